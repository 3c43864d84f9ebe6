#!/bin/bash
# start / end of gno_dh_pc_kernel, gno_stg_kernel<true> and gno_px_gather_kernel inside the last reverse passes of configs[3]
# (kernel trace): do the gather and S^T g run side by side?   ATHENA_MP_LIB=<variant> selects another library build.
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/gather_tl
rm -rf /tmp/kt_tl
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_tl -- python3 $R/scripts/bench_secondary.py --config c4 --reps 2 --timing-only > /dev/null 2> $R/gpurun_out/gather_tl/err.txt)
f=$(find /tmp/kt_tl -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<PY | tee $R/gpurun_out/gather_tl/timeline.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gno_stg_kernel<true>" in r["Kernel_Name"] or "gno_px_gather" in r["Kernel_Name"] or "gno_dh_pc" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-8:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print("  %-60s start %8.3f ms  end %8.3f ms" % (r["Kernel_Name"][:60], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6))
PY
