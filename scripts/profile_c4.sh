#!/bin/bash
# C4 (GNO) evidence: timings, kernel trace, SQ counters of the three fused kernels -> gpurun_out/prof_c4_$TAG/
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_c4_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python3 scripts/bench_configs.py --config c4 --reps 3 > $OUT/config_c4.jsonl 2> $OUT/err.txt
rm -rf /tmp/ktc4 && (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktc4 -- python3 $OLDPWD/scripts/bench_configs.py --config c4 --reps 3 --no-cpu > $OUT/config_c4_profiled.json 2>> $OUT/err.txt)
find /tmp/ktc4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/c4_kernel_stats.csv
: > $OUT/c4_pmc.txt
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  D=/tmp/pmc_c4; rm -rf $D
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d $D -- python3 $OLDPWD/scripts/bench_configs.py --config c4 --no-cpu --reps 1 > /dev/null 2>> $OUT/err.txt)
  python3 scripts/pmc_summarise.py $D gno_ >> $OUT/c4_pmc.txt
done
ls -la $OUT
