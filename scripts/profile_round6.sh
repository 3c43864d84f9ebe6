#!/bin/bash
OUT=$PWD/gpurun_out/r06pmc; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
: > $OUT/pmc_summary.txt
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  D=/tmp/pmc_$(echo $C | tr ' ' '_'); rm -rf $D
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d $D -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>> $OUT/pmc.err)
  python3 scripts/pmc_summarise.py $D agg_gemm >> $OUT/pmc_summary.txt
done
timeout 900 python3 scripts/bench_secondary.py --config c3 > $OUT/c3_secondary.json 2>> $OUT/err.txt
rm -rf /tmp/kt3 && (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -- python3 $R/scripts/bench_secondary.py --config c3 --reps 3 --timing-only > /dev/null 2>> $OUT/err.txt)
find /tmp/kt3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/c3_kernel_stats.csv
tail -5 $OUT/pmc_summary.txt
