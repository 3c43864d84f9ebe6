#!/bin/bash
# Round 4's profile evidence in one call on the GPU box -> gpurun_out/prof_$TAG/ (copied to profiles/r04_* by hand):
#   bench line (with the secondary block), kernel trace + PMC traffic of the headline step, kernel trace + PMC traffic of
#   configs[3] (the reverse pass from one contraction), the Fortran driver's step.
# rocprofv3 always gets the program itself after `--` (python3 ...), counters in passes of their own.
TAG=${1:-r04}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
timeout 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
rm -rf /tmp/kt && (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench_profiled.json 2> $OUT/kt.err)
find /tmp/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_kernel_stats.csv
: > $OUT/pmc_summary.txt
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  D=/tmp/pmc_$(echo $C | tr ' ' '_'); rm -rf $D
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d $D -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>> $OUT/pmc.err)
  python3 scripts/pmc_summarise.py $D agg_gemm >> $OUT/pmc_summary.txt
done
# configs[3]
timeout 600 python3 scripts/bench_secondary.py --config c4 > $OUT/c4_config.json 2>> $OUT/bench.err
timeout 600 python3 scripts/bench_secondary.py --config c4 --mesh-order cells >> $OUT/c4_config.json 2>> $OUT/bench.err
rm -rf /tmp/kt4 && (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt4 -- python3 $R/scripts/bench_secondary.py --config c4 --reps 3 --timing-only > /dev/null 2>> $OUT/kt.err)
find /tmp/kt4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/c4_gno_kernel_stats.csv
: > $OUT/c4_gno_pmc_traffic.txt
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
  D=/tmp/pmc4_$(echo $C | tr ' ' '_'); rm -rf $D
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d $D -- python3 $R/scripts/bench_secondary.py --config c4 --reps 1 --timing-only > /dev/null 2>> $OUT/pmc.err)
  python3 scripts/pmc_summarise.py $D gno_ >> $OUT/c4_gno_pmc_traffic.txt
done
timeout 300 ./athena_amd/fortran/bench_kipf_layer > $OUT/fortran_bench.txt 2>&1
ls -la $OUT
