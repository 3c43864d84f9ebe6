#!/bin/bash
# Collects the round's profile evidence on the GPU box into gpurun_out/prof_$TAG/ (copied to profiles/ by hand).
#   bash scripts/profile_round.sh r02a
# rocprofv3 always gets the program itself after `--` (python3 ...), counters in passes of their own.
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# 1. the bench line, un-profiled
timeout 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
# 2. kernel trace of the same command
rm -rf /tmp/kt && (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $OLDPWD/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_profiled.json 2> $OUT/kt.err)
find /tmp/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_kernel_stats.csv
# 3. HBM traffic counters, one pass each
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  D=/tmp/pmc_$(echo $C | tr ' ' '_'); rm -rf $D
  (cd /tmp && timeout 900 rocprofv3 --pmc $C --output-format csv -d $D -- python3 $OLDPWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>> $OUT/pmc.err)
  python3 scripts/pmc_summarise.py $D agg_gemm >> $OUT/pmc_summary.txt
done
# 4. the 256-wide shard step (C5 width) under the kernel trace
rm -rf /tmp/kt2 && (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python3 $OLDPWD/scripts/gpu_f256.py > $OUT/f256.txt 2>> $OUT/kt.err)
find /tmp/kt2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/f256_kernel_stats.csv
# 5. secondary configurations
timeout 900 python3 scripts/bench_configs.py --config c3 > $OUT/configs_c3_c4.jsonl 2>> $OUT/bench.err
timeout 1200 python3 scripts/bench_configs.py --config c4 --reps 3 >> $OUT/configs_c3_c4.jsonl 2>> $OUT/bench.err
# 6. the headline step driven from Fortran
timeout 300 ./athena_amd/fortran/bench_kipf_layer > $OUT/fortran_bench.txt 2>&1
timeout 300 ./athena_amd/fortran/bench_kipf_layer >> $OUT/fortran_bench.txt 2>&1
ls -la $OUT
