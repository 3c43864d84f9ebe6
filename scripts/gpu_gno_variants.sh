#!/bin/bash
# Times one leg of C4 (scripts/bench_configs.py --only) with the stock library and every variants/libathena_mp_<tag>.so named.
#   scripts/gpu_gno_variants.sh <leg> <tag> [<tag> ...]     -> gpurun_out/gno_variants.txt
leg=$1; shift
mkdir -p gpurun_out
out=gpurun_out/gno_variants.txt
: > $out
python3 scripts/bench_configs.py --config c4 --no-cpu --reps 3 --only $leg >> $out 2>&1
for t in "$@"; do
  ATHENA_MP_LIB=$PWD/variants/libathena_mp_$t.so python3 scripts/bench_configs.py --config c4 --no-cpu --reps 3 --only $leg 2>&1 | grep '^{' >> $out
done
cat $out
