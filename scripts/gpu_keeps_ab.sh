#!/bin/bash
# A/B of how gno_pc_kernel<true> writes the S it keeps (variants/libathena_mp_ks_*.so, scripts/build_variants.sh)
cd "$(dirname "$0")/.."
for leg in fwd_save fwd_save fwd; do
  for tag in "" ks_plain ks_buf19 ks_buf2 ks_buf17 ks_nostore; do
    if [ -z "$tag" ]; then lib=""; else lib=$PWD/variants/libathena_mp_$tag.so; [ -f "$lib" ] || continue; fi
    [ "$leg" = fwd ] && [ -n "$tag" ] && continue
    if [ -z "$lib" ]; then unset ATHENA_MP_LIB; else export ATHENA_MP_LIB=$lib; fi
    timeout 300 python scripts/bench_configs.py --config c4 --only $leg --reps 5 --no-cpu 2>&1 | tail -1
  done
done
