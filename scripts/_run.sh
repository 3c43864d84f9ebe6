for v in "" un3 nopipe; do
  if [ -n "$v" ]; then export ATHENA_MP_LIB=$PWD/variants/libathena_mp_$v.so; else unset ATHENA_MP_LIB; fi
  echo "variant [$v]"; timeout 900 python3 scripts/bench_configs.py --config c4 --reps 3 --no-cpu 2>/dev/null | cut -c1-120
done
