timeout 900 python -m pytest tests -m gpu -x -q -k "duvenaud or c3 or fuzz or chemical" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for v in "" nofold ""; do
  if [ -n "$v" ]; then export ATHENA_MP_LIB=$PWD/variants/libathena_mp_$v.so; else unset ATHENA_MP_LIB; fi
  echo "variant [$v]"; timeout 600 python3 scripts/bench_configs.py --config c3 --no-cpu 2>/dev/null | cut -c1-260
done
