timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_layers.py -x -q -k "gno" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for v in 0 1 0; do
  if [ $v = 1 ]; then export ATHENA_MP_GNO_SERIAL_TILES=1; else unset ATHENA_MP_GNO_SERIAL_TILES; fi
  echo "serial [$v]"; timeout 900 python3 scripts/bench_configs.py --config c4 --reps 3 --no-cpu 2>/dev/null | cut -c1-120
done
