for v in "" nostag; do
  if [ -n "$v" ]; then export ATHENA_MP_LIB=$PWD/variants/libathena_mp_$v.so; else unset ATHENA_MP_LIB; fi
  echo "variant [$v]"; timeout 300 python3 scripts/_t.py 2>/dev/null
done
unset ATHENA_MP_LIB
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -x -q -k "reverse_pass" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
