for v in 0 1 0 1; do
  if [ $v = 1 ]; then export ATHENA_MP_NO_LENGTH_ORDER=1; else unset ATHENA_MP_NO_LENGTH_ORDER; fi
  echo "noorder [$v]"; timeout 300 python3 bench.py --no-cpu-baseline --steps 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
unset ATHENA_MP_NO_LENGTH_ORDER
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_dist.py -x -q -k "fused or reverse_pass or multi_rank" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
