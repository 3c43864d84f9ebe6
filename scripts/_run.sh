timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_dist.py -x -q -k "fused or reverse_pass or multi_rank" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
timeout 600 python scripts/gpu_f256.py 2>&1 | grep -E "fused|step"
ATHENA_MP_NO_BUFFER_LOADS=1 timeout 600 python scripts/gpu_f256.py 2>&1 | grep -E "fused|step"
