set -x
mkdir -p gpurun_out/r2a
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2a/pytest.txt
timeout 300 python bench.py > gpurun_out/r2a/bench1.json 2> gpurun_out/r2a/bench1.err
ATHENA_MP_BENCH_ONE_DEVICE=1 ATHENA_MP_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 4 --steps 5 --warmup 2 > gpurun_out/r2a/bench4_dry.json 2> gpurun_out/r2a/bench4_dry.err
timeout 120 python scripts/gpu_rccl_probe.py 2 > gpurun_out/r2a/rccl_probe2.txt 2>&1
timeout 120 python scripts/gpu_rccl_probe.py 1 > gpurun_out/r2a/rccl_probe1.txt 2>&1
tail -3 gpurun_out/r2a/*.txt gpurun_out/r2a/*.err
cat gpurun_out/r2a/bench1.json gpurun_out/r2a/bench4_dry.json
