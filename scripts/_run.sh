mkdir -p gpurun_out/r2d
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r2d/pytest.txt
timeout 600 python scripts/gpu_f256.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2d/f256.txt
timeout 300 python bench.py 2>gpurun_out/r2d/bench.err | tee gpurun_out/r2d/bench1.json | cut -c1-600
