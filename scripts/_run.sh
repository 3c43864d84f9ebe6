mkdir -p gpurun_out/r2e
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -60 > gpurun_out/r2e/pytest.txt
tail -60 gpurun_out/r2e/pytest.txt
