"""dev aid: time the dense kernels at C2 (1M x 128 x 128)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import ops, synth, _capi
_capi.init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda:0")
x, w, dz = synth.kipf_inputs(N, F)
xd, wd, dzd = [torch.from_numpy(t).to(dev) for t in (x, w, dz)]
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
Z = torch.empty((N, F), device=dev); dW = torch.empty(F * F, device=dev)
t = timeit(lambda: ops.matmul(wd, xd, F, out=Z)); print(f"gemm_fwd {t:.4f} ms {2*N*F*F/t/1e9:.1f} TF")
t = timeit(lambda: ops.matmul_dx(wd, dzd, F, out=Z)); print(f"gemm_dx  {t:.4f} ms {2*N*F*F/t/1e9:.1f} TF")
t = timeit(lambda: ops.matmul_dw(xd, dzd, out=dW)); print(f"gemm_dw  {t:.4f} ms {2*N*F*F/t/1e9:.1f} TF")
ref = (xd.double().T @ dzd.double()).float().reshape(-1)
print("dw rel err", ((dW - ref).abs().max() / ref.abs().max()).item())
