"""Quick on-GPU sanity + timing of the Kipf path against the oracle (dev aid, not a test)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import DeviceGraph, ops, synth
from oracle import oracle

def cmp(name, a, b):
    a = a.cpu().numpy() if hasattr(a, "cpu") else a
    err = np.abs(a - b).max(); sc = np.abs(b).max()
    print(f"{name}: max|d|={err:.3e} scale={sc:.3e} rel={err/sc:.3e} bitexact={np.array_equal(a,b)}")

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ia, ja = synth.random_graph_csr(N, int(4.5 * N))
x, w, dz = synth.kipf_inputs(N, F)
g = DeviceGraph(ia, ja)
dev = torch.device("cuda:0")
xd, wd, dzd = [torch.from_numpy(t).to(dev) for t in (x, w, dz)]
nnz = ja.shape[1]
check = N <= 200000
y = ops.kipf_propagate(g, xd)
if check: cmp("kipf fwd", y, oracle.kipf_propagate(x, ia, ja))
z = ops.matmul(wd, y, F)
if check: cmp("gemm fwd", z, oracle.matmul(w, y.cpu().numpy(), F))
dw = ops.matmul_dw(y, dzd)
if check: cmp("gemm dw", dw, oracle.matmul_dw(dz, y.cpu().numpy()))
dp = ops.matmul_dx(wd, dzd, F)
if check: cmp("gemm dx", dp, oracle.matmul_dx(w, dz, F))
dx = ops.kipf_propagate_bwd(g, dp)
if check: cmp("kipf bwd", dx, oracle.kipf_propagate_bwd(dp.cpu().numpy(), ia, ja))
dxe = ops.kipf_propagate_bwd(g, dp, exact=True)
if check: cmp("kipf bwd exact", dxe, oracle.kipf_propagate_bwd(dp.cpu().numpy(), ia, ja, exact=True))
t = {}
t["agg_fwd"] = timeit(lambda: ops.kipf_propagate(g, xd, out=y))
t["gemm_fwd"] = timeit(lambda: ops.matmul(wd, y, F, out=z))
t["gemm_dw"] = timeit(lambda: ops.matmul_dw(y, dzd, out=dw))
t["gemm_dx"] = timeit(lambda: ops.matmul_dx(wd, dzd, F, out=dp))
t["agg_bwd"] = timeit(lambda: ops.kipf_propagate_bwd(g, dp, out=dx))
bytes_fwd = nnz * (4 * F + 8) + N * (4 * F + 8)
for k, v in t.items(): print(f"{k}: {v:.4f} ms")
print(f"agg fwd algorithmic GB/s: {bytes_fwd / t['agg_fwd'] / 1e6:.1f}  ({bytes_fwd/t['agg_fwd']/1e6/8000*100:.1f}% of 8 TB/s)")
print(f"gemm fwd TFLOP/s: {2*N*F*F/t['gemm_fwd']/1e9:.1f}")
tot = sum(t.values()); print(f"total {tot:.3f} ms -> {nnz/tot/1e6:.2f} G edges/s")
