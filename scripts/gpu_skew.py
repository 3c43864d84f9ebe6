"""dev aid: fused vs unfused on a heavy-tailed (power-law) graph"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import DeviceGraph, ops
N, F = 1000000, 128
rng = np.random.default_rng(0)
deg = np.minimum((rng.pareto(1.5, N) * 4).astype(np.int64) + 1, 20000)
deg = (deg * (10_000_000 / deg.sum())).astype(np.int64) + 1
ia = np.concatenate([[1], 1 + np.cumsum(deg)]).astype(np.int32)
nnz = int(deg.sum())
ja = np.zeros((2, nnz), np.int32, order="F"); ja[0] = rng.integers(1, N + 1, nnz)
print("nnz", nnz, "max deg", deg.max(), "p99", np.percentile(deg, 99))
dev = torch.device("cuda:0")
g = DeviceGraph(ia, ja, n_edge_cols=0)
x = torch.from_numpy(rng.uniform(-1, 1, (N, F)).astype(np.float32)).to(dev)
w = torch.from_numpy((rng.standard_normal(F * F) * 0.1).astype(np.float32)).to(dev)
P = torch.empty((N, F), device=dev); Z = torch.empty((N, F), device=dev)
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
print("unfused fwd: %.3f ms" % timeit(lambda: (ops.kipf_propagate(g, x, out=P), ops.matmul(w, P, F, out=Z))))
print("fused   fwd: %.3f ms" % timeit(lambda: ops.kipf_layer_fwd(g, x, w, F, P=P, Z=Z)))
print("agg only   : %.3f ms" % timeit(lambda: ops.kipf_propagate(g, x, out=P)))
