"""C5-shaped shard on one GPU: Kipf layer fwd+bwd at F = 256: the two-kernel pieces, then the fused step (W in registers)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import DeviceGraph, ops, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1250000
F = 256
dev = torch.device("cuda:0")
ia, ja = synth.random_graph_csr(N, 7 * N)
g = DeviceGraph(ia, ja, n_edge_cols=0)
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.uniform(-1, 1, (N, F)).astype(np.float32)).to(dev)
w = torch.from_numpy((rng.standard_normal(F * F) * 0.08).astype(np.float32)).to(dev)
dz = torch.from_numpy(rng.uniform(-1, 1, (N, F)).astype(np.float32)).to(dev)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
P = ops.kipf_propagate(g, x)
print("entries", ja.shape[1])
print("agg fwd   %.3f ms" % t(lambda: ops.kipf_propagate(g, x)))
print("gemm fwd  %.3f ms  (%.1f TFLOP/s)" % ((tt := t(lambda: ops.matmul(w, P, F))), 2 * N * F * F / tt / 1e9))
print("gemm dx   %.3f ms" % t(lambda: ops.matmul_dx(w, dz, F)))
print("gemm dw   %.3f ms" % t(lambda: ops.matmul_dw(P, dz)))
print("agg bwd   %.3f ms" % t(lambda: ops.kipf_propagate_bwd(g, dz)))
def step():
    p, z = ops.kipf_layer_fwd(g, x, w, F); ops.matmul_dw(p, dz); ops.kipf_layer_bwd_x(g, dz, w, F)
print("fused fwd %.3f ms" % t(lambda: ops.kipf_layer_fwd(g, x, w, F)))
print("fused bwd %.3f ms" % t(lambda: ops.kipf_layer_bwd_x(g, dz, w, F)))
ts = t(step)
nnz = ja.shape[1]
print("step      %.3f ms  (%.2f G edges/s; fused fwd bytes %.1f GB)" % (ts, nnz / ts / 1e6, (nnz * (4 * F + 8) + N * (4 * F + 8) + N * 4 * F) / 1e9))
