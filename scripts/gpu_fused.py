"""dev aid: fused Kipf layer kernels vs the unfused sequence (correctness + time) at C2"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import DeviceGraph, ops, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
F = 128
dev = torch.device("cuda:0")
ia, ja = synth.random_graph_csr(N, int(4.5 * N))
x, w, dz = synth.kipf_inputs(N, F)
g = DeviceGraph(ia, ja, n_edge_cols=0)
xd, wd, dzd = [torch.from_numpy(t).to(dev) for t in (x, w, dz)]
P0 = ops.kipf_propagate(g, xd); Z0 = ops.matmul(wd, P0, F, act="relu")
P1, Z1 = ops.kipf_layer_fwd(g, xd, wd, F, act="relu")
torch.cuda.synchronize()
print("P bit-exact:", torch.equal(P0, P1), " Z max rel:", ((Z0 - Z1).abs().max() / Z0.abs().max()).item())
dP = ops.matmul_dx(wd, dzd, F); dX0 = ops.kipf_propagate_bwd(g, dP)
dX1 = ops.kipf_layer_bwd_x(g, dzd, wd, F)
print("dX max rel:", ((dX0 - dX1).abs().max() / dX0.abs().max()).item())
dXe0 = ops.kipf_propagate_bwd(g, dP, exact=True); dXe1 = ops.kipf_layer_bwd_x(g, dzd, wd, F, exact=True)
print("dX exact max rel:", ((dXe0 - dXe1).abs().max() / dXe0.abs().max()).item())
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
Z = torch.empty_like(Z0); P = torch.empty_like(P0); dX = torch.empty_like(dX0); dPb = torch.empty_like(dP)
print("unfused fwd (agg+gemm): %.3f ms" % timeit(lambda: (ops.kipf_propagate(g, xd, out=P), ops.matmul(wd, P, F, out=Z))))
print("fused   fwd           : %.3f ms" % timeit(lambda: ops.kipf_layer_fwd(g, xd, wd, F, P=P, Z=Z)))
print("unfused bwd (gemm+agg): %.3f ms" % timeit(lambda: (ops.matmul_dx(wd, dzd, F, out=dPb), ops.kipf_propagate_bwd(g, dPb, out=dX))))
print("fused   bwd           : %.3f ms" % timeit(lambda: ops.kipf_layer_bwd_x(g, dzd, wd, F, out=dX)))
