"""dev aid: do an HBM-bound aggregation and an MFMA-bound GEMM overlap on two streams?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import DeviceGraph, ops, synth
N, F = 1000000, 128
dev = torch.device("cuda:0")
ia, ja = synth.random_graph_csr(N, int(4.5 * N))
x, w, dz = synth.kipf_inputs(N, F)
g = DeviceGraph(ia, ja, n_edge_cols=0)
xd, wd, dzd = [torch.from_numpy(t).to(dev) for t in (x, w, dz)]
P = torch.empty((N, F), device=dev); Z = torch.empty((N, F), device=dev); dW = torch.empty(F*F, device=dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def seq():
    ops.kipf_propagate(g, xd, out=P); ops.matmul(wd, dzd, F, out=Z); ops.matmul_dw(xd, dzd, out=dW)
def conc():
    cur = torch.cuda.current_stream()
    sA.wait_stream(cur); sB.wait_stream(cur)
    with torch.cuda.stream(sA): ops.kipf_propagate(g, xd, out=P)
    with torch.cuda.stream(sB): ops.matmul(wd, dzd, F, out=Z); ops.matmul_dw(xd, dzd, out=dW)
    cur.wait_stream(sA); cur.wait_stream(sB)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
print("sequential agg+gemm+dw: %.3f ms" % timeit(seq))
print("concurrent agg || (gemm, dw): %.3f ms" % timeit(conc))
for env in ("ATHENA_MP_GEMM_4WAVE",):
    pass
