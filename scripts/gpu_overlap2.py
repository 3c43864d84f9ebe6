"""dev aid: fused backward || dW on two streams"""
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
P = ops.kipf_propagate(g, xd); dW = torch.empty(F * F, device=dev); dX = torch.empty((N, F), device=dev)
side = torch.cuda.Stream()
def seq():
    ops.matmul_dw(P, dzd, out=dW); ops.kipf_layer_bwd_x(g, dzd, wd, F, out=dX)
def conc():
    cur = torch.cuda.current_stream(); side.wait_stream(cur)
    with torch.cuda.stream(side): ops.matmul_dw(P, dzd, out=dW)
    ops.kipf_layer_bwd_x(g, dzd, wd, F, out=dX)
    cur.wait_stream(side)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
print("dW then fused bwd : %.3f ms" % timeit(seq))
print("dW || fused bwd   : %.3f ms" % timeit(conc))
