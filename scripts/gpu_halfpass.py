"""dev aid: does gathering one 64-feature half at a time (256 MB table, fits the Infinity Cache) beat one 128-feature pass?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import DeviceGraph, ops, synth
N, F = 1000000, 128
dev = torch.device("cuda:0")
ia, ja = synth.random_graph_csr(N, int(4.5 * N))
g = DeviceGraph(ia, ja, n_edge_cols=0)
x = torch.rand((N, F), device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
out = torch.empty((N, F), device=dev)
print("full 128-wide pull       : %.3f ms" % timeit(lambda: ops.kipf_propagate_bwd(g, x, out=out)))
print("first 64 columns (ld 128): %.3f ms" % timeit(lambda: ops.duvenaud_propagate_bwd_x(g, x, 64)))
x64 = torch.rand((N, 64), device=dev); o64 = torch.empty((N, 64), device=dev)
print("64-wide contiguous table : %.3f ms" % timeit(lambda: ops.kipf_propagate_bwd(g, x64, out=o64)))
