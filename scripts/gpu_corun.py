"""dev aid: does the fused kernel co-run gracefully with a copy-like kernel on another stream
(a stand-in for RCCL send/recv kernels during the halo exchange)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import DeviceGraph, ops, synth
N, F = 640000, 128
dev = torch.device("cuda:0")
ia, ja = synth.random_graph_csr(N, int(4.5 * N))
x, w, dz = synth.kipf_inputs(N, F)
g = DeviceGraph(ia, ja, n_edge_cols=0)
xd, wd = torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev)
P = torch.empty((N, F), device=dev); Z = torch.empty((N, F), device=dev)
src = torch.empty(225_000_000 // 4, device=dev); dst = torch.empty_like(src)
side = torch.cuda.Stream()
def fused(): ops.kipf_layer_fwd(g, xd, wd, F, P=P, Z=Z)
def copy():
    with torch.cuda.stream(side):
        for _ in range(4): dst.copy_(src)      # ~4 x 0.08 ms of copy kernels
def both():
    cur = torch.cuda.current_stream(); side.wait_stream(cur)
    copy(); fused(); cur.wait_stream(side)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
def copy_only():
    cur = torch.cuda.current_stream(); side.wait_stream(cur); copy(); cur.wait_stream(side)
print("fused alone   : %.3f ms" % timeit(fused))
print("copies alone  : %.3f ms" % timeit(copy_only))
print("fused || copy : %.3f ms" % timeit(both))
