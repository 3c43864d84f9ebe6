"""dev aid: is the fused step capturable into a hipGraph (torch.cuda.CUDAGraph)?"""
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
P = torch.empty((N, F), device=dev); Z = torch.empty((N, F), device=dev); dW = torch.empty(F*F, device=dev); dX = torch.empty((N, F), device=dev)
def step():
    ops.kipf_layer_fwd(g, xd, wd, F, P=P, Z=Z); ops.matmul_dw(P, dzd, out=dW); ops.kipf_layer_bwd_x(g, dzd, wd, F, out=dX)
for _ in range(3): step()
torch.cuda.synchronize()
ref = (Z.clone(), dW.clone(), dX.clone())
gr = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
    torch.cuda.synchronize()
    with torch.cuda.graph(gr, stream=s):
        step()
torch.cuda.current_stream().wait_stream(s)
Z.zero_(); dW.zero_(); dX.zero_()
gr.replay(); torch.cuda.synchronize()
print("replay matches eager:", torch.equal(Z, ref[0]), torch.equal(dW, ref[1]), torch.equal(dX, ref[2]))
def timeit(fn, n=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print("eager  step: %.4f ms" % timeit(step))
print("graph  step: %.4f ms" % timeit(gr.replay))
