"""Sizing experiment (timing only): the two independent launches of the Kipf reverse pass -- dW = dZ . P^T (matrix pipe + 1 GB of
reads) and dX = (A^T dZ) W (HBM bound) -- on two streams against one after the other (configs[1])."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import ops, synth
from athena_amd.graph import DeviceGraph

dev = torch.device("cuda:0")
N, F = 1_000_000, 128
ia, ja = synth.random_graph_csr(N, 4_500_000)
x, w, dz = synth.kipf_inputs(N, F)
g = DeviceGraph(ia, ja, n_edge_cols=0, device=0)
xd, wd, dzd = (torch.from_numpy(t).to(dev) for t in (x, w, dz))
P, Z = torch.empty((N, F), device=dev), torch.empty((N, F), device=dev)
dW, dX = torch.empty(F * F, device=dev), torch.empty((N, F), device=dev)
ops.kipf_layer_fwd(g, xd, wd, F, P=P, Z=Z)
sb = torch.cuda.Stream(device=dev)


def serial():
    ops.kipf_layer_fwd(g, xd, wd, F, P=P, Z=Z)
    ops.matmul_dw(P, dzd, out=dW)
    ops.kipf_layer_bwd_x(g, dzd, wd, F, out=dX)


def two(first_dw):
    ops.kipf_layer_fwd(g, xd, wd, F, P=P, Z=Z)
    cur = torch.cuda.current_stream(dev)
    sb.wait_stream(cur)
    if first_dw:
        with torch.cuda.stream(sb):
            ops.matmul_dw(P, dzd, out=dW)
        ops.kipf_layer_bwd_x(g, dzd, wd, F, out=dX)
    else:
        with torch.cuda.stream(sb):
            ops.kipf_layer_bwd_x(g, dzd, wd, F, out=dX)
        ops.matmul_dw(P, dzd, out=dW)
    cur.wait_stream(sb)


def timed(fn, reps=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for i in range(3):
    print(f"one stream {timed(serial):.4f} ms   dW on the side stream {timed(lambda: two(True)):.4f} ms   pull on the side stream {timed(lambda: two(False)):.4f} ms")
