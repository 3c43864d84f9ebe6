"""soak: thousands of launches of the persistent ticket-scheduled fused kernels on graphs of many sizes (incl. fewer
chunks than workgroups, empty graphs, one-row graphs), results checked against the two-kernel route each time"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import DeviceGraph, ops, synth
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
t0 = time.time(); launches = 0
for trial in range(60):
    F = int(rng.choice([64, 128]))
    n = int(rng.choice([1, 2, 31, 32, 33, 500, 4097, 20000, 100000]))
    pairs = int(n * rng.choice([0, 1, 5, 20]))
    ia, ja = synth.random_graph_csr(n, pairs, seed=trial) if n > 1 else (np.array([1, 2], np.int32), np.array([[1], [0]], np.int32, order="F"))
    g = DeviceGraph(ia, ja, n_edge_cols=0)
    x = torch.rand((n, F), device=dev) - 0.5; w = (torch.rand(F * F, device=dev) - 0.5) * 0.2; dz = torch.rand((n, F), device=dev) - 0.5
    pr = ops.kipf_propagate(g, x); zr = ops.matmul(w, pr, F)
    dxr = ops.kipf_propagate_bwd(g, ops.matmul_dx(w, dz, F))
    for rep in range(50):
        p, z = ops.kipf_layer_fwd(g, x, w, F)
        dx = ops.kipf_layer_bwd_x(g, dz, w, F)
        launches += 2
    torch.cuda.synchronize()
    assert torch.equal(p, pr), (trial, n, F)
    assert (z - zr).abs().max() <= 1e-5 * max(zr.abs().max().item(), 1e-30), (trial, n, F)
    assert (dx - dxr).abs().max() <= 1e-5 * max(dxr.abs().max().item(), 1e-30), (trial, n, F)
    g.close()
# the headline size, many steps back to back
ia, ja = synth.random_graph_csr(1000000, 4500000)
g = DeviceGraph(ia, ja, n_edge_cols=0)
x = torch.rand((1000000, 128), device=dev); w = torch.rand(128 * 128, device=dev) * 0.1; dz = torch.rand((1000000, 128), device=dev)
for _ in range(1500):
    ops.kipf_layer_fwd(g, x, w, 128); ops.kipf_layer_bwd_x(g, dz, w, 128); launches += 2
torch.cuda.synchronize()
print("soak ok: %d fused launches in %.1f s" % (launches, time.time() - t0))
