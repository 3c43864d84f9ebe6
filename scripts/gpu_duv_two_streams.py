"""Sizing experiment (timing only): does a configs[2] step get faster when the batch is cut in two halves whose launch chains run
on two streams -- propagate of one half beside update of the other?  The library's workspaces are shared by all calls of a device,
so the RESULTS of the two-stream run are not valid; only the elapsed time is read."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import ops, synth
from athena_amd.graph import DeviceGraph

dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
Fv, Fe, O, mn, mx = 64, 8, 10, 1, 10
Fc, D = Fv + Fe, 10
rng = np.random.default_rng(0)
W = T(rng.standard_normal(Fv * Fc * D).astype(np.float32) * 0.1)
R = T(rng.standard_normal(O * Fv).astype(np.float32) * 0.1)


def make(S, seed):
    ia, ja, voff, E = synth.molecule_batch(S) if seed == 0 else synth.molecule_batch(S)
    N = ia.size - 1
    g = DeviceGraph(ia, ja, n_edge_cols=E)
    r = np.random.default_rng(seed)
    d = dict(g=g, x=T(r.random((N, Fv), np.float32)), e=T(r.random((E, Fe), np.float32)), seg=T(voff),
             gout=T(r.standard_normal((S, O)).astype(np.float32)), N=N)
    d["a"] = ops.duvenaud_propagate(g, d["x"], d["e"])
    return d


def chain(d):
    g = d["g"]
    ops.duvenaud_propagate(g, d["x"], d["e"], out=d["a"])
    z, p = ops.duvenaud_update_act_readout(g, d["a"], W, mn, mx, Fv, R, O, act="sigmoid")
    ops.segment_sum(p, d["seg"])
    dc, dR = ops.duvenaud_readout_bwd(R, z, p, d["seg"], d["gout"], act="sigmoid")
    da, dW = ops.duvenaud_update_bwd(g, dc, d["a"], W, mn, mx)
    ops.duvenaud_propagate_bwd_x(g, da, Fv)
    ops.duvenaud_propagate_bwd_e(g, da, Fv)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


whole = make(130_000, 0)
h0, h1 = make(65_000, 1), make(65_000, 2)
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)


def two_streams():
    cur = torch.cuda.current_stream(dev)
    sa.wait_stream(cur); sb.wait_stream(cur)
    with torch.cuda.stream(sa):
        chain(h0)
    with torch.cuda.stream(sb):
        chain(h1)
    cur.wait_stream(sa); cur.wait_stream(sb)


for i in range(2):
    print(f"whole batch, one chain back to back      {timed(lambda: chain(whole)):.3f} ms")
    print(f"two halves, one stream                    {timed(lambda: (chain(h0), chain(h1))):.3f} ms")
    print(f"two halves, two streams (timing only)     {timed(two_streams):.3f} ms")
