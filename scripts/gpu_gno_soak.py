"""Soak of the producer / consumer GNO kernels: radius graphs from 50 to 300 000 vertices, every entry point launched
REPS times; every launch must reproduce the first one bit for bit (the barrier protocol, the slab reductions and the id
queues are deterministic by construction) and stay finite.  Run under `timeout`: a lost barrier would hang."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import ops, synth
from athena_amd.graph import DeviceGraph

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(a).to(dev)
Fi = Fo = H = 64
tot = 0
t0 = time.time()
for N, d in [(50, 3), (777, 2), (4099, 3), (33000, 1), (300000, 3)]:
    ia, ja, coords3 = synth.radius_graph(N)
    coords = np.ascontiguousarray(coords3[:, :d])
    rng = np.random.default_rng(N)
    g = DeviceGraph(ia, ja, n_edge_cols=coords.shape[0])
    x = T(rng.uniform(-1, 1, (N, Fi)).astype(np.float32)); co = T(coords)
    theta = T((0.3 * rng.standard_normal(H * d + H + Fo * Fi * H + Fo * Fi)).astype(np.float32))
    gup = T(rng.uniform(-1, 1, (N, Fo)).astype(np.float32))
    legs = {"fwd": lambda: ops.gno_aggregate(g, theta, co, x, d, H, Fo),
            "dx": lambda: ops.gno_aggregate_bwd_x(g, theta, co, gup, d, H, Fi),
            "dtheta": lambda: ops.gno_aggregate_bwd_theta(g, theta, co, x, gup, d, H),
            "dcoords": lambda: ops.gno_aggregate_bwd_coords(g, theta, co, x, gup, d, H)}
    # the training-mode pair: the forward pass keeps S (buffer reused, poisoned in between), the reverse pass streams it
    keep = [None]
    def fwd_save():
        if keep[0] is not None: keep[0].fill_(float("nan"))
        m, keep[0] = ops.gno_aggregate_save(g, theta, co, x, d, H, Fo, s_save=keep[0])
        return m
    legs["fwd_save"] = fwd_save
    legs["dtheta_saved"] = lambda: ops.gno_aggregate_bwd_theta(g, theta, co, x, gup, d, H, s_save=keep[0])
    # round 4: the reverse pass from ONE contraction (per-entry partials of dx through HBM + gather); its three outputs as one
    # tensor so that every launch is compared in full
    def bwd_one():
        dx1, dth1, dc1, fused = ops.gno_aggregate_bwd(g, theta, co, x, gup, d, H, s_save=keep[0], need_dcoords=True)
        assert fused
        return torch.cat([dx1.reshape(-1), dth1.reshape(-1), dc1.reshape(-1)])
    legs["bwd_one_contraction"] = bwd_one
    for name, fn in legs.items():
        first = fn()
        assert torch.isfinite(first).all(), (N, name)
        for _ in range(REPS - 1):
            assert torch.equal(fn(), first), (N, name, "differs between launches")
        tot += REPS
    assert torch.equal(legs["fwd_save"](), legs["fwd"]()) and torch.equal(legs["dtheta_saved"](), legs["dtheta"]())
    one = legs["bwd_one_contraction"]()
    dx_sep = legs["dx"]().reshape(-1)
    assert (one[:dx_sep.numel()] - dx_sep).abs().max().item() <= 1e-5 * dx_sep.abs().max().item(), (N, "one-contraction dx vs the dx launch")
    assert torch.equal(one[dx_sep.numel():dx_sep.numel() + theta.numel()], legs["dtheta"]()), (N, "dtheta bits")
    print(f"N={N} d={d}: {len(legs) * REPS} launches identical", flush=True)
print(f"soak ok: {tot} launches in {time.time() - t0:.1f} s")
