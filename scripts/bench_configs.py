"""Secondary measurements (not the bench.py contract): BASELINE configs[2] (Duvenaud, QM9-shaped batch)
and configs[3] (GNO on a 3-D radius graph).  Prints per-op HIP-event times and fwd+bwd entries/s."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import DeviceGraph, ops, synth, _capi

def timeit(fn, n=5, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c3")
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
_capi.init(0)
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
res = {}
if a.config == "c3":
    S = int(130000 * a.scale)
    ia, ja, voff, E = synth.molecule_batch(S)
    N, nnz = ia.size - 1, ja.shape[1]
    Fv, Fe, O, mn, mx = 64, 8, 10, 1, 10
    D = mx - mn + 1
    g = DeviceGraph(ia, ja, n_edge_cols=E)
    x, e = T(rng.random((N, Fv), np.float32)), T(rng.random((E, Fe), np.float32))
    W = T(rng.standard_normal(Fv * (Fv + Fe) * D).astype(np.float32) * 0.1)
    R = T(rng.standard_normal(O * Fv).astype(np.float32) * 0.1)
    seg = T(voff)
    gout = T(rng.standard_normal((S, O)).astype(np.float32))
    a_ = ops.duvenaud_propagate(g, x, e); c = ops.duvenaud_update(g, a_, W, mn, mx, Fv); z = ops.activation("sigmoid", c)
    lg = ops.matmul(R, z, O); p, out = ops.softmax_segsum(lg, seg)
    dl = ops.softmax_segsum_bwd(p, seg, gout); dz = ops.matmul_dx(R, dl, Fv); dc = ops.activation_bwd("sigmoid", z, dz)
    da = ops.duvenaud_update_bwd_a(g, dc, W, mn, mx, Fv + Fe)
    t = {}
    t["propagate"] = timeit(lambda: ops.duvenaud_propagate(g, x, e, out=a_), a.reps)
    t["update"] = timeit(lambda: ops.duvenaud_update(g, a_, W, mn, mx, Fv), a.reps)
    t["sigmoid"] = timeit(lambda: ops.activation("sigmoid", c, out=z), a.reps)
    t["readout_gemm"] = timeit(lambda: ops.matmul(R, z, O), a.reps)
    t["softmax_segsum"] = timeit(lambda: ops.softmax_segsum(lg, seg), a.reps)
    t["softmax_segsum_bwd"] = timeit(lambda: ops.softmax_segsum_bwd(p, seg, gout), a.reps)
    t["readout_dw"] = timeit(lambda: ops.matmul_dw(z, dl), a.reps)
    t["readout_dx"] = timeit(lambda: ops.matmul_dx(R, dl, Fv), a.reps)
    t["sigmoid_bwd"] = timeit(lambda: ops.activation_bwd("sigmoid", z, dz), a.reps)
    t["update_bwd_w"] = timeit(lambda: ops.duvenaud_update_bwd_w(g, dc, a_, mn, mx), a.reps)
    t["update_bwd_a"] = timeit(lambda: ops.duvenaud_update_bwd_a(g, dc, W, mn, mx, Fv + Fe), a.reps)
    t["propagate_bwd_x"] = timeit(lambda: ops.duvenaud_propagate_bwd_x(g, da, Fv), a.reps)
    t["propagate_bwd_e"] = timeit(lambda: ops.duvenaud_propagate_bwd_e(g, da, Fv), a.reps)
    tot = sum(t.values())
    res = {"config": "C3 Duvenaud one time step fwd+bwd", "graphs": S, "vertices": N, "entries": nnz, "F_v": Fv, "F_e": Fe,
           "ms": {k: round(v, 4) for k, v in t.items()}, "total_ms": tot, "entries_per_s": nnz / tot * 1e3}
else:
    N = int(2_000_000 * a.scale)
    t0 = time.time(); ia, ja, coords = synth.radius_graph(N); tg = time.time() - t0
    nnz, E = ja.shape[1], coords.shape[0]
    Fi = Fo = H = 64; d = 3
    g = DeviceGraph(ia, ja, n_edge_cols=E)
    x = T(rng.uniform(-1, 1, (N, Fi)).astype(np.float32)); co = T(coords)
    theta = T((0.3 * rng.standard_normal(H * d + H + Fo * Fi * H + Fo * Fi)).astype(np.float32))
    gup = T(rng.uniform(-1, 1, (N, Fo)).astype(np.float32))
    t = {}
    t["fwd"] = timeit(lambda: ops.gno_aggregate(g, theta, co, x, d, H, Fo), a.reps)
    t["bwd_x"] = timeit(lambda: ops.gno_aggregate_bwd_x(g, theta, co, gup, d, H, Fi), a.reps)
    t["bwd_theta"] = timeit(lambda: ops.gno_aggregate_bwd_theta(g, theta, co, x, gup, d, H), a.reps)
    tot = sum(t.values())
    res = {"config": "C4 GNO aggregate fwd + dx + dtheta", "vertices": N, "entries": nnz, "edge_columns": E, "F": Fi, "H": H,
           "graph_build_s": tg, "ms": {k: round(v, 3) for k, v in t.items()}, "total_ms": tot, "entries_per_s": nnz / tot * 1e3}
print(json.dumps(res))
