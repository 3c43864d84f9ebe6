"""Secondary measurements (not the bench.py contract): BASELINE configs[2] (Duvenaud, QM9-shaped batch)
and configs[3] (GNO on a 3-D radius graph).  Prints per-op HIP-event times and fwd+bwd entries/s.
The CPU leg (`cpu_baseline`, skipped with --no-cpu) times the oracle beside the GPU exactly as bench.py's
cpu_baseline leg does for configs[1]: the oracle is the thing COMPARED WITH here, never part of the measured path."""
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
ap.add_argument("--only", default="", help="c4: time only this leg (fwd | bwd_x | bwd_theta)")
ap.add_argument("--no-cpu", action="store_true", help="skip the CPU oracle leg (for rocprofv3 runs)")
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
    t["update_sigmoid"] = timeit(lambda: ops.duvenaud_update_act(g, a_, W, mn, mx, Fv, act="sigmoid"), a.reps)
    t["readout"] = timeit(lambda: ops.duvenaud_readout(R, z, seg, O), a.reps)
    t["update_sigmoid_readout_p(fused)"] = timeit(lambda: ops.duvenaud_update_act_readout(g, a_, W, mn, mx, Fv, R, O, act="sigmoid"), a.reps)
    p_f = ops.duvenaud_update_act_readout(g, a_, W, mn, mx, Fv, R, O, act="sigmoid")[1]
    t["segment_sum"] = timeit(lambda: ops.segment_sum(p_f, seg), a.reps)
    t["readout_bwd"] = timeit(lambda: ops.duvenaud_readout_bwd(R, z, p, seg, gout, act="sigmoid"), a.reps)   # the last time step's form
    t["readout_bwd(+dz_next)"] = timeit(lambda: ops.duvenaud_readout_bwd(R, z, p, seg, gout, act="sigmoid", dz_next=dz), a.reps)   # steps t < T
    t["update_bwd_w"] = timeit(lambda: ops.duvenaud_update_bwd_w(g, dc, a_, mn, mx), a.reps)
    t["update_bwd_a"] = timeit(lambda: ops.duvenaud_update_bwd_a(g, dc, W, mn, mx, Fv + Fe), a.reps)
    t["update_bwd_fused(w+a)"] = timeit(lambda: ops.duvenaud_update_bwd(g, dc, a_, W, mn, mx), a.reps)
    # round 5: the readout's reverse and the update's reverse as ONE launch (what the layer mirrors run; dc never in HBM)
    t["readout_bwd+update_bwd(one launch)"] = timeit(lambda: ops.duvenaud_readout_update_bwd(g, R, z, p, seg, gout, a_, W, mn, mx, Fv, act="sigmoid"), a.reps)
    t["readout_bwd+update_bwd(one launch, +dz_next)"] = timeit(lambda: ops.duvenaud_readout_update_bwd(g, R, z, p, seg, gout, a_, W, mn, mx, Fv, act="sigmoid", dz_next=dz), a.reps)
    # round 5: a and da kept split -- the vertex parts alone, gathered from LDS on this banded batch (csr_gather_banded64)
    a_x = ops.neighbour_sum(g, x); da_x = da[:, :Fv].contiguous()
    t["propagate(vertex part)"] = timeit(lambda: ops.neighbour_sum(g, x, out=a_x), a.reps)
    t["propagate_bwd_x(split da)"] = timeit(lambda: ops.duvenaud_propagate_bwd_x(g, da_x, Fv), a.reps)
    t["propagate_bwd_x"] = timeit(lambda: ops.duvenaud_propagate_bwd_x(g, da, Fv), a.reps)
    t["propagate_bwd_e"] = timeit(lambda: ops.duvenaud_propagate_bwd_e(g, da, Fv), a.reps)
    # the step as the layer mirrors run it: the two update partials in ONE launch (the separate launches stay listed)
    tot = sum(v for k, v in t.items() if k not in ("update_bwd_w", "update_bwd_a", "update_sigmoid", "readout", "readout_bwd(+dz_next)", "readout_bwd",
                                                   "update_bwd_fused(w+a)", "readout_bwd+update_bwd(one launch, +dz_next)",
                                                   "propagate(vertex part)", "propagate_bwd_x(split da)"))
    tot_mid = tot - t["readout_bwd+update_bwd(one launch)"] + t["readout_bwd+update_bwd(one launch, +dz_next)"]   # a step that also receives the next step's dz
    Fc = Fv + Fe
    # algorithmic bytes: gather kernels per entry, dense/elementwise ops = tensors read + written once
    alg = {"propagate": nnz * (4 * Fv + 4 * Fe + 8) + N * (4 * Fc + 4),
           "propagate_bwd_x": nnz * (4 * Fv + 4) + N * (4 * Fv + 4),
           "update_sigmoid": N * 4 * (Fc + Fv), "update_bwd_a": N * 4 * (Fc + Fv), "update_bwd_w": N * 4 * (Fc + Fv),
           "update_bwd_fused(w+a)": N * 4 * (2 * Fc + Fv), "update_sigmoid_readout_p(fused)": N * 4 * (Fc + Fv + O),
           "readout": N * 4 * (Fv + O) + S * 4 * O, "readout_bwd": N * 4 * (2 * Fv + O + 1) + S * 4 * O}
    # A batch of ~18-vertex molecules is block-diagonal: the rows a vertex gathers sit in the same few cache lines as
    # its own, so the per-entry model counts bytes that never leave L2.  The roofline of the two gather kernels is
    # therefore taken on COMPULSORY bytes (every tensor read or written once, indices included); the per-entry figure
    # stays beside it for comparison with configs[1] ("say so rather than claiming > 100 %").
    comp = dict(alg)
    comp["propagate"] = N * 4 * Fv + E * 4 * Fe + nnz * 8 + N * 4 + N * 4 * Fc
    comp["propagate_bwd_x"] = N * 4 * Fc + nnz * 4 + N * 4 + N * 4 * Fv
    roof = {k: {"GBps": round(comp[k] / (t[k] * 1e-3) / 1e9, 1), "frac_of_8TBps": round(comp[k] / (t[k] * 1e-3) / 8e12, 3),
                "bytes": "compulsory" if comp[k] != alg[k] else "algorithmic = compulsory (streaming op)",
                **({"per_entry_model_GBps": round(alg[k] / (t[k] * 1e-3) / 1e9, 1)} if comp[k] != alg[k] else {})} for k in alg}
    if a.no_cpu:
        print(json.dumps({"ms": {k: round(v, 4) for k, v in t.items()}, "total_ms": tot, "total_ms_step_below_the_last": tot_mid, "roofline": roof})); sys.exit(0)
    # CPU oracle (1 thread), same ops, same order
    from oracle import oracle as o
    NG = 130000 if S >= 130000 else S; ns = int(voff[NG]); es = int(ja[1, : ia[ns] - 1].max()); ias = ia[: ns + 1]; jas = np.asfortranarray(ja[:, : ia[ns] - 1])
    xs, es_, Wh, Rh = x[:ns].cpu().numpy(), e[:es].cpu().numpy(), W.cpu().numpy(), R.cpu().numpy()
    segs = voff[:NG + 1]; gouts = gout[:NG].cpu().numpy()
    t0 = time.perf_counter()
    a_h = o.duvenaud_propagate(xs, es_, ias, jas); c_h = o.duvenaud_update(a_h, Wh, ias, mn, mx, Fv); z_h = o.activation("sigmoid", c_h)
    p_h = o.softmax_cols(o.matmul(Rh, z_h, O)); out_h = o.segment_sum(p_h, segs)
    dl_h = o.softmax_cols_bwd(p_h, np.repeat(gouts, np.diff(segs), axis=0)); o.matmul_dw(dl_h, z_h); dz_h = o.matmul_dx(Rh, dl_h, Fv)
    dc_h = o.activation_bwd("sigmoid", z_h, dz_h); o.duvenaud_update_bwd_w(dc_h, a_h, ias, mn, mx)
    da_h = o.duvenaud_update_bwd_a(dc_h, Wh, ias, mn, mx, Fc); o.duvenaud_propagate_bwd_x(da_h, Fv, ias, jas); o.duvenaud_propagate_bwd_e(da_h, Fv, es, ias, jas)
    tc = time.perf_counter() - t0
    ents = int(ias[-1] - 1)
    res = {"config": "C3 Duvenaud one time step fwd+bwd", "graphs": S, "vertices": N, "entries": nnz, "F_v": Fv, "F_e": Fe,
           "ms": {k: round(v, 4) for k, v in t.items()}, "total_ms": tot, "total_ms_step_below_the_last": tot_mid,
           "entries_per_s": nnz / tot * 1e3, "roofline": roof,
           "cpu_baseline": {"value": ents / tc, "unit": "entries/s", "cores": 1, "kind": "port",
                            "sample": f"oracle, {NG} graphs = {ents} entries, {tc:.2f} s"}}
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
    legs = {"fwd": lambda: ops.gno_aggregate(g, theta, co, x, d, H, Fo),
            "bwd_x": lambda: ops.gno_aggregate_bwd_x(g, theta, co, gup, d, H, Fi),
            "bwd_theta": lambda: ops.gno_aggregate_bwd_theta(g, theta, co, x, gup, d, H)}
    # training-mode pair: the forward pass keeps S (33 GB here) and the reverse pass's S^T g streams it
    s_keep = [None]
    def fwd_save():
        m, s_keep[0] = ops.gno_aggregate_save(g, theta, co, x, d, H, Fo, s_save=s_keep[0])
        return m
    legs["fwd_save"] = fwd_save
    legs["bwd_theta_saved"] = lambda: ops.gno_aggregate_bwd_theta(g, theta, co, x, gup, d, H, s_save=s_keep[0])
    if a.only:
        if a.only == "bwd_theta_saved": fwd_save()
        print(json.dumps({"only": a.only, "lib": os.environ.get("ATHENA_MP_LIB", ""), "ms": round(timeit(legs[a.only], a.reps), 3)})); sys.exit(0)
    for k, fn in legs.items(): t[k] = timeit(fn, a.reps)
    t_keep = {"fwd_save": t.pop("fwd_save"), "bwd_theta_saved": t.pop("bwd_theta_saved")}
    tot = sum(t.values())
    tot_keep = t_keep["fwd_save"] + t["bwd_x"] + t_keep["bwd_theta_saved"]
    keep = {"ms": {k: round(v, 3) for k, v in t_keep.items()}, "total_ms": round(tot_keep, 3),
            "s_save_GB": round(ops.gno_saved_bytes(g, d, H, Fi, Fo) / 1e9, 2)}
    R = (H + 1) * Fi
    # re-associated algorithm, fused (S stays on chip): the contraction N*2*Fo*(H+1)*Fi bounds it on the fp32 matrix pipe
    alg_fwd = nnz * (4 * Fi + 4 * d + 8) + N * (4 * Fo + 4)
    flops_fwd = nnz * 2 * H * (Fi + d) + N * 2 * Fo * R
    # dx: the same two pieces over the transposed CSR.  dtheta: outer product + S^T g, and G = g Vmat^T + per-entry dh
    flops = {"fwd": flops_fwd, "bwd_x": nnz * 2 * H * (Fo + d) + N * 2 * Fi * (H + 1) * Fo,
             "bwd_theta": nnz * 2 * H * (Fi + d) + N * 2 * Fo * R + N * 2 * Fo * H * Fi + nnz * 2 * H * (Fi + d + 1)}
    roof = {k: {"bound": "mfma", "TFLOP": round(flops[k] / 1e12, 3), "TFLOPs": round(flops[k] / (t[k] * 1e-3) / 1e12, 1),
                "frac_of_157_TFLOPs_fp32_mfma": round(flops[k] / (t[k] * 1e-3) / 157e12, 3)} for k in t}
    roof["fwd"].update({"algorithmic_GB": round(alg_fwd / 1e9, 1), "GBps": round(alg_fwd / (t["fwd"] * 1e-3) / 1e9, 1)})
    if a.no_cpu:
        print(json.dumps({"ms": {k: round(v, 3) for k, v in t.items()}, "total_ms": tot, "keep_s": keep, "roofline": roof})); sys.exit(0)
    # CPU: the reference's MATERIALISING algorithm is infeasible at this size (246 GB kernel tensor); time it
    # with the oracle on the first 20 000 vertices of the same graph
    from oracle import oracle as o
    ns = 5000
    sia, sja, cs = synth.radius_graph(ns, seed=9)
    rs = np.random.default_rng(5)
    th = theta.cpu().numpy(); xs = rs.uniform(-1, 1, (ns, Fi)).astype(np.float32); gs = rs.uniform(-1, 1, (ns, Fo)).astype(np.float32)
    t0 = time.perf_counter()
    kap = o.gno_kernel_eval(cs, th, H, Fo * Fi); o.gno_aggregate(xs, kap, sia, sja, Fo)
    o.gno_aggregate_bwd_x(gs, kap, sia, sja, Fi); dk = o.gno_aggregate_bwd_k(gs, xs, cs.shape[0], sia, sja); o.gno_kernel_bwd_theta(cs, th, dk, H)
    tc = time.perf_counter() - t0
    res = {"config": "C4 GNO aggregate fwd + dx + dtheta", "vertices": N, "entries": nnz, "edge_columns": E, "F": Fi, "H": H,
           "synthetic_graph_generation_host_s": tg, "ms": {k: round(v, 3) for k, v in t.items()}, "total_ms": tot, "entries_per_s": nnz / tot * 1e3,
           "keep_s": keep, "roofline": roof,
           "cpu_baseline": {"value": sja.shape[1] / tc, "unit": "entries/s", "cores": 1, "kind": "port",
                            "sample": f"materialising oracle (the reference's algorithm), radius graph of {ns} vertices = {sja.shape[1]} entries / {cs.shape[0]} edge columns, same widths, {tc:.1f} s"}}
print(json.dumps(res))
