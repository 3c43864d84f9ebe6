#!/usr/bin/env python3
"""The secondary BASELINE configurations as ONE json object each, for bench.py's `secondary` block (N = 1 only):

    python scripts/bench_secondary.py --config c3     configs[2]: Duvenaud, 130 k QM9-shaped graphs, F_v = 64, F_e = 8
    python scripts/bench_secondary.py --config c4     configs[3]: GNO layer step on a 2 M-point radius mesh, 64 features
    python scripts/bench_secondary.py --config c5     configs[4] on ONE GPU: Kipf 10 M / 150 M / 256 (the 8-GPU config's
                                                      whole graph on one device: 50 GB of resident tensors)
    python scripts/bench_secondary.py --config kb     the Kipf layer on a block-diagonal batch of 130 k molecule-sized graphs (64 and
                                                      128 features): the LDS-staged gather with the coefficient (round 6)

Each prints {"config", "workload", "step_ms", "ops": {name: {"ms", "bound", "frac", ...}}, "parity": {"ok", ...}}:
per-op times from HIP events (median of --reps launches after one warm-up), the roofline fraction of every op against
the bound SURVEY.md 8d names for it (HBM 8 TB/s on algorithmic bytes; fp32 MFMA 157.3 TFLOP/s for the GNO contractions),
and a parity flag of the SAME device results against the CPU oracle on a sample (block-diagonal batches: the first
graphs are an exact sub-problem; meshes and the random graph: sampled rows / columns as compact sub-problems).
bench.py runs this as a CHILD process after its timed loop, so that a failure here is `secondary.error` in the line and
never touches the headline value or the exit code.  The oracle is the checker here, never the thing measured."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from athena_amd import DeviceGraph, _capi, ops, synth

HBM, MFMA = 8000.0, 157.3
TOL = 1e-5
TIMING_ONLY = False
MESH_ORDER = None


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return float(np.median(ts))


def rel(a, b):
    b = np.asarray(b, np.float64)
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


def ew_worst(a, b, scale):
    """max over elements of |a - b| / scale, scale = the sum of the magnitudes of each element's own terms (oracle on |operands|)"""
    a, b, scale = (np.asarray(t, np.float64) for t in (a, b, scale))
    return float((np.abs(a - b) / np.maximum(scale, 1e-30)).max()) if a.size else 0.0


def gno_kappa_magnitude(o, coords, th, d, H, F):
    """sum_k |V[f,k]| h_k + |b_v[f]| per edge column: the kernel MLP with |V|, |b_v| (its hidden h = relu(U dx + b_u) >= 0 as is)"""
    t2 = np.array(th, np.float32, copy=True)
    offV = H * d + H
    t2[offV:] = np.abs(t2[offV:])
    return o.gno_kernel_eval(coords, t2, H, F)


def hbm_op(ms, nbytes, note=None):
    gbs = nbytes / (ms * 1e-3) / 1e9
    out = {"ms": round(ms, 4), "bound": "hbm", "GBps": round(gbs, 1), "frac": round(gbs / HBM, 3), "bytes": int(nbytes)}
    if note:
        out["bytes_model"] = note
    return out


def mfma_op(ms, flops):
    tf = flops / (ms * 1e-3) / 1e12
    return {"ms": round(ms, 3), "bound": "mfma", "TFLOPs": round(tf, 1), "frac": round(tf / MFMA, 3), "TFLOP": round(flops / 1e12, 3)}


def run_c3(dev, reps):
    """one Duvenaud time step as the layer mirror runs it (update_message_duvenaud + update_readout_duvenaud,
    athena_duvenaud_msgpass_layer.f90:755-859, and its reverse pass): 6 launches"""
    from oracle import oracle as o

    S = 130_000
    ia, ja, voff, E = synth.molecule_batch(S)
    N, nnz = ia.size - 1, ja.shape[1]
    Fv, Fe, O, mn, mx = 64, 8, 10, 1, 10
    Fc, D = Fv + Fe, 10
    rng = np.random.default_rng(0)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    g = DeviceGraph(ia, ja, n_edge_cols=E)
    x, e = T(rng.random((N, Fv), np.float32)), T(rng.random((E, Fe), np.float32))
    W = T(rng.standard_normal(Fv * Fc * D).astype(np.float32) * 0.1)
    R = T(rng.standard_normal(O * Fv).astype(np.float32) * 0.1)
    seg = T(voff)
    gout = T(rng.standard_normal((S, O)).astype(np.float32))
    a_ = ops.duvenaud_propagate(g, x, e)
    z, p = ops.duvenaud_update_act_readout(g, a_, W, mn, mx, Fv, R, O, act="sigmoid")
    out = ops.segment_sum(p, seg)
    dc, dR = ops.duvenaud_readout_bwd(R, z, p, seg, gout, act="sigmoid")
    # da split where it is written (da_x [N, 64] + da_e [N, 8], round 5): what both layer mirrors run at F_v = 64
    da_x, da_e, dW = ops.duvenaud_update_bwd_split(g, dc, a_, W, mn, mx, Fv)
    da = torch.cat([da_x, da_e], dim=1)                          # the packed form, for the parity block only
    dx = ops.duvenaud_propagate_bwd_x(g, da_x, Fv)
    de = ops.duvenaud_propagate_bwd_e(g, da_e, 0)
    # the two reverse launches as ONE (round 5: dc never in HBM) -- what both layer mirrors run; the two above feed the parity block
    # a kept SPLIT (round 5; what both layer mirrors run at these widths): a_x [N, 64] gathered per time step, a_e [N, 8] once per layer
    a_x, a_e = ops.neighbour_sum(g, x), ops.duvenaud_propagate_edges(g, e)
    split_bits = bool(torch.equal(a_x, a_[:, :Fv]) and torch.equal(a_e, a_[:, Fv:]))
    z_s, p_s = ops.duvenaud_update_act_readout_split(g, a_x, a_e, W, mn, mx, Fv, R, O, act="sigmoid")
    split_bits = bool(split_bits and torch.equal(z_s, z) and torch.equal(p_s, p))
    del z_s, p_s
    one = lambda: ops.duvenaud_readout_update_bwd(g, R, z, p, seg, gout, a_x, W, mn, mx, Fv, act="sigmoid", a_e=a_e)
    o_dax, o_dae, o_dW, o_dR = one()
    one_vs_two = {"split_a_bit_identical_to_packed": split_bits, "da_bit_identical": bool(torch.equal(o_dax, da_x) and torch.equal(o_dae, da_e)),
                  "dW_rel": rel(o_dW.cpu().numpy(), dW.cpu().numpy()), "dR_rel": rel(o_dR.cpu().numpy(), dR.cpu().numpy())}
    del o_dax, o_dae, o_dW, o_dR
    t_two = {"readout_bwd": timeit(lambda: ops.duvenaud_readout_bwd(R, z, p, seg, gout, act="sigmoid"), reps),
             "update_bwd_fused(w+a)": timeit(lambda: ops.duvenaud_update_bwd_split(g, dc, a_, W, mn, mx, Fv), reps)}
    t_packed = {"propagate": timeit(lambda: ops.duvenaud_propagate(g, x, e, out=a_), reps),
                "update_sigmoid_readout_p(fused)": timeit(lambda: ops.duvenaud_update_act_readout(g, a_, W, mn, mx, Fv, R, O, act="sigmoid"), reps)}
    t_edge_once = timeit(lambda: ops.duvenaud_propagate_edges(g, e, out=a_e), reps)
    t = {"propagate": timeit(lambda: ops.neighbour_sum(g, x, out=a_x), reps),
         "update_sigmoid_readout_p(fused)": timeit(lambda: ops.duvenaud_update_act_readout_split(g, a_x, a_e, W, mn, mx, Fv, R, O, act="sigmoid"), reps),
         "segment_sum": timeit(lambda: ops.segment_sum(p, seg), reps),
         "readout_bwd+update_bwd(one launch)": timeit(one, reps),
         "propagate_bwd_x": timeit(lambda: ops.duvenaud_propagate_bwd_x(g, da_x, Fv), reps),
         "propagate_bwd_e": timeit(lambda: ops.duvenaud_propagate_bwd_e(g, da_e, 0), reps)}
    # a batch of ~18-vertex molecules is block-diagonal: a vertex's neighbours sit in the cache lines next to its own, so the
    # two gathers are priced on COMPULSORY bytes (every tensor once, indices included), the streaming ops on their tensors
    comp = {"propagate": N * 4 * Fv + nnz * 4 + N * 4 + N * 4 * Fv,      # the vertex part alone (the edge part: once per layer, below)
            "update_sigmoid_readout_p(fused)": N * 4 * (Fc + Fv + O), "segment_sum": N * 4 * O + S * 4 * O,
            "readout_bwd+update_bwd(one launch)": N * 4 * (Fv + O + 1 + 2 * Fc) + S * 4 * O,
            "propagate_bwd_x": N * 4 * Fc + nnz * 4 + N * 4 + N * 4 * Fv, "propagate_bwd_e": N * 4 * Fe + nnz * 8 + E * 4 * Fe}
    opsd = {k: hbm_op(t[k], comp[k], "compulsory" if k.startswith("propagate") else "tensors read + written once") for k in t}
    # the two update launches are bound by the fp32 matrix pipe, not by their bytes (timing-only builds without any HBM traffic keep
    # 0.48 of 0.54 ms, profiles/r05_c3_bwd_probe.txt): both fractions are reported, "bound" names the one that holds
    for k, fl in (("update_sigmoid_readout_p(fused)", 2.0 * N * Fv * (Fc + O)),
                  ("readout_bwd+update_bwd(one launch)", 4.0 * N * Fv * Fc + 4.0 * N * Fv * O)):
        m = mfma_op(t[k], fl)
        opsd[k]["hbm_frac"] = opsd[k].pop("frac")
        opsd[k].update({"bound": "mfma", "frac": m["frac"], "TFLOPs": m["TFLOPs"], "TFLOP": m["TFLOP"],
                        "flops_model": "2 x vertices x outputs x inputs per product, unpadded (the 16-wide fragments pad 72 to 80)"})
    # parity: the first 2 000 graphs are an exact sub-problem of every op (dW / dR sum over all vertices: a float64 device sum)
    NG = 2000
    nv = int(voff[NG]); ias = ia[:nv + 1]; jas = np.asfortranarray(ja[:, :ia[nv] - 1]); ne = int(jas[1].max())
    Wh, Rh = W.cpu().numpy(), R.cpu().numpy()
    a_h = o.duvenaud_propagate(x[:nv].cpu().numpy(), e[:ne].cpu().numpy(), ias, jas)
    z_h = o.activation("sigmoid", o.duvenaud_update(a_h, Wh, ias, mn, mx, Fv))
    p_h = o.softmax_cols(o.matmul(Rh, z_h, O))
    out_h = o.segment_sum(p_h, voff[:NG + 1])
    dl_h = o.softmax_cols_bwd(p_h, np.repeat(gout[:NG].cpu().numpy(), np.diff(voff[:NG + 1]), axis=0))
    dc_h = o.activation_bwd("sigmoid", z_h, o.matmul_dx(Rh, dl_h, Fv))
    da_h = o.duvenaud_update_bwd_a(dc_h, Wh, ias, mn, mx, Fc)
    dx_h = o.duvenaud_propagate_bwd_x(da_h, Fv, ias, jas)
    de_h = o.duvenaud_propagate_bwd_e(da_h, Fv, ne, ias, jas)
    eids = np.unique(jas[1][jas[1] > 0]).astype(np.int64) - 1
    par = {"against": f"oracle on the first {NG} graphs ({nv} vertices) of the same batch",
           "propagate_bit_exact": bool(np.array_equal(a_[:nv].cpu().numpy(), a_h)),
           "z_rel": rel(z[:nv].cpu().numpy(), z_h), "readout_rel": rel(out[:NG].cpu().numpy(), out_h),
           "dc_rel": rel(dc[:nv].cpu().numpy(), dc_h), "da_rel": rel(da[:nv].cpu().numpy(), da_h),
           "dx_rel": rel(dx[:nv].cpu().numpy(), dx_h), "de_rel": rel(de[torch.from_numpy(eids).to(dev)].cpu().numpy(), de_h[eids]), "tol": TOL}
    # element-wise against the magnitude of each element's own terms: the update c = W_d (a / d) behind the
    # sigmoid (|dz| <= |dc| / 4) and its reverse da = (dc W_d) / d
    mag_c = o.duvenaud_update(np.abs(a_h), np.abs(Wh), ias, mn, mx, Fv)
    par["z_elementwise_worst"] = ew_worst(z[:nv].cpu().numpy(), z_h, 0.25 * mag_c + np.abs(z_h))
    par["da_elementwise_worst"] = ew_worst(da[:nv].cpu().numpy(), da_h, o.duvenaud_update_bwd_a(np.abs(dc_h), np.abs(Wh), ias, mn, mx, Fc))
    par["ok"] = bool(par["propagate_bit_exact"] and all(par[k] <= TOL for k in ("z_rel", "readout_rel", "dc_rel", "da_rel", "dx_rel", "de_rel",
                                                                                 "z_elementwise_worst", "da_elementwise_worst")))
    step = sum(t.values())
    opsd["readout_bwd+update_bwd(one launch)"]["as_two_launches_ms"] = {k: round(v, 4) for k, v in t_two.items()}
    opsd["propagate"]["packed_with_the_edge_part_ms"] = round(t_packed["propagate"], 4)
    opsd["propagate"]["edge_part_alone_once_per_layer_ms"] = round(t_edge_once, 4)
    opsd["update_sigmoid_readout_p(fused)"]["packed_a_ms"] = round(t_packed["update_sigmoid_readout_p(fused)"], 4)
    opsd["readout_bwd+update_bwd(one launch)"]["against_the_two_launches"] = one_vs_two
    par["one_launch_reverse"] = one_vs_two
    par["ok"] = bool(par["ok"] and one_vs_two["da_bit_identical"] and one_vs_two["dW_rel"] <= TOL and one_vs_two["dR_rel"] <= TOL)
    # the CPU path beside it: the same ops, same order, the oracle on one thread, first 20 000 graphs
    NC = 20000
    nvc = int(voff[NC]); iac = ia[:nvc + 1]; jac = np.asfortranarray(ja[:, :ia[nvc] - 1]); nec = int(jac[1].max())
    xc, ec, gc = x[:nvc].cpu().numpy(), e[:nec].cpu().numpy(), gout[:NC].cpu().numpy()
    t0 = time.perf_counter()
    a_c = o.duvenaud_propagate(xc, ec, iac, jac); z_c = o.activation("sigmoid", o.duvenaud_update(a_c, Wh, iac, mn, mx, Fv))
    p_c = o.softmax_cols(o.matmul(Rh, z_c, O)); o.segment_sum(p_c, voff[:NC + 1])
    dl_c = o.softmax_cols_bwd(p_c, np.repeat(gc, np.diff(voff[:NC + 1]), axis=0)); o.matmul_dw(dl_c, z_c)
    dc_c = o.activation_bwd("sigmoid", z_c, o.matmul_dx(Rh, dl_c, Fv)); o.duvenaud_update_bwd_w(dc_c, a_c, iac, mn, mx)
    da_c = o.duvenaud_update_bwd_a(dc_c, Wh, iac, mn, mx, Fc); o.duvenaud_propagate_bwd_x(da_c, Fv, iac, jac); o.duvenaud_propagate_bwd_e(da_c, Fv, nec, iac, jac)
    tc = time.perf_counter() - t0
    entc = int(iac[-1] - 1)
    cpu = {"value": entc / tc, "unit": "entries/s", "cores": 1, "kind": "port",
           "sample": f"oracle, first {NC} graphs = {entc} entries, one time step fwd+bwd, {tc:.2f} s"}
    res = {"config": "configs[2]", "workload": f"Duvenaud msgpass, {S} QM9-shaped graphs = {N} vertices / {nnz} entries, F_v = {Fv}, F_e = {Fe}, "
           f"{O} outputs, one time step + readout, fwd+bwd (a kept split: the edge part of a, gathered once per layer, is listed beside the step)", "step_ms": round(step, 4), "entries_per_s": nnz / step * 1e3, "ops": opsd, "parity": par,
           "cpu_baseline": cpu}
    del a_, z, p, dc, da, da_x, da_e, dx, de, a_x, a_e
    res["layer_T4"] = c3_layer_T4(dev, reps, ia, ja, voff, E, x, e)
    return res


def c3_layer_T4(dev, reps, ia, ja, voff, E, x, e):
    """BASELINE configs[2] as the config states it (T = 4 + readout, example/msgpass_chemical/src/main.f90:129-138):
    duvenaud_msgpass_layer_type(F_v 64, F_e 8, 4 time steps, degrees 1..10, 10 outputs), forward + the whole reverse pass
    (input, edge-feature and all 8 parameter gradients) through the layer mirror.  Parity of the layer output, dx, de and EVERY
    parameter gradient against the oracle's per-sample composition (oracle/layers.py) on the first 2 000 graphs, run through
    the same layer code; the full-size run is tied to it by the graph-local outputs of those graphs and by additivity of
    the parameter gradients over chunks of the batch (float64 sum of 65 chunk runs against the one full-size run)."""
    from athena_amd.graph import graph_type
    from athena_amd.layers import duvenaud_msgpass_layer_type
    from oracle import layers as ol

    S, N, nnz = voff.size - 1, ia.size - 1, ja.shape[1]
    Fv, Fe, O, T, mn, mx = 64, 8, 10, 4, 1, 10
    rng = np.random.default_rng(7)

    def make():
        return duvenaud_msgpass_layer_type(num_vertex_features=[Fv], num_edge_features=[Fe], num_time_steps=T, max_vertex_degree=mx,
                                           num_outputs=O, min_vertex_degree=mn, seed=1)

    layer = make()
    g = graph_type.from_csr(ia, ja, num_edges=E).freeze()
    layer.set_graph_batched(g, voff)
    up = torch.from_numpy(rng.standard_normal((S, O)).astype(np.float32)).to(dev)

    def step():
        layer.forward(x, e)
        return layer.backward(up, need_input_grad=True, need_edge_grad=True)

    ms = timeit(step, reps)
    out_full = layer.output.cpu().numpy().copy()
    dx_full, de_full = (t.cpu().numpy() for t in step())
    grads_full = layer.get_gradients().astype(np.float64)
    params = layer.get_params()
    # bytes one step moves, per time step the six launches of `ops` above (the last step's reverse has no dz_next; steps 2..T of the
    # reverse read it; dc between the readout's and the update's reverse no longer exists): compulsory tensors only
    Fc = Fv + Fe
    # (a kept split: the vertex part of propagate per time step, the edge part and its reverse once per layer)
    per_t = (N * 4 * Fv + nnz * 4 + N * 4 + N * 4 * Fv) + N * 4 * (Fc + Fv + O) + (N * 4 * O + S * 4 * O) \
        + (N * 4 * (Fv + O + 1 + 2 * Fc) + S * 4 * O) + (N * 4 * Fv + nnz * 4 + N * 4 + N * 4 * Fv)
    once = (E * 4 * Fe + nnz * 4 + N * 4 + N * 4 * Fe) + (N * 4 * Fe + nnz * 8 + E * 4 * Fe)
    nbytes = T * per_t + once + (T - 1) * N * 4 * Fv

    # --- the same layer code on the first NG graphs, against the oracle's per-sample composition
    NG = 2000

    def sub_batch(g0, g1):
        v0, v1 = int(voff[g0]), int(voff[g1])
        w0, w1 = int(ia[v0]) - 1, int(ia[v1]) - 1
        js = ja[:, w0:w1].astype(np.int64)
        eids = np.unique(js[1][js[1] > 0])                       # global edge ids (1-based) these graphs use, ascending
        remap = np.zeros(int(eids.max()) + 1 if eids.size else 1, np.int64)
        remap[eids] = np.arange(1, eids.size + 1)
        jl = np.empty_like(js)
        jl[0] = js[0] - v0
        jl[1] = np.where(js[1] > 0, remap[np.minimum(js[1], remap.size - 1)], 0)
        ias = (ia[v0:v1 + 1].astype(np.int64) - w0).astype(np.int32)
        return v0, v1, ias, np.asfortranarray(jl.astype(np.int32)), eids - 1

    v0, v1, ias, jas, eidx = sub_batch(0, NG)
    sub = make()
    sub.set_params(params)
    gsub = graph_type.from_csr(ias, jas, num_edges=int(eidx.size))
    sub.set_graph_batched(gsub, voff[:NG + 1] - voff[0])
    xs, es = x[v0:v1].contiguous(), e[torch.from_numpy(eidx).to(dev)].contiguous()
    sub.forward(xs, es)
    dxs, des = (t.cpu().numpy() for t in sub.backward(up[:NG].contiguous(), need_input_grad=True, need_edge_grad=True))
    out_s, gr_s = sub.output.cpu().numpy(), sub.get_gradients()
    # oracle: the reference's loop over samples (athena_duvenaud_msgpass_layer.f90:792-855) and grad_reverse's walk over it
    graphs, xl, el = [], [], []
    xh, eh, uph = xs.cpu().numpy(), es.cpu().numpy(), up[:NG].cpu().numpy()
    for s in range(NG):
        a0, a1 = int(voff[s]), int(voff[s + 1])
        w0, w1 = int(ias[a0]) - 1, int(ias[a1]) - 1
        jj = jas[:, w0:w1].astype(np.int64)
        ed = np.unique(jj[1][jj[1] > 0])
        rm = np.zeros(int(ed.max()) + 1 if ed.size else 1, np.int64)
        rm[ed] = np.arange(1, ed.size + 1)
        gl = graph_type.from_csr((ias[a0:a1 + 1].astype(np.int64) - w0).astype(np.int32),
                                 np.asfortranarray(np.stack([jj[0] - a0, np.where(jj[1] > 0, rm[np.minimum(jj[1], rm.size - 1)], 0)]).astype(np.int32)),
                                 num_edges=int(ed.size))
        graphs.append(gl); xl.append(xh[a0:a1]); el.append(eh[ed - 1])
    sizes = [Fv * (Fv + Fe) * (mx - mn + 1)] * T + [O * Fv] * T
    plist = np.split(params, np.cumsum(sizes)[:-1])
    nvf = [Fv] * (T + 1)
    t0 = time.perf_counter()
    out_o, tapes = ol.duvenaud_forward(graphs, xl, el, plist, nvf, Fe, mn, mx, O, "sigmoid")
    dx_o, de_o, gr_o = ol.duvenaud_backward(graphs, el, tapes, plist, nvf, Fe, mn, mx, O, "sigmoid", uph)
    t_cpu = time.perf_counter() - t0
    de_o_flat = np.zeros_like(des)
    for s in range(NG):                                           # per-sample de back into the sub-batch's edge columns
        a0, a1 = int(voff[s]), int(voff[s + 1])
        jj = jas[1, int(ias[a0]) - 1:int(ias[a1]) - 1]
        ed = np.unique(jj[jj > 0])
        de_o_flat[ed - 1] = de_o[s]
    gr_o = np.concatenate(gr_o)
    names = [f"dW_{t}" for t in range(1, T + 1)] + [f"dR_{t}" for t in range(1, T + 1)]
    gsplit_s, gsplit_o = np.split(gr_s, np.cumsum(sizes)[:-1]), np.split(gr_o, np.cumsum(sizes)[:-1])
    par = {"against": f"oracle/layers.py (the reference's per-sample loop and grad_reverse's walk) on the first {NG} graphs through the same layer code",
           "output_rel": rel(out_s, out_o), "dx_rel": rel(dxs, np.concatenate(dx_o)), "de_rel": rel(des, de_o_flat),
           "param_grad_rel": {n: rel(a, b) for n, a, b in zip(names, gsplit_s, gsplit_o)}, "tol": TOL}
    # --- the full-size run tied to it
    par["full_size_output_of_those_graphs_rel"] = rel(out_full[:NG], out_o)
    par["full_size_dx_of_those_graphs_rel"] = rel(dx_full[v0:v1], np.concatenate(dx_o))
    acc = np.zeros(grads_full.size, np.float64)
    CH = 2000
    for g0 in range(0, S, CH):
        g1 = min(S, g0 + CH)
        c0, c1, iac, jac, ec = sub_batch(g0, g1)
        sub.set_graph_batched(graph_type.from_csr(iac, jac, num_edges=int(ec.size)), voff[g0:g1 + 1] - voff[g0])
        sub.forward(x[c0:c1].contiguous(), e[torch.from_numpy(ec).to(dev)].contiguous())
        sub.backward(up[g0:g1].contiguous(), need_input_grad=False, need_edge_grad=False)
        acc += sub.get_gradients().astype(np.float64)
    fsplit, asplit = np.split(grads_full, np.cumsum(sizes)[:-1]), np.split(acc, np.cumsum(sizes)[:-1])
    par["full_size_param_grads_vs_sum_of_chunk_runs_rel"] = {n: rel(a, b) for n, a, b in zip(names, fsplit, asplit)}
    par["ok"] = bool(max(par["output_rel"], par["dx_rel"], par["de_rel"], par["full_size_output_of_those_graphs_rel"],
                         par["full_size_dx_of_those_graphs_rel"], *par["param_grad_rel"].values(),
                         *par["full_size_param_grads_vs_sum_of_chunk_runs_rel"].values()) <= TOL)
    ent = int(ias[-1] - 1) * T
    return {"workload": f"duvenaud_msgpass_layer_type, T = {T} time steps + readout, F_v = {Fv}, F_e = {Fe}, degrees {mn}..{mx}, {O} outputs, "
            f"{S} graphs = {N} vertices / {nnz} entries; forward + reverse (dx, de, {2 * T} parameter gradients)",
            "step_ms": round(ms, 4), "entries_per_s": T * nnz / ms * 1e3, "launches_per_step": 6 * T,
            "hbm": hbm_op(ms, nbytes, "compulsory tensors of the 6 T launches"), "parity": par,
            "cpu_baseline": {"value": ent / t_cpu, "unit": "entries/s (entries x time steps)", "cores": 1, "kind": "port",
                             "sample": f"oracle/layers.py, the reference's loop over samples, first {NG} graphs x {T} time steps = {ent} entry visits, "
                                       f"forward + reverse, {t_cpu:.2f} s"}}


def run_c4(dev, reps):
    """graph_nop_layer's aggregation, training mode (S kept): forward, reverse to x, reverse to theta"""
    from oracle import oracle as o

    N = 2_000_000
    ia, ja, coords = synth.radius_graph(N, order=MESH_ORDER)
    nnz, E = ja.shape[1], coords.shape[0]
    Fi = Fo = H = 64; d = 3
    rng = np.random.default_rng(0)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    g = DeviceGraph(ia, ja, n_edge_cols=E)
    x, co = T(rng.uniform(-1, 1, (N, Fi)).astype(np.float32)), T(coords)
    theta = T((0.3 * rng.standard_normal(H * d + H + Fo * Fi * H + Fo * Fi)).astype(np.float32))
    gup = T(rng.uniform(-1, 1, (N, Fo)).astype(np.float32))
    keep = [None]

    def fwd_save():
        m, keep[0] = ops.gno_aggregate_save(g, theta, co, x, d, H, Fo, s_save=keep[0])
        return m

    t = {"fwd_inference": timeit(lambda: ops.gno_aggregate(g, theta, co, x, d, H, Fo), reps),
         "fwd_keeps_S": timeit(fwd_save, reps),
         "bwd_x": timeit(lambda: ops.gno_aggregate_bwd_x(g, theta, co, gup, d, H, Fi), reps),
         "bwd_theta_S_kept": timeit(lambda: ops.gno_aggregate_bwd_theta(g, theta, co, x, gup, d, H, s_save=keep[0]), reps),
         "bwd_x+theta_one_contraction": timeit(lambda: ops.gno_aggregate_bwd(g, theta, co, x, gup, d, H, s_save=keep[0]), reps)}
    R = (H + 1) * Fi
    f_fwd = nnz * 2 * H * (Fi + d) + N * 2 * Fo * R
    flops = {"fwd_inference": f_fwd, "fwd_keeps_S": f_fwd, "bwd_x": nnz * 2 * H * (Fo + d) + N * 2 * Fi * (H + 1) * Fo,
             "bwd_theta_S_kept": N * 2 * Fo * R + N * 2 * Fo * H * Fi + nnz * 2 * H * (Fi + d + 1),
             # S^T g + G = g Vmat^T + per entry: dh = G x_j, the partial h^T G of dx, the h MFMAs
             "bwd_x+theta_one_contraction": N * 2 * Fo * R + N * 2 * Fo * H * Fi + nnz * 2 * H * (2 * Fi + 2 * (d + 1))}
    opsd = {k: mfma_op(t[k], flops[k]) for k in t}
    if TIMING_ONLY:
        return {"config": "configs[3]", "ops": opsd}
    s_bytes = ops.gno_saved_bytes(g, d, H, Fi, Fo)
    alg_fwd = nnz * (4 * Fi + 4 * d + 8) + N * (4 * Fo + 4)
    opsd["fwd_keeps_S"]["hbm"] = hbm_op(t["fwd_keeps_S"], alg_fwd + s_bytes, "algorithmic + the S it writes")
    # parity: 300 sampled rows of m and 300 sampled columns of dx against the MATERIALISING oracle (kappa only for the edge
    # columns they touch); d theta by the adjoint identity <m, g> = <Vaug, dVaug> (a global sum: the oracle is infeasible)
    m = fwd_save()
    th = theta.cpu().numpy()
    rows = np.sort(rng.choice(N, 300, replace=False))
    ent = np.concatenate([np.arange(ia[r] - 1, ia[r + 1] - 1) for r in rows])
    ecols, einv = np.unique(ja[1, ent], return_inverse=True)
    ncols, cinv = np.unique(ja[0, ent], return_inverse=True)
    sia = np.concatenate([[1], 1 + np.cumsum(ia[rows + 1] - ia[rows])]).astype(np.int32)
    sja = np.zeros((2, ent.size), np.int32, order="F"); sja[0] = cinv + 1; sja[1] = einv + 1
    kap = o.gno_kernel_eval(coords[ecols - 1], th, H, Fo * Fi)
    nsq = max(rows.size, ncols.size)
    xs = np.zeros((nsq, Fi), np.float32); xs[:ncols.size] = x[torch.from_numpy(ncols - 1).to(dev)].cpu().numpy()
    sia_sq = np.concatenate([sia, np.full(nsq - rows.size, sia[-1], np.int32)])
    m_ref = o.gno_aggregate(xs, kap, sia_sq, sja, Fo)[:rows.size]
    par = {"against": "materialising oracle on 300 sampled rows (m) and 300 sampled columns (dx); dtheta: adjoint identity",
           "m_rel": rel(m[torch.from_numpy(rows).to(dev)].cpu().numpy(), m_ref), "tol": TOL}
    kmag = gno_kappa_magnitude(o, coords[ecols - 1], th, d, H, Fo * Fi)
    par["m_elementwise_worst"] = ew_worst(m[torch.from_numpy(rows).to(dev)].cpu().numpy(), m_ref,
                                          o.gno_aggregate(np.abs(xs), kmag, sia_sq, sja, Fo)[:rows.size])
    dx, dth, _, fused = ops.gno_aggregate_bwd(g, theta, co, x, gup, d, H, s_save=keep[0])      # what the layer's reverse pass calls
    csel = np.sort(rng.choice(N, 300, replace=False))
    lut = np.full(N, -1, np.int64); lut[csel] = np.arange(csel.size)
    hit = np.nonzero(lut[ja[0].astype(np.int64) - 1] >= 0)[0]
    src_rows = np.searchsorted(ia, hit + 1, side="right") - 1          # row of every entry that points at a chosen column
    src, sinv = np.unique(src_rows, return_inverse=True)
    order = np.argsort(sinv, kind="stable")
    cia = np.concatenate([[1], 1 + np.cumsum(np.bincount(sinv, minlength=src.size))]).astype(np.int32)
    ecols_c, einv_c = np.unique(ja[1, hit], return_inverse=True)
    cja = np.zeros((2, hit.size), np.int32, order="F")
    cja[0] = lut[ja[0, hit[order]].astype(np.int64) - 1] + 1; cja[1] = einv_c[order] + 1
    kap_c = o.gno_kernel_eval(coords[ecols_c - 1], th, H, Fo * Fi)
    nsq = max(src.size, csel.size)
    cia_sq = np.concatenate([cia, np.full(nsq - src.size, cia[-1], np.int32)])
    g_sq = np.zeros((nsq, Fo), np.float32); g_sq[:src.size] = gup[torch.from_numpy(src).to(dev)].cpu().numpy()
    dx_ref = o.gno_aggregate_bwd_x(g_sq, kap_c, cia_sq, cja, Fi)[:csel.size]
    par["dx_rel"] = rel(dx[torch.from_numpy(csel).to(dev)].cpu().numpy(), dx_ref)
    par["dx_elementwise_worst"] = ew_worst(dx[torch.from_numpy(csel).to(dev)].cpu().numpy(), dx_ref,
                                           o.gno_aggregate_bwd_x(np.abs(g_sq), gno_kappa_magnitude(o, coords[ecols_c - 1], th, d, H, Fo * Fi),
                                                                 cia_sq, cja, Fi)[:csel.size])
    par["reverse_pass"] = "athena_mp_gno_aggregate_bwd: dx and dtheta from one G" if fused else "separate entry points"
    dx_sep = ops.gno_aggregate_bwd_x(g, theta, co, gup, d, H, Fi)
    par["dx_fused_vs_separate_rel"] = float((dx - dx_sep).abs().max().item() / dx_sep.abs().max().item())
    del dx_sep
    offV = H * d + H
    lhs = (m.double() * gup.double()).sum().item()
    scale = (m.double().abs() * gup.double().abs()).sum().item()
    par["dtheta_adjoint_rel"] = abs(lhs - (theta[offV:].double() * dth[offV:].double()).sum().item()) / scale
    par["ok"] = bool(par["m_rel"] <= TOL and par["dx_rel"] <= TOL and par["dx_fused_vs_separate_rel"] <= TOL
                     and par["m_elementwise_worst"] <= TOL and par["dx_elementwise_worst"] <= TOL
                     and par["dtheta_adjoint_rel"] <= TOL and torch.isfinite(dth).all().item())
    step = t["fwd_keeps_S"] + t["bwd_x+theta_one_contraction"]
    step_sep = t["fwd_keeps_S"] + t["bwd_x"] + t["bwd_theta_S_kept"]
    # the CPU path beside it: the reference's MATERIALISING algorithm (kappa [F_out F_in, E]) is infeasible at this size (246 GB);
    # the oracle runs it on a 2 000-vertex mesh of the same generator and widths
    ns = 2000
    sia_c, sja_c, cs_c = synth.radius_graph(ns, seed=9)
    rs = np.random.default_rng(5)
    xs_c = rs.uniform(-1, 1, (ns, Fi)).astype(np.float32); gs_c = rs.uniform(-1, 1, (ns, Fo)).astype(np.float32)
    t0 = time.perf_counter()
    kap_s = o.gno_kernel_eval(cs_c, th, H, Fo * Fi); o.gno_aggregate(xs_c, kap_s, sia_c, sja_c, Fo)
    o.gno_aggregate_bwd_x(gs_c, kap_s, sia_c, sja_c, Fi); dk_s = o.gno_aggregate_bwd_k(gs_c, xs_c, cs_c.shape[0], sia_c, sja_c)
    o.gno_kernel_bwd_theta(cs_c, th, dk_s, H)
    tc = time.perf_counter() - t0
    cpu = {"value": sja_c.shape[1] / tc, "unit": "entries/s", "cores": 1, "kind": "port",
           "sample": f"materialising oracle (the reference's algorithm), radius mesh of {ns} vertices = {sja_c.shape[1]} entries / {cs_c.shape[0]} edge "
                     f"columns, same widths, forward + dx + dtheta, {tc:.1f} s"}
    return {"config": "configs[3]" + (" (vertices in cell order)" if MESH_ORDER else ""), "workload": f"GNO aggregation, radius mesh {N} vertices / {nnz} entries / {E} edge columns, F_in = F_out = H = 64, d = 3, "
            "training step = forward (keeps S) + reverse pass (dx and dtheta from one contraction)", "step_ms": round(step, 3),
            "step_ms_separate_reverse_launches": round(step_sep, 3), "entries_per_s": nnz / step * 1e3,
            "s_kept_GB": round(s_bytes / 1e9, 2), "ops": opsd, "parity": par, "cpu_baseline": cpu}


def run_c5(dev, reps):
    """BASELINE configs[4]'s whole graph on ONE GPU (the 8-GPU run is the driver's): the headline's step at 10 M / 150 M / 256"""
    from oracle import oracle as o

    N, pairs, F = 10_000_000, 70_000_000, 256
    ia, ja = synth.random_graph_csr(N, pairs)
    nnz = ja.shape[1]
    g = DeviceGraph(ia, ja, n_edge_cols=0)
    gen = torch.Generator(device=dev).manual_seed(1)
    x = torch.rand((N, F), device=dev, generator=gen).mul_(2.0).sub_(1.0)
    dz = torch.rand((N, F), device=dev, generator=gen).mul_(2.0).sub_(1.0)
    w_h = synth.kipf_weight(F)
    w = torch.from_numpy(w_h).to(dev)
    P = torch.empty((N, F), device=dev); Z = torch.empty((N, F), device=dev); dX = torch.empty((N, F), device=dev)
    dW = torch.empty(F * F, device=dev)
    t = {"fwd (kipf_propagate + matmul, fused)": timeit(lambda: ops.kipf_layer_fwd(g, x, w, F, P=P, Z=Z), reps),
         "dW = dZ . P^T": timeit(lambda: ops.matmul_dw(P, dz, out=dW), reps),
         "dX = (A^T dZ) W (fused pull)": timeit(lambda: ops.kipf_layer_bwd_x(g, dz, w, F, out=dX), reps)}
    names = list(t)
    agg = nnz * (4 * F + 8) + N * (4 * F + 8)
    opsd = {names[0]: hbm_op(t[names[0]], agg + N * 4 * F, "SURVEY.md 8d per-entry model + the Z rows written"),
            names[1]: mfma_op(t[names[1]], 2.0 * N * F * F),
            names[2]: hbm_op(t[names[2]], nnz * (4 * F + 4) + N * (4 * F + 4), "SURVEY.md 8d per-entry model (pull over the transposed CSR)")}
    rng = np.random.default_rng(3)
    rows = np.sort(rng.choice(N, 4000, replace=False))
    deg = np.diff(ia).astype(np.int32)
    ent = np.concatenate([np.arange(ia[r] - 1, ia[r + 1] - 1) for r in rows])
    cols, inv = np.unique(ja[0, ent].astype(np.int64) - 1, return_inverse=True)
    sia = np.concatenate([[1], 1 + np.cumsum(ia[rows + 1] - ia[rows])]).astype(np.int32)
    sja = np.zeros((2, ent.size), np.int32, order="F"); sja[0] = inv + 1
    rsel = torch.from_numpy(rows).to(dev)
    xc = x[torch.from_numpy(cols).to(dev)].cpu().numpy()
    p_ref = o.kipf_propagate_rect(xc, sia, sja, deg[rows], deg[cols])
    par = {"against": "oracle on 4 000 sampled rows (compact sub-problems); dW against float64 on the device",
           "P_bit_exact": bool(np.array_equal(P[rsel].cpu().numpy(), p_ref)), "Z_rel": rel(Z[rsel].cpu().numpy(), o.matmul(w_h, p_ref, F)), "tol": TOL}
    # the reverse pull of the sampled rows: the graph is symmetric as a multiset, so column v's sources are row v's neighbours
    dzc = dz[torch.from_numpy(cols).to(dev)].cpu().numpy()
    ones = np.ones(max(rows.size, cols.size), np.int32)
    dx_ref = o.kipf_propagate_rect(o.matmul_dx(w_h, dzc, F), sia, sja, ones[:rows.size], ones[:cols.size])
    par["dX_rel"] = rel(dX[rsel].cpu().numpy(), dx_ref)
    # element-wise against the magnitude of each element's own terms (|P| |Wt|; the pull of |dZ| |W|)
    wt_abs = np.abs(w_h.astype(np.float64)).reshape(F, F)
    par["Z_elementwise_worst"] = ew_worst(Z[rsel].cpu().numpy(), o.matmul(w_h, p_ref, F), np.abs(p_ref.astype(np.float64)) @ wt_abs)
    dp_mag = (np.abs(dzc.astype(np.float64)) @ wt_abs.T).astype(np.float32)
    par["dX_elementwise_worst"] = ew_worst(dX[rsel].cpu().numpy(), dx_ref,
                                           o.kipf_propagate_rect(dp_mag, sia, sja, ones[:rows.size], ones[:cols.size]))
    d64 = torch.zeros((F, F), device=dev, dtype=torch.float64)
    for r0 in range(0, N, 1 << 20):
        d64 += P[r0:r0 + (1 << 20)].double().T @ dz[r0:r0 + (1 << 20)].double()
    par["dW_rel_vs_float64"] = float((dW.double() - d64.reshape(-1)).abs().max().item() / d64.abs().max().item())
    par["ok"] = bool(par["P_bit_exact"] and max(par["Z_rel"], par["dX_rel"], par["dW_rel_vs_float64"], par["Z_elementwise_worst"],
                                                par["dX_elementwise_worst"]) <= TOL)
    step = sum(t.values())
    return {"config": "configs[4] on one GPU", "workload": f"Kipf GCN layer fwd+bwd, random graph {N} vertices / {nnz} entries, {F} features, fp32, ONE GPU "
            "(the 8-way partition is the driver's multi-GPU run)", "step_ms": round(step, 3), "entries_per_s": nnz / step * 1e3, "ops": opsd, "parity": par}


def run_kb(dev, reps):
    """the Kipf layer on a BATCH of small graphs (msgpass_chemical / onnx_gnn-shaped: BASELINE configs[0]'s layer at configs[2]'s batch):
    130 000 molecule-sized graphs as one block-diagonal graph, 64 and 128 features, the headline's step -- forward (aggregation from
    LDS + dense step + relu), dW, reverse to x.  Parity: the first 2 000 graphs are an exact sub-problem for the oracle."""
    from oracle import oracle as o

    S = 130000
    ia, ja, voff, E = synth.molecule_batch(S)
    N, nnz = ia.size - 1, ja.shape[1]
    g = DeviceGraph(ia, ja, n_edge_cols=0)
    NG = 2000
    v1 = int(voff[NG]); w1 = int(ia[v1]) - 1
    sia, sja = ia[:v1 + 1].copy(), np.asfortranarray(ja[:, :w1])
    res = {"config": "Kipf on a block-diagonal batch", "workload": f"Kipf GCN layer fwd+bwd on {S} molecule-sized graphs = {N} vertices / {nnz} entries "
           "as one block-diagonal graph (band 29), relu, fp32", "widths": {}}
    ok = True
    for F in (64, 128):
        gen = torch.Generator(device=dev).manual_seed(F)
        x = torch.rand((N, F), device=dev, generator=gen).mul_(2.0).sub_(1.0)
        dy = torch.rand((N, F), device=dev, generator=gen).mul_(2.0).sub_(1.0)
        w_h = synth.kipf_weight(F)
        w = torch.from_numpy(w_h).to(dev)
        P = torch.empty((N, F), device=dev); Y = torch.empty((N, F), device=dev); dX = torch.empty((N, F), device=dev)
        dZ = torch.empty((N, F), device=dev); dW = torch.empty(F * F, device=dev)
        t = {"fwd (banded aggregation + dense step + relu; one launch at 64)": timeit(lambda: ops.kipf_layer_fwd(g, x, w, F, act="relu", P=P, Z=Y), reps),
             "relu reverse factor": timeit(lambda: ops.activation_bwd("relu", Y, dy, out=dZ), reps),
             "dW = dZ . P^T": timeit(lambda: ops.matmul_dw(P, dZ, out=dW), reps),
             "dX = (A^T dZ) W (banded pull + dense step; one launch at 64)": timeit(lambda: ops.kipf_layer_bwd_x(g, dZ, w, F, out=dX), reps)}
        names = list(t)
        agg = nnz * 8 + 2 * N * 4 * F          # ids + coefficients, rows in once (banded: every row staged once per block), rows out
        one = F == 64                          # 64 -> 64: ONE launch (banded_fused.hip), the aggregated tile never re-read from HBM
        opsd = {names[0]: hbm_op(t[names[0]], agg + (1 if one else 2) * N * 4 * F,
                                 "rows in, P out, Z out, ids + coefficients" + ("" if one else "; P read again by the dense step's launch")),
                names[1]: hbm_op(t[names[1]], 3 * N * 4 * F, "three tensors"),
                names[2]: mfma_op(t[names[2]], 2.0 * N * F * F),
                names[3]: hbm_op(t[names[3]], nnz * 4 + (2 if one else 4) * N * 4 * F,
                                 "rows in, dX out, ids" + ("" if one else "; the pulled rows out and in again between the two launches"))}
        step = sum(t.values())
        par = {}
        if not TIMING_ONLY:
            xs = x[:v1].cpu().numpy()
            p_ref = o.kipf_propagate(xs, sia, sja)
            z_ref = o.activation("relu", o.matmul(w_h, p_ref, F))
            # (the reverse factor from the DEVICE's Y: where z is within rounding of relu's kink the oracle's own z may sit on the other
            # side, and the whole dy of that element would count as an error of the kernels behind it)
            dz_h = o.activation_bwd("relu", Y[:v1].cpu().numpy(), dy[:v1].cpu().numpy())
            dx_ref = o.kipf_propagate_bwd(o.matmul_dx(w_h, dz_h, F), sia, sja)
            par = {"against": f"oracle on the first {NG} graphs ({v1} vertices: an exact sub-problem of the block-diagonal batch)",
                   "P_bit_exact": bool(np.array_equal(P[:v1].cpu().numpy(), p_ref)), "Z_rel": rel(Y[:v1].cpu().numpy(), z_ref),
                   "dX_rel": rel(dX[:v1].cpu().numpy(), dx_ref), "tol": TOL}
            d64 = torch.zeros((F, F), device=dev, dtype=torch.float64)
            for r0 in range(0, N, 1 << 20):
                d64 += P[r0:r0 + (1 << 20)].double().T @ dZ[r0:r0 + (1 << 20)].double()
            par["dW_rel_vs_float64"] = float((dW.double() - d64.reshape(-1)).abs().max().item() / d64.abs().max().item())
            par["ok"] = bool(par["P_bit_exact"] and max(par["Z_rel"], par["dX_rel"], par["dW_rel_vs_float64"]) <= TOL)
            ok = ok and par["ok"]
        res["widths"][str(F)] = {"step_ms": round(step, 4), "entries_per_s": nnz / step * 1e3, "ops": opsd, "parity": par}
    res["step_ms"] = res["widths"]["64"]["step_ms"]
    res["parity"] = {"ok": bool(ok), "per_width": "see widths"}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=["c3", "c4", "c5", "kb"], required=True)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--timing-only", action="store_true", help="c4: stop after the timed launches (rocprofv3 runs)")
    ap.add_argument("--mesh-order", choices=["drawn", "cells"], default="drawn",
                    help="c4: vertices numbered as drawn (BASELINE configs[3]: every gather a random row) or in Morton order of "
                         "their cell (the numbering a mesher / partitioner gives; what the 8-way row partition uses)")
    a = ap.parse_args()
    global TIMING_ONLY, MESH_ORDER
    TIMING_ONLY = a.timing_only
    MESH_ORDER = None if a.mesh_order == "drawn" else "cells"
    _capi.init(0)
    dev = torch.device("cuda:0")
    t0 = time.perf_counter()
    res = {"c3": run_c3, "c4": run_c4, "c5": run_c5, "kb": run_kb}[a.config](dev, a.reps)
    res["wall_s"] = round(time.perf_counter() - t0, 1)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
