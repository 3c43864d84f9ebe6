"""GPU: seeded fuzz of the Duvenaud and GNO op chains against the oracle -- random molecule-like batches
(ragged graph sizes, empty graphs, isolated vertices, self loops with edge id 0), random feature counts
on both sides of every kernel-selection threshold (VALU kernels / register-resident MFMA / tiled
fallback), random degree clamps."""
import numpy as np
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu


def T(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def H(t):
    return t.detach().cpu().numpy()


def _batch(rng, n_graphs, max_size, self_loops, isolated_frac):
    """block-diagonal batch of random connected-ish graphs; returns adj_ia, adj_ja, seg, n_edges"""
    rows, cols, eids, seg = [], [], [], [0]
    v0 = e0 = 0
    for _ in range(n_graphs):
        nv = int(rng.integers(0, max_size + 1))
        pairs = [(i, i + 1) for i in range(nv - 1) if rng.random() > isolated_frac]
        pairs += [tuple(rng.integers(0, nv, 2)) for _ in range(int(rng.integers(0, nv + 1)))] if nv > 1 else []
        pairs = [(int(a), int(b)) for a, b in pairs if a != b]
        for k, (a, b) in enumerate(pairs):
            rows += [v0 + a, v0 + b]; cols += [v0 + b, v0 + a]; eids += [e0 + k + 1] * 2
        if self_loops:
            rows += list(range(v0, v0 + nv)); cols += list(range(v0, v0 + nv)); eids += [0] * nv
        v0 += nv; e0 += len(pairs)
        seg.append(v0)
    rows, cols, eids = np.array(rows, np.int64), np.array(cols, np.int64), np.array(eids, np.int64)
    order = np.lexsort((eids, rows))
    rows, cols, eids = rows[order], cols[order], eids[order]
    ia = np.concatenate([[1], 1 + np.cumsum(np.bincount(rows, minlength=v0))]).astype(np.int32)
    ja = np.zeros((2, rows.size), np.int32, order="F")
    ja[0] = cols + 1; ja[1] = eids
    return ia, ja, np.array(seg, np.int32), e0


@pytest.mark.parametrize("seed", range(16))
def test_duvenaud_chain_fuzz(dev, oracle, seed):
    from athena_amd import DeviceGraph, ops

    rng = np.random.default_rng(5000 + seed)
    big = seed % 4 == 0                                   # enough vertices for the MFMA route (n >= 1024)
    ia, ja, seg, E = _batch(rng, int(rng.integers(120, 200)) if big else int(rng.integers(1, 30)),
                            24 if big else 12, self_loops=bool(seed % 2), isolated_frac=0.15)
    N = ia.size - 1
    if N == 0 or E == 0:
        pytest.skip("degenerate draw")
    Fv = int(rng.choice([3, 6, 8, 16, 20, 32, 64]))
    Fe = int(rng.choice([1, 2, 4, 8]))
    Fo = int(rng.choice([4, 7, 16, 32, 64]))
    O = int(rng.choice([1, 3, 10, 16, 20]))
    mn = int(rng.integers(1, 3)); mx = mn + int(rng.integers(0, 6))
    act = str(rng.choice(["sigmoid", "tanh", "relu", "none"]))
    g = DeviceGraph(ia, ja, n_edge_cols=E)
    x = rng.uniform(0, 1, (N, Fv)).astype(np.float32)
    e = rng.uniform(0, 1, (E, Fe)).astype(np.float32)
    w = (rng.standard_normal(Fo * (Fv + Fe) * (mx - mn + 1)) * 0.3).astype(np.float32)
    R = (rng.standard_normal(O * Fo) * 0.5).astype(np.float32)
    gout = rng.standard_normal((seg.size - 1, O)).astype(np.float32)

    a = ops.duvenaud_propagate(g, T(x, dev), T(e, dev))
    ao = oracle.duvenaud_propagate(x, e, ia, ja)
    assert np.array_equal(H(a), ao)
    z = ops.duvenaud_update_act(g, a, T(w, dev), mn, mx, Fo, act=act)
    from oracle import oracle64 as o64       # float64 twin: yardstick of the anchored 1e-5 (helpers.assert_close)
    zo = oracle.activation(act, oracle.duvenaud_update(ao, w, ia, mn, mx, Fo))
    assert_close(H(z), zo, 1e-5, "update+act", f64=lambda: o64.activation(act, o64.duvenaud_update(ao, w, ia, mn, mx, Fo)))
    p, out = ops.duvenaud_readout(T(R, dev), T(zo, dev), T(seg, dev), O)
    po = oracle.softmax_cols(oracle.matmul(R, zo, O))
    assert_close(H(p), po, 1e-5, "readout p", f64=lambda: o64.softmax_cols(o64.matmul(R, zo, O)))
    assert_close(H(out), oracle.segment_sum(po, seg), 1e-5, "readout out", f64=lambda: o64.segment_sum(po, seg))
    dc, dR = ops.duvenaud_readout_bwd(T(R, dev), T(zo, dev), T(po, dev), T(seg, dev), T(gout, dev), act=act)
    dl = oracle.softmax_cols_bwd(po, np.repeat(gout, np.diff(seg), axis=0))
    dco = oracle.activation_bwd(act, zo, oracle.matmul_dx(R, dl, Fo))
    dl64 = lambda: o64.softmax_cols_bwd(po, np.repeat(gout, np.diff(seg), axis=0))
    assert_close(H(dc), dco, 1e-5, "readout reverse dc", f64=lambda: o64.activation_bwd(act, zo, o64.matmul_dx(R, dl64(), Fo)))
    assert_close(H(dR), oracle.matmul_dw(dl, zo), 1e-5, "readout reverse dR", f64=lambda: o64.matmul_dw(dl64(), zo))
    dw = ops.duvenaud_update_bwd_w(g, T(dco, dev), T(ao, dev), mn, mx)
    assert_close(H(dw), oracle.duvenaud_update_bwd_w(dco, ao, ia, mn, mx), 1e-5, "dW", f64=lambda: o64.duvenaud_update_bwd_w(dco, ao, ia, mn, mx))
    da = ops.duvenaud_update_bwd_a(g, T(dco, dev), T(w, dev), mn, mx, Fv + Fe)
    dao = oracle.duvenaud_update_bwd_a(dco, w, ia, mn, mx, Fv + Fe)
    assert_close(H(da), dao, 1e-5, "da", f64=lambda: o64.duvenaud_update_bwd_a(dco, w, ia, mn, mx, Fv + Fe))
    assert np.array_equal(H(ops.duvenaud_propagate_bwd_x(g, T(dao, dev), Fv)), oracle.duvenaud_propagate_bwd_x(dao, Fv, ia, ja))
    assert np.array_equal(H(ops.duvenaud_propagate_bwd_e(g, T(dao, dev), Fv)), oracle.duvenaud_propagate_bwd_e(dao, Fv, E, ia, ja))


@pytest.mark.parametrize("seed", range(12))
def test_duvenaud_one_call_reverse_fuzz(dev, seed):
    """athena_mp_duvenaud_readout_update_bwd on random batches (isolated vertices, self loops or none, random degree windows, every
    F_e / O / activation the one launch takes and shapes it hands to the two launches): da_x / da_e carry the bits of
    duvenaud_readout_bwd + duvenaud_update_bwd_split, dW / dR their sums in another order; accumulate_da_e adds exactly."""
    from athena_amd import DeviceGraph, ops

    rng = np.random.default_rng(9100 + seed)
    big = seed % 3 != 2                                   # two in three draws reach the one launch (n >= 1024)
    ia, ja, seg, E = _batch(rng, int(rng.integers(90, 260)) if big else int(rng.integers(1, 30)), 24 if big else 12,
                            self_loops=bool(seed % 2), isolated_frac=0.15)
    N = ia.size - 1
    if N == 0 or E == 0:
        pytest.skip("degenerate draw")
    Fv = 64 if seed % 4 != 3 else int(rng.choice([16, 32, 128]))
    Fe = int(rng.choice([4, 8, 12, 16, 24, 32, 36]))      # 36: F_v + F_e = 100 > 96 -> the two launches
    O = int(rng.choice([1, 2, 5, 10, 13, 16, 20]))
    mn = int(rng.integers(1, 3)); mx = mn + int(rng.integers(0, 8))
    act = str(rng.choice(["sigmoid", "tanh", "relu", "none"]))
    g = DeviceGraph(ia, ja, n_edge_cols=E)
    Fc = Fv + Fe
    a = T(rng.uniform(-1, 1, (N, Fc)).astype(np.float32), dev)
    W = T((0.3 * rng.standard_normal(Fv * Fc * (mx - mn + 1))).astype(np.float32), dev)
    R = T((0.5 * rng.standard_normal(O * Fv)).astype(np.float32), dev)
    gout = T(rng.standard_normal((seg.size - 1, O)).astype(np.float32), dev)
    segd = T(seg, dev)
    dzn = T(rng.standard_normal((N, Fv)).astype(np.float32), dev) if seed % 2 else None
    z = ops.duvenaud_update_act(g, a, W, mn, mx, Fv, act=act)
    p, _ = ops.duvenaud_readout(R, z, segd, O)
    dc, dR2 = ops.duvenaud_readout_bwd(R, z, p, segd, gout, act=act, dz_next=dzn)
    da_x2, da_e2, dW2 = ops.duvenaud_update_bwd_split(g, dc, a, W, mn, mx, Fv)
    da_x, da_e, dW, dR = ops.duvenaud_readout_update_bwd(g, R, z, p, segd, gout, a, W, mn, mx, Fv, act=act, dz_next=dzn)
    assert torch.equal(da_x, da_x2) and torch.equal(da_e, da_e2)
    assert_close(H(dW), H(dW2), 1e-5, "dW")
    assert_close(H(dR), H(dR2), 1e-5, "dR")
    base = T(rng.uniform(-1, 1, (N, Fe)).astype(np.float32), dev)
    da_x3, da_e3, dW3, _ = ops.duvenaud_readout_update_bwd(g, R, z, p, segd, gout, a, W, mn, mx, Fv, act=act, dz_next=dzn, da_e=base.clone())
    assert torch.equal(da_x3, da_x) and torch.equal(da_e3, base + da_e) and torch.equal(dW3, dW)


@pytest.mark.parametrize("seed", range(10))
def test_gno_fuzz(dev, oracle, seed):
    from athena_amd import DeviceGraph, ops

    rng = np.random.default_rng(7000 + seed)
    ia, ja, _, E = _batch(rng, 1, int(rng.integers(5, 400)), self_loops=bool(seed % 3 == 0), isolated_frac=0.1)
    N = ia.size - 1
    if N < 2 or E == 0:
        pytest.skip("degenerate draw")
    d = int(rng.integers(1, 4)); Hh = int(rng.choice([3, 8, 16, 32, 64]))
    Fi = int(rng.choice([2, 5, 16, 32, 64])); Fo = int(rng.choice([3, 8, 32, 64]))
    g = DeviceGraph(ia, ja, n_edge_cols=E)
    coords = rng.standard_normal((E, d)).astype(np.float32)
    x = rng.uniform(-1, 1, (N, Fi)).astype(np.float32)
    theta = (0.4 * rng.standard_normal(Hh * d + Hh + Fo * Fi * Hh + Fo * Fi)).astype(np.float32)
    up = rng.uniform(-1, 1, (N, Fo)).astype(np.float32)
    from oracle import oracle64 as o64       # float64 twin of the materialising oracle: yardstick of the anchored 1e-5
    kap = oracle.gno_kernel_eval(coords, theta, Hh, Fo * Fi)
    kap64 = lambda: o64.gno_kernel_eval(coords, theta, Hh, Fo * Fi)
    dk64 = lambda: o64.gno_aggregate_bwd_k(up, x, E, ia, ja)
    m = ops.gno_aggregate(g, T(theta, dev), T(coords, dev), T(x, dev), d, Hh, Fo)
    assert_close(H(m), oracle.gno_aggregate(x, kap, ia, ja, Fo), 1e-5, "gno fwd", f64=lambda: o64.gno_aggregate(x, kap64(), ia, ja, Fo))
    dx = ops.gno_aggregate_bwd_x(g, T(theta, dev), T(coords, dev), T(up, dev), d, Hh, Fi)
    assert_close(H(dx), oracle.gno_aggregate_bwd_x(up, kap, ia, ja, Fi), 1e-5, "gno dx", f64=lambda: o64.gno_aggregate_bwd_x(up, kap64(), ia, ja, Fi))
    dk = oracle.gno_aggregate_bwd_k(up, x, E, ia, ja)
    dth = ops.gno_aggregate_bwd_theta(g, T(theta, dev), T(coords, dev), T(x, dev), T(up, dev), d, Hh)
    assert_close(H(dth), oracle.gno_kernel_bwd_theta(coords, theta, dk, Hh), 1e-5, "gno dtheta",
                 f64=lambda: o64.gno_kernel_bwd_theta(coords, theta, dk64(), Hh))
    dco = ops.gno_aggregate_bwd_coords(g, T(theta, dev), T(coords, dev), T(x, dev), T(up, dev), d, Hh)
    assert_close(H(dco), oracle.gno_kernel_bwd_coords(coords, theta, dk, Hh), 1e-5, "gno dcoords",
                 f64=lambda: o64.gno_kernel_bwd_coords(coords, theta, dk64(), Hh))


@pytest.mark.parametrize("seed", range(12))
def test_kipf_layer_order_fuzz(dev, seed):
    """seeded fuzz of the Kipf layer mirror: random step counts, feature widths on both sides of the narrowing
    threshold, fused and separate activations, hub vertices -- the layer in each order (aggregate_first,
    transform_first, auto) against the per-sample oracle (1e-5; gradients through up to three chained steps anchored on
    the oracle's float64 twin: |gpu - f64| <= |oracle - f64| + 1e-5)"""
    import oracle_layers as ol
    from helpers import csr_from_index_list
    from athena_amd import ops
    from athena_amd.layers import kipf_msgpass_layer_type

    rng = np.random.default_rng(1000 + seed)
    T_ = int(rng.integers(1, 4))
    nvf = [int(rng.choice([1, 3, 8, 17, 32, 64, 96, 128])) for _ in range(T_ + 1)]
    act = [("none"), ("relu"), ("tanh"), ("swish"), ops.actv_type("leaky_relu", alpha=0.2), ("softmax")][seed % 6]
    gs = []
    for n in rng.integers(2, 60, int(rng.integers(1, 5))):
        n = int(n)
        pairs = [[i, i + 1] for i in range(1, n)]
        pairs += [[int(a), int(b)] for a, b in rng.integers(1, n + 1, (int(rng.integers(0, 3 * n)), 2)) if a != b]
        gs.append(csr_from_index_list(n, np.array(pairs).T, self_loops=True))
    if seed % 3 == 0:      # one graph with a hub vertex (> 512 entries: segment plan)
        n = 700
        pairs = [[1, k] for k in range(2, n + 1)] + [[i, i + 1] for i in range(2, n)]
        gs.append(csr_from_index_list(n, np.array(pairs).T, self_loops=True))
    xs = [rng.uniform(-1, 1, (g.num_vertices, nvf[0])).astype(np.float32) for g in gs]
    ref_layer = kipf_msgpass_layer_type(num_vertex_features=nvf, num_time_steps=T_, activation=act, seed=seed)
    params = ref_layer.get_params()
    plist, o_ = [], 0
    for t in range(1, T_ + 1):
        plist.append(params[o_:o_ + nvf[t] * nvf[t - 1]]); o_ += nvf[t] * nvf[t - 1]
    outs, tapes = ol.kipf_forward(gs, xs, plist, nvf, act)
    ups = [rng.uniform(-1, 1, o.shape).astype(np.float32) for o in outs]
    exact = bool(seed & 1)
    dxs, grads = ol.kipf_backward(gs, tapes, plist, nvf, act, ups, exact=exact)
    import functools

    @functools.lru_cache(None)
    def hi():
        with ol.double_precision():
            _, t64 = ol.kipf_forward(gs, xs, plist, nvf, act)
            a, b = ol.kipf_backward(gs, t64, plist, nvf, act, ups, exact=exact)
        return np.concatenate(a), np.concatenate(b)
    for order in ("aggregate_first", "transform_first", "auto"):
        layer = kipf_msgpass_layer_type(num_vertex_features=nvf, num_time_steps=T_, activation=act, seed=seed, order=order)
        layer.set_params(params)
        layer.set_graph(gs)
        assert_close(H(layer.forward(xs)), np.concatenate(outs), 1e-5, f"seed {seed} {order} {nvf} fwd")
        dx = layer.backward(np.concatenate(ups), exact=exact)
        assert_close(H(dx), np.concatenate(dxs), 1e-5, f"seed {seed} {order} {nvf} dx", f64=lambda: hi()[0])
        assert_close(layer.get_gradients(), np.concatenate(grads), 1e-5, f"seed {seed} {order} {nvf} dW", f64=lambda: hi()[1])


@pytest.mark.parametrize("seed", range(8))
def test_gno_fused_shape_fuzz(dev, oracle, seed):
    """the producer / consumer GNO kernels (H = 64, widths 64) over random vertex counts, average row lengths from 1 to 40
    (every step-count variant of the sparse waves, the 17..32 and the > 32 entry row classes of the kernel-MLP backward),
    d = 1..3, with and without self-loop entries (no edge column); all four entry points against the oracle"""
    from athena_amd import DeviceGraph, ops
    from helpers import csr_from_index_list
    from oracle import oracle64 as o64

    rng = np.random.default_rng(9100 + seed)
    N = int(rng.integers(40, 900))
    avg = float(rng.choice([0.6, 2.0, 4.5, 8.0, 12.0, 20.0]))
    n_pairs = max(1, int(N * avg / 2))
    pairs = rng.integers(1, N + 1, (n_pairs, 2))
    if seed % 2:                                        # a few long rows on top
        hub = int(rng.integers(1, N + 1))
        pairs = np.concatenate([pairs, np.stack([np.full(50, hub), rng.integers(1, N + 1, 50)], 1)])
    pairs = pairs[pairs[:, 0] != pairs[:, 1]].T
    g0 = csr_from_index_list(N, pairs, self_loops=bool(seed % 3 == 0))
    ia, ja, E = g0.adj_ia, g0.adj_ja, pairs.shape[1]
    d = int(rng.integers(1, 4)); Hh = Fi = Fo = 64
    g = DeviceGraph(ia, ja, n_edge_cols=E)
    coords = rng.standard_normal((E, d)).astype(np.float32)
    x = rng.uniform(-1, 1, (N, Fi)).astype(np.float32)
    theta = (0.3 * rng.standard_normal(Hh * d + Hh + Fo * Fi * Hh + Fo * Fi)).astype(np.float32)
    up = rng.uniform(-1, 1, (N, Fo)).astype(np.float32)
    kap = oracle.gno_kernel_eval(coords, theta, Hh, Fo * Fi)
    th, co, xd, gd = T(theta, dev), T(coords, dev), T(x, dev), T(up, dev)
    assert_close(H(ops.gno_aggregate(g, th, co, xd, d, Hh, Fo)), oracle.gno_aggregate(x, kap, ia, ja, Fo), 1e-5, "gno fwd")
    assert_close(H(ops.gno_aggregate_bwd_x(g, th, co, gd, d, Hh, Fi)), oracle.gno_aggregate_bwd_x(up, kap, ia, ja, Fi), 1e-5, "gno dx")
    dk = oracle.gno_aggregate_bwd_k(up, x, E, ia, ja)
    dk64 = lambda: o64.gno_aggregate_bwd_k(up, x, E, ia, ja)
    assert_close(H(ops.gno_aggregate_bwd_theta(g, th, co, xd, gd, d, Hh)), oracle.gno_kernel_bwd_theta(coords, theta, dk, Hh), 1e-5,
                 "gno dtheta", f64=lambda: o64.gno_kernel_bwd_theta(coords, theta, dk64(), Hh))
    assert_close(H(ops.gno_aggregate_bwd_coords(g, th, co, xd, gd, d, Hh)), oracle.gno_kernel_bwd_coords(coords, theta, dk, Hh), 1e-5,
                 "gno dcoords", f64=lambda: o64.gno_kernel_bwd_coords(coords, theta, dk64(), Hh))


@pytest.mark.parametrize("seed", range(12))
def test_kipf_on_random_batches_fuzz(dev, oracle, seed):
    """round 6: kipf_propagate forward / reverse (both forms) bit for bit and the layer step at 1e-5 on random block-diagonal batches --
    banded ones (the LDS-staged gather with the coefficient at F = 64 / 128: small graphs, rows of <= 8 entries) and ones that are
    not (bigger graphs, denser rows, other widths: the general kernels), whichever route the library picks"""
    from athena_amd import DeviceGraph, ops
    from oracle import oracle64 as o64

    rng = np.random.default_rng(9000 + seed)
    banded = seed % 3 != 2
    ia, ja, seg, E = _batch(rng, int(rng.integers(40, 400)), 8 if banded else 40, self_loops=True, isolated_frac=0.1)
    N = ia.size - 1
    if N == 0:
        pytest.skip("degenerate draw")
    F = int(rng.choice([64, 128] if seed % 4 else [16, 32, 96, 256]))
    g = DeviceGraph(ia, ja)
    x = rng.uniform(-1, 1, (N, F)).astype(np.float32)
    up = rng.uniform(-1, 1, (N, F)).astype(np.float32)
    xd, upd = T(x, dev), T(up, dev)
    assert np.array_equal(H(ops.kipf_propagate(g, xd)), oracle.kipf_propagate(x, ia, ja))
    assert np.array_equal(H(ops.kipf_propagate_bwd(g, upd)), oracle.kipf_propagate_bwd(up, ia, ja))
    assert np.array_equal(H(ops.kipf_propagate_bwd(g, upd, exact=True)), oracle.kipf_propagate_bwd(up, ia, ja, exact=True))
    W = (rng.standard_normal(F * F) * np.sqrt(2.0 / F)).astype(np.float32)
    act = str(rng.choice(["none", "relu", "sigmoid", "tanh"]))
    P, Z = ops.kipf_layer_fwd(g, xd, T(W, dev), F, act=act)
    p_ref = oracle.kipf_propagate(x, ia, ja)
    assert np.array_equal(H(P), p_ref)
    assert_close(H(Z), oracle.activation(act, oracle.matmul(W, p_ref, F)), 1e-5, f"Z ({act})",
                 f64=lambda: o64.activation(act, o64.matmul(W, o64.kipf_propagate(x, ia, ja), F)))
    dX = ops.kipf_layer_bwd_x(g, upd, T(W, dev), F)
    assert_close(H(dX), oracle.kipf_propagate_bwd(oracle.matmul_dx(W, up, F), ia, ja), 1e-5, "dX",
                 f64=lambda: o64.kipf_propagate_bwd(o64.matmul_dx(W, up, F), ia, ja))


@pytest.mark.parametrize("seed", range(10))
def test_device_graph_builder_fuzz(dev, seed):
    """round 6: the device graph builder (hand-written radix passes, graph_build.hip / radix_sort.h) against the host builder, array for
    array, on random multigraphs of random size -- entry counts from a few to several 4 096-entry tiles, few and many edge columns
    (1 .. 3 radix passes), duplicate entries, empty rows"""
    import os
    from athena_amd import DeviceGraph

    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(1, 6000))
    deg = rng.poisson(float(rng.choice([0.5, 3.0, 9.0])), n).astype(np.int64)
    nnz = int(deg.sum())
    ia = np.concatenate([[1], 1 + np.cumsum(deg)]).astype(np.int32)
    ja = np.zeros((2, nnz), np.int32, order="F")
    n_edge_cols = int(rng.choice([0, 3, 300, 70000]))
    if nnz:
        ja[0] = rng.integers(1, n + 1, nnz)
        ja[1] = rng.integers(0, n_edge_cols + 1, nnz) if n_edge_cols else 0
    names = ("rowptr", "col", "eid", "coef", "t_rowptr", "t_src", "t_eid", "t_coef", "e_rowptr", "e_row", "e_entry", "deg_row", "deg_col")
    built = {}
    old = os.environ.get("ATHENA_MP_GRAPH_BUILD")
    try:
        for how in ("host", "device"):
            os.environ["ATHENA_MP_GRAPH_BUILD"] = how
            h = DeviceGraph(ia, ja, n_edge_cols=n_edge_cols)
            built[how] = {k: h.export(k) for k in names}
            h.close()
    finally:
        if old is None:
            os.environ.pop("ATHENA_MP_GRAPH_BUILD", None)
        else:
            os.environ["ATHENA_MP_GRAPH_BUILD"] = old
    for k in names:
        a, b = built["host"][k], built["device"][k]
        assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), f"{k} differs between the host and the device builder (seed {seed})"
