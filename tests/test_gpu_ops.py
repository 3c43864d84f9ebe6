"""GPU parity tests proper: every call goes through the C ABI (libathena_mp.so) and is compared
with the CPU oracle on the same seeded inputs.  Tolerance: the north star's 1e-5 relative fp32;
the aggregation kernels are additionally required to be BIT-EXACT (same summation order, strict
fp32) against the oracle."""
import os

import numpy as np
import pytest
import torch

from helpers import assert_close, assert_close_elementwise, csr_from_index_list, golden, random_graph, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def H(t):
    return t.cpu().numpy()


def test_native_library_is_loaded(dev):
    from athena_amd import _capi

    lib = _capi.load()
    assert lib.athena_mp_version() >= 100
    with open("/proc/self/maps") as fh:
        assert "libathena_mp.so" in fh.read()


def test_reference_kat_identity_graph(dev):
    from athena_amd import DeviceGraph, ops

    k = golden("reference_kat_kipf_identity.json")
    g = DeviceGraph(np.array(k["adj_ia"], np.int32), np.array(k["adj_ja"], np.int32), n_edge_cols=0)
    x = T(np.array(k["x"], np.float32), dev)
    assert np.abs(H(ops.kipf_propagate(g, x)) - np.array(k["forward"])).max() <= k["tol_abs"]
    assert np.abs(H(ops.kipf_propagate_bwd(g, torch.ones_like(x))) - np.array(k["grad_of_sum"])).max() <= k["tol_abs"]
    up = T(np.array(k["upstream"], np.float32), dev)
    assert np.abs(H(ops.kipf_propagate_bwd(g, up)) - np.array(k["reverse_partial"])).max() <= k["tol_abs"]
    # reverse_kipf_propagate (..._sub_kipf.f90:116-206): value = the same scatter; its function-form partial is
    # kipf_propagate itself (the KAT's forward_partial, test_diffstruc_extd_kipf.f90), its _val form the scatter
    assert np.abs(H(ops.reverse_kipf_propagate(g, up)) - np.array(k["reverse_partial"])).max() <= k["tol_abs"]
    up2 = T(np.array(k["second_upstream"], np.float32), dev)
    assert np.abs(H(ops.reverse_kipf_propagate_partial(g, up2)) - np.array(k["forward_partial"])).max() <= k["tol_abs"]
    assert np.array_equal(H(ops.reverse_kipf_propagate_partial(g, up2, val_form=True)), H(ops.kipf_propagate_bwd(g, up2)))


def test_reverse_kipf_propagate_named_entry_points_on_a_random_graph(dev, oracle):
    """a3 beyond the identity-graph KAT: on a random multigraph with hubs of degree > 1 the three named entry points
    equal the oracle's scatter / propagate bit for bit (athena_diffstruc_extd_sub_kipf.f90:116-205)"""
    from athena_amd import DeviceGraph, ops

    ia, ja = random_graph(3000, 14000, seed=21)
    g = DeviceGraph(ia, ja, n_edge_cols=0)
    a = np.random.default_rng(2).uniform(-1, 1, (3000, 48)).astype(np.float32)
    ad = T(a, dev)
    assert np.array_equal(H(ops.reverse_kipf_propagate(g, ad)), oracle.kipf_propagate_bwd(a, ia, ja))
    assert np.array_equal(H(ops.reverse_kipf_propagate_partial(g, ad)), oracle.kipf_propagate(a, ia, ja))
    assert np.array_equal(H(ops.reverse_kipf_propagate_partial(g, ad, val_form=True)), oracle.kipf_propagate_bwd(a, ia, ja))


def test_survey_recorded_reference_outputs(dev):
    from athena_amd import DeviceGraph, ops

    s = golden("survey_recorded_path_graph.json")
    ia, ja = np.array(s["graph"]["adj_ia"], np.int32), np.array(s["graph"]["adj_ja"], np.int32)
    g = DeviceGraph(ia, ja)
    x = T(np.array(s["x"], np.float32), dev)
    tol = s["tol_rel"]
    assert rel_err(H(ops.kipf_propagate(g, x)), s["kipf_fwd"]) <= tol
    assert rel_err(H(ops.kipf_propagate_bwd(g, T(np.array(s["kipf_bwd_upstream"], np.float32), dev))), s["kipf_bwd"]) <= tol
    a = ops.duvenaud_propagate(g, x, T(np.array(s["edge_features"], np.float32), dev))
    assert rel_err(H(a), s["duvenaud_propagate"]) <= tol
    du = s["duvenaud_update"]
    c = ops.duvenaud_update(g, a, T(np.array(du["weight"], np.float32), dev), du["min_degree"], du["max_degree"], du["num_outputs"])
    assert rel_err(H(c), du["out"]) <= tol


@pytest.mark.parametrize("F", [1, 3, 6, 7, 16, 64, 72, 128, 130, 256, 300, 516])
def test_kipf_fwd_bwd_bit_exact_vs_oracle(dev, oracle, F):
    from athena_amd import DeviceGraph, ops

    n = 2000 if F <= 130 else 500
    ia, ja = random_graph(n, 5 * n, seed=F, self_loops=True, isolated=7)
    rng = np.random.default_rng(F)
    x = rng.uniform(-1, 1, (n, F)).astype(np.float32)
    gr = rng.uniform(-1, 1, (n, F)).astype(np.float32)
    g = DeviceGraph(ia, ja)
    y = H(ops.kipf_propagate(g, T(x, dev)))
    yo = oracle.kipf_propagate(x, ia, ja)
    assert_close(y, yo, 1e-5, "fwd")
    assert np.array_equal(y, yo), f"fwd not bit-exact: {np.abs(y - yo).max()}"
    assert np.all(y[-7:] == 0)
    for exact in (False, True):
        d = H(ops.kipf_propagate_bwd(g, T(gr, dev), exact=exact))
        do = oracle.kipf_propagate_bwd(gr, ia, ja, exact=exact)
        assert_close(d, do, 1e-5, f"bwd exact={exact}")
        assert np.array_equal(d, do), f"bwd exact={exact} not bit-exact"


def test_kipf_ragged_degrees_and_hubs(dev, oracle):
    """power-law-ish rows: a few hubs with thousands of entries, many rows of length 0/1"""
    from athena_amd import DeviceGraph, ops

    rng = np.random.default_rng(11)
    n = 3000
    deg = np.minimum((rng.pareto(1.2, n) * 2).astype(np.int64), 2500)
    deg[:3] = [2500, 1500, 0]
    ia = np.concatenate([[1], 1 + np.cumsum(deg)]).astype(np.int32)
    nnz = int(deg.sum())
    ja = np.zeros((2, nnz), np.int32, order="F")
    ja[0] = rng.choice(np.nonzero(deg > 0)[0] + 1, nnz)      # neighbours have degree > 0 (finite coefficients)
    x = rng.uniform(-1, 1, (n, 128)).astype(np.float32)
    g = DeviceGraph(ia, ja, n_edge_cols=0)
    # rows of more than 512 entries are summed as ordered 512-entry segments (parallel lane groups):
    # same terms, different association -> tolerance there, bit-exact everywhere else
    y = H(ops.kipf_propagate(g, T(x, dev)))
    yo = oracle.kipf_propagate(x, ia, ja)
    short = deg <= 512
    assert np.isfinite(yo).all() and (~short).sum() >= 2
    assert np.array_equal(y[short], yo[short])
    assert_close(y[~short], yo[~short], 1e-5, "hub rows")
    # ... and element by element against the magnitude of each element's own terms (sum of |coef x|): an element whose
    # terms cancel is not excused by the tensor's maximum
    mag = oracle.kipf_propagate(np.abs(x), ia, ja)
    assert_close_elementwise(y[~short], yo[~short], mag[~short], 1e-5, "hub rows, element-wise")
    d = H(ops.kipf_propagate_bwd(g, T(x, dev)))
    do = oracle.kipf_propagate_bwd(x, ia, ja)
    cdeg = np.bincount(ja[0] - 1, minlength=n)
    assert np.array_equal(d[cdeg <= 512], do[cdeg <= 512])
    assert_close(d, do, 1e-5, "hub columns")
    assert_close_elementwise(d, do, oracle.kipf_propagate_bwd(np.abs(x), ia, ja), 1e-5, "hub columns, element-wise")


def test_kipf_empty_and_single(dev, oracle):
    from athena_amd import DeviceGraph, ops

    g = DeviceGraph(np.ones(1, np.int32), np.zeros((2, 0), np.int32), n_edge_cols=0)
    assert ops.kipf_propagate(g, torch.zeros((0, 8), device=dev)).shape == (0, 8)
    g1 = DeviceGraph(np.array([1, 1], np.int32), np.zeros((2, 0), np.int32), n_edge_cols=0)
    y = ops.kipf_propagate(g1, torch.ones((1, 8), device=dev))
    assert torch.all(y == 0)


@pytest.mark.parametrize("F", [1, 3, 6, 16, 32, 64, 130])
def test_kipf_reverse_both_forms_from_one_gather(dev, oracle, F):
    """athena_mp_kipf_propagate_bwd_dual: the coefficient-free scatter of the reference and the adjoint of the
    forward from ONE pass over the upstream rows -- each BIT-equal to its own launch and to the oracle; a graph
    with hub columns (segment plan) takes the two-pass route behind the same entry point"""
    from athena_amd import DeviceGraph, ops

    for hubs in (False, True):
        rng = np.random.default_rng(F + hubs)
        n = 900
        deg = rng.integers(1, 12, n)     # no empty rows: a zero-degree column would make the coefficient infinite
        if hubs:
            deg[[5, 400]] = [1500, 700]
        ia = np.concatenate([[1], 1 + np.cumsum(deg)]).astype(np.int32)
        ja = np.zeros((2, int(deg.sum())), np.int32, order="F")
        ja[0] = rng.integers(1, n + 1, ja.shape[1])
        if hubs:
            ja[0, :1200] = 17       # column 17 becomes a hub of the transposed structure
        g = DeviceGraph(ia, ja, n_edge_cols=0)
        up = rng.uniform(-1, 1, (n, F)).astype(np.float32)
        plain, coef = ops.kipf_propagate_bwd_dual(g, T(up, dev))
        p1, c1 = ops.kipf_propagate_bwd(g, T(up, dev)), ops.kipf_propagate_bwd(g, T(up, dev), exact=True)
        assert torch.equal(plain, p1) and torch.equal(coef, c1)
        po, co = oracle.kipf_propagate_bwd(up, ia, ja), oracle.kipf_propagate_bwd(up, ia, ja, exact=True)
        if hubs:      # segmented hub sums: a 1200-term fp32 sum in another order than the sequential oracle's.  1e-5, anchored
            # on the float64 evaluation of the same loops where the fp32 oracle's own rounding is larger than that
            from oracle import oracle64 as o64
            assert_close(H(plain), po, 1e-5, "plain", f64=lambda: o64.kipf_propagate_bwd(up, ia, ja))
            assert_close(H(coef), co, 1e-5, "coef", f64=lambda: o64.kipf_propagate_bwd(up, ia, ja, exact=True))
        else:
            assert np.array_equal(H(plain), po) and np.array_equal(H(coef), co)


@pytest.mark.parametrize("act", ["none", "relu", "sigmoid", "tanh"])
@pytest.mark.parametrize("F", [5, 32, 64])
def test_kipf_propagate_with_the_activation_in_its_store(dev, oracle, act, F):
    """athena_mp_kipf_propagate_act_fwd = activation(kipf_propagate(x)) in one launch, hub rows (segment plan +
    ordered combine) included: equal to the two launches bit for bit"""
    from athena_amd import DeviceGraph, ops

    rng = np.random.default_rng(F)
    n = 1200
    deg = rng.integers(0, 9, n)
    deg[[3, 700]] = [900, 1300]
    ia = np.concatenate([[1], 1 + np.cumsum(deg)]).astype(np.int32)
    ja = np.zeros((2, int(deg.sum())), np.int32, order="F")
    ja[0] = rng.integers(1, n + 1, ja.shape[1])
    g = DeviceGraph(ia, ja, n_edge_cols=0, row_deg=(deg + 1).astype(np.int32), col_deg=rng.integers(1, 9, n).astype(np.int32))
    x = T(rng.uniform(-1, 1, (n, F)).astype(np.float32), dev)
    one = ops.kipf_propagate_act(g, x, act=act)
    two = ops.kipf_propagate(g, x)
    if act != "none":
        two = ops.activation(act, two)
    assert torch.equal(one, two)


def test_kipf_rectangular_block_with_explicit_degrees(dev, oracle):
    """row partition with halo columns: degrees supplied (the multi-GPU shard shape)"""
    from athena_amd import DeviceGraph, ops

    rng = np.random.default_rng(5)
    n_rows, n_cols, F = 300, 800, 64
    deg = rng.integers(0, 20, n_rows)
    ia = np.concatenate([[1], 1 + np.cumsum(deg)]).astype(np.int32)
    ja = np.zeros((2, int(deg.sum())), np.int32, order="F")
    ja[0] = rng.integers(1, n_cols + 1, ja.shape[1])
    row_deg = (deg + rng.integers(0, 3, n_rows)).astype(np.int32) + 1
    col_deg = rng.integers(1, 30, n_cols).astype(np.int32)
    x = rng.uniform(-1, 1, (n_cols, F)).astype(np.float32)
    g = DeviceGraph(ia, ja, n_cols=n_cols, n_edge_cols=0, row_deg=row_deg, col_deg=col_deg)
    y = H(ops.kipf_propagate(g, T(x, dev)))
    assert np.array_equal(y, oracle.kipf_propagate_rect(x, ia, ja, row_deg, col_deg))
    gr = rng.uniform(-1, 1, (n_rows, F)).astype(np.float32)
    d = H(ops.kipf_propagate_bwd(g, T(gr, dev)))
    assert np.array_equal(d, oracle.kipf_propagate_bwd(gr, ia, ja, n_out=n_cols))


@pytest.mark.parametrize("N,Fi,Fo", [(1000, 128, 128), (4097, 64, 128), (333, 128, 64), (31, 32, 32), (5000, 64, 64),
                                     (700, 6, 10), (513, 7, 6), (100, 256, 256), (257, 96, 40), (2000, 128, 32),
                                     (3001, 256, 256), (2000, 72, 64), (5000, 64, 10), (5000, 10, 64), (1500, 4160, 64),
                                     (900, 130, 258), (129, 9, 33),
                                     # few outputs (a bias gradient is Fi = 1): the block's threads split the rows as well
                                     (300000, 1, 64), (70001, 1, 100), (9000, 3, 5), (255, 1, 1),
                                     # dW in 128 x 128 blocks on the register-resident kernel (N >= 4096; configs[4]'s 256 x 256)
                                     (4099, 256, 256), (5001, 256, 128), (4100, 128, 384), (6000, 512, 256)])
def test_gemm_family_vs_float64(dev, oracle, N, Fi, Fo):
    """matmul is diffstruc's (unpinned by the reference's tests): exact-math fp32 GEMM, checked
    against float64 at 1e-5 and against the oracle's k-ordered fp32 sum"""
    from athena_amd import ops

    rng = np.random.default_rng(N + Fi)
    P = rng.uniform(-1, 1, (N, Fi)).astype(np.float32)
    W = (rng.standard_normal(Fo * Fi) * np.sqrt(2.0 / Fi)).astype(np.float32)
    dZ = rng.uniform(-1, 1, (N, Fo)).astype(np.float32)
    Wt = W.reshape(Fi, Fo).astype(np.float64)
    Z = H(ops.matmul(T(W, dev), T(P, dev), Fo))
    assert_close(Z, P @ Wt, 1e-5, "fwd vs f64")
    assert_close(Z, oracle.matmul(W, P, Fo), 1e-5, "fwd vs oracle")
    dP = H(ops.matmul_dx(T(W, dev), T(dZ, dev), Fi))
    assert_close(dP, dZ @ Wt.T, 1e-5, "dx vs f64")
    dW = H(ops.matmul_dw(T(P, dev), T(dZ, dev)))
    assert_close(dW.reshape(Fi, Fo), P.astype(np.float64).T @ dZ, 1e-5, "dw vs f64")
    b = rng.standard_normal(Fo).astype(np.float32)
    Zb = H(ops.matmul(T(W, dev), T(P, dev), Fo, bias=T(b, dev), act="relu"))
    assert_close(Zb, np.maximum(P @ Wt + b, 0), 1e-5, "fused bias+relu")


def test_weight_gradient_in_blocks_repeats_and_agrees_with_the_tiled_kernel(dev):
    """gemm_dw_dispatch's blocked form (Fi, Fo multiples of 128 beyond 128): the same bits on every launch, and the generic
    tiled kernel (the route below 4096 rows) agrees on the same data"""
    from athena_amd import ops

    rng = np.random.default_rng(9)
    N, Fi, Fo = 9000, 256, 256
    P = T(rng.uniform(-1, 1, (N, Fi)).astype(np.float32), dev)
    dZ = T(rng.uniform(-1, 1, (N, Fo)).astype(np.float32), dev)
    dW = ops.matmul_dw(P, dZ)
    assert torch.equal(dW, ops.matmul_dw(P, dZ))
    ref = (P.double().T @ dZ.double()).reshape(-1)
    assert (dW.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    lo = ops.matmul_dw(P[:4000].contiguous(), dZ[:4000].contiguous()) + ops.matmul_dw(P[4000:8000].contiguous(), dZ[4000:8000].contiguous()) \
        + ops.matmul_dw(P[8000:].contiguous(), dZ[8000:].contiguous())         # three launches of the generic kernel
    assert (lo.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize("kind", ["none", "relu", "sigmoid", "tanh"])
def test_activations(dev, oracle, kind):
    from athena_amd import ops

    z = np.random.default_rng(1).standard_normal((1000, 37)).astype(np.float32) * 3
    g = np.random.default_rng(2).standard_normal(z.shape).astype(np.float32)
    y = ops.activation(kind, T(z, dev))
    yo = oracle.activation(kind, z)
    assert_close(H(y), yo, 1e-5)
    assert_close(H(ops.activation_bwd(kind, y, T(g, dev))), oracle.activation_bwd(kind, yo, g), 1e-5)
    if kind == "relu":
        assert torch.all(y >= 0)                       # test_kipf_msgpass_layer.f90 property
    if kind == "sigmoid":
        assert torch.all((y >= 0) & (y <= 1))          # test_duvenaud_msgpass_layer.f90 property


@pytest.mark.parametrize("Fv,Fe", [(6, 1), (8, 2), (64, 8), (5, 3), (128, 4)])
def test_duvenaud_propagate_and_grads_bit_exact(dev, oracle, Fv, Fe):
    from athena_amd import DeviceGraph, ops

    n = 1500
    ia, ja = random_graph(n, 2 * n, seed=Fv, self_loops=True, isolated=3)   # self loops carry edge id 0
    E = 2 * n
    rng = np.random.default_rng(Fv)
    x = rng.uniform(0, 1, (n, Fv)).astype(np.float32)
    e = rng.uniform(0, 1, (E, Fe)).astype(np.float32)
    g = DeviceGraph(ia, ja, n_edge_cols=E)
    c = ops.duvenaud_propagate(g, T(x, dev), T(e, dev))
    assert np.array_equal(H(c), oracle.duvenaud_propagate(x, e, ia, ja))
    up = rng.uniform(-1, 1, (n, Fv + Fe)).astype(np.float32)
    assert np.array_equal(H(ops.duvenaud_propagate_bwd_x(g, T(up, dev), Fv)), oracle.duvenaud_propagate_bwd_x(up, Fv, ia, ja))
    assert np.array_equal(H(ops.duvenaud_propagate_bwd_e(g, T(up, dev), Fv)), oracle.duvenaud_propagate_bwd_e(up, Fv, E, ia, ja))


@pytest.mark.parametrize("Fi,Fo,mn,mx,n", [(7, 6, 1, 10, 1200), (10, 4, 2, 3, 1200), (72, 64, 1, 10, 1200), (9, 5, 1, 1, 1200),
                                             (72, 64, 1, 10, 70000), (40, 32, 1, 6, 40000), (20, 16, 2, 5, 3000),
                                             (96, 64, 1, 4, 2000), (68, 48, 1, 3, 2000), (72, 64, 3, 3, 5000),
                                             (64, 96, 1, 5, 2000), (16, 16, 1, 2, 1024), (132, 64, 1, 4, 1500),
                                             (70, 64, 1, 4, 1500), (72, 64, 5, 12, 1500)])
def test_duvenaud_update_and_grads(dev, oracle, Fi, Fo, mn, mx, n):
    """small layers: exact VALU kernels; matrix-sized layers: register-resident-weight MFMA kernels over
    16-vertex bucket tiles (several tiles per wave and bucket changes inside a wave at n >= 40000; a
    single bucket; buckets that stay empty at (5,12)); (132,*) and (70,*) take the tiled fallback"""
    from athena_amd import DeviceGraph, ops

    ia, ja = random_graph(n, int(1.5 * n), seed=Fi, self_loops=True, isolated=2)
    rng = np.random.default_rng(Fi)
    a = rng.uniform(0, 4, (n, Fi)).astype(np.float32)
    w = rng.standard_normal(Fo * Fi * (mx - mn + 1)).astype(np.float32)
    up = rng.uniform(-1, 1, (n, Fo)).astype(np.float32)
    g = DeviceGraph(ia, ja)
    c = H(ops.duvenaud_update(g, T(a, dev), T(w, dev), mn, mx, Fo))
    co = oracle.duvenaud_update(a, w, ia, mn, mx, Fo)
    da = H(ops.duvenaud_update_bwd_a(g, T(up, dev), T(w, dev), mn, mx, Fi))
    dao = oracle.duvenaud_update_bwd_a(up, w, ia, mn, mx, Fi)
    if Fi < 16 or Fo < 16:   # small layers: VALU kernels in the reference's operation order, bit-exact
        assert np.array_equal(c, co), f"update fwd not bit exact ({rel_err(c, co):.2e})"
        assert np.array_equal(da, dao)
    else:                    # matrix-sized layers: degree-bucketed MFMA contraction (fma chain)
        assert_close(c, co, 1e-5, "update fwd")
        assert_close(da, dao, 1e-5, "update bwd_a")
    dw = H(ops.duvenaud_update_bwd_w(g, T(up, dev), T(a, dev), mn, mx))
    dwo = oracle.duvenaud_update_bwd_w(up, a, ia, mn, mx)
    assert_close(dw, dwo, 1e-5, "dW")
    # both reverse products from one pass over the gradient rows (one launch at 64 .. 96 features on both sides,
    # the two launches above elsewhere): same partials, and the same bits on a second call
    da2, dw2 = ops.duvenaud_update_bwd(g, T(up, dev), T(a, dev), T(w, dev), mn, mx)
    if Fi < 16 or Fo < 16:
        assert np.array_equal(H(da2), dao)
    else:
        assert_close(H(da2), dao, 1e-5, "fused reverse: da")
    assert_close(H(dw2), dwo, 1e-5, "fused reverse: dW")
    da3, dw3 = ops.duvenaud_update_bwd(g, T(up, dev), T(a, dev), T(w, dev), mn, mx)
    assert torch.equal(da2, da3) and torch.equal(dw2, dw3)
    if Fi >= 16 and Fo >= 16:            # update + activation + the readout's softmax(R z) in one launch == the two launches
        O = 10
        R = rng.standard_normal(O * Fo).astype(np.float32) * 0.3
        for act in ("sigmoid", "none"):
            z2, p2 = ops.duvenaud_update_act_readout(g, T(a, dev), T(w, dev), mn, mx, Fo, T(R, dev), O, act=act)
            z1 = ops.duvenaud_update_act(g, T(a, dev), T(w, dev), mn, mx, Fo, act=act)
            assert torch.equal(z1, z2)
            zo = oracle.activation(act, co)
            po = oracle.softmax_cols(oracle.matmul(R, zo, O))
            from oracle import oracle64 as o64
            assert_close(H(p2), po, 1e-5, "update+readout p", f64=lambda: o64.softmax_cols(o64.matmul(R, o64.activation(act, o64.duvenaud_update(a, w, ia, mn, mx, Fo)), O)))
            seg = torch.from_numpy(np.array([0, 3, 3, n // 2, n], np.int32)).to(dev)
            p1, out1 = ops.duvenaud_readout(T(R, dev), z1, seg, O)
            assert_close(H(p2), H(p1), 1e-5, "p: fused against the readout launch",
                         f64=lambda: o64.softmax_cols(o64.matmul(R, o64.activation(act, o64.duvenaud_update(a, w, ia, mn, mx, Fo)), O)))
            assert_close(H(ops.segment_sum(p2, seg)), oracle.segment_sum(H(p2), H(seg)), 1e-5, "per-graph sums of the fused p")
    for act in ("sigmoid", "relu"):      # activation in the epilogue == update followed by the activation op
        z = H(ops.duvenaud_update_act(g, T(a, dev), T(w, dev), mn, mx, Fo, act=act))
        from oracle import oracle64 as o64   # float64 twin: yardstick of the anchored 1e-5 (helpers.assert_close)
        assert_close(z, oracle.activation(act, co), 1e-5, "update+" + act,
                     f64=lambda: o64.activation(act, o64.duvenaud_update(a, w, ia, mn, mx, Fo)))


def test_softmax_segment_sum_readout(dev, oracle):
    from athena_amd import ops

    rng = np.random.default_rng(8)
    sizes = rng.integers(0, 30, 200)
    seg = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    N, O = int(seg[-1]), 10
    z = rng.standard_normal((N, O)).astype(np.float32) * 2
    p, out = ops.softmax_segsum(T(z, dev), T(seg, dev))
    po = oracle.softmax_cols(z)
    assert_close(H(p), po, 1e-5)
    assert_close(H(out), oracle.segment_sum(po, seg), 1e-5)
    # accumulate over a second "time step" (athena_duvenaud_msgpass_layer.f90:849-852)
    p2, out2 = ops.softmax_segsum(T(z * 0.5, dev), T(seg, dev), out=out.clone())
    assert_close(H(out2), oracle.segment_sum(oracle.softmax_cols(z * 0.5), seg, out=oracle.segment_sum(po, seg)), 1e-5)
    gout = rng.standard_normal((200, O)).astype(np.float32)
    dz = H(ops.softmax_segsum_bwd(p, T(seg, dev), T(gout, dev)))
    gv = np.repeat(gout, sizes, axis=0)
    assert_close(dz, oracle.softmax_cols_bwd(po, gv), 1e-5)


@pytest.mark.parametrize("Fv,O,act,carry", [(64, 10, "sigmoid", True), (64, 10, "none", False), (16, 3, "tanh", True),
                                             (32, 16, "relu", False), (128, 1, "sigmoid", True),
                                             (24, 10, "sigmoid", True), (64, 20, "tanh", False)])
def test_duvenaud_readout_one_launch_matches_the_composed_chain(dev, oracle, Fv, O, act, carry):
    """fused readout (matmul -> softmax -> per-graph sum) and its reverse through the message
    activation, against the oracle's op-by-op chain; ragged graph sizes incl. empty graphs, vertex
    count not a multiple of the 16-vertex tile; (24,*) and (*,20) take the composed fallback"""
    from athena_amd import ops

    rng = np.random.default_rng(80 + Fv + O)
    sizes = rng.integers(0, 40, 300)
    sizes[[0, 17, 299]] = 0
    seg = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    N, S = int(seg[-1]), sizes.size
    assert N % 16 != 0
    z = oracle.activation(act, rng.standard_normal((N, Fv)).astype(np.float32))
    R = (rng.standard_normal(O * Fv) * 0.5).astype(np.float32)
    p, out = ops.duvenaud_readout(T(R, dev), T(z, dev), T(seg, dev), O)
    from oracle import oracle64 as o64       # float64 twin: yardstick of the anchored 1e-5 (helpers.assert_close)
    po = oracle.softmax_cols(oracle.matmul(R, z, O))
    po64 = lambda: o64.softmax_cols(o64.matmul(R, z, O))
    assert_close(H(p), po, 1e-5, "p", f64=po64)
    assert_close(H(out), oracle.segment_sum(po, seg), 1e-5, "out", f64=lambda: o64.segment_sum(po64(), seg))
    _, out2 = ops.duvenaud_readout(T(R, dev), T(z, dev), T(seg, dev), O, out=out.clone())
    assert_close(H(out2), oracle.segment_sum(po, seg, out=oracle.segment_sum(po, seg)), 1e-5, "accumulated out",
                 f64=lambda: 2.0 * o64.segment_sum(po64(), seg))

    gout = rng.standard_normal((S, O)).astype(np.float32)
    dzn = rng.standard_normal((N, Fv)).astype(np.float32) if carry else None
    dc, dR = ops.duvenaud_readout_bwd(T(R, dev), T(z, dev), p, T(seg, dev), T(gout, dev), act=act,
                                      dz_next=T(dzn, dev) if carry else None)
    dl = oracle.softmax_cols_bwd(H(p), np.repeat(gout, sizes, axis=0))
    dz = oracle.matmul_dx(R, dl, Fv)
    if carry:
        dz = dz + dzn
    dl64 = lambda: o64.softmax_cols_bwd(H(p), np.repeat(gout, sizes, axis=0))
    dz64 = lambda: o64.matmul_dx(R, dl64(), Fv) + (dzn if carry else 0.0)
    assert_close(H(dc), oracle.activation_bwd(act, z, dz), 1e-5, "dc", f64=lambda: o64.activation_bwd(act, z, dz64()))
    assert_close(H(dR), oracle.matmul_dw(dl, z), 1e-5, "dR", f64=lambda: o64.matmul_dw(dl64(), z))


def test_errors_surface_as_exceptions_not_aborts(dev):
    from athena_amd import DeviceGraph, _capi

    with pytest.raises(_capi.AthenaMPError, match="adj_ia"):
        DeviceGraph(np.array([0, 1], np.int32), np.array([[1], [0]], np.int32))
    with pytest.raises(_capi.AthenaMPError, match="outside"):
        DeviceGraph(np.array([1, 2], np.int32), np.array([[5], [0]], np.int32))
    # the entry points added later keep the convention: status + message, never an abort
    import ctypes as C
    from athena_amd import ops
    with pytest.raises(_capi.AthenaMPError, match=r"index_list\(:,2\) = \(3, 9\) outside \[1,4\]"):
        DeviceGraph.from_edges(4, np.array([[1, 3], [2, 9]], np.int32))
    g = DeviceGraph.from_edges(4, np.array([[1, 3], [2, 4]], np.int32), add_self_loops=True)
    x = torch.zeros((4, 8), device=dev)
    with pytest.raises(_capi.AthenaMPError, match="unknown activation 9"):
        _capi.call("athena_mp_kipf_propagate_act_fwd", g.handle, 8, C.c_void_p(x.data_ptr()), 9, C.c_void_p(x.data_ptr()))
    with pytest.raises(_capi.AthenaMPError, match="unknown activation 42"):
        _capi.call("athena_mp_activation_param_fwd", 42, 32, 1.0, 0.0, 0.0, C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()))
    with pytest.raises(_capi.AthenaMPError, match="null tensor"):
        _capi.call("athena_mp_kipf_propagate_bwd_dual", g.handle, 8, C.c_void_p(x.data_ptr()), None, None)
    with pytest.raises(ValueError, match="shape"):
        ops.kipf_propagate_bwd_dual(g, torch.zeros((5, 8), device=dev))
    with pytest.raises(ValueError, match="unknown activation"):
        ops.actv_type("mish")


def _gno_case(seed, N, d, H, Fi, Fo, extra_pairs=0, self_loops=False):
    rng = np.random.default_rng(seed)
    pairs = [[i, i + 1] for i in range(1, N)] + [[1, N]]
    for _ in range(extra_pairs):
        a, b = rng.integers(1, N + 1, 2)
        if a != b:
            pairs.append([int(a), int(b)])
    pairs = np.array(pairs).T
    g = csr_from_index_list(N, pairs, self_loops=self_loops)   # self loops carry edge id 0 => zero kernel
    E = pairs.shape[1]
    coords = rng.standard_normal((E, d)).astype(np.float32)
    x = rng.uniform(-1, 1, (N, Fi)).astype(np.float32)
    theta = (0.5 * rng.standard_normal(H * d + H + Fo * Fi * H + Fo * Fi)).astype(np.float32)
    up = rng.uniform(-1, 1, (N, Fo)).astype(np.float32)
    return g, E, coords, x, theta, up


@pytest.mark.parametrize("N,d,H,Fi,Fo,extra", [(9, 3, 5, 4, 3, 2), (20, 1, 8, 3, 3, 0), (300, 3, 16, 8, 8, 600),
                                               (400, 2, 32, 32, 32, 1500), (64, 3, 64, 64, 64, 300), (50, 3, 7, 5, 9, 60)])
def test_gno_reassociated_vs_materialising_oracle(dev, oracle, N, d, H, Fi, Fo, extra):
    """the HIP path never forms kappa [F_out*F_in, E]; the oracle does (athena_diffstruc_extd_sub_nop.f90)"""
    from athena_amd import DeviceGraph, ops

    g, E, coords, x, theta, up = _gno_case(N + H, N, d, H, Fi, Fo, extra)
    ia, ja = g.adj_ia, g.adj_ja
    dg = DeviceGraph(ia, ja, n_edge_cols=E)
    F = Fo * Fi
    kap = oracle.gno_kernel_eval(coords, theta, H, F)
    m_ref = oracle.gno_aggregate(x, kap, ia, ja, Fo)
    th, co, xd, gd = T(theta, dev), T(coords, dev), T(x, dev), T(up, dev)
    m = H_(ops.gno_aggregate(dg, th, co, xd, d, H, Fo))
    assert_close(m, m_ref, 1e-5, "gno fwd")
    dx_ref = oracle.gno_aggregate_bwd_x(up, kap, ia, ja, Fi)
    assert_close(H_(ops.gno_aggregate_bwd_x(dg, th, co, gd, d, H, Fi)), dx_ref, 1e-5, "gno dx")
    dk = oracle.gno_aggregate_bwd_k(up, x, E, ia, ja)
    from oracle import oracle64 as o64       # float64 twin of the materialising oracle: yardstick of the anchored 1e-5
    dk64 = lambda: o64.gno_aggregate_bwd_k(up, x, E, ia, ja)
    assert_close(H_(ops.gno_aggregate_bwd_theta(dg, th, co, xd, gd, d, H)), oracle.gno_kernel_bwd_theta(coords, theta, dk, H), 1e-5,
                 "gno dtheta", f64=lambda: o64.gno_kernel_bwd_theta(coords, theta, dk64(), H))
    assert_close(H_(ops.gno_aggregate_bwd_coords(dg, th, co, xd, gd, d, H)), oracle.gno_kernel_bwd_coords(coords, theta, dk, H), 1e-5,
                 "gno dcoords", f64=lambda: o64.gno_kernel_bwd_coords(coords, theta, dk64(), H))


def H_(t):
    return t.cpu().numpy()


@pytest.mark.parametrize("M,N", [(5000, 1024), (1024, 512), (4099, 4096)])
def test_short_contraction_wide_output_gemm(dev, oracle, M, N):
    """C[M, N] = A[M, 64] . B^T with B stored [N][64] (the GNO 'G = g . Vmat^T' shape): the A-resident kernel
    that walks the column blocks, incl. a ragged last row block; against the oracle's matmul_dx"""
    from athena_amd import ops

    rng = np.random.default_rng(M)
    dz = rng.uniform(-1, 1, (M, 64)).astype(np.float32)
    w = (rng.standard_normal(64 * N) * 0.2).astype(np.float32)            # W(Fo = 64, Fi = N) column-major
    got = H_(ops.matmul_dx(T(w, dev), T(dz, dev), N))
    ref = oracle.matmul_dx(w, dz, N)
    assert got.shape == (M, N)
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()


@pytest.mark.parametrize("d,loops", [(3, False), (2, True), (4, False)])
def test_gno_fused_kernel_hub_rows_tail_tiles_and_self_loops(dev, oracle, d, loops):
    """the one-launch GNO aggregate (H = 64, widths 64): vertices with more than 64 entries (ids beyond the
    prefetched block), a vertex count that is not a multiple of the 16-vertex tile, isolated vertices, and
    self-loop entries without an edge id (zero kernel); forward and dx against the materialising oracle"""
    from athena_amd import DeviceGraph, ops

    rng = np.random.default_rng(40 + d)
    N = 203
    pairs = [[i, i + 1] for i in range(1, N - 5)]                 # the last vertices stay isolated
    pairs += [[7, int(v)] for v in rng.choice(np.arange(9, N - 5), 150, replace=False)]     # hub: 150+ entries
    pairs += [[60, int(v)] for v in rng.choice(np.arange(62, N - 5), 70, replace=False)]    # 70+ entries
    pairs += [[int(a), int(b)] for a, b in rng.integers(1, N - 4, (300, 2)) if a != b]
    pairs = np.array(pairs).T
    g = csr_from_index_list(N, pairs, self_loops=loops)
    E = pairs.shape[1]
    assert np.diff(g.adj_ia).max() > 128 and N % 16 != 0
    Hh = Fi = Fo = 64
    coords = rng.standard_normal((E, d)).astype(np.float32)
    x = rng.uniform(-1, 1, (N, Fi)).astype(np.float32)
    theta = (0.3 * rng.standard_normal(Hh * d + Hh + Fo * Fi * Hh + Fo * Fi)).astype(np.float32)
    up = rng.uniform(-1, 1, (N, Fo)).astype(np.float32)
    dg = DeviceGraph(g.adj_ia, g.adj_ja, n_edge_cols=E)
    kap = oracle.gno_kernel_eval(coords, theta, Hh, Fo * Fi)
    th, co = T(theta, dev), T(coords, dev)
    m = ops.gno_aggregate(dg, th, co, T(x, dev), d, Hh, Fo)
    assert_close(H_(m), oracle.gno_aggregate(x, kap, g.adj_ia, g.adj_ja, Fo), 1e-5, "fused gno fwd")
    assert not H_(m)[N - 4:].any()                                # isolated vertices: exact zeros
    dx = ops.gno_aggregate_bwd_x(dg, th, co, T(up, dev), d, Hh, Fi)
    assert_close(H_(dx), oracle.gno_aggregate_bwd_x(up, kap, g.adj_ia, g.adj_ja, Fi), 1e-5, "fused gno dx")
    assert torch_equal_twice(lambda: ops.gno_aggregate(dg, th, co, T(x, dev), d, Hh, Fo))   # deterministic


@pytest.mark.parametrize("d,loops", [(3, False), (2, True), (1, False)])
def test_gno_fused_kernels_many_tiles_per_workgroup(dev, oracle, d, loops):
    """the producer / consumer GNO kernels (aggregate, dx, and the S^T g half of dtheta) on a graph whose 32-vertex
    tiles outnumber the workgroups' tile classes (several tiles per workgroup, id queue in steady state), with rows
    of 33 .. 150 entries (blocks beyond the 32 prefetched ones), isolated vertices and a ragged last tile; forward,
    dx, dtheta and dcoords against the materialising oracle"""
    from athena_amd import DeviceGraph, ops
    from oracle import oracle64 as o64

    rng = np.random.default_rng(70 + d)
    N = 2101
    pairs = [[i, i + 1] for i in range(1, N - 6)]
    pairs += [[11, int(v)] for v in rng.choice(np.arange(13, N - 6), 150, replace=False)]
    for hub in (400, 900, 1500):
        pairs += [[hub, int(v)] for v in rng.choice(np.arange(hub + 2, N - 6), 33 + hub % 7, replace=False)]
    pairs += [[int(a), int(b)] for a, b in rng.integers(1, N - 5, (2500, 2)) if a != b]
    pairs = np.array(pairs).T
    g = csr_from_index_list(N, pairs, self_loops=loops)
    E = pairs.shape[1]
    deg = np.diff(g.adj_ia)
    assert deg.max() > 128 and ((deg > 32) & (deg < 64)).any() and N % 32 != 0 and N // 32 > 64
    Hh = Fi = Fo = 64
    coords = rng.standard_normal((E, d)).astype(np.float32)
    x = rng.uniform(-1, 1, (N, Fi)).astype(np.float32)
    theta = (0.3 * rng.standard_normal(Hh * d + Hh + Fo * Fi * Hh + Fo * Fi)).astype(np.float32)
    up = rng.uniform(-1, 1, (N, Fo)).astype(np.float32)
    ia, ja = g.adj_ia, g.adj_ja
    dg = DeviceGraph(ia, ja, n_edge_cols=E)
    kap = oracle.gno_kernel_eval(coords, theta, Hh, Fo * Fi)
    th, co, xd, gd = T(theta, dev), T(coords, dev), T(x, dev), T(up, dev)
    assert_close(H_(ops.gno_aggregate(dg, th, co, xd, d, Hh, Fo)), oracle.gno_aggregate(x, kap, ia, ja, Fo), 1e-5, "gno fwd")
    assert_close(H_(ops.gno_aggregate_bwd_x(dg, th, co, gd, d, Hh, Fi)), oracle.gno_aggregate_bwd_x(up, kap, ia, ja, Fi), 1e-5, "gno dx")
    dk = oracle.gno_aggregate_bwd_k(up, x, E, ia, ja)
    dk64 = lambda: o64.gno_aggregate_bwd_k(up, x, E, ia, ja)
    dth = ops.gno_aggregate_bwd_theta(dg, th, co, xd, gd, d, Hh)
    assert_close(H_(dth), oracle.gno_kernel_bwd_theta(coords, theta, dk, Hh), 1e-5, "gno dtheta",
                 f64=lambda: o64.gno_kernel_bwd_theta(coords, theta, dk64(), Hh))
    assert torch.equal(dth, ops.gno_aggregate_bwd_theta(dg, th, co, xd, gd, d, Hh))     # deterministic
    assert_close(H_(ops.gno_aggregate_bwd_coords(dg, th, co, xd, gd, d, Hh)), oracle.gno_kernel_bwd_coords(coords, theta, dk, Hh), 1e-5,
                 "gno dcoords", f64=lambda: o64.gno_kernel_bwd_coords(coords, theta, dk64(), Hh))


def test_gno_fused_kernels_rectangular_rows_subset(dev, oracle):
    """the fused GNO kernels on a RECTANGULAR graph (a shard's rows against all columns: gathered features have n_cols
    rows, gradients n_rows): aggregate, dx (n_cols rows out) and dtheta against the oracle on the same graph padded
    with empty rows to square"""
    from athena_amd import DeviceGraph, ops

    rng = np.random.default_rng(91)
    N, d, Hh, Fi, Fo = 1500, 3, 64, 64, 64
    pairs = [[i, i + 1] for i in range(1, N)]
    pairs += [[20, int(v)] for v in rng.choice(np.arange(30, N), 40, replace=False)]
    pairs += [[int(a), int(b)] for a, b in rng.integers(1, N + 1, (2600, 2)) if a != b]
    pairs = np.array(pairs).T
    g = csr_from_index_list(N, pairs)
    E = pairs.shape[1]
    R = 701                                                   # the shard: the first R rows
    ia, ja = g.adj_ia, g.adj_ja
    ia_r = ia[: R + 1].copy()
    ja_r = np.asfortranarray(ja[:, : ia_r[-1] - 1])
    ia_sq = np.concatenate([ia_r, np.full(N - R, ia_r[-1], np.int32)])
    coords = rng.standard_normal((E, d)).astype(np.float32)
    x = rng.uniform(-1, 1, (N, Fi)).astype(np.float32)
    theta = (0.3 * rng.standard_normal(Hh * d + Hh + Fo * Fi * Hh + Fo * Fi)).astype(np.float32)
    up = rng.uniform(-1, 1, (R, Fo)).astype(np.float32)
    up_sq = np.zeros((N, Fo), np.float32); up_sq[:R] = up
    dg = DeviceGraph(ia_r, ja_r, n_cols=N, n_edge_cols=E, row_deg=np.diff(ia_r), col_deg=np.diff(ia))   # (degrees: unused by GNO)
    kap = oracle.gno_kernel_eval(coords, theta, Hh, Fo * Fi)
    th, co, xd, gd = T(theta, dev), T(coords, dev), T(x, dev), T(up, dev)
    m = H_(ops.gno_aggregate(dg, th, co, xd, d, Hh, Fo))
    assert m.shape == (R, Fo)
    assert_close(m, oracle.gno_aggregate(x, kap, ia_sq, ja_r, Fo)[:R], 1e-5, "gno fwd, rectangular")
    dx = H_(ops.gno_aggregate_bwd_x(dg, th, co, gd, d, Hh, Fi))
    assert dx.shape == (N, Fi)
    assert_close(dx, oracle.gno_aggregate_bwd_x(up_sq, kap, ia_sq, ja_r, Fi), 1e-5, "gno dx, rectangular")
    dk = oracle.gno_aggregate_bwd_k(up_sq, x, E, ia_sq, ja_r)
    from oracle import oracle64 as o64
    assert_close(H_(ops.gno_aggregate_bwd_theta(dg, th, co, xd, gd, d, Hh)), oracle.gno_kernel_bwd_theta(coords, theta, dk, Hh), 1e-5,
                 "gno dtheta, rectangular",
                 f64=lambda: o64.gno_kernel_bwd_theta(coords, theta, o64.gno_aggregate_bwd_k(up_sq, x, E, ia_sq, ja_r), Hh))


@pytest.mark.parametrize("d,loops,N", [(3, False, 2101), (2, True, 2101), (1, False, 203), (3, False, 8200)])
def test_gno_training_pair_keeps_s_for_the_reverse_pass(dev, oracle, d, loops, N):
    """athena_mp_gno_aggregate_fwd_save / _bwd_theta_saved: the forward pass keeps S (every piece as it lies in LDS, bias
    rows behind them), the reverse pass's S^T g streams it.  Same bits as the pair that rebuilds S -- m, and dtheta -- and
    dtheta against the materialising oracle; hub rows (blocks beyond the 32 prefetched entries), isolated vertices, a
    ragged last tile, several tiles per workgroup, a reused buffer"""
    from athena_amd import DeviceGraph, ops
    from oracle import oracle64 as o64

    rng = np.random.default_rng(170 + d + N)
    pairs = [[i, i + 1] for i in range(1, N - 6)]
    pairs += [[11, int(v)] for v in rng.choice(np.arange(13, N - 6), 150, replace=False)]
    pairs += [[90, int(v)] for v in rng.choice(np.arange(92, N - 6), 37, replace=False)]
    pairs += [[int(a), int(b)] for a, b in rng.integers(1, N - 5, (int(1.2 * N), 2)) if a != b]
    pairs = np.array(pairs).T
    g = csr_from_index_list(N, pairs, self_loops=loops)
    E = pairs.shape[1]
    Hh = Fi = Fo = 64
    coords = rng.standard_normal((E, d)).astype(np.float32)
    x = rng.uniform(-1, 1, (N, Fi)).astype(np.float32)
    theta = (0.3 * rng.standard_normal(Hh * d + Hh + Fo * Fi * Hh + Fo * Fi)).astype(np.float32)
    up = rng.uniform(-1, 1, (N, Fo)).astype(np.float32)
    ia, ja = g.adj_ia, g.adj_ja
    dg = DeviceGraph(ia, ja, n_edge_cols=E)
    th, co, xd, gd = T(theta, dev), T(coords, dev), T(x, dev), T(up, dev)
    nbytes = ops.gno_saved_bytes(dg, d, Hh, Fi, Fo)
    assert nbytes == 4 * ((N + 31) // 32) * (8 * 32 * 512 + 32 * 64)
    m, s_save = ops.gno_aggregate_save(dg, th, co, xd, d, Hh, Fo)
    assert torch.equal(m, ops.gno_aggregate(dg, th, co, xd, d, Hh, Fo))
    dth = ops.gno_aggregate_bwd_theta(dg, th, co, xd, gd, d, Hh, s_save=s_save)
    assert torch.equal(dth, ops.gno_aggregate_bwd_theta(dg, th, co, xd, gd, d, Hh))
    dk = oracle.gno_aggregate_bwd_k(up, x, E, ia, ja)
    assert_close(H_(dth), oracle.gno_kernel_bwd_theta(coords, theta, dk, Hh), 1e-5, "gno dtheta from the kept S",
                 f64=lambda: o64.gno_kernel_bwd_theta(coords, theta, o64.gno_aggregate_bwd_k(up, x, E, ia, ja), Hh))
    # the buffer of one step serves the next (other features): nothing of the old S survives
    x2 = T(rng.uniform(-1, 1, (N, Fi)).astype(np.float32), dev)
    s_save.fill_(float("nan"))
    m2, s2 = ops.gno_aggregate_save(dg, th, co, x2, d, Hh, Fo, s_save=s_save)
    assert s2.data_ptr() == s_save.data_ptr() and torch.isfinite(s2[: nbytes // 4]).all()
    assert torch.equal(ops.gno_aggregate_bwd_theta(dg, th, co, x2, gd, d, Hh, s_save=s2),
                       ops.gno_aggregate_bwd_theta(dg, th, co, x2, gd, d, Hh))


def test_gno_training_pair_says_which_shapes_it_serves(dev):
    """shapes that do not take the kernels that keep S: zero bytes, and the pair refuses them with a message"""
    import ctypes as C
    from athena_amd import DeviceGraph, ops, _capi

    g, E, coords, x, theta, up = _gno_case(5, 40, 4, 64, 64, 64, 30)          # d = 4: the round-1 kernel
    dg = DeviceGraph(g.adj_ia, g.adj_ja, n_edge_cols=E)
    assert ops.gno_saved_bytes(dg, 4, 64, 64, 64) == 0
    g2, E2, coords2, x2, theta2, up2 = _gno_case(6, 40, 3, 16, 8, 8, 30)      # small widths: generic kernels
    dg2 = DeviceGraph(g2.adj_ia, g2.adj_ja, n_edge_cols=E2)
    assert ops.gno_saved_bytes(dg2, 3, 16, 8, 8) == 0
    with pytest.raises(ValueError, match="does not keep S"):
        ops.gno_aggregate_save(dg2, T(theta2, dev), T(coords2, dev), T(x2, dev), 3, 16, 8)
    buf = torch.zeros(16, device=dev)
    with pytest.raises(_capi.AthenaMPError, match="does not keep S"):
        _capi.call("athena_mp_gno_aggregate_fwd_save", dg2.handle, 3, 16, 8, 8, C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr()),
                   C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr()))


def torch_equal_twice(fn):
    import torch
    return torch.equal(fn(), fn())


@pytest.mark.parametrize("act,F", [("none", 128), ("relu", 128), ("none", 64), ("tanh", 64), ("none", 256), ("sigmoid", 256)])
def test_fused_kipf_layer_kernels(dev, oracle, act, F):
    """one-launch aggregation + dense step: P bit-exact vs the oracle's kipf_propagate, Z and dX within
    1e-5 of the oracle's unfused order (incl. a hub row, zero-degree rows, a ragged tail chunk)"""
    from athena_amd import DeviceGraph, ops

    n = 4133
    ia, ja = random_graph(n, 5 * n, seed=77, self_loops=True, isolated=9)
    ia0, ja0 = ia.copy(), ja.copy()
    # add a hub: vertex 1 gets 700 extra symmetric neighbours
    rng = np.random.default_rng(3)
    extra = rng.integers(2, n - 9, 700)
    rows = np.repeat(np.arange(1, n + 1), np.diff(ia))
    src = np.concatenate([rows, np.ones(700, np.int64), extra]); dst = np.concatenate([ja[0], extra, np.ones(700, np.int64)])
    order = np.argsort(src, kind="stable")
    ia = np.concatenate([[1], 1 + np.cumsum(np.bincount(src - 1, minlength=n))]).astype(np.int32)
    ja = np.zeros((2, src.size), np.int32, order="F"); ja[0] = dst[order]
    x = rng.uniform(-1, 1, (n, F)).astype(np.float32)
    w = (rng.standard_normal(F * F) * np.sqrt(2.0 / F)).astype(np.float32)
    b = rng.standard_normal(F).astype(np.float32)
    dz = rng.uniform(-1, 1, (n, F)).astype(np.float32)
    g = DeviceGraph(ia, ja, n_edge_cols=0)
    P, Z = ops.kipf_layer_fwd(g, T(x, dev), T(w, dev), F, bias=T(b, dev), act=act)
    Po = oracle.kipf_propagate(x, ia, ja)
    assert np.array_equal(H(P)[1:], Po[1:])            # row 0 is the 700-entry hub (segmented sum)
    assert_close(H(P), Po, 1e-5, "P incl. hub")
    assert_close_elementwise(H(P), Po, oracle.kipf_propagate(np.abs(x), ia, ja), 1e-5, "P incl. hub, element-wise")
    assert_close(H(Z), oracle.activation(act, oracle.add_bias_rows(oracle.matmul(w, Po, F), b)), 1e-5, "fused Z")
    # the same graph without the hub goes through the one-launch kernel: P bit-exact everywhere
    g0 = DeviceGraph(ia0, ja0, n_edge_cols=0)
    P0, Z0 = ops.kipf_layer_fwd(g0, T(x, dev), T(w, dev), F, bias=T(b, dev), act=act)
    Po0 = oracle.kipf_propagate(x, ia0, ja0)
    assert np.array_equal(H(P0), Po0)
    assert_close(H(Z0), oracle.activation(act, oracle.add_bias_rows(oracle.matmul(w, Po0, F), b)), 1e-5, "fused Z (one launch)")
    assert_close(H(ops.kipf_layer_bwd_x(g0, T(dz, dev), T(w, dev), F)),
                 oracle.kipf_propagate_bwd(oracle.matmul_dx(w, dz, F), ia0, ja0), 1e-5, "fused dX (one launch)")
    for exact in (False, True):
        dX = H(ops.kipf_layer_bwd_x(g, T(dz, dev), T(w, dev), F, exact=exact))
        ref = oracle.kipf_propagate_bwd(oracle.matmul_dx(w, dz, F), ia, ja, exact=exact)
        assert_close(dX, ref, 1e-5, f"fused dX exact={exact}")
    # rows of 17..40 entries exercise the second index block of the 16-lane (F = 64) mapping
    ia3, ja3 = random_graph(n, 12 * n, seed=78, self_loops=True)
    g3 = DeviceGraph(ia3, ja3, n_edge_cols=0)
    P3, Z3 = ops.kipf_layer_fwd(g3, T(x, dev), T(w, dev), F, act=act)
    Po3 = oracle.kipf_propagate(x, ia3, ja3)
    assert np.array_equal(H(P3), Po3)
    assert_close(H(Z3), oracle.activation(act, oracle.matmul(w, Po3, F)), 1e-5, "fused Z (dense rows)")
    if F == 256:   # rows of 65..160 entries: the second 64-entry index block of the wave-per-row mapping (W in registers)
        ia4, ja4 = random_graph(1500, 50 * 1500, seed=79, self_loops=True)
        assert np.diff(ia4).max() > 64
        g4 = DeviceGraph(ia4, ja4, n_edge_cols=0)
        P4, Z4 = ops.kipf_layer_fwd(g4, T(x[:1500], dev), T(w, dev), F, act=act)
        Po4 = oracle.kipf_propagate(x[:1500], ia4, ja4)
        assert np.array_equal(H(P4), Po4)
        assert_close(H(Z4), oracle.activation(act, oracle.matmul(w, Po4, F)), 1e-5, "fused Z (64+ entry rows)")
        assert_close(H(ops.kipf_layer_bwd_x(g4, T(dz[:1500], dev), T(w, dev), F)),
                     oracle.kipf_propagate_bwd(oracle.matmul_dx(w, dz[:1500], F), ia4, ja4), 1e-5, "fused dX (64+ entry rows)")
    # other widths take the two-kernel route behind the same entry points
    x2 = rng.uniform(-1, 1, (n, 64)).astype(np.float32); w2 = rng.standard_normal(64 * 32).astype(np.float32)
    P2, Z2 = ops.kipf_layer_fwd(g, T(x2, dev), T(w2, dev), 32)
    assert np.array_equal(H(P2)[1:], oracle.kipf_propagate(x2, ia, ja)[1:])
    assert_close(H(P2), oracle.kipf_propagate(x2, ia, ja), 1e-5)
    assert_close(H(Z2), oracle.matmul(w2, H(P2), 32), 1e-5)
    dz2 = rng.uniform(-1, 1, (n, 32)).astype(np.float32)
    assert_close(H(ops.kipf_layer_bwd_x(g, T(dz2, dev), T(w2, dev), 64)),
                 oracle.kipf_propagate_bwd(oracle.matmul_dx(w2, dz2, 64), ia, ja), 1e-5)


@pytest.mark.parametrize("F,exact", [(128, False), (128, True), (64, False), (24, True)])
def test_kipf_layer_reverse_pass_with_dw_from_the_step_input(dev, oracle, F, exact):
    """athena_mp_kipf_layer_bwd: dX and dW of one step from its INPUT x (no stored P) -- both sums from one gather of dZ
    (128 -> 128: the one-launch kernel of fused_dw.hip; other widths and hub graphs: dual gather + two contractions).
    Against the reference order: dW = dZ . kipf_propagate(x)^T, dX = scatter(W^T dZ) (athena_kipf_msgpass_layer.f90:951,
    athena_diffstruc_extd_sub_kipf.f90:85-111)"""
    from athena_amd import DeviceGraph, ops
    from oracle import oracle64 as o64

    rng = np.random.default_rng(300 + F)
    n = 4133
    ia, ja = random_graph(n, 5 * n, seed=81, self_loops=True, isolated=9)
    x = rng.uniform(-1, 1, (n, F)).astype(np.float32)
    w = (rng.standard_normal(F * F) * np.sqrt(2.0 / F)).astype(np.float32)
    dz = rng.uniform(-1, 1, (n, F)).astype(np.float32)

    def check(ia, ja, tag):
        g = DeviceGraph(ia, ja, n_edge_cols=0)
        dX, dW = ops.kipf_layer_bwd(g, T(dz, dev), T(w, dev), T(x, dev), exact=exact)
        dx_ref = oracle.kipf_propagate_bwd(oracle.matmul_dx(w, dz, F), ia, ja, exact=exact)
        assert_close(H(dX), dx_ref, 1e-5, f"{tag}: dX",
                     f64=lambda: o64.kipf_propagate_bwd(o64.matmul_dx(w, dz, F), ia, ja, exact=exact))
        dw_ref = oracle.matmul_dw(dz, oracle.kipf_propagate(x, ia, ja))
        assert_close(H(dW), dw_ref, 1e-5, f"{tag}: dW", f64=lambda: o64.matmul_dw(dz, o64.kipf_propagate(x, ia, ja)))
        _, dW2 = ops.kipf_layer_bwd(g, T(dz, dev), T(w, dev), T(x, dev), exact=exact, need_dx=False)
        assert torch.equal(dW, dW2)                       # same weight gradient without the input gradient
        d3, w3 = ops.kipf_layer_bwd(g, T(dz, dev), T(w, dev), T(x, dev), exact=exact)
        assert torch.equal(dX, d3) and torch.equal(dW, w3)  # deterministic (ticket-scheduled chunks, ordered slab sum)
        # forward without a stored P gives the same Z
        P, Z = ops.kipf_layer_fwd(g, T(x, dev), T(w, dev), F)
        none, Z2 = ops.kipf_layer_fwd(g, T(x, dev), T(w, dev), F, keep_P=False)
        assert none is None and torch.equal(Z, Z2)

    check(ia, ja, "random graph")
    # rows of 40..90 entries (second index block of the half-wave mapping) and a hub column (> 512: two-kernel route)
    ia2, ja2 = random_graph(1500, 30 * 1500, seed=82, self_loops=True)
    x, dz = x[:1500], dz[:1500]
    check(ia2, ja2, "dense rows")
    extra = rng.integers(2, 1500, 700)
    rows = np.repeat(np.arange(1, 1501), np.diff(ia2))
    src = np.concatenate([rows, np.ones(700, np.int64), extra]); dst = np.concatenate([ja2[0], extra, np.ones(700, np.int64)])
    order = np.argsort(src, kind="stable")
    ia3 = np.concatenate([[1], 1 + np.cumsum(np.bincount(src - 1, minlength=1500))]).astype(np.int32)
    ja3 = np.zeros((2, src.size), np.int32, order="F"); ja3[0] = dst[order]
    check(ia3, ja3, "hub vertex")


@pytest.mark.parametrize("Fi,Fo", [(48, 20), (20, 48), (64, 32), (32, 64), (7, 130)])
def test_reverse_step_picks_the_narrower_side_for_the_scatter(dev, oracle, Fi, Fo):
    """kipf_layer_bwd_x / pull_gemm evaluate A^T(dZ W) or (A^T dZ) W -- whichever moves the narrower rows through
    the scatter; both associations agree with the oracle's (the reference's order) to 1e-5, on a rectangular
    row block (the shard shape) with supplied degrees"""
    from athena_amd import DeviceGraph, ops

    rng = np.random.default_rng(Fi * 3 + Fo)
    n_rows, n_cols = 400, 700
    deg = rng.integers(0, 15, n_rows)
    ia = np.concatenate([[1], 1 + np.cumsum(deg)]).astype(np.int32)
    ja = np.zeros((2, int(deg.sum())), np.int32, order="F")
    ja[0] = rng.integers(1, n_cols + 1, ja.shape[1])
    row_deg = (deg + 1).astype(np.int32)
    col_deg = rng.integers(1, 30, n_cols).astype(np.int32)
    g = DeviceGraph(ia, ja, n_cols=n_cols, n_edge_cols=0, row_deg=row_deg, col_deg=col_deg)
    w = (rng.standard_normal(Fo * Fi) * 0.2).astype(np.float32)
    dz = rng.uniform(-1, 1, (n_rows, Fo)).astype(np.float32)
    dx = H(ops.kipf_layer_bwd_x(g, T(dz, dev), T(w, dev), Fi))
    assert_close(dx, oracle.kipf_propagate_bwd(oracle.matmul_dx(w, dz, Fi), ia, ja, n_out=n_cols), 1e-5, "bwd_x")
    # pull form: rows of g list the SOURCES; dX[v] = (sum_w dZ[col w]) W
    dzs = rng.uniform(-1, 1, (n_cols, Fo)).astype(np.float32)
    pulled = H(ops.pull_gemm(g, T(dzs, dev), T(w, dev), Fi))
    dense = np.zeros((n_rows, n_cols))
    for v in range(n_rows):
        for k in range(ia[v] - 1, ia[v + 1] - 1):
            dense[v, ja[0, k] - 1] += 1.0
    assert_close(pulled, dense @ dzs.astype(np.float64) @ w.reshape(Fi, Fo).astype(np.float64).T, 1e-5, "pull_gemm")


def test_host_pointer_variants(dev, oracle):
    """the *_host entry points a Fortran caller holding array_type%val would use (phase-1 staging)"""
    import ctypes as C

    from athena_amd import DeviceGraph, _capi

    def P_(a):
        return a.ctypes.data_as(C.c_void_p)

    t = golden("reference_test_topologies.json")["network_5v6e"]
    g = csr_from_index_list(5, t["index_list"], self_loops=True)
    dg = DeviceGraph(g.adj_ia, g.adj_ja, n_edge_cols=6)
    x = np.ascontiguousarray(np.array(t["vertex_features_rows"], np.float32).T)
    e = np.ascontiguousarray(np.array(t["edge_features_rows"], np.float32).T)
    c = np.empty((5, 10), np.float32)
    _capi.call("athena_mp_duvenaud_propagate_fwd_host", dg.handle, 8, 2, P_(x), P_(e), P_(c))
    assert np.array_equal(c, oracle.duvenaud_propagate(x, e, g.adj_ia, g.adj_ja))
    w = np.random.default_rng(0).standard_normal(4 * 10 * 3).astype(np.float32)
    u = np.empty((5, 4), np.float32)
    _capi.call("athena_mp_duvenaud_update_fwd_host", dg.handle, 10, 4, 2, 4, P_(c), P_(w), P_(u))
    assert np.array_equal(u, oracle.duvenaud_update(c, w, g.adj_ia, 2, 4, 4))
    up = np.random.default_rng(1).standard_normal((5, 4)).astype(np.float32)
    dw = np.empty_like(w)
    _capi.call("athena_mp_duvenaud_update_bwd_w_host", dg.handle, 10, 4, 2, 4, P_(up), P_(c), P_(dw))
    assert_close(dw, oracle.duvenaud_update_bwd_w(up, c, g.adj_ia, 2, 4), 1e-5)
    seg = np.array([0, 2, 5], np.int32)
    p = np.empty_like(u); out = np.empty((2, 4), np.float32)
    _capi.call("athena_mp_softmax_segsum_fwd_host", 4, 5, 2, P_(seg), P_(u), P_(p), P_(out), 0)
    assert_close(out, oracle.segment_sum(oracle.softmax_cols(u), seg), 1e-5)
    # Duvenaud composites and shaped activations through host arrays
    z = np.empty((5, 4), np.float32)
    _capi.call("athena_mp_duvenaud_update_act_fwd_host", dg.handle, 10, 4, 2, 4, P_(c), P_(w), 2, P_(z))
    assert_close(z, oracle.activation("sigmoid", oracle.duvenaud_update(c, w, g.adj_ia, 2, 4, 4)), 1e-6)
    R = np.random.default_rng(2).standard_normal(3 * 4).astype(np.float32)
    pr = np.empty((5, 3), np.float32); ro = np.empty((2, 3), np.float32)
    _capi.call("athena_mp_duvenaud_readout_fwd_host", 5, 4, 3, 2, P_(seg), P_(z), P_(R), P_(pr), P_(ro), 0)
    po = oracle.softmax_cols(oracle.matmul(R, z, 3))
    assert_close(pr, po, 1e-6); assert_close(ro, oracle.segment_sum(po, seg), 1e-6)
    go = np.random.default_rng(3).standard_normal((2, 3)).astype(np.float32)
    dcc = np.empty((5, 4), np.float32); dRR = np.empty(12, np.float32)
    _capi.call("athena_mp_duvenaud_readout_bwd_host", 5, 4, 3, 2, P_(seg), P_(z), P_(R), P_(po), P_(go), None, 2, P_(dcc), P_(dRR), 0)
    dl = oracle.softmax_cols_bwd(po, np.repeat(go, [2, 3], axis=0))
    assert_close(dcc, oracle.activation_bwd("sigmoid", z, oracle.matmul_dx(R, dl, 4)), 1e-5)
    assert_close(dRR, oracle.matmul_dw(dl, z), 1e-5)
    sm = np.empty_like(u)
    _capi.call("athena_mp_softmax_fwd_host", 5, 4, P_(u), P_(sm))
    assert_close(sm, oracle.softmax_cols(u), 1e-6)
    sw = np.empty_like(u)
    _capi.call("athena_mp_swish_fwd_host", u.size, 1.0, P_(u), P_(sw))
    assert_close(sw, oracle.swish(u), 1e-6)
    ga, dga = np.empty_like(u), np.empty_like(u)
    _capi.call("athena_mp_activation_param_fwd_host", 6, u.size, 2.0, 0.7, 0.1, P_(u), P_(ga))     # gaussian
    assert_close(ga, oracle.activation_param("gaussian", u, 2.0, 0.7, 0.1), 2e-6)
    _capi.call("athena_mp_activation_param_bwd_host", 5, u.size, 1.0, 1.67326, 1.0507, P_(u), P_(sm), P_(dga))   # selu
    assert_close(dga, oracle.activation_param_bwd("selu", u, sm, 1.0, 1.67326, 1.0507), 2e-6)
    # GNO through host arrays
    gg, E, coords, xx, theta, upg = _gno_case(5, 30, 3, 8, 4, 6, 20)
    d2 = DeviceGraph(gg.adj_ia, gg.adj_ja, n_edge_cols=E)
    m = np.empty((30, 6), np.float32)
    _capi.call("athena_mp_gno_aggregate_fwd_host", d2.handle, 3, 8, 4, 6, P_(theta), P_(coords), P_(xx), P_(m))
    kap = oracle.gno_kernel_eval(coords, theta, 8, 24)
    assert_close(m, oracle.gno_aggregate(xx, kap, gg.adj_ia, gg.adj_ja, 6), 1e-5)
    dth = np.empty_like(theta)
    _capi.call("athena_mp_gno_aggregate_bwd_theta_host", d2.handle, 3, 8, 4, 6, P_(theta), P_(coords), P_(xx), P_(upg), P_(dth))
    dk = oracle.gno_aggregate_bwd_k(upg, xx, E, gg.adj_ia, gg.adj_ja)
    from oracle import oracle64 as o64
    assert_close(dth, oracle.gno_kernel_bwd_theta(coords, theta, dk, 8), 1e-5,
                 f64=lambda: o64.gno_kernel_bwd_theta(coords, theta, o64.gno_aggregate_bwd_k(upg, xx, E, gg.adj_ia, gg.adj_ja), 8))


@pytest.mark.parametrize("seed", range(24))
def test_kipf_fuzz_shapes_and_degree_distributions(dev, oracle, seed):
    """seeded fuzz: vertex counts 1..3000, feature counts 1..200, mean degrees 0..40, optional hubs
    (> 512 entries), zero-degree rows, duplicate entries, rectangular column spaces"""
    from athena_amd import DeviceGraph, ops

    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 3000))
    F = int(rng.choice([1, 2, 3, 5, 8, 17, 32, 64, 100, 128, 132, 200]))
    mean_deg = float(rng.choice([0.0, 0.5, 3.0, 10.0, 40.0]))
    deg = rng.poisson(mean_deg, n).astype(np.int64)
    if seed % 3 == 0 and n > 10:
        deg[rng.integers(0, n, 2)] = rng.integers(513, 1500, 2)       # hubs
    rect = seed % 4 == 1
    n_cols = n + int(rng.integers(1, 500)) if rect else n
    ia = np.concatenate([[1], 1 + np.cumsum(deg)]).astype(np.int32)
    nnz = int(deg.sum())
    ja = np.zeros((2, nnz), np.int32, order="F")
    ja[0] = rng.integers(1, n_cols + 1, nnz)
    row_deg = np.maximum(deg, 1).astype(np.int32)                       # avoid 0**-0.5 = inf in the comparison
    col_deg = rng.integers(1, 50, n_cols).astype(np.int32)
    if not rect:
        col_deg = row_deg
    x = rng.uniform(-1, 1, (n_cols, F)).astype(np.float32)
    gr = rng.uniform(-1, 1, (n, F)).astype(np.float32)
    g = DeviceGraph(ia, ja, n_cols=n_cols, n_edge_cols=0, row_deg=row_deg, col_deg=col_deg)
    y = H(ops.kipf_propagate(g, T(x, dev)))
    yo = oracle.kipf_propagate_rect(x, ia, ja, row_deg, col_deg)
    short = deg <= 512
    assert np.array_equal(y[short], yo[short])
    assert_close(y, yo, 1e-5, "fwd") if nnz else None
    d = H(ops.kipf_propagate_bwd(g, T(gr, dev)))
    do = oracle.kipf_propagate_bwd(gr, ia, ja, n_out=n_cols)
    cdeg = np.bincount(ja[0] - 1, minlength=n_cols)
    assert np.array_equal(d[cdeg <= 512], do[cdeg <= 512])
    if nnz:
        assert_close(d, do, 1e-5, "bwd")


@pytest.mark.parametrize("d,loops,N,hubs,directed", [(3, False, 2101, (150, 37), False), (2, True, 2101, (40,), False), (1, False, 203, (), False),
                                                     (3, False, 8200, (33,), False), (3, False, 2101, (), True), (3, True, 70, (), False)])
def test_gno_reverse_pass_from_one_contraction(dev, oracle, d, loops, N, hubs, directed):
    """athena_mp_gno_aggregate_bwd: dx, dtheta and dcoords from ONE G = g . Vmat^T -- the kernel that holds a piece of G_i in
    LDS also emits every entry's partial of dx ([2][nnz][64]) and a gather over the transposed CSR sums them
    (athena_diffstruc_extd_sub_nop.f90:419-458 features, :235-325 parameters, :137-216 coordinates).  Against the
    materialising oracle and against the separate entry points; rows of 17 .. 32 and of more than 32 entries (the plain
    kernel for the few long rows), self-loop entries without an edge column, isolated vertices, a ragged last tile, a
    DIRECTED graph (the entry point does not assume symmetry), S kept and S rebuilt; deterministic."""
    from athena_amd import DeviceGraph, ops
    from oracle import oracle64 as o64

    rng = np.random.default_rng(300 + d + N)
    pairs = [[i, i + 1] for i in range(1, N - 6)]
    for k, h in enumerate(hubs):
        hub = 11 + 80 * k
        pairs += [[hub, int(v)] for v in rng.choice(np.arange(hub + 2, N - 6), h, replace=False)]
    # (the 8 200-vertex case is there for its 257 tiles of 32 short rows -- a second tile per workgroup: the prefetched ids, the
    # c rows' double-buffered LDS strip -- and keeps its rows short so that the materialising oracle stays cheap)
    pairs += [[int(a), int(b)] for a, b in rng.integers(1, N - 5, (int((3.5 if N < 5000 else 1.2) * N), 2)) if a != b]
    pairs = np.array(pairs).T
    g = csr_from_index_list(N, pairs, self_loops=loops)
    E = pairs.shape[1]
    ia, ja = g.adj_ia, g.adj_ja
    if directed:       # keep only the entries that point "upwards" (+ a few back): rows and columns differ
        rows = np.repeat(np.arange(1, N + 1), np.diff(ia))
        keep = (ja[0] > rows) | (rng.random(ja.shape[1]) < 0.2)
        ja = np.asfortranarray(ja[:, keep])
        ia = np.concatenate([[1], 1 + np.cumsum(np.bincount(rows[keep] - 1, minlength=N))]).astype(np.int32)
    Hh = Fi = Fo = 64
    coords = rng.standard_normal((E, d)).astype(np.float32)
    x = rng.uniform(-1, 1, (N, Fi)).astype(np.float32)
    theta = (0.3 * rng.standard_normal(Hh * d + Hh + Fo * Fi * Hh + Fo * Fi)).astype(np.float32)
    up = rng.uniform(-1, 1, (N, Fo)).astype(np.float32)
    dg = DeviceGraph(ia, ja, n_edge_cols=E)
    th, co, xd, gd = T(theta, dev), T(coords, dev), T(x, dev), T(up, dev)
    dx, dth, dc, fused = ops.gno_aggregate_bwd(dg, th, co, xd, gd, d, Hh, need_dcoords=True)
    assert fused
    kap = oracle.gno_kernel_eval(coords, theta, Hh, Fo * Fi)
    assert_close(H_(dx), oracle.gno_aggregate_bwd_x(up, kap, ia, ja, Fi), 1e-5, "gno dx (one contraction)")
    dk = oracle.gno_aggregate_bwd_k(up, x, E, ia, ja)
    dk64 = lambda: o64.gno_aggregate_bwd_k(up, x, E, ia, ja)
    assert_close(H_(dth), oracle.gno_kernel_bwd_theta(coords, theta, dk, Hh), 1e-5, "gno dtheta (one contraction)",
                 f64=lambda: o64.gno_kernel_bwd_theta(coords, theta, dk64(), Hh))
    assert_close(H_(dc), oracle.gno_kernel_bwd_coords(coords, theta, dk, Hh), 1e-5, "gno dcoords (one contraction)",
                 f64=lambda: o64.gno_kernel_bwd_coords(coords, theta, dk64(), Hh))
    # the parameter gradient is the SAME launches as the separate entry point's: same bits; dx associates differently
    assert torch.equal(dth, ops.gno_aggregate_bwd_theta(dg, th, co, xd, gd, d, Hh))
    dx_sep = ops.gno_aggregate_bwd_x(dg, th, co, gd, d, Hh, Fi)
    assert (dx - dx_sep).abs().max().item() <= 1e-5 * dx_sep.abs().max().item()
    # S kept by the forward pass: the same dtheta bits, the same dx bits
    _, s_save = ops.gno_aggregate_save(dg, th, co, xd, d, Hh, Fo)
    dx2, dth2, _, fused2 = ops.gno_aggregate_bwd(dg, th, co, xd, gd, d, Hh, s_save=s_save)
    assert fused2 and torch.equal(dx2, dx) and torch.equal(dth2, dth)
    # dx alone / dtheta alone
    dx3, none, _, _ = ops.gno_aggregate_bwd(dg, th, co, xd, gd, d, Hh, need_dtheta=False)
    assert none is None and torch.equal(dx3, dx)
    none, dth3, _, f3 = ops.gno_aggregate_bwd(dg, th, co, xd, gd, d, Hh, need_dx=False)
    assert none is None and not f3 and torch.equal(dth3, dth)


_BWD_DIGEST = """
import sys, hashlib, numpy as np, torch
sys.path.insert(0, {root!r})
from athena_amd import DeviceGraph, ops, synth
ia, ja, c3 = synth.radius_graph({N})
rng = np.random.default_rng(5)
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(a).to(dev)
g = DeviceGraph(ia, ja, n_edge_cols=c3.shape[0])
x = T(rng.uniform(-1, 1, ({N}, 64)).astype(np.float32)); co = T(np.ascontiguousarray(c3))
th = T((0.3 * rng.standard_normal(64 * 3 + 64 + 64 * 64 * 64 + 64 * 64)).astype(np.float32))
up = T(rng.uniform(-1, 1, ({N}, 64)).astype(np.float32))
_, keep = ops.gno_aggregate_save(g, th, co, x, 3, 64, 64)
def run():
    poison = torch.full(({N}, 64), float("nan"), device=dev); del poison          # the block torch.empty hands out next
    dx, dth, dc, fused = ops.gno_aggregate_bwd(g, th, co, x, up, 3, 64, s_save=keep, need_dcoords=True)
    assert fused
    got = [t.clone() for t in (dx, dth, dc)]        # read on the caller's stream at once: the second stream must have joined
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(got, (dx, dth, dc))) and all(torch.isfinite(t).all() for t in got)
    return hashlib.sha256(b"".join(t.cpu().numpy().tobytes() for t in got)).hexdigest()
first = run()
assert all(run() == first for _ in range(4))
print("DIGEST", first)
"""


def test_gno_reverse_pass_second_stream_joins_and_changes_no_bit(dev):
    """athena_mp_gno_aggregate_bwd runs the partials' gather on the library's second stream beside S^T g: the caller's stream
    has joined it when the call returns -- outputs read at once, from blocks poisoned just before, are complete -- and two
    processes give the same bits (the pipeline is deterministic whatever the two streams' relative timing)"""
    import subprocess, sys
    prog = _BWD_DIGEST.format(root=ROOT, N=60000)
    out = []
    for _ in range(2):
        r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
        out.append([l for l in r.stdout.splitlines() if l.startswith("DIGEST")][-1])
    assert out[0] == out[1]


def test_gno_reverse_pass_second_stream_is_capturable(dev):
    """the fork / join onto the library's second stream inside athena_mp_gno_aggregate_bwd records into a HIP graph (the
    pattern network.capture_step relies on): the replayed graph refills poisoned output buffers with the eager call's bits"""
    from athena_amd import DeviceGraph, ops, synth

    N = 20000
    ia, ja, c3 = synth.radius_graph(N)
    rng = np.random.default_rng(11)
    g = DeviceGraph(ia, ja, n_edge_cols=c3.shape[0])
    x, co = T(rng.uniform(-1, 1, (N, 64)).astype(np.float32), dev), T(c3, dev)
    th = T((0.3 * rng.standard_normal(64 * 3 + 64 + 64 * 64 * 64 + 64 * 64)).astype(np.float32), dev)
    up = T(rng.uniform(-1, 1, (N, 64)).astype(np.float32), dev)
    _, keep = ops.gno_aggregate_save(g, th, co, x, 3, 64, 64)
    dx_ref, dth_ref, _, fused = ops.gno_aggregate_bwd(g, th, co, x, up, 3, 64, s_save=keep)     # eager: also grows the workspaces
    assert fused
    torch.cuda.synchronize()
    dx_buf, dth_buf = torch.empty_like(dx_ref), torch.empty_like(dth_ref)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            ops.gno_aggregate_bwd(g, th, co, x, up, 3, 64, s_save=keep, dx_out=dx_buf, dtheta_out=dth_buf)
    torch.cuda.current_stream(dev).wait_stream(side)
    for _ in range(3):
        dx_buf.fill_(float("nan")); dth_buf.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(dx_buf, dx_ref) and torch.equal(dth_buf, dth_ref)


def test_gno_reverse_pass_falls_back_outside_the_fused_shapes(dev, oracle):
    """shapes the fused kernels do not serve (d = 4; generic widths; a graph where more than 1 row in 64 is longer than 32
    entries) run the separate entry points behind the same call"""
    from athena_amd import DeviceGraph, ops

    for (N, d, H, Fi, Fo, extra) in ((40, 4, 64, 64, 64, 30), (50, 3, 7, 5, 9, 60), (64, 3, 64, 64, 64, 1800)):
        g, E, coords, x, theta, up = _gno_case(N + d, N, d, H, Fi, Fo, extra)
        dg = DeviceGraph(g.adj_ia, g.adj_ja, n_edge_cols=E)
        th, co, xd, gd = T(theta, dev), T(coords, dev), T(x, dev), T(up, dev)
        dx, dth, dc, fused = ops.gno_aggregate_bwd(dg, th, co, xd, gd, d, H, need_dcoords=True)
        assert not fused
        assert torch.equal(dx, ops.gno_aggregate_bwd_x(dg, th, co, gd, d, H, Fi))
        assert torch.equal(dth, ops.gno_aggregate_bwd_theta(dg, th, co, xd, gd, d, H))
        assert torch.equal(dc, ops.gno_aggregate_bwd_coords(dg, th, co, xd, gd, d, H))


@pytest.mark.parametrize("Fv,Fe,Fo", [(64, 8, 64), (64, 16, 80), (32, 4, 48), (64, 8, 24), (6, 1, 7)])
def test_duvenaud_update_bwd_split_matches_the_packed_form(dev, oracle, Fv, Fe, Fo):
    """athena_mp_duvenaud_update_bwd_split: da written as da_x [n, Fv] + da_e [n, Fe] where it is produced (one launch at
    F_v = 64 and the fused kernel's widths, the packed form + two strided copies elsewhere) -- the same bits as the packed da,
    and the two propagate partials taken from the halves equal those taken from the packed rows"""
    from athena_amd import DeviceGraph, ops, synth

    rng = np.random.default_rng(Fv + Fe + Fo)
    ia, ja, voff, E = synth.molecule_batch(300, seed=11)
    N = ia.size - 1
    g = DeviceGraph(ia, ja, n_edge_cols=E)
    mn, mx = 1, 10
    T = lambda a: torch.from_numpy(a).to(dev)
    a = T(rng.uniform(-1, 1, (N, Fv + Fe)).astype(np.float32))
    W = T((0.3 * rng.standard_normal(Fo * (Fv + Fe) * (mx - mn + 1))).astype(np.float32))
    gup = T(rng.uniform(-1, 1, (N, Fo)).astype(np.float32))
    da, dW = ops.duvenaud_update_bwd(g, gup, a, W, mn, mx)
    da_x, da_e, dW2 = ops.duvenaud_update_bwd_split(g, gup, a, W, mn, mx, Fv)
    assert torch.equal(da_x, da[:, :Fv]) and torch.equal(da_e, da[:, Fv:]) and torch.equal(dW2, dW)
    assert_close(da_x.cpu().numpy(), oracle.duvenaud_update_bwd_a(gup.cpu().numpy(), W.cpu().numpy(), ia, mn, mx, Fv + Fe)[:, :Fv], 1e-5)
    assert torch.equal(ops.duvenaud_propagate_bwd_x(g, da_x, Fv), ops.duvenaud_propagate_bwd_x(g, da, Fv))
    assert torch.equal(ops.duvenaud_propagate_bwd_e(g, da_e, 0), ops.duvenaud_propagate_bwd_e(g, da, Fv))


@pytest.mark.parametrize("Fv,Fe,O,act,dz", [(64, 8, 10, "sigmoid", True), (64, 8, 10, "sigmoid", False), (64, 4, 1, "relu", True),
                                             (64, 16, 16, "tanh", True), (64, 32, 7, "none", False), (64, 8, 3, "relu", False),
                                             (32, 8, 10, "sigmoid", True), (64, 8, 20, "sigmoid", True), (128, 8, 10, "tanh", False)])
def test_duvenaud_readout_update_bwd_one_call(dev, oracle, Fv, Fe, O, act, dz):
    """athena_mp_duvenaud_readout_update_bwd: the readout's reverse and the update's reverse of one time step in one call -- ONE
    launch (dc never in HBM) at F_v = 64, F_v + F_e <= 96, O <= 16, the two launches through a workspace elsewhere (the last three
    cases).  da_x / da_e have the bits of athena_mp_duvenaud_readout_bwd + athena_mp_duvenaud_update_bwd_split (the same dc, the same
    products), dW / dR differ by the order of their sums only; everything against the oracle composed op by op."""
    from athena_amd import DeviceGraph, ops, synth

    rng = np.random.default_rng(Fv + 3 * Fe + 5 * O + (7 if dz else 0))
    ia, ja, voff, E = synth.molecule_batch(400, seed=5)      # 7 000 vertices: several tiles in the big buckets, one in the small
    N, S = ia.size - 1, voff.size - 1
    g = DeviceGraph(ia, ja, n_edge_cols=E)
    mn, mx = 1, 10
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    Fc = Fv + Fe
    a_h = rng.uniform(-1, 1, (N, Fc)).astype(np.float32)
    W_h = (0.3 * rng.standard_normal(Fv * Fc * (mx - mn + 1))).astype(np.float32)
    R_h = (0.3 * rng.standard_normal(O * Fv)).astype(np.float32)
    gout_h = rng.standard_normal((S, O)).astype(np.float32)
    dzn_h = rng.standard_normal((N, Fv)).astype(np.float32) if dz else None
    a, W, R, gout, seg = T(a_h), T(W_h), T(R_h), T(gout_h), T(voff)
    dzn = T(dzn_h) if dz else None
    if O <= 16:
        z, p = ops.duvenaud_update_act_readout(g, a, W, mn, mx, Fv, R, O, act=act)
    else:
        z = ops.duvenaud_update_act(g, a, W, mn, mx, Fv, act=act)
        p, _ = ops.duvenaud_readout(R, z, seg, O)
    dc, dR2 = ops.duvenaud_readout_bwd(R, z, p, seg, gout, act=act, dz_next=dzn)
    da_x2, da_e2, dW2 = ops.duvenaud_update_bwd_split(g, dc, a, W, mn, mx, Fv)
    da_x, da_e, dW, dR = ops.duvenaud_readout_update_bwd(g, R, z, p, seg, gout, a, W, mn, mx, Fv, act=act, dz_next=dzn)
    assert torch.equal(da_x, da_x2) and torch.equal(da_e, da_e2)
    assert_close(dW.cpu().numpy(), dW2.cpu().numpy(), 1e-5)
    assert_close(dR.cpu().numpy(), dR2.cpu().numpy(), 1e-5)
    # accumulate_dR: added to what the caller holds
    base = torch.full_like(dR, 0.5)
    _, _, _, dR3 = ops.duvenaud_readout_update_bwd(g, R, z, p, seg, gout, a, W, mn, mx, Fv, act=act, dz_next=dzn, dR=base.clone())
    assert_close((dR3 - 0.5).cpu().numpy(), dR.cpu().numpy(), 1e-5)
    # accumulate_da_e: the edge part of da is added to the caller's buffer (one fp32 add per element), da_x as before
    base_e = T(rng.uniform(-1, 1, (N, Fe)).astype(np.float32))
    da_x4, da_e4, _, _ = ops.duvenaud_readout_update_bwd(g, R, z, p, seg, gout, a, W, mn, mx, Fv, act=act, dz_next=dzn, da_e=base_e.clone())
    assert torch.equal(da_x4, da_x) and torch.equal(da_e4, base_e + da_e)
    # the oracle, op by op, from the device's z and p
    z_h, p_h = z.cpu().numpy(), p.cpu().numpy()
    dl_h = oracle.softmax_cols_bwd(p_h, np.repeat(gout_h, np.diff(voff), axis=0))
    dzt = oracle.matmul_dx(R_h, dl_h, Fv)
    if dz:
        dzt = dzt + dzn_h
    dc_h = oracle.activation_bwd(act, z_h, dzt)
    da_h = oracle.duvenaud_update_bwd_a(dc_h, W_h, ia, mn, mx, Fc)
    assert_close(da_x.cpu().numpy(), da_h[:, :Fv], 1e-5)
    assert_close(da_e.cpu().numpy(), da_h[:, Fv:], 1e-5)
    # parameter gradients are sums over 7 000 vertices: 1e-5, anchored on the float64 twin of the same formulas (helpers.assert_close)
    from oracle import oracle64 as o64

    def hi():
        dl64 = o64.softmax_cols_bwd(p_h, np.repeat(gout_h, np.diff(voff), axis=0))
        dzt64 = o64.matmul_dx(R_h, dl64, Fv)
        if dz:
            dzt64 = dzt64 + dzn_h
        return dl64, o64.activation_bwd(act, z_h, dzt64)

    assert_close(dW.cpu().numpy(), oracle.duvenaud_update_bwd_w(dc_h, a_h, ia, mn, mx), 1e-5, "dW",
                 f64=lambda: o64.duvenaud_update_bwd_w(hi()[1], a_h, ia, mn, mx))
    assert_close(dR.cpu().numpy(), oracle.matmul_dw(dl_h, z_h), 1e-5, "dR", f64=lambda: o64.matmul_dw(hi()[0], z_h))


@pytest.mark.parametrize("Fv,Fe,O,act", [(64, 8, 10, "sigmoid"), (64, 32, 16, "tanh"), (64, 4, 3, "relu"), (32, 8, 10, "sigmoid"),
                                         (64, 8, 20, "none"), (128, 8, 5, "sigmoid")])
def test_duvenaud_split_a_matches_the_packed_form(dev, Fv, Fe, O, act):
    """a = [a_x | a_e] kept split (round 5): the two halves of duvenaud_propagate on their own (neighbour_sum,
    duvenaud_propagate_edges), the update + activation + readout launch and the one-call reverse reading them -- the bits of the
    packed form everywhere (one launch at F_v = 64 and the fused kernels' widths, a packed copy in a workspace elsewhere)."""
    from athena_amd import DeviceGraph, ops, synth

    rng = np.random.default_rng(11 * Fv + 3 * Fe + O)
    ia, ja, voff, E = synth.molecule_batch(400, seed=9)
    N, S = ia.size - 1, voff.size - 1
    g = DeviceGraph(ia, ja, n_edge_cols=E)
    mn, mx = 1, 10
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    Fc = Fv + Fe
    x, e = T(rng.uniform(-1, 1, (N, Fv)).astype(np.float32)), T(rng.uniform(-1, 1, (E, Fe)).astype(np.float32))
    W = T((0.3 * rng.standard_normal(Fv * Fc * (mx - mn + 1))).astype(np.float32))
    R = T((0.3 * rng.standard_normal(O * Fv)).astype(np.float32))
    gout, seg = T(rng.standard_normal((S, O)).astype(np.float32)), T(voff)
    dzn = T(rng.standard_normal((N, Fv)).astype(np.float32))
    a = ops.duvenaud_propagate(g, x, e)
    a_x, a_e = ops.neighbour_sum(g, x), ops.duvenaud_propagate_edges(g, e)
    assert torch.equal(a_x, a[:, :Fv]) and torch.equal(a_e, a[:, Fv:])
    if O <= 16:
        z, p = ops.duvenaud_update_act_readout(g, a, W, mn, mx, Fv, R, O, act=act)
        z2, p2 = ops.duvenaud_update_act_readout_split(g, a_x, a_e, W, mn, mx, Fv, R, O, act=act)
        assert torch.equal(z, z2) and torch.equal(p, p2)
    else:
        z = ops.duvenaud_update_act(g, a, W, mn, mx, Fv, act=act)
        p, _ = ops.duvenaud_readout(R, z, seg, O)
    one = ops.duvenaud_readout_update_bwd(g, R, z, p, seg, gout, a, W, mn, mx, Fv, act=act, dz_next=dzn)
    two = ops.duvenaud_readout_update_bwd(g, R, z, p, seg, gout, a_x, W, mn, mx, Fv, act=act, dz_next=dzn, a_e=a_e)
    for u, v in zip(one, two):
        assert torch.equal(u, v)


@pytest.mark.parametrize("n,band,max_len", [(1, 0, 1), (127, 5, 8), (128, 32, 8), (129, 32, 3), (1000, 31, 8), (5000, 17, 6), (3001, 33, 4),
                                            (3001, 12, 9), (70000, 29, 8)])
def test_banded_gather_matches_the_oracle(dev, oracle, n, band, max_len):
    """csr_gather_banded64 (agg.hip): graphs whose every neighbour lies within 32 rows and whose rows hold at most 8 entries take
    the LDS-staged gather at F = 64 -- forward neighbour sums and the reverse pull over the transposed CSR, bit for bit the
    oracle's CSR-order sums; a wider band (33) or a longer row (9) takes the general kernels, same bits.  Sizes on both sides of
    the 128-row workgroup, empty rows, the first and last block."""
    from athena_amd import DeviceGraph, ops

    rng = np.random.default_rng(n + 7 * band + max_len)
    rows, cols = [], []
    for v in range(n):
        k = int(rng.integers(0, max_len + 1)) if n > 1 else 1
        lo, hi = max(0, v - band), min(n - 1, v + band)
        c = rng.integers(lo, hi + 1, k)
        if k and v % 97 == 0:
            c[0] = lo                      # the band's edge is reached
        if k and v % 89 == 0:
            c[-1] = hi
        rows += [v] * k; cols += list(c)
    rows, cols = np.asarray(rows, np.int64), np.asarray(cols, np.int64)
    ia = np.concatenate([[1], 1 + np.cumsum(np.bincount(rows, minlength=n))]).astype(np.int32)
    ja = np.zeros((2, rows.size), np.int32, order="F"); ja[0] = cols + 1
    g = DeviceGraph(ia, ja, n_edge_cols=0)
    x = rng.uniform(-1, 1, (n, 64)).astype(np.float32)
    xd = torch.from_numpy(x).to(dev)
    y = ops.neighbour_sum(g, xd)
    e0 = np.zeros((1, 1), np.float32)
    want = oracle.duvenaud_propagate(x, e0, ia, ja)[:, :64] if rows.size else np.zeros_like(x)
    assert np.array_equal(y.cpu().numpy(), want)
    gup = rng.uniform(-1, 1, (n, 64)).astype(np.float32)
    dx = ops.duvenaud_propagate_bwd_x(g, torch.from_numpy(gup).to(dev), 64)
    want_dx = oracle.duvenaud_propagate_bwd_x(gup, 64, ia, ja) if rows.size else np.zeros_like(x)
    assert np.array_equal(dx.cpu().numpy(), want_dx)
    # the vertex part of PACKED gradient rows [n, 64 + 8]: the staging loads skip the edge part of every row
    packed = np.concatenate([gup, rng.uniform(-1, 1, (n, 8)).astype(np.float32)], axis=1)
    dx_p = ops.duvenaud_propagate_bwd_x(g, torch.from_numpy(packed).to(dev), 64)
    assert np.array_equal(dx_p.cpu().numpy(), want_dx)


@pytest.mark.parametrize("F", [64, 128])
@pytest.mark.parametrize("n,band,max_len", [(1, 0, 1), (127, 5, 8), (129, 32, 3), (5000, 17, 6), (3001, 33, 4), (3001, 12, 9), (40000, 29, 8)])
def test_banded_kipf_gather_matches_the_oracle(dev, oracle, n, band, max_len, F):
    """round 6: the LDS-staged gather with the per-entry Kipf coefficient, F = 64 and 128 (128: two launches of 64 columns) --
    kipf_propagate forward, its reverse in both forms (the reference's coefficient-free scatter, athena_diffstruc_extd_sub_kipf.f90:85-111,
    and the exact adjoint), bit for bit the oracle's CSR-order sums; the layer step on the same graphs (aggregation from LDS, dense step
    as its own launch) and its reverse to x at 1e-5.  A wider band (33) or a longer row (9) takes the general kernels, same bits."""
    from athena_amd import DeviceGraph, ops
    from oracle import oracle64 as o64

    rng = np.random.default_rng(n + 7 * band + max_len + F)
    rows, cols = [], []
    for v in range(n):
        k = int(rng.integers(0 if n == 3001 else 1, max_len + 1)) if n > 1 else 1     # n = 3001: empty rows too
        lo, hi = max(0, v - band), min(n - 1, v + band)
        c = rng.integers(lo, hi + 1, k)
        if k and v % 97 == 0:
            c[0] = lo
        if k and v % 89 == 0:
            c[-1] = hi
        rows += [v] * k; cols += list(c)
    rows, cols = np.asarray(rows, np.int64), np.asarray(cols, np.int64)
    ia = np.concatenate([[1], 1 + np.cumsum(np.bincount(rows, minlength=n))]).astype(np.int32)
    ja = np.zeros((2, rows.size), np.int32, order="F"); ja[0] = cols + 1
    g = DeviceGraph(ia, ja, n_edge_cols=0)
    x = rng.uniform(-1, 1, (n, F)).astype(np.float32)
    up = rng.uniform(-1, 1, (n, F)).astype(np.float32)
    xd, upd = torch.from_numpy(x).to(dev), torch.from_numpy(up).to(dev)
    # (a neighbour whose own row is empty has degree 0: the reference's coefficient is (deg_v * 0) ** -0.5 = inf there, and inf - inf
    # = NaN in a sum -- the same in the oracle and on the device, hence equal_nan)
    same = lambda a, b: np.array_equal(a, b, equal_nan=True)
    assert same(ops.kipf_propagate(g, xd).cpu().numpy(), oracle.kipf_propagate(x, ia, ja))
    assert same(ops.kipf_propagate_bwd(g, upd).cpu().numpy(), oracle.kipf_propagate_bwd(up, ia, ja))
    assert same(ops.kipf_propagate_bwd(g, upd, exact=True).cpu().numpy(), oracle.kipf_propagate_bwd(up, ia, ja, exact=True))
    if n == 3001:
        return            # (infinite coefficients: nothing to hold the dense step to)
    # the layer step and its reverse to x on the same graph
    W = (rng.standard_normal(F * F) * np.sqrt(2.0 / F)).astype(np.float32)
    Wd = torch.from_numpy(W).to(dev)
    P, Z = ops.kipf_layer_fwd(g, xd, Wd, F, act="relu")
    p_ref = oracle.kipf_propagate(x, ia, ja)
    assert np.array_equal(P.cpu().numpy(), p_ref)
    assert_close(Z.cpu().numpy(), oracle.activation("relu", oracle.matmul(W, p_ref, F)), 1e-5, "Z",
                 f64=lambda: o64.activation("relu", o64.matmul(W, o64.kipf_propagate(x, ia, ja), F)))
    dX = ops.kipf_layer_bwd_x(g, upd, Wd, F)
    assert_close(dX.cpu().numpy(), oracle.kipf_propagate_bwd(oracle.matmul_dx(W, up, F), ia, ja), 1e-5, "dX",
                 f64=lambda: o64.kipf_propagate_bwd(o64.matmul_dx(W, up, F), ia, ja))
