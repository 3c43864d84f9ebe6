"""CPU: pin the oracle (oracle/athena_oracle.c) before anything trusts it.

1. the reference's own known-answer test (test/test_diffstruc_extd_kipf.f90)
2. the values the survey recorded from the reference
3. an independent float64 dense formulation + adjoint identities / finite differences
"""
import numpy as np
import pytest

from helpers import assert_close, csr_from_index_list, dense_adjacency, golden, kipf_dense, random_graph, rel_err


def test_reference_kat_identity_graph(oracle):
    k = golden("reference_kat_kipf_identity.json")
    ia, ja = np.array(k["adj_ia"], np.int32), np.array(k["adj_ja"], np.int32)
    x = np.array(k["x"], np.float32)
    y = oracle.kipf_propagate(x, ia, ja)
    assert np.abs(y - np.array(k["forward"])).max() <= k["tol_abs"]          # :33-37
    g = oracle.kipf_propagate_bwd(np.ones_like(x), ia, ja)
    assert np.abs(g - np.array(k["grad_of_sum"])).max() <= k["tol_abs"]      # :39-45
    up = np.array(k["upstream"], np.float32)
    assert np.abs(oracle.kipf_propagate_bwd(up, ia, ja) - np.array(k["reverse_partial"])).max() <= k["tol_abs"]
    up2 = np.array(k["second_upstream"], np.float32)
    # partial of reverse_kipf_propagate is kipf_propagate itself (..._sub_kipf.f90:174-176)
    assert np.abs(oracle.kipf_propagate(up2, ia, ja) - np.array(k["forward_partial"])).max() <= k["tol_abs"]


def test_survey_recorded_reference_outputs(oracle):
    s = golden("survey_recorded_path_graph.json")
    ia, ja = np.array(s["graph"]["adj_ia"], np.int32), np.array(s["graph"]["adj_ja"], np.int32)
    x = np.array(s["x"], np.float32)
    tol = s["tol_rel"]
    assert rel_err(oracle.kipf_propagate(x, ia, ja), s["kipf_fwd"]) <= tol
    # backward WITHOUT the coefficient
    assert rel_err(oracle.kipf_propagate_bwd(np.array(s["kipf_bwd_upstream"], np.float32), ia, ja), s["kipf_bwd"]) <= tol
    e = np.array(s["edge_features"], np.float32)
    a = oracle.duvenaud_propagate(x, e, ia, ja)
    assert rel_err(a, s["duvenaud_propagate"]) <= tol
    du = s["duvenaud_update"]
    c = oracle.duvenaud_update(a, np.array(du["weight"], np.float32), ia, du["min_degree"], du["max_degree"], du["num_outputs"])
    assert rel_err(c, du["out"]) <= tol


@pytest.mark.parametrize("name,self_loops", [("kipf_layer_6v8e", False), ("kipf_layer_6v8e", True),
                                             ("network_5v6e", True), ("onnx_gnn_4v5e", False)])
def test_kipf_vs_dense_float64_on_reference_topologies(oracle, name, self_loops):
    t = golden("reference_test_topologies.json")[name]
    g = csr_from_index_list(t["num_vertices"], t["index_list"], self_loops)
    rng = np.random.default_rng(7)
    x = rng.uniform(-1, 1, (t["num_vertices"], 5)).astype(np.float32)
    assert_close(oracle.kipf_propagate(x, g.adj_ia, g.adj_ja), kipf_dense(g.adj_ia, g.adj_ja, x), 2e-6, name)
    # reference backward = A^T g (no coefficient)
    A = dense_adjacency(g.adj_ia, g.adj_ja)
    assert_close(oracle.kipf_propagate_bwd(x, g.adj_ia, g.adj_ja), A.T @ x.astype(np.float64), 2e-6, name)


def test_kipf_random_and_degenerate(oracle):
    ia, ja = random_graph(300, 900, seed=3, self_loops=True, isolated=5)
    x = np.random.default_rng(1).uniform(-1, 1, (300, 7)).astype(np.float32)
    y = oracle.kipf_propagate(x, ia, ja)
    assert_close(y, kipf_dense(ia, ja, x), 2e-6, "random")
    assert np.all(y[-5:] == 0.0)  # zero-degree rows give exactly 0 (..._sub_kipf.f90:31)
    # exact-gradient flag is the true adjoint of the forward: <A x, g> == <x, A^T g>
    g = np.random.default_rng(2).uniform(-1, 1, y.shape).astype(np.float32)
    lhs = float((y.astype(np.float64) * g).sum())
    rhs = float((x.astype(np.float64) * oracle.kipf_propagate_bwd(g, ia, ja, exact=True)).sum())
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs)
    # empty graph
    e_ia = np.ones(1, np.int32)
    assert oracle.kipf_propagate(np.zeros((0, 4), np.float32), e_ia, np.zeros((2, 0), np.int32)).shape == (0, 4)


def test_matmul_family_vs_float64(oracle):
    rng = np.random.default_rng(5)
    N, Fi, Fo = 257, 13, 9
    P = rng.uniform(-1, 1, (N, Fi)).astype(np.float32)
    W = rng.standard_normal(Fo * Fi).astype(np.float32)
    dZ = rng.uniform(-1, 1, (N, Fo)).astype(np.float32)
    Wt = W.reshape(Fi, Fo).astype(np.float64)  # Wt[i,o] = W(o,i)
    assert_close(oracle.matmul(W, P, Fo), P @ Wt, 1e-6)
    assert_close(oracle.matmul_dx(W, dZ, Fi), dZ @ Wt.T, 1e-6)
    assert_close(oracle.matmul_dw(dZ, P).reshape(Fi, Fo), P.astype(np.float64).T @ dZ, 1e-6)


def test_duvenaud_ops_vs_float64_and_adjoints(oracle):
    t = golden("reference_test_topologies.json")["network_5v6e"]
    g = csr_from_index_list(5, t["index_list"], self_loops=True)   # self-loop entries carry edge id 0
    ia, ja = g.adj_ia, g.adj_ja
    x = np.array(t["vertex_features_rows"], np.float32).T.copy()   # [5, 8]
    e = np.array(t["edge_features_rows"], np.float32).T.copy()     # [6, 2]
    a = oracle.duvenaud_propagate(x, e, ia, ja)
    A = dense_adjacency(ia, ja)
    assert_close(a[:, :8], A @ x.astype(np.float64), 1e-6)
    # incidence for edge features: B[v,e] = multiplicity of edge e in row v (id 0 -> none)
    B = np.zeros((5, 6))
    for v in range(5):
        for w in range(ia[v] - 1, ia[v + 1] - 1):
            if ja[1, w] > 0:
                B[v, ja[1, w] - 1] += 1
    assert_close(a[:, 8:], B @ e.astype(np.float64), 1e-6)
    up = np.random.default_rng(0).uniform(-1, 1, a.shape).astype(np.float32)
    assert_close(oracle.duvenaud_propagate_bwd_x(up, 8, ia, ja), A.T @ up[:, :8].astype(np.float64), 1e-6)
    assert_close(oracle.duvenaud_propagate_bwd_e(up, 8, 6, ia, ja), B.T @ up[:, 8:].astype(np.float64), 1e-6)
    # update: bucket index is selector AND divisor
    Fi, Fo, mn, mx = 10, 4, 2, 3
    w = np.random.default_rng(1).standard_normal(Fo * Fi * (mx - mn + 1)).astype(np.float32)
    c = oracle.duvenaud_update(a, w, ia, mn, mx, Fo)
    deg = np.diff(ia)
    ref = np.zeros((5, Fo))
    for v in range(5):
        d = max(mn, min(deg[v], mx)) - mn + 1
        Wd = w[(d - 1) * Fo * Fi: d * Fo * Fi].reshape(Fi, Fo).T.astype(np.float64)  # W(o,i)
        ref[v] = Wd @ (a[v].astype(np.float64) / d)
    assert_close(c, ref, 1e-6)
    gup = np.random.default_rng(2).uniform(-1, 1, c.shape).astype(np.float32)
    da = oracle.duvenaud_update_bwd_a(gup, w, ia, mn, mx, Fi)
    dw = oracle.duvenaud_update_bwd_w(gup, a, ia, mn, mx)
    # adjoint identities of the bilinear map c = U(a, w)
    assert abs((c.astype(np.float64) * gup).sum() - (a.astype(np.float64) * da).sum()) <= 1e-5 * abs((c * gup).sum())
    assert abs((c.astype(np.float64) * gup).sum() - (w.astype(np.float64) * dw).sum()) <= 1e-5 * abs((c * gup).sum())


def test_softmax_segment_sum(oracle):
    rng = np.random.default_rng(3)
    z = rng.standard_normal((11, 10)).astype(np.float32)
    p = oracle.softmax_cols(z)
    z64 = z.astype(np.float64)
    ref = np.exp(z64 - z64.max(1, keepdims=True)); ref /= ref.sum(1, keepdims=True)
    assert_close(p, ref, 1e-6)
    assert np.all(p > 0) and np.all(p < 1) and np.allclose(p.sum(1), 1, atol=1e-6)
    seg = np.array([0, 4, 4, 11], np.int32)  # middle graph is empty
    out = oracle.segment_sum(p, seg)
    assert_close(out, np.stack([ref[0:4].sum(0), np.zeros(10), ref[4:].sum(0)]), 1e-6)
    g = rng.standard_normal(p.shape).astype(np.float32)
    dz = oracle.softmax_cols_bwd(p, g)
    eps = 1e-3
    zp = z.copy(); zp[2, 3] += eps
    zm = z.copy(); zm[2, 3] -= eps
    fd = ((oracle.softmax_cols(zp).astype(np.float64) * g).sum() - (oracle.softmax_cols(zm).astype(np.float64) * g).sum()) / (2 * eps)
    assert abs(fd - dz[2, 3]) < 2e-3 * max(1.0, abs(fd))


def _gno_problem(seed=0, N=9, d=3, H=5, Fi=4, Fo=3):
    rng = np.random.default_rng(seed)
    pairs = np.array([[i, i + 1] for i in range(1, N)] + [[1, N], [2, 5]]).T   # chain + 2 chords
    g = csr_from_index_list(N, pairs)
    E = pairs.shape[1]
    coords = rng.standard_normal((E, d)).astype(np.float32)
    x = rng.uniform(-1, 1, (N, Fi)).astype(np.float32)
    theta = (0.5 * rng.standard_normal(H * d + H + Fo * Fi * H + Fo * Fi)).astype(np.float32)
    return g, coords, x, theta, (d, H, Fi, Fo, E, N)


def test_gno_ops_vs_float64_and_finite_differences(oracle):
    g, coords, x, theta, (d, H, Fi, Fo, E, N) = _gno_problem()
    ia, ja = g.adj_ia, g.adj_ja
    F = Fo * Fi
    kap = oracle.gno_kernel_eval(coords, theta, H, F)
    U = theta[:H * d].reshape(d, H).T.astype(np.float64)
    bu = theta[H * d:H * d + H].astype(np.float64)
    V = theta[H * d + H:H * d + H + F * H].reshape(H, F).T.astype(np.float64)
    bv = theta[H * d + H + F * H:].astype(np.float64)
    hid = np.maximum(coords.astype(np.float64) @ U.T + bu, 0)
    kref = hid @ V.T + bv
    assert_close(kap, kref, 2e-6)
    m = oracle.gno_aggregate(x, kap, ia, ja, Fo)
    mref = np.zeros((N, Fo))
    for i in range(N):
        for w in range(ia[i] - 1, ia[i + 1] - 1):
            K = kref[ja[1, w] - 1].reshape(Fi, Fo).T
            mref[i] += K @ x[ja[0, w] - 1].astype(np.float64)
    assert_close(m, mref, 2e-6)
    up = np.random.default_rng(9).uniform(-1, 1, m.shape).astype(np.float32)
    dx = oracle.gno_aggregate_bwd_x(up, kap, ia, ja, Fi)
    dk = oracle.gno_aggregate_bwd_k(up, x, E, ia, ja)
    dth = oracle.gno_kernel_bwd_theta(coords, theta, dk, H)
    dco = oracle.gno_kernel_bwd_coords(coords, theta, dk, H)

    def loss(th, co, xx):
        k = oracle.gno_kernel_eval(co, th, H, F)
        return float((oracle.gno_aggregate(xx, k, ia, ja, Fo).astype(np.float64) * up).sum())

    rng = np.random.default_rng(4)
    eps = 1e-2
    for arr, grad in ((theta, dth), (coords, dco), (x, dx)):
        for _ in range(6):
            idx = tuple(rng.integers(0, s) for s in arr.shape)
            ap, am = arr.copy(), arr.copy()
            ap[idx] += eps; am[idx] -= eps
            args_p = [theta, coords, x]; args_m = [theta, coords, x]
            pos = 0 if arr is theta else (1 if arr is coords else 2)
            args_p[pos], args_m[pos] = ap, am
            fd = (loss(*args_p) - loss(*args_m)) / (2 * eps)
            assert abs(fd - grad[idx]) <= 2e-2 * max(1.0, abs(fd)), (pos, idx, fd, grad[idx])


def test_oracle_has_not_drifted(oracle):
    """tests/golden/oracle_regression.json (self-generated, see make_oracle_regression.py)"""
    r = golden("oracle_regression.json")
    pairs = np.array([[1, 2], [1, 3], [2, 3], [2, 4], [3, 5], [4, 5], [4, 6], [5, 6]]).T
    g = csr_from_index_list(6, pairs, self_loops=True)
    g2 = csr_from_index_list(6, pairs)
    I = {k: np.array(v, np.float32) for k, v in r["inputs"].items()}
    a = oracle.duvenaud_propagate(I["x"], I["e"], g.adj_ia, g.adj_ja)
    c = oracle.duvenaud_update(a, I["w"], g.adj_ia, 2, 5, 3)
    kap = oracle.gno_kernel_eval(I["coords"], I["theta"], 3, 8)
    dk = oracle.gno_aggregate_bwd_k(I["gm"], I["x"], 8, g2.adj_ia, g2.adj_ja)
    got = {"kipf_fwd": oracle.kipf_propagate(I["x"], g.adj_ia, g.adj_ja), "kipf_bwd": oracle.kipf_propagate_bwd(I["x"], g.adj_ia, g.adj_ja),
           "duvenaud_propagate": a, "duvenaud_update": c, "duvenaud_update_bwd_w": oracle.duvenaud_update_bwd_w(c, a, g.adj_ia, 2, 5),
           "gno_aggregate": oracle.gno_aggregate(I["x"], kap, g2.adj_ia, g2.adj_ja, 2),
           "gno_dtheta": oracle.gno_kernel_bwd_theta(I["coords"], I["theta"], dk, 3),
           "gno_dcoords": oracle.gno_kernel_bwd_coords(I["coords"], I["theta"], dk, 3)}
    for k, v in got.items():
        assert np.allclose(v, np.array(r[k], np.float32), rtol=2e-6, atol=1e-7), k


def test_update_step_restatements_against_float64_closed_forms(oracle):
    """clip / SGD / Adam / MSE (athena_clipper.f90:165-210, athena_optimiser.f90:634-673, :1027-1091,
    athena_loss.f90:393-430) against the textbook formulas evaluated in float64"""
    rng = np.random.default_rng(11)
    n = 2000
    p = rng.standard_normal(n).astype(np.float32); g = rng.standard_normal(n).astype(np.float32)
    P, G = p.astype(np.float64), g.astype(np.float64)
    # Adam, two iterations, no regulariser: p -= lr * m_hat / (sqrt(v_hat) + eps)
    lr, b1, b2, eps = 0.01, 0.9, 0.999, 1e-8
    m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    M = np.zeros(n); V = np.zeros(n); Pq = P.copy()
    pq = p.copy()
    for it in (1, 2):
        pq, _, m, v = oracle.adam_step(pq, g, m, v, lr, it)
        M = b1 * M + (1 - b1) * G; V = b2 * V + (1 - b2) * G * G
        Pq = Pq - lr * (M / (1 - b1 ** it)) / (np.sqrt(V / (1 - b2 ** it)) + eps)
        assert np.allclose(pq, Pq, rtol=2e-6, atol=1e-7)
    # first Adam step is lr * sign(g) whatever the scale of g
    assert np.allclose(p - oracle.adam_step(p, g * 1e3, np.zeros(n), np.zeros(n), lr, 1)[0], lr * np.sign(g), rtol=1e-4)
    # AdamW decoupled decay multiplies the parameter by (1 - lr*l2) first (:1069-1072)
    pw = oracle.adam_step(p, g, np.zeros(n), np.zeros(n), lr, 1, reg="l2", l2=0.1, decoupled=True)[0]
    g_reg = G + lr * 2 * 0.1 * P                    # regularise_l2 also touched the gradient (:1045-1046)
    assert np.allclose(pw, P * (1 - lr * 0.1) - lr * np.sign(g_reg), rtol=1e-5, atol=1e-6)
    # SGD with momentum / Nesterov
    vel = rng.standard_normal(n).astype(np.float32)
    ps, gs, vs = oracle.sgd_step(p, g, vel, 0.1, momentum=0.9)
    assert np.allclose(vs, 0.9 * vel - 0.1 * G, rtol=1e-6, atol=1e-7) and np.allclose(ps, P + vs, rtol=1e-6, atol=1e-7)
    assert np.allclose(gs, -0.1 * G, rtol=1e-6)
    pn, _, vn = oracle.sgd_step(p, g, vel, 0.1, momentum=0.9, nesterov=True)
    assert np.allclose(pn, P + 0.9 * vn - 0.1 * G, rtol=1e-5, atol=1e-6)
    p0, _, v0 = oracle.sgd_step(p, g, vel, 0.1)     # momentum 0: velocity is just the step
    assert np.allclose(p0, P - 0.1 * G, rtol=1e-6, atol=1e-7) and np.allclose(v0, -0.1 * G, rtol=1e-6)
    # L1 uses sign(1, p): +1 at +0, -1 at -0 (Fortran SIGN transfers the sign bit)
    z = np.array([0.0, -0.0], np.float32)
    assert np.array_equal(oracle.sgd_step(z, np.zeros(2), np.zeros(2), 1.0, reg="l1", l1=0.5)[1], np.array([-0.5, 0.5], np.float32))
    # clipping
    c = oracle.clip(g * 5, -1.0, 2.0)
    assert c.min() == -1.0 and c.max() == 2.0
    c = oracle.clip(g, clip_norm=3.0)
    assert abs(np.sqrt((c.astype(np.float64) ** 2).sum()) - 3.0) < 1e-4
    assert np.array_equal(oracle.clip(g * 1e-3, clip_norm=3.0), (g * 1e-3).astype(np.float32))
    # MSE
    e = rng.standard_normal(n).astype(np.float32)
    lo, d = oracle.mse(p, e)
    assert abs(lo - ((P - e) ** 2).mean() / 2) < 1e-5 and np.allclose(d, (P - e) / n, rtol=1e-5, atol=1e-9)
    # real ** integer by binary powering
    f = oracle.lib().oracle_powi
    import ctypes as C
    f.restype = C.c_float
    for it in (1, 2, 3, 10, 1000):
        assert abs(f(C.c_float(0.999), C.c_int(it)) - 0.999 ** it) < 1e-5


def test_threaded_context_baseline_equals_the_single_thread_oracle(oracle):
    """athena_oracle_omp.c (bench.py's all-cores context number): same step, rows shared out over threads"""
    rng = np.random.default_rng(21)
    n, F = 700, 16
    ia, ja = random_graph(n, 2500, seed=21, self_loops=True, isolated=4)
    x = rng.uniform(-1, 1, (n, F)).astype(np.float32)
    w = rng.standard_normal(F * F).astype(np.float32) * 0.3
    dz = rng.uniform(-1, 1, (n, F)).astype(np.float32)
    P, Z, dW, dX, dt, threads = oracle.omp_kipf_step(x, w, dz, ia, ja)
    Pr = oracle.kipf_propagate(x, ia, ja)
    assert np.allclose(P, Pr, rtol=1e-6, atol=1e-7)          # numpy's float32 power vs libm powf: <= 1 ulp on coefficients
    assert np.allclose(Z, oracle.matmul(w, Pr, F), rtol=1e-5, atol=1e-6)
    assert np.allclose(dW, oracle.matmul_dw(dz, Pr), rtol=1e-4, atol=1e-5)
    assert np.allclose(dX, oracle.kipf_propagate_bwd(oracle.matmul_dx(w, dz, F), ia, ja), rtol=1e-5, atol=1e-6)
    assert threads >= 1 and dt > 0


def test_attributed_activations_against_the_reference_test_formulas(oracle):
    """test/test_activations.f90:395-480 (compare_output / compare_derivative) states the expected values of the
    activations at their default attributes; the float64 forms below restate those lines."""
    x = np.linspace(-3, 3, 61).astype(np.float32)
    x64 = x.astype(np.float64)
    g = np.ones_like(x)
    # gaussian, sigma 1.5, mu 0 (:403-405, :457-460)
    f = 1.0 / (np.sqrt(8.0 * np.arctan(1.0)) * 1.5) * np.exp(-0.5 * (x64 / 1.5) ** 2)
    assert np.abs(oracle.activation_param("gaussian", x, 1.0, 1.5, 0.0) - f).max() <= 1e-6
    assert np.abs(oracle.activation_param_bwd("gaussian", x, g, 1.0, 1.5, 0.0) - (-x64 / 1.5 ** 2 * f)).max() <= 1e-6
    # leaky_relu / relu / piecewise at a positive input inside the limit return the input (:406-411)
    pos = np.array([0.5], np.float32)
    for kind, p0, p1 in (("leaky_relu", 0.01, 0.0), ("relu", 0.0, 0.0), ("piecewise", 0.1, 1.0), ("linear", 0.0, 0.0)):
        assert oracle.activation_param(kind, pos, 1.0, p0, p1)[0] == pos[0]
    # piecewise derivative is 1 where |x| >= 1 (:465-468); as written in get_partial_piecewise_val it is 1 everywhere
    assert np.array_equal(oracle.activation_param_bwd("piecewise", x, g, 1.0, 0.1, 1.0), g)
    # piecewise forward: slope `gradient` outside +-limit, continuous at the limit (piecewise_array :216-254)
    pw = oracle.activation_param("piecewise", x, 1.0, 0.1, 1.0)
    ref = np.where(x64 >= 1, 0.1 * (x64 - 1) + 1, np.where(x64 <= -1, 0.1 * (x64 + 1) - 1, x64))
    assert np.abs(pw - ref).max() <= 1e-6
    # selu / leaky_relu closed forms and their derivatives by central differences in float64
    al, lam = 1.67326, 1.0507
    selu = np.where(x64 > 0, lam * x64, al * lam * (np.exp(x64) - 1))
    assert np.abs(oracle.activation_param("selu", x, 1.0, al, lam) - selu).max() <= 2e-6
    dselu = np.where(x64 > 0, lam, al * lam * np.exp(x64))
    assert np.abs(oracle.activation_param_bwd("selu", x, g, 1.0, al, lam) - dselu).max() <= 2e-6
    lr = oracle.activation_param("leaky_relu", x, 1.0, 0.01, 0.0)
    assert np.abs(lr - np.where(x64 > 0, x64, 0.01 * x64)).max() <= 1e-7
    nz = np.abs(x) > 1e-6     # the tie of max() at 0 is diffstruc's choice (absent): not pinned
    assert np.allclose(oracle.activation_param_bwd("leaky_relu", x, g, 1.0, 0.01, 0.0)[nz], np.where(x > 0, 1.0, 0.01)[nz])
    # scale multiplies output and gradient; relu threshold clamps from below (athena_activation_relu.f90:201)
    assert np.allclose(oracle.activation_param("relu", x, 2.0, 0.5, 0.0), 2.0 * np.maximum(x, 0.5))
    assert np.allclose(oracle.activation_param_bwd("relu", x, g, 2.0, 0.5, 0.0), np.where(x > 0.5, 2.0, 0.0))
    for kind in ("sigmoid", "tanh"):
        assert np.abs(oracle.activation_param(kind, x, 1.0) - oracle.activation(kind, x)).max() == 0
        assert np.abs(oracle.activation_param_bwd(kind, x, g, 1.0) - oracle.activation_bwd(kind, oracle.activation(kind, x), g)).max() <= 1e-7


def test_float64_twin_of_the_oracle_and_the_anchored_tolerance(oracle):
    """oracle/oracle64.py is the same C source in double precision: it reproduces the reference's identity-graph KAT,
    agrees with the fp32 oracle to fp32 rounding on every op family, and is what helpers.assert_close anchors on"""
    from oracle import oracle64 as o64
    from helpers import assert_close_elementwise

    k = golden("reference_kat_kipf_identity.json")
    ia, ja = np.array(k["adj_ia"], np.int32), np.array(k["adj_ja"], np.int32)
    y = o64.kipf_propagate(np.array(k["x"], np.float32), ia, ja)
    assert y.dtype == np.float64 and np.abs(y - np.array(k["forward"])).max() <= k["tol_abs"]
    rng = np.random.default_rng(5)
    ia, ja = random_graph(300, 1200, seed=9)
    x = rng.uniform(-1, 1, (300, 12)).astype(np.float32)
    w = rng.standard_normal(12 * 7).astype(np.float32)
    p32, p64 = oracle.kipf_propagate(x, ia, ja), o64.kipf_propagate(x, ia, ja)
    assert rel_err(p32, p64) < 1e-6 and rel_err(oracle.matmul(w, p32, 7), o64.matmul(w, p64, 7)) < 1e-6
    assert rel_err(oracle.kipf_propagate_bwd(x, ia, ja), o64.kipf_propagate_bwd(x, ia, ja)) < 1e-6
    assert rel_err(oracle.activation("tanh", x), o64.activation("tanh", x)) < 1e-6
    th = rng.standard_normal(8 * 3 + 8 + 6 * 8 + 6).astype(np.float32); co = rng.standard_normal((40, 3)).astype(np.float32)
    assert rel_err(oracle.gno_kernel_eval(co, th, 8, 6), o64.gno_kernel_eval(co, th, 8, 6)) < 1e-6
    # anchored criterion: passes a result that sits between the oracle and float64, fails one that is simply wrong
    ref64 = np.linspace(1.0, 2.0, 50)
    orc = (ref64 * (1 + 4e-5)).astype(np.float32)                  # an fp32 oracle 4e-5 off exact arithmetic
    assert_close((ref64 * (1 + 1e-6)).astype(np.float32), orc, 1e-5, "anchored", f64=ref64)
    with pytest.raises(AssertionError):
        assert_close((ref64 * (1 + 9e-5)).astype(np.float32), orc, 1e-5, "anchored", f64=ref64)
    with pytest.raises(AssertionError):
        assert_close((ref64 * (1 + 1e-6)).astype(np.float32), orc, 1e-5, "not anchored")
    assert_close_elementwise(np.array([1.0, 1e-9]), np.array([1.0, 2e-9]), np.array([2.0, 1.0]))
    with pytest.raises(AssertionError):
        assert_close_elementwise(np.array([1.0, 1e-3]), np.array([1.0, 2e-3]), np.array([2.0, 1.0]))
