"""GPU: the FORTRAN host side (athena_amd/fortran/athena_mp_layers.f90 -- kipf / duvenaud / graph_nop layer types
with the reference's layer API over the C ABI) against the oracle.  Each test writes a case file, runs
athena_mp_layer_run (built by __graft_entry__.build(), travels with the snapshot) and compares what the Fortran
layer computed -- forward, input / edge gradients, parameter gradients, accessors -- with the per-sample oracle
restatement of the layer (tests/oracle_layers.py).  Tolerance: the north star's 1e-5 relative; gradients
that chain several contractions are anchored on the float64 twin of the oracle (helpers.assert_close, f64=)."""
import os
import subprocess

import numpy as np
import pytest

import functools

import oracle_layers as ol
from helpers import assert_close, csr_from_index_list, rel_err as rel_err_

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNNER = os.path.join(ROOT, "athena_amd", "fortran", "athena_mp_layer_run")


class _actv:
    """what oracle_layers' activation helpers read (name, scale, apply_scaling, p, beta)"""

    def __init__(self, name, scale=1.0, p0=0.0, p1=0.0):
        self.name, self.scale, self.p = name, float(scale), [float(p0), float(p1)]
        self.apply_scaling = abs(np.float32(scale) - np.float32(1)) > np.float32(1e-6)
        self.beta = float(p0) if name == "swish" else 1.0


def _oracle_act(a):
    plain = not a.apply_scaling and a.p == [0.0, 0.0]
    if a.name in ("none", "relu", "sigmoid", "tanh", "softmax") and plain:
        return a.name
    if a.name == "swish" and not a.apply_scaling and a.p[0] == 1.0:
        return "swish"
    return a


def _graphs(rng, sizes, self_loops):
    gs = []
    for n in sizes:
        pairs = [[i, i + 1] for i in range(1, n)]
        for _ in range(n // 2):
            a, b = rng.integers(1, n + 1, 2)
            if a != b:
                pairs.append([int(a), int(b)])
        gs.append(csr_from_index_list(n, np.array(pairs).T, self_loops=self_loops))
    return gs


def _i(*v):
    return np.asarray(v, np.int32).tobytes()


def _mat(a):
    """a is [elements, features] row-major == Fortran (features, elements)"""
    a = np.ascontiguousarray(a, np.float32)
    return _i(a.shape[1], a.shape[0]) + a.tobytes()


def _act_bytes(a):
    return a.name.ljust(16).encode() + np.asarray([a.scale, a.p[0], a.p[1]], np.float32).tobytes()


def _case_header(kind, gs):
    out = _i(kind, len(gs))
    for g in gs:
        out += _i(g.num_vertices, g.num_edges, g.nnz) + np.asarray(g.adj_ia, np.int32).tobytes()
        out += np.asfortranarray(g.adj_ja, dtype=np.int32).tobytes(order="F")
    return out


class _reader:
    def __init__(self, path):
        self.b = open(path, "rb").read()
        self.o = 0

    def ints(self, n):
        v = np.frombuffer(self.b, np.int32, n, self.o); self.o += 4 * n
        return v

    def matrix(self):
        f, n = self.ints(2)
        v = np.frombuffer(self.b, np.float32, int(f) * int(n), self.o).reshape(int(n), int(f)); self.o += 4 * int(f) * int(n)
        return v

    def vector(self):
        n = int(self.ints(1)[0])
        v = np.frombuffer(self.b, np.float32, n, self.o); self.o += 4 * n
        return v


def _run(tmp_path, blob, env=None):
    if not os.path.exists(RUNNER):
        pytest.fail("athena_mp_layer_run is not built: __graft_entry__.build() compiles the Fortran host side")
    case, res = str(tmp_path / "case.bin"), str(tmp_path / "result.bin")
    with open(case, "wb") as f:
        f.write(blob)
    r = subprocess.run([RUNNER, case, res], capture_output=True, text=True, timeout=300, env={**os.environ, **(env or {})})
    assert r.returncode == 0, f"athena_mp_layer_run failed ({r.returncode}): {r.stderr[-2000:]}"
    return _reader(res)


KIPF_CASES = [
    ([5], 1, _actv("none"), 0, 0),
    ([8, 16, 4], 2, _actv("relu"), 0, 0),
    ([64, 128, 64], 2, _actv("tanh"), 1, 0),
    ([48, 12, 20, 5], 3, _actv("swish", p0=1.0), 0, 0),           # auto: steps 1 and 3 run transform-first
    ([48, 12, 20, 5], 3, _actv("sigmoid"), 2, 1),                  # transform-first everywhere, exact reverse
    ([6, 9], 1, _actv("leaky_relu", p0=0.1, scale=1.5), 0, 0),
    ([7, 7, 7], 2, _actv("softmax"), 1, 0),
]


@pytest.mark.parametrize("nvf,T,act,order,exact", KIPF_CASES)
def test_fortran_kipf_layer(dev, tmp_path, nvf, T, act, order, exact):
    rng = np.random.default_rng(len(nvf) * 7 + T + order)
    gs = _graphs(rng, [6, 17, 40, 9], self_loops=True)
    full = nvf * (T + 1) if len(nvf) == 1 else nvf
    plist = [(rng.standard_normal(full[t] * full[t - 1]) * np.sqrt(2.0 / full[t - 1])).astype(np.float32) for t in range(1, T + 1)]
    xs = [rng.uniform(-1, 1, (g.num_vertices, full[0])).astype(np.float32) for g in gs]
    oa = _oracle_act(act)
    outs, tapes = ol.kipf_forward(gs, xs, plist, full, oa)
    ups = [rng.uniform(-1, 1, o.shape).astype(np.float32) for o in outs]
    blob = _case_header(1, gs) + _i(T, len(nvf)) + _i(*nvf) + _act_bytes(act) + _i(order, exact, 1)
    blob += _i(sum(p.size for p in plist)) + np.concatenate(plist).tobytes()
    blob += _mat(np.concatenate(xs)) + _mat(np.concatenate(ups))
    r = _run(tmp_path, blob)
    assert np.all(r.vector() == 0)                                   # get_gradients before a reverse pass: zeros
    out = r.matrix()
    assert_close(out, np.concatenate(outs), 1e-5, "fortran kipf forward")
    dx = r.matrix()
    dxs, grads = ol.kipf_backward(gs, tapes, plist, full, oa, ups, exact=bool(exact))

    @functools.lru_cache(None)
    def hi():   # the same composition on the oracle's float64 twin: yardstick of the anchored 1e-5 (helpers.assert_close)
        with ol.double_precision():
            _, t64 = ol.kipf_forward(gs, xs, plist, full, oa)
            d64, g64 = ol.kipf_backward(gs, t64, plist, full, oa, ups, exact=bool(exact))
        return np.concatenate(d64), np.concatenate(g64)
    assert_close(dx, np.concatenate(dxs), 1e-5, "fortran kipf dx", f64=lambda: hi()[0])
    assert_close(r.vector(), np.concatenate(grads), 1e-5, "fortran kipf dW", f64=lambda: hi()[1])
    assert np.array_equal(r.vector(), np.concatenate(plist))         # get_params returns what set_params stored
    assert np.array_equal(r.matrix(), out)                           # deterministic forward


@pytest.mark.parametrize("act,act_r", [(_actv("sigmoid"), _actv("softmax")), (_actv("tanh"), _actv("sigmoid")),
                                       (_actv("selu", p0=1.67326, p1=1.0507), _actv("softmax")),
                                       (_actv("relu"), _actv("none")), (_actv("gaussian", p0=1.0), _actv("swish", p0=1.0))])
def test_fortran_duvenaud_layer(dev, tmp_path, act, act_r):
    rng = np.random.default_rng(33)
    gs = _graphs(rng, [9, 14, 5, 20, 11], self_loops=False)
    Fv, Fe, T, D, nout = 8, 2, 3, 6, 4
    nvf = [Fv] * (T + 1)
    plist = [(rng.standard_normal(nvf[t] * (nvf[t - 1] + Fe) * D) * 0.3).astype(np.float32) for t in range(1, T + 1)]
    plist += [(rng.standard_normal(nout * nvf[t]) * 0.3).astype(np.float32) for t in range(1, T + 1)]
    xs = [rng.uniform(0, 1, (g.num_vertices, Fv)).astype(np.float32) for g in gs]
    es = [rng.uniform(0, 1, (g.num_edges, Fe)).astype(np.float32) for g in gs]
    oa, oar = _oracle_act(act), _oracle_act(act_r)
    outs, tapes = ol.duvenaud_forward(gs, xs, es, plist, nvf, Fe, 1, D, nout, oa, act_readout=oar)
    up = rng.uniform(-1, 1, outs.shape).astype(np.float32)
    blob = _case_header(2, gs) + _i(T, Fv, Fe, 1, D, nout) + _act_bytes(act) + _act_bytes(act_r)
    blob += _i(sum(p.size for p in plist)) + np.concatenate(plist).tobytes()
    blob += _mat(np.concatenate(xs)) + _mat(np.concatenate(es)) + _mat(up)
    r = _run(tmp_path, blob)
    assert_close(r.matrix(), outs, 1e-5, "fortran duvenaud forward")
    dxs, des, grads = ol.duvenaud_backward(gs, es, tapes, plist, nvf, Fe, 1, D, nout, oa, up, act_readout=oar)

    @functools.lru_cache(None)
    def hi():
        with ol.double_precision():
            _, t64 = ol.duvenaud_forward(gs, xs, es, plist, nvf, Fe, 1, D, nout, oa, act_readout=oar)
            a, b, c = ol.duvenaud_backward(gs, es, t64, plist, nvf, Fe, 1, D, nout, oa, up, act_readout=oar)
        return np.concatenate(a), np.concatenate(b), np.concatenate(c)
    assert_close(r.matrix(), np.concatenate(dxs), 1e-5, "fortran duvenaud dx", f64=lambda: hi()[0])
    assert_close(r.matrix(), np.concatenate(des), 1e-5, "fortran duvenaud de", f64=lambda: hi()[1])
    assert_close(r.vector(), np.concatenate(grads), 1e-5, "fortran duvenaud gradients", f64=lambda: hi()[2])


@pytest.mark.parametrize("Fi,Fo,d,H,bias,act", [(3, 5, 1, 8, 1, _actv("none")), (8, 8, 3, 16, 0, _actv("relu")),
                                               (32, 32, 3, 32, 1, _actv("tanh")), (4, 6, 2, 8, 1, _actv("piecewise", p0=0.3, p1=0.4)),
                                               (64, 64, 3, 64, 1, _actv("relu"))])   # the widths whose forward pass keeps S
def test_fortran_graph_nop_layer(dev, tmp_path, Fi, Fo, d, H, bias, act):
    rng = np.random.default_rng(Fi + H)
    gs = _graphs(rng, [20, 11, 33], self_loops=False)
    F = Fo * Fi
    sizes = [H * d + H + F * H + F, F] + ([Fo] if bias else [])
    plist = [(rng.standard_normal(n) * 0.2).astype(np.float32) for n in sizes]
    xs = [rng.uniform(-1, 1, (g.num_vertices, Fi)).astype(np.float32) for g in gs]
    cs = [rng.standard_normal((g.num_edges, d)).astype(np.float32) for g in gs]
    oa = _oracle_act(act)
    outs, tapes = ol.gno_forward(gs, xs, cs, plist, Fi, Fo, d, H, bool(bias), oa)
    ups = [rng.uniform(-1, 1, o.shape).astype(np.float32) for o in outs]
    blob = _case_header(3, gs) + _i(Fi, Fo, d, H, bias) + _act_bytes(act)
    blob += _i(sum(sizes)) + np.concatenate(plist).tobytes()
    blob += _mat(np.concatenate(xs)) + _mat(np.concatenate(cs)) + _mat(np.concatenate(ups))
    r = _run(tmp_path, blob)
    assert_close(r.matrix(), np.concatenate(outs), 1e-5, "fortran graph_nop forward")
    dxs, dcs, grads = ol.gno_backward(gs, xs, cs, tapes, plist, Fi, Fo, d, H, bool(bias), oa, ups)

    @functools.lru_cache(None)
    def hi():
        with ol.double_precision():
            _, t64 = ol.gno_forward(gs, xs, cs, plist, Fi, Fo, d, H, bool(bias), oa)
            a, b, c = ol.gno_backward(gs, xs, cs, t64, plist, Fi, Fo, d, H, bool(bias), oa, ups)
        return np.concatenate(a), np.concatenate(b), np.concatenate(c)
    assert_close(r.matrix(), np.concatenate(dxs), 1e-5, "fortran graph_nop dx", f64=lambda: hi()[0])
    assert_close(r.matrix(), np.concatenate(dcs), 1e-5, "fortran graph_nop dcoords", f64=lambda: hi()[1])
    gv = r.vector()
    assert_close(gv, np.concatenate(grads), 1e-5, "fortran graph_nop gradients", f64=lambda: hi()[2])
    if H == 64:   # S kept by the forward pass (the default) against S rebuilt by the reverse pass: the same bits
        r2 = _run(tmp_path, blob, env={"ATHENA_MP_GNO_KEEP_S_MAX_GB": "0"})
        r2.matrix(); r2.matrix(); r2.matrix()
        assert np.array_equal(r2.vector(), gv)


def _relabelled(rng, g, self_loops):
    """the same graph with its vertices renamed: equal vertex, edge and entry counts, different adjacency"""
    n = g.num_vertices
    while True:
        perm = rng.permutation(n) + 1
        if n < 2 or not np.array_equal(perm, np.arange(1, n + 1)):
            break
    rows = np.repeat(np.arange(1, n + 1), np.diff(g.adj_ia))
    pairs = {}
    for v, u, e in zip(rows, g.adj_ja[0], g.adj_ja[1]):
        if e > 0:
            pairs.setdefault(int(e), (int(perm[v - 1]), int(perm[u - 1])))
    idx = np.array([pairs[e] for e in sorted(pairs)]).T
    return csr_from_index_list(n, idx, self_loops=self_loops)


def test_fortran_set_graph_before_every_forward_rebuilds_only_on_change(dev, tmp_path):
    """athena_network_sub.f90:2727-2730 calls set_graph before EVERY forward.  Two batches with equal vertex and entry
    counts but different adjacency must give two different (oracle-correct) results -- a cache keyed on the entry count
    alone would serve the first batch's adjacency to the second -- and an unchanged batch must not rebuild the handle.
    The key is the CONTENT of adj_ia / adj_ja (athena_mp_graph_key), so an in-place overwrite is seen too."""
    rng = np.random.default_rng(2024)
    ga = _graphs(rng, [12, 30, 7], self_loops=True)
    gb = [_relabelled(rng, g, True) for g in ga]
    for a, b in zip(ga, gb):
        assert a.num_vertices == b.num_vertices and a.nnz == b.nnz and a.num_edges == b.num_edges
        assert not (np.array_equal(a.adj_ja, b.adj_ja) and np.array_equal(a.adj_ia, b.adj_ia))
    nvf, T, act = [16, 24, 8], 2, _actv("tanh")
    plist = [(rng.standard_normal(nvf[t] * nvf[t - 1]) * 0.4).astype(np.float32) for t in range(1, T + 1)]
    xs = [rng.uniform(-1, 1, (g.num_vertices, nvf[0])).astype(np.float32) for g in ga]
    out_a = np.concatenate(ol.kipf_forward(ga, xs, plist, nvf, "tanh")[0])
    out_b = np.concatenate(ol.kipf_forward(gb, xs, plist, nvf, "tanh")[0])
    assert rel_err_(out_a, out_b) > 1e-2, "the two batches must be distinguishable"
    blob = _case_header(5, ga) + _i(T, len(nvf)) + _i(*nvf) + _act_bytes(act)
    blob += _i(sum(p.size for p in plist)) + np.concatenate(plist).tobytes()
    blob += _case_header(5, gb)[8:]                      # the second batch, without the (kind, batch) words
    blob += _mat(np.concatenate(xs))
    r = _run(tmp_path, blob)
    assert_close(r.matrix(), out_a, 1e-5, "batch A")
    assert_close(r.matrix(), out_b, 1e-5, "batch B (same sizes as A, other adjacency)")
    assert_close(r.matrix(), out_a, 1e-5, "batch A again")
    assert_close(r.matrix(), out_a, 1e-5, "a second layer sharing A's handle")
    assert_close(r.matrix(), out_b, 1e-5, "A's arrays overwritten in place with B's content")
    counts = r.ints(6).tolist()
    # layer-level rebuilds: A, B, A, (unchanged: none), in-place change; device handles built: A and B once each --
    # the returns to A / B are served by the handle cache behind athena_mp_graph_acquire
    assert counts == [1, 2, 3, 3, 4, 2], counts


def test_fortran_set_graph_per_forward_costs_a_key_not_a_build(dev):
    """BASELINE configs[1] size from Fortran (bench_kipf_layer): `steps` x (set_graph + forward + backward) on one
    1 M-vertex / 10 M-entry graph_type builds ONE device handle, and the step with set_graph in front of every forward
    costs at most 1.1 x the resident step (the reference re-copies 84 MB of CSR per set_graph,
    athena_msgpass_layer_sub.f90:144-174; a rebuild here would be ~28 ms against a 2 ms step)."""
    import json
    exe = os.path.join(ROOT, "athena_amd", "fortran", "bench_kipf_layer")
    if not os.path.exists(exe):
        pytest.fail("bench_kipf_layer is not built: __graft_entry__.build() compiles the Fortran host side")
    r = subprocess.run([exe, "1000000", "4500000", "128", "40"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["handles_built_by_the_set_graph_calls"] == 1, line
    assert line["ms_per_step_set_graph_every_forward"] <= 1.1 * line["ms_per_step"], line


@pytest.mark.parametrize("a1,a2,dims", [(_actv("relu"), _actv("none"), (16, 64, 64)), (_actv("swish", p0=1.0), _actv("tanh"), (64, 24, 7))])
def test_fortran_kipf_layers_chained_on_the_device(dev, tmp_path, a1, a2, dims):
    """forward_dev / backward_dev: two Kipf layers stacked from Fortran with the activations between them kept in
    HBM (the first layer's tape feeds the second, the second's input gradient feeds the first)"""
    rng = np.random.default_rng(sum(dims))
    gs = _graphs(rng, [30, 12, 45], self_loops=True)
    f0, f1, f2 = dims
    p1 = (rng.standard_normal(f1 * f0) * np.sqrt(2.0 / f0)).astype(np.float32)
    p2 = (rng.standard_normal(f2 * f1) * np.sqrt(2.0 / f1)).astype(np.float32)
    xs = [rng.uniform(-1, 1, (g.num_vertices, f0)).astype(np.float32) for g in gs]
    o1, t1 = ol.kipf_forward(gs, xs, [p1], [f0, f1], _oracle_act(a1))
    o2, t2 = ol.kipf_forward(gs, o1, [p2], [f1, f2], _oracle_act(a2))
    ups = [rng.uniform(-1, 1, o.shape).astype(np.float32) for o in o2]
    g1, gr2 = ol.kipf_backward(gs, t2, [p2], [f1, f2], _oracle_act(a2), ups)
    g0, gr1 = ol.kipf_backward(gs, t1, [p1], [f0, f1], _oracle_act(a1), g1)
    blob = _case_header(4, gs) + _i(f0, f1, f2) + _act_bytes(a1) + _act_bytes(a2)
    blob += _i(p1.size) + p1.tobytes() + _i(p2.size) + p2.tobytes()
    blob += _mat(np.concatenate(xs)) + _mat(np.concatenate(ups))
    r = _run(tmp_path, blob)
    assert_close(r.matrix(), np.concatenate(o2), 1e-5, "chained forward")

    @functools.lru_cache(None)
    def hi():
        with ol.double_precision():
            q1, u1 = ol.kipf_forward(gs, xs, [p1], [f0, f1], _oracle_act(a1))
            _, u2 = ol.kipf_forward(gs, q1, [p2], [f1, f2], _oracle_act(a2))
            h1, w2 = ol.kipf_backward(gs, u2, [p2], [f1, f2], _oracle_act(a2), ups)
            h0, w1 = ol.kipf_backward(gs, u1, [p1], [f0, f1], _oracle_act(a1), h1)
        return np.concatenate(h0), w1[0], w2[0]
    assert_close(r.matrix(), np.concatenate(g0), 1e-5, "chained dx", f64=lambda: hi()[0])
    assert_close(r.vector(), gr1[0], 1e-5, "dW of the first layer", f64=lambda: hi()[1])
    assert_close(r.vector(), gr2[0], 1e-5, "dW of the second layer", f64=lambda: hi()[2])


def test_fortran_layers_with_an_edgeless_graph_in_the_batch(dev, tmp_path):
    """a batch whose middle graph has vertices but no edges (and one single-vertex graph): empty CSR rows, zero
    edge-feature columns of its own -- Duvenaud through the Fortran layer and through the Python mirror, both against
    the oracle"""
    from athena_amd.layers import duvenaud_msgpass_layer_type

    rng = np.random.default_rng(91)
    gs = _graphs(rng, [7, 12], self_loops=False)
    lonely = csr_from_index_list(3, np.zeros((2, 0), np.int64), self_loops=False)
    single = csr_from_index_list(1, np.zeros((2, 0), np.int64), self_loops=False)
    gs = [gs[0], lonely, gs[1], single]
    Fv, Fe, T, D, nout = 5, 2, 2, 4, 3
    nvf = [Fv] * (T + 1)
    plist = [(rng.standard_normal(nvf[t] * (nvf[t - 1] + Fe) * D) * 0.3).astype(np.float32) for t in range(1, T + 1)]
    plist += [(rng.standard_normal(nout * nvf[t]) * 0.3).astype(np.float32) for t in range(1, T + 1)]
    xs = [rng.uniform(0, 1, (g.num_vertices, Fv)).astype(np.float32) for g in gs]
    es = [rng.uniform(0, 1, (g.num_edges, Fe)).astype(np.float32) for g in gs]
    act, act_r = _actv("sigmoid"), _actv("softmax")
    outs, tapes = ol.duvenaud_forward(gs, xs, es, plist, nvf, Fe, 1, D, nout, "sigmoid")
    up = rng.uniform(-1, 1, outs.shape).astype(np.float32)
    dxs, des, grads = ol.duvenaud_backward(gs, es, tapes, plist, nvf, Fe, 1, D, nout, "sigmoid", up)
    blob = _case_header(2, gs) + _i(T, Fv, Fe, 1, D, nout) + _act_bytes(act) + _act_bytes(act_r)
    blob += _i(sum(p.size for p in plist)) + np.concatenate(plist).tobytes()
    blob += _mat(np.concatenate(xs)) + _mat(np.concatenate(es)) + _mat(up)
    r = _run(tmp_path, blob)

    @functools.lru_cache(None)
    def hi():
        with ol.double_precision():
            _, t64 = ol.duvenaud_forward(gs, xs, es, plist, nvf, Fe, 1, D, nout, "sigmoid")
            a, b, c = ol.duvenaud_backward(gs, es, t64, plist, nvf, Fe, 1, D, nout, "sigmoid", up)
        return np.concatenate(a), np.concatenate(b), np.concatenate(c)
    assert_close(r.matrix(), outs, 1e-5, "fortran forward")
    assert_close(r.matrix(), np.concatenate(dxs), 1e-5, "fortran dx", f64=lambda: hi()[0])
    assert_close(r.matrix(), np.concatenate(des), 1e-5, "fortran de", f64=lambda: hi()[1])
    assert_close(r.vector(), np.concatenate(grads), 1e-5, "fortran gradients", f64=lambda: hi()[2])
    layer = duvenaud_msgpass_layer_type(num_vertex_features=[Fv], num_edge_features=[Fe], num_time_steps=T,
                                        max_vertex_degree=D, num_outputs=nout, min_vertex_degree=1, seed=1)
    layer.set_params(np.concatenate(plist))
    layer.set_graph(gs)
    assert_close(layer.forward(xs, es).cpu().numpy(), outs, 1e-5, "python forward")
    dx, de = layer.backward(up, need_edge_grad=True)
    assert_close(dx.cpu().numpy(), np.concatenate(dxs), 1e-5, "python dx", f64=lambda: hi()[0])
    assert_close(de.cpu().numpy(), np.concatenate(des), 1e-5, "python de", f64=lambda: hi()[1])
    assert_close(layer.get_gradients(), np.concatenate(grads), 1e-5, "python gradients", f64=lambda: hi()[2])


def test_fortran_gno_regression_example(dev):
    """athena_amd/fortran/example_gno_regression.f90: the reference's example/gno_regression driven from Fortran with
    the layers chained on the device, athena_mp_mse_loss and minimise_base -- 200 epochs, the loss falls from O(1) to
    the plateau the tiny model reaches (exit status 0 = below a tenth of the first value)"""
    exe = os.path.join(ROOT, "athena_amd", "fortran", "example_gno_regression")
    if not os.path.exists(exe):
        pytest.fail("example_gno_regression is not built: __graft_entry__.build() compiles the Fortran host side")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "Number of parameters: 282" in " ".join(r.stdout.split())
    final = float(r.stdout.split("Final loss:")[1].split()[0])
    assert 0.0 < final < 0.5
