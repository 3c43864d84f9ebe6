"""GPU: shaped activations (softmax over features, swish), the 'concatenate' merge, and the chained
network of example/msgpass_euler (seven Kipf layers, each fed [input features || previous output])
kept resident in HBM -- forward, reverse pass and Adam updates against the oracle composed on the host."""
import os

import numpy as np
import pytest

import oracle_layers as ol
from helpers import assert_close, csr_from_index_list

pytestmark = pytest.mark.gpu


def T(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def H(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("N,F", [(1000, 3), (777, 7), (513, 14), (300, 32), (129, 64), (65, 200), (1, 5)])
def test_softmax_activation_over_features(dev, oracle, N, F):
    from athena_amd import ops

    rng = np.random.default_rng(N + F)
    z = (rng.standard_normal((N, F)) * 3).astype(np.float32)
    g = rng.standard_normal((N, F)).astype(np.float32)
    y = ops.activation("softmax", T(z, dev))
    yo = oracle.softmax_cols(z)
    assert_close(H(y), yo, 2e-6, "softmax")
    assert np.allclose(H(y).sum(1), 1.0, atol=1e-5)
    assert_close(H(ops.activation_bwd("softmax", y, T(g, dev))), oracle.softmax_cols_bwd(H(y), g), 2e-6, "softmax bwd")


@pytest.mark.parametrize("beta", [1.0, 0.5, 2.0])
def test_swish_activation(dev, oracle, beta):
    from athena_amd import ops

    rng = np.random.default_rng(1)
    x = (rng.standard_normal(100003) * 4).astype(np.float32)
    g = rng.standard_normal(100003).astype(np.float32)
    assert_close(H(ops.activation("swish", T(x, dev), beta=beta)), oracle.swish(x, beta), 1e-6, "swish")
    d = ops.activation_bwd("swish", None if False else T(x, dev), T(g, dev), z=T(x, dev), beta=beta)
    assert_close(H(d), oracle.swish_bwd(x, g, beta), 1e-6, "swish bwd")
    with pytest.raises(ValueError, match="input"):
        ops.activation_bwd("swish", T(x, dev), T(g, dev))


ATTRIBUTED = [("leaky_relu", {}), ("leaky_relu", {"alpha": 0.2, "scale": 1.5}), ("selu", {}),
              ("selu", {"alpha": 1.2, "lambda": 0.9}), ("gaussian", {}), ("gaussian", {"sigma": 0.7, "mu": 0.3, "scale": 2.0}),
              ("piecewise", {}), ("piecewise", {"gradient": 0.25, "limit": 0.5}), ("relu", {"threshold": 0.25}),
              ("relu", {"scale": 3.0}), ("linear", {"scale": 0.5}), ("sigmoid", {"scale": 2.0}), ("tanh", {"scale": 1.0000005})]


@pytest.mark.parametrize("name,attrs", ATTRIBUTED)
def test_activations_with_attributes(dev, oracle, name, attrs):
    """athena_activation_*.f90 with scale / threshold / alpha / lambda / sigma / mu / gradient / limit, given by
    name (reference defaults) and as ops.actv_type; value and reverse factor against the oracle.
    Tolerance 2e-6 relative to the largest value: the device's expf / tanhf differ from libm's by a few ulp."""
    from athena_amd import ops

    rng = np.random.default_rng(7)
    x = (rng.standard_normal(70001) * 2).astype(np.float32)
    g = rng.standard_normal(70001).astype(np.float32)
    a = ops.actv_type(name, **attrs)
    kind = ops.resolve_activation(a)
    scale = a.scale if a.apply_scaling else 1.0
    y = ops.activation(kind, T(x, dev))
    assert_close(H(y), oracle.activation_param(name, x, scale, a.p[0], a.p[1]), 2e-6, f"{name} {attrs}")
    d = ops.activation_bwd(kind, y, T(g, dev), z=T(x, dev))
    assert_close(H(d), oracle.activation_param_bwd(name, x, g, scale, a.p[0], a.p[1]), 2e-6, f"{name} {attrs} bwd")
    if not attrs:      # by name: the reference's reset_* defaults
        assert np.array_equal(H(ops.activation(name, T(x, dev))), H(y))
    if name == "tanh":  # |scale - 1| <= 1e-6 does not scale (apply_scaling, athena_activation_tanh.f90 initialise)
        assert kind == "tanh"
    with pytest.raises(ValueError):
        ops.actv_type(name, no_such_attribute=1.0)


def test_concatenate_merge_and_its_split(dev, oracle):
    from athena_amd import ops

    rng = np.random.default_rng(2)
    a = rng.standard_normal((1001, 3)).astype(np.float32); b = rng.standard_normal((1001, 14)).astype(np.float32)
    out = ops.concat_features(T(a, dev), T(b, dev))
    assert np.array_equal(H(out), oracle.concat(a, b))
    da, db = ops.concat_features_bwd(out, 3)
    assert np.array_equal(H(da), a) and np.array_equal(H(db), b)
    da, db = ops.concat_features_bwd(out, 3, need_a=False)
    assert da is None and np.array_equal(H(db), b)


EULER = [([3, 6], "softmax"), ([9, 14], "softmax"), ([17, 32], "softmax"), ([35, 64], "softmax"),
         ([67, 32], "softmax"), ([35, 14], "softmax"), ([17, 7], "swish")]     # main.f90:183-255


def _oracle_net_forward(g, x, params):
    outs, tapes, cur = [x], [], None
    for k, ((fi, fo), act) in enumerate(EULER):
        inp = x if k == 0 else oracle_concat(x, outs[-1])
        o_, tape = ol.kipf_forward([g], [inp], [params[k]], [fi, fo], act)
        outs.append(o_[0]); tapes.append(tape)
    return outs, tapes


def oracle_concat(a, b):
    return ol.o.concat(a, b)          # ol.o: the fp32 oracle, or its float64 twin under ol.double_precision()


def _trajectory(host_pass):
    """the host-side reference of a short training run, computed twice: with the fp32 oracle (what the device results are
    held against) and -- lazily -- with its float64 twin, the yardstick of helpers.assert_close(..., f64=).
    host_pass() runs ALL passes and returns a list of dicts (one per pass)."""
    import contextlib

    class _T:
        ref = host_pass()
        _hi = None

        @classmethod
        def hi(cls, it, key, k=None):
            def get():
                if cls._hi is None:
                    with ol.double_precision():
                        cls._hi = host_pass()
                v = cls._hi[it][key]
                return v if k is None else v[k]
            return get
    return _T


def _scalar_close(a, b, rtol, f64=None):
    """|a - b| <= rtol |b|, or (anchored) a no further from the float64 value than b is, plus rtol"""
    if abs(a - b) <= rtol * abs(b):
        return True
    if f64 is None:
        return False
    h = float(f64())
    return abs(a - h) <= abs(b - h) + rtol * abs(h)


def _oracle_net_backward(g, params, tapes, up):
    grads = [None] * len(EULER)
    gc = up
    for k in range(len(EULER) - 1, -1, -1):
        (fi, fo), act = EULER[k]
        dxs, gr = ol.kipf_backward([g], tapes[k], [params[k]], [fi, fo], act, [gc])
        grads[k] = gr[0]
        if k > 0:
            gc = np.ascontiguousarray(dxs[0][:, 3:])       # the input's share has requires_grad = .false.
    return grads


def test_msgpass_euler_network_resident_chain(dev, oracle):
    from athena_amd import optim
    from athena_amd.layers import kipf_msgpass_layer_type
    from athena_amd.network import network_type

    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "euler_mesh_edges.npz"))
    n = int(d["num_vertices"])
    g = csr_from_index_list(n, d["index_list"], self_loops=True)
    rng = np.random.default_rng(9)
    x = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    y = rng.uniform(-1, 1, (n, 7)).astype(np.float32)
    net = network_type()
    for k, (nvf, act) in enumerate(EULER):
        layer = kipf_msgpass_layer_type(num_vertex_features=nvf, num_time_steps=1, activation=act, seed=10 + k)
        net.add(layer) if k == 0 else net.add(layer, input_list=[0, -1], operator="concatenate")
    with pytest.raises(ValueError, match="out of range"):
        net.add(kipf_msgpass_layer_type(num_vertex_features=[7, 7], num_time_steps=1), input_list=[-9])
    assert net.parents[1] == [0, 1] and net.parents[6] == [0, 6]
    net.set_graph(g)
    net.compile(optim.adam_optimiser_type(learning_rate=0.005, clip_dict=optim.clip_type(clip_norm=1.0)))
    assert net.get_num_params() == sum(a * b for (a, b), _ in EULER)
    params0 = [l.get_params().copy() for l in net.layers]

    def host_pass():
        params = [p.astype(ol._REAL) for p in params0]
        m = np.zeros(net.get_num_params(), ol._REAL); v = np.zeros_like(m)
        rec = []
        for it in (1, 2):
            outs, tapes = _oracle_net_forward(g, x, params)
            lo, do = ol.o.mse(outs[-1], y)
            grads = _oracle_net_backward(g, params, tapes, do)
            flat = ol.o.clip(np.concatenate(grads), clip_norm=1.0)
            pf, _, m, v = ol.o.adam_step(np.concatenate(params), flat, m, v, 0.005, it)
            o_, new = 0, []
            for k in range(len(params)):
                new.append(pf[o_:o_ + params[k].size]); o_ += params[k].size
            params = new
            rec.append(dict(out=outs[-1], loss=lo, grads=grads, params=params))
        return rec
    tr = _trajectory(host_pass)
    for it in (1, 2):
        ref = tr.ref[it - 1]
        out = net.forward([x])
        assert_close(H(out), ref["out"], 1e-5, f"euler network forward, pass {it}", f64=tr.hi(it - 1, "out"))
        loss, dl = optim.mse_loss_type().compute(out, T(y, dev))
        assert _scalar_close(float(loss.item()), ref["loss"], 1e-5, tr.hi(it - 1, "loss"))
        net.backward(dl)
        for k, l in enumerate(net.layers):
            assert_close(l.get_gradients(), ref["grads"][k], 1e-5, f"dW of layer {k + 1}, pass {it}", f64=tr.hi(it - 1, "grads", k))
        net.update()
        for k in range(len(params0)):
            assert_close(net.layers[k].get_params(), ref["params"][k], 1e-5, f"parameters of layer {k + 1} after update {it}",
                         f64=tr.hi(it - 1, "params", k))


def test_captured_step_replays_the_eager_step(dev, oracle):
    """forward -> mse -> reverse pass recorded once into a HIP graph: replay writes the same loss and the same
    gradients as the eager step, for new inputs copied into the captured buffers"""
    import torch
    from athena_amd import optim
    from athena_amd.layers import kipf_msgpass_layer_type
    from athena_amd.network import network_type

    rng = np.random.default_rng(12)
    n = 500
    pairs = np.array([[i, i + 1] for i in range(1, n)] + [[int(a), int(b)] for a, b in rng.integers(1, n + 1, (700, 2)) if a != b]).T
    g = csr_from_index_list(n, pairs, self_loops=True)
    net = network_type()
    net.add(kipf_msgpass_layer_type(num_vertex_features=[3, 16], num_time_steps=1, activation="softmax", seed=1))
    net.add(kipf_msgpass_layer_type(num_vertex_features=[19, 64], num_time_steps=1, activation="tanh", seed=2), input_list=[0, -1])
    net.add(kipf_msgpass_layer_type(num_vertex_features=[64, 64], num_time_steps=1, activation="relu", seed=3))
    net.add(kipf_msgpass_layer_type(num_vertex_features=[67, 5], num_time_steps=1, activation="swish", seed=4), input_list=[0, -1])
    net.set_graph(g)
    loss = optim.mse_loss_type()
    x0 = rng.uniform(-1, 1, (n, 3)).astype(np.float32); y0 = rng.uniform(-1, 1, (n, 5)).astype(np.float32)
    replay, xb, tb, lb = net.capture_step(x0, y0, loss)
    for trial in range(2):
        x = rng.uniform(-1, 1, (n, 3)).astype(np.float32); y = rng.uniform(-1, 1, (n, 5)).astype(np.float32)
        out = net.forward([x]); l, d = loss.compute(out, T(y, dev)); net.backward(d)
        eager = [l_.get_gradients().copy() for l_ in net.layers]
        le = float(l.item())
        for l_ in net.layers:
            l_.grads = [None] * len(l_.params)
        xb.copy_(T(x, dev)); tb.copy_(T(y, dev))
        replay()
        torch.cuda.synchronize()
        assert float(lb.item()) == le
        for k, l_ in enumerate(net.layers):
            assert np.array_equal(l_.get_gradients(), eager[k]), f"layer {k + 1} gradients differ under replay"


def _molecules(rng, sizes):
    gs = []
    for n in sizes:
        pairs = [[i, i + 1] for i in range(1, n)]
        for _ in range(max(1, n // 4)):
            a, b = rng.integers(1, n + 1, 2)
            if a != b:
                pairs.append([int(a), int(b)])
        gs.append(csr_from_index_list(n, np.array(pairs).T, self_loops=False))
    return gs


def test_msgpass_chemical_network_resident(dev, oracle):
    """example/msgpass_chemical/src/main.f90:129-157: a Duvenaud layer (T = 4, degrees 1..10, softmax readout into
    num_dense_inputs outputs) feeding three full layers with leaky_relu -- forward, mse, every gradient and two
    Adam updates with the whole step resident on the device, against the oracle composed per sample on the host;
    then the step recorded into a HIP graph replays to the same gradients"""
    import torch
    from athena_amd import optim
    from athena_amd.layers import duvenaud_msgpass_layer_type, full_layer_type, read_layer
    from athena_amd.network import network_type

    rng = np.random.default_rng(21)
    gs = _molecules(rng, [9, 14, 21, 6, 17, 11, 25, 8])
    Fv, Fe, T_, D, nd = 6, 1, 4, 10, 10
    xs = [rng.uniform(0, 1, (g.num_vertices, Fv)).astype(np.float32) for g in gs]
    es = [rng.uniform(0, 1, (g.num_edges, Fe)).astype(np.float32) for g in gs]
    y = rng.uniform(-1, 1, (len(gs), 1)).astype(np.float32)
    net = network_type()
    net.add(duvenaud_msgpass_layer_type(num_vertex_features=[Fv], num_edge_features=[Fe], num_time_steps=T_,
                                        max_vertex_degree=D, num_outputs=nd, min_vertex_degree=1,
                                        kernel_initialiser="glorot_normal", readout_activation="softmax", seed=1))
    dense = [(nd, 128), (128, 64), (64, 1)]
    for k, (fi, fo) in enumerate(dense):
        net.add(full_layer_type(num_outputs=fo, num_inputs=fi if k == 0 else None, activation="leaky_relu",
                                kernel_initialiser="he_normal", bias_initialiser="ones", seed=2 + k))
    net.set_graph(gs)
    net.compile(optim.adam_optimiser_type(learning_rate=0.01))
    out = net.forward(xs, es)               # first pass also sizes the full layers that take their width from upstream
    assert tuple(out.shape) == (len(gs), 1)
    assert net.get_num_params() == net.layers[0].get_num_params() + sum((fi + 1) * fo for fi, fo in dense)
    assert np.all(net.layers[1].get_params()[-128:] == 1.0)           # bias_initialiser = 'ones'
    nvf = net.layers[0].num_vertex_features

    def split_duv(flat):
        pl, o_ = [], 0
        for t in range(1, T_ + 1):
            k = nvf[t] * (nvf[t - 1] + Fe) * D
            pl.append(flat[o_:o_ + k]); o_ += k
        for t in range(1, T_ + 1):
            pl.append(flat[o_:o_ + nd * nvf[t]]); o_ += nd * nvf[t]
        return pl

    params0 = [l.get_params().copy() for l in net.layers]

    def host_pass():
        params = [p.astype(ol._REAL) for p in params0]
        m = np.zeros(net.get_num_params(), ol._REAL); v = np.zeros_like(m)
        rec = []
        for it in (1, 2):
            pl = split_duv(params[0])
            h, tapes = ol.duvenaud_forward(gs, xs, es, pl, nvf, Fe, 1, D, nd, "sigmoid")
            acts, keep = [h], []
            for k, (fi, fo) in enumerate(dense):
                W, b = params[1 + k][:fi * fo], params[1 + k][fi * fo:]
                yk, zk = ol.full_forward(acts[-1], W, b, fo, "leaky_relu")
                keep.append((W, b, yk, zk)); acts.append(yk)
            lo, do = ol.o.mse(acts[-1], y)
            gc, grads = do, [None] * 4
            for k in range(2, -1, -1):
                W, b, yk, zk = keep[k]
                gc, gk = ol.full_backward(acts[k], W, b, yk, zk, "leaky_relu", gc)
                grads[1 + k] = np.concatenate(gk)
            _, _, gd = ol.duvenaud_backward(gs, es, tapes, pl, nvf, Fe, 1, D, nd, "sigmoid", gc)
            grads[0] = np.concatenate(gd)
            pf, _, m, v = ol.o.adam_step(np.concatenate(params), np.concatenate(grads), m, v, 0.01, it)
            o_, new = 0, []
            for k in range(len(params)):
                new.append(pf[o_:o_ + params[k].size]); o_ += params[k].size
            params = new
            rec.append(dict(out=acts[-1], loss=lo, grads=grads, params=params))
        return rec
    tr = _trajectory(host_pass)
    for it in (1, 2):
        ref = tr.ref[it - 1]
        out = net.forward(xs, es)
        assert_close(H(out), ref["out"], 1e-5, f"chemical network forward, pass {it}", f64=tr.hi(it - 1, "out"))
        loss, dl = optim.mse_loss_type().compute(out, T(y, dev))
        assert _scalar_close(float(loss.item()), ref["loss"], 1e-5, tr.hi(it - 1, "loss"))
        net.backward(dl)
        for k, l in enumerate(net.layers):
            assert_close(l.get_gradients(), ref["grads"][k], 1e-5, f"gradients of layer {k + 1}, pass {it}", f64=tr.hi(it - 1, "grads", k))
        net.update()
        for k in range(len(params0)):
            assert_close(net.layers[k].get_params(), ref["params"][k], 1e-5, f"parameters of layer {k + 1} after update {it}",
                         f64=tr.hi(it - 1, "params", k))
    # FULL card round trip (print_to_unit_full / read_full)
    card = net.layers[1].print()
    assert card.startswith("FULL\n   NUM_INPUTS = 10\n   NUM_OUTPUTS = 128\n   USE_BIAS = T\n   ACTIVATION\n      name = leaky_relu")
    assert "alpha = 0.010000" in card and read_layer(card).print() == card
    assert np.allclose(read_layer(card).get_params(), net.layers[1].get_params(), rtol=1e-7, atol=0)   # eight digits
    # the same step recorded into a HIP graph: replay on new features reproduces the eager gradients bit for bit
    replay, xb, tb, lb = net.capture_step(xs, y, optim.mse_loss_type(), edge_features=es)
    xs2 = [rng.uniform(0, 1, a.shape).astype(np.float32) for a in xs]
    es2 = [rng.uniform(0, 1, a.shape).astype(np.float32) for a in es]
    out = net.forward(xs2, es2); l, d = optim.mse_loss_type().compute(out, T(y, dev)); net.backward(d)
    eager = [l_.get_gradients().copy() for l_ in net.layers]
    for l_ in net.layers:
        l_.grads = [None] * len(l_.params)
    xb.copy_(T(np.concatenate(xs2), dev)); net.captured_edge_buf.copy_(T(np.concatenate(es2), dev))
    replay()
    torch.cuda.synchronize()
    assert float(lb.item()) == float(l.item())
    for l_, ge in zip(net.layers, eager):
        assert np.array_equal(l_.get_gradients(), ge)


def test_gno_regression_network_resident(dev, oracle):
    """example/gno_regression/src/main.f90: a 20-vertex chain, edge feature = coordinate difference of the pair,
    two stacked graph_nop layers (1 -> 8 relu -> 2, kernel_hidden 8; the second takes its input width from the
    first, the edge geometry is forwarded unchanged), mse, plain gradient descent (base_optimiser_type) --
    three training steps against the oracle composed on the host"""
    from athena_amd import optim
    from athena_amd.layers import graph_nop_layer_type
    from athena_amd.network import network_type

    nv, F_in, F_h, F_out, Hk, d, lr = 20, 1, 8, 2, 8, 1, 0.01
    pairs = np.array([[i, i + 1] for i in range(1, nv)]).T
    g = csr_from_index_list(nv, pairs, self_loops=False)
    coords = (np.arange(nv, dtype=np.float32) / np.float32(nv - 1) * np.float32(2 * np.pi)).astype(np.float32)
    x = np.ones((nv, F_in), np.float32)
    tgt = np.stack([np.sin(coords), np.cos(coords)], axis=1).astype(np.float32)
    ec = (coords[:-1] - coords[1:]).reshape(nv - 1, d).astype(np.float32)
    net = network_type()
    net.add(graph_nop_layer_type(num_inputs=F_in, num_outputs=F_h, coord_dim=d, kernel_hidden=Hk, activation="relu", seed=1))
    net.add(graph_nop_layer_type(num_outputs=F_out, coord_dim=d, kernel_hidden=Hk, seed=2))
    net.set_graph(g)
    net.compile(optim.base_optimiser_type(learning_rate=lr))
    out = net.forward([x], [ec])
    assert tuple(out.shape) == (nv, F_out)
    dims = [(F_in, F_h, "relu"), (F_h, F_out, "none")]
    assert net.get_num_params() == sum(Hk * d + Hk + fo * fi * Hk + fo * fi + fo * fi + fo for fi, fo, _ in dims)

    def split(flat, fi, fo):
        F = fo * fi
        sizes = [Hk * d + Hk + F * Hk + F, F, fo]
        pl, o_ = [], 0
        for n in sizes:
            pl.append(flat[o_:o_ + n]); o_ += n
        return pl

    # non-zero biases so that every gradient path carries signal
    rng = np.random.default_rng(3)
    for l in net.layers:
        l.set_params(l.get_params() + rng.standard_normal(l.get_num_params()).astype(np.float32) * 0.05)
    params0 = [l.get_params().copy() for l in net.layers]

    def host_pass():
        params = [p.astype(ol._REAL) for p in params0]
        rec = []
        for it in (1, 2, 3):
            cur, keep = x, []
            for (fi, fo, act), pf in zip(dims, params):
                pl = split(pf, fi, fo)
                o_, tp = ol.gno_forward([g], [cur], [ec], pl, fi, fo, d, Hk, True, act)
                keep.append((pl, cur, tp)); cur = o_[0]
            lo, do = ol.o.mse(cur, tgt)
            gc, grads = do, [None, None]
            for k in (1, 0):
                fi, fo, act = dims[k]
                pl, xin, tp = keep[k]
                dxs, _, gk = ol.gno_backward([g], [xin], [ec], tp, pl, fi, fo, d, Hk, True, act, [gc])
                grads[k] = np.concatenate(gk); gc = dxs[0]
            params = [(params[k] - ol._REAL(np.float32(lr)) * grads[k]).astype(ol._REAL) for k in range(2)]   # minimise_base :396-418
            rec.append(dict(out=cur, loss=lo, grads=grads, params=params))
        return rec
    tr = _trajectory(host_pass)
    for it in (1, 2, 3):
        ref = tr.ref[it - 1]
        out = net.forward([x], [ec])
        assert_close(H(out), ref["out"], 1e-5, f"gno network forward, step {it}", f64=tr.hi(it - 1, "out"))
        loss, dl = optim.mse_loss_type().compute(out, T(tgt, dev))
        assert _scalar_close(float(loss.item()), ref["loss"], 1e-5, tr.hi(it - 1, "loss"))
        net.backward(dl)
        for k, l in enumerate(net.layers):
            assert_close(l.get_gradients(), ref["grads"][k], 1e-5, f"gradients of layer {k + 1}, step {it}", f64=tr.hi(it - 1, "grads", k))
        net.update()
        for k in range(2):
            assert_close(net.layers[k].get_params(), ref["params"][k], 1e-5, f"parameters of layer {k + 1} after step {it}",
                         f64=tr.hi(it - 1, "params", k))


def test_network_accessors_of_the_reference_msgpass_network_test(dev):
    """test_msgpass_network.f90: num_layers counts the inserted input layer (:55, :117), get_num_params / get_params
    over the whole network (:162-188), reset empties it (:207-211)"""
    from athena_amd import optim
    from athena_amd.layers import duvenaud_msgpass_layer_type, full_layer_type, kipf_msgpass_layer_type
    from athena_amd.network import network_type

    net = network_type()
    assert net.num_layers == 0 and net.get_num_params() == 0 and net.get_params().size == 0
    net.add(kipf_msgpass_layer_type(num_vertex_features=[3, 4], num_time_steps=1, activation="relu", seed=1))
    net.compile(optim.sgd_optimiser_type(learning_rate=0.01))
    assert net.num_layers == 2 and net.get_num_params() == 12
    p = net.get_params()
    assert p.size == 12 and np.array_equal(p, net.layers[0].get_params())
    net.set_params(p * 2)
    assert np.array_equal(net.get_params(), p * 2)
    with pytest.raises(ValueError, match="wrong number of parameters"):
        net.set_params(p[:5])
    net.reset()
    assert net.num_layers == 0 and net.optimiser is None
    d = network_type()
    d.add(duvenaud_msgpass_layer_type(num_vertex_features=[4], num_edge_features=[2], num_time_steps=2,
                                      max_vertex_degree=3, num_outputs=5, seed=2))
    d.add(full_layer_type(num_inputs=5, num_outputs=1, seed=3))
    assert d.num_layers == 3
    assert d.get_num_params() == 2 * 4 * 6 * 3 + 2 * 5 * 4 + 6 and d.get_params().size == d.get_num_params()
