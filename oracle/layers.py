"""Oracle-side restatement of the three layers' update_message / update_readout and of the tape walk
diffstruc's grad_reverse performs over them -- per sample, looping on the host exactly like the
reference (athena_kipf_msgpass_layer.f90:940-957, athena_duvenaud_msgpass_layer.f90:792-855,
athena_graph_nop_layer.f90:740-786).  Test infrastructure only."""
import numpy as np

import contextlib

from oracle import oracle as o


@contextlib.contextmanager
def double_precision():
    """run the same per-sample compositions on the float64 twin of the oracle (oracle/oracle64.py): the yardstick of
    helpers.assert_close(..., f64=...)"""
    global o, _REAL
    from oracle import oracle64
    keep, o, _REAL = o, oracle64, np.float64
    try:
        yield
    finally:
        o, _REAL = keep, np.float32


class f64_lazy:
    """yardstick for helpers.assert_close(..., f64=...): `fn` (a closure over the test's inputs that re-runs the oracle
    composition and returns a tuple of lists / arrays) is evaluated once, under double_precision(), only if some
    comparison needs it; hi(i) is the callable for element i (lists are concatenated)"""

    def __init__(self, fn):
        self.fn, self.val = fn, None

    def _get(self, i):
        if self.val is None:
            with double_precision():
                self.val = self.fn()
        v = self.val[i]
        return np.concatenate(v) if isinstance(v, (list, tuple)) else v

    def __call__(self, i):
        return lambda: self._get(i)


_REAL = np.float32      # accumulator type of the host-side sums below (float64 under double_precision)


_ATTRIBUTED = ("leaky_relu", "selu", "gaussian", "piecewise")


def _spec(act):
    """(name, scale, p0, p1) of an activation given by name (reference defaults) or as an object with
    .name / .scale / .apply_scaling / .p (athena_amd.ops.actv_type)"""
    if hasattr(act, "name"):
        return act.name, (act.scale if act.apply_scaling else 1.0), act.p[0], act.p[1]
    d = {"leaky_relu": (0.01, 0.0), "selu": (1.67326, 1.0507), "gaussian": (1.5, 0.0), "piecewise": (0.1, 1.0)}
    return (act, 1.0) + d[act]


def act_fwd(act, z):
    """athena_activation_*.f90 apply, incl. the shaped ones the euler example uses and the attributed ones"""
    if hasattr(act, "name") or act in _ATTRIBUTED:
        if getattr(act, "name", act) == "swish":
            return o.swish(z, act.beta)
        n, sc, p0, p1 = _spec(act)
        return o.activation_param(n, z, sc, p0, p1)
    if act == "softmax":
        return o.softmax_cols(z)
    if act == "swish":
        return o.swish(z)
    return o.activation(act, z)


def act_bwd(act, y, g, z):
    if hasattr(act, "name") or act in _ATTRIBUTED:
        if getattr(act, "name", act) == "swish":
            return o.swish_bwd(z, g, act.beta)
        n, sc, p0, p1 = _spec(act)
        return o.activation_param_bwd(n, z, g, sc, p0, p1)
    if act == "softmax":
        return o.softmax_cols_bwd(y, g)
    if act == "swish":
        return o.swish_bwd(z, g)       # differentiates at the input
    return o.activation_bwd(act, y, g)


def kipf_forward(graphs, xs, params, nvf, act):
    outs, tapes = [], []
    for g, x in zip(graphs, xs):
        cur, tape = x, []
        for t in range(1, len(nvf)):
            p = o.kipf_propagate(cur, g.adj_ia, g.adj_ja)
            z = o.matmul(params[t - 1], p, nvf[t])
            y = act_fwd(act, z)
            tape.append((p, y, z))
            cur = y
        outs.append(cur); tapes.append(tape)
    return outs, tapes


def kipf_backward(graphs, tapes, params, nvf, act, ups, exact=False):
    grads = [np.zeros(p.shape, _REAL) for p in params]
    dxs = []
    for g, tape, up in zip(graphs, tapes, ups):
        gc = up
        for t in range(len(nvf) - 1, 0, -1):
            p, y, z = tape[t - 1]
            dz = act_bwd(act, y, gc, z)
            grads[t - 1] += o.matmul_dw(dz, p)
            dp = o.matmul_dx(params[t - 1], dz, nvf[t - 1])
            gc = o.kipf_propagate_bwd(dp, g.adj_ia, g.adj_ja, exact=exact)
        dxs.append(gc)
    return dxs, grads


def duvenaud_forward(graphs, xs, es, params, nvf, Fe, mn, mx, nout, act, act_readout="softmax"):
    T = len(nvf) - 1
    out = np.zeros((len(graphs), nout), _REAL)
    tapes = []
    for s, (g, x, e) in enumerate(zip(graphs, xs, es)):
        cur, A, Z = x, [], []
        for t in range(1, T + 1):
            a = o.duvenaud_propagate(cur, e, g.adj_ia, g.adj_ja)
            c = o.duvenaud_update(a, params[t - 1], g.adj_ia, mn, mx, nvf[t])
            z = act_fwd(act, c)
            A.append(a); Z.append((z, c)); cur = z
        P = []
        for t in range(1, T + 1):
            lg = o.matmul(params[T + t - 1], Z[t - 1][0], nout)
            p = act_fwd(act_readout, lg)
            out[s] += o.segment_sum(p, np.array([0, p.shape[0]], np.int32))[0]
            P.append((p, lg))
        tapes.append((A, Z, P))
    return out, tapes


def duvenaud_backward(graphs, es, tapes, params, nvf, Fe, mn, mx, nout, act, gout, act_readout="softmax"):
    T = len(nvf) - 1
    grads = [np.zeros(p.shape, _REAL) for p in params]
    dxs, des = [], []
    for s, (g, e, (A, Z, P)) in enumerate(zip(graphs, es, tapes)):
        n = Z[0][0].shape[0]
        gv = np.repeat(gout[s:s + 1], n, axis=0)
        dz_next = None
        de = np.zeros(e.shape, _REAL)
        for t in range(T, 0, -1):
            dl = act_bwd(act_readout, P[t - 1][0], gv, P[t - 1][1])
            grads[T + t - 1] += o.matmul_dw(dl, Z[t - 1][0])
            dz = o.matmul_dx(params[T + t - 1], dl, nvf[t])
            if dz_next is not None:
                dz = dz + dz_next
            dc = act_bwd(act, Z[t - 1][0], dz, Z[t - 1][1])
            grads[t - 1] += o.duvenaud_update_bwd_w(dc, A[t - 1], g.adj_ia, mn, mx)
            da = o.duvenaud_update_bwd_a(dc, params[t - 1], g.adj_ia, mn, mx, A[t - 1].shape[1])
            de += o.duvenaud_propagate_bwd_e(da, nvf[t - 1], e.shape[0], g.adj_ia, g.adj_ja)
            dz_next = o.duvenaud_propagate_bwd_x(da, nvf[t - 1], g.adj_ia, g.adj_ja)
        dxs.append(dz_next); des.append(de)
    return dxs, des, grads


def gno_forward(graphs, xs, cs, params, Fi, Fo, d, H, use_bias, act):
    outs, tapes = [], []
    for g, x, c in zip(graphs, xs, cs):
        kap = o.gno_kernel_eval(c, params[0], H, Fo * Fi)
        m = o.gno_aggregate(x, kap, g.adj_ia, g.adj_ja, Fo)
        z = m + o.matmul(params[1], x, Fo)
        if use_bias:
            z = o.add_bias_rows(z, params[2])
        y = act_fwd(act, z)
        outs.append(y); tapes.append((kap, y, z))
    return outs, tapes


def gno_backward(graphs, xs, cs, tapes, params, Fi, Fo, d, H, use_bias, act, ups):
    grads = [np.zeros(p.shape, _REAL) for p in params]
    dxs, dcs = [], []
    for g, x, c, (kap, y, z), up in zip(graphs, xs, cs, tapes, ups):
        dz = act_bwd(act, y, up, z)
        if use_bias:
            grads[2] += dz.sum(axis=0)
        grads[1] += o.matmul_dw(dz, x)
        dk = o.gno_aggregate_bwd_k(dz, x, c.shape[0], g.adj_ia, g.adj_ja)
        grads[0] += o.gno_kernel_bwd_theta(c, params[0], dk, H)
        dxs.append(o.matmul_dx(params[1], dz, Fi) + o.gno_aggregate_bwd_x(dz, kap, g.adj_ia, g.adj_ja, Fi))
        dcs.append(o.gno_kernel_bwd_coords(c, params[0], dk, H))
    return dxs, dcs, grads


def full_forward(x, W, b, Fo, act):
    """forward_full, athena_full_layer.f90:835-875: act(matmul(params(1), input) + params(2)); returns (y, z)"""
    z = o.matmul(W, x, Fo)
    if b is not None:
        z = o.add_bias_rows(z, b)
    return act_fwd(act, z), z


def full_backward(x, W, b, y, z, act, up):
    """the tape walk through activation -> + bias -> matmul; returns (dx, [dW, db])"""
    dz = act_bwd(act, y, up, z)
    grads = [o.matmul_dw(dz, x)]
    if b is not None:
        db = np.zeros(dz.shape[1], _REAL)
        for r in range(dz.shape[0]):
            db = db + dz[r]                       # batch rows in order, fp32
        grads.append(db)
    return o.matmul_dx(W, dz, x.shape[1]), grads
