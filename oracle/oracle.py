"""ctypes front-end of oracle/liboracle.so (athena_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of athena_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.

Array conventions are the reference's (Fortran), expressed in numpy:
  features  x : float32 [N, F] C-contiguous  == Fortran val(F, N)
  adj_ia      : int32 [N+1], 1-based row pointers
  adj_ja      : int32 [2, nnz] (Fortran order), 1-based neighbour / edge column
  weights     : flat float32, column-major W(Fo, Fi) == params(t)%val(:,1)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ACT = {"none": 0, "linear": 0, "relu": 1, "sigmoid": 2, "tanh": 3}


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "athena_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.c_void_p)


def _ia(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.c_void_p)


def _ja(a):
    a = np.asfortranarray(a, dtype=np.int32)
    assert a.ndim == 2 and a.shape[0] == 2, "adj_ja must be [2, nnz]"
    return a, a.ctypes.data_as(C.c_void_p)


def kipf_propagate(x, adj_ia, adj_ja):
    """athena_diffstruc_extd_sub_kipf.f90:7-59"""
    x, px = _f(x); ia, pia = _ia(adj_ia); ja, pja = _ja(adj_ja)
    N, F = x.shape
    y = np.empty_like(x)
    lib().oracle_kipf_propagate(C.c_int(F), C.c_int(N), px, pia, pja, y.ctypes.data_as(C.c_void_p))
    return y


def kipf_propagate_rect(x, adj_ia, adj_ja, row_deg, col_deg):
    x, px = _f(x); ia, pia = _ia(adj_ia); ja, pja = _ja(adj_ja)
    rd, prd = _ia(row_deg); cd, pcd = _ia(col_deg)
    n_rows = ia.shape[0] - 1
    F = x.shape[1]
    y = np.empty((n_rows, F), np.float32)
    lib().oracle_kipf_propagate_rect(C.c_int(F), C.c_int(n_rows), px, pia, pja, prd, pcd,
                                     y.ctypes.data_as(C.c_void_p))
    return y


def kipf_propagate_bwd(g, adj_ia, adj_ja, exact=False, n_out=None):
    """athena_diffstruc_extd_sub_kipf.f90:85-111 (exact=False is the reference)"""
    g, pg = _f(g); ia, pia = _ia(adj_ia); ja, pja = _ja(adj_ja)
    N, F = g.shape
    n_out = N if n_out is None else n_out
    out = np.empty((n_out, F), np.float32)
    lib().oracle_kipf_propagate_bwd(C.c_int(F), C.c_int(N), C.c_int(n_out), pg, pia, pja,
                                    C.c_int(int(exact)), out.ctypes.data_as(C.c_void_p))
    return out


def matmul(W, P, Fo):
    """Z = W(Fo,Fi) P ; W flat column-major (athena_kipf_msgpass_layer.f90:951)"""
    P, pP = _f(P); W, pW = _f(W)
    N, Fi = P.shape
    assert W.size == Fo * Fi
    Z = np.empty((N, Fo), np.float32)
    lib().oracle_matmul(C.c_int(Fo), C.c_int(Fi), C.c_int(N), pW, pP, Z.ctypes.data_as(C.c_void_p))
    return Z


def matmul_dw(dZ, P):
    dZ, pz = _f(dZ); P, pP = _f(P)
    N, Fo = dZ.shape; Fi = P.shape[1]
    dW = np.empty(Fo * Fi, np.float32)
    lib().oracle_matmul_dw(C.c_int(Fo), C.c_int(Fi), C.c_int(N), pz, pP, dW.ctypes.data_as(C.c_void_p))
    return dW


def matmul_dx(W, dZ, Fi):
    dZ, pz = _f(dZ); W, pW = _f(W)
    N, Fo = dZ.shape
    dP = np.empty((N, Fi), np.float32)
    lib().oracle_matmul_dx(C.c_int(Fo), C.c_int(Fi), C.c_int(N), pW, pz, dP.ctypes.data_as(C.c_void_p))
    return dP


def duvenaud_propagate(x, e, adj_ia, adj_ja):
    """athena_diffstruc_extd_sub_duvenaud.f90:7-59"""
    x, px = _f(x); e, pe = _f(e); ia, pia = _ia(adj_ia); ja, pja = _ja(adj_ja)
    N, Fv = x.shape; Fe = e.shape[1]
    c = np.empty((N, Fv + Fe), np.float32)
    lib().oracle_duvenaud_propagate(C.c_int(Fv), C.c_int(Fe), C.c_int(N), px, pe, pia, pja,
                                    c.ctypes.data_as(C.c_void_p))
    return c


def duvenaud_propagate_bwd_x(g, Fv, adj_ia, adj_ja):
    g, pg = _f(g); ia, pia = _ia(adj_ia); ja, pja = _ja(adj_ja)
    N, Fc = g.shape
    dx = np.empty((N, Fv), np.float32)
    lib().oracle_duvenaud_propagate_bwd_x(C.c_int(Fv), C.c_int(Fc - Fv), C.c_int(N), pg, pia, pja,
                                          dx.ctypes.data_as(C.c_void_p))
    return dx


def duvenaud_propagate_bwd_e(g, Fv, E, adj_ia, adj_ja):
    g, pg = _f(g); ia, pia = _ia(adj_ia); ja, pja = _ja(adj_ja)
    N, Fc = g.shape
    de = np.empty((E, Fc - Fv), np.float32)
    lib().oracle_duvenaud_propagate_bwd_e(C.c_int(Fv), C.c_int(Fc - Fv), C.c_int(N), C.c_int(E), pg,
                                          pia, pja, de.ctypes.data_as(C.c_void_p))
    return de


def duvenaud_update(a, weight, adj_ia, min_deg, max_deg, Fo):
    """athena_diffstruc_extd_sub_duvenaud.f90:176-228"""
    a, pa = _f(a); w, pw = _f(weight); ia, pia = _ia(adj_ia)
    N, Fi = a.shape
    assert w.size == Fo * Fi * (max_deg - min_deg + 1)
    c = np.empty((N, Fo), np.float32)
    lib().oracle_duvenaud_update(C.c_int(Fo), C.c_int(Fi), C.c_int(N), pa, pw, pia, C.c_int(min_deg),
                                 C.c_int(max_deg), c.ctypes.data_as(C.c_void_p))
    return c


def duvenaud_update_bwd_a(g, weight, adj_ia, min_deg, max_deg, Fi):
    g, pg = _f(g); w, pw = _f(weight); ia, pia = _ia(adj_ia)
    N, Fo = g.shape
    da = np.empty((N, Fi), np.float32)
    lib().oracle_duvenaud_update_bwd_a(C.c_int(Fo), C.c_int(Fi), C.c_int(N), pg, pw, pia,
                                       C.c_int(min_deg), C.c_int(max_deg), da.ctypes.data_as(C.c_void_p))
    return da


def duvenaud_update_bwd_w(g, a, adj_ia, min_deg, max_deg):
    g, pg = _f(g); a, pa = _f(a); ia, pia = _ia(adj_ia)
    N, Fo = g.shape; Fi = a.shape[1]
    dW = np.empty(Fo * Fi * (max_deg - min_deg + 1), np.float32)
    lib().oracle_duvenaud_update_bwd_w(C.c_int(Fo), C.c_int(Fi), C.c_int(N), pg, pa, pia,
                                       C.c_int(min_deg), C.c_int(max_deg), dW.ctypes.data_as(C.c_void_p))
    return dW


def activation(kind, z):
    z, pz = _f(z)
    y = np.empty_like(z)
    lib().oracle_activation(C.c_int(ACT[kind]), C.c_size_t(z.size), pz, y.ctypes.data_as(C.c_void_p))
    return y


def activation_bwd(kind, y, g):
    y, py = _f(y); g, pg = _f(g)
    dz = np.empty_like(y)
    lib().oracle_activation_bwd(C.c_int(ACT[kind]), C.c_size_t(y.size), py, pg, dz.ctypes.data_as(C.c_void_p))
    return dz


def softmax_cols(z):
    """athena_diffstruc_extd_sub.f90:309-313 (dim=2: per vertex, over features)"""
    z, pz = _f(z)
    N, F = z.shape
    y = np.empty_like(z)
    lib().oracle_softmax_cols(C.c_int(F), C.c_int(N), pz, y.ctypes.data_as(C.c_void_p))
    return y


def softmax_cols_bwd(y, g):
    y, py = _f(y); g, pg = _f(g)
    N, F = y.shape
    dz = np.empty_like(y)
    lib().oracle_softmax_cols_bwd(C.c_int(F), C.c_int(N), py, pg, dz.ctypes.data_as(C.c_void_p))
    return dz


def segment_sum(p, seg, out=None):
    """athena_duvenaud_msgpass_layer.f90:846-852 (sum over the vertices of each graph)"""
    p, pp = _f(p); seg, ps = _ia(seg)
    S = seg.shape[0] - 1
    O = p.shape[1]
    acc = out is not None
    if out is None:
        out = np.empty((S, O), np.float32)
    assert out.dtype == np.float32 and out.flags.c_contiguous
    lib().oracle_segment_sum(C.c_int(O), C.c_int(S), ps, pp, C.c_int(int(acc)), out.ctypes.data_as(C.c_void_p))
    return out


def gno_kernel_eval(coords, theta, H, F):
    """athena_diffstruc_extd_sub_nop.f90:26-115"""
    coords, pc = _f(coords); theta, pt = _f(theta)
    E, d = coords.shape
    assert theta.size == H * d + H + F * H + F
    kappa = np.empty((E, F), np.float32)
    lib().oracle_gno_kernel_eval(C.c_int(d), C.c_int(H), C.c_int(F), C.c_int(E), pc, pt,
                                 kappa.ctypes.data_as(C.c_void_p))
    return kappa


def gno_aggregate(x, kappa, adj_ia, adj_ja, Fo):
    """athena_diffstruc_extd_sub_nop.f90:330-397"""
    x, px = _f(x); kappa, pk = _f(kappa); ia, pia = _ia(adj_ia); ja, pja = _ja(adj_ja)
    N, Fi = x.shape
    c = np.empty((N, Fo), np.float32)
    lib().oracle_gno_aggregate(C.c_int(Fi), C.c_int(Fo), C.c_int(N), px, pk, pia, pja,
                               c.ctypes.data_as(C.c_void_p))
    return c


def gno_aggregate_bwd_x(g, kappa, adj_ia, adj_ja, Fi):
    g, pg = _f(g); kappa, pk = _f(kappa); ia, pia = _ia(adj_ia); ja, pja = _ja(adj_ja)
    N, Fo = g.shape
    dx = np.empty((N, Fi), np.float32)
    lib().oracle_gno_aggregate_bwd_x(C.c_int(Fi), C.c_int(Fo), C.c_int(N), pg, pk, pia, pja,
                                     dx.ctypes.data_as(C.c_void_p))
    return dx


def gno_aggregate_bwd_k(g, x, E, adj_ia, adj_ja):
    g, pg = _f(g); x, px = _f(x); ia, pia = _ia(adj_ia); ja, pja = _ja(adj_ja)
    N, Fo = g.shape; Fi = x.shape[1]
    dk = np.empty((E, Fo * Fi), np.float32)
    lib().oracle_gno_aggregate_bwd_k(C.c_int(Fi), C.c_int(Fo), C.c_int(N), C.c_int(E), pg, px, pia,
                                     pja, dk.ctypes.data_as(C.c_void_p))
    return dk


def gno_kernel_bwd_theta(coords, theta, dkappa, H):
    coords, pc = _f(coords); theta, pt = _f(theta); dkappa, pd = _f(dkappa)
    E, d = coords.shape; F = dkappa.shape[1]
    dth = np.empty(theta.size, np.float32)
    lib().oracle_gno_kernel_bwd_theta(C.c_int(d), C.c_int(H), C.c_int(F), C.c_int(E), pc, pt, pd,
                                      dth.ctypes.data_as(C.c_void_p))
    return dth


def gno_kernel_bwd_coords(coords, theta, dkappa, H):
    coords, pc = _f(coords); theta, pt = _f(theta); dkappa, pd = _f(dkappa)
    E, d = coords.shape; F = dkappa.shape[1]
    dc = np.empty((E, d), np.float32)
    lib().oracle_gno_kernel_bwd_coords(C.c_int(d), C.c_int(H), C.c_int(F), C.c_int(E), pc, pt, pd,
                                       dc.ctypes.data_as(C.c_void_p))
    return dc


def add_bias_rows(y, b):
    y = np.array(y, dtype=np.float32, order="C", copy=True)
    b, pb = _f(b)
    N, F = y.shape
    lib().oracle_add_bias_rows(C.c_int(F), C.c_int(N), pb, y.ctypes.data_as(C.c_void_p))
    return y


# ---- parameter update of a train step -------------------------------------------
REG = {None: 0, "none": 0, "l1": 1, "l2": 2, "l1l2": 3}


def clip(g, clip_min=None, clip_max=None, clip_norm=None):
    """athena_clipper.f90:165-210 (in place on a copy)"""
    g = np.array(g, dtype=np.float32, copy=True)
    mm = clip_min is not None or clip_max is not None
    lib().oracle_clip(C.c_size_t(g.size), g.ctypes.data_as(C.c_void_p), C.c_int(int(mm)),
                      C.c_float(clip_min if clip_min is not None else -np.finfo(np.float32).max),
                      C.c_float(clip_max if clip_max is not None else np.finfo(np.float32).max),
                      C.c_int(int(clip_norm is not None)), C.c_float(clip_norm or 0.0))
    return g


def sgd_step(param, grad, velocity, lr, momentum=0.0, nesterov=False, reg=None, l1=0.0, l2=0.0):
    """athena_optimiser.f90:634-673; returns (param, grad, velocity) copies after the step"""
    p, g, v = (np.array(a, dtype=np.float32, copy=True) for a in (param, grad, velocity))
    lib().oracle_sgd_step(C.c_size_t(p.size), C.c_float(lr), C.c_float(momentum), C.c_int(int(nesterov)),
                          C.c_int(REG[reg]), C.c_float(l1), C.c_float(l2), p.ctypes.data_as(C.c_void_p),
                          g.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p))
    return p, g, v


def adam_step(param, grad, m, v, lr, it, beta1=0.9, beta2=0.999, epsilon=1e-8, reg=None, l1=0.0, l2=0.0,
              decoupled=False):
    """athena_optimiser.f90:1027-1091; returns (param, grad, m, v) copies after the step"""
    p, g, m_, v_ = (np.array(a, dtype=np.float32, copy=True) for a in (param, grad, m, v))
    lib().oracle_adam_step(C.c_size_t(p.size), C.c_float(lr), C.c_float(beta1), C.c_float(beta2), C.c_float(epsilon),
                           C.c_int(it), C.c_int(REG[reg]), C.c_float(l1), C.c_float(l2), C.c_int(int(decoupled)),
                           p.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.c_void_p),
                           m_.ctypes.data_as(C.c_void_p), v_.ctypes.data_as(C.c_void_p))
    return p, g, m_, v_


def mse(pred, expected):
    """athena_loss.f90:393-430; returns (loss, dloss/dpred)"""
    p, pp = _f(pred); e, pe = _f(expected)
    d = np.empty_like(p)
    f = lib().oracle_mse
    f.restype = C.c_float
    return float(f(C.c_size_t(p.size), pp, pe, d.ctypes.data_as(C.c_void_p))), d


def swish(x, beta=1.0):
    """athena_diffstruc_extd_sub.f90:424-455"""
    x, px = _f(x)
    y = np.empty_like(x)
    lib().oracle_swish(C.c_size_t(x.size), C.c_float(beta), px, y.ctypes.data_as(C.c_void_p))
    return y


def swish_bwd(x, g, beta=1.0):
    """athena_diffstruc_extd_sub.f90:477-492 (differentiates at the input x)"""
    x, px = _f(x); g, pg = _f(g)
    d = np.empty_like(x)
    lib().oracle_swish_bwd(C.c_size_t(x.size), C.c_float(beta), px, pg, d.ctypes.data_as(C.c_void_p))
    return d


ACTP = {"linear": 0, "relu": 1, "sigmoid": 2, "tanh": 3, "leaky_relu": 4, "selu": 5, "gaussian": 6, "piecewise": 7}


def activation_param(kind, x, scale=1.0, p0=0.0, p1=0.0):
    """athena_activation_*.f90 apply with attributes (see athena_oracle.c)"""
    x, px = _f(x)
    y = np.empty_like(x)
    lib().oracle_activation_param(ACTP[kind], C.c_size_t(x.size), C.c_float(scale), C.c_float(p0), C.c_float(p1), px,
                                  y.ctypes.data_as(C.c_void_p))
    return y


def activation_param_bwd(kind, x, g, scale=1.0, p0=0.0, p1=0.0):
    x, px = _f(x); g, pg = _f(g)
    d = np.empty_like(x)
    lib().oracle_activation_param_bwd(ACTP[kind], C.c_size_t(x.size), C.c_float(scale), C.c_float(p0), C.c_float(p1),
                                      px, pg, d.ctypes.data_as(C.c_void_p))
    return d


def concat(a, b):
    a, pa = _f(a); b, pb = _f(b)
    out = np.empty((a.shape[0], a.shape[1] + b.shape[1]), np.float32)
    lib().oracle_concat(C.c_int(a.shape[0]), C.c_int(a.shape[1]), C.c_int(b.shape[1]), pa, pb, out.ctypes.data_as(C.c_void_p))
    return out


# ---- all-cores context number (athena_oracle_omp.c; measurement aid of bench.py's cpu_baseline leg) ---------
_LIB_OMP = None


def lib_omp():
    global _LIB_OMP
    if _LIB_OMP is None:
        so = os.path.join(_HERE, "liboracle_omp.so")
        src = os.path.join(_HERE, "athena_oracle_omp.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_omp.so"], stdout=subprocess.DEVNULL)
        _LIB_OMP = C.CDLL(so)
    return _LIB_OMP


def omp_kipf_step(x, W, dz, adj_ia, adj_ja):
    """one Kipf layer fwd+bwd step threaded over rows (pull-form backward); returns
    (P, Z, dW, dX, seconds of the timed call, threads).  The transposed CSR and the coefficients are prepared
    outside the timed call."""
    import time
    x, px = _f(x); W, pW = _f(W); dz, pdz = _f(dz)
    ia = np.ascontiguousarray(adj_ia, np.int32)
    N, F = x.shape
    rowptr = (ia - 1).astype(np.int32)
    col = np.ascontiguousarray(np.asarray(adj_ja)[0] - 1, dtype=np.int32)
    deg = np.diff(ia).astype(np.int64)
    rows = np.repeat(np.arange(N, dtype=np.int64), deg)
    coef = np.power((deg[rows] * deg[col]).astype(np.float32), np.float32(-0.5)).astype(np.float32)
    order = np.argsort(col, kind="stable")                     # sources ascending inside each transposed row
    t_src = rows[order].astype(np.int32)
    t_rowptr = np.concatenate([[0], np.cumsum(np.bincount(col, minlength=N))]).astype(np.int32)
    P = np.empty((N, F), np.float32); Z = np.empty((N, F), np.float32); dP = np.empty((N, F), np.float32)
    dX = np.empty((N, F), np.float32); dW = np.empty(F * F, np.float32)
    L = lib_omp()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    t0 = time.perf_counter()
    L.oracle_omp_kipf_step(C.c_int(N), C.c_int(F), vp(rowptr), vp(col), vp(coef), vp(t_rowptr), vp(t_src), px, pW, pdz,
                           vp(P), vp(Z), vp(dW), vp(dP), vp(dX))
    dt = time.perf_counter() - t0
    return P, Z, dW, dX, dt, int(L.oracle_omp_threads())
