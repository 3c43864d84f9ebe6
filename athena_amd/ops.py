"""The autodiff ops of the path on device tensors -- thin wrappers over the C ABI.

Names follow the reference's op library (athena_diffstruc_extd.f90:251-318): kipf_propagate,
duvenaud_propagate, duvenaud_update, gno_*; `matmul` is diffstruc's.  Each `*_bwd*` function is
the `get_partial_*_val` callback of the corresponding op.  Tensors are torch CUDA(HIP) float32,
row-major [N, F] == athena's val(F, N).  Kernels go to torch's current stream.
"""
import ctypes as C

import numpy as np

import torch

from . import _capi
from .graph import DeviceGraph

ACT = {"none": 0, "linear": 0, "relu": 1, "sigmoid": 2, "tanh": 3}


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _chk(t, shape=None, dtype=torch.float32):
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise ValueError("expect contiguous device tensor")
    if shape is not None:
        if not (tuple(t.shape) == tuple(shape)):
            raise ValueError(f"shape {tuple(t.shape)} != {tuple(shape)}")
    return t


def _go():
    _capi.use_torch_stream()


# ---- Kipf -------------------------------------------------------------------------------------
def kipf_propagate(g: DeviceGraph, x, out=None):
    """athena_diffstruc_extd_sub_kipf.f90:7-59"""
    F = x.shape[1]
    _chk(x, (g.n_cols, F))
    y = out if out is not None else torch.empty((g.n_rows, F), device=x.device, dtype=torch.float32)
    _chk(y, (g.n_rows, F))
    _go()
    _capi.call("athena_mp_kipf_propagate_fwd", g.handle, F, _p(x), _p(y))
    return y


def kipf_propagate_act(g: DeviceGraph, x, act="none", out=None):
    """act(kipf_propagate(x)) in one launch (act in ops.ACT) -- the step of a layer that runs its dense step first"""
    F = x.shape[1]
    _chk(x, (g.n_cols, F))
    y = out if out is not None else torch.empty((g.n_rows, F), device=x.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_kipf_propagate_act_fwd", g.handle, F, _p(x), ACT[act], _p(_chk(y, (g.n_rows, F))))
    return y


def kipf_propagate_bwd(g: DeviceGraph, grad, exact=False, out=None):
    """get_partial_kipf_propagate_left_val, ..._sub_kipf.f90:85-111 (exact=False: no coefficient)"""
    F = grad.shape[1]
    _chk(grad, (g.n_rows, F))
    dx = out if out is not None else torch.empty((g.n_cols, F), device=grad.device, dtype=torch.float32)
    _chk(dx, (g.n_cols, F))
    _go()
    _capi.call("athena_mp_kipf_propagate_bwd", g.handle, F, _p(grad), _p(dx), int(bool(exact)))
    return dx


def reverse_kipf_propagate(g: DeviceGraph, a, out=None):
    """reverse_kipf_propagate, athena_diffstruc_extd_sub_kipf.f90:116-158: c[u] = sum of a[v] over the entries (v -> u),
    no coefficient (the node of the higher-order path; its value is the scatter of get_partial_kipf_propagate_left_val)"""
    F = a.shape[1]
    _chk(a, (g.n_rows, F))
    c = out if out is not None else torch.empty((g.n_cols, F), device=a.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_reverse_kipf_propagate_fwd", g.handle, F, _p(a), _p(_chk(c, (g.n_cols, F))))
    return c


def reverse_kipf_propagate_partial(g: DeviceGraph, upstream, val_form=False, out=None):
    """partial of reverse_kipf_propagate wrt its operand: the function form (:159-175) is kipf_propagate(upstream) WITH
    the coefficient, the `_val` form (:176-205) is the coefficient-free scatter again -- the reference as written"""
    F = upstream.shape[1]
    if val_form:
        _chk(upstream, (g.n_rows, F))
        y = out if out is not None else torch.empty((g.n_cols, F), device=upstream.device, dtype=torch.float32)
        _go()
        _capi.call("athena_mp_reverse_kipf_propagate_partial_val", g.handle, F, _p(upstream), _p(_chk(y, (g.n_cols, F))))
        return y
    _chk(upstream, (g.n_cols, F))
    y = out if out is not None else torch.empty((g.n_rows, F), device=upstream.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_reverse_kipf_propagate_partial", g.handle, F, _p(upstream), _p(_chk(y, (g.n_rows, F))))
    return y


def kipf_propagate_bwd_dual(g: DeviceGraph, grad):
    """both reverse forms of kipf_propagate from one gather of the upstream rows: (coefficient-free scatter of the
    reference, adjoint of the forward)"""
    F = grad.shape[1]
    _chk(grad, (g.n_rows, F))
    plain = torch.empty((g.n_cols, F), device=grad.device, dtype=torch.float32)
    coef = torch.empty((g.n_cols, F), device=grad.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_kipf_propagate_bwd_dual", g.handle, F, _p(grad), _p(plain), _p(coef))
    return plain, coef


def pull_dual(g: DeviceGraph, x, plain=None, coef=None):
    """over the forward rows of g: (sum of the listed rows, coefficient-weighted sum) from one gather -- the pull
    form of kipf_propagate_bwd_dual on a symmetric row shard (dist.py)"""
    F = x.shape[1]
    _chk(x, (g.n_cols, F))
    plain = plain if plain is not None else torch.empty((g.n_rows, F), device=x.device, dtype=torch.float32)
    coef = coef if coef is not None else torch.empty((g.n_rows, F), device=x.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_kipf_propagate_fwd_dual", g.handle, F, _p(x), _p(_chk(plain, (g.n_rows, F))),
               _p(_chk(coef, (g.n_rows, F))))
    return plain, coef


def neighbour_sum(g: DeviceGraph, x, out=None):
    """y[v,:] = sum_{w in row v} x[col[w],:] (no coefficient): the pull form of the reference's
    coefficient-free backward on a symmetric graph shard (= duvenaud_propagate with F_e = 0)"""
    F = x.shape[1]
    _chk(x, (g.n_cols, F))
    y = out if out is not None else torch.empty((g.n_rows, F), device=x.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_duvenaud_propagate_fwd", g.handle, F, 0, _p(x), None, _p(y))
    return y


def gather_rows(x, idx, out=None):
    """out[r,:] = x[idx[r],:] -- halo packing"""
    n, F = idx.numel(), x.shape[1]
    _chk(x); _chk(idx, dtype=torch.int32)
    y = out if out is not None else torch.empty((n, F), device=x.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_gather_rows", n, F, _p(idx), _p(x), _p(y))
    return y


# ---- dense contraction (diffstruc matmul) ---------------------------------------------------------
def matmul(W, P, Fo, bias=None, act="none", out=None):
    """Z[N,Fo] = act(P[N,Fi] . Wt + bias);  W flat params%val(:,1) = W(Fo,Fi) column-major."""
    N, Fi = P.shape
    _chk(P)
    if not (W.numel() == Fo * Fi):
        raise ValueError('expected: W.numel() == Fo * Fi')
    Z = out if out is not None else torch.empty((N, Fo), device=P.device, dtype=torch.float32)
    _chk(Z, (N, Fo))
    if bias is not None:
        _chk(bias, (Fo,))
    _go()
    _capi.call("athena_mp_gemm_fwd", N, Fi, Fo, _p(P), _p(_chk(W)), _p(bias), ACT[act], _p(Z))
    return Z


def matmul_dw(P, dZ, out=None):
    """dW(Fo,Fi) flat = sum_v dZ[v,:] (x) P[v,:]"""
    N, Fi = P.shape
    Fo = dZ.shape[1]
    _chk(P); _chk(dZ, (N, Fo))
    dW = out if out is not None else torch.empty(Fo * Fi, device=P.device, dtype=torch.float32)
    if _chk(dW).numel() != Fo * Fi:
        raise ValueError(f"dW holds {dW.numel()} values, expected {Fo * Fi}")
    _go()
    _capi.call("athena_mp_gemm_dw", N, Fi, Fo, _p(P), _p(dZ), _p(dW))
    return dW


def matmul_dx(W, dZ, Fi, out=None):
    """dP[N,Fi] = dZ[N,Fo] . W"""
    N, Fo = dZ.shape
    _chk(dZ)
    if not (W.numel() == Fo * Fi):
        raise ValueError('expected: W.numel() == Fo * Fi')
    dP = out if out is not None else torch.empty((N, Fi), device=dZ.device, dtype=torch.float32)
    _chk(dP, (N, Fi))
    _go()
    _capi.call("athena_mp_gemm_dx", N, Fi, Fo, _p(dZ), _p(_chk(W)), _p(dP))
    return dP


def kipf_layer_fwd(g: DeviceGraph, x, W, Fo, bias=None, act="none", P=None, Z=None, keep_P=True):
    """one Kipf time step in one launch: P = kipf_propagate(x), Z = act(P . Wt + bias).  keep_P=False: P is not
    stored (returns (None, Z)) -- for a reverse pass through kipf_layer_bwd, which works from x"""
    Fi = x.shape[1]
    _chk(x, (g.n_cols, Fi))
    if keep_P:
        P = P if P is not None else torch.empty((g.n_rows, Fi), device=x.device, dtype=torch.float32)
        _chk(P, (g.n_rows, Fi))
    else:
        P = None
    Z = Z if Z is not None else torch.empty((g.n_rows, Fo), device=x.device, dtype=torch.float32)
    _chk(Z, (g.n_rows, Fo))
    if _chk(W).numel() != Fo * Fi:
        raise ValueError(f"W holds {W.numel()} values, expected Fo*Fi = {Fo * Fi}")
    if bias is not None:
        _chk(bias, (Fo,))
    _go()
    _capi.call("athena_mp_kipf_layer_fwd", g.handle, Fi, Fo, _p(x), _p(_chk(W)), _p(bias), ACT[act], _p(P), _p(Z))
    return P, Z


def kipf_layer_bwd_x(g: DeviceGraph, dZ, W, Fi, exact=False, out=None):
    """dX = A^T (dZ . W) in one launch (aggregate first, then contract)"""
    Fo = dZ.shape[1]
    _chk(dZ, (g.n_rows, Fo))
    dX = out if out is not None else torch.empty((g.n_cols, Fi), device=dZ.device, dtype=torch.float32)
    _chk(dX, (g.n_cols, Fi))
    if _chk(W).numel() != Fo * Fi:
        raise ValueError(f"W holds {W.numel()} values, expected Fo*Fi = {Fo * Fi}")
    _go()
    _capi.call("athena_mp_kipf_layer_bwd_x", g.handle, Fi, Fo, _p(dZ), _p(_chk(W)), int(bool(exact)), _p(dX))
    return dX


def kipf_layer_bwd(g: DeviceGraph, dZ, W, x, exact=False, need_dx=True, dX=None, dW=None):
    """the whole reverse pass of one Kipf step from its input x (no stored P): returns (dX or None, dW) with
    dX = (A^T dZ) . W (exact=False: the reference's coefficient-free scatter) and dW = dZ . (A^ x)^T"""
    Fo, Fi = dZ.shape[1], x.shape[1]
    _chk(dZ, (g.n_rows, Fo)); _chk(x, (g.n_cols, Fi))
    if _chk(W).numel() != Fo * Fi:
        raise ValueError(f"W holds {W.numel()} values, expected Fo*Fi = {Fo * Fi}")
    if need_dx:
        dX = dX if dX is not None else torch.empty((g.n_cols, Fi), device=dZ.device, dtype=torch.float32)
        _chk(dX, (g.n_cols, Fi))
    else:
        dX = None
    dW = dW if dW is not None else torch.empty(Fo * Fi, device=dZ.device, dtype=torch.float32)
    if _chk(dW).numel() != Fo * Fi:
        raise ValueError(f"dW holds {dW.numel()} values, expected {Fo * Fi}")
    _go()
    _capi.call("athena_mp_kipf_layer_bwd", g.handle, Fi, Fo, _p(dZ), _p(W), _p(x), int(bool(exact)), _p(dX), _p(dW))
    return dX, dW


def pull_gemm(g: DeviceGraph, dZ, W, Fi, exact=False, out=None):
    """dX[v,:] = (sum over the forward row v of g of dZ[col,:]) . W -- shard backward (dist.py)"""
    Fo = dZ.shape[1]
    _chk(dZ, (g.n_cols, Fo))
    dX = out if out is not None else torch.empty((g.n_rows, Fi), device=dZ.device, dtype=torch.float32)
    _chk(dX, (g.n_rows, Fi))
    if _chk(W).numel() != Fo * Fi:
        raise ValueError(f"W holds {W.numel()} values, expected Fo*Fi = {Fo * Fi}")
    _go()
    _capi.call("athena_mp_pull_gemm", g.handle, Fi, Fo, _p(dZ), _p(_chk(W)), int(bool(exact)), _p(dX))
    return dX


def _act_code(kind):
    if kind not in ACT:
        raise ValueError(f"activation '{kind}' is not built on the HIP path (built: {sorted(ACT)} + softmax, swish)")
    return ACT[kind]


# activations whose reverse pass needs the INPUT (pre-activation) rather than the output
NEEDS_INPUT = ("swish", "leaky_relu", "selu", "gaussian", "piecewise")

# activations with attributes: name -> (C-ABI kind, (first attribute, default), (second attribute, default));
# defaults are the reference's reset_* values (athena_activation_leaky_relu.f90:83-85, _selu.f90:95-99,
# _gaussian.f90:92-96, _piecewise.f90:80-84, _relu.f90:79-80)
ACTP = {"linear": (0, None, None), "none": (0, None, None), "relu": (1, ("threshold", 0.0), None),
        "sigmoid": (2, None, None), "tanh": (3, None, None), "leaky_relu": (4, ("alpha", 0.01), None),
        "selu": (5, ("alpha", 1.67326), ("lambda_", 1.0507)), "gaussian": (6, ("sigma", 1.5), ("mu", 0.0)),
        "piecewise": (7, ("gradient", 0.1), ("limit", 1.0))}


class actv_type:
    """base_actv_type with its attributes (athena_misc_types / athena_activation_*.f90 `initialise`): a layer's
    `activation` argument may be a name or one of these.  `scale` multiplies the output only when it differs from
    1 by more than 1e-6, as `apply_scaling` does."""

    def __init__(self, name, scale=1.0, **attrs):
        if name not in ACTP and name not in ("swish", "softmax"):
            raise ValueError(f"unknown activation '{name}'")
        self.name = name
        self.scale = float(scale)
        self.apply_scaling = abs(np.float32(self.scale) - np.float32(1)) > np.float32(1e-6)
        self.beta = float(attrs.pop("beta", 1.0))                      # swish
        self.p = [0.0, 0.0]
        if name in ACTP:
            for k, spec in enumerate(ACTP[name][1:]):
                if spec is not None:
                    self.p[k] = float(attrs.pop(spec[0], attrs.pop(spec[0].rstrip("_"), spec[1])))
        if attrs:
            raise ValueError(f"activation '{name}' has no attribute(s) {sorted(attrs)}")

    def simple(self):
        """the plain name when the attributes are the defaults of an activation the fused epilogues know"""
        plain = not self.apply_scaling and self.p == [0.0, 0.0] and self.beta == 1.0
        return self.name if plain and (self.name in ACT or self.name in ("softmax", "swish")) else None

    def __repr__(self):
        return f"actv_type({self.name!r}, scale={self.scale}, p={self.p})"


def resolve_activation(a):
    """a name stays a name; an actv_type collapses to its name when it carries no non-default attribute"""
    if isinstance(a, actv_type):
        return a.simple() or a
    return a


def needs_input(kind):
    """does the reverse pass of this activation take the pre-activation? (else it takes the output)"""
    return isinstance(kind, actv_type) or kind in NEEDS_INPUT


def _actp_args(kind):
    a = kind if isinstance(kind, actv_type) else actv_type(kind)
    if a.name not in ACTP:
        raise ValueError(f"activation '{a.name}' takes no scale / attributes on the HIP path")
    return ACTP[a.name][0], (a.scale if a.apply_scaling else 1.0), a.p[0], a.p[1]


def activation(kind, z, out=None, beta=1.0):
    """athena_activation_*.f90 apply: element-wise kinds, plus 'softmax' (over the features of each vertex,
    z must be [N, F]) and 'swish' (x * sigmoid(beta x))"""
    y = out if out is not None else torch.empty_like(z)
    _go()
    kind = resolve_activation(kind)
    if isinstance(kind, actv_type) and kind.name == "swish" and not kind.apply_scaling:
        kind, beta = "swish", kind.beta
    if isinstance(kind, actv_type) or kind in ("leaky_relu", "selu", "gaussian", "piecewise"):
        code, scale, p0, p1 = _actp_args(kind)
        _capi.call("athena_mp_activation_param_fwd", code, z.numel(), scale, p0, p1, _p(_chk(z)), _p(y))
    elif kind == "softmax":
        if not (z.dim() == 2):
            raise ValueError('expected: z.dim() == 2')
        _capi.call("athena_mp_softmax_fwd", z.shape[0], z.shape[1], _p(_chk(z)), _p(y))
    elif kind == "swish":
        _capi.call("athena_mp_swish_fwd", z.numel(), float(beta), _p(_chk(z)), _p(y))
    else:
        _capi.call("athena_mp_activation_fwd", _act_code(kind), z.numel(), _p(_chk(z)), _p(y))
    return y


def activation_bwd(kind, y, g, out=None, z=None, beta=1.0):
    """reverse factor; y = the activation's OUTPUT, z = its input (required by 'swish' only)"""
    dz = out if out is not None else torch.empty_like(y)
    _go()
    kind = resolve_activation(kind)
    if isinstance(kind, actv_type) and kind.name == "swish" and not kind.apply_scaling:
        kind, beta = "swish", kind.beta
    if isinstance(kind, actv_type) or kind in ("leaky_relu", "selu", "gaussian", "piecewise"):
        if z is None:
            raise ValueError("activations with attributes differentiate at their input: pass z")
        code, scale, p0, p1 = _actp_args(kind)
        _capi.call("athena_mp_activation_param_bwd", code, z.numel(), scale, p0, p1, _p(_chk(z)), _p(_chk(g)), _p(dz))
    elif kind == "softmax":
        if not (y.dim() == 2):
            raise ValueError('expected: y.dim() == 2')
        _capi.call("athena_mp_softmax_bwd", y.shape[0], y.shape[1], _p(_chk(y)), _p(_chk(g)), _p(dz))
    elif kind == "swish":
        if z is None:
            raise ValueError("swish differentiates at its input: pass z")
        _capi.call("athena_mp_swish_bwd", z.numel(), float(beta), _p(_chk(z)), _p(_chk(g)), _p(dz))
    else:
        _capi.call("athena_mp_activation_bwd", _act_code(kind), y.numel(), _p(_chk(y)), _p(_chk(g)), _p(dz))
    return dz


def concat_features(a, b):
    """'concatenate' merge of two layer inputs: out[v] = [a[v], b[v]]"""
    _chk(a); _chk(b)
    if not (a.shape[0] == b.shape[0]):
        raise ValueError('expected: a.shape[0] == b.shape[0]')
    out = torch.empty((a.shape[0], a.shape[1] + b.shape[1]), device=a.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_concat_fwd", a.shape[0], a.shape[1], b.shape[1], _p(a), _p(b), _p(out))
    return out


def concat_features_bwd(grad, Fa, need_a=True, need_b=True):
    """split the gradient of concat_features; returns (da, db) (None where not needed)"""
    _chk(grad)
    N, F = grad.shape
    da = torch.empty((N, Fa), device=grad.device, dtype=torch.float32) if need_a else None
    db = torch.empty((N, F - Fa), device=grad.device, dtype=torch.float32) if need_b else None
    _go()
    _capi.call("athena_mp_concat_bwd", N, Fa, F - Fa, _p(grad), _p(da) if need_a else None, _p(db) if need_b else None)
    return da, db


def axpy(alpha, x, y):
    _go()
    _capi.call("athena_mp_axpy", x.numel(), float(alpha), _p(_chk(x)), _p(_chk(y)))
    return y


# ---- Duvenaud -------------------------------------------------------------------------------------
def duvenaud_propagate(g: DeviceGraph, x, e, out=None):
    """athena_diffstruc_extd_sub_duvenaud.f90:7-59"""
    Fv, Fe = x.shape[1], e.shape[1]
    _chk(x, (g.n_cols, Fv)); _chk(e, (g.n_edge_cols, Fe))
    c = out if out is not None else torch.empty((g.n_rows, Fv + Fe), device=x.device, dtype=torch.float32)
    _chk(c, (g.n_rows, Fv + Fe))
    _go()
    _capi.call("athena_mp_duvenaud_propagate_fwd", g.handle, Fv, Fe, _p(x), _p(e), _p(c))
    return c


def duvenaud_propagate_edges(g: DeviceGraph, e, out=None):
    """the edge part of duvenaud_propagate alone: a_e[v, :] = sum of e[edge of entry w, :] over row v, [n_rows, Fe] -- the same sums in
    the same order as columns Fv .. of the packed form (the x part alone: neighbour_sum)"""
    Fe = e.shape[1]
    _chk(e, (g.n_edge_cols, Fe))
    c = out if out is not None else torch.empty((g.n_rows, Fe), device=e.device, dtype=torch.float32)
    _chk(c, (g.n_rows, Fe))
    _go()
    _capi.call("athena_mp_duvenaud_propagate_fwd", g.handle, 0, Fe, None, _p(e), _p(c))
    return c


def duvenaud_propagate_bwd_x(g: DeviceGraph, grad, Fv):
    Fe = grad.shape[1] - Fv
    _chk(grad, (g.n_rows, Fv + Fe))
    dx = torch.empty((g.n_cols, Fv), device=grad.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_duvenaud_propagate_bwd_x", g.handle, Fv, Fe, _p(grad), _p(dx))
    return dx


def duvenaud_propagate_bwd_e(g: DeviceGraph, grad, Fv):
    Fe = grad.shape[1] - Fv
    _chk(grad, (g.n_rows, Fv + Fe))
    de = torch.empty((g.n_edge_cols, Fe), device=grad.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_duvenaud_propagate_bwd_e", g.handle, Fv, Fe, _p(grad), _p(de))
    return de


def duvenaud_update(g: DeviceGraph, a, weight, min_deg, max_deg, Fo):
    """athena_diffstruc_extd_sub_duvenaud.f90:176-228"""
    Fi = a.shape[1]
    _chk(a, (g.n_rows, Fi))
    if not (weight.numel() == Fo * Fi * (max_deg - min_deg + 1)):
        raise ValueError('expected: weight.numel() == Fo * Fi * (max_deg - min_deg + 1)')
    c = torch.empty((g.n_rows, Fo), device=a.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_duvenaud_update_fwd", g.handle, Fi, Fo, min_deg, max_deg, _p(a), _p(_chk(weight)), _p(c))
    return c


def duvenaud_update_act(g: DeviceGraph, a, weight, min_deg, max_deg, Fo, act="none"):
    """z = act(duvenaud_update(a)) with the activation in the kernel epilogue"""
    Fi = a.shape[1]
    _chk(a, (g.n_rows, Fi))
    if not (weight.numel() == Fo * Fi * (max_deg - min_deg + 1)):
        raise ValueError('expected: weight.numel() == Fo * Fi * (max_deg - min_deg + 1)')
    z = torch.empty((g.n_rows, Fo), device=a.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_duvenaud_update_act_fwd", g.handle, Fi, Fo, min_deg, max_deg, _p(a), _p(_chk(weight)), ACT[act], _p(z))
    return z


def duvenaud_update_act_readout(g: DeviceGraph, a, weight, min_deg, max_deg, Fo, R, O, act="none"):
    """(z, p): z = act(duvenaud_update(a)) and the readout's p = softmax(z R^T) per vertex from one launch"""
    Fi = a.shape[1]
    _chk(a, (g.n_rows, Fi))
    if not (weight.numel() == Fo * Fi * (max_deg - min_deg + 1) and R.numel() == O * Fo):
        raise ValueError('expected: weight.numel() == Fo * Fi * D and R.numel() == O * Fo')
    z = torch.empty((g.n_rows, Fo), device=a.device, dtype=torch.float32)
    p = torch.empty((g.n_rows, O), device=a.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_duvenaud_update_readout_fwd", g.handle, Fi, Fo, min_deg, max_deg, _p(a), _p(_chk(weight)), ACT[act],
               _p(z), O, _p(_chk(R)), _p(p))
    return z, p


def duvenaud_update_act_readout_split(g: DeviceGraph, a_x, a_e, weight, min_deg, max_deg, Fo, R, O, act="none"):
    """duvenaud_update_act_readout with a split: a_x [n, Fv] (neighbour_sum of the vertex features) and a_e [n, Fe]
    (duvenaud_propagate_edges: the same at every time step of a layer)"""
    Fv, Fe = a_x.shape[1], a_e.shape[1]
    _chk(a_x, (g.n_rows, Fv)); _chk(a_e, (g.n_rows, Fe))
    if not (weight.numel() == Fo * (Fv + Fe) * (max_deg - min_deg + 1) and R.numel() == O * Fo):
        raise ValueError('expected: weight.numel() == Fo * (Fv + Fe) * D and R.numel() == O * Fo')
    z = torch.empty((g.n_rows, Fo), device=a_x.device, dtype=torch.float32)
    p = torch.empty((g.n_rows, O), device=a_x.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_duvenaud_update_readout_fwd_split", g.handle, Fv, Fe, Fo, min_deg, max_deg, _p(a_x), _p(a_e), _p(_chk(weight)),
               ACT[act], _p(z), O, _p(_chk(R)), _p(p))
    return z, p


def duvenaud_update_bwd_a(g: DeviceGraph, grad, weight, min_deg, max_deg, Fi):
    Fo = grad.shape[1]
    _chk(grad, (g.n_rows, Fo))
    da = torch.empty((g.n_rows, Fi), device=grad.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_duvenaud_update_bwd_a", g.handle, Fi, Fo, min_deg, max_deg, _p(grad), _p(_chk(weight)), _p(da))
    return da


def duvenaud_update_bwd_w(g: DeviceGraph, grad, a, min_deg, max_deg):
    Fo, Fi = grad.shape[1], a.shape[1]
    _chk(grad, (g.n_rows, Fo)); _chk(a, (g.n_rows, Fi))
    dW = torch.empty(Fo * Fi * (max_deg - min_deg + 1), device=grad.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_duvenaud_update_bwd_w", g.handle, Fi, Fo, min_deg, max_deg, _p(grad), _p(a), _p(dW))
    return dW


def duvenaud_update_bwd(g: DeviceGraph, grad, a, weight, min_deg, max_deg):
    """(da, dW): both reverse products of duvenaud_update from one pass over grad (athena_mp_duvenaud_update_bwd)"""
    Fo, Fi = grad.shape[1], a.shape[1]
    _chk(grad, (g.n_rows, Fo)); _chk(a, (g.n_rows, Fi))
    if not (weight.numel() == Fo * Fi * (max_deg - min_deg + 1)):
        raise ValueError('expected: weight.numel() == Fo * Fi * (max_deg - min_deg + 1)')
    da = torch.empty((g.n_rows, Fi), device=grad.device, dtype=torch.float32)
    dW = torch.empty(Fo * Fi * (max_deg - min_deg + 1), device=grad.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_duvenaud_update_bwd", g.handle, Fi, Fo, min_deg, max_deg, _p(grad), _p(a), _p(_chk(weight)), _p(da), _p(dW))
    return da, dW


def duvenaud_update_bwd_split(g: DeviceGraph, grad, a, weight, min_deg, max_deg, Fv):
    """(da_x [n, Fv], da_e [n, Fe], dW): duvenaud_update_bwd with da written split where it is produced
    (athena_mp_duvenaud_update_bwd_split); duvenaud_propagate_bwd_x(g, da_x, Fv) and duvenaud_propagate_bwd_e(g, da_e, 0)
    take the halves -- same sums in the same order as from the packed da"""
    Fo, Fi = grad.shape[1], a.shape[1]
    Fe = Fi - Fv
    if not (0 < Fv < Fi):
        raise ValueError("duvenaud_update_bwd_split: needs vertex AND edge columns")
    _chk(grad, (g.n_rows, Fo)); _chk(a, (g.n_rows, Fi))
    if not (weight.numel() == Fo * Fi * (max_deg - min_deg + 1)):
        raise ValueError('expected: weight.numel() == Fo * Fi * (max_deg - min_deg + 1)')
    da_x = torch.empty((g.n_rows, Fv), device=grad.device, dtype=torch.float32)
    da_e = torch.empty((g.n_rows, Fe), device=grad.device, dtype=torch.float32)
    dW = torch.empty(Fo * Fi * (max_deg - min_deg + 1), device=grad.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_duvenaud_update_bwd_split", g.handle, Fv, Fe, Fo, min_deg, max_deg, _p(grad), _p(a), _p(_chk(weight)), _p(da_x),
               _p(da_e), _p(dW))
    return da_x, da_e, dW


def softmax_segsum(logits, seg, out=None):
    """p = softmax over outputs per vertex; out[s] (+)= sum of p over graph s.  Returns (p, out)."""
    N, O = logits.shape
    S = seg.numel() - 1
    _chk(logits); _chk(seg, dtype=torch.int32)
    p = torch.empty_like(logits)
    acc = out is not None
    if out is None:
        out = torch.empty((S, O), device=logits.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_softmax_segsum_fwd", O, N, S, _p(seg), _p(logits), _p(p), _p(out), int(acc))
    return p, out


def softmax_segsum_bwd(p, seg, gout):
    N, O = p.shape
    S = seg.numel() - 1
    _chk(p); _chk(gout, (S, O))
    dz = torch.empty_like(p)
    _go()
    _capi.call("athena_mp_softmax_segsum_bwd", O, N, S, _p(seg), _p(p), _p(gout), _p(dz))
    return dz


def segment_sum(p, seg, out=None):
    """out[s] (+)= sum of p over the vertices of graph s"""
    N, O = p.shape
    S = seg.numel() - 1
    _chk(p); _chk(seg, dtype=torch.int32)
    acc = out is not None
    if out is None:
        out = torch.empty((S, O), device=p.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_segment_sum", O, N, S, _p(seg), _p(p), _p(out), int(acc))
    return out


def segment_sum_bwd(gout, seg, N):
    S, O = gout.shape
    _chk(gout); _chk(seg, dtype=torch.int32)
    dp = torch.empty((N, O), device=gout.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_segment_sum_bwd", O, N, S, _p(seg), _p(gout), _p(dp))
    return dp


def duvenaud_readout(R, z, seg, O, out=None):
    """One launch: p = softmax(z R^T) per vertex, out[s] (+)= sum of p over graph s.  Returns (p, out)."""
    N, Fv = z.shape
    S = seg.numel() - 1
    _chk(z); _chk(seg, dtype=torch.int32); _chk(R)
    if not (R.numel() == O * Fv):
        raise ValueError('expected: R.numel() == O * Fv')
    p = torch.empty((N, O), device=z.device, dtype=torch.float32)
    acc = out is not None
    if out is None:
        out = torch.empty((S, O), device=z.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_duvenaud_readout_fwd", N, Fv, O, S, _p(seg), _p(z), _p(R), _p(p), _p(out), int(acc))
    return p, out


def duvenaud_readout_bwd(R, z, p, seg, gout, act="none", dz_next=None):
    """Reverse of duvenaud_readout and of the activation that produced z.
    Returns (dc, dR) with dc = act'(z) * (dl R + dz_next)."""
    N, Fv = z.shape
    O = p.shape[1]
    S = seg.numel() - 1
    _chk(z); _chk(p, (N, O)); _chk(gout, (S, O)); _chk(R)
    if dz_next is not None:
        _chk(dz_next, (N, Fv))
    dc = torch.empty_like(z)
    dR = torch.empty(O * Fv, device=z.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_duvenaud_readout_bwd", N, Fv, O, S, _p(seg), _p(z), _p(R), _p(p), _p(gout),
               _p(dz_next) if dz_next is not None else None, ACT[act], _p(dc), _p(dR), 0)
    return dc, dR


def duvenaud_readout_update_bwd(g: DeviceGraph, R, z, p, seg, gout, a, weight, min_deg, max_deg, Fv, act="none", dz_next=None,
                                dR=None, da_e=None, a_e=None):
    """One time step of the layer's reverse pass in one call: duvenaud_readout_bwd then duvenaud_update_bwd_split without dc in
    HBM where the shape allows it.  Returns (da_x, da_e, dW, dR); dR given: added to; da_e given: added to (the edge-feature
    gradient is linear in it: sum over the time steps, scatter once); a_e given: a arrives split, a = a_x [n, Fv]."""
    N = a.shape[0]
    if a_e is not None:
        _chk(a_e, (N, a_e.shape[1]))
        Fi = Fv + a_e.shape[1]
        _chk(a, (g.n_rows, Fv))
    else:
        Fi = a.shape[1]
        _chk(a, (g.n_rows, Fi))
    Fe = Fi - Fv
    O = p.shape[1]
    S = seg.numel() - 1
    _chk(z, (N, Fv)); _chk(p, (N, O)); _chk(gout, (S, O)); _chk(R); _chk(weight)
    if not (Fe > 0 and R.numel() == O * Fv and weight.numel() == (max_deg - min_deg + 1) * Fv * Fi):
        raise ValueError('expected: Fe > 0, R.numel() == O * Fv, weight.numel() == buckets * Fv * (Fv + Fe)')
    if dz_next is not None:
        _chk(dz_next, (N, Fv))
    da_x = torch.empty((N, Fv), device=a.device, dtype=torch.float32)
    acc_e = da_e is not None
    if acc_e:
        _chk(da_e, (N, Fe))
    else:
        da_e = torch.empty((N, Fe), device=a.device, dtype=torch.float32)
    dW = torch.empty(weight.numel(), device=a.device, dtype=torch.float32)
    acc = dR is not None
    if not acc:
        dR = torch.empty(O * Fv, device=a.device, dtype=torch.float32)
    else:
        _chk(dR)
    _go()
    _capi.call("athena_mp_duvenaud_readout_update_bwd", g.handle, Fv, Fe, min_deg, max_deg, O, S, _p(seg), _p(z), _p(R), _p(p), _p(gout),
               _p(dz_next) if dz_next is not None else None, ACT[act], _p(a), _p(weight), _p(da_x), _p(da_e), _p(dW), _p(dR), 1 if acc else 0,
               1 if acc_e else 0, _p(a_e) if a_e is not None else None)
    return da_x, da_e, dW, dR


# ---- graph neural operator -----------------------------------------------------------------------
def gno_aggregate(g: DeviceGraph, theta, coords, x, d, H, Fo, out=None):
    """gno_kernel_eval + gno_aggregate (athena_diffstruc_extd_sub_nop.f90:26-115, :330-397) fused"""
    Fi = x.shape[1]
    _chk(x, (g.n_cols, Fi)); _chk(coords, (g.n_edge_cols, d)); _chk(theta)
    if not (theta.numel() == H * d + H + Fo * Fi * H + Fo * Fi):
        raise ValueError('expected: theta.numel() == H * d + H + Fo * Fi * H + Fo * Fi')
    m = out if out is not None else torch.empty((g.n_rows, Fo), device=x.device, dtype=torch.float32)
    _chk(m, (g.n_rows, Fo))
    _go()
    _capi.call("athena_mp_gno_aggregate_fwd", g.handle, d, H, Fi, Fo, _p(theta), _p(coords), _p(x), _p(m))
    return m


def gno_saved_bytes(g: DeviceGraph, d, H, Fi, Fo) -> int:
    """size of the S the training-mode forward keeps for this graph and shape; 0: the shape does not keep S"""
    import ctypes
    b = ctypes.c_int64(0)
    _capi.call("athena_mp_gno_saved_bytes", g.handle, d, H, Fi, Fo, ctypes.byref(b))
    return int(b.value)


def gno_aggregate_save(g: DeviceGraph, theta, coords, x, d, H, Fo, s_save=None, out=None):
    """`gno_aggregate` that also keeps S = sum_e [h_e;1] x_j^T per vertex for `gno_aggregate_bwd_theta(..., s_save=)`
    (the reference keeps kappa [Fo*Fi, E] on its tape, athena_diffstruc_extd_sub_nop.f90:330-397).  Returns (m, s_save);
    `s_save` may be a buffer from an earlier step of at least gno_saved_bytes() bytes."""
    Fi = x.shape[1]
    _chk(x, (g.n_cols, Fi)); _chk(coords, (g.n_edge_cols, d)); _chk(theta)
    if not (theta.numel() == H * d + H + Fo * Fi * H + Fo * Fi):
        raise ValueError('expected: theta.numel() == H * d + H + Fo * Fi * H + Fo * Fi')
    nbytes = gno_saved_bytes(g, d, H, Fi, Fo)
    if nbytes == 0:
        raise ValueError("gno_aggregate_save: this shape does not keep S (gno_saved_bytes() == 0)")
    if s_save is None or s_save.numel() * 4 < nbytes:
        s_save = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
    _chk(s_save)
    m = out if out is not None else torch.empty((g.n_rows, Fo), device=x.device, dtype=torch.float32)
    _chk(m, (g.n_rows, Fo))
    _go()
    _capi.call("athena_mp_gno_aggregate_fwd_save", g.handle, d, H, Fi, Fo, _p(theta), _p(coords), _p(x), _p(m), _p(s_save))
    return m, s_save


def gno_aggregate_bwd_x(g: DeviceGraph, theta, coords, grad, d, H, Fi):
    Fo = grad.shape[1]
    _chk(grad, (g.n_rows, Fo)); _chk(coords, (g.n_edge_cols, d))
    if _chk(theta).numel() != H * d + H + Fo * Fi * H + Fo * Fi:
        raise ValueError("theta: expected H*d + H + Fo*Fi*H + Fo*Fi values")
    dx = torch.empty((g.n_cols, Fi), device=grad.device, dtype=torch.float32)
    _go()
    _capi.call("athena_mp_gno_aggregate_bwd_x", g.handle, d, H, Fi, Fo, _p(theta), _p(coords), _p(grad), _p(dx))
    return dx


def gno_aggregate_bwd_x_pull(g: DeviceGraph, theta, coords, grad_ext, d, H, Fi, out=None):
    """the same gradient as a pull over the graph's OWN rows: dx[v] = sum_{w in row v} K_e^T grad_ext[col[w]] -- what a
    row block of a partitioned undirected graph evaluates once the halo rows of grad have arrived (dist.GnoShardStep)"""
    Fo = grad_ext.shape[1]
    _chk(grad_ext, (g.n_cols, Fo)); _chk(coords, (g.n_edge_cols, d))
    if _chk(theta).numel() != H * d + H + Fo * Fi * H + Fo * Fi:
        raise ValueError("theta: expected H*d + H + Fo*Fi*H + Fo*Fi values")
    dx = out if out is not None else torch.empty((g.n_rows, Fi), device=grad_ext.device, dtype=torch.float32)
    _chk(dx, (g.n_rows, Fi))
    _go()
    _capi.call("athena_mp_gno_aggregate_bwd_x_pull", g.handle, d, H, Fi, Fo, _p(theta), _p(coords), _p(grad_ext), _p(dx))
    return dx


def gno_aggregate_bwd_theta(g: DeviceGraph, theta, coords, x, grad, d, H, s_save=None):
    Fi, Fo = x.shape[1], grad.shape[1]
    _chk(x, (g.n_cols, Fi)); _chk(grad, (g.n_rows, Fo)); _chk(coords, (g.n_edge_cols, d))
    if _chk(theta).numel() != H * d + H + Fo * Fi * H + Fo * Fi:
        raise ValueError("theta: expected H*d + H + Fo*Fi*H + Fo*Fi values")
    dth = torch.empty_like(theta)
    _go()
    if s_save is not None:   # S of the forward pass of the SAME graph, theta, coords, x (gno_aggregate_save)
        if _chk(s_save).numel() * 4 < gno_saved_bytes(g, d, H, Fi, Fo):
            raise ValueError("s_save: smaller than gno_saved_bytes()")
        _capi.call("athena_mp_gno_aggregate_bwd_theta_saved", g.handle, d, H, Fi, Fo, _p(theta), _p(coords), _p(x), _p(grad),
                   _p(s_save), _p(dth))
        return dth
    _capi.call("athena_mp_gno_aggregate_bwd_theta", g.handle, d, H, Fi, Fo, _p(theta), _p(coords), _p(x), _p(grad), _p(dth))
    return dth


def gno_aggregate_bwd(g: DeviceGraph, theta, coords, x, grad, d, H, s_save=None, need_dx=True, need_dtheta=True, need_dcoords=False,
                      dx_out=None, dtheta_out=None):
    """the whole reverse pass of gno_aggregate from ONE G = g . Vmat^T (athena_mp_gno_aggregate_bwd): returns
    (dx, dtheta, dcoords, fused) with None for what was not asked; `fused` says whether the shape took the fused kernels.
    dx_out [n_cols, Fi] / dtheta_out (theta's size): written in place of fresh tensors (a resident step keeps its buffers)"""
    import ctypes
    Fi, Fo = x.shape[1], grad.shape[1]
    _chk(x, (g.n_cols, Fi)); _chk(grad, (g.n_rows, Fo)); _chk(coords, (g.n_edge_cols, d))
    if _chk(theta).numel() != H * d + H + Fo * Fi * H + Fo * Fi:
        raise ValueError("theta: expected H*d + H + Fo*Fi*H + Fo*Fi values")
    if s_save is not None and _chk(s_save).numel() * 4 < gno_saved_bytes(g, d, H, Fi, Fo):
        raise ValueError("s_save: smaller than gno_saved_bytes()")
    dx = (_chk(dx_out, (g.n_cols, Fi)) if dx_out is not None else torch.empty((g.n_cols, Fi), device=x.device, dtype=torch.float32)) \
        if need_dx else None
    if dtheta_out is not None and _chk(dtheta_out).numel() != theta.numel():
        raise ValueError("dtheta_out: expected theta's size")
    dth = (dtheta_out if dtheta_out is not None else torch.empty_like(theta)) if need_dtheta else None
    dc = torch.empty_like(coords) if need_dcoords else None
    fused = ctypes.c_int32(0)
    _go()
    _capi.call("athena_mp_gno_aggregate_bwd", g.handle, d, H, Fi, Fo, _p(theta), _p(coords), _p(x), _p(grad), _p(s_save), _p(dx), _p(dth),
               _p(dc), ctypes.byref(fused))
    return dx, dth, dc, bool(fused.value)


def gno_aggregate_bwd_coords(g: DeviceGraph, theta, coords, x, grad, d, H):
    Fi, Fo = x.shape[1], grad.shape[1]
    _chk(x, (g.n_cols, Fi)); _chk(grad, (g.n_rows, Fo)); _chk(coords, (g.n_edge_cols, d))
    if _chk(theta).numel() != H * d + H + Fo * Fi * H + Fo * Fi:
        raise ValueError("theta: expected H*d + H + Fo*Fi*H + Fo*Fi values")
    dc = torch.empty_like(coords)
    _go()
    _capi.call("athena_mp_gno_aggregate_bwd_coords", g.handle, d, H, Fi, Fo, _p(theta), _p(coords), _p(x), _p(grad), _p(dc))
    return dc
