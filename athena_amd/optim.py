"""Device-resident tail of a train step: loss, clipping, optimiser update.

Mirrors, name for name, the reference types that network_type%update drives
(athena_network_sub.f90:2816-2929): clip_type (athena_clipper.f90), the learning-rate decays
(athena_lr_decay.f90:200-275), the regularisers (athena_regulariser.f90:40-138), sgd / adam
(athena_optimiser.f90:634-673, :1027-1091) and mse_loss_type (athena_loss.f90:393-430).
Parameters, gradients and optimiser state are flat float32 device vectors; every update is one
launch of libathena_mp (athena_amd/csrc/train.hip) -- there is no host fallback.
"""
import math

import numpy as np
import torch

from . import _capi
from .ops import _chk, _go, _p

_f32 = np.float32
_HUGE = float(np.finfo(np.float32).max)


# ---- learning-rate decay (host scalars, real32 arithmetic as in the reference) --------------------
class base_lr_decay_type:
    iterate_per_epoch = False

    def get_lr(self, learning_rate, iteration):   # lr_decay_none :200-216
        return float(_f32(learning_rate))


class exp_lr_decay_type(base_lr_decay_type):
    def __init__(self, decay_rate=0.9):
        self.decay_rate = _f32(decay_rate)

    def get_lr(self, learning_rate, iteration):   # :218-234
        return float(_f32(learning_rate) * _f32(math.exp(float(-_f32(iteration) * self.decay_rate))))


class step_lr_decay_type(base_lr_decay_type):
    def __init__(self, decay_rate=0.1, decay_steps=100):
        self.decay_rate, self.decay_steps = _f32(decay_rate), int(decay_steps)

    def get_lr(self, learning_rate, iteration):   # :236-252  (integer division in the exponent)
        return float(_f32(learning_rate) * _f32(self.decay_rate) ** _f32(iteration // self.decay_steps))


class inv_lr_decay_type(base_lr_decay_type):
    def __init__(self, decay_rate=0.001, decay_power=1.0):
        self.decay_rate, self.decay_power = _f32(decay_rate), _f32(decay_power)

    def get_lr(self, learning_rate, iteration):   # :254-272
        return float(_f32(learning_rate) * (_f32(1) + self.decay_rate * _f32(iteration)) ** (-self.decay_power))


# ---- regularisers ---------------------------------------------------------------------------------
class l1_regulariser_type:
    kind = 1

    def __init__(self, l1=0.01):
        self.l1, self.l2, self.decoupled = float(l1), 0.0, False


class l2_regulariser_type:
    kind = 2

    def __init__(self, l2=0.01, decoupled=True):   # decoupled defaults to .true. (athena_regulariser.f90:62)
        self.l1, self.l2, self.decoupled = 0.0, float(l2), bool(decoupled)


class l1l2_regulariser_type:
    kind = 3

    def __init__(self, l1=0.01, l2=0.01):
        self.l1, self.l2, self.decoupled = float(l1), float(l2), False


# ---- clipping -------------------------------------------------------------------------------------
class clip_type:
    """athena_clipper.f90: min/max clamp, then global L2-norm scaling of the flat gradient vector"""

    def __init__(self, clip_min=None, clip_max=None, clip_norm=None):
        self.l_min_max = clip_min is not None or clip_max is not None
        self.l_norm = clip_norm is not None
        self.min = -_HUGE if clip_min is None else float(clip_min)
        self.max = _HUGE if clip_max is None else float(clip_max)
        self.norm = _HUGE if clip_norm is None else float(clip_norm)

    def apply(self, gradient):
        _chk(gradient)
        if self.l_min_max or self.l_norm:
            _go()
            _capi.call("athena_mp_clip", gradient.numel(), _p(gradient), int(self.l_min_max), self.min, self.max,
                       int(self.l_norm), self.norm)
        return gradient


# ---- optimisers -----------------------------------------------------------------------------------
class base_optimiser_type:
    name = "base"

    def __init__(self, learning_rate=0.01, regulariser=None, clip_dict=None, lr_decay=None):
        self.learning_rate = float(learning_rate)
        self.regulariser = regulariser
        self.regularisation = regulariser is not None
        self.clip_dict = clip_dict or clip_type()
        self.lr_decay = lr_decay or base_lr_decay_type()
        self.iter = 0
        self.epoch = 0

    def _lr(self):
        return self.lr_decay.get_lr(self.learning_rate, self.iter)

    def _reg(self):
        r = self.regulariser
        return (r.kind, r.l1, r.l2, int(r.decoupled)) if r is not None else (0, 0.0, 0.0, 0)

    def minimise(self, param, gradient):   # minimise_base :396-418  param = param - lr * gradient
        from . import ops
        ops.axpy(-self._lr(), gradient, param)


class sgd_optimiser_type(base_optimiser_type):
    name = "sgd"

    def __init__(self, learning_rate=0.01, momentum=0.0, nesterov=False, **kw):
        super().__init__(learning_rate, **kw)
        self.momentum, self.nesterov = float(momentum), bool(nesterov)
        self.velocity = None

    def minimise(self, param, gradient):
        _chk(param); _chk(gradient, tuple(param.shape))
        if self.velocity is None:
            self.velocity = torch.zeros_like(param)       # init_gradients_sgd :615-629
        kind, l1, l2, _ = self._reg()
        _go()
        _capi.call("athena_mp_sgd_step", param.numel(), self._lr(), self.momentum, int(self.nesterov), kind, l1, l2,
                   _p(param), _p(gradient), _p(self.velocity))


class adam_optimiser_type(base_optimiser_type):
    name = "adam"

    def __init__(self, learning_rate=0.01, beta1=0.9, beta2=0.999, epsilon=1e-8, **kw):
        super().__init__(learning_rate, **kw)
        self.beta1, self.beta2, self.epsilon = float(beta1), float(beta2), float(epsilon)
        self.m = self.v = None

    def minimise(self, param, gradient):
        _chk(param); _chk(gradient, tuple(param.shape))
        if self.m is None:
            self.m, self.v = torch.zeros_like(param), torch.zeros_like(param)   # init_gradients_adam :1006-1022
        kind, l1, l2, dec = self._reg()
        _go()
        _capi.call("athena_mp_adam_step", param.numel(), self._lr(), self.beta1, self.beta2, self.epsilon, self.iter,
                   kind, l1, l2, dec, _p(param), _p(gradient), _p(self.m), _p(self.v))


# ---- loss -----------------------------------------------------------------------------------------
class mse_loss_type:
    """compute_mse (athena_loss.f90:393-430): mean((predicted - expected)^2) / 2"""

    def compute(self, predicted, expected, need_grad=True):
        _chk(predicted); _chk(expected, tuple(predicted.shape))
        loss = torch.empty(1, device=predicted.device, dtype=torch.float32)
        d = torch.empty_like(predicted) if need_grad else None
        _go()
        _capi.call("athena_mp_mse_loss", predicted.numel(), _p(predicted), _p(expected), _p(loss),
                   _p(d) if d is not None else None)
        return loss, d


# ---- network%update -------------------------------------------------------------------------------
def update(layers, optimiser, epoch=None):
    """athena_network_sub.f90:2816-2929: bump the iteration counter, gather the learnable parameters and
    gradients of every layer (layer order, then parameter-index order) into flat vectors, clip, minimise,
    write the parameters back and reset the gradients.  `layers` expose .params / .grads (lists of flat
    device tensors), as the layer mirrors in athena_amd.layers do."""
    if optimiser.lr_decay.iterate_per_epoch:
        if epoch is not None and epoch > optimiser.epoch:
            optimiser.epoch = epoch
            optimiser.iter += 1
    else:
        optimiser.iter += 1
    tensors = [(l, i) for l in layers for i in range(len(l.params))]
    for l, i in tensors:
        if l.grads[i] is None:
            raise RuntimeError("Gradient not allocated for parameters")   # :2857-2861
    params = torch.cat([l.params[i].reshape(-1) for l, i in tensors])
    grads = torch.cat([l.grads[i].reshape(-1) for l, i in tensors])
    optimiser.clip_dict.apply(grads)
    optimiser.minimise(params, grads)
    off = 0
    for l, i in tensors:
        n = l.params[i].numel()
        l.params[i].copy_(params[off:off + n].view_as(l.params[i]))
        l.grads[i] = None                                                  # reset_gradients
        off += n
    return params
