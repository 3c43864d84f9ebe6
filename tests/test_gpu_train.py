"""GPU: the device-resident tail of a train step (loss, clipping, SGD / Adam on flat vectors) against the
oracle's restatement of athena_loss.f90 / athena_clipper.f90 / athena_regulariser.f90 /
athena_optimiser.f90, and a few whole train steps of a Kipf layer driven like network_type%train does
(forward -> mse -> reverse pass -> network%update)."""
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


@pytest.mark.parametrize("momentum,nesterov,reg", [(0.0, False, None), (0.9, False, None), (0.9, True, None),
                                                   (0.0, False, "l1"), (0.5, False, "l2"), (0.9, True, "l1l2")])
def test_sgd_steps_bit_exact(dev, oracle, momentum, nesterov, reg):
    from athena_amd import optim

    rng = np.random.default_rng(3)
    n = 70001
    p = rng.standard_normal(n).astype(np.float32); p[:3] = [0.0, -0.0, 1e-30]
    regs = {None: None, "l1": optim.l1_regulariser_type(0.02), "l2": optim.l2_regulariser_type(0.03),
            "l1l2": optim.l1l2_regulariser_type(0.02, 0.03)}
    opt = optim.sgd_optimiser_type(learning_rate=0.05, momentum=momentum, nesterov=nesterov, regulariser=regs[reg])
    pd = T(p, dev)
    v = np.zeros(n, np.float32)
    for it in range(1, 4):
        g = rng.standard_normal(n).astype(np.float32)
        opt.iter = it
        gd = T(g, dev)
        opt.minimise(pd, gd)
        p, g2, v = oracle.sgd_step(p, g, v, 0.05, momentum, nesterov, reg, 0.02, 0.03)
        assert np.array_equal(H(pd), p), f"param differs at iteration {it}"
        assert np.array_equal(H(opt.velocity), v)
        assert np.array_equal(H(gd), g2)            # the reference overwrites gradient with -lr*gradient


@pytest.mark.parametrize("reg,decoupled", [(None, False), ("l2", True), ("l2", False), ("l1", False), ("l1l2", False)])
def test_adam_steps_bit_exact(dev, oracle, reg, decoupled):
    from athena_amd import optim

    rng = np.random.default_rng(4)
    n = 50003
    p = rng.standard_normal(n).astype(np.float32)
    regs = {None: None, "l1": optim.l1_regulariser_type(0.02), "l2": optim.l2_regulariser_type(0.03, decoupled),
            "l1l2": optim.l1l2_regulariser_type(0.02, 0.03)}
    opt = optim.adam_optimiser_type(learning_rate=0.01, regulariser=regs[reg])
    pd = T(p, dev)
    m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    for it in (1, 2, 3, 50):
        g = rng.standard_normal(n).astype(np.float32)
        if it == 2:
            g[:10] = 0.0                       # sqrt(v_hat) + eps with a vanishing second moment
        opt.iter = it
        opt.minimise(pd, T(g, dev))
        p, _, m, v = oracle.adam_step(p, g, m, v, 0.01, it, reg=reg, l1=0.02, l2=0.03, decoupled=decoupled)
        assert np.array_equal(H(opt.m), m) and np.array_equal(H(opt.v), v)
        assert np.array_equal(H(pd), p), f"param differs at iteration {it}: {np.abs(H(pd) - p).max()}"


def test_adam_rejects_iteration_zero(dev):
    import torch
    from athena_amd import optim, _capi

    opt = optim.adam_optimiser_type()
    with pytest.raises(_capi.AthenaMPError, match="iteration"):
        opt.minimise(torch.zeros(4, device=dev), torch.zeros(4, device=dev))


def test_clip_and_lr_decay(dev, oracle):
    from athena_amd import optim

    rng = np.random.default_rng(5)
    g = (rng.standard_normal(300001) * 3).astype(np.float32)
    out = H(optim.clip_type(clip_min=-1.5, clip_max=2.0).apply(T(g, dev)))
    assert np.array_equal(out, oracle.clip(g, -1.5, 2.0))
    out = H(optim.clip_type(clip_norm=10.0).apply(T(g, dev)))
    # the norm is a 300 001-term fp32 sum: the oracle adds sequentially (as Fortran's sum does), the device
    # pairwise; the two differ by the rounding noise of the sequential sum (~5e-5 here), and the device
    # value is the one closer to the float64 norm
    exact = g.astype(np.float64) * min(1.0, 10.0 / np.sqrt((g.astype(np.float64) ** 2).sum()))
    assert_close(out, oracle.clip(g, clip_norm=10.0), 1e-5, "norm clip vs sequential fp32 sum (anchored on float64)", f64=exact)
    assert_close(out, exact.astype(np.float32), 1e-6, "norm clip vs float64")
    assert abs(float(np.sqrt((out.astype(np.float64) ** 2).sum())) - 10.0) < 1e-3
    small = (g * 1e-4).astype(np.float32)                      # already inside the ball: untouched
    assert np.array_equal(H(optim.clip_type(clip_norm=10.0).apply(T(small, dev))), small)
    both = H(optim.clip_type(clip_min=-1.0, clip_max=1.0, clip_norm=5.0).apply(T(g, dev)))
    gc64 = np.clip(g.astype(np.float64), -1.0, 1.0)
    assert_close(both, oracle.clip(g, -1.0, 1.0, 5.0), 1e-5, "clamp then norm (anchored on float64)",
                 f64=gc64 * min(1.0, 5.0 / np.sqrt((gc64 ** 2).sum())))
    # learning-rate decays (athena_lr_decay.f90:200-272), real32 arithmetic
    assert optim.base_lr_decay_type().get_lr(0.1, 7) == float(np.float32(0.1))
    assert np.isclose(optim.exp_lr_decay_type(0.05).get_lr(0.1, 10), 0.1 * np.exp(-0.5), rtol=1e-6)
    assert np.isclose(optim.step_lr_decay_type(0.5, 4).get_lr(0.1, 9), 0.1 * 0.25, rtol=1e-6)     # 9/4 = 2
    assert np.isclose(optim.inv_lr_decay_type(0.01, 2.0).get_lr(0.1, 100), 0.1 / 4.0, rtol=1e-6)


def test_mse_loss_and_gradient(dev, oracle):
    from athena_amd import optim

    rng = np.random.default_rng(6)
    p = rng.standard_normal((4097, 10)).astype(np.float32); e = rng.standard_normal((4097, 10)).astype(np.float32)
    loss, d = optim.mse_loss_type().compute(T(p, dev), T(e, dev))
    lo, do = oracle.mse(p, e)
    exact = float(((p.astype(np.float64) - e) ** 2).mean() / 2)
    assert abs(float(loss.item()) - exact) <= 1e-6 * exact          # pairwise device sum vs float64
    # vs the oracle's sequential fp32 sum: within 1e-5, or no further from float64 than the oracle is, plus 1e-5
    assert abs(float(loss.item()) - lo) <= 1e-5 * abs(lo) or abs(float(loss.item()) - exact) <= abs(lo - exact) + 1e-5 * exact
    assert np.array_equal(H(d), do)


def test_train_steps_of_a_kipf_layer_follow_the_oracle(dev, oracle):
    """forward -> mse -> reverse -> clip -> adam, three times, exactly as the oracle restatement does on the host"""
    from athena_amd import optim
    from athena_amd.layers import kipf_msgpass_layer_type

    rng = np.random.default_rng(7)
    n = 300
    pairs = np.array([[i, i + 1] for i in range(1, n)] + [[int(a), int(b)] for a, b in rng.integers(1, n + 1, (400, 2)) if a != b]).T
    g = csr_from_index_list(n, pairs, self_loops=True)
    nvf = [16, 32, 8]
    layer = kipf_msgpass_layer_type(num_vertex_features=nvf, num_time_steps=2, activation="tanh", seed=3)
    layer.set_graph([g])
    x = rng.uniform(-1, 1, (n, 16)).astype(np.float32)
    y = rng.uniform(-1, 1, (n, 8)).astype(np.float32)
    opt = optim.adam_optimiser_type(learning_rate=0.01, clip_dict=optim.clip_type(clip_norm=0.5),
                                    regulariser=optim.l2_regulariser_type(1e-3, decoupled=True))
    params0 = layer.get_params().copy()

    def host_pass():     # the oracle restatement of the three updates (fp32 oracle, or its float64 twin as the yardstick)
        params = params0.astype(ol._REAL)
        m = np.zeros_like(params); v = np.zeros_like(params)
        rec = []
        for it in range(1, 4):
            plist = [params[:16 * 32], params[16 * 32:]]
            outs, tapes = ol.kipf_forward([g], [x], plist, nvf, "tanh")
            lo, do = ol.o.mse(outs[0], y)
            _, grads = ol.kipf_backward([g], tapes, plist, nvf, "tanh", [do])
            gflat = ol.o.clip(np.concatenate(grads), clip_norm=0.5)
            params, _, m, v = ol.o.adam_step(params, gflat, m, v, 0.01, it, reg="l2", l2=1e-3, decoupled=True)
            rec.append((lo, params))
        return rec
    ref = host_pass()
    hi = ol.f64_lazy(lambda: [p for _, p in host_pass()])
    losses = []
    for it in range(1, 4):
        out = layer.forward([x])
        loss, d = optim.mse_loss_type().compute(out, T(y, dev))
        layer.backward(d, need_input_grad=False)
        optim.update([layer], opt)
        lo, params = ref[it - 1]
        losses.append(lo)
        assert abs(float(loss.item()) - lo) <= 1e-5 * abs(lo)
        # Adam's m/sqrt(v) is scale-free in the first steps, so rounding differences of the gradients are carried (not
        # damped) from update to update -- in the fp32 oracle as much as on the device: anchored on the float64 trajectory
        assert_close(layer.get_params(), params, 1e-5, f"parameters after update {it}", f64=hi(it - 1))
    assert losses[2] < losses[0]
