"""The *_pair_host entry points a shim inside athena binds in diffstruc's two `pure` partial callbacks
(athena_amd/fortran/athena_dropin/athena_hip_msgpass_ops.f90): both partials of a node from ONE device
pass, handed over below the C ABI -- and never handed to a request they were not computed for."""
import ctypes as C

import numpy as np
import pytest

from helpers import assert_close, csr_from_index_list

pytestmark = pytest.mark.gpu


def P_(a):
    return a.ctypes.data_as(C.c_void_p)


def _stats():
    from athena_amd import _capi

    f, h = C.c_int64(0), C.c_int64(0)
    _capi.call("athena_mp_pair_stats", C.byref(f), C.byref(h))
    return f.value, h.value


@pytest.mark.parametrize("Fi,Fo,act", [(10, 4, 0), (72, 64, 2), (72, 64, 1), (24, 16, 3)])
@pytest.mark.parametrize("resident", [0, 1])
def test_duvenaud_update_pair(dev, oracle, Fi, Fo, act, resident):
    from athena_amd import DeviceGraph, _capi
    from athena_amd import synth

    rng = np.random.default_rng(Fi * 100 + Fo + act)
    ia, ja, seg, n_edges = synth.molecule_batch(60, seed=5)
    N = ia.size - 1
    dg = DeviceGraph(ia, ja, n_edge_cols=n_edges)
    mn, mx = 1, 6
    D = mx - mn + 1
    a = rng.uniform(-1, 1, (N, Fi)).astype(np.float32)
    w = (0.3 * rng.standard_normal(Fo * Fi * D)).astype(np.float32)
    g = rng.uniform(-1, 1, (N, Fo)).astype(np.float32)
    names = {0: "none", 1: "relu", 2: "sigmoid", 3: "tanh"}
    z = oracle.activation(names[act], oracle.duvenaud_update(a, w, ia, mn, mx, Fo)) if act else np.zeros((N, Fo), np.float32)
    dc = oracle.activation_bwd(names[act], z, g) if act else g
    da_ref = oracle.duvenaud_update_bwd_a(dc, w, ia, mn, mx, Fi)
    dw_ref = oracle.duvenaud_update_bwd_w(dc, a, ia, mn, mx)
    _capi.call("athena_mp_resident_mode", resident)
    try:
        f0, h0 = _stats()
        da = np.empty((N, Fi), np.float32)
        dw = np.empty_like(w)
        args = (dg.handle, Fi, Fo, mn, mx, act, P_(z), P_(g), P_(a), P_(w))
        _capi.call("athena_mp_duvenaud_update_bwd_pair_host", *args, 0, P_(da))      # first request: the fused pass
        _capi.call("athena_mp_duvenaud_update_bwd_pair_host", *args, 1, P_(dw))      # second: handed over
        f1, h1 = _stats()
        assert (f1 - f0, h1 - h0) == (1, 1)
        if resident:
            _capi.call("athena_mp_resident_flush", None)
        assert_close(da, da_ref, 1e-5, "da")
        assert_close(dw, dw_ref, 1e-5, "dw")
        # the other order, and a slot is handed over ONCE
        dw2, da2, da3 = np.empty_like(w), np.empty((N, Fi), np.float32), np.empty((N, Fi), np.float32)
        _capi.call("athena_mp_duvenaud_update_bwd_pair_host", *args, 1, P_(dw2))
        _capi.call("athena_mp_duvenaud_update_bwd_pair_host", *args, 0, P_(da2))
        _capi.call("athena_mp_duvenaud_update_bwd_pair_host", *args, 0, P_(da3))     # nothing parked any more: a fresh pass
        f2, h2 = _stats()
        assert (f2 - f1, h2 - h1) == (2, 1)
        if resident:
            _capi.call("athena_mp_resident_flush", None)
        assert np.array_equal(dw2, dw) and np.array_equal(da2, da) and np.array_equal(da3, da)
        # a request with OTHER content in the same arrays must not get the parked partial
        _capi.call("athena_mp_duvenaud_update_bwd_pair_host", *args, 0, P_(da3))     # parks dw for (g, a, w)
        g *= 2.0                                                                      # same address, new content
        dw3 = np.empty_like(w)
        _capi.call("athena_mp_duvenaud_update_bwd_pair_host", *args, 1, P_(dw3))
        f3, h3 = _stats()
        assert (f3 - f2, h3 - h2) == (2, 0), "a parked partial was handed to a request with different operand content"
        if resident:
            _capi.call("athena_mp_resident_flush", None)
        dc2 = oracle.activation_bwd(names[act], z, g) if act else g
        assert_close(dw3, oracle.duvenaud_update_bwd_w(dc2, a, ia, mn, mx), 1e-5, "dw after the edit")
    finally:
        _capi.call("athena_mp_resident_mode", 0)


def test_duvenaud_update_readout_fwd_host(dev, oracle):
    from athena_amd import DeviceGraph, _capi, synth

    rng = np.random.default_rng(3)
    ia, ja, seg, n_edges = synth.molecule_batch(40, seed=9)
    N = ia.size - 1
    dg = DeviceGraph(ia, ja, n_edge_cols=n_edges)
    Fi, Fo, O, mn, mx = 72, 64, 10, 1, 6
    a = rng.uniform(0, 1, (N, Fi)).astype(np.float32)
    w = (0.3 * rng.standard_normal(Fo * Fi * (mx - mn + 1))).astype(np.float32)
    R = (0.3 * rng.standard_normal(O * Fo)).astype(np.float32)
    z, p = np.empty((N, Fo), np.float32), np.empty((N, O), np.float32)
    _capi.call("athena_mp_duvenaud_update_readout_fwd_host", dg.handle, Fi, Fo, mn, mx, P_(a), P_(w), 2, P_(z), O, P_(R), P_(p))
    z_ref = oracle.activation("sigmoid", oracle.duvenaud_update(a, w, ia, mn, mx, Fo))
    assert_close(z, z_ref, 1e-5, "z")
    assert_close(p, oracle.softmax_cols(oracle.matmul(R, z_ref, O)), 1e-5, "p")


def test_gno_aggregate_pair(dev, oracle):
    from athena_amd import DeviceGraph, _capi
    from oracle import oracle64 as o64
    from test_gpu_ops import _gno_case

    for (N, d, H, Fi, Fo, extra) in [(50, 3, 7, 5, 9, 60), (300, 3, 64, 64, 64, 900)]:
        g, E, coords, x, theta, up = _gno_case(N + H, N, d, H, Fi, Fo, extra)
        ia, ja = g.adj_ia, g.adj_ja
        dg = DeviceGraph(ia, ja, n_edge_cols=E)
        args = (dg.handle, d, H, Fi, Fo, P_(theta), P_(coords), P_(x), P_(up))
        dx, dth, dco = np.empty((N, Fi), np.float32), np.empty_like(theta), np.empty((E, d), np.float32)
        f0, h0 = _stats()
        _capi.call("athena_mp_gno_aggregate_bwd_pair_host", *args, 1, P_(dth))       # theta first (either order works)
        _capi.call("athena_mp_gno_aggregate_bwd_pair_host", *args, 0, P_(dx))
        f1, h1 = _stats()
        assert (f1 - f0, h1 - h0) == (1, 1)
        kap = oracle.gno_kernel_eval(coords, theta, H, Fi * Fo)
        assert_close(dx, oracle.gno_aggregate_bwd_x(up, kap, ia, ja, Fi), 1e-5, "dx",
                     f64=lambda: o64.gno_aggregate_bwd_x(up, o64.gno_kernel_eval(coords, theta, H, Fi * Fo), ia, ja, Fi))
        dk = oracle.gno_aggregate_bwd_k(up, x, E, ia, ja)
        assert_close(dth, oracle.gno_kernel_bwd_theta(coords, theta, dk, H), 1e-5, "dtheta",
                     f64=lambda: o64.gno_kernel_bwd_theta(coords, theta, o64.gno_aggregate_bwd_k(up, x, E, ia, ja), H))
        _capi.call("athena_mp_gno_aggregate_bwd_pair_host", *args, 2, P_(dco))       # coordinates: computed when asked for
        assert_close(dco, oracle.gno_kernel_bwd_coords(coords, theta, dk, H), 1e-5, "dcoords",
                     f64=lambda: o64.gno_kernel_bwd_coords(coords, theta, o64.gno_aggregate_bwd_k(up, x, E, ia, ja), H))
        # ... and it parked dx and dtheta of the same operands
        dx2, dth2 = np.empty_like(dx), np.empty_like(dth)
        _capi.call("athena_mp_gno_aggregate_bwd_pair_host", *args, 0, P_(dx2))
        _capi.call("athena_mp_gno_aggregate_bwd_pair_host", *args, 1, P_(dth2))
        f2, h2 = _stats()
        assert (f2 - f1, h2 - h1) == (1, 2)
        assert_close(dx2, dx, 1e-6); assert_close(dth2, dth, 1e-6)
        # another graph handle with the same shapes and the same operand arrays is a different request
        dg2 = DeviceGraph(ia, ja, n_edge_cols=E)
        _capi.call("athena_mp_gno_aggregate_bwd_pair_host", *args, 0, P_(dx2))       # parks dtheta for dg
        _capi.call("athena_mp_gno_aggregate_bwd_pair_host", dg2.handle, *args[1:], 1, P_(dth2))
        f3, h3 = _stats()
        assert (f3 - f2, h3 - h2) == (2, 0)
