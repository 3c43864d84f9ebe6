"""latency of the host-pointer (phase-1 Fortran) entry points on a mini-batch sized graph"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from athena_amd import DeviceGraph, synth, _capi
_capi.init(0)
ia, ja, voff, E = synth.molecule_batch(32)
N = ia.size - 1
g = DeviceGraph(ia, ja, n_edge_cols=E)
x = np.random.default_rng(0).random((N, 64), np.float32); y = np.empty_like(x)
P = lambda a: a.ctypes.data_as(C.c_void_p)
for _ in range(20): _capi.call("athena_mp_kipf_propagate_fwd_host", g.handle, 64, P(x), P(y))
t = time.perf_counter(); reps = 500
for _ in range(reps): _capi.call("athena_mp_kipf_propagate_fwd_host", g.handle, 64, P(x), P(y))
print("kipf_propagate_fwd_host, %d vertices x 64: %.1f us per call" % (N, (time.perf_counter() - t) / reps * 1e6))
w = np.random.default_rng(1).standard_normal(64 * 64).astype(np.float32); z = np.empty_like(x)
for _ in range(20): _capi.call("athena_mp_gemm_dx_host", N, 64, 64, P(x), P(w), P(z))
t = time.perf_counter()
for _ in range(reps): _capi.call("athena_mp_gemm_dx_host", N, 64, 64, P(x), P(w), P(z))
print("gemm_dx_host (pooled staging), %d x 64 x 64: %.1f us per call" % (N, (time.perf_counter() - t) / reps * 1e6))
