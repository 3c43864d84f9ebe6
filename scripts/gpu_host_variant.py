"""dev aid: PCIe-inclusive rate of the *_host staging entry points at C2 (never the bench value)"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from athena_amd import DeviceGraph, synth, _capi
N, F = 1000000, 128
ia, ja = synth.random_graph_csr(N, 4500000)
x, w, dz = synth.kipf_inputs(N, F)
g = DeviceGraph(ia, ja, n_edge_cols=0)
y = np.empty_like(x)
P = lambda a: a.ctypes.data_as(C.c_void_p)
for _ in range(2): _capi.call("athena_mp_kipf_propagate_fwd_host", g.handle, F, P(x), P(y))
t0 = time.perf_counter()
for _ in range(5): _capi.call("athena_mp_kipf_propagate_fwd_host", g.handle, F, P(x), P(y))
dt = (time.perf_counter() - t0) / 5
print(f"kipf_propagate_fwd_host: {dt*1e3:.1f} ms per call = {ja.shape[1]/dt/1e6:.1f} M edges/s (pageable host memory, 512 MB each way)")
