"""dev aid: PCIe-inclusive cost of the *_host entry points at configs[1] size (never the bench value) -- one op, and a
3-step Kipf chain (kipf_propagate + matmul per step, athena_kipf_msgpass_layer.f90:943-952) staged op by op against the
same chain with the residency table on (athena_mp_resident_mode: X up once, Z down once)"""
import os, sys, time, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from athena_amd import DeviceGraph, synth, _capi
N, F = 1000000, 128
ia, ja = synth.random_graph_csr(N, 4500000)
x, w, dz = synth.kipf_inputs(N, F)
g = DeviceGraph(ia, ja, n_edge_cols=0)
P = lambda a: a.ctypes.data_as(C.c_void_p)
bufs = [np.empty_like(x) for _ in range(6)]
ws = [w, (w * 0.5).astype(np.float32), (w * 0.25).astype(np.float32)]

def chain():
    src = x
    for k in range(3):
        _capi.call("athena_mp_kipf_propagate_fwd_host", g.handle, F, P(src), P(bufs[2 * k]))
        _capi.call("athena_mp_gemm_fwd_host", N, F, F, P(bufs[2 * k]), P(ws[k]), None, 0, P(bufs[2 * k + 1]))
        src = bufs[2 * k + 1]
    return src

def timed(fn, n=4):
    fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e3

out = {}
out["one_op_kipf_propagate_fwd_host_ms"] = timed(lambda: _capi.call("athena_mp_kipf_propagate_fwd_host", g.handle, F, P(x), P(bufs[0])))
out["chain3_staged_ms"] = timed(chain)
ref = chain().copy()
_capi.call("athena_mp_resident_mode", 1)
def chain_resident():
    z = chain()
    _capi.call("athena_mp_resident_flush", P(z))
    x[0, 0] = x[0, 0]          # host code does not touch X between steps; it is re-uploaded anyway (first touch rule)
out["chain3_resident_ms"] = timed(chain_resident)
st = [C.c_int64() for _ in range(5)]
_capi.call("athena_mp_resident_stats", *[C.byref(s) for s in st])
out["resident_stats"] = dict(zip(("arrays", "h2d_bytes", "d2h_bytes", "reused_inputs", "lazy_outputs"), [s.value for s in st]))
out["same_bits"] = bool(np.array_equal(bufs[5], ref))
_capi.call("athena_mp_resident_mode", 0)
import torch
from athena_amd import ops
xd, wd = torch.from_numpy(x).cuda(), [torch.from_numpy(t).cuda() for t in ws]
def chain_dev():
    s = xd
    for k in range(3):
        _, s = ops.kipf_layer_fwd(g, s, wd[k], F)
    torch.cuda.synchronize()
out["chain3_device_resident_fused_ms"] = timed(chain_dev)
print(json.dumps(out))
