"""A/B of the EXPERIMENT duv_gather_update_kernel (variants/libathena_mp_gather.so, built with -DDUV_GATHER=1 from
scripts/experiments/duv_gather_update.patch): duvenaud_propagate folded into the update launch at BASELINE configs[2] --
against the two shipped launches on the same box, same tensors.

    ATHENA_MP_LIB=$PWD/variants/libathena_mp_gather.so python scripts/gpu_duv_gather_ab.py [graphs]
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from athena_amd import DeviceGraph, _capi, ops, synth

S = int(sys.argv[1]) if len(sys.argv) > 1 else 130000
dev = torch.device("cuda:0")
ia, ja, voff, E = synth.molecule_batch(S)
N, nnz = ia.size - 1, ja.shape[1]
Fv, Fe, O, mn, mx = 64, 8, 10, 1, 10
rng = np.random.default_rng(0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
g = DeviceGraph(ia, ja, n_edge_cols=E)
x = T(rng.random((N, Fv), np.float32))
e_ext = torch.zeros((E + 1, Fe), device=dev)
e_ext[:E] = T(rng.random((E, Fe), np.float32))
e = e_ext[:E]
W = T(rng.standard_normal(Fv * (Fv + Fe) * 10).astype(np.float32) * 0.1)
R = T(rng.standard_normal(O * Fv).astype(np.float32) * 0.1)
lib = _capi.load()
fn = lib.athena_mp_duvenaud_propagate_update_fwd
fn.restype = C.c_int
vp = C.c_void_p
fn.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp, C.c_int32, vp, vp, vp, C.c_int32, vp]
_capi.use_torch_stream()


def fused(z, a_out=None, p=None):
    rc = fn(g.handle, mn, mx, x.data_ptr(), e_ext.data_ptr(), W.data_ptr(), 2, z.data_ptr(), a_out.data_ptr() if a_out is not None else None,
            R.data_ptr() if p is not None else None, O, p.data_ptr() if p is not None else None)
    if rc:
        raise RuntimeError(lib.athena_mp_last_error().decode())


def timeit(f, reps=20):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


a_ref = ops.duvenaud_propagate(g, x, e)
z_ref, p_ref = ops.duvenaud_update_act_readout(g, a_ref, W, mn, mx, Fv, R, O, act="sigmoid")
z_plain = ops.duvenaud_update_act(g, a_ref, W, mn, mx, Fv, act="sigmoid")
z, a_out, p = torch.empty_like(z_ref), torch.empty_like(a_ref), torch.empty_like(p_ref)
fused(z, a_out, p)
torch.cuda.synchronize()
out = {"graphs": S, "vertices": N, "entries": nnz,
       "parity": {"a_bit_exact": bool(torch.equal(a_out, a_ref)), "z_bit_exact_vs_shipped": bool(torch.equal(z, z_ref)),
                  "z_max_abs_diff": float((z - z_ref).abs().max()), "p_max_abs_diff": float((p - p_ref).abs().max())}}
z2 = torch.empty_like(z)
fused(z2)
out["parity"]["z_without_a_and_p_equal"] = bool(torch.equal(z2, z))
rounds = []
for _ in range(3):           # alternating, three times: boxes drift
    r = {"shipped_propagate_ms": timeit(lambda: ops.duvenaud_propagate(g, x, e, out=a_ref)),
         "shipped_update_sigmoid_readout_ms": timeit(lambda: ops.duvenaud_update_act_readout(g, a_ref, W, mn, mx, Fv, R, O, act="sigmoid")),
         "shipped_update_sigmoid_ms": timeit(lambda: ops.duvenaud_update_act(g, a_ref, W, mn, mx, Fv, act="sigmoid")),
         "fused_z_ms": timeit(lambda: fused(z)),
         "fused_z_keep_a_ms": timeit(lambda: fused(z, a_out)),
         "fused_z_p_ms": timeit(lambda: fused(z, None, p)),
         "fused_z_p_keep_a_ms": timeit(lambda: fused(z, a_out, p))}
    rounds.append({k: round(v, 4) for k, v in r.items()})
out["rounds"] = rounds
med = {k: float(np.median([r[k] for r in rounds])) for k in rounds[0]}
out["median"] = {k: round(v, 4) for k, v in med.items()}
out["shipped_pair_with_readout_ms"] = round(med["shipped_propagate_ms"] + med["shipped_update_sigmoid_readout_ms"], 4)
out["fused_with_readout_keep_a_ms"] = med["fused_z_p_keep_a_ms"]
print(json.dumps(out))
