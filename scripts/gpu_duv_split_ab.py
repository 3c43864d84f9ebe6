"""A/B at BASELINE configs[2] sizes: da of the Duvenaud update's reverse pass written SPLIT (athena_mp_duvenaud_update_bwd_split:
da_x [N, 64], 256-byte rows = whole cache lines whatever the order, + da_e [N, 8]) against the packed [N, 72] rows of 288
bytes the bucket-ordered kernel writes as partial lines.  Three launches each way: update reverse (da + dW), propagate reverse
to x, propagate reverse to e.  profiles/r05_c3_split_da_ab.txt holds the run that decided it (0.936 -> 0.872 ms).
    python scripts/gpu_duv_split_ab.py packed ; python scripts/gpu_duv_split_ab.py split"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import DeviceGraph, _capi, ops, synth

mode = sys.argv[1] if len(sys.argv) > 1 else "packed"
dev = torch.device("cuda:0")
S = 130000
ia, ja, voff, E = synth.molecule_batch(S)
N = ia.size - 1
Fv, Fe, mn, mx = 64, 8, 1, 10
Fc = Fv + Fe
rng = np.random.default_rng(0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
g = DeviceGraph(ia, ja, n_edge_cols=E)
a_ = T(rng.random((N, Fc), np.float32)); dc = T(rng.standard_normal((N, Fv)).astype(np.float32))
W = T(rng.standard_normal(Fv * Fc * 10).astype(np.float32) * 0.1)
_capi.use_torch_stream()


def timeit(f, reps=20):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return float(np.median(ts))


dx, de = torch.empty((N, Fv), device=dev), torch.empty((E, Fe), device=dev)
if mode == "split":
    upd = lambda: ops.duvenaud_update_bwd_split(g, dc, a_, W, mn, mx, Fv)
    da_x, da_e, dW = upd()
    fx = lambda: _capi.call("athena_mp_duvenaud_propagate_bwd_x", g.handle, Fv, 0, da_x.data_ptr(), dx.data_ptr())
    fe = lambda: _capi.call("athena_mp_duvenaud_propagate_bwd_e", g.handle, 0, Fe, da_e.data_ptr(), de.data_ptr())
else:
    upd = lambda: ops.duvenaud_update_bwd(g, dc, a_, W, mn, mx)
    da, dW = upd()
    fx = lambda: _capi.call("athena_mp_duvenaud_propagate_bwd_x", g.handle, Fv, Fe, da.data_ptr(), dx.data_ptr())
    fe = lambda: _capi.call("athena_mp_duvenaud_propagate_bwd_e", g.handle, Fv, Fe, da.data_ptr(), de.data_ptr())
torch.cuda.synchronize()
fx(); fe(); torch.cuda.synchronize()
out = {"mode": mode, "checks": {"dx_sum": float(dx.double().sum()), "de_sum": float(de.double().sum()), "dW_sum": float(dW.double().sum()),
                                 "dx_abs": float(dx.double().abs().sum()), "de_abs": float(de.double().abs().sum())}}
rounds = []
for _ in range(3):
    rounds.append({"update_bwd_fused_ms": round(timeit(upd), 4),
                   "propagate_bwd_x_ms": round(timeit(fx), 4), "propagate_bwd_e_ms": round(timeit(fe), 4)})
out["rounds"] = rounds
out["median"] = {k: float(np.median([r[k] for r in rounds])) for k in rounds[0]}
out["median"]["sum_ms"] = round(sum(out["median"].values()), 4)
print(json.dumps(out))
