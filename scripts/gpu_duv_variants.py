"""timing of the Duvenaud update family at configs[2] size for one library build (ATHENA_MP_LIB selects a variant)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import DeviceGraph, ops, synth, _capi
_capi.init(0)
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
ia, ja, voff, E = synth.molecule_batch(130000)
N = ia.size - 1
Fv, Fe, mn, mx = 64, 8, 1, 10
g = DeviceGraph(ia, ja, n_edge_cols=E)
a_ = T(rng.random((N, Fv + Fe), np.float32)); dc = T(rng.standard_normal((N, Fv)).astype(np.float32))
W = T(rng.standard_normal(Fv * (Fv + Fe) * 10).astype(np.float32) * 0.1)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return round(s.elapsed_time(e) / n, 4)
out = {"lib": os.environ.get("ATHENA_MP_LIB", "stock")}
out["update_sigmoid"] = timeit(lambda: ops.duvenaud_update_act(g, a_, W, mn, mx, Fv, act="sigmoid"))
out["update_none"] = timeit(lambda: ops.duvenaud_update_act(g, a_, W, mn, mx, Fv, act="none"))
out["update_bwd_a"] = timeit(lambda: ops.duvenaud_update_bwd_a(g, dc, W, mn, mx, Fv + Fe))
out["update_bwd_w"] = timeit(lambda: ops.duvenaud_update_bwd_w(g, dc, a_, mn, mx))
R = T(rng.standard_normal(10 * Fv).astype(np.float32) * 0.1)
if hasattr(_capi.load(), "athena_mp_duvenaud_update_readout_fwd"):
    out["update_sigmoid_readout"] = timeit(lambda: ops.duvenaud_update_act_readout(g, a_, W, mn, mx, Fv, R, 10, act="sigmoid"))
if hasattr(ops, "duvenaud_update_bwd") and hasattr(_capi.load(), "athena_mp_duvenaud_update_bwd"):
    out["update_bwd_fused"] = timeit(lambda: ops.duvenaud_update_bwd(g, dc, a_, W, mn, mx))
print(json.dumps(out))
