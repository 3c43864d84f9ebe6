"""A/B of round 6: Kipf on a block-diagonal batch (130 000 molecule-shaped graphs, 2.34 M vertices / 7.1 M entries, band 29) at F = 64 and
128 -- kipf_propagate forward and reverse (reference scatter and exact adjoint) and the layer step (propagate + dense step + relu; its
reverse to x).  ATHENA_MP_LIB picks the build: the product (LDS-staged banded gather with the per-entry coefficient), `agg_no_banded`
(scripts/build_variants.sh agg.hip agg_no_banded -DAGG_NO_BANDED=1: the general row-chasing gather) and `layer_fused`
(... fused.hip layer_fused -DKIPF_LAYER_BANDED=0: the layer through the one-launch kernel).  Checksums say whether the bits agree."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import DeviceGraph, _capi, ops, synth

dev = torch.device("cuda:0")
ia, ja, voff, E = synth.molecule_batch(130000)
N = ia.size - 1
g = DeviceGraph(ia, ja)
_capi.use_torch_stream()
rng = np.random.default_rng(0)


def timeit(f, reps=20):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return float(np.median(ts))


cs = lambda t: [float(t.double().sum()), float(t.double().abs().sum())]
out = {"lib": os.path.basename(os.environ.get("ATHENA_MP_LIB", "product")), "vertices": int(N), "entries": int(ja.shape[1])}
for F in (64, 128):
    x = torch.from_numpy(rng.standard_normal((N, F)).astype(np.float32)).to(dev)
    w = torch.from_numpy((rng.standard_normal(F * F) * 0.1).astype(np.float32)).to(dev)
    P, Z = torch.empty((N, F), device=dev), torch.empty((N, F), device=dev)
    fns = {"propagate_fwd": lambda: ops.kipf_propagate(g, x),
           "propagate_bwd_reference": lambda: ops.kipf_propagate_bwd(g, x),
           "propagate_bwd_exact": lambda: ops.kipf_propagate_bwd(g, x, exact=True),
           "layer_fwd_relu": lambda: ops.kipf_layer_fwd(g, x, w, F, act="relu", P=P, Z=Z),
           "layer_bwd_x": lambda: ops.kipf_layer_bwd_x(g, x, w, F)}
    r = {}
    for k, f in fns.items():
        v = f()
        v = v[1] if isinstance(v, tuple) else v
        r[k] = {"ms": round(float(np.median([timeit(f) for _ in range(3)])), 4), "check": cs(v)}
    r["layer_fwd_relu"]["check_P"] = cs(P)
    out[f"F{F}"] = r
print(json.dumps(out))
