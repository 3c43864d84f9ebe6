"""Round 5: the readout's reverse folded into the update's reverse (athena_mp_duvenaud_readout_update_bwd, ONE launch, dc never in HBM)
against athena_mp_duvenaud_readout_bwd + athena_mp_duvenaud_update_bwd_split at configs[2] sizes, with and without a dz_next.
    python scripts/gpu_duv_ro_fused_ab.py [graphs [F_e [outputs [activation]]]]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import DeviceGraph, _capi, ops, synth

dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 130000
ia, ja, voff, E = synth.molecule_batch(S)
N = ia.size - 1
Fv, mn, mx = 64, 1, 10
Fe = int(sys.argv[2]) if len(sys.argv) > 2 else 8
O = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ACT = sys.argv[4] if len(sys.argv) > 4 else "sigmoid"
Fc = Fv + Fe
rng = np.random.default_rng(0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
g = DeviceGraph(ia, ja, n_edge_cols=E)
x, e = T(rng.random((N, Fv), np.float32)), T(rng.random((E, Fe), np.float32))
W = T(rng.standard_normal(Fv * Fc * 10).astype(np.float32) * 0.1)
R = T(rng.standard_normal(O * Fv).astype(np.float32) * 0.1)
seg = T(voff)
gout = T(rng.standard_normal((S, O)).astype(np.float32))
dzn = T(rng.standard_normal((N, Fv)).astype(np.float32))
a_ = ops.duvenaud_propagate(g, x, e)
z, p = ops.duvenaud_update_act_readout(g, a_, W, mn, mx, Fv, R, O, act=ACT)


def timeit(f, reps=20):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        s, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e_.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e_))
    return round(float(np.median(ts)), 4)


def rel(a, b):
    b = b.double(); return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))


out = {"graphs": S, "vertices": int(N), "F_e": Fe, "outputs": O, "activation": ACT}
for name, dz in (("no_dz_next", None), ("dz_next", dzn)):
    def two():
        dc, dR = ops.duvenaud_readout_bwd(R, z, p, seg, gout, act=ACT, dz_next=dz)
        return ops.duvenaud_update_bwd_split(g, dc, a_, W, mn, mx, Fv) + (dR,)
    one = lambda: ops.duvenaud_readout_update_bwd(g, R, z, p, seg, gout, a_, W, mn, mx, Fv, act=ACT, dz_next=dz)
    r2, r1 = two(), one()
    torch.cuda.synchronize()
    chk = {k: rel(u, v) for k, u, v in zip(("da_x", "da_e", "dW", "dR"), r1, r2)}
    rounds = [(timeit(two), timeit(one)) for _ in range(3)]
    out[name] = {"one_launch_vs_two_rel": chk, "two_launches_ms": float(np.median([r[0] for r in rounds])),
                 "one_launch_ms": float(np.median([r[1] for r in rounds])), "rounds": rounds}
print(json.dumps(out))
