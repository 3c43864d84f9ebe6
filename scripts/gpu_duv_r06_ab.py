"""A/B of round 6 at BASELINE configs[2] sizes (130 000 graphs, F_v 64 / F_e 8, 10 outputs, degrees 1..10): the two matrix launches of a
Duvenaud time step as the layer runs them -- update + sigmoid + readout p with `a` split, and the one-launch reverse (with and
without a dz_next) -- timed with events on the launch stream, medians of 3 x 20, with checksums so that variants can be compared for
bits.  ATHENA_MP_LIB picks the build (scripts/build_variants.sh duv_mfma.hip plain_tail -DDUV_PLAIN_TAIL=1: the plain k-mapping of the
forward's tail fragment; ... ro_nw8 -DDUV_RO_NW=8: the one-launch reverse as 8 waves per workgroup)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import DeviceGraph, _capi, ops, synth

dev = torch.device("cuda:0")
S = 130000
ia, ja, voff, E = synth.molecule_batch(S)
N = ia.size - 1
Fv, Fe, O, mn, mx = 64, 8, 10, 1, 10
rng = np.random.default_rng(0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
g = DeviceGraph(ia, ja, n_edge_cols=E)
a_x, a_e = T(rng.random((N, Fv), np.float32)), T(rng.random((N, Fe), np.float32))
W = T((rng.standard_normal(Fv * (Fv + Fe) * 10) * 0.1).astype(np.float32))
R = T((rng.standard_normal(O * Fv) * 0.3).astype(np.float32))
gout = T(rng.standard_normal((S, O)).astype(np.float32))
dzn = T(rng.standard_normal((N, Fv)).astype(np.float32))
seg = T(voff)
_capi.use_torch_stream()


def timeit(f, reps=20):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return float(np.median(ts))


fwd = lambda: ops.duvenaud_update_act_readout_split(g, a_x, a_e, W, mn, mx, Fv, R, O, act="sigmoid")
z, p = fwd()
bwd0 = lambda: ops.duvenaud_readout_update_bwd(g, R, z, p, seg, gout, a_x, W, mn, mx, Fv, act="sigmoid", a_e=a_e)
bwd1 = lambda: ops.duvenaud_readout_update_bwd(g, R, z, p, seg, gout, a_x, W, mn, mx, Fv, act="sigmoid", dz_next=dzn, a_e=a_e)
da_x, da_e, dW, dR = bwd1()
cs = lambda t: [float(t.double().sum()), float(t.double().abs().sum())]
out = {"lib": os.path.basename(os.environ.get("ATHENA_MP_LIB", "product")),
       "checks": {"z": cs(z), "p": cs(p), "da_x": cs(da_x), "da_e": cs(da_e), "dW": cs(dW), "dR": cs(dR)}}
rounds = [{"fwd_split_ms": round(timeit(fwd), 4), "bwd_one_launch_ms": round(timeit(bwd0), 4), "bwd_one_launch_dz_ms": round(timeit(bwd1), 4)}
          for _ in range(3)]
out["rounds"] = rounds
out["median"] = {k: float(np.median([r[k] for r in rounds])) for k in rounds[0]}
print(json.dumps(out))
