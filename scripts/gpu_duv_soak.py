"""Soak of round 5's Duvenaud kernels on random batches: the banded LDS gather against the oracle's CSR-order sums (forward and reverse
pull), split a against packed, the one-call reverse against the two launches (da bits, dW / dR to 1e-5), accumulate_da_e.
    python scripts/gpu_duv_soak.py [draws]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import DeviceGraph, ops, synth
from oracle import oracle

dev = torch.device("cuda:0")
draws = int(sys.argv[1]) if len(sys.argv) > 1 else 60
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bad = 0
for it in range(draws):
    rng = np.random.default_rng(31000 + it)
    S = int(rng.choice([1, 3, 40, 200, 700, 3000]))
    ia, ja, voff, E = synth.molecule_batch(S, seed=int(rng.integers(1, 10 ** 6)))
    N = ia.size - 1
    Fv = 64
    Fe = int(rng.choice([4, 8, 12, 16, 32]))
    O = int(rng.integers(1, 17))
    mn = int(rng.integers(1, 3)); mx = mn + int(rng.integers(0, 9))
    act = str(rng.choice(["sigmoid", "tanh", "relu", "none"]))
    g = DeviceGraph(ia, ja, n_edge_cols=E)
    x = rng.uniform(-1, 1, (N, Fv)).astype(np.float32); e = rng.uniform(-1, 1, (E, Fe)).astype(np.float32)
    xd, ed = T(x), T(e)
    a = ops.duvenaud_propagate(g, xd, ed)
    a_x, a_e = ops.neighbour_sum(g, xd), ops.duvenaud_propagate_edges(g, ed)
    ok = torch.equal(a_x, a[:, :Fv]) and torch.equal(a_e, a[:, Fv:])
    ok = ok and np.array_equal(a.cpu().numpy(), oracle.duvenaud_propagate(x, e, ia, ja))
    W = T((0.3 * rng.standard_normal(Fv * (Fv + Fe) * (mx - mn + 1))).astype(np.float32))
    R = T((0.4 * rng.standard_normal(O * Fv)).astype(np.float32))
    gout, seg = T(rng.standard_normal((voff.size - 1, O)).astype(np.float32)), T(voff)
    dzn = T(rng.standard_normal((N, Fv)).astype(np.float32)) if it % 2 else None
    z, p = ops.duvenaud_update_act_readout(g, a, W, mn, mx, Fv, R, O, act=act)
    z2, p2 = ops.duvenaud_update_act_readout_split(g, a_x, a_e, W, mn, mx, Fv, R, O, act=act)
    ok = ok and torch.equal(z, z2) and torch.equal(p, p2)
    dc, dR2 = ops.duvenaud_readout_bwd(R, z, p, seg, gout, act=act, dz_next=dzn)
    da_x2, da_e2, dW2 = ops.duvenaud_update_bwd_split(g, dc, a, W, mn, mx, Fv)
    base = T(rng.uniform(-1, 1, (N, Fe)).astype(np.float32))
    da_x, da_e, dW, dR = ops.duvenaud_readout_update_bwd(g, R, z, p, seg, gout, a_x, W, mn, mx, Fv, act=act, dz_next=dzn, a_e=a_e, da_e=base.clone())
    rel = lambda u, v: float((u.double() - v.double()).abs().max() / v.double().abs().max().clamp_min(1e-30))
    ok = ok and torch.equal(da_x, da_x2) and torch.equal(da_e, base + da_e2) and rel(dW, dW2) <= 1e-5 and rel(dR, dR2) <= 1e-5
    dx = ops.duvenaud_propagate_bwd_x(g, da_x, Fv)
    ok = ok and np.array_equal(dx.cpu().numpy(), oracle.duvenaud_propagate_bwd_x(da_x.cpu().numpy(), Fv, ia, ja))
    if not ok:
        bad += 1
        print("MISMATCH draw", it, dict(S=S, N=N, Fe=Fe, O=O, mn=mn, mx=mx, act=act, dz=dzn is not None))
print(f"{draws} draws, {bad} mismatches")
sys.exit(1 if bad else 0)
