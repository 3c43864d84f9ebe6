"""Measured-first probe (round 5): do the matrix-bound launches of a Duvenaud time step (update, fused reverse: fp32 MFMA at the
clock the chip holds under it) and its gather launches (propagate, its reverse: latency / HBM) OVERLAP when they run on two HIP
streams over independent data (two halves of a batch of independent graphs)?  Times A alone, B alone, A || B.
    python scripts/gpu_c3_overlap.py            (ATHENA_MP_LIB selects a variant build, e.g. one matrix workgroup per CU)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import DeviceGraph, _capi, ops, synth

dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 130000
ia, ja, voff, E = synth.molecule_batch(S)
N = ia.size - 1
Fv, Fe, mn, mx, O = 64, 8, 1, 10, 10
Fc = Fv + Fe
rng = np.random.default_rng(0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
g = DeviceGraph(ia, ja, n_edge_cols=E)
a_ = T(rng.random((N, Fc), np.float32)); dc = T(rng.standard_normal((N, Fv)).astype(np.float32))
W = T(rng.standard_normal(Fv * Fc * 10).astype(np.float32) * 0.1)
R = T(rng.standard_normal(O * Fv).astype(np.float32) * 0.1)
x, e = T(rng.random((N, Fv), np.float32)), T(rng.random((E, Fe), np.float32))
a_out = torch.empty((N, Fc), device=dev)
dax = T(rng.standard_normal((N, Fv)).astype(np.float32))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def on(stream, f):
    with torch.cuda.stream(stream):            # ops.* enqueue on torch's current stream
        _capi.use_torch_stream()
        f()


A = {"update_bwd": lambda: ops.duvenaud_update_bwd_split(g, dc, a_, W, mn, mx, Fv),
     "update_fwd_readout": lambda: ops.duvenaud_update_act_readout(g, a_, W, mn, mx, Fv, R, O, act="sigmoid")}
B = {"propagate": lambda: ops.duvenaud_propagate(g, x, e, out=a_out),
     "propagate_bwd_x": lambda: _capi.call("athena_mp_duvenaud_propagate_bwd_x", g.handle, Fv, 0, dax.data_ptr(), x_out.data_ptr())}
x_out = torch.empty((N, Fv), device=dev)


def timed(fa, fb, reps=20):
    ts = []
    for _ in range(reps + 2):
        torch.cuda.synchronize()
        st, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        st.record(s1)
        s2.wait_event(st)
        if fa: on(s1, fa)
        if fb: on(s2, fb)
        e1.record(s1); e2.record(s2)
        torch.cuda.synchronize()
        ts.append(max(st.elapsed_time(e1), st.elapsed_time(e2)))
    return round(float(np.median(ts[2:])), 4)


out = {"lib": os.environ.get("ATHENA_MP_LIB", "stock"), "graphs": S}
for na, fa in A.items():
    for nb, fb in B.items():
        ta, tb, tab = timed(fa, None), timed(None, fb), timed(fa, fb)
        out[f"{na} || {nb}"] = {"A_ms": ta, "B_ms": tb, "both_ms": tab, "sum_ms": round(ta + tb, 4), "max_ms": max(ta, tb)}
_capi.use_torch_stream()
print(json.dumps(out))
