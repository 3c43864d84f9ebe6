"""Round 5: a = [a_x | a_e] kept SPLIT for a whole Duvenaud layer (the edge part gathered once per forward pass) against the packed
form, launch by launch at configs[2] sizes: propagate, update + sigmoid + readout p, the one-call reverse.
    python scripts/gpu_duv_split_a_ab.py [graphs]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import DeviceGraph, ops, synth

dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 130000
ia, ja, voff, E = synth.molecule_batch(S)
N = ia.size - 1
Fv, Fe, mn, mx, O = 64, 8, 1, 10, 10
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
a = ops.duvenaud_propagate(g, x, e)
a_x, a_e = ops.neighbour_sum(g, x), ops.duvenaud_propagate_edges(g, e)
z, p = ops.duvenaud_update_act_readout(g, a, W, mn, mx, Fv, R, O, act="sigmoid")


def timeit(f, reps=20):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        s, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e_.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e_))
    return float(np.median(ts))


rows = {"propagate": (lambda: ops.duvenaud_propagate(g, x, e, out=a), lambda: ops.neighbour_sum(g, x, out=a_x)),
        "edge part alone (once per layer)": (None, lambda: ops.duvenaud_propagate_edges(g, e, out=a_e)),
        "update + sigmoid + readout p": (lambda: ops.duvenaud_update_act_readout(g, a, W, mn, mx, Fv, R, O, act="sigmoid"),
                                         lambda: ops.duvenaud_update_act_readout_split(g, a_x, a_e, W, mn, mx, Fv, R, O, act="sigmoid")),
        "one-call reverse (+dz_next)": (lambda: ops.duvenaud_readout_update_bwd(g, R, z, p, seg, gout, a, W, mn, mx, Fv, act="sigmoid", dz_next=dzn),
                                        lambda: ops.duvenaud_readout_update_bwd(g, R, z, p, seg, gout, a_x, W, mn, mx, Fv, act="sigmoid", dz_next=dzn, a_e=a_e))}
out = {"graphs": S, "vertices": int(N), "ms": {}}
for k, (fp, fs) in rows.items():
    r = [(timeit(fp) if fp else None, timeit(fs)) for _ in range(3)]
    out["ms"][k] = {"packed": (round(float(np.median([v[0] for v in r])), 4) if fp else None), "split": round(float(np.median([v[1] for v in r])), 4)}
print(json.dumps(out))

# ---- the T = 4 layer step, split against packed, alternating on this box
from athena_amd.graph import graph_type
from athena_amd.layers import duvenaud_msgpass_layer_type

del a, a_x, a_e, z, p
layer = duvenaud_msgpass_layer_type(num_vertex_features=[Fv], num_edge_features=[Fe], num_time_steps=4, max_vertex_degree=mx,
                                    num_outputs=O, min_vertex_degree=mn, seed=1)
layer.set_graph_batched(graph_type.from_csr(ia, ja, num_edges=E).freeze(), voff)
up = T(rng.standard_normal((S, O)).astype(np.float32))


def step():
    layer.forward(x, e)
    return layer.backward(up, need_input_grad=True, need_edge_grad=True)


res = {"split": [], "packed": []}
split_fn = type(layer)._split_a
for rnd in range(3):
    for mode in ("split", "packed"):
        type(layer)._split_a = split_fn if mode == "split" else (lambda self: False)
        res[mode].append(round(timeit(step, 10), 4))
type(layer)._split_a = split_fn
print(json.dumps({"layer_T4_ms": res, "median": {k: float(np.median(v)) for k, v in res.items()}}))
