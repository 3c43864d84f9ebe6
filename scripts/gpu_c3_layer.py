"""BASELINE configs[2] as the config states it -- the T = 4 Duvenaud LAYER (F_v 64, F_e 8, degrees 1..10, 10 outputs) -- driven
from FORTRAN (duvenaud_mp_layer_type: set_graph on the batch of 130 000 graph objects, forward_dev / backward_dev on tensors
resident in HBM; athena_amd/fortran/athena_mp_layer_run.f90, case kind 6) beside the Python mirror on the same batch and
parameters.  One json object on stdout (kept as profiles/r05_c3_layer_fortran.json).

    python scripts/gpu_c3_layer.py [graphs] [reps]
"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from athena_amd import synth
from athena_amd.graph import graph_type
from athena_amd.layers import duvenaud_msgpass_layer_type

RUNNER = os.path.join(ROOT, "athena_amd", "fortran", "athena_mp_layer_run")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 130000
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
Fv, Fe, O, T, mn, mx = 64, 8, 10, 4, 1, 10


def rel(a, b):
    b = np.asarray(b, np.float64)
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


ia, ja, voff, E = synth.molecule_batch(S)
N, nnz = ia.size - 1, ja.shape[1]
dev = torch.device("cuda:0")
layer = duvenaud_msgpass_layer_type(num_vertex_features=[Fv], num_edge_features=[Fe], num_time_steps=T, max_vertex_degree=mx,
                                    num_outputs=O, min_vertex_degree=mn, seed=1)
layer.set_graph_batched(graph_type.from_csr(ia, ja, num_edges=E).freeze(), voff)
rng = np.random.default_rng(0)
xh, eh = rng.random((N, Fv), np.float32), rng.random((E, Fe), np.float32)
uph = rng.standard_normal((S, O)).astype(np.float32)
x, e, up = (torch.from_numpy(t).to(dev) for t in (xh, eh, uph))


def step():
    layer.forward(x, e)
    return layer.backward(up, need_input_grad=True, need_edge_grad=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
a.record()
for _ in range(REPS):
    dx_py, de_py = step()
b.record()
torch.cuda.synchronize()
py_ms, py_wall = a.elapsed_time(b) / REPS, (time.perf_counter() - t0) / REPS * 1e3
out_py, g_py = layer.output.cpu().numpy(), layer.get_gradients()
dx_py, de_py = dx_py.cpu().numpy(), de_py.cpu().numpy()

# ---- the same batch as the 130 000 graph objects the reference's API takes (local vertex numbers, local edge ids: the edge
#      columns of graph s are its own edges in ascending global id, graph after graph)
gid_of_vertex = np.repeat(np.arange(S), np.diff(voff))
row_of_entry = np.repeat(np.arange(N), np.diff(ia))
eid = ja[1].astype(np.int64)
has = eid > 0
edge_graph = np.zeros(E + 1, np.int64)
edge_graph[eid[has]] = gid_of_vertex[row_of_entry[has]]
order = np.lexsort((np.arange(1, E + 1), edge_graph[1:]))        # global edge ids (0-based) sorted by (graph, id)
new_col = np.empty(E, np.int64)
new_col[order] = np.arange(E)                                     # global id -> column in the Fortran layer's batch
ne = np.bincount(edge_graph[1:], minlength=S)
eoff = np.concatenate([[0], np.cumsum(ne)])
local_eid = np.where(has, new_col[np.maximum(eid, 1) - 1] - eoff[gid_of_vertex[row_of_entry]] + 1, 0).astype(np.int32)
local_col = (ja[0].astype(np.int64) - voff[gid_of_vertex[row_of_entry]]).astype(np.int32)


def i32(*v):
    return np.asarray(v, np.int32).tobytes()


def mat(m):
    m = np.ascontiguousarray(m, np.float32)
    return i32(m.shape[1], m.shape[0]) + m.tobytes()


def actv(name):
    return name.ljust(16).encode() + np.asarray([1.0, 0.0, 0.0], np.float32).tobytes()     # name, scale, p0, p1


parts = [i32(6, S)]
for s in range(S):
    v0, v1 = int(voff[s]), int(voff[s + 1])
    w0, w1 = int(ia[v0]) - 1, int(ia[v1]) - 1
    parts.append(i32(v1 - v0, int(ne[s]), w1 - w0))
    parts.append((ia[v0:v1 + 1].astype(np.int64) - w0).astype(np.int32).tobytes())
    parts.append(np.stack([local_col[w0:w1], local_eid[w0:w1]], axis=1).astype(np.int32).tobytes())   # (2, nnz) column-major
params = layer.get_params()
parts += [i32(T, Fv, Fe, mn, mx, O), actv("sigmoid"), actv("softmax"), i32(params.size), params.tobytes(),
          mat(xh), mat(eh[order]), mat(uph), i32(REPS)]
with tempfile.TemporaryDirectory() as td:
    case, res = os.path.join(td, "case.bin"), os.path.join(td, "result.bin")
    with open(case, "wb") as f:
        for p in parts:
            f.write(p)
    t0 = time.perf_counter()
    r = subprocess.run([RUNNER, case, res], capture_output=True, text=True, timeout=1500)
    wall = time.perf_counter() - t0
    if r.returncode != 0:
        print(json.dumps({"ok": False, "error": f"athena_mp_layer_run rc {r.returncode}: {r.stderr[-600:]}"}))
        sys.exit(1)
    buf = open(res, "rb").read()
off = 0


def ints(n):
    global off
    v = np.frombuffer(buf, np.int32, n, off)
    off += 4 * n
    return v


def rmat():
    global off
    f, n = (int(t) for t in ints(2))
    v = np.frombuffer(buf, np.float32, f * n, off).reshape(n, f)
    off += 4 * f * n
    return v


def rvec():
    global off
    n = int(ints(1)[0])
    v = np.frombuffer(buf, np.float32, n, off)
    off += 4 * n
    return v


out_f, dx_f, de_f, g_f = rmat(), rmat(), rmat(), rvec()
f_ms = float(rvec()[0])
g_f2 = rvec()
line = {
    "workload": f"duvenaud layer, T = {T}, F_v {Fv}, F_e {Fe}, degrees {mn}..{mx}, {O} outputs; {S} graphs = {N} vertices / {nnz} entries; "
                f"forward + reverse (dx, de, {2 * T} parameter gradients)",
    "fortran": {"driver": "duvenaud_mp_layer_type%forward_dev / backward_dev (athena_mp_layer_run kind 6), set_graph on the batch of "
                          f"{S} mp_graph_type objects", "ms_per_step": round(f_ms, 4), "entry_visits_per_s": T * nnz / f_ms * 1e3,
                "reps": REPS, "program_wall_s": round(wall, 1)},
    "python": {"driver": "athena_amd.layers.duvenaud_msgpass_layer_type.forward / backward", "ms_per_step_gpu_events": round(py_ms, 4),
               "ms_per_step_wall": round(py_wall, 4)},
    "fortran_vs_python": {"output_rel": rel(out_f, out_py), "dx_rel": rel(dx_f, dx_py), "de_rel": rel(de_f, de_py[order]),
                          "param_grads_rel": rel(g_f, g_py), "gradients_after_timed_loop_bit_identical": bool(np.array_equal(g_f, g_f2))},
}
line["ok"] = bool(max(line["fortran_vs_python"][k] for k in ("output_rel", "dx_rel", "de_rel", "param_grads_rel")) <= 1e-5
                  and line["fortran_vs_python"]["gradients_after_timed_loop_bit_identical"])
print(json.dumps(line))
