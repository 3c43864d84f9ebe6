"""BASELINE configs[2] through the LAYER mirror (T = 4 Duvenaud time steps + readout, forward and reverse pass)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import synth
from athena_amd.graph import graph_type
from athena_amd.layers import duvenaud_msgpass_layer_type
S = int(sys.argv[1]) if len(sys.argv) > 1 else 130000
ia, ja, voff, E = synth.molecule_batch(S)
g = graph_type.from_csr(ia, ja, num_edges=E)
N = ia.size - 1
layer = duvenaud_msgpass_layer_type(num_vertex_features=[64], num_edge_features=[8], num_time_steps=4, max_vertex_degree=10,
                                    num_outputs=10, min_vertex_degree=1, seed=1)
layer.set_graph([g])
layer._seg = torch.from_numpy(voff.astype(np.int32)).to(layer.device); layer.graph.batch = S
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.random((N, 64), np.float32)).to(layer.device)
e = torch.from_numpy(rng.random((E, 8), np.float32)).to(layer.device)
up = torch.from_numpy(rng.standard_normal((S, 10)).astype(np.float32)).to(layer.device)
def step():
    layer.forward(x, e); layer.backward(up, need_input_grad=True)
for _ in range(3): step()
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); a.record()
for _ in range(10): step()
b.record(); torch.cuda.synchronize()
print("Duvenaud layer, T = 4, %d graphs / %d vertices / %d entries: %.3f ms per fwd+bwd (GPU events), %.3f ms wall" % (
    S, N, ja.shape[1], a.elapsed_time(b) / 10, (time.perf_counter() - t0) / 10 * 1e3))
