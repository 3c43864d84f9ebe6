"""msgpass_euler network (7 Kipf layers, [input || previous], 12 800-vertex mesh): train-step latency,
eager (one Python call + launch per op) vs the step captured into a HIP graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import optim
from athena_amd.graph import graph_type
from athena_amd.layers import kipf_msgpass_layer_type
from athena_amd.network import network_type

EULER = [([3, 6], "softmax"), ([9, 14], "softmax"), ([17, 32], "softmax"), ([35, 64], "softmax"),
         ([67, 32], "softmax"), ([35, 14], "softmax"), ([17, 7], "swish")]
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "euler_mesh_edges.npz"))
n = int(d["num_vertices"])
g = graph_type(); g.set_num_vertices(n, 3); g.generate_adjacency(d["index_list"]); g.add_self_loops()
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (n, 3)).astype(np.float32); y = rng.uniform(-1, 1, (n, 7)).astype(np.float32)
net = network_type()
for k, (nvf, act) in enumerate(EULER):
    l = kipf_msgpass_layer_type(num_vertex_features=nvf, num_time_steps=1, activation=act, seed=10 + k)
    net.add(l) if k == 0 else net.add(l, input_list=[0, -1], operator="concatenate")
net.set_graph(g)
net.compile(optim.adam_optimiser_type(learning_rate=0.005, clip_dict=optim.clip_type(clip_norm=1.0)))
loss = optim.mse_loss_type()
dev = net.layers[0].device
xd = torch.from_numpy(x).to(dev); yd = torch.from_numpy(y).to(dev)

def eager():
    out = net.forward(xd); l, dl = loss.compute(out, yd); net.backward(dl); net.update(); return l

for _ in range(3): eager()
torch.cuda.synchronize()
t = time.perf_counter(); reps = 50
for _ in range(reps): l = eager()
torch.cuda.synchronize()
print("eager   step: %.3f ms (wall)  loss %.6f" % ((time.perf_counter() - t) / reps * 1e3, l.item()))

replay, xb, tb, lb = net.capture_step(x, y, loss)
def graphed():
    replay(); net.update(); return lb
for _ in range(3): graphed()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps): l = graphed()
torch.cuda.synchronize()
print("graphed step: %.3f ms (wall)  loss %.6f" % ((time.perf_counter() - t) / reps * 1e3, l.item()))
t = time.perf_counter()
for _ in range(reps): replay()
torch.cuda.synchronize()
print("graph replay only (fwd+loss+bwd): %.3f ms" % ((time.perf_counter() - t) / reps * 1e3))
