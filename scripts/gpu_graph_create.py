"""graph handle creation time: host builder vs device builder (ATHENA_MP_GRAPH_BUILD) at C2 and C4 sizes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import DeviceGraph, synth, _capi
_capi.init(0)
def t(label, fn, reps=3):
    fn().close()
    best = 1e9
    for _ in range(reps):
        s = time.perf_counter(); g = fn(); best = min(best, time.perf_counter() - s); g.close()
    print("%-58s %8.1f ms" % (label, best * 1e3))
ia, ja = synth.random_graph_csr(1000000, 4500000)
for mode in ("host", "device"):
    os.environ["ATHENA_MP_GRAPH_BUILD"] = mode
    t(f"C2 1M vertices / 10M entries, no edge ids   [{mode}]", lambda: DeviceGraph(ia, ja, n_edge_cols=0))
    t(f"C2 1M vertices / 10M entries, with edge ids [{mode}]", lambda: DeviceGraph(ia, ja))
ia, ja, coords = synth.radius_graph(2000000)
for mode in ("host", "device"):
    os.environ["ATHENA_MP_GRAPH_BUILD"] = mode
    t(f"C4 2M points / {ja.shape[1]/1e6:.1f}M entries / {coords.shape[0]/1e6:.1f}M edge columns [{mode}]",
      lambda: DeviceGraph(ia, ja, n_edge_cols=coords.shape[0]), reps=2)
# edge list -> CSR: numpy mirror vs the device builder
import numpy as np
from athena_amd.graph import graph_type
rng = np.random.default_rng(0)
for n, E in ((1000000, 4500000), (2000000, 14800000)):
    idx = rng.integers(1, n + 1, (2, E)).astype(np.int32)
    for name, fn in (("host (numpy lexsort)", lambda g: (g.generate_adjacency(idx), g.add_self_loops())),
                     ("device (radix sort)", lambda g: g.generate_adjacency_device(idx, add_self_loops=True))):
        g = graph_type(); g.set_num_vertices(n, 1)
        if name.startswith("device"):
            fn(g); g = graph_type(); g.set_num_vertices(n, 1)     # warm
        s = time.perf_counter(); fn(g)
        print("edge list -> CSR + self loops, %d vertices / %d pairs [%s] %9.1f ms" % (n, E, name, (time.perf_counter() - s) * 1e3))
