"""edge list -> device handle: the two-call route (csr_from_edges to host arrays, graph_create from them) against
athena_mp_graph_create_from_edges (entries stay in HBM), at the C2 and C4 edge counts"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import DeviceGraph
from athena_amd.graph import graph_type

torch.zeros(1, device="cuda:0")
for n, pairs in ((1000000, 4500000), (2000000, 14800000)):
    rng = np.random.default_rng(1)
    idx = rng.integers(1, n + 1, (2, pairs)).astype(np.int32)
    idx[1, idx[0] == idx[1]] = idx[1, idx[0] == idx[1]] % n + 1
    idx = np.asfortranarray(idx)
    for with_ids in (False, True):
        def two():
            g = graph_type(); g.num_vertices = n
            g.generate_adjacency_device(idx, add_self_loops=True)
            return DeviceGraph(g.adj_ia, g.adj_ja, n_edge_cols=pairs if with_ids else 0)
        def one():
            return DeviceGraph.from_edges(n, idx, add_self_loops=True, with_edge_ids=with_ids)
        def one_adj():
            return DeviceGraph.from_edges(n, idx, add_self_loops=True, with_edge_ids=with_ids, want_adjacency=True)[0]
        res = {}
        for name, fn in (("two calls", two), ("one call", one), ("one call + adjacency back", one_adj)):
            fn().close()
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter(); h = fn(); torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0); h.close()
            res[name] = best * 1e3
        print(f"{n} vertices / {pairs} pairs, edge ids {with_ids}: " + ", ".join(f"{k} {v:.1f} ms" for k, v in res.items()), flush=True)
