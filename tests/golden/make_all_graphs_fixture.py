"""Writes tests/golden/all_graphs_small.txt: four small molecule-like graphs in the reference's
all_graphs.txt layout (example/msgpass_chemical/src/main.f90:353-396), the format statements followed by
hand (ES16.8E2 / I0).  Deterministic; run from the repo root."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from athena_amd.graph import graph_type
from athena_amd.io import write_all_graphs

rng = np.random.default_rng(2024)
graphs, labels = [], []
for nv, pairs in ((3, [[1, 2], [2, 3]]), (5, [[1, 2], [2, 3], [3, 4], [4, 5], [5, 1]]),
                  (4, [[1, 2], [1, 3], [1, 4]]), (6, [[1, 2], [2, 3], [3, 1], [4, 5], [5, 6], [3, 4]])):
    g = graph_type()
    g.set_num_vertices(nv, 6)
    g.set_num_edges(len(pairs), 1)
    g.generate_adjacency(np.array(pairs).T)
    g.add_self_loops()
    g.vertex_features = rng.random((nv, 6)).astype(np.float32)
    g.edge_features = rng.random((len(pairs), 1)).astype(np.float32)
    graphs.append(g); labels.append(np.float32(rng.standard_normal()))
write_all_graphs(os.path.join(os.path.dirname(os.path.abspath(__file__)), "all_graphs_small.txt"), graphs, labels)
