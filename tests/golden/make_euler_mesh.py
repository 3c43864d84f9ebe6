"""Converts the reference's only realistic in-tree graph -- the Euler "bump" mesh topology,
/root/reference/example/msgpass_euler/data/bump_edgeData_1.txt (37 681 undirected edges over 12 800
vertices; first line = edge count, then one 1-based vertex pair per line, read by
example/msgpass_euler/src/main.f90 `read_graph`) -- into tests/golden/euler_mesh_edges.npz.
Data only (the vertex pairs), stored as int32; no reference source is copied."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
src = "/root/reference/example/msgpass_euler/data/bump_edgeData_1.txt"
with open(src) as fh:
    n = int(fh.readline())
    pairs = np.loadtxt(fh, dtype=np.int32)
assert pairs.shape == (n, 2)
np.savez_compressed(os.path.join(HERE, "euler_mesh_edges.npz"), index_list=pairs.T.copy(), num_vertices=np.int32(pairs.max()))
print("edges", n, "vertices", pairs.max())
