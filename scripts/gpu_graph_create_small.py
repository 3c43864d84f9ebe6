"""graph handle creation latency for mini-batch sized graphs (block-diagonal batch of molecules)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from athena_amd import DeviceGraph, synth, _capi
_capi.init(0)
for S in (1, 32, 256, 4096):
    ia, ja, voff, E = synth.molecule_batch(S)
    for _ in range(3): DeviceGraph(ia, ja, n_edge_cols=E).close()
    t = time.perf_counter(); reps = 50
    for _ in range(reps): g = DeviceGraph(ia, ja, n_edge_cols=E); g.close()
    print(f"{S:5d} graphs, {ia.size - 1:6d} vertices, {ja.shape[1]:7d} entries: create+destroy {(time.perf_counter() - t) / reps * 1e3:.3f} ms")
