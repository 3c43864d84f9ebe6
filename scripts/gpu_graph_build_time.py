"""how long a device graph handle takes to build at BASELINE configs[1] size (1 M vertices / 10 M entries): from the host CSR
(athena_mp_graph_create: upload + the device builder of graph_build.hip) and from the edge list in one call
(athena_mp_graph_create_from_edges); best of 5, cache bypassed.  Under rocprofv3 --kernel-trace --stats the radix passes show up
as radix_hist / radix_scan_rows / radix_scan_bins / radix_scatter."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import DeviceGraph, synth

torch.zeros(1, device="cuda:0")
n, pairs = 1_000_000, 4_500_000
ia, ja = synth.random_graph_csr(n, pairs, seed=1)
rng = np.random.default_rng(1)
idx = rng.integers(1, n + 1, (2, pairs)).astype(np.int32)
idx[1, idx[0] == idx[1]] = idx[1, idx[0] == idx[1]] % n + 1
idx = np.asfortranarray(idx)


def best(fn, reps=5):
    fn().close()
    b = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); h = fn(); torch.cuda.synchronize()
        b = min(b, time.perf_counter() - t0); h.close()
    return b * 1e3


print(f"host CSR -> handle ({ja.shape[1]} entries, no edge columns): {best(lambda: DeviceGraph(ia, ja)):.1f} ms")
print(f"host CSR -> handle ({ja.shape[1]} entries, edge columns):    {best(lambda: DeviceGraph(ia, ja, n_edge_cols=pairs)):.1f} ms")
print(f"edge list -> handle in one call ({pairs} pairs):             {best(lambda: DeviceGraph.from_edges(n, idx, add_self_loops=True)):.1f} ms")
