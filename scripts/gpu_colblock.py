"""experiment: does restricting the gathered columns to a slice that fits the 256 MiB Infinity Cache make the
aggregation faster than its share of the entries?  (column-blocked two-pass aggregation idea)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from athena_amd import DeviceGraph, ops, synth
N, F = 1000000, 128
dev = torch.device("cuda:0")
ia, ja = synth.random_graph_csr(N, 4500000)
x = torch.rand((N, F), device=dev)
rows = np.repeat(np.arange(N), np.diff(ia))
deg = np.diff(ia).astype(np.int32)
def t(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
g = DeviceGraph(ia, ja, n_edge_cols=0)
full = t(lambda: ops.kipf_propagate(g, x))
print("full graph: %d entries  %.3f ms  (%.1f ns/kentry)" % (ja.shape[1], full, full * 1e6 / ja.shape[1] * 1e3 / 1e3))
for frac in (0.5, 0.25, 0.125, 0.03):
    keep = ja[0] <= int(N * frac)
    r = rows[keep]
    sia = np.concatenate([[1], 1 + np.cumsum(np.bincount(r, minlength=N))]).astype(np.int32)
    sja = np.asfortranarray(ja[:, keep])
    gs = DeviceGraph(sia, sja, n_edge_cols=0, row_deg=deg, col_deg=deg)
    tt = t(lambda: ops.kipf_propagate(gs, x))
    print("columns < %.3f N (%4.0f MB slice): %8d entries  %.3f ms = %.2f of full time for %.2f of the entries" % (
        frac, N * frac * F * 4 / 1e6, sja.shape[1], tt, tt / full, sja.shape[1] / ja.shape[1]))
