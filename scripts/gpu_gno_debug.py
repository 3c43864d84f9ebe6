"""Debug aid: GNO aggregate of a radius graph with the default fused kernel and with ATHENA_MP_GNO_FUSED_V1=1
(each in its own process: the switch is read once), rows that differ printed with their lengths and tile slots."""
import os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
if len(sys.argv) > 2:
    import torch
    from athena_amd import ops, synth
    from athena_amd.graph import DeviceGraph
    ia, ja, coords = synth.radius_graph(N)
    rng = np.random.default_rng(1)
    Fi = Fo = H = 64; d = 3
    g = DeviceGraph(ia, ja, n_edge_cols=coords.shape[0])
    T = lambda a: torch.from_numpy(a).to("cuda:0")
    x = T(rng.uniform(-1, 1, (N, Fi)).astype(np.float32)); co = T(coords)
    theta = T((0.3 * rng.standard_normal(H * d + H + Fo * Fi * H + Fo * Fi)).astype(np.float32))
    m = ops.gno_aggregate(g, theta, co, x, d, H, Fo)
    np.save(sys.argv[2], m.cpu().numpy()); np.save(sys.argv[2] + ".deg.npy", np.diff(ia))
    sys.exit(0)
env = dict(os.environ)
subprocess.check_call([sys.executable, __file__, str(N), "/tmp/gno_new.npy"], env=env)
env["ATHENA_MP_GNO_FUSED_V1"] = "1"
subprocess.check_call([sys.executable, __file__, str(N), "/tmp/gno_old.npy"], env=env)
a, b, deg = np.load("/tmp/gno_new.npy"), np.load("/tmp/gno_old.npy"), np.load("/tmp/gno_new.npy.deg.npy")
err = np.abs(a - b).max(axis=1) / np.abs(b).max()
bad = np.nonzero(err > 1e-5)[0]
print("rows", N, "bad", bad.size, "max rel err", err.max())
order = np.argsort(-deg, kind="stable"); slot = np.empty(N, np.int64); slot[order] = np.arange(N)
print("degree histogram of bad rows:", np.bincount(deg[bad]) if bad.size else None)
print("degree histogram of all rows:", np.bincount(deg))
for r in bad[:40]:
    s = slot[r]
    print(f"row {r} deg {deg[r]} slot {s} tile {s // 32} (tile % 256 = {(s // 32) % 256}, round {(s // 32) // 256}) in-tile {s % 32} wave {(s % 32) // 4} err {err[r]:.3g}")
