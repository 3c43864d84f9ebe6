"""athena_mp_gno_aggregate_bwd against the separate dx entry point: which columns differ, and what rows feed them
(row length class of the sources, entry position inside the row)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from athena_amd import DeviceGraph, ops, _capi
from helpers import csr_from_index_list
_capi.init(0)
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
d = 3
rng = np.random.default_rng(1)
pairs = [[i, i + 1] for i in range(1, N - 6)]
pairs += [[int(a), int(b)] for a, b in rng.integers(1, N - 5, (int(float(sys.argv[2]) * N) if len(sys.argv) > 2 else 3 * N, 2)) if a != b]
for k, h in enumerate([int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 and sys.argv[3] else []):
    hub = 11 + 80 * k
    pairs += [[hub, int(v)] for v in rng.choice(np.arange(hub + 2, N - 6), h, replace=False)]
pairs = np.array(pairs).T
g = csr_from_index_list(N, pairs)
E = pairs.shape[1]
ia, ja = g.adj_ia, g.adj_ja
H = Fi = Fo = 64
coords = rng.standard_normal((E, d)).astype(np.float32)
x = rng.uniform(-1, 1, (N, Fi)).astype(np.float32)
theta = (0.3 * rng.standard_normal(H * d + H + Fo * Fi * H + Fo * Fi)).astype(np.float32)
up = rng.uniform(-1, 1, (N, Fo)).astype(np.float32)
if os.environ.get("ZERO_BV"):
    theta[H * d + H + Fo * Fi * H:] = 0
if os.environ.get("ZERO_V"):
    theta[H * d + H:H * d + H + Fo * Fi * H] = 0
dg = DeviceGraph(ia, ja, n_edge_cols=E)
T = lambda a: torch.from_numpy(a).to(dev)
th, co, xd, gd = T(theta), T(coords), T(x), T(up)
dx, dth, _, fused = ops.gno_aggregate_bwd(dg, th, co, xd, gd, d, H)
ref = ops.gno_aggregate_bwd_x(dg, th, co, gd, d, H, Fi)
err = (dx - ref).abs().cpu().numpy()
scale = ref.abs().max().item()
print("fused", fused, "max rel", err.max() / scale)
deg = np.diff(ia)
rows = np.repeat(np.arange(N), deg)
bad_cols = np.nonzero(err.max(1) > 1e-5 * scale)[0]
print("bad columns", bad_cols.size, "of", N, " features with error per bad column (first 5):")
for u in bad_cols[:5]:
    src = rows[ja[0] - 1 == u]
    print(" col", u, "bad features", np.nonzero(err[u] > 1e-5 * scale)[0].tolist()[:20], "source rows lens", deg[src].tolist())
print("feature histogram of errors:", (err > 1e-5 * scale).sum(0).tolist())
# per-entry expectation in float64: px_e = h_e^T G_i + c_i
