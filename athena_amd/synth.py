"""Seeded synthetic workloads of SURVEY.md 8d (the reference ships no benchmark inputs)."""
import numpy as np


def random_graph_csr(n, n_pairs, seed=20260424, self_loops=True):
    """C2/C5 generator: n_pairs i.i.d. uniform undirected pairs (u != v, duplicates kept) + one
    self-loop per vertex; rows in vertex order, neighbours in generation order; adj_ja(2,.) =
    pair id (1-based), 0 for self-loops.  Returns Fortran-convention (adj_ia, adj_ja)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    u = rng.integers(0, n, n_pairs, dtype=np.int64)
    v = rng.integers(0, n - 1, n_pairs, dtype=np.int64)
    v = v + (v >= u)  # u != v, still uniform
    pid = np.arange(1, n_pairs + 1, dtype=np.int64)
    src = np.concatenate([u, v])
    dst = np.concatenate([v, u])
    eid = np.concatenate([pid, pid])
    if self_loops:
        loops = np.arange(n, dtype=np.int64)
        src = np.concatenate([loops, src])  # self-loop first in each row
        dst = np.concatenate([loops, dst])
        eid = np.concatenate([np.zeros(n, np.int64), eid])
    order = np.argsort(src, kind="stable")
    src, dst, eid = src[order], dst[order], eid[order]
    counts = np.bincount(src, minlength=n)
    adj_ia = np.concatenate([[1], 1 + np.cumsum(counts)]).astype(np.int32)
    adj_ja = np.empty((2, src.size), np.int32, order="F")
    adj_ja[0] = dst + 1
    adj_ja[1] = eid
    return adj_ia, adj_ja


def kipf_inputs(n, F, seed=1):
    """X ~ U(-1,1) (seed), W ~ N(0, 2/F) (seed+1), dZ ~ U(-1,1) (seed+2); fp32."""
    x = np.random.Generator(np.random.PCG64(seed)).uniform(-1, 1, (n, F)).astype(np.float32)
    w = (np.random.Generator(np.random.PCG64(seed + 1)).standard_normal(F * F) * np.sqrt(2.0 / F)).astype(np.float32)
    dz = np.random.Generator(np.random.PCG64(seed + 2)).uniform(-1, 1, (n, F)).astype(np.float32)
    return x, w, dz
