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


def kipf_weight(F, seed=2):
    """W ~ N(0, 2/F), flat W(F, F) column-major; fp32"""
    return (np.random.Generator(np.random.PCG64(seed)).standard_normal(F * F) * np.sqrt(2.0 / F)).astype(np.float32)


def kipf_inputs(n, F, seed=1):
    """X ~ U(-1,1) (seed), W ~ N(0, 2/F) (seed+1), dZ ~ U(-1,1) (seed+2); fp32."""
    def stream(sd):      # the same values as one (n, F) draw, produced 2^20 rows at a time (the float64 draw of a
        out = np.empty((n, F), np.float32)          # configs[4]-sized tensor would take 20 GB of host memory at once)
        for r0 in range(0, n, 1 << 20):
            r1 = min(n, r0 + (1 << 20))
            out[r0:r1] = feature_block(sd, r0, r1, F)
        return out
    if n * F <= (1 << 28):
        x = np.random.Generator(np.random.PCG64(seed)).uniform(-1, 1, (n, F)).astype(np.float32)
        dz = np.random.Generator(np.random.PCG64(seed + 2)).uniform(-1, 1, (n, F)).astype(np.float32)
    else:
        x, dz = stream(seed), stream(seed + 2)
    w = kipf_weight(F, seed + 1)
    return x, w, dz


def molecule_batch(n_graphs, seed=3):
    """C3: QM9-shaped batch as ONE block-diagonal graph.  Vertices per graph ~
    clip(round(N(18,3)),4,29); bonds = a random spanning tree (parent among the two previous atoms) +
    Poisson(1.5) ring-closing bonds (5/6-rings) where both atoms have degree < 4; every vertex gets a
    self-loop (edge id 0).  Returns (adj_ia, adj_ja, vertex_offsets[S+1], num_edges)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nv = np.clip(np.rint(rng.normal(18, 3, n_graphs)), 4, 29).astype(np.int64)
    voff = np.concatenate([[0], np.cumsum(nv)])
    N = int(voff[-1])
    gid = np.repeat(np.arange(n_graphs), nv)
    loc = np.arange(N) - voff[gid]
    # spanning tree
    child = np.nonzero(loc >= 1)[0]
    step = np.where(loc[child] >= 2, rng.integers(1, 3, child.size), 1)
    parent = child - step
    u = [parent]; v = [child]
    deg = np.bincount(np.concatenate([parent, child]), minlength=N)
    # ring closures
    n_extra = rng.poisson(1.5, n_graphs)
    g_rep = np.repeat(np.arange(n_graphs), n_extra)
    span = rng.integers(4, 6, g_rep.size)
    room = nv[g_rep] - span
    ok = room > 0
    g_rep, span, room = g_rep[ok], span[ok], room[ok]
    a = voff[g_rep] + (rng.random(g_rep.size) * room).astype(np.int64)
    b = a + span
    key = a * 8 + span
    _, first = np.unique(key, return_index=True)
    a, b = a[np.sort(first)], b[np.sort(first)]
    keep = (deg[a] < 4) & (deg[b] < 4)
    a, b = a[keep], b[keep]
    u.append(a); v.append(b)
    u = np.concatenate(u); v = np.concatenate(v)
    E = u.size
    eid = np.arange(1, E + 1, dtype=np.int64)
    loops = np.arange(N, dtype=np.int64)
    src = np.concatenate([loops, u, v])
    dst = np.concatenate([loops, v, u])
    ee = np.concatenate([np.zeros(N, np.int64), eid, eid])
    order = np.argsort(src, kind="stable")
    src, dst, ee = src[order], dst[order], ee[order]
    adj_ia = np.concatenate([[1], 1 + np.cumsum(np.bincount(src, minlength=N))]).astype(np.int32)
    adj_ja = np.empty((2, src.size), np.int32, order="F")
    adj_ja[0] = dst + 1
    adj_ja[1] = ee
    return adj_ia, adj_ja, voff.astype(np.int32), E


def morton_order(pts, bits=7):
    """permutation that sorts points of the unit cube by the Morton (Z-order) code of their cell on a 2^bits grid per
    axis: consecutive vertex ids are spatial neighbours, so a contiguous row block is a compact region of the mesh
    (what a mesher's own numbering, or any partitioner, gives a real mesh; 2^k equal blocks are the octants)"""
    g = np.minimum((np.asarray(pts) * (1 << bits)).astype(np.int64), (1 << bits) - 1)
    code = np.zeros(g.shape[0], np.int64)
    dim = g.shape[1]
    for b in range(bits):
        for a in range(dim):
            code |= ((g[:, a] >> b) & 1) << (b * dim + (dim - 1 - a))
    return np.argsort(code, kind="stable")


def radius_graph(n_points, mean_degree=15.0, seed=4, dim=3, order=None):
    """C4: points uniform in the unit cube, undirected radius graph with the radius
    chosen for the requested mean degree, no self-loops; edge feature = x_i - x_j for the pair as
    generated, shared by both directions (the reference's convention).
    order=None: vertices numbered as drawn (the single-GPU workload of BASELINE configs[3]: every gather is a random
    row); order="cells": the SAME points numbered in Morton order of their cell (`morton_order`) -- the mesh as a row
    partition wants it: a contiguous block is a compact region and its halo a thin shell.
    Returns (adj_ia, adj_ja, coords[E, dim])."""
    from scipy.spatial import cKDTree

    rng = np.random.Generator(np.random.PCG64(seed))
    pts = rng.random((n_points, dim))
    if order == "cells":
        pts = pts[morton_order(pts)]
    elif order is not None:
        raise ValueError("order: None or 'cells'")
    r = (mean_degree / (n_points * 4.0 / 3.0 * np.pi)) ** (1.0 / 3.0)
    pairs = cKDTree(pts).query_pairs(r, output_type="ndarray")
    i, j = pairs[:, 0].astype(np.int64), pairs[:, 1].astype(np.int64)
    E = i.size
    coords = (pts[i] - pts[j]).astype(np.float32)
    eid = np.arange(1, E + 1, dtype=np.int64)
    src = np.concatenate([i, j]); dst = np.concatenate([j, i]); ee = np.concatenate([eid, eid])
    order = np.argsort(src, kind="stable")
    src, dst, ee = src[order], dst[order], ee[order]
    adj_ia = np.concatenate([[1], 1 + np.cumsum(np.bincount(src, minlength=n_points))]).astype(np.int32)
    adj_ja = np.empty((2, src.size), np.int32, order="F")
    adj_ja[0] = dst + 1
    adj_ja[1] = ee
    return adj_ia, adj_ja, coords


def random_graph_csr_rows(n, n_pairs, r0, r1, seed=20260424, locality=None):
    """Rows [r0, r1) of the graph random_graph_csr(n, n_pairs, seed) builds, without building the rest: the same
    pair stream, the same entry order inside a row (self-loop, then the pairs that list the vertex first, then those
    that list it second, each in generation order).  Returns (adj_ia [r1-r0+1] 1-based, cols [nnz] 0-based GLOBAL
    vertex ids).  locality=(w, p): the C5 locality variant of SURVEY.md 8d -- with probability p the second endpoint
    is drawn with 0 < |u - v| <= w (reflected at the ends of the vertex range), otherwise uniform."""
    rng = np.random.Generator(np.random.PCG64(seed))
    u = rng.integers(0, n, n_pairs, dtype=np.int64)
    v = rng.integers(0, n - 1, n_pairs, dtype=np.int64)
    v = v + (v >= u)
    if locality is not None:
        w, p = locality
        near = rng.random(n_pairs) < p
        off = rng.integers(1, w + 1, n_pairs, dtype=np.int64) * np.where(rng.random(n_pairs) < 0.5, -1, 1)
        vn = u + off
        bad = (vn < 0) | (vn >= n)
        vn = np.where(bad, u - off, vn)
        v = np.where(near, vn, v)
    mu = (u >= r0) & (u < r1)
    mv = (v >= r0) & (v < r1)
    loops = np.arange(r0, r1, dtype=np.int64)
    src = np.concatenate([loops, u[mu], v[mv]])
    dst = np.concatenate([loops, v[mu], u[mv]])
    order = np.argsort(src, kind="stable")
    src, dst = src[order], dst[order]
    counts = np.bincount(src - r0, minlength=r1 - r0)
    adj_ia = np.concatenate([[1], 1 + np.cumsum(counts)]).astype(np.int64)
    return adj_ia, dst


def feature_rows(seed, ids, F, lo=-1.0, hi=1.0):
    """rows `ids` of the [n, F] matrix Generator(PCG64(seed)).uniform(lo, hi, (n, F)).astype(float32) (what kipf_inputs
    draws for X and dZ) without drawing the rows before them (PCG64.advance: one 64-bit draw per value)"""
    ids = np.asarray(ids, np.int64)
    out = np.empty((ids.size, F), np.float32)
    for k, i in enumerate(ids):
        bg = np.random.PCG64(seed)
        bg.advance(int(i) * F)
        out[k] = np.random.Generator(bg).uniform(lo, hi, F).astype(np.float32)
    return out


def feature_block(seed, r0, r1, F, lo=-1.0, hi=1.0):
    """rows [r0, r1) of the same matrix as one block"""
    bg = np.random.PCG64(seed)
    bg.advance(int(r0) * F)
    return np.random.Generator(bg).uniform(lo, hi, (r1 - r0, F)).astype(np.float32)
