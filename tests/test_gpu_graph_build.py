"""GPU: the device-side builder of the graph handle (athena_amd/csrc/graph_build.hip) against the host
builder (capi.hip) -- every array of the handle compared element for element (integer / index work: bit
exact; the coefficient too, since both take it from the same host powf), on graphs that exercise edge ids,
self-loop entries without one, rectangular shards with explicit degrees, hub rows, empty rows and columns."""
import os

import numpy as np
import pytest

from helpers import csr_from_index_list, random_graph

pytestmark = pytest.mark.gpu

NAMES = ["rowptr", "col", "eid", "coef", "t_rowptr", "t_src", "t_eid", "t_coef", "e_rowptr", "e_row", "e_entry",
         "deg_row", "deg_col"]


def _build(mode, *a, **kw):
    from athena_amd import DeviceGraph

    old = os.environ.get("ATHENA_MP_GRAPH_BUILD")
    os.environ["ATHENA_MP_GRAPH_BUILD"] = mode
    try:
        return DeviceGraph(*a, **kw)
    finally:
        if old is None:
            del os.environ["ATHENA_MP_GRAPH_BUILD"]
        else:
            os.environ["ATHENA_MP_GRAPH_BUILD"] = old


def _same(a, b):
    for n in NAMES:
        x, y = a.export(n), b.export(n)
        assert x.shape == y.shape, n
        assert np.array_equal(x, y), f"{n} differs between the host and the device builder"


@pytest.mark.parametrize("n,pairs,loops,iso", [(50, 120, True, 3), (5000, 20000, True, 10), (20000, 30000, False, 500), (1, 0, True, 0),
                                               # entry counts on both sides of the radix passes' 4 096-entry tiles (radix_sort.h), one and two tiles
                                               (95, 2000, True, 0), (96, 2000, True, 0), (97, 2000, True, 0), (192, 4000, True, 0), (300, 70000, True, 7)])
def test_device_builder_matches_host_builder(dev, n, pairs, loops, iso):
    ia, ja = random_graph(n, pairs, seed=n, self_loops=loops, isolated=iso)
    E = int(ja[1].max()) if ja.shape[1] else 0
    _same(_build("host", ia, ja, n_edge_cols=E), _build("device", ia, ja, n_edge_cols=E))
    _same(_build("host", ia, ja, n_edge_cols=0), _build("device", ia, ja, n_edge_cols=0))      # Kipf: no edge index


def test_device_builder_rectangular_shard_hubs_and_duplicates(dev, oracle):
    from athena_amd import ops
    import torch

    rng = np.random.default_rng(3)
    n_rows, n_cols = 3000, 4100
    deg = rng.poisson(6, n_rows).astype(np.int64)
    deg[[5, 77]] = [900, 1500]                     # hub rows (> 512 entries): long-row plan
    deg[[0, 1, 2999]] = 0                          # empty rows
    ia = np.concatenate([[1], 1 + np.cumsum(deg)]).astype(np.int32)
    nnz = int(deg.sum())
    ja = np.zeros((2, nnz), np.int32, order="F")
    ja[0] = rng.integers(1, n_cols - 50, nnz)      # the last 50 columns are never referenced; duplicates occur
    ja[1] = rng.integers(0, 40, nnz)               # few edge columns, many entries each, id 0 = none
    rd = rng.integers(1, 30, n_rows).astype(np.int32); cd = rng.integers(1, 30, n_cols).astype(np.int32)
    a = _build("host", ia, ja, n_cols=n_cols, n_edge_cols=39, row_deg=rd, col_deg=cd)
    b = _build("device", ia, ja, n_cols=n_cols, n_edge_cols=39, row_deg=rd, col_deg=cd)
    _same(a, b)
    x = torch.from_numpy(rng.uniform(-1, 1, (n_cols, 24)).astype(np.float32)).to(dev)
    g = torch.from_numpy(rng.uniform(-1, 1, (n_rows, 24)).astype(np.float32)).to(dev)
    assert torch.equal(ops.kipf_propagate(a, x), ops.kipf_propagate(b, x))
    assert torch.equal(ops.kipf_propagate_bwd(a, g), ops.kipf_propagate_bwd(b, g))


def test_device_builder_reports_bad_entries_like_the_host_builder(dev):
    from athena_amd import _capi

    ia = np.array([1, 3, 4], np.int32)
    for bad, msg in ((np.array([[1, 9, 2], [0, 0, 0]], np.int32), r"adj_ja\(1,2\) = 9 outside"),
                     (np.array([[1, 2, 2], [0, 7, 0]], np.int32), r"adj_ja\(2,2\) = 7 outside")):
        for mode in ("host", "device"):
            with pytest.raises(_capi.AthenaMPError, match=msg):
                _build(mode, ia, np.asfortranarray(bad), n_edge_cols=3)


def test_default_choice_builds_large_graphs_on_the_device_with_identical_results(dev, oracle):
    """C2-shaped graph at 1/4 size (2.5 M entries: above the 2^18-entry switch): default build == host build"""
    from athena_amd import DeviceGraph, synth

    ia, ja = synth.random_graph_csr(250_000, 1_125_000)
    a = _build("host", ia, ja, n_edge_cols=0)
    b = DeviceGraph(ia, ja, n_edge_cols=0)
    _same(a, b)


def test_graph_handles_do_not_leak_device_memory(dev):
    """create / use / destroy many handles with both builders (incl. the lazily built bucket tiles and
    length-sorted row orders): free device memory returns to where it started"""
    import torch
    from athena_amd import ops

    ia, ja = random_graph(30000, 120000, seed=9, self_loops=True, isolated=5)
    E = int(ja[1].max())
    x = torch.rand((30000, 64), device=dev)
    e = torch.rand((E, 8), device=dev)
    w = torch.rand(64 * 72 * 4, device=dev)

    def cycle(mode):
        g = _build(mode, ia, ja, n_edge_cols=E)
        ops.kipf_propagate(g, x)
        a = ops.duvenaud_propagate(g, x, e)
        ops.duvenaud_update_act(g, a, w, 1, 4, 64, act="sigmoid")      # builds the bucket tiles
        g.close()

    for mode in ("host", "device"):
        cycle(mode)                                                    # warm: workspaces, pools
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(40):
        cycle("host"); cycle("device")
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 8 << 20, f"{(free0 - free1) >> 20} MiB of device memory lost over 80 graph handles"


@pytest.mark.parametrize("n,E,loops", [(1, 0, True), (7, 5, False), (300, 900, True), (5000, 40000, True), (5000, 40000, False),
                                       (100000, 450000, True)])
def test_edge_list_to_csr_on_the_device_equals_the_host_mirror(dev, n, E, loops):
    """generate_adjacency (+ add_self_loops) built by one device radix sort == athena_amd.graph.graph_type's host
    methods, element for element: self pairs (one entry), duplicate pairs, isolated vertices, vertices that
    already carry a self edge (no second loop), id-less loops first inside a row"""
    from athena_amd import _capi
    from athena_amd.graph import graph_type

    rng = np.random.default_rng(n + E)
    idx = rng.integers(1, n + 1, (2, E)).astype(np.int64)
    if E >= 5:
        idx[:, 0] = [3, 3] if n >= 3 else [1, 1]          # a self pair
        idx[:, 1] = idx[:, 2]                              # a duplicate pair
    h = graph_type(); h.set_num_vertices(n, 1); h.generate_adjacency(idx)
    d = graph_type(); d.set_num_vertices(n, 1); d.generate_adjacency_device(idx, add_self_loops=False)
    assert np.array_equal(h.adj_ia, d.adj_ia) and np.array_equal(h.adj_ja, d.adj_ja)
    if loops:
        h.add_self_loops()
        d2 = graph_type(); d2.set_num_vertices(n, 1); d2.generate_adjacency_device(idx, add_self_loops=True)
        assert np.array_equal(h.adj_ia, d2.adj_ia) and np.array_equal(h.adj_ja, d2.adj_ja)
        assert d2.nnz == h.nnz
    bad = idx.copy()
    if E:
        bad[1, E // 2] = n + 5
        with pytest.raises(_capi.AthenaMPError, match=r"index_list\(:,%d\)" % (E // 2 + 1)):
            graph_type_dev = graph_type(); graph_type_dev.set_num_vertices(n, 1); graph_type_dev.generate_adjacency_device(bad)


@pytest.mark.parametrize("n,pairs,loops,edge_ids,mode", [(60, 150, True, True, "host"), (60, 150, True, False, "device"),
                                                         (40000, 150000, True, True, "device"), (40000, 150000, False, False, None),
                                                         (300000, 200000, True, True, None), (5, 0, True, True, None)])
def test_handle_from_an_edge_list_in_one_call(dev, n, pairs, loops, edge_ids, mode):
    """athena_mp_graph_create_from_edges = generate_adjacency [+ add_self_loops] + set_graph with the CSR entries kept
    in HBM in between: every array of the handle equals the two-call route's (csr_from_edges -> graph_create), and the
    adjacency it can hand back equals generate_adjacency_device's, element for element"""
    from athena_amd import DeviceGraph
    from athena_amd.graph import graph_type

    rng = np.random.default_rng(n + pairs)
    idx = rng.integers(1, n + 1, (2, pairs)).astype(np.int32) if pairs else np.zeros((2, 0), np.int32)
    if pairs:
        idx[1, idx[0] == idx[1]] = idx[1, idx[0] == idx[1]] % n + 1          # a few self pairs stay when n is tiny
    old = os.environ.get("ATHENA_MP_GRAPH_BUILD")
    if mode:
        os.environ["ATHENA_MP_GRAPH_BUILD"] = mode
    try:
        g = graph_type()
        g.num_vertices = n
        g.generate_adjacency_device(idx, add_self_loops=loops)
        two = DeviceGraph(g.adj_ia, g.adj_ja, n_edge_cols=pairs if edge_ids else 0)
        one, ia, ja = DeviceGraph.from_edges(n, idx, add_self_loops=loops, with_edge_ids=edge_ids, want_adjacency=True)
        lean = DeviceGraph.from_edges(n, idx, add_self_loops=loops, with_edge_ids=edge_ids)
    finally:
        if mode:
            if old is None:
                del os.environ["ATHENA_MP_GRAPH_BUILD"]
            else:
                os.environ["ATHENA_MP_GRAPH_BUILD"] = old
    assert np.array_equal(ia, g.adj_ia) and np.array_equal(ja, g.adj_ja)
    assert (one.n_rows, one.nnz, one.n_edge_cols) == (two.n_rows, two.nnz, two.n_edge_cols)
    _same(one, two)
    _same(lean, two)
