"""CPU: host-side logic and the C-ABI library surface (no compute calls without a GPU)."""
import ctypes as C
import os

import numpy as np

from helpers import csr_from_index_list, golden


def test_library_loads_and_exports_every_declared_symbol():
    from athena_amd import _capi

    assert os.path.exists(_capi.LIB_PATH), "build with __graft_entry__.build()"
    lib = _capi.load()
    declared = _capi.declared_symbols()
    assert len(declared) >= 30
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    # every declared function has a ctypes prototype (so the binding cannot drift from the header)
    unbound = [s for s in declared if s not in _capi._PROTOS and s != "athena_mp_last_error"]
    assert not unbound, unbound
    assert lib.athena_mp_version() >= 100


def test_graph_type_matches_reference_test_graph():
    t = golden("reference_test_topologies.json")["kipf_layer_6v8e"]
    g = csr_from_index_list(6, t["index_list"])
    assert g.adj_ia.tolist() == [1, 3, 6, 9, 12, 15, 17]          # degrees 2,3,3,3,3,2
    assert g.nnz == 16 and g.adj_ja.shape == (2, 16)
    # each undirected edge id appears exactly twice, with mirrored endpoints
    rows = np.repeat(np.arange(1, 7), np.diff(g.adj_ia))
    for e in range(1, 9):
        w = np.nonzero(g.adj_ja[1] == e)[0]
        assert w.size == 2
        assert {(rows[w[0]], g.adj_ja[0, w[0]]), (rows[w[1]], g.adj_ja[0, w[1]])} == \
            {(t["index_list"][0][e - 1], t["index_list"][1][e - 1]), (t["index_list"][1][e - 1], t["index_list"][0][e - 1])}
    g.add_self_loops()
    assert g.nnz == 22 and np.diff(g.adj_ia).tolist() == [3, 4, 4, 4, 4, 3]
    rows = np.repeat(np.arange(1, 7), np.diff(g.adj_ia))
    loops = g.adj_ja[0] == rows
    assert loops.sum() == 6 and np.all(g.adj_ja[1, loops] == 0)
    g.add_self_loops()  # idempotent
    assert g.nnz == 22


def test_synth_generator_is_seeded_and_well_formed():
    from athena_amd import synth

    ia, ja = synth.random_graph_csr(1000, 4500)
    ia2, ja2 = synth.random_graph_csr(1000, 4500)
    assert np.array_equal(ia, ia2) and np.array_equal(ja, ja2)
    assert ia[0] == 1 and ia[-1] - 1 == ja.shape[1] == 10000
    rows = np.repeat(np.arange(1, 1001), np.diff(ia))
    assert np.all(ja[0, ia[:-1] - 1] == np.arange(1, 1001))       # self loop first in each row
    assert np.all((ja[1] == 0) == (ja[0] == rows))
    assert ja[0].min() >= 1 and ja[0].max() <= 1000
