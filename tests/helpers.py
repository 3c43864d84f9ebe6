"""Shared test utilities: seeded graphs in athena's CSR convention and an INDEPENDENT float64
dense-matrix formulation of each op (from the layer documentation, not from the oracle's loops)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def csr_from_index_list(n, index_list, self_loops=False):
    from athena_amd.graph import graph_type

    g = graph_type()
    g.set_num_vertices(n, 0)
    g.generate_adjacency(np.asarray(index_list))
    if self_loops:
        g.add_self_loops()
    return g


def random_graph(n, n_pairs, seed, self_loops=True, isolated=0):
    """random multigraph; `isolated` trailing vertices get no entries at all (zero-degree rows)."""
    from athena_amd import synth

    m = n - isolated
    ia, ja = synth.random_graph_csr(m, n_pairs, seed=seed, self_loops=self_loops)
    if isolated:
        ia = np.concatenate([ia, np.full(isolated, ia[-1], np.int32)])
    return ia, ja


def dense_adjacency(ia, ja, n_cols=None):
    """A[v,u] = multiplicity of u in row v (float64)"""
    n = ia.size - 1
    A = np.zeros((n, n_cols or n), np.float64)
    for v in range(n):
        for w in range(ia[v] - 1, ia[v + 1] - 1):
            A[v, ja[0, w] - 1] += 1.0
    return A


def kipf_dense(ia, ja, x):
    """D^-1/2 A D^-1/2 X with D = CSR row length (docs/source/layers/msgpass/kipf_msgpass_layer.rst)"""
    A = dense_adjacency(ia, ja)
    deg = np.diff(ia).astype(np.float64)
    with np.errstate(divide="ignore"):
        dinv = np.where(deg > 0, deg ** -0.5, 0.0)
    return (dinv[:, None] * A * dinv[None, :]) @ x.astype(np.float64)


def rel_err(a, b):
    b = np.asarray(b, np.float64)
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


def assert_close(a, b, rtol=1e-5, what="", f64=None):
    """north-star tolerance: 1e-5 relative fp32 (relative to the tensor's scale) against the fp32 oracle `b`.

    f64 (an array, or a callable evaluated only when needed): the SAME reference formulas evaluated in float64
    (oracle/oracle64.py).  Chains of fp32 roundings (a reduction over thousands of vertices, several layers in a row) can
    put the fp32 oracle itself more than 1e-5 from exact arithmetic; the device result then passes when it is no further
    from the float64 value than the oracle is, plus rtol:  |a - f64| <= |b - f64| + rtol * scale.  A device result that is
    simply wrong fails both tests."""
    e = rel_err(a, b)
    if e <= rtol:
        return
    if f64 is not None:
        ref = np.asarray(f64() if callable(f64) else f64, np.float64)
        scale = max(np.abs(ref).max(), 1e-30)
        e_dev = float(np.abs(np.asarray(a, np.float64) - ref).max() / scale)
        e_orc = float(np.abs(np.asarray(b, np.float64) - ref).max() / scale)
        assert e_dev <= e_orc + rtol, (f"{what}: {e:.3e} from the fp32 oracle; {e_dev:.3e} from float64 where the oracle "
                                       f"itself is {e_orc:.3e} (allowed: oracle's distance + {rtol})")
        return
    assert e <= rtol, f"{what}: rel err {e:.3e} > {rtol}"


def assert_close_elementwise(a, b, scale, rtol=1e-5, what=""):
    """element-wise check for sums: |a - b| <= rtol * scale element by element, where `scale` is the sum of the
    MAGNITUDES of the terms of each element (the quantity fp32 summation error is relative to; an element whose terms
    cancel is not held to a relative error of its tiny value, and is not excused by the tensor's maximum either)."""
    a, b, scale = (np.asarray(t, np.float64) for t in (a, b, scale))
    bad = np.abs(a - b) > rtol * scale + 1e-30
    assert not bad.any(), f"{what}: {int(bad.sum())} element(s) beyond {rtol} of their term magnitudes; worst {float((np.abs(a - b) / np.maximum(scale, 1e-30)).max()):.3e}"


def elementwise_worst(a, b, scale):
    """max over the elements of |a - b| / scale, scale = the sum of the MAGNITUDES of each element's terms (the oracle run on
    |operands|): the strongest element-wise statement fp32 sums allow -- an element whose terms cancel is held to the size of
    what was added, not to its tiny value, and no element hides behind the tensor's maximum"""
    a, b, scale = (np.asarray(t, np.float64) for t in (a, b, scale))
    return float((np.abs(a - b) / np.maximum(scale, 1e-30)).max()) if a.size else 0.0
