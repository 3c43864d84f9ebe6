"""GPU: BASELINE configs[2]'s layer as the config states it -- duvenaud_msgpass_layer_type, T = 4 time steps + readout, F_v = 64,
F_e = 8, degrees 1..10, 10 outputs (example/msgpass_chemical/src/main.f90:129-138) -- on 2 000 molecule-shaped graphs, against
oracle/layers.py (the reference's loop over samples, athena_duvenaud_msgpass_layer.f90:792-855, and grad_reverse's walk over it).

These widths are the PERFORMANCE route all at once: `a` kept split (vertex part per time step, edge part once per forward pass),
the LDS-staged banded gather of the block-diagonal batch, the MFMA update + activation + readout launch, the one-launch reverse
(readout's and update's reverse of a time step, dc never in HBM) with the edge part of da accumulated over the time steps.  The
layer tests elsewhere use F_v 6 / 8 (the VALU route); scripts/bench_secondary.py checks the same composition at full size in the
driver's bench line -- here it is under pytest.  Output, dx, de and all 8 parameter gradients, Python and Fortran mirrors, 1e-5
(float64-anchored: helpers.assert_close)."""
import numpy as np
import pytest
import torch

import oracle_layers as ol
from helpers import assert_close
from test_gpu_fortran_layers import _act_bytes, _actv, _case_header, _i, _mat, _run

pytestmark = pytest.mark.gpu

NG, Fv, Fe, T, MN, MX, O = 2000, 64, 8, 4, 1, 10, 10


def _split_batch(ia, ja, voff):
    """the per-sample graphs of a block-diagonal batch, edge ids renumbered per sample in ascending order of the batch's ids;
    returns (graphs, [batch edge ids (0-based) of each sample's local columns])"""
    from athena_amd.graph import graph_type

    graphs, eids = [], []
    for s in range(voff.size - 1):
        a0, a1 = int(voff[s]), int(voff[s + 1])
        w0, w1 = int(ia[a0]) - 1, int(ia[a1]) - 1
        jj = ja[:, w0:w1].astype(np.int64)
        ed = np.unique(jj[1][jj[1] > 0])
        rm = np.zeros(int(ed.max()) + 1 if ed.size else 1, np.int64)
        rm[ed] = np.arange(1, ed.size + 1)
        local = np.stack([jj[0] - a0, np.where(jj[1] > 0, rm[np.minimum(jj[1], rm.size - 1)], 0)]).astype(np.int32)
        graphs.append(graph_type.from_csr((ia[a0:a1 + 1].astype(np.int64) - w0).astype(np.int32), np.asfortranarray(local),
                                          num_edges=int(ed.size)))
        eids.append(ed - 1)
    return graphs, eids


@pytest.fixture(scope="module")
def problem():
    from athena_amd import synth

    rng = np.random.default_rng(64)
    ia, ja, voff, E = synth.molecule_batch(NG, seed=11)
    gs, eids = _split_batch(ia, ja, voff)
    N = ia.size - 1
    x = rng.uniform(0, 1, (N, Fv)).astype(np.float32)
    e = rng.uniform(0, 1, (E, Fe)).astype(np.float32)
    xs = [x[voff[s]:voff[s + 1]] for s in range(NG)]
    es = [e[ed] for ed in eids]
    plist = [(rng.standard_normal(Fv * (Fv + Fe) * (MX - MN + 1)) * np.sqrt(2.0 / (Fv + Fe))).astype(np.float32) for _ in range(T)]
    plist += [(rng.standard_normal(O * Fv) * 0.3).astype(np.float32) for _ in range(T)]
    up = rng.standard_normal((NG, O)).astype(np.float32)
    nvf = [Fv] * (T + 1)

    def run():
        out, tapes = ol.duvenaud_forward(gs, xs, es, plist, nvf, Fe, MN, MX, O, "sigmoid")
        dxs, des, grads = ol.duvenaud_backward(gs, es, tapes, plist, nvf, Fe, MN, MX, O, "sigmoid", up)
        return out, np.concatenate(dxs), np.concatenate(des), np.concatenate(grads)

    want = run()
    return dict(ia=ia, ja=ja, voff=voff, E=E, gs=gs, eids=eids, x=x, e=e, xs=xs, es=es, plist=plist, up=up, want=want,
                hi=ol.f64_lazy(run))


def _check(p, out, dx, de, grads, who):
    w, hi = p["want"], p["hi"]
    assert_close(out, w[0], 1e-5, f"{who}: output", f64=hi(0))
    assert_close(dx, w[1], 1e-5, f"{who}: dx", f64=hi(1))
    assert_close(de, w[2], 1e-5, f"{who}: de", f64=hi(2))
    sizes = [Fv * (Fv + Fe) * (MX - MN + 1)] * T + [O * Fv] * T
    names = [f"dW_{t}" for t in range(1, T + 1)] + [f"dR_{t}" for t in range(1, T + 1)]
    cuts = np.cumsum(sizes)[:-1]
    for k, (n, a, b) in enumerate(zip(names, np.split(np.asarray(grads), cuts), np.split(w[3], cuts))):
        assert_close(a, b, 1e-5, f"{who}: {n}", f64=(lambda k=k: np.split(hi(3)(), cuts)[k]))


def test_python_mirror_list_of_samples(dev, problem):
    """set_graph(list of 2 000 graphs): the block-diagonal batch the layer assembles (its own edge numbering, sample by sample)"""
    from athena_amd.layers import duvenaud_msgpass_layer_type

    p = problem
    layer = duvenaud_msgpass_layer_type(num_vertex_features=[Fv], num_edge_features=[Fe], num_time_steps=T, max_vertex_degree=MX,
                                        num_outputs=O, min_vertex_degree=MN, seed=1)
    layer.set_params(np.concatenate(p["plist"]))
    layer.set_graph(p["gs"])
    out = layer.forward(p["xs"], p["es"]).cpu().numpy()
    dx, de = (t.cpu().numpy() for t in layer.backward(p["up"], need_input_grad=True, need_edge_grad=True))
    _check(p, out, dx, de, layer.get_gradients(), "python mirror, list of samples")


def test_python_mirror_one_batched_graph_then_the_same_graph_as_one_sample(dev, problem):
    """set_graph_batched (the batch as ONE graph with vertex offsets: what the bench line runs; the batch's own edge numbering), then
    set_graph of the SAME graph object as a single sample: the cut is part of the layer's graph state beside the cached handle, so
    the second call must give ONE readout row -- the sum of the 2 000 (the readout is additive over vertices)"""
    from athena_amd.graph import graph_type
    from athena_amd.layers import duvenaud_msgpass_layer_type

    p = problem
    layer = duvenaud_msgpass_layer_type(num_vertex_features=[Fv], num_edge_features=[Fe], num_time_steps=T, max_vertex_degree=MX,
                                        num_outputs=O, min_vertex_degree=MN, seed=1)
    layer.set_params(np.concatenate(p["plist"]))
    g = graph_type.from_csr(p["ia"], p["ja"], num_edges=p["E"]).freeze()
    x, e = torch.from_numpy(p["x"]).to(dev), torch.from_numpy(p["e"]).to(dev)
    layer.set_graph_batched(g, p["voff"])
    out = layer.forward(x, e).cpu().numpy()
    dx, de_b = (t.cpu().numpy() for t in layer.backward(torch.from_numpy(p["up"]).to(dev), need_input_grad=True, need_edge_grad=True))
    de = np.concatenate([de_b[ed] for ed in p["eids"]])                    # the oracle's de is per sample, local column order
    _check(p, out, dx, de, layer.get_gradients(), "python mirror, one batched graph")
    handle = layer.graph.device
    layer.set_graph(g)                                                     # same object, same key: no rebuild, but ONE sample now
    assert layer.graph.device is handle and layer.graph.batch == 1
    one = layer.forward(x, e).cpu().numpy()
    assert one.shape == (1, O)
    assert_close(one[0], out.astype(np.float64).sum(0), 1e-5, "one sample = the sum of the batch's readout rows")
    layer.set_graph_batched(g, p["voff"])                                  # and back
    assert layer.graph.device is handle and layer.graph.batch == NG
    assert np.array_equal(layer.forward(x, e).cpu().numpy(), out)


def test_fortran_mirror(dev, problem, tmp_path):
    """duvenaud_mp_layer_type (athena_amd/fortran/athena_mp_layers.f90) through athena_mp_layer_run: the same layer from Fortran"""
    p = problem
    blob = _case_header(2, p["gs"]) + _i(T, Fv, Fe, MN, MX, O) + _act_bytes(_actv("sigmoid")) + _act_bytes(_actv("softmax"))
    blob += _i(sum(q.size for q in p["plist"])) + np.concatenate(p["plist"]).tobytes()
    blob += _mat(np.concatenate(p["xs"])) + _mat(np.concatenate(p["es"])) + _mat(p["up"])
    r = _run(tmp_path, blob)
    out, dx, de, grads = r.matrix(), r.matrix(), r.matrix(), r.vector()
    _check(p, out, dx, de, grads, "fortran mirror")
