"""CPU, world_size 2, gloo: the row-partition / halo-exchange logic of athena_amd/dist.py.
The HIP kernels cannot run here, so an oracle-backed compute backend is injected (tests may use the
oracle); what is under test is the sharding: shard generation, column renumbering, halo plan,
point-to-point exchange, degree exchange, dW all-reduce -- the assembled result must equal the
single-process oracle on the assembled global graph (bit-exact for the two aggregations)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleGraph:
    def __init__(self, ia, ja, n_cols, row_deg, col_deg, n_edge_cols=0):
        self.ia, self.ja, self.n_cols, self.row_deg, self.col_deg = ia, np.asfortranarray(ja), n_cols, row_deg, col_deg
        self.n_rows = ia.size - 1
        self.n_edge_cols = n_edge_cols

    def square(self):
        """the block as a square graph of n_cols vertices (rows beyond n_rows are empty): what the oracle's loops take"""
        ia = np.concatenate([self.ia, np.full(self.n_cols - self.n_rows, self.ia[-1], self.ia.dtype)]).astype(np.int32)
        return ia, self.ja


class OracleBackend:
    """same call surface as athena_amd.dist.HipBackend, computed by the CPU oracle"""

    def __init__(self):
        from oracle import oracle
        self.o = oracle

    def make_graph(self, ia, ja, n_cols, row_deg, col_deg, n_edge_cols=0):
        return OracleGraph(ia, ja, n_cols, row_deg, col_deg, n_edge_cols)

    def kipf_propagate(self, g, x, out):
        out.copy_(torch.from_numpy(self.o.kipf_propagate_rect(x.numpy(), g.ia, g.ja, g.row_deg, g.col_deg)))

    def neighbour_sum(self, g, x, out):
        ones_r, ones_c = np.ones(g.n_rows, np.int32), np.ones(g.n_cols, np.int32)   # coefficient 1
        out.copy_(torch.from_numpy(self.o.kipf_propagate_rect(x.numpy(), g.ia, g.ja, ones_r, ones_c)))

    def kipf_layer_fwd(self, g, x, W, Fo, P, Z):
        self.kipf_propagate(g, x, P)
        Z.copy_(torch.from_numpy(self.o.matmul(W.numpy(), P.numpy(), Fo)))

    def pull_gemm(self, g, dz_ext, W, Fi, exact, out):
        t = torch.empty((g.n_rows, dz_ext.shape[1]))
        if exact:
            self.kipf_propagate(g, dz_ext, t)
        else:
            self.neighbour_sum(g, dz_ext, t)
        out.copy_(torch.from_numpy(self.o.matmul_dx(W.numpy(), t.numpy(), Fi)))

    def matmul_dw(self, P, dZ, out):
        out.copy_(torch.from_numpy(self.o.matmul_dw(dZ.numpy(), P.numpy())))

    def matmul(self, W, P, Fo, bias=None, out=None):
        z = self.o.matmul(W.numpy(), P.numpy(), Fo)
        if bias is not None:
            z = self.o.add_bias_rows(z, bias.numpy())
        z = torch.from_numpy(z)
        if out is None:
            return z
        out.copy_(z)
        return out

    def pull_dual(self, g, x, plain, coef):
        self.neighbour_sum(g, x, plain)
        self.kipf_propagate(g, x, coef)

    def matmul_dx(self, W, dZ, Fi, out=None):
        r = torch.from_numpy(self.o.matmul_dx(W.numpy(), dZ.numpy(), Fi))
        if out is None:
            return r
        out.copy_(r)
        return out

    def axpy(self, alpha, x, y):
        y.add_(x, alpha=alpha)
        return y

    def activation(self, kind, z):
        return torch.from_numpy(self.o.activation(kind, z.numpy()))

    def activation_bwd(self, kind, y, g, z=None):
        return torch.from_numpy(self.o.activation_bwd(kind, y.numpy(), g.numpy()))

    # -- graph neural operator on a row block (the MATERIALISING oracle: kappa [E, Fo*Fi] exists here) ----------------------
    def gno_aggregate(self, g, theta, coords, x, d, H, Fo, out=None):
        Fi = x.shape[1]
        kap = self.o.gno_kernel_eval(coords.numpy(), theta.numpy(), H, Fo * Fi)
        ia, ja = g.square()
        m = torch.from_numpy(self.o.gno_aggregate(x.numpy(), kap, ia, ja, Fo)[:g.n_rows])
        if out is None:
            return m
        out.copy_(m)
        return out

    def gno_aggregate_bwd_theta(self, g, theta, coords, x, grad, d, H, s_save=None):
        ia, ja = g.square()
        gp = np.zeros((g.n_cols, grad.shape[1]), np.float32)
        gp[:g.n_rows] = grad.numpy()
        dk = self.o.gno_aggregate_bwd_k(gp, x.numpy(), coords.shape[0], ia, ja)
        return torch.from_numpy(self.o.gno_kernel_bwd_theta(coords.numpy(), theta.numpy(), dk, H))

    def gno_aggregate_bwd(self, g, theta, coords, x, grad, d, H, s_save=None, need_dx=True, need_dtheta=True, need_dcoords=False):
        """the scatter-form reverse pass of a row block: dx of EVERY column its rows touch (n_cols rows), dtheta of its rows"""
        Fi, Fo = x.shape[1], grad.shape[1]
        ia, ja = g.square()
        gp = np.zeros((g.n_cols, Fo), np.float32)
        gp[:g.n_rows] = grad.numpy()
        kap = self.o.gno_kernel_eval(coords.numpy(), theta.numpy(), H, Fo * Fi)
        dx = torch.from_numpy(self.o.gno_aggregate_bwd_x(gp, kap, ia, ja, Fi)) if need_dx else None
        dth = self.gno_aggregate_bwd_theta(g, theta, coords, x, grad, d, H) if need_dtheta else None
        dc = self.gno_aggregate_bwd_coords(g, theta, coords, x, grad, d, H) if need_dcoords else None
        return dx, dth, dc, False

    def gno_aggregate_bwd_coords(self, g, theta, coords, x, grad, d, H):
        ia, ja = g.square()
        gp = np.zeros((g.n_cols, grad.shape[1]), np.float32)
        gp[:g.n_rows] = grad.numpy()
        dk = self.o.gno_aggregate_bwd_k(gp, x.numpy(), coords.shape[0], ia, ja)
        return torch.from_numpy(self.o.gno_kernel_bwd_coords(coords.numpy(), theta.numpy(), dk, H))

    def gno_aggregate_bwd_x_pull(self, g, theta, coords, grad_ext, d, H, Fi, out=None):
        """dx[v] = sum_{w in row v} K_e^T grad_ext[col[w]] in entry order, fp32 (the pull the product path evaluates)"""
        Fo = grad_ext.shape[1]
        kap = self.o.gno_kernel_eval(coords.numpy(), theta.numpy(), H, Fo * Fi).reshape(-1, Fi, Fo)   # K_e[q, o] = kappa[o + Fo q]
        ge = grad_ext.numpy()
        dx = np.zeros((g.n_rows, Fi), np.float32)
        for v in range(g.n_rows):
            for w in range(g.ia[v] - 1, g.ia[v + 1] - 1):
                e = g.ja[1, w] - 1
                if e >= 0:
                    dx[v] += (kap[e] @ ge[g.ja[0, w] - 1]).astype(np.float32)
        dx = torch.from_numpy(dx)
        if out is None:
            return dx
        out.copy_(dx)
        return out

    def gather_rows(self, x, idx, out):
        out.copy_(x[idx.long()])


def _worker(rank, world, port, n, pairs, F, cut, q, Fo=None, order="auto", mode="p2p"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ATHENA_MP_HALO_MODE=mode)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from athena_amd import dist as adist

    dev = torch.device("cpu")
    shard = adist.make_weak_scaling_shard(rank, world, n, pairs, F, cut=cut, device=dev)
    step = adist.KipfShardStep(shard, F, dev, backend=OracleBackend(), Fo=Fo, order=order)
    x_local = step.x_ext[:n].clone().numpy()
    dx = step().clone().numpy()
    q.put((rank, dict(x=x_local, dz=step.dZ.numpy().copy(), w=step.W.numpy().copy(),
                      P=step.P.numpy().copy() if step.P is not None else None, transform_first=step.transform_first,
                      Z=step.Z.numpy().copy(), dW=step.dW.numpy().copy(), dX=dx, n_halo=shard.n_halo,
                      send=int(shard.send_idx.numel()), col_deg=shard.col_deg.copy(), order=shard.order.copy(),
                      n_int=shard.n_int, halo_mode=shard.halo_mode, halo_fraction=shard.halo_fraction)))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("world,cut,Fo,order,mode", [
    (2, None, None, "auto", "p2p"), (2, 0.1, None, "auto", "p2p"), (4, 0.05, None, "auto", "p2p"),
    (8, 0.05, None, "auto", "p2p"),           # the 8-way split of BASELINE configs[4]
    (3, None, None, "auto", "p2p"),
    (2, 0.1, 3, "auto", "p2p"),               # 8 -> 3: dense step before the exchange
    (3, None, 7, "auto", "p2p"),              # 8 -> 7: rectangular, aggregate first
    (2, None, 8, "transform_first", "p2p"),
    # the halo as ONE all-gather of whole blocks (the fallback above a halo fraction of ~0.7)
    (2, None, None, "auto", "allgather"), (3, None, None, "auto", "allgather"), (8, None, None, "auto", "allgather"),
    (2, 0.1, 3, "auto", "allgather"), (3, None, 7, "auto", "allgather"),
    # the rule itself: a uniform graph on 2 ranks needs nearly every remote row -> all-gather; planted partitions -> p2p
    (2, None, None, "auto", "auto"), (4, 0.05, None, "auto", "auto")])
def test_multi_rank_kipf_step_matches_global_oracle(oracle, world, cut, Fo, order, mode):
    from athena_amd import dist as adist

    n, pairs, F = 400, 1500, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, pairs, F, cut, q, Fo, order, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # assemble the global graph from the same generator
    rows, cols = [], []
    for r in range(world):
        rr, cc, _ = adist.shard_entries(r, world, n, pairs, cut)
        rows.append(rr + r * n); cols.append(cc)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    N = world * n
    ia = np.concatenate([[1], 1 + np.cumsum(np.bincount(rows, minlength=N))]).astype(np.int32)
    ja = np.zeros((2, rows.size), np.int32, order="F"); ja[0] = cols + 1
    # the generated graph is symmetric as a multiset (undirected), which the pull-form backward relies on
    A = np.zeros((N, N), np.int64); np.add.at(A, (rows, cols), 1)
    assert np.array_equal(A, A.T)
    # shards number their vertices interior-first: row k of rank r is original local vertex order[k]
    def unperm(key):
        out = []
        for r in range(world):
            a = np.empty_like(res[r][key]); a[res[r]["order"]] = res[r][key]; out.append(a)
        return np.concatenate(out)
    x, dz = unperm("x"), unperm("dz")
    w = res[0]["w"]
    assert all(np.array_equal(w, res[r]["w"]) for r in range(world))
    Fo = F if Fo is None else Fo
    P = oracle.kipf_propagate(x, ia, ja)
    Zo = oracle.matmul(w, P, Fo)
    if res[0]["transform_first"]:
        assert (Fo, order) in ((3, "auto"), (8, "transform_first"))
        assert np.abs(unperm("Z") - Zo).max() <= 1e-5 * np.abs(Zo).max()                   # A (X W^T): re-associated
    else:
        assert np.array_equal(unperm("P"), P)                                              # bit-exact
        assert np.array_equal(unperm("Z"), Zo)
    dP = oracle.matmul_dx(w, dz, F)
    dX = oracle.kipf_propagate_bwd(dP, ia, ja)                                             # reference: no coefficient
    got = unperm("dX")                                                                     # (A^T dZ) W: same map,
    assert np.abs(got - dX).max() <= 1e-5 * np.abs(dX).max()                               # re-associated
    dW = oracle.matmul_dw(dz, P)
    for r in range(world):
        assert np.abs(res[r]["dW"] - dW).max() <= 1e-5 * np.abs(dW).max()                  # all-reduced
        assert res[r]["n_halo"] > 0
        want = mode if mode != "auto" else ("allgather" if res[r]["halo_fraction"] > 0.7 else "p2p")
        assert res[r]["halo_mode"] == want, (res[r]["halo_mode"], res[r]["halo_fraction"])
        assert (res[r]["send"] > 0) == (want == "p2p")                 # the all-gather needs no send list
        assert res[r]["n_halo"] == world * n or want == "p2p"
    if mode == "auto":
        assert res[0]["halo_mode"] == ("allgather" if cut is None else "p2p"), res[0]["halo_fraction"]
    deg = np.diff(ia)
    assert np.array_equal(res[0]["col_deg"][:n], deg[:n][res[0]["order"]])
    assert all(0 <= res[r]["n_int"] < n for r in range(world))
    if cut is not None:
        assert all(res[r]["n_int"] > n // 4 for r in range(world))   # planted partitions leave interior rows


def _worker_global(rank, world, port, n_total, pairs, F, locality, q, mode="p2p"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ATHENA_MP_HALO_MODE=mode)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from athena_amd import dist as adist, synth

    dev = torch.device("cpu")
    shard = adist.make_global_shard(rank, world, n_total, pairs, device=dev, seed=11, locality=locality)
    n = shard.n
    inputs = (synth.feature_block(1, rank * n, (rank + 1) * n, F), synth.feature_block(3, rank * n, (rank + 1) * n, F),
              synth.kipf_weight(F))
    step, nnz, info = adist.build_kipf_step(shard, F, dev, backend=OracleBackend(), inputs=inputs)
    dx = step().clone().numpy()
    held = shard.ext_ids >= 0                      # every row behind the local ones that holds a vertex (not a padding slot)
    halo_ok = bool(np.array_equal(step.x_ext[n:].numpy()[held], synth.feature_rows(1, shard.ext_ids[held], F)) and
                   np.array_equal(step.dZ_ext[n:].numpy()[held], synth.feature_rows(3, shard.ext_ids[held], F)) and
                   np.isin(shard.halo_ids, shard.ext_ids[held]).all())
    q.put((rank, dict(P=step.P.numpy().copy(), Z=step.Z.numpy().copy(), dW=step.dW.numpy().copy(), dX=dx,
                      order=shard.order.copy(), nnz=nnz, halo_ok=halo_ok, n_int=shard.n_int, info=info,
                      halo_mode=shard.halo_mode)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,locality,mode", [(2, None, "p2p"), (8, None, "p2p"), (4, (60, 0.95), "p2p"),
                                                 (8, None, "allgather"), (4, (60, 0.95), "auto"), (2, None, "auto")])
def test_strong_scaling_shards_of_one_fixed_graph_match_the_single_process_oracle(oracle, world, locality, mode):
    """bench.py's N > 1 default: every rank builds its row block of ONE graph (the C2 generator, here 1 600 vertices)
    from the shared pair stream and takes its rows of the SAME X / dZ the single-GPU run uses; assembled results equal
    the oracle on the whole graph, P bit for bit.  8 ranks = the partition of BASELINE configs[4]."""
    from athena_amd import synth

    n_total, pairs, F = 1600, 6000, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_global, args=(r, world, port, n_total, pairs, F, locality, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if locality is None:
        ia, ja = synth.random_graph_csr(n_total, pairs, seed=11)
    else:
        ia64, cols = synth.random_graph_csr_rows(n_total, pairs, 0, n_total, seed=11, locality=locality)
        ia = ia64.astype(np.int32)
        ja = np.zeros((2, cols.size), np.int32, order="F"); ja[0] = cols + 1
    assert sum(res[r]["nnz"] for r in range(world)) == ja.shape[1]
    x, w, dz = synth.feature_block(1, 0, n_total, F), synth.kipf_weight(F), synth.feature_block(3, 0, n_total, F)

    def unperm(key):
        out = []
        for r in range(world):
            a = np.empty_like(res[r][key]); a[res[r]["order"]] = res[r][key]; out.append(a)
        return np.concatenate(out)

    P = oracle.kipf_propagate(x, ia, ja)
    assert np.array_equal(unperm("P"), P)
    assert np.array_equal(unperm("Z"), oracle.matmul(w, P, F))
    dX = oracle.kipf_propagate_bwd(oracle.matmul_dx(w, dz, F), ia, ja)
    assert np.abs(unperm("dX") - dX).max() <= 1e-5 * np.abs(dX).max()
    dW = oracle.matmul_dw(dz, P)
    for r in range(world):
        assert np.abs(res[r]["dW"] - dW).max() <= 1e-5 * np.abs(dW).max()
        assert res[r]["halo_ok"]
        if mode != "auto":
            assert res[r]["halo_mode"] == mode
        else:   # banded graph: a rank needs a sliver of its neighbours' rows -> p2p; uniform on 2 ranks -> all-gather
            assert res[r]["halo_mode"] == ("p2p" if locality is not None else "allgather")
    if locality is not None:
        assert all(res[r]["n_int"] > 0 for r in range(world))       # a banded graph leaves interior rows


def test_shard_generator_balances_entries():
    from athena_amd import dist as adist

    for world in (1, 2, 4, 8):
        for cut in (None, 0.05):
            r, c, used = adist.shard_entries(world - 1, world, 1000, 4500, cut)
            assert abs(r.size - 10000) <= 8
            assert c.min() >= 0 and c.max() < world * 1000
            local = (c >= (world - 1) * 1000)
            if world > 1:
                frac = 1 - (local.sum() - 1000) / (r.size - 1000)
                assert abs(frac - used) < 0.03


# ---- ONE mesh with edge features cut by rows: the graph neural operator ------------------------------------
def gno_problem(n_points, Fi, Fo, d, H, mean_degree=6.0, seed=9):
    """the global problem every rank (and the checking process) derives from the seed: mesh, features, upstream gradient,
    parameters"""
    sys.path.insert(0, ROOT)
    from athena_amd import synth
    ia, ja, coords = synth.radius_graph(n_points, mean_degree, seed, dim=d, order="cells")
    rng = np.random.Generator(np.random.PCG64(seed + 100))
    x = rng.uniform(-1, 1, (n_points, Fi)).astype(np.float32)
    up = rng.uniform(-1, 1, (n_points, Fo)).astype(np.float32)
    theta = (rng.standard_normal(H * d + H + Fo * Fi * H + Fo * Fi) * 0.3).astype(np.float32)
    w = (rng.standard_normal(Fo * Fi) * 0.5).astype(np.float32)
    b = (rng.standard_normal(Fo) * 0.1).astype(np.float32)
    return ia, ja, coords, x, up, theta, w, b


def _worker_gno(rank, world, port, dims, act, mode, q, reverse="pull"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ATHENA_MP_HALO_MODE=mode)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from athena_amd import dist as adist

    n_points, Fi, Fo, d, H = dims
    ia, ja, coords, x, up, theta, w, b = gno_problem(*dims)
    dev = torch.device("cpu")
    shard, c_loc = adist.make_mesh_shard(rank, world, n_points, device=dev, mesh=(ia, ja, coords))
    n = shard.n
    sl = slice(rank * n, (rank + 1) * n)
    step = adist.GnoShardStep(shard, Fi, Fo, d, H, dev, backend=OracleBackend(), inputs=(x[sl], up[sl], theta, w, b, c_loc),
                              activation=act, reverse=reverse, need_coord_grad=True)
    assert step.reverse == reverse
    out = step.forward().clone().numpy()
    dx = step.backward().clone().numpy()
    held = shard.ext_ids >= 0
    halo_ok = bool(np.array_equal(step.x_ext[n:].numpy()[held], x[shard.ext_ids[held]]) and
                   np.isin(shard.halo_ids, shard.ext_ids[held]).all())
    q.put((rank, dict(out=out, dX=dx, grads=step.grad_flat.numpy().copy(), order=shard.order.copy(), n_int=shard.n_int,
                      n_halo=shard.n_halo, halo_mode=shard.halo_mode, halo_fraction=shard.halo_fraction, halo_ok=halo_ok,
                      n_edge_cols=shard.n_edge_cols, coords_ok=bool(np.array_equal(c_loc, coords[shard.edge_ids])),
                      dcoords=step.dcoords.numpy().copy(), edge_ids=shard.edge_ids.copy(),
                      cut_cols=int(sum(a.size for a in shard.edge_share)))))
    dist.barrier()
    dist.destroy_process_group()


def gno_reference(dims, act):
    """the single-process MATERIALISING oracle on the whole mesh (tests/oracle_layers.py composes the layer as
    athena_graph_nop_layer.f90:690-788 does)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_layers as ol
    from athena_amd.graph import graph_type

    n_points, Fi, Fo, d, H = dims
    ia, ja, coords, x, up, theta, w, b = gno_problem(*dims)
    g = graph_type.from_csr(ia, ja, num_edges=coords.shape[0])
    def run():
        outs, tapes = ol.gno_forward([g], [x], [coords], [theta, w, b], Fi, Fo, d, H, True, act)
        dxs, dcs, grads = ol.gno_backward([g], [x], [coords], tapes, [theta, w, b], Fi, Fo, d, H, True, act, [up])
        return outs[0], dxs[0], np.concatenate([np.asarray(a).reshape(-1) for a in grads]), dcs[0]

    out, dx, grads, dcoords = run()
    gno_reference.dcoords = dcoords        # [E, d] of the whole mesh (the callers that check it read it from here)
    gno_reference.hi = ol.f64_lazy(run)    # the same composition on the float64 twin: hi(2) grads, hi(3) dcoords (helpers.assert_close)
    return out, dx, grads.astype(np.float32)


def assert_param_grads(res_r, g_ref, dc_ref, r):
    """a rank's all-reduced parameter gradients and the coordinate gradient of the edge columns it holds: 1e-5 of the whole
    tensor's scale, anchored on the float64 twin (sums over every vertex / entry of the mesh, in an order that depends on the cut)"""
    from helpers import assert_close

    assert_close(res_r["grads"], g_ref, 1e-5, f"rank {r}: [dtheta | dW | db]", f64=gno_reference.hi(2))
    full = dc_ref.copy()
    full[res_r["edge_ids"]] = res_r["dcoords"]
    assert_close(full, dc_ref, 1e-5, f"rank {r}: dcoords", f64=gno_reference.hi(3))


@pytest.mark.parametrize("world,act,mode,reverse", [(2, "none", "p2p", "pull"), (3, "none", "p2p", "pull"), (8, "none", "p2p", "pull"),
                                                    (2, "relu", "p2p", "pull"), (3, "sigmoid", "allgather", "pull"), (8, "none", "allgather", "pull"),
                                                    (4, "none", "auto", "pull"),
                                                    # the scatter-form reverse pass: rows computed for remote vertices travel to their owners
                                                    (2, "none", "p2p", "reduce"), (3, "relu", "p2p", "reduce"), (8, "none", "p2p", "reduce"),
                                                    (3, "none", "allgather", "reduce"), (8, "sigmoid", "allgather", "reduce")])
def test_node_partitioned_gno_layer_matches_the_single_process_oracle(oracle, world, act, mode, reverse):
    """graph_nop_layer on ONE mesh cut by rows (halo exchange of x forward and of dz in reverse, theta
    replicated, d theta / dW / db all-reduced): assembled out and dx, and every rank's all-reduced gradients, equal the
    materialising oracle on the whole mesh; the rank holds exactly the coords its rows reference."""
    dims = (960, 3, 5, 3, 4)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_gno, args=(r, world, port, dims, act, mode, q, reverse)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out_ref, dx_ref, g_ref = gno_reference(dims, act)

    def unperm(key):
        parts = []
        for r in range(world):
            a = np.empty_like(res[r][key]); a[res[r]["order"]] = res[r][key]; parts.append(a)
        return np.concatenate(parts)

    assert np.abs(unperm("out") - out_ref).max() <= 1e-5 * np.abs(out_ref).max()
    assert np.abs(unperm("dX") - dx_ref).max() <= 1e-5 * np.abs(dx_ref).max()
    dc_ref = gno_reference.dcoords
    for r in range(world):
        assert res[r]["halo_ok"] and res[r]["coords_ok"] and res[r]["n_halo"] > 0 and res[r]["n_edge_cols"] > 0
        # the coordinate gradient of every edge column the rank holds -- cut columns completed by the peer's share
        assert res[r]["cut_cols"] > 0
        assert_param_grads(res[r], g_ref, dc_ref, r)
        want = mode if mode != "auto" else ("allgather" if res[r]["halo_fraction"] > 0.7 else "p2p")
        assert res[r]["halo_mode"] == want
    assert sum(res[r]["n_int"] for r in range(world)) > 0 or world == 8      # compact blocks keep interior rows


def test_locality_order_shrinks_the_halo_of_a_mesh_numbered_at_random(oracle):
    """dist.locality_order (reverse Cuthill-McKee on the CSR pattern: no coordinates needed) + dist.permute_csr: the renamed
    graph is the same graph (kipf_propagate of the permuted features is the permuted result, bit for bit; edge ids travel with
    their entries), and contiguous row blocks of it need a fraction of the halo rows of the random numbering."""
    from athena_amd import dist as adist, synth

    n, world = 24000, 8
    ia, ja, coords = synth.radius_graph(n, mean_degree=10.0, seed=6)          # numbered as drawn: no locality at all
    perm = adist.locality_order(ia, ja)
    assert np.array_equal(np.sort(perm), np.arange(n))
    ia2, ja2 = adist.permute_csr(ia, ja, perm)
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (n, 5)).astype(np.float32)
    assert np.array_equal(oracle.kipf_propagate(x[perm], ia2, ja2), oracle.kipf_propagate(x, ia, ja)[perm])
    # the multiset of (edge id) per renamed row is the old row's
    for k in (0, 17, n - 1):
        a = np.sort(ja2[1, ia2[k] - 1:ia2[k + 1] - 1]); b = np.sort(ja[1, ia[perm[k]] - 1:ia[perm[k] + 1] - 1])
        assert np.array_equal(a, b)

    def halo_rows(ia_, ja_):
        blk = n // world
        rows = np.repeat(np.arange(n), np.diff(ia_)); cols = ja_[0].astype(np.int64) - 1
        return sum(np.unique(cols[(rows // blk == r) & (cols // blk != r)]).size for r in range(world))

    before, after = halo_rows(ia, ja), halo_rows(ia2, ja2)
    assert after < 0.25 * before, (before, after)


@pytest.mark.parametrize("tau,locality,want", [("0.001", (60, 0.95), "allgather"), ("0.999", None, "p2p")])
def test_halo_allgather_fraction_moves_the_threshold(oracle, tau, locality, want):
    """ATHENA_MP_HALO_ALLGATHER_FRACTION (the tau of SURVEY.md 8e, read by comm.hip and by the CPU mirror alike): in auto mode a
    banded graph travels as packed rows and a uniform one as whole blocks (test above); with the threshold pushed to either
    end the choice flips, and the results stay the oracle's"""
    from athena_amd import synth

    world, n_total, pairs, F = 2, 1600, 6000, 8
    old = os.environ.get("ATHENA_MP_HALO_ALLGATHER_FRACTION")
    os.environ["ATHENA_MP_HALO_ALLGATHER_FRACTION"] = tau          # spawned ranks inherit it
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker_global, args=(r, world, port, n_total, pairs, F, locality, q, "auto")) for r in range(world)]
        for p in procs:
            p.start()
        res = dict(q.get(timeout=180) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        if old is None:
            os.environ.pop("ATHENA_MP_HALO_ALLGATHER_FRACTION", None)
        else:
            os.environ["ATHENA_MP_HALO_ALLGATHER_FRACTION"] = old
    if locality is None:
        ia, ja = synth.random_graph_csr(n_total, pairs, seed=11)
    else:
        ia64, cols = synth.random_graph_csr_rows(n_total, pairs, 0, n_total, seed=11, locality=locality)
        ia = ia64.astype(np.int32)
        ja = np.zeros((2, cols.size), np.int32, order="F"); ja[0] = cols + 1
    x = synth.feature_block(1, 0, n_total, F)
    out = []
    for r in range(world):
        assert res[r]["halo_mode"] == want and res[r]["halo_ok"]
        a = np.empty_like(res[r]["P"]); a[res[r]["order"]] = res[r]["P"]; out.append(a)
    assert np.array_equal(np.concatenate(out), oracle.kipf_propagate(x, ia, ja))
