"""GPU: the N > 1 row-partitioned Kipf step with the PRODUCT path -- HIP kernels and the C ABI's communicator / shard /
halo exchange (csrc/comm.hip) -- several ranks sharing the one GPU of the box.  RCCL refuses two ranks on one device
("Duplicate GPU detected"), so these runs use comm.hip's host-staged TEST transport (ATHENA_MP_COMM_TRANSPORT=shm):
partition, renumbering, send lists, halo degrees, interior/boundary split, both exchanges and the dW all-reduce are the
product code; only the nccl* calls are replaced.  RCCL itself runs here with one rank (communicator, barrier,
all-reduce) and with one GPU per rank in the driver's multi-GPU bench.  Results are checked against the single-process
oracle on the assembled global graph."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_dist_gloo import ROOT, _free_port

pytestmark = pytest.mark.gpu


def _shm_dirs():
    try:
        return {d for d in os.listdir("/dev/shm") if d.startswith("athena_mp_")}
    except OSError:
        return set()


def _remove_shm_dirs(before):
    """ranks that a watchdog ended on purpose (os._exit / _exit) cannot tidy the test transport's directory: the test does"""
    import shutil
    for d in _shm_dirs() - before:
        shutil.rmtree(os.path.join("/dev/shm", d), ignore_errors=True)


def _worker(rank, world, port, n, pairs, F, cut, q, Fo=None, mode="p2p"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ATHENA_MP_HALO_MODE=mode)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from athena_amd import dist as adist

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    shard = adist.make_weak_scaling_shard(rank, world, n, pairs, F, cut=cut, device=dev)
    step = adist.KipfShardStep(shard, F, dev, Fo=Fo)                # HipBackend
    x_local = step.x_ext[:n].clone().cpu().numpy()
    ev = []
    dx = step(events=ev).clone().cpu().numpy()
    torch.cuda.synchronize()
    q.put((rank, dict(x=x_local, dz=step.dZ.cpu().numpy().copy(), w=step.W.cpu().numpy().copy(),
                      P=step.P.cpu().numpy().copy() if step.P is not None else None, transform_first=step.transform_first,
                      Z=step.Z.cpu().numpy().copy(), dW=step.dW.cpu().numpy().copy(),
                      dX=dx, order=shard.order.copy(), n_int=shard.n_int, n_halo=shard.n_halo,
                      fwd_ms=ev[0][0].elapsed_time(ev[0][1]), transport=shard.transport, halo_mode=shard.halo_mode,
                      halo_fraction=shard.halo_fraction)))
    dist.barrier()
    adist.c_comm_destroy()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,cut,F,Fo,mode", [
    (2, 0.1, 128, None, "p2p"), (3, None, 64, None, "p2p"), (2, 0.05, 24, None, "p2p"),
    (2, 0.1, 128, 32, "p2p"),     # narrowing step: dense step before the exchange
    (3, None, 64, 96, "p2p"),     # widening step: rectangular, aggregate first
    (2, 0.1, 256, None, "p2p"),   # BASELINE configs[4] width (two-kernel route per shard)
    # the halo as one all-gather of whole blocks (comm.hip mode 1)
    (2, None, 128, None, "allgather"), (3, None, 64, None, "allgather"), (8, None, 64, None, "allgather"),
    (2, 0.1, 128, 32, "allgather"), (2, None, 256, None, "allgather"),
    (3, None, 64, None, "auto"), (2, 0.05, 24, None, "auto")])
def test_multi_rank_step_with_hip_backend_matches_global_oracle(dev, oracle, world, cut, F, Fo, mode):
    from athena_amd import dist as adist

    n, pairs = (3000, 12000) if world < 8 else (1000, 4000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, pairs, F, cut, q, Fo, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    Fo = F if Fo is None else Fo
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rows, cols = [], []
    for r in range(world):
        rr, cc, _ = adist.shard_entries(r, world, n, pairs, cut)
        rows.append(rr + r * n); cols.append(cc)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    N = world * n
    ia = np.concatenate([[1], 1 + np.cumsum(np.bincount(rows, minlength=N))]).astype(np.int32)
    ja = np.zeros((2, rows.size), np.int32, order="F"); ja[0] = cols + 1

    def unperm(key):
        out = []
        for r in range(world):
            a = np.empty_like(res[r][key]); a[res[r]["order"]] = res[r][key]; out.append(a)
        return np.concatenate(out)

    x, dz, w = unperm("x"), unperm("dz"), res[0]["w"]
    P = oracle.kipf_propagate(x, ia, ja)
    if res[0]["transform_first"]:
        assert 4 * Fo <= 3 * F
    else:
        assert np.array_equal(unperm("P"), P)                                     # aggregation: bit exact across shards
    Z = oracle.matmul(w, P, Fo)
    assert np.abs(unperm("Z") - Z).max() <= 1e-5 * np.abs(Z).max()
    dX = oracle.kipf_propagate_bwd(oracle.matmul_dx(w, dz, F), ia, ja)
    assert np.abs(unperm("dX") - dX).max() <= 1e-5 * np.abs(dX).max()
    dW = oracle.matmul_dw(dz, P)
    for r in range(world):
        assert np.abs(res[r]["dW"] - dW).max() <= 1e-5 * np.abs(dW).max()
        assert res[r]["n_halo"] > 0 and res[r]["fwd_ms"] > 0
        assert res[r]["transport"].startswith("shm")     # the C-ABI shard / exchange (comm.hip) over its test transport
        want = mode if mode != "auto" else ("allgather" if cut is None else "p2p")   # uniform on 3 ranks: fraction ~0.9
        assert res[r]["halo_mode"] == want, (res[r]["halo_mode"], res[r]["halo_fraction"])


def _dp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from athena_amd import dist as adist, io
    from athena_amd.layers import duvenaud_msgpass_layer_type

    graphs, _ = io.read_all_graphs(os.path.join(ROOT, "tests", "golden", "all_graphs_small.txt"))
    mine, pos = adist.shard_graphs(graphs, rank, world)
    layer = duvenaud_msgpass_layer_type(num_vertex_features=[6], num_edge_features=[1], num_time_steps=2,
                                        max_vertex_degree=10, num_outputs=10, min_vertex_degree=1, seed=5)   # same seed: same weights
    layer.set_graph(mine)
    _, _, _, x, e = io.batch_graphs(mine)
    out = layer.forward(x, e)
    up = np.arange(len(graphs) * 10, dtype=np.float32).reshape(len(graphs), 10) / 40.0 - 0.5
    layer.backward(up[pos])
    adist.allreduce_layer_gradients([layer])
    full = adist.gather_graph_outputs(out, pos, len(graphs))
    q.put((rank, layer.get_gradients(), full.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_graph_sharded_duvenaud_layer_all_reduce_equals_single_process(dev):
    """independent graphs are the sharding unit (no halo): two ranks take alternate graphs of the batch, the
    parameter gradients are all-reduced in one bucket and the per-graph readout gathered -- equal to the
    single-process pass over the whole batch"""
    from athena_amd import io
    from athena_amd.layers import duvenaud_msgpass_layer_type

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: (g, o) for r, g, o in (q.get(timeout=300) for _ in range(world))}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    graphs, _ = io.read_all_graphs(os.path.join(ROOT, "tests", "golden", "all_graphs_small.txt"))
    layer = duvenaud_msgpass_layer_type(num_vertex_features=[6], num_edge_features=[1], num_time_steps=2,
                                        max_vertex_degree=10, num_outputs=10, min_vertex_degree=1, seed=5)
    layer.set_graph(graphs)
    _, _, _, x, e = io.batch_graphs(graphs)
    out = layer.forward(x, e).cpu().numpy()
    up = np.arange(len(graphs) * 10, dtype=np.float32).reshape(len(graphs), 10) / 40.0 - 0.5
    layer.backward(up)
    ref = layer.get_gradients()
    for r in range(world):
        assert np.abs(res[r][0] - ref).max() <= 2e-6 * np.abs(ref).max()        # same sums, different association
        assert np.array_equal(res[r][1], out)                                    # per-graph readout: bit exact


# ---- the C-ABI communicator / shard / halo exchange (csrc/comm.hip) ---------------------------------------------------
def _xcheck_worker(rank, world, port, q, mode):
    """both plans on the same rows: the C ABI's shard (shm test transport) and the python mirror (torch p2p over gloo)"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ATHENA_MP_HALO_MODE=mode)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from athena_amd import dist as adist

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n, pairs = 2000, 9000
    rows, cols, _ = adist.shard_entries(rank, world, n, pairs, 0.2)
    py = adist.build_plan(adist.Shard(rank, world, n, rows, cols), dev)
    ia = np.concatenate([[1], 1 + np.cumsum(np.bincount(rows, minlength=n))])
    cs = adist.CShard(adist.c_comm(dev), ia, cols)
    same = dict(
        dims=(cs.n, cs.n_int, cs.n_halo, cs.nnz) == (py.n, py.n_int, py.n_halo, py.nnz),
        order=bool(np.array_equal(cs.order, py.order)), halo_ids=bool(np.array_equal(cs.halo_ids, py.halo_ids)),
        ext_ids=bool(np.array_equal(cs.ext_ids, py.ext_ids)), mode=cs.halo_mode == py.halo_mode == mode,
        fraction=abs(cs.halo_fraction - py.halo_fraction) < 1e-12, recv_rows=cs.recv_rows == py.recv_rows,
        col_deg=bool(np.array_equal(cs.col_deg, py.col_deg)),
        send_idx=bool(np.array_equal(cs.send_idx.numpy(), py.send_idx.cpu().numpy())),
        fwd=all(np.array_equal(a, b) for a, b in zip(cs.csr(False), py.csr(False))),
        bwd=all(np.array_equal(a, b) for a, b in zip(cs.csr(True), py.csr(True))), transport=cs.transport)
    # one exchange through each transport delivers the same halo rows
    x = torch.arange((n + cs.n_halo) * 8, dtype=torch.float32, device=dev).reshape(-1, 8) + 1000.0 * rank
    x2 = x.clone()
    x[n:] = -1.0; x2[n:] = -1.0
    cs.exchange(8, dev)(x)
    py.exchange(8, dev, adist.HipBackend(dev))(x2)
    torch.cuda.synchronize()
    held = torch.from_numpy(cs.ext_ids >= 0).to(dev)
    same["halo_rows"] = bool(torch.equal(x[n:][held], x2[n:][held])) and bool((x[n:][held] >= 0).all())
    q.put((rank, same))
    dist.barrier()
    cs.close()
    adist.c_comm_destroy()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["p2p", "allgather"])
def test_c_abi_shard_equals_the_python_plan_and_moves_the_same_halo_rows(dev, mode):
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_xcheck_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in range(world):
        assert res[r].pop("transport").startswith("shm")
        assert all(res[r].values()), (r, res[r])


def test_rccl_communicator_single_rank_through_the_c_abi(dev):
    """what a 1-GPU box can run of the RCCL transport itself: librccl is loaded, ncclGetUniqueId / ncclCommInitRank build
    a 1-rank communicator, barrier and all-reduce go through, a 1-rank shard has no halo and its exchange is a no-op"""
    import ctypes as C
    from athena_amd import _capi, synth
    from athena_amd.dist import CShard

    assert os.environ.get("ATHENA_MP_COMM_TRANSPORT") != "shm"
    idb = C.create_string_buffer(128)
    _capi.call("athena_mp_comm_unique_id", idb)
    assert any(idb.raw)
    h = C.c_void_p()
    _capi.call("athena_mp_comm_create", 0, 1, idb, C.byref(h))
    name = C.create_string_buffer(64)
    r, w = C.c_int32(-1), C.c_int32(-1)
    _capi.call("athena_mp_comm_info", h, C.byref(r), C.byref(w), name, 64)
    assert (r.value, w.value, name.value) == (0, 1, b"rccl")
    with open("/proc/self/maps") as fh:
        assert "librccl" in fh.read()
    _capi.call("athena_mp_comm_barrier", h)
    t = torch.arange(16, dtype=torch.float32, device=dev)
    _capi.call("athena_mp_allreduce", h, C.c_void_p(t.data_ptr()), 16)
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(16, dtype=torch.float32))
    comm = type("C", (), {})(); comm.handle, comm.rank, comm.world, comm.transport = h, 0, 1, "rccl"
    ia, ja = synth.random_graph_csr(500, 2000, seed=3)
    sh = CShard(comm, ia, ja[0].astype(np.int64) - 1)
    assert (sh.n, sh.n_int, sh.n_halo) == (500, 500, 0) and np.array_equal(sh.order, np.arange(500))
    x = torch.rand((500, 16), device=dev)
    x0 = x.clone()
    sh.exchange(16, dev)(x)
    assert torch.equal(x, x0)
    sh.close()
    _capi.call("athena_mp_comm_destroy", h)


@pytest.mark.parametrize("world,transport,mode,nv", [(1, "rccl", "auto", 6000), (2, "shm", "p2p", 6000), (3, "shm", "p2p", 6000),
                                                     (3, "shm", "allgather", 6001),    # blocks of 2000 / 2000 / 2001 rows: padded slots
                                                     (8, "shm", "allgather", 6001), (2, "shm", "auto", 6000)])
def test_fortran_processes_run_the_sharded_step_through_the_c_abi(dev, oracle, tmp_path, world, transport, mode, nv):
    """kipf_shard_run.f90: one FORTRAN process per rank -- communicator from an id file, shard, halo exchange of X and
    dZ under the interior rows, dW all-reduce, all through ISO_C_BINDING; assembled results against the oracle on the
    whole graph (P bit for bit).  world = 1 uses RCCL itself; 2 and 3 ranks share the box's one GPU over the test
    transport (RCCL refuses two ranks on one device)."""
    import subprocess

    exe = os.path.join(ROOT, "athena_amd", "fortran", "kipf_shard_run")
    if not os.path.exists(exe):
        pytest.skip("Fortran driver not built (no amdflang)")
    pairs, F = 24000, 64
    env = dict(os.environ)
    env.pop("ATHENA_MP_COMM_TRANSPORT", None)
    env["ATHENA_MP_HALO_MODE"] = mode
    # a file a crashed launch left behind at the rendezvous path (and a stale hello) must not be taken for this launch's
    with open(tmp_path / "id", "wb") as fh:
        fh.write(b"\x5a" * (128 + 8 * world))
    with open(str(tmp_path / "id") + ".hello.1", "wb") as fh:
        fh.write(b"\x11" * 8)
    if transport == "shm":
        env["ATHENA_MP_COMM_TRANSPORT"] = "shm"
    prefix = str(tmp_path / "run")
    procs = [subprocess.Popen([exe, str(r), str(world), "0", str(tmp_path / "id"), str(nv), str(pairs), str(F), prefix],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    with open(prefix + "_problem.bin", "rb") as fh:
        N, nnz, Fr = np.fromfile(fh, np.int32, 3)
        ia = np.fromfile(fh, np.int32, N + 1)
        ja = np.fromfile(fh, np.int32, 2 * nnz).reshape(2, nnz, order="F")
        x = np.fromfile(fh, np.float32, N * F).reshape(N, F)
        dz = np.fromfile(fh, np.float32, N * F).reshape(N, F)
        w = np.fromfile(fh, np.float32, F * F)
    assert (N, nnz, Fr) == (nv, 2 * pairs + nv, F)
    P, Z, dX, dWs, halos = [], [], [], [], []
    for r in range(world):
        with open(f"{prefix}_r{r}.bin", "rb") as fh:
            n, f, n_int, n_halo = np.fromfile(fh, np.int32, 4)
            P.append(np.fromfile(fh, np.float32, n * f).reshape(n, f)); Z.append(np.fromfile(fh, np.float32, n * f).reshape(n, f))
            dX.append(np.fromfile(fh, np.float32, n * f).reshape(n, f)); dWs.append(np.fromfile(fh, np.float32, f * f))
            halos.append(n_halo)
    P, Z, dX = np.concatenate(P), np.concatenate(Z), np.concatenate(dX)
    assert all(h > 0 for h in halos) or world == 1
    p_ref = oracle.kipf_propagate(x, ia, ja)
    assert np.array_equal(P, p_ref)
    z_ref = oracle.matmul(w, p_ref, F)
    assert np.abs(Z - z_ref).max() <= 1e-5 * np.abs(z_ref).max()
    dx_ref = oracle.kipf_propagate_bwd(oracle.matmul_dx(w, dz, F), ia, ja)
    assert np.abs(dX - dx_ref).max() <= 1e-5 * np.abs(dx_ref).max()
    dw_ref = oracle.matmul_dw(dz, p_ref)
    for r in range(world):
        assert np.abs(dWs[r] - dw_ref).max() <= 1e-5 * np.abs(dw_ref).max()


@pytest.mark.parametrize("argv,mode", [(["--gpus", "2"], "auto"),                        # the command the driver issues, C2 graph
                                       (["--gpus", "8"], "auto"),                        # ... at N = 8 (halo fraction 0.68: packed rows)
                                       (["--gpus", "8", "--nodes", "200000", "--pairs", "900000"], "auto"),
                                       (["--gpus", "4", "--nodes", "200000", "--pairs", "900000"], "p2p"),
                                       (["--gpus", "2", "--config", "c5-local", "--nodes", "400000", "--pairs", "2800000"], "auto")])
def test_bench_py_multi_gpu_command_dry_run_on_one_device(dev, argv, mode):
    """`python bench.py --gpus N` -- what the driver launches on an 8-GPU node -- through the one-device dry run
    (ATHENA_MP_BENCH_ONE_DEVICE=1, gloo process group, comm.hip's host-staged test transport): self-launch, strong-scaling
    shards of the metric's own graph, both exchanges, parity of every rank against the oracle, the breakdown.  The
    numbers mean nothing here; the line's structure and its parity do."""
    import json
    import subprocess

    env = dict(os.environ, ATHENA_MP_BENCH_ONE_DEVICE="1", ATHENA_MP_BENCH_BACKEND="gloo", ATHENA_MP_HALO_MODE=mode)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1"] + argv,
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    n = int(argv[1])
    assert line["n_gpus"] == n and line["scaling"] == "strong" and line["unit"] == "edges/s"
    assert line["parity"]["ok"] and line["parity"]["P_bit_exact"] and line["parity"]["halo_rows_bit_exact"]
    assert max(line["parity"]["Z_rel"], line["parity"]["dX_rel"], line["parity"]["dW_rel_vs_float64"]) <= 1e-5
    bd = line["breakdown"]
    for k in ("halo_ms", "interior_ms", "boundary_ms", "dw_ms", "halo_recv_bytes_per_gpu_per_step", "xgmi_recv_GBps_per_gpu",
              "halo_mode", "halo_fraction", "halo_allgather_threshold"):
        assert k in bd, k
    cfg = line["config"]
    assert cfg["transport"].startswith("shm") and cfg["halo_mode"] == bd["halo_mode"]
    if mode == "auto":   # the rule of SURVEY.md 8e: whole blocks above a halo fraction of 0.7
        assert bd["halo_mode"] == ("allgather" if bd["halo_fraction"] > bd["halo_allgather_threshold"] else "p2p")
        if "c5-local" in argv:
            assert bd["halo_mode"] == "p2p"                 # a banded graph needs a sliver of its neighbours' blocks
        elif n == 2:
            assert bd["halo_mode"] == "allgather"           # uniform graph, 2 ranks: ~0.99 of the other block
        elif argv == ["--gpus", "8"]:                       # configs[1] at N = 8: 1 - exp(-1.125) = 0.675 of the remote rows
            assert 0.6 < bd["halo_fraction"] < 0.7 and bd["halo_mode"] == "p2p"
    else:
        assert bd["halo_mode"] == mode
    assert line["roofline"]["frac"] > 0 and line["roofline"]["dense"]["bound"] == "mfma"


def _c5_shard_worker(rank, world, port, variant, check_rank, q):
    """one rank of the 8-way partition of BASELINE configs[4]: builds its row block from the shared pair stream and its shard
    through athena_mp_shard_create (all ranks: it is collective); `check_rank` then runs the layer step's launches on the
    shard's own graph handles and holds sampled rows against the oracle"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ATHENA_MP_HALO_MODE="auto")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from athena_amd import dist as adist, synth

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    N, pairs, F = 10_000_000, 70_000_000, 256
    locality = (50_000, 0.95) if variant == "local" else None
    shard = adist.make_global_shard(rank, world, N, pairs, device=dev, locality=locality)
    assert isinstance(shard, adist.CShard), shard.transport
    out = dict(mode=shard.halo_mode, fraction=shard.halo_fraction, n_int=shard.n_int, n_halo=shard.n_halo)
    if rank == check_rank:
        from oracle import oracle
        n = shard.n
        g_fi, g_fb, g_bi, g_bb = shard.graphs()
        ni = shard.n_int
        gen = torch.Generator(device=dev).manual_seed(1)
        xg = torch.rand((N, F), device=dev, generator=gen).mul_(2.0).sub_(1.0)
        ids = torch.from_numpy(np.concatenate([rank * n + shard.order, np.maximum(shard.ext_ids, 0)])).to(dev)
        x_ext = xg[ids]                                    # what the exchange delivers: the owners' rows
        del xg
        w_h = synth.kipf_weight(F)
        w = torch.from_numpy(w_h).to(dev)
        b = adist.HipBackend(dev)
        P = torch.empty((n, F), device=dev); Z = torch.empty((n, F), device=dev); dX = torch.empty((n, F), device=dev)
        b.kipf_layer_fwd(g_fi, x_ext, w, F, P=P[:ni], Z=Z[:ni])
        b.kipf_layer_fwd(g_fb, x_ext, w, F, P=P[ni:], Z=Z[ni:])
        b.pull_gemm(g_bi, x_ext, w, F, exact=False, out=dX[:ni])          # x_ext doubles as the exchanged dZ
        b.pull_gemm(g_bb, x_ext, w, F, exact=False, out=dX[ni:])
        rng = np.random.default_rng(40 + rank)
        pick = [rng.choice(ni, min(2000, ni), replace=False)] if ni else []
        if n - ni:
            pick.append(ni + rng.choice(n - ni, min(2000, n - ni), replace=False))
        rows = np.unique(np.concatenate(pick + [np.arange(max(n - 16, 0), n)]))
        rsel = torch.from_numpy(rows).to(dev)

        def sub(backward):
            ia, nb = shard.csr(backward)
            ent = np.concatenate([np.arange(ia[r] - 1, ia[r + 1] - 1) for r in rows])
            cols, inv = np.unique(nb[ent].astype(np.int64) - 1, return_inverse=True)
            sia = np.concatenate([[1], 1 + np.cumsum(ia[rows + 1] - ia[rows])]).astype(np.int32)
            sja = np.zeros((2, ent.size), np.int32, order="F"); sja[0] = inv + 1
            return cols, sia, sja

        cols, sia, sja = sub(False)
        xc = x_ext[torch.from_numpy(cols).to(dev)].cpu().numpy()
        p_ref = oracle.kipf_propagate_rect(xc, sia, sja, shard.row_deg[rows], shard.col_deg[cols])
        out["P_bit_exact"] = bool(np.array_equal(P[rsel].cpu().numpy(), p_ref))
        z_ref = oracle.matmul(w_h, p_ref, F)
        out["Z_rel"] = float(np.abs(Z[rsel].cpu().numpy() - z_ref).max() / np.abs(z_ref).max())
        cols, sia, sja = sub(True)
        dzc = x_ext[torch.from_numpy(cols).to(dev)].cpu().numpy()
        ones = np.ones(max(rows.size, cols.size), np.int32)
        dx_ref = oracle.kipf_propagate_rect(oracle.matmul_dx(w_h, dzc, F), sia, sja, ones[:rows.size], ones[:cols.size])
        out["dX_rel"] = float(np.abs(dX[rsel].cpu().numpy() - dx_ref).max() / np.abs(dx_ref).max())
        # the degrees of the halo columns came from their owners: equal to the generator's
        ia_all, _ = None, None
        held = shard.ext_ids >= 0
        out["halo_ids_in_layout"] = bool(np.isin(shard.halo_ids, shard.ext_ids[held]).all())
        torch.cuda.synchronize()
    q.put((rank, out))
    dist.barrier()
    shard.close()
    adist.c_comm_destroy()
    dist.destroy_process_group()


@pytest.mark.parametrize("variant,check_rank,want_mode", [("uniform", 5, "allgather"), ("local", 3, "p2p")])
def test_c5_shard_from_shard_create_with_eight_ranks(dev, variant, check_rank, want_mode):
    """BASELINE configs[4] (10 M / 150 M / 256) partitioned 8 ways by athena_mp_shard_create ITSELF: eight processes on
    the one GPU (shm test transport for the metadata collectives; no feature rows are exchanged -- the checked rank fills
    its halo rows with what the exchange would deliver), every rank builds its real 1.25 M-row shard, the halo rule picks
    the whole-block all-gather on the uniform graph (fraction 0.80) and packed rows on the locality variant, and one rank
    runs the interior + boundary launches of the forward and of the reverse pull on the shard's own handles: P bit for
    bit, Z and dX <= 1e-5 on sampled rows of both blocks against the oracle."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_c5_shard_worker, args=(r, world, port, variant, check_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=1500) for _ in range(world))
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for r in range(world):
        assert res[r]["mode"] == want_mode, (r, res[r])
        assert (res[r]["fraction"] > 0.7) == (want_mode == "allgather")
    c = res[check_rank]
    assert c["P_bit_exact"] and c["halo_ids_in_layout"]
    assert c["Z_rel"] <= 1e-5 and c["dX_rel"] <= 1e-5, c
    if variant == "local":
        assert all(res[r]["n_int"] > 0 for r in range(world))


# ---- ONE mesh with edge features cut by rows: graph_nop_layer through the C-ABI shard (athena_mp_shard_create_edges) ----------
def _gno_worker(rank, world, port, dims, act, mode, q, reverse="pull"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ATHENA_MP_HALO_MODE=mode)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from athena_amd import dist as adist
    from test_dist_gloo import gno_problem

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n_points, Fi, Fo, d, H = dims
    ia, ja, coords, x, up, theta, w, b = gno_problem(*dims)
    shard, c_loc = adist.make_mesh_shard(rank, world, n_points, device=dev, mesh=(ia, ja, coords))
    assert isinstance(shard, adist.CShard), shard.transport
    n = shard.n
    sl = slice(rank * n, (rank + 1) * n)
    # the python plan on the same rows: both must number the edge columns alike
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(ia[rank * n:(rank + 1) * n + 1]))
    e0, e1 = int(ia[rank * n]) - 1, int(ia[(rank + 1) * n]) - 1
    py = adist.Shard(rank, world, n, rows, ja[0, e0:e1].astype(np.int64) - 1, ja[1, e0:e1].astype(np.int64))
    gi, gb = shard.graphs()[0:2]
    eid_c = np.concatenate([gi.export("eid"), gb.export("eid")]) + 1
    same_plan = bool(np.array_equal(shard.edge_ids, py.edge_ids) and np.array_equal(shard.order, py.order)
                     and np.array_equal(eid_c, py.adj_ja[1]))
    step = adist.GnoShardStep(shard, Fi, Fo, d, H, dev, inputs=(x[sl], up[sl], theta, w, b, c_loc), activation=act, reverse=reverse,
                              need_coord_grad=True)
    assert step.reverse == reverse
    out = step.forward().clone().cpu().numpy()
    dx = step.backward().clone().cpu().numpy()
    torch.cuda.synchronize()
    held = shard.ext_ids >= 0
    halo_ok = bool(np.array_equal(step.x_ext[n:].cpu().numpy()[held], x[shard.ext_ids[held]]))
    q.put((rank, dict(out=out, dX=dx, grads=step.grad_flat.cpu().numpy().copy(), order=shard.order.copy(), n_int=shard.n_int,
                      n_halo=shard.n_halo, halo_mode=shard.halo_mode, halo_ok=halo_ok, same_plan=same_plan,
                      kept_s=[t is not None for t in step._s], transport=shard.transport, n_edge_cols=shard.n_edge_cols,
                      dcoords=step.dcoords.cpu().numpy().copy(), edge_ids=shard.edge_ids.copy(),
                      cut_same=bool(np.array_equal(shard._export(8, np.int32), np.concatenate(py.edge_share + [np.zeros(0, np.int64)])))
                      and bool(np.array_equal(shard._export(9, np.int64), np.concatenate([[0], np.cumsum([a.size for a in py.edge_share])]))))))
    dist.barrier()
    shard.close()
    adist.c_comm_destroy()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,dims,act,mode,reverse", [
    (2, (1536, 64, 64, 3, 64), "none", "p2p", "pull"),        # BASELINE configs[3]'s widths: the producer / consumer kernels, S kept
    (3, (1536, 64, 64, 3, 64), "relu", "p2p", "pull"),
    (2, (1536, 64, 64, 3, 64), "none", "allgather", "pull"),
    (8, (2048, 64, 64, 3, 64), "none", "auto", "pull"),
    (2, (960, 3, 5, 3, 4), "sigmoid", "p2p", "pull"),         # generic shapes: the VALU / tiled route
    (3, (960, 32, 32, 2, 32), "none", "p2p", "pull"),
    # the scatter-form reverse pass: dx and dtheta of each block from ONE contraction, remote rows reduced at their owners
    # (athena_mp_halo_reduce_*), no exchange of dz
    (2, (1536, 64, 64, 3, 64), "none", "p2p", "reduce"),
    (3, (1536, 64, 64, 3, 64), "relu", "p2p", "reduce"),
    (3, (1536, 64, 64, 3, 64), "none", "allgather", "reduce"),
    (8, (2048, 64, 64, 3, 64), "sigmoid", "auto", "reduce"),
    (3, (960, 32, 32, 2, 32), "none", "p2p", "reduce")])      # generic widths: the separate entry points behind the same call
def test_node_partitioned_gno_layer_with_hip_backend_matches_the_oracle(dev, world, dims, act, mode, reverse):
    """graph_nop_layer forward + reverse on ONE mesh cut by rows, the product path: athena_mp_shard_create_edges (edge
    columns renumbered per rank, symmetry checked across ranks), halo exchange of x and of dz through comm.hip (shm test
    transport: the ranks share the box's one GPU), athena_mp_gno_aggregate_fwd / _bwd_theta on the forward blocks,
    athena_mp_gno_aggregate_bwd_x_pull on the backward blocks, ONE all-reduce of [dtheta | dW | db].  Assembled results
    against the materialising oracle on the whole mesh (1e-5; the parameter gradients float64-anchored)."""
    from test_dist_gloo import assert_param_grads, gno_reference

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gno_worker, args=(r, world, port, dims, act, mode, q, reverse)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
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
        assert res[r]["halo_ok"] and res[r]["same_plan"] and res[r]["transport"].startswith("shm")
        assert res[r]["cut_same"]                      # the C shard and the python plan cut the same edge columns per peer
        assert_param_grads(res[r], g_ref, dc_ref, r)   # dcoords through athena_mp_shard_edge_reduce
        assert res[r]["n_halo"] > 0 and res[r]["n_edge_cols"] > 0
        if mode != "auto":
            assert res[r]["halo_mode"] == mode
    if dims[1:] == (64, 64, 3, 64):
        assert any(any(res[r]["kept_s"]) for r in range(world))     # the training-mode forward kept S on the blocks


def _asym_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from athena_amd import dist as adist
    from athena_amd._capi import AthenaMPError

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    # 4 vertices, 2 per rank; pair (1, 3) carries edge column 1 in row 1 but column 2 in row 3: not a shared column
    ia = np.array([1, 2, 2], np.int32) if rank == 0 else np.array([1, 2, 2], np.int32)
    cols = np.array([2], np.int64) if rank == 0 else np.array([0], np.int64)
    eids = np.array([1], np.int64) if rank == 0 else np.array([2], np.int64)
    try:
        adist.CShard(adist.c_comm(dev), ia, cols, eids)
        msg = "no error"
    except AthenaMPError as exc:
        msg = str(exc)
    q.put((rank, msg))
    dist.barrier()
    adist.c_comm_destroy()
    dist.destroy_process_group()


def test_shard_create_edges_refuses_a_graph_whose_directions_do_not_share_an_edge_column(dev):
    """the pull form of the reverse pass equals the reference's scatter only on an undirected graph whose two directions
    share one edge column (athena_diffstruc_extd_sub_nop.f90:369-376): athena_mp_shard_create_edges checks it over all
    ranks and every rank gets the error"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_asym_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert "shared edge columns" in res[0] and "shared edge columns" in res[1], res
    assert all("no error" not in res[r] for r in range(world)), res


@pytest.mark.parametrize("world,transport,mode,dims,reverse", [
    (1, "rccl", "auto", (1536, 64, 64, 3, 64), "pull"), (2, "shm", "p2p", (1536, 64, 64, 3, 64), "pull"),
    (3, "shm", "allgather", (1537, 64, 64, 3, 64), "pull"),   # blocks of 512 / 512 / 513
    (3, "shm", "p2p", (961, 3, 5, 3, 4), "pull"),
    # the scatter-form reverse pass from Fortran: athena_mp_gno_aggregate_bwd per block + athena_mp_halo_reduce_*
    (1, "rccl", "auto", (1536, 64, 64, 3, 64), "reduce"), (2, "shm", "p2p", (1536, 64, 64, 3, 64), "reduce"),
    (3, "shm", "allgather", (1537, 64, 64, 3, 64), "reduce"), (3, "shm", "p2p", (1537, 64, 64, 3, 64), "reduce")])
def test_fortran_processes_run_the_node_partitioned_gno_layer_through_the_c_abi(dev, tmp_path, world, transport, mode, dims, reverse):
    """gno_shard_run.f90: one FORTRAN process per rank -- communicator from an id file, athena_mp_shard_create_edges on the
    rank's rows of graph_type%adj_ia / adj_ja (global vertex AND edge ids), the rank's own edge geometry, halo exchange of
    x and of dz under the interior rows, dtheta / dW / db in one all-reduce, all through ISO_C_BINDING; assembled results
    against the materialising oracle on the whole mesh.  world = 1 uses RCCL itself."""
    import subprocess

    from test_dist_gloo import gno_problem, gno_reference

    exe = os.path.join(ROOT, "athena_amd", "fortran", "gno_shard_run")
    if not os.path.exists(exe):
        pytest.skip("Fortran driver not built (no amdflang)")
    n_points, Fi, Fo, d, H = dims
    ia, ja, coords, x, up, theta, w, b = gno_problem(*dims)
    with open(tmp_path / "problem.bin", "wb") as fh:
        np.array([n_points, ja.shape[1], coords.shape[0], Fi, Fo, d, H], np.int32).tofile(fh)
        ia.astype(np.int32).tofile(fh)
        np.asfortranarray(ja, np.int32).T.copy().tofile(fh)        # column-major (2, nnz)
        for a in (coords, x, up, theta, w, b):
            np.ascontiguousarray(a, np.float32).tofile(fh)
    env = dict(os.environ)
    env.pop("ATHENA_MP_COMM_TRANSPORT", None)
    env["ATHENA_MP_HALO_MODE"] = mode
    if transport == "shm":
        env["ATHENA_MP_COMM_TRANSPORT"] = "shm"
    prefix = str(tmp_path / "run")
    procs = [subprocess.Popen([exe, str(r), str(world), "0", str(tmp_path / "id"), str(tmp_path / "problem.bin"), prefix, reverse],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    out, dx, grads, halos = [], [], [], []
    ng = H * d + H + Fo * Fi * H + Fo * Fi + Fo * Fi + Fo
    for r in range(world):
        with open(f"{prefix}_r{r}.bin", "rb") as fh:
            n, n_int, n_halo, n_ec = np.fromfile(fh, np.int32, 4)
            out.append(np.fromfile(fh, np.float32, n * Fo).reshape(n, Fo)); dx.append(np.fromfile(fh, np.float32, n * Fi).reshape(n, Fi))
            grads.append(np.fromfile(fh, np.float32, ng))
            halos.append(n_halo)
            assert n_ec > 0
    out_ref, dx_ref, g_ref = gno_reference(dims, "none")
    assert all(h > 0 for h in halos) or world == 1
    assert np.abs(np.concatenate(out) - out_ref).max() <= 1e-5 * np.abs(out_ref).max()
    assert np.abs(np.concatenate(dx) - dx_ref).max() <= 1e-5 * np.abs(dx_ref).max()
    from helpers import assert_close
    for r in range(world):
        assert_close(grads[r], g_ref, 1e-5, f"rank {r}: [dtheta | dW | db]", f64=gno_reference.hi(2))
    if dims[1:] == (64, 64, 3, 64):
        assert all("S kept T" in o for o in outs), outs


def _c4_shard_worker(rank, world, port, mesh_dir, check_rank, q):
    """one rank of the 8-way row partition of BASELINE configs[3]'s mesh (points in Morton order): builds its shard through
    athena_mp_shard_create_edges (collective); `check_rank` then runs the forward blocks and the reverse pull on the
    shard's own handles with the halo rows filled with what the exchange delivers, and holds sampled rows of both blocks
    against the materialising oracle"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ATHENA_MP_HALO_MODE="auto")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from athena_amd import dist as adist, ops

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ia = np.load(os.path.join(mesh_dir, "ia.npy"), mmap_mode="r")
    ja = np.load(os.path.join(mesh_dir, "ja.npy"), mmap_mode="r")
    coords = np.load(os.path.join(mesh_dir, "coords.npy"), mmap_mode="r")
    N = ia.size - 1
    shard, c_loc = adist.make_mesh_shard(rank, world, N, device=dev, mesh=(ia, ja, coords))
    assert isinstance(shard, adist.CShard), shard.transport
    n, ni = shard.n, shard.n_int
    out = dict(mode=shard.halo_mode, fraction=shard.halo_fraction, n=n, n_int=ni, n_halo=shard.n_halo, n_edge_cols=shard.n_edge_cols,
               nnz=shard.nnz, recv_rows=shard.recv_rows)
    if rank == check_rank:
        from oracle import oracle
        Fi = Fo = H = 64; d = 3
        rng = np.random.default_rng(7)
        gen = torch.Generator(device=dev).manual_seed(1)
        xg = torch.rand((N, Fi), device=dev, generator=gen).mul_(2.0).sub_(1.0)
        ids = torch.from_numpy(np.concatenate([rank * n + shard.order, np.maximum(shard.ext_ids, 0)])).to(dev)
        x_ext = xg[ids]                                    # local rows interior-first | what the exchange delivers
        del xg
        theta_h = (0.3 * rng.standard_normal(H * d + H + Fo * Fi * H + Fo * Fi)).astype(np.float32)
        theta = torch.from_numpy(theta_h).to(dev)
        co = torch.from_numpy(c_loc).to(dev)
        g_fi, g_fb, g_bi, g_bb = shard.graphs()
        m = torch.empty((n, Fo), device=dev)
        s_int = s_bnd = None
        if ni:
            _, s_int = ops.gno_aggregate_save(g_fi, theta, co, x_ext, d, H, Fo, out=m[:ni])
        if n - ni:
            _, s_bnd = ops.gno_aggregate_save(g_fb, theta, co, x_ext, d, H, Fo, out=m[ni:])
        dx = torch.empty((n, Fi), device=dev)
        if ni:
            ops.gno_aggregate_bwd_x_pull(g_bi, theta, co, x_ext, d, H, Fi, out=dx[:ni])      # x_ext doubles as the exchanged dz
        if n - ni:
            ops.gno_aggregate_bwd_x_pull(g_bb, theta, co, x_ext, d, H, Fi, out=dx[ni:])
        pick = [rng.choice(ni, min(150, ni), replace=False)] if ni else []
        if n - ni:
            pick.append(ni + rng.choice(n - ni, min(150, n - ni), replace=False))
        rows = np.unique(np.concatenate(pick + [np.arange(max(n - 8, 0), n)]))
        rsel = torch.from_numpy(rows).to(dev)

        def sub(backward):
            gi, gb = shard.graphs()[2:4] if backward else shard.graphs()[0:2]
            rp = np.concatenate([gi.export("rowptr"), gb.export("rowptr")[1:] + gi.nnz]).astype(np.int64)
            col = np.concatenate([gi.export("col"), gb.export("col")])
            eid = np.concatenate([gi.export("eid"), gb.export("eid")])
            ent = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in rows])
            cols, cinv = np.unique(col[ent], return_inverse=True)
            ecols, einv = np.unique(eid[ent], return_inverse=True)
            assert ecols.min() >= 0
            sia = np.concatenate([[1], 1 + np.cumsum(rp[rows + 1] - rp[rows])]).astype(np.int32)
            sja = np.zeros((2, ent.size), np.int32, order="F"); sja[0] = cinv + 1; sja[1] = einv + 1
            return cols, ecols, sia, sja

        cols, ecols, sia, sja = sub(False)
        kap = oracle.gno_kernel_eval(c_loc[ecols], theta_h, H, Fo * Fi)
        nsq = max(rows.size, cols.size)
        xs = np.zeros((nsq, Fi), np.float32); xs[:cols.size] = x_ext[torch.from_numpy(cols).to(dev)].cpu().numpy()
        sia_sq = np.concatenate([sia, np.full(nsq - rows.size, sia[-1], np.int32)])
        ref = oracle.gno_aggregate(xs, kap, sia_sq, sja, Fo)[:rows.size]
        out["m_rel"] = float(np.abs(m[rsel].cpu().numpy() - ref).max() / np.abs(ref).max())
        # the pull: dx_v = sum_{(u,e) in row v} K_e^T dz_u, float64 per entry
        cols, ecols, sia, sja = sub(True)
        kap = oracle.gno_kernel_eval(c_loc[ecols], theta_h, H, Fo * Fi).reshape(-1, Fi, Fo).astype(np.float64)
        gs = x_ext[torch.from_numpy(cols).to(dev)].cpu().numpy().astype(np.float64)
        dref = np.zeros((rows.size, Fi))
        for k in range(rows.size):
            for w in range(sia[k] - 1, sia[k + 1] - 1):
                dref[k] += kap[sja[1, w] - 1] @ gs[sja[0, w] - 1]
        out["dx_rel"] = float(np.abs(dx[rsel].cpu().numpy() - dref).max() / np.abs(dref).max())
        # d theta of the two blocks, S kept against S rebuilt: the same bits; finite
        parts = []
        for g, r0, r1, s in ((g_fi, 0, ni, s_int), (g_fb, ni, n, s_bnd)):
            if r1 > r0:
                gsl = x_ext[r0:r1].contiguous()
                a = ops.gno_aggregate_bwd_theta(g, theta, co, x_ext, gsl, d, H, s_save=s)
                b = ops.gno_aggregate_bwd_theta(g, theta, co, x_ext, gsl, d, H)
                parts.append(bool(torch.equal(a, b) and torch.isfinite(a).all()))
        out["dtheta_kept_equals_rebuilt"] = all(parts)
        # the scatter-form reverse pass of the two blocks (what GnoShardStep(reverse="reduce") launches): its local rows + what
        # the pull delivers must agree where no peer contributes -- on the INTERIOR rows every neighbour is local
        dxa_i, dth_i, _, fused_i = ops.gno_aggregate_bwd(g_fi, theta, co, x_ext, x_ext[:ni].contiguous(), d, H, s_save=s_int)
        dxa_b, dth_b, _, fused_b = ops.gno_aggregate_bwd(g_fb, theta, co, x_ext, x_ext[ni:n].contiguous(), d, H, s_save=s_bnd)
        out["fused"] = bool(fused_i and fused_b)

        def timed(fn, reps=3):
            fn(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        gi_, gb_ = x_ext[:ni].contiguous(), x_ext[ni:n].contiguous()
        out["reverse_pull_ms"] = timed(lambda: (ops.gno_aggregate_bwd_theta(g_fi, theta, co, x_ext, gi_, d, H, s_save=s_int),
                                                ops.gno_aggregate_bwd_theta(g_fb, theta, co, x_ext, gb_, d, H, s_save=s_bnd),
                                                ops.gno_aggregate_bwd_x_pull(g_bi, theta, co, x_ext, d, H, Fi, out=dx[:ni]),
                                                ops.gno_aggregate_bwd_x_pull(g_bb, theta, co, x_ext, d, H, Fi, out=dx[ni:])))
        out["reverse_reduce_ms"] = timed(lambda: (ops.gno_aggregate_bwd(g_fi, theta, co, x_ext, gi_, d, H, s_save=s_int),
                                                  ops.gno_aggregate_bwd(g_fb, theta, co, x_ext, gb_, d, H, s_save=s_bnd)))
        out["forward_ms"] = timed(lambda: (ops.gno_aggregate_save(g_fi, theta, co, x_ext, d, H, Fo, s_save=s_int, out=m[:ni]),
                                           ops.gno_aggregate_save(g_fb, theta, co, x_ext, d, H, Fo, s_save=s_bnd, out=m[ni:])))
        torch.cuda.synchronize()
    q.put((rank, out))
    dist.barrier()
    shard.close()
    adist.c_comm_destroy()
    dist.destroy_process_group()


def test_c4_mesh_cut_eight_ways_by_shard_create_edges(dev, tmp_path):
    """BASELINE configs[3] (2 M vertices / ~30 M entries / 64 features, d = 3, H = 64) partitioned 8 ways by
    athena_mp_shard_create_edges ITSELF: eight processes on the one GPU (shm test transport for the metadata collectives),
    every rank builds its real 250 000-row shard with its own edge columns; a mesh numbered in cell order is the locality
    case of SURVEY.md 8e -- the halo is a thin shell, so the rule picks packed rows; one rank runs the forward blocks
    (S kept), the reverse pull and d theta on the shard's own handles: sampled rows of both blocks <= 1e-5 against the
    materialising oracle.  Prints the halo fraction of the mesh."""
    from athena_amd import synth

    world, check_rank = 8, 3
    ia, ja, coords = synth.radius_graph(2_000_000, order="cells")
    assert 25e6 < ja.shape[1] < 35e6
    np.save(tmp_path / "ia.npy", ia); np.save(tmp_path / "ja.npy", ja); np.save(tmp_path / "coords.npy", coords)
    del ia, ja, coords
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_c4_shard_worker, args=(r, world, port, str(tmp_path), check_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=1500) for _ in range(world))
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    halo = [res[r]["n_halo"] / res[r]["n"] for r in range(world)]
    print(f"configs[3] mesh, 8 row blocks in cell order: halo fraction {res[0]['fraction']:.4f} of all remote rows "
          f"(per rank {min(halo):.3f} .. {max(halo):.3f} of its own rows), interior rows "
          f"{min(res[r]['n_int'] for r in range(world))} .. {max(res[r]['n_int'] for r in range(world))} of {res[0]['n']}, "
          f"mode {res[0]['mode']}")
    for r in range(world):
        assert res[r]["mode"] == "p2p" and res[r]["fraction"] < 0.1, res[r]
        assert res[r]["n_int"] > res[r]["n"] // 2 and 0 < res[r]["n_halo"] < res[r]["n"] // 4, res[r]
    c = res[check_rank]
    assert c["m_rel"] <= 1e-5 and c["dx_rel"] <= 1e-5 and c["dtheta_kept_equals_rebuilt"], c
    print(f"one real 1/8 shard of configs[3] (rank {check_rank}; eight processes share the GPU, so the times are upper bounds): forward "
          f"{c['forward_ms']:.2f} ms, reverse by pull {c['reverse_pull_ms']:.2f} ms, reverse by one contraction + reduce {c['reverse_reduce_ms']:.2f} ms")
    assert c["fused"]


@pytest.mark.parametrize("where,phase", [("halo", "first halo exchange"), ("setup", "communicator creation")])
def test_bench_py_fails_fast_when_a_rank_stalls(dev, where, phase):
    """no RCCL collective has a completion deadline of its own; bench.py's per-rank watchdog (athena_amd.dist.Watchdog) does:
    rank 1 is put to sleep on purpose before its first halo exchange (or before the communicator is built), and the launch
    ends within the deadline with {"ok": false, "error": "rank 0 stalled in <phase> ..."} and a non-zero exit code instead
    of hanging until the driver's limit"""
    import json
    import subprocess
    import time

    env = dict(os.environ, ATHENA_MP_BENCH_ONE_DEVICE="1", ATHENA_MP_BENCH_BACKEND="gloo", ATHENA_MP_COLLECTIVE_TIMEOUT_S="6",
               ATHENA_MP_BENCH_STALL=f"1:{where}")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    t0 = time.time()
    shm_before = _shm_dirs()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--nodes", "20000", "--pairs", "90000"], env=env, capture_output=True, text=True, timeout=600)
    took = time.time() - t0
    _remove_shm_dirs(shm_before)
    assert r.returncode != 0, r.stdout[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines and lines[-1]["ok"] is False, (r.stdout[-1500:], r.stderr[-1500:])
    assert "rank 0 stalled in" in lines[-1]["error"] and phase in lines[-1]["error"], lines[-1]
    assert took < 240, took


@pytest.mark.parametrize("gpus", [1, 2, 8])
def test_bench_py_gno_mesh_config(dev, gpus):
    """`python bench.py --config c4-mesh --gpus N`: graph_nop_layer forward + reverse on the configs[3] mesh generator (here
    48 000 points) cut by rows -- one rank (empty halo) and the one-device dry run at 2 and 8 ranks (shm test transport);
    the line's structure, the parity of every rank against the materialising oracle, S kept on the blocks."""
    import json
    import subprocess

    env = dict(os.environ, ATHENA_MP_BENCH_ONE_DEVICE="1", ATHENA_MP_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c4-mesh", "--gpus", str(gpus), "--nodes", "48000",
                        "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == gpus and line["unit"] == "edges/s" and line["roofline"]["bound"] == "mfma"
    assert line["parity"]["ok"] and line["parity"]["halo_rows_bit_exact"]
    assert max(line["parity"]["out_rel"], line["parity"]["dX_rel"]) <= 1e-5
    cfg = line["config"]
    assert cfg["vertices"] == 48000 and any(cfg["S_kept_per_block"]) and cfg["edge_columns_per_gpu"] > 0
    if gpus > 1:
        assert cfg["transport"].startswith("shm") and cfg["halo_rows_per_gpu"] > 0 and line["scaling"] == "strong"
        for k in ("halo_ms", "fwd_interior_ms", "fwd_boundary_ms", "bwd_dtheta_ms", "bwd_pull_interior_ms", "bwd_pull_boundary_ms",
                  "halo_recv_bytes_per_gpu_per_step", "xgmi_recv_GBps_per_gpu"):
            assert k in line["breakdown"], k
    else:
        assert cfg["halo_rows_per_gpu"] == 0


def test_c_abi_transfers_have_a_deadline_of_their_own(dev, tmp_path):
    """comm.hip's monitor thread (for hosts that are not bench.py: the Fortran drivers): every transfer put on the communication
    stream is watched through its completion event, and one still pending after ATHENA_MP_COLLECTIVE_TIMEOUT_S (library default
    1800 s; here 2) ends the process with a message naming the rank and the transfer on STDERR and exit code 3.  Exercised with kipf_shard_run on two ranks (shm test
    transport) and the test hook that parks a BOUNDED 6 s spin kernel in front of the completion event, deadline 2 s."""
    import subprocess

    exe = os.path.join(ROOT, "athena_amd", "fortran", "kipf_shard_run")
    if not os.path.exists(exe):
        pytest.skip("Fortran driver not built (no amdflang)")
    env = dict(os.environ, ATHENA_MP_COMM_TRANSPORT="shm", ATHENA_MP_COLLECTIVE_TIMEOUT_S="2", ATHENA_MP_COMM_TEST_DELAY_MS="6000",
               ATHENA_MP_HALO_MODE="p2p")
    prefix = str(tmp_path / "run")
    shm_before = _shm_dirs()
    procs = [subprocess.Popen([exe, str(r), "2", "0", str(tmp_path / "id"), "4000", "16000", "64", prefix], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    _remove_shm_dirs(shm_before)
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 3, (r, p.returncode, so.decode()[-500:], se.decode()[-800:])
        assert f"rank {r} stalled in the halo exchange in slot 0" in se.decode()
        assert '"ok": false' not in so.decode()      # stderr + exit code only: the host program's stdout is not the library's
    # ... and with the deadline above the delay the same run completes
    env["ATHENA_MP_COLLECTIVE_TIMEOUT_S"] = "60"
    env["ATHENA_MP_COMM_TEST_DELAY_MS"] = "300"
    procs = [subprocess.Popen([exe, str(r), "2", "0", str(tmp_path / "id2"), "4000", "16000", "64", prefix], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_bench_line_on_one_gpu_carries_what_the_driver_reads(dev):
    """`python bench.py` (N = 1, the command the driver issues; here with one CPU run and only configs[2] in the secondary block
    to keep it short): ONE json line with the contract's fields, `roofline` incl. the copy ceiling measured in the same run,
    `cpu_baseline` with its per-run times, `parity.ok`, the relu-epilogue step reported separately, and
    `secondary` as a block of its own whose failure would not touch `value`."""
    import json
    import subprocess

    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "ATHENA_MP_BENCH_ONE_DEVICE", "ATHENA_MP_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--cpu-runs", "1",
                        "--secondary", "c3"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "parity"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["unit"] == "edges/s" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert "configs[1]" in d["config"]["workload"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and 0.5 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert 3000 < rf["measured_copy_ceiling_GBps"] < 8000 and rf["dense"]["bound"] == "mfma"
    assert rf["reverse"]["bound"] == "hbm" and 0.5 < rf["reverse"]["frac"] < 1.0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and len(d["cpu_baseline"]["runs_s"]) == 1
    assert d["parity"]["ok"] and d["parity"]["P_bit_exact"]
    assert d["relu_epilogue"]["ms_per_step"] > d["ms_per_step"] * 0.9 and d["relu_epilogue"]["Z_rel_first_rows"] <= 1e-5
    sec = d["secondary"]
    assert "error" not in sec and sec["configs[2]"]["parity"]["ok"] and sec["configs[2]"]["step_ms"] > 0
    assert "cpu_baseline" in sec["configs[2]"] and set(sec) >= {"note", "configs[2]", "wall_s"}


def _directed_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from athena_amd import dist as adist
    from athena_amd._capi import AthenaMPError

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    # 4 vertices, 2 per rank; vertex 0 lists vertex 2, vertex 2 does not list vertex 0
    ia = np.array([1, 2, 2], np.int32) if rank == 0 else np.array([1, 1, 1], np.int32)
    cols = np.array([2], np.int64) if rank == 0 else np.zeros(0, np.int64)
    try:
        adist.CShard(adist.c_comm(dev), ia, cols)
        msg = "no error"
    except AthenaMPError as exc:
        msg = str(exc)
    q.put((rank, msg))
    dist.barrier()
    adist.c_comm_destroy()
    dist.destroy_process_group()


def test_shard_create_refuses_a_directed_graph(dev):
    """the Kipf shard's reverse pass is a pull over the rank's own rows (athena_diffstruc_extd_sub_kipf.f90:85-111 read from the
    receiving side), exact for undirected graphs only: athena_mp_shard_create checks it over all ranks -- an error on every
    rank instead of a silently different dX"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_directed_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all("not undirected" in res[r] for r in range(world)), res


def _fuzz_problem(seed, n_total, Fi, Fo, d, H):
    """a random undirected MULTIgraph with edge columns: duplicate pairs carry distinct columns, every vertex a self loop without a
    column (edge id 0, as add_self_loops leaves it), a few self loops WITH one, isolated vertices at the end"""
    rng = np.random.default_rng(seed)
    live = n_total - rng.integers(0, 6)
    E = int(rng.integers(2, 5) * live)
    u = rng.integers(0, live, E); v = rng.integers(0, live, E)
    dup = rng.integers(0, E, E // 6)                       # repeat some pairs under new columns
    u = np.concatenate([u, u[dup]]); v = np.concatenate([v, v[dup]])
    E = u.size
    eid = np.arange(1, E + 1)
    loops = u == v                                         # a self pair is ONE entry carrying its column
    src = np.concatenate([u, v[~loops], np.arange(live)]); dst = np.concatenate([v, u[~loops], np.arange(live)])
    ee = np.concatenate([eid, eid[~loops], np.zeros(live, np.int64)])
    order = np.lexsort((rng.random(src.size), src))        # random entry order inside a row
    src, dst, ee = src[order], dst[order], ee[order]
    ia = np.concatenate([[1], 1 + np.cumsum(np.bincount(src, minlength=n_total))]).astype(np.int32)
    ja = np.zeros((2, src.size), np.int32, order="F"); ja[0] = dst + 1; ja[1] = ee
    coords = rng.standard_normal((E, d)).astype(np.float32)
    x = rng.uniform(-1, 1, (n_total, Fi)).astype(np.float32); up = rng.uniform(-1, 1, (n_total, Fo)).astype(np.float32)
    theta = (0.4 * rng.standard_normal(H * d + H + Fo * Fi * H + Fo * Fi)).astype(np.float32)
    w = (0.5 * rng.standard_normal(Fo * Fi)).astype(np.float32); b = (0.1 * rng.standard_normal(Fo)).astype(np.float32)
    return ia, ja, coords, x, up, theta, w, b


def _fuzz_worker(rank, world, port, seed, dims, reverse, mode, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ATHENA_MP_HALO_MODE=mode)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from athena_amd import dist as adist

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n_total, Fi, Fo, d, H = dims
    ia, ja, coords, x, up, theta, w, b = _fuzz_problem(seed, *dims)
    shard, c_loc = adist.make_mesh_shard(rank, world, n_total, device=dev, mesh=(ia, ja, coords))
    n = shard.n
    sl = slice(rank * n, (rank + 1) * n)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(ia[rank * n:(rank + 1) * n + 1]))
    e0, e1 = int(ia[rank * n]) - 1, int(ia[(rank + 1) * n]) - 1
    py = adist.Shard(rank, world, n, rows, ja[0, e0:e1].astype(np.int64) - 1, ja[1, e0:e1].astype(np.int64))
    gi, gb = shard.graphs()[0:2]
    same = bool(np.array_equal(shard.edge_ids, py.edge_ids) and np.array_equal(shard.order, py.order)
                and np.array_equal(np.concatenate([gi.export("eid"), gb.export("eid")]) + 1, py.adj_ja[1])
                and np.array_equal(shard._export(8, np.int32), np.concatenate(py.edge_share + [np.zeros(0, np.int64)])))
    step = adist.GnoShardStep(shard, Fi, Fo, d, H, dev, inputs=(x[sl], up[sl], theta, w, b, c_loc), reverse=reverse, need_coord_grad=True)
    out = step.forward().clone().cpu().numpy()
    dx = step.backward().clone().cpu().numpy()
    q.put((rank, dict(out=out, dX=dx, grads=step.grad_flat.cpu().numpy().copy(), order=shard.order.copy(), same=same,
                      dcoords=step.dcoords.cpu().numpy().copy(), edge_ids=shard.edge_ids.copy())))
    dist.barrier()
    shard.close()
    adist.c_comm_destroy()
    dist.destroy_process_group()


@pytest.mark.parametrize("seed,world,reverse,mode", [(1, 2, "pull", "p2p"), (2, 3, "reduce", "p2p"), (3, 2, "reduce", "allgather"),
                                                     (4, 3, "pull", "allgather"), (5, 4, "reduce", "auto"), (6, 2, "pull", "auto")])
def test_node_partitioned_gno_fuzz_multigraphs_with_self_loops(dev, seed, world, reverse, mode):
    """random undirected MULTIgraphs cut by rows: duplicate pairs under distinct edge columns, self loops with and without a
    column, random entry order inside the rows, isolated vertices -- the C shard against the python plan (order, edge columns,
    per-entry columns, cut columns) and the layer step in both reverse forms against the materialising oracle, dcoords included"""
    import oracle_layers as ol
    from athena_amd.graph import graph_type

    dims = (120 * world, 4, 8, 2, 4)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fuzz_worker, args=(r, world, port, seed, dims, reverse, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n_total, Fi, Fo, d, H = dims
    ia, ja, coords, x, up, theta, w, b = _fuzz_problem(seed, *dims)
    g = graph_type.from_csr(ia, ja, num_edges=coords.shape[0])
    def run():
        outs, tapes = ol.gno_forward([g], [x], [coords], [theta, w, b], Fi, Fo, d, H, True, "none")
        dxs, dcs, grads = ol.gno_backward([g], [x], [coords], tapes, [theta, w, b], Fi, Fo, d, H, True, "none", [up])
        return outs, dxs, dcs, np.concatenate([np.asarray(a).reshape(-1) for a in grads])

    outs, dxs, dcs, g_ref = run()
    hi = ol.f64_lazy(run)
    from helpers import assert_close

    def unperm(key):
        parts = []
        for r in range(world):
            a = np.empty_like(res[r][key]); a[res[r]["order"]] = res[r][key]; parts.append(a)
        return np.concatenate(parts)

    assert np.abs(unperm("out") - outs[0]).max() <= 1e-5 * np.abs(outs[0]).max()
    assert np.abs(unperm("dX") - dxs[0]).max() <= 1e-5 * np.abs(dxs[0]).max()
    for r in range(world):
        assert res[r]["same"], r
        assert_close(res[r]["grads"], g_ref, 1e-5, f"rank {r}: [dtheta | dW | db]", f64=hi(3))
        full = dcs[0].copy()
        full[res[r]["edge_ids"]] = res[r]["dcoords"]
        assert_close(full, dcs[0], 1e-5, f"rank {r}: dcoords", f64=hi(2))


def _failing_shard_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from athena_amd import dist as adist
    from athena_amd import synth

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n_total, n = 4000, 4000 // world
    ia, cols = synth.random_graph_csr_rows(n_total, 12000, rank * n, (rank + 1) * n)
    if rank == 1:
        cols = cols.copy()
        cols[3] = n_total + 17                       # a neighbour that does not exist: athena_mp_shard_create refuses it
    try:
        adist._c_shard(dev, ia, cols)
        q.put((rank, "no error"))
    except RuntimeError as exc:
        q.put((rank, str(exc)))
    dist.barrier()
    adist.c_comm_destroy()
    dist.destroy_process_group()


def test_a_c_abi_shard_failure_raises_on_every_rank_with_the_c_error_text(dev):
    """no second implementation behind the product path: when one rank's athena_mp_shard_create fails,
    that rank raises with the C ABI's message and the others raise too (told by the agreement all-reduce) -- nothing falls
    back to dist.py's python plan, which is the CPU mirror of the gloo tests only"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    env_before = os.environ.get("ATHENA_MP_COMM_TRANSPORT")
    procs = [ctx.Process(target=_failing_shard_worker, args=(r, world, port, q)) for r in range(world)]
    shm_before = _shm_dirs()
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
    _remove_shm_dirs(shm_before)
    if env_before is None:
        os.environ.pop("ATHENA_MP_COMM_TRANSPORT", None)
    assert "C-ABI communicator / shard failed" in got[1] and "AthenaMPError" in got[1] and "outside [1,4000]" in got[1], got
    # rank 0's own C call fails too (athena_mp_shard_create agrees on argument checks across ranks before its first data
    # collective): it raises with the C ABI's text naming the rank at fault
    assert "C-ABI communicator / shard failed" in got[0] and "rank 1 rejected its arguments" in got[0], got
    src = open(os.path.join(ROOT, "athena_amd", "dist.py")).read()
    assert "FALLBACK" not in src and "_c_shard_or_none" not in src


def test_comm_stats_count_what_went_on_the_wire(dev):
    """athena_mp_comm_stats through dist.c_comm_stats(): one rank over RCCL -- ranks_seen is RCCL's own ncclCommCount, the
    version its ncclGetVersion; the all-reduce payload is counted"""
    import subprocess
    prog = r"""
import os, sys, json
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%r)
import torch, torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
from athena_amd import dist as adist, _capi
c = adist.c_comm(torch.device("cuda", 0))
buf = torch.ones(1000, device="cuda")
_capi.call("athena_mp_allreduce", c.handle, buf.data_ptr(), 1000)
torch.cuda.synchronize()
print(json.dumps(adist.c_comm_stats()))
adist.c_comm_destroy(); dist.destroy_process_group()
""" % (ROOT, str(_free_port()))
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    st = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert st["transport"] == "rccl" and st["ranks_seen"] == 1 and st["version"] > 20000, st
    assert st["sent_bytes_per_peer"] == [0]


def test_stall_handler_gets_the_hosts_last_word_before_exit_code_3(dev, tmp_path):
    """athena_mp_set_stall_handler: when comm.hip's monitor finds a transfer past its deadline it calls the host's handler ONCE,
    on the monitor thread, before the process leaves with exit code 3 -- the host's chance to flush its own state.  Two python
    ranks over the test transport, a bounded 6 s spin kernel in front of the first halo exchange's completion event, deadline
    2 s; the handler writes a file naming the rank and the transfer.  Nothing is printed on stdout by the library."""
    import subprocess
    port = _free_port()
    prog = r'''
import os, sys, ctypes as C
rank = int(sys.argv[1]); out = sys.argv[2]
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%r, ATHENA_MP_HALO_MODE="p2p")
import torch, torch.distributed as dist
dist.init_process_group("gloo", rank=rank, world_size=2)
torch.cuda.set_device(0)
from athena_amd import dist as adist, _capi
dev = torch.device("cuda", 0)
lib = _capi.load()
CB = C.CFUNCTYPE(None, C.c_int32, C.c_char_p, C.c_double)
def on_stall(r, what, seconds):
    with open(out, "w") as f:
        f.write("rank %%d stalled in %%s after %%.0f s\n" %% (r, what.decode(), seconds))
cb = CB(on_stall)
lib.athena_mp_set_stall_handler.argtypes = [CB]
assert lib.athena_mp_set_stall_handler(cb) == 0
shard = adist.make_weak_scaling_shard(rank, 2, 4000, 16000, 64, cut=0.1, device=dev)
step = adist.KipfShardStep(shard, 64, dev)
step()
torch.cuda.synchronize()
print("finished without a stall", flush=True)
''' % (ROOT, str(port))
    env = dict(os.environ, ATHENA_MP_COLLECTIVE_TIMEOUT_S="2", ATHENA_MP_COMM_TEST_DELAY_MS="6000")
    shm_before = _shm_dirs()
    procs = [subprocess.Popen([sys.executable, "-c", prog, str(r), str(tmp_path / f"stall_{r}.txt")], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    _remove_shm_dirs(shm_before)
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 3, (r, p.returncode, so.decode()[-300:], se.decode()[-800:])
        assert "finished without a stall" not in so.decode() and '"ok"' not in so.decode()
        note = (tmp_path / f"stall_{r}.txt").read_text()
        assert note.startswith(f"rank {r} stalled in") and "halo exchange" in note, note
