"""Row-partitioned Kipf layer step across the GPUs of one node (one process per GPU, torch.distributed
over RCCL/xGMI).  The reference has nothing distributed: this is new.

Partition: rank r owns the contiguous vertex block [r*n, (r+1)*n) -- its rows of X, the CSR rows of
its vertices and the matching output rows.  Per propagate ONE exchange step moves the *distinct*
remote rows a rank's CSR rows reference (halo rows), as grouped point-to-point send/recv to every
peer (torch `batch_isend_irecv` = ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd: one transfer
per xGMI link, not a ring).  Forward exchanges X, backward exchanges dP (pull form: athena's graphs
are undirected, so the transposed rows of a vertex are its own rows sorted by source id); dW is
all-reduced.  W is replicated.

The compute calls go through a backend object: `HipBackend` (the product path, libathena_mp.so).
tests/ inject a CPU backend to exercise the partition / halo logic under gloo.
"""
import numpy as np
import torch
import torch.distributed as dist


# --------------------------------------------------------------------------------------------------
# deterministic shard generator for the weak-scaling workload (every rank builds only its own rows)
# --------------------------------------------------------------------------------------------------
def _block_pairs(a, b, n, count, seed):
    """`count` undirected pairs with one endpoint in partition a and the other in partition b (a<=b),
    local ids; identical on both owners (seeded by the block)."""
    rng = np.random.Generator(np.random.PCG64([seed, a, b]))
    u = rng.integers(0, n, count, dtype=np.int64)
    if a == b:
        v = rng.integers(0, n - 1, count, dtype=np.int64)
        v = v + (v >= u)
    else:
        v = rng.integers(0, n, count, dtype=np.int64)
    return u, v


def shard_entries(rank, world, n, pairs, cut=None, seed=20260424):
    """CSR entries (row_local, col_global) of rank's rows for a graph of world*n vertices with
    world*pairs undirected pairs (+ one self-loop per vertex).  cut = fraction of pairs that cross
    partitions; None = (world-1)/world, i.e. both endpoints uniform over the whole graph."""
    if world == 1:
        cut = 0.0
    elif cut is None:
        cut = (world - 1) / world
    n_cross_block = int(round(2.0 * pairs * cut / (world - 1))) if world > 1 else 0
    n_in = pairs - (n_cross_block * (world - 1)) // 2
    rows = [np.arange(n, dtype=np.int64)]
    cols = [np.arange(n, dtype=np.int64) + rank * n]                 # self loop first in each row
    u, v = _block_pairs(rank, rank, n, n_in, seed)
    rows += [u, v]
    cols += [v + rank * n, u + rank * n]
    for b in range(world):
        if b == rank:
            continue
        lo, hi = min(rank, b), max(rank, b)
        u, v = _block_pairs(lo, hi, n, n_cross_block, seed)
        mine, theirs = (u, v) if rank == lo else (v, u)
        rows.append(mine)
        cols.append(theirs + b * n)
    rows = np.concatenate(rows)
    cols = np.concatenate(cols)
    order = np.argsort(rows, kind="stable")
    return rows[order], cols[order], cut


def halo_tau():
    """threshold of the all-gather fallback: same environment variable as comm.hip"""
    import os
    try:
        t = float(os.environ.get("ATHENA_MP_HALO_ALLGATHER_FRACTION", "0.7"))
    except ValueError:
        t = 0.7
    return t if t > 0 else 0.7


def halo_mode_env():
    """ATHENA_MP_HALO_MODE: None = auto, else "p2p" / "allgather" (comm.hip reads the same variable)"""
    import os
    m = os.environ.get("ATHENA_MP_HALO_MODE", "auto")
    return None if m in ("", "auto") else ("allgather" if m == "allgather" else "p2p")


class Shard:
    """One rank's rows with columns renumbered [local | halo] and the halo exchange plan.

    Local vertices are renumbered INTERIOR FIRST (rows that reference no remote vertex), so the rows
    [0, n_int) can be processed while the halo is still in flight and only [n_int, n) wait for it.
    `order[k]` = original local id of the vertex now called k (a layout choice of the partition; the
    original order is recovered with `order`)."""

    def __init__(self, rank, world, n, rows, cols_global, eids_global=None):
        """eids_global: optional GLOBAL edge id (1-based, 0 = none) of every entry -- graph_type%adj_ja(2,:) of a graph
        with edge features cut by rows (GNO on a mesh).  The shard's graphs then keep the edge columns, renumbered to
        the rank's own set (`edge_ids`: their global ids, 0-based, ascending), as athena_mp_shard_create_edges does."""
        self.rank, self.world, self.n = rank, world, n
        lo = rank * n
        local = (cols_global >= lo) & (cols_global < lo + n)
        is_bnd = np.zeros(n, bool)
        is_bnd[rows[~local]] = True
        self.order = np.argsort(is_bnd, kind="stable")                     # interior rows first
        new_of_old = np.empty(n, np.int64)
        new_of_old[self.order] = np.arange(n)
        self.new_of_old = new_of_old
        self.n_int = int(n - is_bnd.sum())
        rows_new = new_of_old[rows]
        perm = np.argsort(rows_new, kind="stable")                          # entry order inside a row is kept
        rows_new, cols_global, local = rows_new[perm], cols_global[perm], local[perm]
        if eids_global is not None:
            eg = np.asarray(eids_global, np.int64)[perm]
            self.edge_ids = np.unique(eg[eg > 0]) - 1
            self._eid_local = np.where(eg > 0, 1 + np.searchsorted(self.edge_ids, eg - 1), 0).astype(np.int32)
        else:
            self.edge_ids, self._eid_local = np.zeros(0, np.int64), None
        self.n_edge_cols = int(self.edge_ids.size)
        # edge columns the partition cuts, per peer (local ids, ascending global id: the peer lists the same columns in the same order)
        self.edge_share = [np.zeros(0, np.int64) for _ in range(world)]
        if eids_global is not None and world > 1:
            rem = ~local & (self._eid_local > 0)
            owner_e = (cols_global[rem] // n).astype(np.int64)
            le = self._eid_local[rem].astype(np.int64) - 1
            self.edge_share = [np.unique(le[owner_e == p]) for p in range(world)]
        counts = np.bincount(rows_new, minlength=n)
        self.adj_ia = np.concatenate([[1], 1 + np.cumsum(counts)]).astype(np.int32)
        self.cols_global = cols_global
        self.row_deg = counts.astype(np.int32)
        self.halo_ids = np.unique(cols_global[~local])                      # sorted => grouped by owner
        owner = self.halo_ids // n
        self.recv_counts = np.bincount(owner, minlength=world).astype(np.int64)
        col = np.where(local, new_of_old[np.where(local, cols_global - lo, 0)],
                       n + np.searchsorted(self.halo_ids, cols_global))
        self.n_halo = int(self.halo_ids.size)
        self.ext_ids = self.halo_ids          # global id held by each row of x_ext beyond the local ones (-1: padding)
        self.nnz = int(cols_global.size)
        self._local, self._rows_new = local, rows_new
        self._set_columns(col)
        self.halo_mode, self.halo_fraction, self.halo_tau = "p2p", 0.0, halo_tau()
        self.recv_rows = self.n_halo          # rows that cross a link into this rank per exchange
        self.send_idx = None     # filled by build_plan
        self.send_counts = None
        self.col_deg = None

    def _set_columns(self, col):
        n, world = self.n, self.world
        ja = np.zeros((2, self.nnz), np.int32, order="F")
        ja[0] = col + 1
        if self._eid_local is not None:
            ja[1] = self._eid_local
        self.adj_ja = ja
        # backward (pull) graph: same rows, entries ordered by global source id (the order the
        # reference's scatter accumulates in, athena_diffstruc_extd_sub_kipf.f90:101-109)
        key = self._rows_new * (np.int64(world) * n) + self.cols_global
        order = np.argsort(key, kind="stable")
        jb = np.zeros((2, self.nnz), np.int32, order="F")
        jb[0] = col[order] + 1
        if self._eid_local is not None:
            jb[1] = self._eid_local[order]
        self.adj_ja_bwd = jb

    def use_allgather(self, all_new_of_old, all_deg_new):
        """switch to the all-gather layout (comm.hip, mode 1): x_ext = [n local | world blocks of n rows, block p in
        rank p's own interior-first order]; a halo row is a view into its owner's block -- no pack, no send lists.
        all_new_of_old[p] = rank p's new position of each of its original local vertices, all_deg_new[p] = the degrees
        of rank p's vertices in their new order."""
        n, world, lo = self.n, self.world, self.rank * self.n
        cg, local = self.cols_global, self._local
        owner = cg // n
        pos = np.stack(all_new_of_old)[owner, cg - owner * n]
        col = np.where(local, self.new_of_old[np.where(local, cg - lo, 0)], n + owner * n + pos)
        self._set_columns(col)
        self.n_halo = world * n
        ext = np.empty(world * n, np.int64)
        for p in range(world):
            ext[p * n + all_new_of_old[p]] = p * n + np.arange(n)
        self.ext_ids = ext
        self.col_deg = np.concatenate([self.row_deg] + [np.asarray(d, np.int32) for d in all_deg_new])
        self.halo_mode = "allgather"
        self.recv_rows = (world - 1) * n

    def row_block(self, adj_ja, r0, r1):
        """CSR of rows [r0, r1) as its own (adj_ia, adj_ja) pair"""
        e0, e1 = int(self.adj_ia[r0]) - 1, int(self.adj_ia[r1]) - 1
        ia = (self.adj_ia[r0:r1 + 1] - e0).astype(np.int32)
        return ia, np.asfortranarray(adj_ja[:, e0:e1])

    # -- the surface KipfShardStep drives (CShard offers the same, backed by the C ABI) -----------------------------
    transport = "torch.distributed point-to-point (python plan)"

    def graphs(self, backend):
        """(fwd interior, fwd boundary, bwd interior, bwd boundary) row blocks as backend graphs"""
        n, ni, nc = self.n, self.n_int, self.n + self.n_halo
        def block(adj_ja, r0, r1):
            ia, ja = self.row_block(adj_ja, r0, r1)
            if self.n_edge_cols:
                return backend.make_graph(ia, ja, nc, self.row_deg[r0:r1], self.col_deg, n_edge_cols=self.n_edge_cols)
            return backend.make_graph(ia, ja, nc, self.row_deg[r0:r1], self.col_deg)
        return (block(self.adj_ja, 0, ni), block(self.adj_ja, ni, n), block(self.adj_ja_bwd, 0, ni), block(self.adj_ja_bwd, ni, n))

    def exchange(self, F, device, backend):
        return HaloExchange(self, F, device, backend)

    def allreduce_start(self, t):
        """sum t over the ranks; returns an object with wait() or None (already done)"""
        if self.world == 1:
            return None
        if _host_staged(t):
            h = t.cpu()
            dist.all_reduce(h)
            t.copy_(h)
            return None
        return dist.all_reduce(t, async_op=True)

    def edge_reduce(self, e):
        """sum over the partition of a per-edge-column quantity e [n_edge_cols, F] (mirror of athena_mp_shard_edge_reduce): cut
        columns exchanged with the peer that shares them and added in"""
        if self.world == 1:
            return e
        idx = [torch.from_numpy(a).to(e.device) for a in self.edge_share]
        sends = [e[idx[p]].contiguous() if p != self.rank and idx[p].numel() else None for p in range(self.world)]
        recvs = [torch.empty_like(t) if t is not None else None for t in sends]
        _p2p_exchange(sends, recvs, self.rank, self.world)
        for p in range(self.world):
            if recvs[p] is not None:
                e.index_add_(0, idx[p], recvs[p])
        return e

    @property
    def interior_entries(self):
        return int(self.adj_ia[self.n_int]) - 1

    def csr(self, backward=False):
        """(adj_ia [n+1] 1-based, neighbour ids [nnz] 1-based in [local | halo] numbering) of all n rows"""
        return self.adj_ia, (self.adj_ja_bwd if backward else self.adj_ja)[0]


def _host_staged(t):
    """gloo moves host memory only.  It is the transport of the CPU tests and of the single-GPU dry run of the
    N > 1 bench path (all ranks on one device, ATHENA_MP_BENCH_BACKEND=gloo); device tensors are then staged
    through the host.  With RCCL ("nccl") nothing is staged."""
    return t.is_cuda and dist.get_backend() == "gloo"


class _StagedRecv:
    def __init__(self, req, host, dst):
        self.req, self.host, self.dst = req, host, dst

    def wait(self):
        self.req.wait()
        if self.host is not None:
            self.dst.copy_(self.host)


class _StagedGather:
    def __init__(self, req, hosts, dsts):
        self.req, self.hosts, self.dsts = req, hosts, dsts

    def wait(self):
        self.req.wait()
        for h, d in zip(self.hosts, self.dsts):
            d.copy_(h)


def _p2p_start(send_bufs, recv_bufs, rank, world):
    ops, pending = [], []
    for p in range(world):
        if p == rank:
            continue
        if send_bufs[p] is not None and send_bufs[p].numel() > 0:
            sb = send_bufs[p].cpu() if _host_staged(send_bufs[p]) else send_bufs[p]
            ops.append(dist.P2POp(dist.isend, sb, p))
            pending.append((None, None))
        if recv_bufs[p] is not None and recv_bufs[p].numel() > 0:
            if _host_staged(recv_bufs[p]):
                hb = torch.empty(recv_bufs[p].shape, dtype=recv_bufs[p].dtype)
                ops.append(dist.P2POp(dist.irecv, hb, p))
                pending.append((hb, recv_bufs[p]))
            else:
                ops.append(dist.P2POp(dist.irecv, recv_bufs[p], p))
                pending.append((None, None))
    if not ops:
        return []
    return [_StagedRecv(r, h, d) for r, (h, d) in zip(dist.batch_isend_irecv(ops), pending)]


def _p2p_exchange(send_bufs, recv_bufs, rank, world):
    for r in _p2p_start(send_bufs, recv_bufs, rank, world):
        r.wait()


# --------------------------------------------------------------------------------------------------
# a deadline around everything that waits for another rank
# --------------------------------------------------------------------------------------------------
class Watchdog:
    """No RCCL collective has a completion deadline of its own: two ranks that disagree about a transfer wait for each
    other until whoever launched them gives up (the driver: 1800 s).  A monitor thread per rank arms a deadline around
    every phase that waits for a peer -- process group, communicator, shard build, first contact of each exchange, the
    timed loop, parity -- and on expiry prints ONE json line {"ok": false, "error": "rank r stalled in <phase> ..."} and
    leaves with os._exit(3): no retry, no re-exec, no cleanup that could itself block.
    ATHENA_MP_COLLECTIVE_TIMEOUT_S (default 120) is the deadline of one phase; phases that also do host work proportional
    to the graph (shard build) pass their own factor."""

    def __init__(self, rank, timeout_s=None, exit_fn=None):
        import os
        import threading
        self.rank = rank
        try:
            self.timeout = float(os.environ.get("ATHENA_MP_COLLECTIVE_TIMEOUT_S", "120")) if timeout_s is None else float(timeout_s)
        except ValueError:
            self.timeout = 120.0
        self._lock = threading.Lock()
        self._stack = []          # (phase, deadline)
        self._exit = exit_fn or self._die
        self._stop = False
        self._thread = threading.Thread(target=self._watch, name="athena_mp-watchdog", daemon=True)
        self._thread.start()

    def _die(self, phase, waited):
        import json
        import os
        import sys
        msg = {"ok": False, "error": f"rank {self.rank} stalled in {phase} for more than {waited:.0f} s "
                                     "(ATHENA_MP_COLLECTIVE_TIMEOUT_S); a peer is missing, late or in another collective"}
        try:
            sys.stdout.write(json.dumps(msg) + "\n"); sys.stdout.flush()
            sys.stderr.write(f"[athena_mp watchdog] {msg['error']}\n"); sys.stderr.flush()
        finally:
            os._exit(3)

    def _watch(self):
        import time
        while not self._stop:
            time.sleep(0.25)
            now = time.monotonic()
            with self._lock:     # an OUTER phase's earlier deadline counts too: the earliest expired one anywhere in the stack
                late = [p for p in self._stack if now > p[1]]
                hit = min(late, key=lambda p: p[1]) if late else None
            if hit is not None:
                self._exit(hit[0], hit[2])
                return

    def phase(self, name, factor=1.0):
        """context manager: `name` must finish within factor * timeout seconds"""
        wd = self

        class _P:
            def __enter__(self_inner):
                import time
                with wd._lock:
                    wd._stack.append((name, time.monotonic() + wd.timeout * factor, wd.timeout * factor))

            def __exit__(self_inner, *exc):
                with wd._lock:
                    wd._stack.pop()
                return False
        return _P()

    def close(self):
        self._stop = True


class _NoWatch:
    def phase(self, name, factor=1.0):
        import contextlib
        return contextlib.nullcontext()


_WATCH = _NoWatch()


def set_watchdog(wd):
    """bench.py (and any multi-rank driver) installs its Watchdog here: communicator creation and the collective part of
    the shard build arm it themselves"""
    global _WATCH
    _WATCH = wd if wd is not None else _NoWatch()


# --------------------------------------------------------------------------------------------------
# the product path: communicator, shard and halo exchange behind the C ABI (csrc/comm.hip, RCCL)
# --------------------------------------------------------------------------------------------------
_COMM = None


def c_comm(device):
    """the process's athena_mp_comm (created once).  The 128-byte id is drawn by rank 0 through the C ABI and handed
    round with torch.distributed's object broadcast (any backend -- gloo by default in bench.py).  The transport is RCCL;
    the host-staged TEST transport (ATHENA_MP_COMM_TRANSPORT=shm) is taken when asked for by name or when the ranks of this
    node outnumber its devices -- the one-GPU dry run of bench.py and tests/test_gpu_dist.py, where RCCL refuses several
    ranks on one device."""
    global _COMM
    if _COMM is not None:
        return _COMM
    import ctypes as C
    import os
    from . import _capi
    _capi.init(device.index or 0)
    rank, world = dist.get_rank(), dist.get_world_size()
    # The data plane is RCCL whatever torch's group runs on (gloo is the default control plane of bench.py: one RCCL communicator
    # per rank).  Only where RCCL cannot work -- more ranks on this node than devices, i.e. ranks SHARING a GPU ("Duplicate GPU
    # detected"): the one-GPU test boxes -- and nobody chose a transport by name does the test transport step in.
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    if "ATHENA_MP_COMM_TRANSPORT" not in os.environ and local_world > torch.cuda.device_count():
        os.environ["ATHENA_MP_COMM_TRANSPORT"] = "shm"
    buf = C.create_string_buffer(128)
    if rank == 0:
        _capi.call("athena_mp_comm_unique_id", buf)
    box = [bytes(buf.raw)]
    h = C.c_void_p()
    with _WATCH.phase("communicator creation (id broadcast + athena_mp_comm_create / ncclCommInitRank)"):
        dist.broadcast_object_list(box, src=0)
        _capi.call("athena_mp_comm_create", rank, world, C.create_string_buffer(box[0], 128), C.byref(h))
    name = C.create_string_buffer(96)
    _capi.call("athena_mp_comm_info", h, None, None, name, 96)
    _COMM = type("CComm", (), {})()
    _COMM.handle, _COMM.rank, _COMM.world, _COMM.transport = h, rank, world, name.value.decode()
    return _COMM


def c_comm_stats():
    """What this rank's C-ABI communicator has done so far (athena_mp_comm_stats): {"transport", "ranks_seen" (RCCL's own
    ncclCommCount), "version" (ncclGetVersion), "transfers", "allreduce_bytes", "sent_bytes_per_peer"} -- the N > 1 bench line
    carries it so that the first real multi-GPU run proves what it used.  None before a communicator exists."""
    if _COMM is None:
        return None
    import ctypes as C
    from . import _capi
    seen, ver, tr, arb = C.c_int32(0), C.c_int32(0), C.c_int64(0), C.c_int64(0)
    per = (C.c_int64 * _COMM.world)()
    _capi.call("athena_mp_comm_stats", _COMM.handle, C.byref(seen), C.byref(ver), C.byref(tr), C.byref(arb), per, _COMM.world)
    return {"transport": _COMM.transport, "ranks_seen": int(seen.value), "version": int(ver.value), "transfers": int(tr.value),
            "allreduce_bytes": int(arb.value), "sent_bytes_per_peer": [int(v) for v in per]}


def c_comm_destroy():
    global _COMM
    if _COMM is not None:
        from . import _capi
        _capi.call("athena_mp_comm_destroy", _COMM.handle)
        _COMM = None


class _CExchange:
    def __init__(self, shard, F, slot):
        self.s, self.F, self.slot = shard, F, slot

    def start(self, x_ext):
        from . import _capi
        import ctypes as C
        if tuple(x_ext.shape) != (self.s.n + self.s.n_halo, self.F) or not x_ext.is_contiguous():
            raise ValueError(f"halo exchange: x_ext must be contiguous [{self.s.n + self.s.n_halo}, {self.F}]")
        _capi.use_torch_stream()
        _capi.call("athena_mp_halo_start", self.s.handle, self.slot, self.F, C.c_void_p(x_ext.data_ptr()))
        return self

    def finish(self, _reqs=None):
        from . import _capi
        _capi.use_torch_stream()
        _capi.call("athena_mp_halo_finish", self.s.handle, self.slot)

    def __call__(self, x_ext):
        self.start(x_ext)
        self.finish()
        return x_ext

    # the transpose: rows computed for remote vertices go to their owners and are added there (athena_mp_halo_reduce_*)
    def reduce_start(self, y_ext):
        from . import _capi
        import ctypes as C
        if tuple(y_ext.shape) != (self.s.n + self.s.n_halo, self.F) or not y_ext.is_contiguous():
            raise ValueError(f"halo reduce: y_ext must be contiguous [{self.s.n + self.s.n_halo}, {self.F}]")
        _capi.use_torch_stream()
        _capi.call("athena_mp_halo_reduce_start", self.s.handle, self.slot, self.F, C.c_void_p(y_ext.data_ptr()))
        return self

    def reduce_finish(self, y_local, _reqs=None):
        from . import _capi
        import ctypes as C
        if tuple(y_local.shape) != (self.s.n, self.F) or not y_local.is_contiguous():
            raise ValueError(f"halo reduce: y_local must be contiguous [{self.s.n}, {self.F}]")
        _capi.use_torch_stream()
        _capi.call("athena_mp_halo_reduce_finish", self.s.handle, self.slot, C.c_void_p(y_local.data_ptr()))


class _CReduce:
    def __init__(self, comm):
        self.comm = comm

    def wait(self):
        from . import _capi
        _capi.use_torch_stream()
        _capi.call("athena_mp_allreduce_finish", self.comm.handle)


class CShard:
    """athena_mp_shard: the [local | halo] renumbering, send lists, halo degrees and the four row-block graph handles
    are built by the C ABI (athena_mp_shard_create); this class only holds the handle and reads its arrays back.
    n_halo = rows of x_ext beyond the local ones: the distinct remote rows in p2p mode, the padded blocks of every rank
    in all-gather mode (halo_mode; the C ABI decides from the halo fraction)."""

    def __init__(self, comm, adj_ia, cols_global, eids_global=None):
        """eids_global (1-based GLOBAL edge ids, 0 = none): athena_mp_shard_create_edges -- the row blocks keep their edge
        columns, renumbered to the rank's own set (`edge_ids`)"""
        import ctypes as C
        from . import _capi
        self.comm, self.rank, self.world = comm, comm.rank, comm.world
        ia = np.ascontiguousarray(adj_ia, np.int32)
        ja = np.zeros((2, cols_global.size), np.int32, order="F")
        ja[0] = cols_global + 1
        if eids_global is not None:
            ja[1] = eids_global
        h = C.c_void_p()
        _capi.call("athena_mp_shard_create_edges" if eids_global is not None else "athena_mp_shard_create", comm.handle,
                   ia.size - 1, ja.shape[1], ia.ctypes.data_as(C.c_void_p), ja.ctypes.data_as(C.c_void_p), C.byref(h))
        self.handle = h
        self.edge_ids = self._export(7, np.int64) if eids_global is not None else np.zeros(0, np.int64)
        self.n_edge_cols = int(self.edge_ids.size)
        n, ni, nh, nnz, roff, ntot = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64(), C.c_int64()
        _capi.call("athena_mp_shard_dims", h, C.byref(n), C.byref(ni), C.byref(nh), C.byref(nnz), C.byref(roff), C.byref(ntot))
        self.n, self.n_int, self.n_halo, self.nnz, self.row_offset = n.value, ni.value, nh.value, nnz.value, roff.value
        self.order = self._export(0, np.int32).astype(np.int64)
        self.halo_ids = self._export(1, np.int64)            # the distinct remote rows referenced
        self.ext_ids = self._export(6, np.int64)             # what each row of x_ext beyond the local ones holds (-1: padding)
        mode, frac, tau, rr = C.c_int32(), C.c_double(), C.c_double(), C.c_int64()
        _capi.call("athena_mp_shard_info", h, C.byref(mode), C.byref(frac), C.byref(tau), C.byref(rr))
        self.halo_mode = "allgather" if mode.value == 1 else "p2p"
        self.halo_fraction, self.halo_tau, self.recv_rows = frac.value, tau.value, rr.value
        self.col_deg = self._export(3, np.int32)
        self.row_deg = self.col_deg[:self.n]
        self.transport = comm.transport
        self._graphs = None

    def _export(self, which, dt):
        import ctypes as C
        from . import _capi
        cnt = C.c_int64()
        _capi.call("athena_mp_shard_export", self.handle, which, None, 0, C.byref(cnt))
        out = np.empty(cnt.value, dt)
        _capi.call("athena_mp_shard_export", self.handle, which, out.ctypes.data_as(C.c_void_p), cnt.value, C.byref(cnt))
        return out

    @property
    def send_idx(self):
        return torch.from_numpy(self._export(2, np.int32))

    def graphs(self, backend=None):
        import ctypes as C
        from . import _capi
        from .graph import DeviceGraph
        if self._graphs is None:
            out = []
            for which in range(4):
                g = C.c_void_p()
                _capi.call("athena_mp_shard_graph", self.handle, which, C.byref(g))
                out.append(DeviceGraph.borrow(g, owner=self))
            self._graphs = tuple(out)
        return self._graphs

    def exchange(self, F, device, backend=None):
        self._slots = getattr(self, "_slots", 0) + 1
        return _CExchange(self, F, (self._slots - 1) % 2)

    def allreduce_start(self, t):
        import ctypes as C
        from . import _capi
        if self.world == 1:
            return None
        _capi.use_torch_stream()
        _capi.call("athena_mp_allreduce_start", self.comm.handle, C.c_void_p(t.data_ptr()), t.numel())
        return _CReduce(self.comm)

    def edge_reduce(self, e):
        import ctypes as C
        from . import _capi
        if tuple(e.shape)[0] != self.n_edge_cols or not e.is_contiguous():
            raise ValueError(f"edge reduce: e must be contiguous [{self.n_edge_cols}, F]")
        _capi.use_torch_stream()
        _capi.call("athena_mp_shard_edge_reduce", self.handle, int(e.shape[1]), C.c_void_p(e.data_ptr()))
        return e

    @property
    def interior_entries(self):
        return int(self.graphs()[0].nnz)

    def csr(self, backward=False):
        gi, gb = self.graphs()[2:4] if backward else self.graphs()[0:2]
        rp = np.concatenate([gi.export("rowptr"), gb.export("rowptr")[1:] + gi.nnz])
        return (rp + 1).astype(np.int32), np.concatenate([gi.export("col"), gb.export("col")]) + 1

    def close(self):
        if getattr(self, "handle", None):
            from . import _capi
            self._graphs = None
            _capi.call("athena_mp_shard_destroy", self.handle)
            self.handle = None


def _c_shard(device, adj_ia, cols_global, eids_global=None):
    """The C-ABI shard of the product path (cuda).  A failure on ANY rank -- communicator, collective plan, graph handles --
    ends the job at the point of failure: the failing rank raises with the C ABI's error text, the other ranks learn of it
    from the agreement all-reduce below and raise too (a rank that died before reaching it is the watchdog's case: exit
    code 3 naming the phase).  There is no second implementation to fall back to on a GPU: dist.py's python plan is the
    CPU mirror the gloo tests hold the C ABI's arithmetic against."""
    err = None
    sh = None
    try:
        comm = c_comm(torch.device(device))
        # collective, plus host work proportional to the shard (renumbering, four graph handles): a longer leash
        with _WATCH.phase("athena_mp_shard_create (collective plan + graph handles)", factor=5.0):
            sh = CShard(comm, adj_ia, cols_global, eids_global)
    except Exception as exc:      # AthenaMPError from the C ABI, OSError from the loader ...
        err = f"{type(exc).__name__}: {exc}"[:600]
    flag = torch.tensor([0 if err is None else 1], dtype=torch.int32)
    if dist.get_backend() == "nccl":
        flag = flag.to(device)
    with _WATCH.phase("agreeing on the shard build (all_reduce of the error flag)", factor=5.0):
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if int(flag.item()) == 0:
        return sh
    if sh is not None:
        sh.close()
    raise RuntimeError(f"rank {dist.get_rank()}: C-ABI communicator / shard failed: {err}" if err else
                       f"rank {dist.get_rank()}: another rank failed to build its C-ABI communicator / shard (its message names the cause)")


def _use_c_abi(device):
    return device is not None and torch.device(device).type == "cuda" and dist.is_initialized()


def build_plan(shard, device):
    """Tell every owner which of its rows we need; learn which of ours the peers need; fetch the
    global degrees of our halo columns (the Kipf coefficient needs deg of the remote endpoint)."""
    rank, world, n = shard.rank, shard.world, shard.n
    if world == 1:
        shard.send_counts = np.zeros(1, np.int64)
        shard.send_idx = torch.zeros(0, dtype=torch.int32, device=device)
        shard.col_deg = shard.row_deg.copy()
        return shard
    rc = torch.from_numpy(shard.recv_counts)
    if dist.get_backend() != "gloo":
        rc = rc.to(device)
    allc = [torch.empty_like(rc) for _ in range(world)]
    dist.all_gather(allc, rc)
    allc = torch.stack(allc).cpu().numpy()                 # allc[q][p] = rows q needs from p
    # how the halo travels (same rule as comm.hip): all-gather of whole blocks when the ranks together need more than
    # tau of all remote rows anyway
    shard.halo_fraction = float(allc.sum()) / float((world - 1) * world * n)
    forced = halo_mode_env()
    if (forced or ("allgather" if shard.halo_fraction > shard.halo_tau else "p2p")) == "allgather":
        mine = torch.from_numpy(np.stack([shard.new_of_old, shard.row_deg.astype(np.int64)]))
        if dist.get_backend() != "gloo":
            mine = mine.to(device)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        parts = [t.cpu().numpy() for t in parts]
        shard.use_allgather([t[0] for t in parts], [t[1] for t in parts])
        shard.send_counts = np.zeros(world, np.int64)
        shard.send_idx = torch.zeros(0, dtype=torch.int32, device=device)
        return shard
    shard.send_counts = allc[:, rank].copy()
    shard.send_counts[rank] = 0
    roff = np.concatenate([[0], np.cumsum(shard.recv_counts)])
    soff = np.concatenate([[0], np.cumsum(shard.send_counts)])
    want = torch.from_numpy((shard.halo_ids % n).astype(np.int64)).to(device)     # owner-local ids
    asked = torch.empty(int(soff[-1]), dtype=torch.int64, device=device)
    _p2p_exchange([want[roff[p]:roff[p + 1]] if p != rank else None for p in range(world)],
                  [asked[soff[p]:soff[p + 1]] if p != rank else None for p in range(world)], rank, world)
    new_of_old = torch.from_numpy(shard.new_of_old).to(device)
    asked = new_of_old[asked] if asked.numel() else asked     # peers ask by original local id
    shard.send_idx = asked.to(torch.int32).contiguous()
    shard._roff, shard._soff = roff, soff
    # degrees of halo columns
    deg = torch.from_numpy(shard.row_deg.astype(np.int64)).to(device)
    sdeg = deg[asked] if asked.numel() else asked
    hdeg = torch.empty(shard.n_halo, dtype=torch.int64, device=device)
    _p2p_exchange([sdeg[soff[p]:soff[p + 1]] if p != rank else None for p in range(world)],
                  [hdeg[roff[p]:roff[p + 1]] if p != rank else None for p in range(world)], rank, world)
    shard.col_deg = np.concatenate([shard.row_deg, hdeg.cpu().numpy().astype(np.int32)])
    return shard


class HaloExchange:
    """x_ext = [local rows | halo rows]: fills the halo part from the owners."""

    def __init__(self, shard, F, device, backend):
        self.s, self.F, self.backend = shard, F, backend
        self.send_buf = torch.empty((int(shard.send_idx.numel()), F), dtype=torch.float32, device=device)

    def start(self, x_ext):
        """pack + post the grouped sends/receives; returns the requests (RCCL runs them on its own
        stream, so kernels enqueued before finish() overlap the transfer)"""
        s = self.s
        if s.world == 1:
            return []
        n = s.n
        if s.halo_mode == "allgather":     # whole blocks, as they lie: no pack
            outs = [x_ext[n + p * n:n + (p + 1) * n] for p in range(s.world)]
            if _host_staged(x_ext):
                hs = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
                return [_StagedGather(dist.all_gather(hs, x_ext[:n].cpu(), async_op=True), hs, outs)]
            return [dist.all_gather(outs, x_ext[:n], async_op=True)]
        if self.send_buf.shape[0]:
            self.backend.gather_rows(x_ext[:n], s.send_idx, out=self.send_buf)   # pack (HIP gather kernel)
        roff, soff = s._roff, s._soff
        return _p2p_start([self.send_buf[soff[p]:soff[p + 1]] if p != s.rank else None for p in range(s.world)],
                          [x_ext[n + roff[p]:n + roff[p + 1]] if p != s.rank else None for p in range(s.world)],
                          s.rank, s.world)

    @staticmethod
    def finish(reqs):
        for r in reqs:
            r.wait()

    def __call__(self, x_ext):
        self.finish(self.start(x_ext))
        return x_ext

    # the transpose (mirror of athena_mp_halo_reduce_*): every rank sends what it computed for remote vertices to their owners
    def reduce_start(self, y_ext):
        s = self.s
        if s.world == 1:
            return []
        n = s.n
        if s.halo_mode == "allgather":       # whole blocks: block p of y_ext goes to rank p
            self._red = torch.empty((s.world, n, self.F), dtype=y_ext.dtype, device=y_ext.device)
            sends = [y_ext[n + p * n:n + (p + 1) * n] if p != s.rank else None for p in range(s.world)]
            recvs = [self._red[p] if p != s.rank else None for p in range(s.world)]
        else:
            roff, soff = s._roff, s._soff
            self._red = torch.empty((int(soff[-1]), self.F), dtype=y_ext.dtype, device=y_ext.device)
            sends = [y_ext[n + roff[p]:n + roff[p + 1]] if p != s.rank else None for p in range(s.world)]
            recvs = [self._red[soff[p]:soff[p + 1]] if p != s.rank else None for p in range(s.world)]
        return _p2p_start([t.contiguous() if t is not None else None for t in sends], recvs, s.rank, s.world)

    def reduce_finish(self, y_local, reqs=None):
        s = self.s
        if s.world == 1:
            return
        for r in reqs or []:
            r.wait()
        for p in range(s.world):               # peer by peer in rank order: the order the C ABI adds in
            if p == s.rank:
                continue
            if s.halo_mode == "allgather":
                y_local.add_(self._red[p])
            else:
                a, b = int(s._soff[p]), int(s._soff[p + 1])
                if b > a:
                    y_local.index_add_(0, s.send_idx[a:b].long(), self._red[a:b])


class HipBackend:
    """the product path: hand-written HIP kernels behind the C ABI"""
    writes_in_place = True   # gno_aggregate_bwd(dx_out=, dtheta_out=): results land in the step's own buffers

    def __init__(self, device):
        from . import _capi
        _capi.init(device.index or 0)
        self.device = device

    def make_graph(self, adj_ia, adj_ja, n_cols, row_deg, col_deg, n_edge_cols=0):
        from .graph import DeviceGraph
        return DeviceGraph(adj_ia, adj_ja, n_cols=n_cols, n_edge_cols=n_edge_cols, row_deg=row_deg, col_deg=col_deg,
                           device=self.device.index or 0)

    def __getattr__(self, name):
        from . import ops
        return getattr(ops, name)


def make_weak_scaling_shard(rank, world, n, pairs, F, cut=None, device=None, seed=20260424):
    rows, cols, cut = shard_entries(rank, world, n, pairs, cut, seed)
    if _use_c_abi(device):
        ia = np.concatenate([[1], 1 + np.cumsum(np.bincount(rows, minlength=n))])
        sh = _c_shard(device, ia, cols)
    else:                                                  # CPU mirror (gloo tests)
        sh = build_plan(Shard(rank, world, n, rows, cols), device)
    sh.cut = cut
    return sh


def make_global_shard(rank, world, n_total, pairs, device=None, seed=20260424, locality=None):
    """STRONG scaling: rank's contiguous row block of ONE fixed graph -- synth.random_graph_csr(n_total, pairs, seed),
    the generator of BASELINE configs[1] / configs[4] -- built from the shared pair stream without
    materialising the other ranks' rows.  n_total must divide by world."""
    from . import synth
    if n_total % world:
        raise ValueError(f"{n_total} vertices do not split into {world} equal row blocks")
    n = n_total // world
    ia, cols = synth.random_graph_csr_rows(n_total, pairs, rank * n, (rank + 1) * n, seed=seed, locality=locality)
    if _use_c_abi(device):
        sh = _c_shard(device, ia, cols)
    else:                                                  # CPU mirror (gloo tests)
        rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(ia))
        sh = build_plan(Shard(rank, world, n, rows, cols), device)
    sh.cut = None
    sh.n_total = n_total
    return sh


class KipfShardStep:
    """One interior Kipf layer fwd+bwd on a row shard (the bench step), F -> Fo features:
         aggregate first (the reference's association):
             exchange(X); (P, Z) = fused(A^ X, . W); exchange(dZ) under dW = dZ P^T (all-reduce);
             dX = (A^T dZ) . W  (fused pull over the shard's rows)
         transform first (4 Fo <= 3 F): the dense step runs on the local rows BEFORE the exchange, so
         the forward halo carries Fo-wide rows instead of F-wide ones:
             Y = X W^T; exchange(Y); Z = A^ Y; exchange(dZ); (Qp, Qc) = one dual pull of dZ;
             dW = Qc^T X (all-reduce); dX = Qp W"""

    def __init__(self, shard, F, device, backend=None, seed=1, exact=False, Fo=None, order="auto", inputs=None):
        """inputs: optional (x [n, F], dz [n, Fo], w [Fo*F]) host arrays for the rank's rows in their ORIGINAL local
        order (the shard renumbers them interior-first); default: seeded random per rank"""
        self.s, self.F, self.device = shard, F, device
        self.Fo = Fo = F if Fo is None else int(Fo)
        if not (order in ("auto", "aggregate_first", "transform_first")):
            raise ValueError('expected: order in ("auto", "aggregate_first", "transform_first")')
        self.transform_first = order == "transform_first" or (order == "auto" and 4 * Fo <= 3 * F)
        self.b = backend or HipBackend(device)
        n, nh = shard.n, shard.n_halo
        ni = shard.n_int
        # interior rows touch no halo column: they run while the exchange is in flight
        self.g_fwd_int, self.g_fwd_bnd, self.g_bwd_int, self.g_bwd_bnd = shard.graphs(self.b)
        self.exact = exact
        rng = np.random.Generator(np.random.PCG64([seed, shard.rank]))
        xw = Fo if self.transform_first else F            # width of the rows the forward exchange moves
        self.x_ext = torch.empty((n + nh, F) if not self.transform_first else (n, F), dtype=torch.float32, device=device)
        if inputs is not None:
            x_h, dz_h, w_h = inputs
            if not (x_h.shape == (n, F) and dz_h.shape == (n, Fo) and w_h.size == Fo * F):
                raise ValueError("inputs: expected x [n, F], dz [n, Fo], w [Fo*F] for this rank's rows")
            x_h, dz_h = x_h[shard.order], dz_h[shard.order]
        else:
            x_h = rng.uniform(-1, 1, (n, F)).astype(np.float32)
            dz_h = rng.uniform(-1, 1, (n, Fo)).astype(np.float32)
            wr = np.random.Generator(np.random.PCG64(seed + 1))          # W identical on every rank
            w_h = (wr.standard_normal(Fo * F) * np.sqrt(2.0 / F)).astype(np.float32)
        self.x_ext[:n] = torch.from_numpy(np.ascontiguousarray(x_h, np.float32)).to(device)
        self.dZ_ext = torch.empty((n + nh, Fo), dtype=torch.float32, device=device)
        self.dZ_ext[:n] = torch.from_numpy(np.ascontiguousarray(dz_h, np.float32)).to(device)
        self.dZ = self.dZ_ext[:n]
        self.W = torch.from_numpy(np.ascontiguousarray(w_h, np.float32).reshape(-1)).to(device)
        self.Z = torch.empty((n, Fo), dtype=torch.float32, device=device)
        self.dW = torch.empty(Fo * F, dtype=torch.float32, device=device)
        self.dX = torch.empty((n, F), dtype=torch.float32, device=device)
        if self.transform_first:
            self.y_ext = torch.empty((n + nh, Fo), dtype=torch.float32, device=device)
            self.Qp = torch.empty((n, Fo), dtype=torch.float32, device=device)
            self.Qc = self.Qp if exact else torch.empty((n, Fo), dtype=torch.float32, device=device)
            self.P = None
        else:
            self.P = torch.empty((n, F), dtype=torch.float32, device=device)
        self.xchg = shard.exchange(xw, device, self.b)                    # forward rows
        self.xchg_o = self.xchg if xw == Fo else shard.exchange(Fo, device, self.b)       # dZ rows

    def _allreduce_dw(self):
        # asynchronous: a blocking all_reduce would make the compute stream wait for the collective, and the
        # collective queues behind the halo transfer on the communication stream -- the interior rows would then
        # start only after the exchange they are meant to hide
        return self.s.allreduce_start(self.dW)

    def __call__(self, events=None, events_bnd=None):
        """events / events_bnd: optional lists; a (start, end) pair of torch.cuda events around the interior / the
        boundary forward launch is appended (bench.py's roofline at N > 1)"""
        if self.transform_first:
            return self._step_transform_first(events)
        s, b, F, Fo, n, ni = self.s, self.b, self.F, self.Fo, self.s.n, self.s.n_int
        reqs = self.xchg.start(self.x_ext)                                        # halo of X in flight ...
        if events is not None:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        b.kipf_layer_fwd(self.g_fwd_int, self.x_ext, self.W, Fo, P=self.P[:ni], Z=self.Z[:ni])   # ... under the interior rows
        if events is not None:
            e1.record()
            events.append((e0, e1))
        self.xchg.finish(reqs)
        if events_bnd is not None:
            e2 = torch.cuda.Event(enable_timing=True); e3 = torch.cuda.Event(enable_timing=True)
            e2.record()
        b.kipf_layer_fwd(self.g_fwd_bnd, self.x_ext, self.W, Fo, P=self.P[ni:], Z=self.Z[ni:])
        if events_bnd is not None:
            e3.record()
            events_bnd.append((e2, e3))
        reqs = self.xchg_o.start(self.dZ_ext)                                     # halo of dZ in flight ...
        b.matmul_dw(self.P, self.dZ, out=self.dW)                                 # ... under dW
        red = self._allreduce_dw()
        b.pull_gemm(self.g_bwd_int, self.dZ_ext, self.W, F, exact=self.exact, out=self.dX[:ni])   # ... and the interior rows
        self.xchg_o.finish(reqs)
        b.pull_gemm(self.g_bwd_bnd, self.dZ_ext, self.W, F, exact=self.exact, out=self.dX[ni:])
        if red is not None:
            red.wait()
        return self.dX

    def _step_transform_first(self, events=None):
        s, b, F, Fo, n, ni = self.s, self.b, self.F, self.Fo, self.s.n, self.s.n_int
        b.matmul(self.W, self.x_ext[:n], Fo, out=self.y_ext[:n])                  # dense step on the local rows
        reqs = self.xchg.start(self.y_ext)                                        # halo of Y (Fo wide) in flight ...
        if events is not None:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        b.kipf_propagate(self.g_fwd_int, self.y_ext, out=self.Z[:ni])             # ... under the interior rows
        if events is not None:
            e1.record()
            events.append((e0, e1))
        self.xchg.finish(reqs)
        b.kipf_propagate(self.g_fwd_bnd, self.y_ext, out=self.Z[ni:])
        reqs = self.xchg_o.start(self.dZ_ext)                                     # halo of dZ in flight ...
        self._pull(self.g_bwd_int, 0, ni)                                         # ... under the interior rows
        self.xchg_o.finish(reqs)
        self._pull(self.g_bwd_bnd, ni, n)
        b.matmul_dw(self.x_ext[:n], self.Qc, out=self.dW)                         # dW = (A^^T dZ)^T X
        red = self._allreduce_dw()
        b.matmul_dx(self.W, self.Qp, F, out=self.dX)                              # dX = (A^T dZ) W, the reference's scatter
        if red is not None:
            red.wait()
        return self.dX

    def _pull(self, g, r0, r1):
        if self.exact:
            self.b.kipf_propagate(g, self.dZ_ext, out=self.Qp[r0:r1])
        else:
            self.b.pull_dual(g, self.dZ_ext, plain=self.Qp[r0:r1], coef=self.Qc[r0:r1])


def first_contact(step, watch, sync):
    """every transfer of a sharded step (KipfShardStep / GnoShardStep) ONCE on its own, each under the watchdog and with
    the device drained behind it (`sync`): the first time two ranks meet in a collective is where a mismatch shows, and a
    stall then names the rank and the transfer instead of hanging the whole launch in its first step"""
    s = step.s
    if s.world == 1:
        return
    fwd = getattr(step, "y_ext", None) if getattr(step, "transform_first", False) else step.x_ext
    with watch.phase("first halo exchange (forward rows)"):
        step.xchg.finish(step.xchg.start(fwd))
        sync()
    rev = step.dZ_ext if hasattr(step, "dZ_ext") else step.g_ext
    with watch.phase("first halo exchange (gradient rows)"):
        step.xchg_o.finish(step.xchg_o.start(rev))
        sync()
    if getattr(step, "xchg_r", None) is not None:
        with watch.phase("first halo reduce (feature-gradient rows to their owners)"):
            buf = torch.zeros((s.n + s.n_halo, step.Fi), dtype=torch.float32, device=step.device)
            step.xchg_r.reduce_finish(step.dX, step.xchg_r.reduce_start(buf))
            sync()
    grads = step.dW if hasattr(step, "dZ_ext") else step.grad_flat
    with watch.phase("first all-reduce of the parameter gradients"):
        grads.zero_()
        red = s.allreduce_start(grads)
        if red is not None:
            red.wait()
        sync()


def build_kipf_step(shard, F, device, backend=None, Fo=None, order="auto", inputs=None):
    step = KipfShardStep(shard, F, device, backend, Fo=Fo, order=order, inputs=inputs)
    halo_bytes = shard.recv_rows * 4 * ((step.Fo if step.transform_first else F) + step.Fo)
    if getattr(shard, "n_total", None) is not None:
        graph = f"contiguous row block of ONE fixed graph of {shard.n_total} vertices (the single-GPU workload, strong scaling)"
    elif shard.world > 1 and abs(shard.cut - (shard.world - 1) / shard.world) < 1e-9:
        graph = "random graph, both endpoints uniform over all N*vertices_per_gpu vertices (no partition structure)"
    else:
        graph = (f"stochastic block model, one block per GPU, fixed inter-block density: {shard.cut:.4f} of the undirected "
                 "pairs cross partitions at this N")
    info = {"graph": graph,
            "halo_rows_per_gpu": int(shard.halo_ids.size), "halo_recv_rows_per_gpu": int(shard.recv_rows),
            "halo_bytes_per_gpu_per_step": halo_bytes,
            "halo_mode": shard.halo_mode, "halo_fraction": round(float(shard.halo_fraction), 4),
            "halo_allgather_threshold": shard.halo_tau,
            "interior_rows_per_gpu": shard.n_int, "interior_entries_per_gpu": shard.interior_entries,
            "transport": shard.transport}
    return step, shard.nnz, info


def _serial_breakdown():
    import os
    return bool(os.environ.get("ATHENA_MP_BENCH_SERIAL_BREAKDOWN"))


def _make_timed(s, iters):
    """timed(fn, collective=False): mean device time of fn over `iters` calls after one warm-up.  Under
    ATHENA_MP_BENCH_SERIAL_BREAKDOWN the NON-collective parts are timed one rank at a time (the others wait at a barrier), so
    that the one-device dry run -- all ranks on one GPU -- still yields each shard's real compute times."""
    serial = _serial_breakdown() and s.world > 1

    def once(fn):
        fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    def timed(fn, collective=False):
        if collective or not serial:
            fn()
            torch.cuda.synchronize()
            if s.world > 1:
                dist.barrier()
            return once(fn)
        t = 0.0
        for r in range(s.world):
            dist.barrier()
            if r == s.rank:
                t = once(fn)
        dist.barrier()
        return t

    return timed


def measure_breakdown(step, iters=5):
    """Each part of the sharded step timed ALONE (device events, after the timed loop of bench.py): the two halo
    exchanges (pack + grouped send/recv + wait), the interior launches and the boundary launches of both passes, the
    dW contraction.  In the step itself the exchanges run under the interior launches; the parts are timed apart so a
    scaling curve can be read (what is communication, what is compute).  Aggregate-first steps only."""
    if step.transform_first:
        return {}
    s, b, F, Fo, n, ni = step.s, step.b, step.F, step.Fo, step.s.n, step.s.n_int

    timed = _make_timed(s, iters)

    def halo():
        step.xchg.finish(step.xchg.start(step.x_ext))
        step.xchg_o.finish(step.xchg_o.start(step.dZ_ext))

    def interior():
        b.kipf_layer_fwd(step.g_fwd_int, step.x_ext, step.W, Fo, P=step.P[:ni], Z=step.Z[:ni])
        b.pull_gemm(step.g_bwd_int, step.dZ_ext, step.W, F, exact=step.exact, out=step.dX[:ni])

    def boundary():
        b.kipf_layer_fwd(step.g_fwd_bnd, step.x_ext, step.W, Fo, P=step.P[ni:], Z=step.Z[ni:])
        b.pull_gemm(step.g_bwd_bnd, step.dZ_ext, step.W, F, exact=step.exact, out=step.dX[ni:])

    def dw():
        b.matmul_dw(step.P, step.dZ, out=step.dW)

    out = {"halo_ms": timed(halo, collective=True) if s.world > 1 else 0.0, "interior_ms": timed(interior), "boundary_ms": timed(boundary),
           "dw_ms": timed(dw)}
    if _serial_breakdown():
        out["compute_parts_timed"] = "one rank at a time (ATHENA_MP_BENCH_SERIAL_BREAKDOWN): valid per-shard numbers even when the ranks share a device"
    recv_bytes = s.recv_rows * 4 * (F + Fo)
    out["halo_mode"], out["halo_fraction"], out["halo_allgather_threshold"] = s.halo_mode, round(float(s.halo_fraction), 4), s.halo_tau
    out["halo_recv_bytes_per_gpu_per_step"] = recv_bytes
    out["xgmi_recv_GBps_per_gpu"] = (recv_bytes / (out["halo_ms"] * 1e-3) / 1e9) if out["halo_ms"] > 0 else None
    return out


# --------------------------------------------------------------------------------------------------
# ONE graph with edge features cut by rows: the graph neural operator on a partitioned mesh (SURVEY.md 8e:
# "GNO: as Kipf plus replicated theta and all-reduce of dtheta"; BASELINE configs[3] is one 2 M-vertex mesh)
# --------------------------------------------------------------------------------------------------
def locality_order(adj_ia, adj_ja):
    """A vertex numbering under which a CONTIGUOUS row block is a compact piece of the graph, for graphs that arrive in an
    arbitrary order and carry no coordinates (the layers only see edge geometry): reverse Cuthill-McKee on the CSR pattern
    (scipy, host, 0.05 s per 200 k vertices) -- the optional offline reordering SURVEY.md 8e names.  Returns `perm` with
    perm[k] = old id (0-based) of the vertex numbered k.  On the configs[3] generator's mesh numbered as drawn, 8 row blocks
    need 5.8 x their own row count as halo rows; under this order 0.12 - 0.44 x; under the Morton order of the points' cells
    (synth.radius_graph(order="cells"), which needs the coordinates) 0.11 - 0.13 x."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    ia = np.asarray(adj_ia, np.int64) - 1
    n = ia.size - 1
    a = sp.csr_matrix((np.ones(ia[-1], np.int8), np.asarray(adj_ja[0], np.int64) - 1, ia), shape=(n, n))
    return np.asarray(reverse_cuthill_mckee(a, symmetric_mode=True), np.int64)


def permute_csr(adj_ia, adj_ja, perm):
    """the same graph with vertex perm[k] renamed k: rows re-ordered, neighbour ids renamed, the order of the entries inside a
    row and their edge ids kept (edge features are untouched; vertex features follow as x[perm]).  Fortran-convention arrays."""
    perm = np.asarray(perm, np.int64)
    n = perm.size
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)
    ia = np.asarray(adj_ia, np.int64) - 1
    counts = (ia[1:] - ia[:-1])[perm]
    new_ia = np.concatenate([[0], np.cumsum(counts)])
    src = np.repeat(ia[:-1][perm], counts) + (np.arange(new_ia[-1]) - np.repeat(new_ia[:-1], counts))   # old entry of each new entry
    ja = np.empty((2, src.size), np.int32, order="F")
    ja[0] = inv[np.asarray(adj_ja[0], np.int64)[src] - 1] + 1
    ja[1] = np.asarray(adj_ja[1])[src]
    return (new_ia + 1).astype(np.int32), ja


def make_mesh_shard(rank, world, n_points, device=None, mean_degree=15.0, seed=4, mesh=None):
    """rank's contiguous row block of the radius mesh synth.radius_graph(n_points, order="cells") -- BASELINE configs[3]'s
    generator with the points numbered in Morton order of their cell, so a block is a compact region and its halo a thin
    shell (SURVEY.md 8e "locality graphs: near-linear").  n_points must divide by world.  Returns (shard, coords_local):
    the rank's edge geometry [n_edge_cols, d], i.e. the rows `shard.edge_ids` of the global coords.
    mesh: optional prebuilt (adj_ia, adj_ja, coords) of the whole graph (tests)."""
    from . import synth
    if n_points % world:
        raise ValueError(f"{n_points} vertices do not split into {world} equal row blocks")
    n = n_points // world
    ia, ja, coords = mesh if mesh is not None else synth.radius_graph(n_points, mean_degree, seed, order="cells")
    e0, e1 = int(ia[rank * n]) - 1, int(ia[(rank + 1) * n]) - 1
    ia_l = (ia[rank * n:(rank + 1) * n + 1].astype(np.int64) - e0).astype(np.int32)
    cols = ja[0, e0:e1].astype(np.int64) - 1
    eids = ja[1, e0:e1].astype(np.int64)
    if _use_c_abi(device):
        sh = _c_shard(device, ia_l, cols, eids)
    else:                                                  # CPU mirror (gloo tests)
        rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(ia_l))
        sh = build_plan(Shard(rank, world, n, rows, cols, eids), device)
    sh.cut = None
    sh.n_total = n_points
    return sh, np.ascontiguousarray(coords[sh.edge_ids], np.float32)


class GnoShardStep:
    """graph_nop_layer forward + reverse on a row shard of ONE mesh (update_message_gno, athena_graph_nop_layer.f90:690-788,
    and the _val gradients of athena_diffstruc_extd_sub_nop.f90 composed as the layer mirror composes them):
        forward   exchange(x);  m = gno_aggregate(kappa(coords), x_ext): interior rows under the transfer, boundary rows
                  after it;  out = act(m + W x + b)
        reverse   dz = act'(.) up;  exchange(dz);  under the transfer everything that needs LOCAL rows of dz only:
                  db, dW = dz^T x, dtheta = the two blocks' S^T dz + kernel-MLP halves, and the interior rows of
                  dx = sum_{(u,e) in row v} K_e^T dz_u (the pull over the rank's own rows: both directions of a pair share
                  one edge column, :369-376);  then the boundary rows of dx, dx += dz W.
                  theta, W, b are replicated; [dtheta | dW | db] is summed over the ranks in ONE all-reduce.
    The edge geometry stays where its rows are: a rank holds coords for the edge columns its rows reference."""

    def __init__(self, shard, Fi, Fo, d, H, device, backend=None, inputs=None, activation="none", use_bias=True, keep_s=None,
                 seed=1, reverse="auto", need_coord_grad=False):
        """inputs: (x [n, Fi], up [n, Fo], theta, W [Fo*Fi], b [Fo] or None, coords [n_edge_cols, d]) host arrays, x / up
        for the rank's rows in their ORIGINAL local order; default: seeded random (theta, W, b identical on every rank)"""
        self.s, self.Fi, self.Fo, self.d, self.H, self.device = shard, Fi, Fo, d, H, device
        self.b = backend or HipBackend(device)
        self.act, self.use_bias = activation, use_bias
        n, nh, ne = shard.n, shard.n_halo, shard.n_edge_cols
        self.g_fwd_int, self.g_fwd_bnd, self.g_bwd_int, self.g_bwd_bnd = shard.graphs(self.b)
        nth = H * d + H + Fo * Fi * H + Fo * Fi
        if inputs is not None:
            x_h, up_h, th_h, w_h, b_h, c_h = inputs
            x_h, up_h = np.asarray(x_h, np.float32)[shard.order], np.asarray(up_h, np.float32)[shard.order]
        else:
            rng = np.random.Generator(np.random.PCG64([seed, shard.rank]))
            x_h = rng.uniform(-1, 1, (n, Fi)).astype(np.float32)
            up_h = rng.uniform(-1, 1, (n, Fo)).astype(np.float32)
            c_h = rng.uniform(-0.05, 0.05, (ne, d)).astype(np.float32)
            wr = np.random.Generator(np.random.PCG64(seed + 1))
            th_h = (wr.standard_normal(nth) * 0.1).astype(np.float32)
            w_h = (wr.standard_normal(Fo * Fi) * np.sqrt(2.0 / Fi)).astype(np.float32)
            b_h = (wr.standard_normal(Fo) * 0.1).astype(np.float32)
        if not (x_h.shape == (n, Fi) and up_h.shape == (n, Fo) and th_h.size == nth and w_h.size == Fo * Fi and c_h.shape == (ne, d)):
            raise ValueError("inputs: expected x [n, Fi], up [n, Fo], theta, W [Fo*Fi], b, coords [n_edge_cols, d]")
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(device)
        self.x_ext = torch.empty((n + nh, Fi), dtype=torch.float32, device=device)
        self.x_ext[:n] = t(x_h)
        self.g_ext = torch.empty((n + nh, Fo), dtype=torch.float32, device=device)     # dz: local rows | halo rows
        self.up = t(up_h)
        self.theta, self.W, self.coords = t(th_h).reshape(-1), t(w_h).reshape(-1), t(c_h)
        self.bias = t(b_h).reshape(-1) if use_bias else None
        self.m = torch.empty((n, Fo), dtype=torch.float32, device=device)
        self.out = None
        self.dX = torch.empty((n, Fi), dtype=torch.float32, device=device)
        self.grad_flat = torch.zeros(nth + Fo * Fi + (Fo if use_bias else 0), dtype=torch.float32, device=device)
        self.dtheta, self.dW = self.grad_flat[:nth], self.grad_flat[nth:nth + Fo * Fi]
        self.db = self.grad_flat[nth + Fo * Fi:] if use_bias else None
        self.ones = torch.ones((n, 1), dtype=torch.float32, device=device)
        self.xchg = shard.exchange(Fi, device, self.b)
        self.xchg_o = self.xchg if Fi == Fo else shard.exchange(Fo, device, self.b)
        # the forward pass may keep S per block for the reverse pass's S^T dz; None: when the backend offers it
        self.keep_s = keep_s
        self._s = [None, None]
        # how the feature gradient crosses the partition:
        #   "pull"    halo exchange of dz, then dx_v = sum_{(u,e) in row v} K_e^T dz_u over the rank's own rows (needs the graph's
        #             symmetry, checked at shard creation) -- dtheta and dx from separate launches;
        #   "reduce"  no exchange of dz at all: athena_mp_gno_aggregate_bwd on each forward block gives dtheta AND the scatter-form
        #             dx of every column the block's rows touch, local or not, from ONE contraction; the rows
        #             computed for remote vertices travel to their owners (athena_mp_halo_reduce_*) under the interior block
        if reverse not in ("auto", "pull", "reduce"):
            raise ValueError('reverse: "auto", "pull" or "reduce"')
        self.reverse = ("reduce" if hasattr(self.b, "gno_aggregate_bwd") and Fi % 4 == 0 else "pull") if reverse == "auto" else reverse
        self.xchg_r = shard.exchange(Fi, device, self.b) if self.reverse == "reduce" else None
        # the coordinate gradient (athena_diffstruc_extd_sub_nop.f90:137-216) on request: each block's share over the rank's
        # edge columns, summed, then the columns the partition cuts completed by the peer's share (shard.edge_reduce)
        self.need_coord_grad = need_coord_grad
        self.dcoords = None

    def _keeps(self, g):
        if self.keep_s is False or not hasattr(self.b, "gno_saved_bytes") or g.n_rows == 0:
            return False
        return self.b.gno_saved_bytes(g, self.d, self.H, self.Fi, self.Fo) > 0

    def _fwd_block(self, k, g, r0, r1):
        b = self.b
        if r1 == r0:
            return
        if self._keeps(g):
            _, self._s[k] = b.gno_aggregate_save(g, self.theta, self.coords, self.x_ext, self.d, self.H, self.Fo, s_save=self._s[k],
                                                 out=self.m[r0:r1])
        else:
            self._s[k] = None
            b.gno_aggregate(g, self.theta, self.coords, self.x_ext, self.d, self.H, self.Fo, out=self.m[r0:r1])

    def forward(self, events=None):
        s, b, n, ni = self.s, self.b, self.s.n, self.s.n_int
        reqs = self.xchg.start(self.x_ext)                                         # halo of x in flight ...
        if events is not None:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        self._fwd_block(0, self.g_fwd_int, 0, ni)                                  # ... under the interior rows
        if events is not None:
            e1.record()
            events.append((e0, e1))
        z = b.matmul(self.W, self.x_ext[:n], self.Fo, bias=self.bias)              # ... and the bypass W x + b
        self.xchg.finish(reqs)
        self._fwd_block(1, self.g_fwd_bnd, ni, n)
        b.axpy(1.0, self.m, z)
        self.z = z
        self.out = z if self.act in ("none", "linear") else b.activation(self.act, z)
        return self.out

    def backward(self):
        s, b, n, ni = self.s, self.b, self.s.n, self.s.n_int
        d, H, Fi, Fo = self.d, self.H, self.Fi, self.Fo
        one = s.world == 1 and ni == n and hasattr(b, "gno_aggregate_bwd")
        if one:      # nothing travels: dz stays where it is
            dz = self.up if self.act in ("none", "linear") else b.activation_bwd(self.act, self.out, self.up, z=self.z)
        else:        # the ranks of a partition exchange (or reduce into) the rows behind the local ones
            dz = self.g_ext[:n]
            if self.act in ("none", "linear"):
                dz.copy_(self.up)
            else:
                dz.copy_(b.activation_bwd(self.act, self.out, self.up, z=self.z))
        self.dz = dz                                                               # (what a checker of dX reads)
        if one:
            # one rank, nothing to exchange: the whole reverse pass of the aggregation from ONE contraction
            # (athena_mp_gno_aggregate_bwd) -- on a shard its per-entry partials would have to travel to the
            # column's owner, so the ranks of a partition take the pull below
            if self.use_bias:
                b.matmul_dw(self.ones, dz, out=self.db)
            b.matmul_dw(self.x_ext[:n], dz, out=self.dW)
            if getattr(b, "writes_in_place", False):
                _, _, dc, _ = b.gno_aggregate_bwd(self.g_fwd_int, self.theta, self.coords, self.x_ext, dz, d, H, s_save=self._s[0],
                                                  need_dcoords=self.need_coord_grad, dx_out=self.dX, dtheta_out=self.dtheta)
            else:
                dxa, dth, dc, _ = b.gno_aggregate_bwd(self.g_fwd_int, self.theta, self.coords, self.x_ext, dz, d, H, s_save=self._s[0],
                                                      need_dcoords=self.need_coord_grad)
                self.dtheta.copy_(dth)
                self.dX.copy_(dxa)
            self.dcoords = dc
            b.axpy(1.0, b.matmul_dx(self.W, dz, Fi), self.dX)
            return self.dX
        if self.reverse == "reduce":
            return self._backward_reduce(dz)
        reqs = self.xchg_o.start(self.g_ext)                                       # halo of dz in flight; under it:
        if self.use_bias:
            b.matmul_dw(self.ones, dz, out=self.db)                                # db[o] = sum_v dz[v,o]
        b.matmul_dw(self.x_ext[:n], dz, out=self.dW)
        first = True
        for k, (g, r0, r1) in enumerate(((self.g_fwd_int, 0, ni), (self.g_fwd_bnd, ni, n))):
            if r1 == r0:
                continue
            part = b.gno_aggregate_bwd_theta(g, self.theta, self.coords, self.x_ext, dz[r0:r1], d, H, s_save=self._s[k])
            if first:
                self.dtheta.copy_(part)
            else:
                self.dtheta.add_(part)
            first = False
        if first:
            self.dtheta.zero_()
        if self.need_coord_grad:
            self.dcoords = torch.zeros_like(self.coords)
            for g, r0, r1 in ((self.g_fwd_int, 0, ni), (self.g_fwd_bnd, ni, n)):
                if r1 > r0:
                    self.dcoords.add_(b.gno_aggregate_bwd_coords(g, self.theta, self.coords, self.x_ext, dz[r0:r1], d, H))
            s.edge_reduce(self.dcoords)
        red = s.allreduce_start(self.grad_flat)                                    # [dtheta | dW | db]: one collective
        if ni:
            b.gno_aggregate_bwd_x_pull(self.g_bwd_int, self.theta, self.coords, self.g_ext, d, H, Fi, out=self.dX[:ni])
        self.xchg_o.finish(reqs)
        if n - ni:
            b.gno_aggregate_bwd_x_pull(self.g_bwd_bnd, self.theta, self.coords, self.g_ext, d, H, Fi, out=self.dX[ni:])
        b.axpy(1.0, b.matmul_dx(self.W, dz, Fi), self.dX)
        if red is not None:
            red.wait()
        return self.dX

    def _backward_reduce(self, dz):
        s, b, n, ni = self.s, self.b, self.s.n, self.s.n_int
        d, H, Fi, Fo = self.d, self.H, self.Fi, self.Fo
        if self.use_bias:
            b.matmul_dw(self.ones, dz, out=self.db)
        b.matmul_dw(self.x_ext[:n], dz, out=self.dW)
        # the boundary block first: its rows are the only ones that touch remote columns
        wc = self.need_coord_grad
        dc_b = dc_i = None
        if n - ni:
            dx_b, dth_b, dc_b, _ = b.gno_aggregate_bwd(self.g_fwd_bnd, self.theta, self.coords, self.x_ext, dz[ni:], d, H, s_save=self._s[1],
                                                       need_dcoords=wc)
        else:           # a rank without boundary rows sends nothing but still receives what its peers computed for it
            dx_b = torch.zeros((n + s.n_halo, Fi), dtype=torch.float32, device=self.device)
            dth_b = None
        reqs = self.xchg_r.reduce_start(dx_b)                                      # remote rows on their way to their owners ...
        dth_i = None
        dx_i = None
        if ni:                                                                     # ... under the interior block
            dx_i, dth_i, dc_i, _ = b.gno_aggregate_bwd(self.g_fwd_int, self.theta, self.coords, self.x_ext, dz[:ni], d, H, s_save=self._s[0],
                                                       need_dcoords=wc)
        if wc:
            self.dcoords = torch.zeros_like(self.coords)
            for t in (dc_b, dc_i):
                if t is not None:
                    self.dcoords.add_(t)
            s.edge_reduce(self.dcoords)
        if dth_b is not None:
            self.dtheta.copy_(dth_b)
            if dth_i is not None:
                self.dtheta.add_(dth_i)
        elif dth_i is not None:
            self.dtheta.copy_(dth_i)
        else:
            self.dtheta.zero_()
        red = s.allreduce_start(self.grad_flat)                                    # [dtheta | dW | db]: one collective
        b.matmul_dx(self.W, dz, Fi, out=self.dX)                                   # the bypass
        self.dX.add_(dx_b[:n])
        if dx_i is not None:
            self.dX.add_(dx_i[:n])
        self.xchg_r.reduce_finish(self.dX, reqs)                                   # + what the peers computed for these rows
        if red is not None:
            red.wait()
        return self.dX

    def __call__(self, events=None):
        self.forward(events)
        return self.backward()


def measure_breakdown_gno(step, iters=3):
    """the parts of a GnoShardStep timed ALONE (device events, after the timed loop): both halo exchanges, the interior and the
    boundary launches of the forward aggregation, the reverse pass's local part (d theta of both blocks, dW, db) and its two
    pulls.  In the step the exchanges run under the interior work; timed apart, a scaling curve can be read."""
    s, b, n, ni = step.s, step.b, step.s.n, step.s.n_int
    d, H, Fi, Fo = step.d, step.H, step.Fi, step.Fo

    timed = _make_timed(s, iters)

    def halo():
        step.xchg.finish(step.xchg.start(step.x_ext))
        step.xchg_o.finish(step.xchg_o.start(step.g_ext))

    dz = step.g_ext[:n]

    def dtheta():
        for k, (g, r0, r1) in enumerate(((step.g_fwd_int, 0, ni), (step.g_fwd_bnd, ni, n))):
            if r1 > r0:
                b.gno_aggregate_bwd_theta(g, step.theta, step.coords, step.x_ext, dz[r0:r1], d, H, s_save=step._s[k])

    out = {"halo_ms": timed(halo, collective=True) if s.world > 1 else 0.0,
           "fwd_interior_ms": timed(lambda: step._fwd_block(0, step.g_fwd_int, 0, ni)),
           "fwd_boundary_ms": timed(lambda: step._fwd_block(1, step.g_fwd_bnd, ni, n)),
           "bwd_dtheta_ms": timed(dtheta),
           "bwd_pull_interior_ms": timed(lambda: b.gno_aggregate_bwd_x_pull(step.g_bwd_int, step.theta, step.coords, step.g_ext, d, H, Fi,
                                                                           out=step.dX[:ni])) if ni else 0.0,
           "bwd_pull_boundary_ms": timed(lambda: b.gno_aggregate_bwd_x_pull(step.g_bwd_bnd, step.theta, step.coords, step.g_ext, d, H, Fi,
                                                                           out=step.dX[ni:])) if n - ni else 0.0}
    if step.reverse == "reduce":
        # the reverse pass the step actually runs: each block's dx and dtheta from ONE contraction, remote rows reduced at their owners
        def red():
            buf = torch.zeros((n + s.n_halo, Fi), dtype=torch.float32, device=step.device)
            step.xchg_r.reduce_finish(step.dX, step.xchg_r.reduce_start(buf))
        out["halo_reduce_ms"] = timed(red, collective=True) if s.world > 1 else 0.0
        out["bwd_one_contraction_interior_ms"] = timed(lambda: b.gno_aggregate_bwd(step.g_fwd_int, step.theta, step.coords, step.x_ext, dz[:ni],
                                                                               d, H, s_save=step._s[0])) if ni else 0.0
        out["bwd_one_contraction_boundary_ms"] = timed(lambda: b.gno_aggregate_bwd(step.g_fwd_bnd, step.theta, step.coords, step.x_ext, dz[ni:],
                                                                               d, H, s_save=step._s[1])) if n - ni else 0.0
    recv = s.recv_rows * 4 * (Fi + (Fi if step.reverse == "reduce" else Fo))
    out["halo_recv_bytes_per_gpu_per_step"] = int(recv)
    out["xgmi_recv_GBps_per_gpu"] = (recv / (out["halo_ms"] * 1e-3) / 1e9) if out["halo_ms"] > 0 else None
    out["halo_mode"], out["halo_fraction"] = s.halo_mode, round(float(s.halo_fraction), 4)
    return out


# --------------------------------------------------------------------------------------------------
# data parallelism over independent graphs (Duvenaud / GNO mini-batches)
# --------------------------------------------------------------------------------------------------
def shard_graphs(graphs, rank, world):
    """the graphs of a batch are the sharding unit: rank r takes graphs r, r + world, ... (no halo, no
    collective on the data path); returns (its graphs, their positions in the batch)"""
    pos = list(range(rank, len(graphs), world))
    return [graphs[i] for i in pos], pos


def _c_allreduce_(t):
    """in-place float32 sum of a contiguous DEVICE tensor over the ranks through comm.hip's communicator (athena_mp_allreduce_start /
    _finish: RCCL on the communication stream, ordered behind and in front of torch's current stream; the host never blocks) --
    the same communicator the halo exchange uses, whatever backend torch.distributed's control-plane group runs on"""
    import ctypes as C
    from . import _capi
    comm = c_comm(t.device)
    _capi.use_torch_stream()
    _capi.call("athena_mp_allreduce_start", comm.handle, C.c_void_p(t.data_ptr()), t.numel())
    _capi.call("athena_mp_allreduce_finish", comm.handle)


def allreduce_layer_gradients(layers):
    """sum the parameter gradients of `layers` over all ranks in ONE bucketed all-reduce (the only collective a
    batch of independent graphs needs: dW / dR of the Duvenaud layer, dtheta / dW / db of the GNO layer), then
    hand each layer its slice back.  Gradients a rank did not produce count as zero.  Device tensors travel through the C
    ABI's communicator (RCCL); host tensors (the gloo CPU mirror) through torch.distributed."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    slots = [(l, i) for l in layers for i in range(len(l.params))]
    if not slots:
        return
    flat = torch.cat([(l.grads[i] if l.grads[i] is not None else torch.zeros_like(l.params[i])).reshape(-1)
                      for l, i in slots]).float().contiguous()
    if flat.is_cuda:
        _c_allreduce_(flat)
    else:
        dist.all_reduce(flat)
    off = 0
    for l, i in slots:
        n = l.params[i].numel()
        l.grads[i] = flat[off:off + n].view_as(l.params[i]).clone()
        off += n


def gather_graph_outputs(local_out, positions, n_graphs):
    """assemble the per-graph readout [batch, num_outputs] from the ranks' shards (rank r holds the rows
    `positions`); every rank receives the full tensor"""
    pos = torch.as_tensor(positions, dtype=torch.long)
    full = torch.zeros((n_graphs, local_out.shape[1]), dtype=local_out.dtype, device=local_out.device)
    full[pos.to(local_out.device)] = local_out.detach()
    if full.is_cuda and full.dtype == torch.float32:
        _c_allreduce_(full)        # disjoint rows: the sum is the gather (x + 0 is exact); nothing leaves the device
    else:
        dist.all_reduce(full)
    return full
