#!/usr/bin/env python3
"""bench.py -- msgpass fwd+bwd edges/sec on BASELINE.json configs[1]:
Kipf GCN layer, synthetic random graph 1M vertices / 10M CSR entries / 128 features, fp32.

A "step" is one interior Kipf layer forward+backward over the whole graph:
    P = A^ X ; Z = W P            (kipf_propagate, matmul)            one fused launch
    dW = dZ P^T                   (matmul reverse)                     MFMA reduction
    dX = A^T (W^T dZ)             (matmul reverse + get_partial_kipf_propagate_left_val -- reference:
                                   no coefficient), evaluated as (A^T dZ) W in one fused launch
"edges" = CSR entries (nnz).  Inputs are resident in HBM before the timed region.

`python3 bench.py --gpus N` launches itself: with WORLD_SIZE unset and N > 1 the parent starts
`python -m torch.distributed.run --nproc-per-node N ... bench.py` as a CHILD before anything touches the
GPU and exits with its code (a driver that starts the ranks itself sets WORLD_SIZE and skips this).

Workloads (--config):
  c2 (default)   N = 1: BASELINE configs[1].  N > 1: STRONG scaling of the SAME graph (SURVEY.md 8d generator:
                 uniform pairs, PCG64 seed 20260424) under a contiguous row partition -- the metric's
                 "1M-node/10M-edge, 1/2/4/8 GPU".  A uniformly random graph has no partition structure: every rank
                 needs most remote rows, so this line is xGMI bound by construction; the line says how much
                 (halo_ms / interior_ms / boundary_ms / bytes / achieved GB/s per GPU).
  c2-weak-sbm    WEAK scaling, 1M vertices / 10M entries per GPU, stochastic block model with a fixed inter-block
                 density (--cut8; what a node partitioner leaves on meshes and molecule batches).
  c5, c5-local   BASELINE configs[4]: 10M vertices / 150M entries / 256 features split over the ranks (8-way in the
                 config; any N that divides 10M runs); c5-local = the locality variant of SURVEY.md 8d
                 (|u-v| <= 50 000 with probability 0.95).
  c4-mesh        BASELINE configs[3] as a LAYER step at any N: graph_nop_layer forward + reverse (aggregation with the kernel
                 MLP, bypass W x + b) on the 2M-point radius mesh numbered in cell order, cut by rows
                 (athena_mp_shard_create_edges); a step = forward + reverse, "edges" = CSR entries.  Not the headline.
Every line carries "parity": the device results of the LAST timed step against the CPU oracle (N = 1: the whole
workload; N > 1: sampled rows of every rank + the transported halo rows against the generator); the process exits
non-zero when parity is above 1e-5 (or P is not bit-exact).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the pool's host driver only supports dmabuf IPC: without this RCCL's peer-to-peer setup between the ranks' processes fails
# with "hipIpcGetMemHandle: invalid argument" (already exported on the GPU boxes; kept here for launchers that scrub the env)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md "Chip-level parameters")
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32-input MFMA peak (MI355X_MICROARCH.md)
TOL = 1e-5             # north_star: 1e-5 relative fp32

CONFIGS = {
    "c2": dict(nodes=1_000_000, pairs=4_500_000, feat=128, locality=None),
    "c2-weak-sbm": dict(nodes=1_000_000, pairs=4_500_000, feat=128, locality=None),
    "c5": dict(nodes=10_000_000, pairs=70_000_000, feat=256, locality=None),
    "c5-local": dict(nodes=10_000_000, pairs=70_000_000, feat=256, locality=(50_000, 0.95)),
    # BASELINE configs[3]: graph_nop_layer fwd+bwd on the 2 M-point radius mesh (points numbered in cell order), row-partitioned
    # over the ranks (athena_mp_shard_create_edges; halo of x and of dz, [dtheta | dW | db] all-reduced); --nodes = points
    "c4-mesh": dict(nodes=2_000_000, pairs=0, feat=64, locality=None),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--nodes", type=int, default=None, help="vertices (c2-weak-sbm: per GPU; otherwise of the whole graph)")
    ap.add_argument("--pairs", type=int, default=None, help="undirected pairs (nnz = 2*pairs + nodes)")
    ap.add_argument("--feat", type=int, default=None)
    ap.add_argument("--cut8", dest="cut", type=float, default=0.05,
                    help="c2-weak-sbm: fraction of undirected pairs crossing partitions at 8 blocks (fixed inter-block "
                         "density; the fraction at N blocks is cut8*(N-1)/7).  -1: structure-free uniform random graph")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle (no cpu_baseline, N=1 parity null)")
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000,
                    help="rows of the workload the 1-thread CPU oracle runs (default: all of C2, ~7 s per run)")
    ap.add_argument("--cpu-runs", type=int, default=5, help="timed runs of the CPU oracle after one warm-up (median reported)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="N = 1: skip the `secondary` block (configs[2], configs[3], configs[4] on one GPU; ~2-4 min of child processes)")
    ap.add_argument("--secondary", default="c3,c4,c5,kb", help="which secondary configurations to run (comma separated)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    args.custom = any(v is not None for v in (args.nodes, args.pairs, args.feat))
    args.nodes = cfg["nodes"] if args.nodes is None else args.nodes
    args.pairs = cfg["pairs"] if args.pairs is None else args.pairs
    args.feat = cfg["feat"] if args.feat is None else args.feat
    args.locality = cfg["locality"]
    return args


def self_launch(args):
    """N > 1 from a bare shell: start the ranks as a child process (never exec: nothing here has touched the GPU, and
    the children are fresh processes)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def _stall_for_test(rank, where):
    """test hook (tests/test_gpu_dist.py): ATHENA_MP_BENCH_STALL="<rank>:<where>" makes that rank sleep at `where`, so the
    OTHER ranks' watchdogs can be seen to fire.  Never set outside the tests."""
    spec = os.environ.get("ATHENA_MP_BENCH_STALL", "")
    if spec and spec == f"{rank}:{where}":
        time.sleep(3600)


def init_control_plane(dev, one_device):
    """N > 1: the CONTROL plane of the bench (barriers around the timed loop, the max of the step time, scalar agreement, the id
    broadcast that seeds comm.hip's communicator) is a gloo group by default, so that librccl holds exactly ONE communicator per
    rank -- comm.hip's, the data plane (halo exchange, dW all-reduce).  ATHENA_MP_BENCH_CONTROL=nccl opts into torch's RCCL
    group instead (a second communicator on the same devices).  The one-device dry run asks for comm.hip's host-staged TEST
    transport by name; nothing infers the data plane's transport from the control plane's backend.  Returns the backend."""
    import torch.distributed as dist

    control = os.environ.get("ATHENA_MP_BENCH_CONTROL") or os.environ.get("ATHENA_MP_BENCH_BACKEND") or "gloo"
    if control not in ("gloo", "nccl"):
        sys.exit(f"bench.py: ATHENA_MP_BENCH_CONTROL={control}: gloo or nccl")
    if one_device:
        os.environ.setdefault("ATHENA_MP_COMM_TRANSPORT", "shm")
    if control == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost") and os.path.isdir("/sys/class/net/lo"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")       # one node: the container's hostname need not resolve
        dist.init_process_group("gloo")
    return control


def control_plane_note(control):
    return (f"{control} (torch.distributed): barriers, max of the step time, scalar agreement, communicator id broadcast; data plane "
            f"= comm.hip's own communicator" + ("" if control == "gloo" else " -- beside torch's RCCL group: two communicators per rank"))


def transport_error(transport, one_device):
    """N > 1 lines must have moved their halos over RCCL.  comm.hip's host-staged test transport ("shm ...") and the
    python fallback plan exist for boxes with one GPU; a line that used them without the one-device dry-run switch
    (say, a stray ATHENA_MP_COMM_TRANSPORT on a real node) is an error, not a slow measurement."""
    if transport.startswith("rccl") or one_device:
        return None
    return (f"transport is '{transport}', not RCCL: only the one-device dry run (ATHENA_MP_BENCH_ONE_DEVICE=1) may use "
            "the test transport")


def rel(a, b):
    import numpy as np
    b = np.asarray(b, np.float64)
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


def cpu_baseline(ia, ja, x, w, dz, F, sample_rows, runs=3):
    """The oracle (a line-by-line C port of the reference's loops), 1 thread, on the first
    `sample_rows` rows of the same workload (they gather from / scatter into the full tensors).
    Returns (the cpu_baseline object, the oracle's results for the parity block)."""
    import numpy as np
    from oracle import oracle

    n = ia.size - 1
    rows = min(sample_rows, n)
    sia = ia[: rows + 1].copy()
    sja = np.asfortranarray(ja[:, : sia[-1] - 1])
    deg = np.diff(ia).astype(np.int32)
    ent = int(sia[-1] - 1)
    def one_run(r, keep):
        ria = sia[: r + 1]
        rja = sja[:, : ria[-1] - 1]
        t0 = time.perf_counter()
        p = oracle.kipf_propagate_rect(x, ria, rja, deg[:r], deg)
        z = oracle.matmul(w, p, F)
        dw = oracle.matmul_dw(dz[:r], p)
        dp = oracle.matmul_dx(w, dz[:r], F)
        dx = oracle.kipf_propagate_bwd(dp, ria, rja, n_out=n)
        return time.perf_counter() - t0, (dict(p=p, z=z, dw=dw, dx=dx) if keep else None)

    one_run(min(rows, 50_000), False)                 # warm-up: library, page tables, caches ("after 1 warm-up")
    times, res = [], None
    for k in range(max(1, runs)):
        t, r_ = one_run(rows, k == 0)
        times.append(t)
        res = res or r_
    t = float(np.median(times))
    p, z, dw, dx = res["p"], res["z"], res["dw"], res["dx"]
    out = {"value": ent / t, "unit": "edges/s", "cores": 1, "kind": "port",
           "sample": f"oracle (C port of the reference loops), first {rows} of {n} rows = {ent} of {ja.shape[1]} entries, "
                     f"full fwd+bwd step, median of {len(times)} timed runs after 1 warm-up "
                     f"({', '.join(f'{v:.2f}' for v in times)} s) on {os.cpu_count()}-core host, 1 thread",
           "runs_s": [round(v, 3) for v in times]}
    # context only: the same step threaded over rows on ALL host cores.  The reference itself has no
    # threading, so the single-thread figure above is the baseline; this one shows what the whole socket pair reaches.
    try:
        best = None
        for _ in range(3):
            *_, dt, threads = oracle.omp_kipf_step(x, w, dz, ia, ja)
            best = dt if best is None else min(best, dt)
        out["all_cores_context"] = {"value": ja.shape[1] / best, "unit": "edges/s", "cores": threads,
                                    "kind": "port, OpenMP over rows, pull-form backward (not the reference's algorithm)",
                                    "sample": f"whole workload, best of 3, {best * 1e3:.0f} ms"}
    except Exception as exc:   # no OpenMP runtime on the host: the contract fields above are complete without it
        out["all_cores_context"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    return out, dict(rows=rows, full=(rows == n), p=p, z=z, dw=dw, dx=dx, w=w, ia=ia, ja=ja)


def parity_single(ref, P, Z, dW, dX, dz_host):
    """device results of the last timed step against the oracle's (whole workload when the oracle ran all rows).
    dW is a sum over 10^6 vertices: the oracle adds them sequentially in fp32 and is itself ~3e-5 from the exact sum,
    so dW passes when it is within 1e-5 of the oracle OR no further from float64 than the oracle is plus 1e-5
    (|gpu - f64| <= |oracle - f64| + 1e-5 scale); all three distances are reported."""
    import numpy as np
    r = ref["rows"]
    out = {"against": "oracle (C port of the reference loops), " + ("all rows" if ref["full"] else f"first {r} rows (dW, dX need all rows: null)"),
           "P_bit_exact": bool(np.array_equal(P[:r].cpu().numpy(), ref["p"])),
           "Z_rel": rel(Z[:r].cpu().numpy(), ref["z"]),
           "dW_rel": None, "dW_rel_vs_float64": None, "dW_oracle_rel_vs_float64": None,
           "dX_rel": rel(dX.cpu().numpy(), ref["dx"]) if ref["full"] else None, "tol": TOL}
    dw_ok = True
    if ref["full"]:
        dw = dW.cpu().numpy()
        f64 = (ref["p"].astype(np.float64).T @ dz_host.astype(np.float64)).reshape(-1)   # dW(Fo,Fi) column-major = P^T dZ
        out["dW_rel"], out["dW_rel_vs_float64"], out["dW_oracle_rel_vs_float64"] = rel(dw, ref["dw"]), rel(dw, f64), rel(ref["dw"], f64)
        dw_ok = out["dW_rel"] <= TOL or out["dW_rel_vs_float64"] <= out["dW_oracle_rel_vs_float64"] + TOL
    # ELEMENT-WISE: every element of the contraction outputs against the magnitude of ITS OWN terms --
    # |gpu - oracle| <= 1e-5 * sum_k |term_k| -- so that a wrong small element cannot hide behind the tensor's maximum.
    # Z[v,o] = sum_i P[v,i] Wt[i,o]: terms |P| |Wt|.  dX[u] = sum over the entries that list u of (dZ W)[v]: terms
    # A^T (|dZ| |W|) (float64 BLAS / scipy sparse: the magnitudes need no particular summation order).  dW sums 10^6 terms
    # per element; there the fp32 oracle itself is the outlier, so the device is held to float64 instead.
    out["Z_elementwise_worst"] = out["dX_elementwise_worst"] = out["dW_elementwise_worst_vs_float64"] = None
    ew_ok = True
    try:
        if "w" in ref:
            F = ref["p"].shape[1]
            wt_abs = np.abs(ref["w"].astype(np.float64)).reshape(F, -1)       # Wt[i][o] = params.val[o + Fo i]
            z_mag = np.abs(ref["p"].astype(np.float64)) @ wt_abs
            out["Z_elementwise_worst"] = float((np.abs(Z[:r].cpu().numpy().astype(np.float64) - ref["z"]) / np.maximum(z_mag, 1e-30)).max())
            if ref["full"]:
                import scipy.sparse as sp
                dp_mag = np.abs(dz_host.astype(np.float64)) @ wt_abs.T        # |dZ| |W| : [N, Fi]
                ia, ja = ref["ia"], ref["ja"]
                n = ia.size - 1
                A = sp.csr_matrix((np.ones(ja.shape[1]), ja[0].astype(np.int64) - 1, ia.astype(np.int64) - 1), shape=(n, n))
                dx_mag = A.T @ dp_mag                                          # the coefficient-free scatter of the reference
                out["dX_elementwise_worst"] = float((np.abs(dX.cpu().numpy().astype(np.float64) - ref["dx"]) / np.maximum(dx_mag, 1e-30)).max())
                dw_mag = (np.abs(ref["p"].astype(np.float64)).T @ np.abs(dz_host.astype(np.float64))).reshape(-1)
                f64 = (ref["p"].astype(np.float64).T @ dz_host.astype(np.float64)).reshape(-1)
                out["dW_elementwise_worst_vs_float64"] = float((np.abs(dW.cpu().numpy().astype(np.float64) - f64) / np.maximum(dw_mag, 1e-30)).max())
            ew_ok = all(v is None or v <= TOL for v in (out["Z_elementwise_worst"], out["dX_elementwise_worst"], out["dW_elementwise_worst_vs_float64"]))
    except Exception as exc:      # scipy missing on the host: the tensor-level fields above stand on their own
        out["elementwise_error"] = f"{type(exc).__name__}: {exc}"[:200]
    out["ok"] = bool(out["P_bit_exact"] and dw_ok and ew_ok and all(v is None or v <= TOL for v in (out["Z_rel"], out["dX_rel"])))
    return out


def parity_sharded(step, shard, seeds, dev, n_sample=4000):
    """N > 1: (1) the halo rows the exchanges delivered against the generator (transport, bit-exact); (2) sampled rows
    of this rank: P bit-exact / Z / dX <= 1e-5 against the oracle on the compacted sub-problem; (3) the all-reduced dW
    against a float64 contraction on the device (a yardstick, not the oracle: the oracle would need every rank's P)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from athena_amd import synth
    from oracle import oracle

    s, F, Fo, n = shard, step.F, step.Fo, shard.n
    rng = np.random.default_rng(100 + s.rank)
    res = {}
    torch.cuda.synchronize()
    # (1) transported rows
    ok_halo = True
    held = np.flatnonzero(s.ext_ids >= 0)         # rows behind the local ones that hold a vertex (all-gather layout: not the padding slots)
    if held.size and seeds is not None:
        k = np.sort(rng.choice(held, min(500, held.size), replace=False))
        kd = torch.from_numpy(n + k).to(dev)
        ok_halo = bool(np.array_equal(step.x_ext[kd].cpu().numpy(), synth.feature_rows(seeds[0], s.ext_ids[k], F)) and
                       np.array_equal(step.dZ_ext[kd].cpu().numpy(), synth.feature_rows(seeds[1], s.ext_ids[k], Fo)) and
                       np.isin(s.halo_ids, s.ext_ids[held]).all())
    res["halo_rows_bit_exact"] = ok_halo
    # (2) sampled rows
    rows = np.sort(rng.choice(n, min(n_sample, n), replace=False))
    W = step.W.cpu().numpy()

    def sub(backward):
        ia, nb = s.csr(backward)                      # the shard's own CSR ([local | halo] numbering), read back
        ent = np.concatenate([np.arange(ia[r] - 1, ia[r + 1] - 1) for r in rows])
        cols, inv = np.unique(nb[ent].astype(np.int64) - 1, return_inverse=True)
        sia = np.concatenate([[1], 1 + np.cumsum(ia[rows + 1] - ia[rows])]).astype(np.int32)
        sja = np.zeros((2, ent.size), np.int32, order="F"); sja[0] = inv + 1
        return cols, sia, sja

    rsel = torch.from_numpy(rows).to(dev)
    cols, sia, sja = sub(False)
    xc = step.x_ext[torch.from_numpy(cols).to(dev)].cpu().numpy()
    p_ref = oracle.kipf_propagate_rect(xc, sia, sja, s.row_deg[rows], s.col_deg[cols])
    res["P_bit_exact"] = bool(np.array_equal(step.P[rsel].cpu().numpy(), p_ref))
    res["Z_rel"] = rel(step.Z[rsel].cpu().numpy(), oracle.matmul(W, p_ref, Fo))
    cols, sia, sja = sub(True)
    dzc = step.dZ_ext[torch.from_numpy(cols).to(dev)].cpu().numpy()
    dp = oracle.matmul_dx(W, dzc, F)                                            # reference order: W^T dZ, then the scatter
    ones = np.ones(max(rows.size, cols.size), np.int32)
    dx_ref = oracle.kipf_propagate_rect(dp, sia, sja, ones[:rows.size], ones[:cols.size])   # coefficient 1: plain sums
    res["dX_rel"] = rel(step.dX[rsel].cpu().numpy(), dx_ref)
    # (3) dW
    d64 = (step.P.double().T @ step.dZ.double()).reshape(-1)                     # dW(Fo,Fi) column-major == [Fi][Fo] row-major = P^T dZ
    if dist.get_backend() == "nccl":
        dist.all_reduce(d64)
    else:
        h = d64.cpu(); dist.all_reduce(h); d64 = h.to(dev)
    res["dW_rel_vs_float64"] = float((step.dW.double() - d64).abs().max().item() / max(d64.abs().max().item(), 1e-30))
    flags = torch.tensor([float(not (res["halo_rows_bit_exact"] and res["P_bit_exact"])), res["Z_rel"], res["dX_rel"],
                          res["dW_rel_vs_float64"]], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        flags = flags.to(dev)
    dist.all_reduce(flags, op=dist.ReduceOp.MAX)
    flags = flags.cpu().tolist()
    out = {"against": f"oracle on {rows.size} sampled rows of every rank (max over ranks); halo rows against the generator; "
                      "dW against float64 on the device",
           "halo_rows_bit_exact": flags[0] == 0.0 and res["halo_rows_bit_exact"], "P_bit_exact": flags[0] == 0.0,
           "Z_rel": flags[1], "dX_rel": flags[2], "dW_rel_vs_float64": flags[3], "tol": TOL}
    out["ok"] = bool(flags[0] == 0.0 and max(flags[1:]) <= TOL)
    return out


def measure_copy_ceiling(dev, gib=1, reps=10):
    """the streaming ceiling of THIS box, measured in this run: a 1 GiB device copy, one 16-byte element per thread in
    launch order (athena_mp_device_copy: the form that reaches the ceiling, profiles/r03_ubench_stream_rows.txt), read +
    written bytes over the mean launch time of `reps` launches after one warm-up (HIP events on the launch stream)"""
    import ctypes as C

    import torch
    from athena_amd import _capi

    n = gib << 30
    src = torch.empty(n, dtype=torch.uint8, device=dev)
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    src.zero_(); dst.zero_()
    _capi.use_torch_stream()
    cp = lambda: _capi.call("athena_mp_device_copy", C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), n)
    cp()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        cp()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del src, dst
    torch.cuda.empty_cache()
    return 2.0 * n / (ms * 1e-3) / 1e9


def run_secondary(which, budget_s=540.0):
    """configs[2], configs[3] and configs[4]-on-one-GPU as CHILD processes (scripts/bench_secondary.py) after the
    headline's timed loop: step ms, per-op roofline fractions and a parity flag against the oracle each.  Anything that
    goes wrong in a child -- a crash, a timeout, a parity failure -- is recorded here and changes neither the headline
    nor the exit code."""
    out = {"note": "measured after the headline's timed loop, each configuration in a child process of its own; "
                   "never part of `value`"}
    names = {"c3": "configs[2]", "c4": "configs[3]", "c5": "configs[4]_one_gpu", "kb": "kipf_block_diagonal_batch"}
    t_start = time.perf_counter()
    for key in which:
        if key not in names:
            continue
        left = budget_s - (time.perf_counter() - t_start)
        if left < 30:
            out[names[key]] = {"error": "skipped: the secondary block's time budget is spent"}
            continue
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bench_secondary.py"), "--config", key],
                               capture_output=True, text=True, timeout=left)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                out[names[key]] = {"error": f"rc {r.returncode}: " + (r.stderr.strip().splitlines() or ["no output"])[-1][:300]}
            else:
                out[names[key]] = json.loads(lines[-1])
        except subprocess.TimeoutExpired:
            out[names[key]] = {"error": f"timed out after {left:.0f} s"}
        except Exception as exc:      # a broken child must never take the headline with it
            out[names[key]] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    bad = [k for k, v in out.items() if isinstance(v, dict) and ("error" in v or not v.get("parity", {}).get("ok", False))]
    if bad:
        out["error"] = "failed or parity not ok: " + ", ".join(bad)
    out["wall_s"] = round(time.perf_counter() - t_start, 1)
    return out


def main_gno(args, world, rank, dev, one_device):
    """--config c4-mesh: graph_nop_layer forward + reverse on the configs[3] mesh, row-partitioned over the ranks
    (dist.GnoShardStep over athena_mp_shard_create_edges).  One json line: whole-job entries/s, the interior forward launch
    against the fp32-MFMA peak, parity of sampled rows of every rank against the MATERIALISING oracle, the halo's share."""
    import contextlib

    import numpy as np
    import torch
    import torch.distributed as dist

    from athena_amd import dist as adist
    from athena_amd import synth

    Fi = Fo = H = args.feat
    d = 3
    watch = None
    backend = "none"
    if world > 1:
        os.environ.setdefault("ATHENA_MP_COLLECTIVE_TIMEOUT_S", "120")   # the bench's own deadline, for the python watchdog AND
        watch = adist.Watchdog(rank)                                     # comm.hip's monitor (library default: 1800 s)
        adist.set_watchdog(watch)
        with watch.phase("process group creation"):
            backend = init_control_plane(dev, one_device)

    def phase(name, factor=1.0):
        return watch.phase(name, factor) if watch is not None else contextlib.nullcontext()

    N = args.nodes
    mesh = synth.radius_graph(N, order="cells")                  # every rank derives the same mesh from the seed
    ia, ja, coords = mesh
    if world > 1:
        shard, c_loc = adist.make_mesh_shard(rank, world, N, device=dev, mesh=mesh)
    else:
        # one rank: the same code path with an empty halo (python plan; nothing to exchange)
        rows = np.repeat(np.arange(N, dtype=np.int64), np.diff(ia))
        shard = adist.build_plan(adist.Shard(0, 1, N, rows, ja[0].astype(np.int64) - 1, ja[1].astype(np.int64)), dev)
        shard.n_total = N
        c_loc = np.ascontiguousarray(coords[shard.edge_ids], np.float32)
    n = shard.n
    rng = np.random.Generator(np.random.PCG64(7))                 # the global problem, the same on every rank
    theta = (0.3 * rng.standard_normal(H * d + H + Fo * Fi * H + Fo * Fi)).astype(np.float32)
    w = (rng.standard_normal(Fo * Fi) * np.sqrt(2.0 / Fi)).astype(np.float32)
    b = (rng.standard_normal(Fo) * 0.1).astype(np.float32)
    lo = rank * n
    x_l = synth.feature_block(1, lo, lo + n, Fi)
    up_l = synth.feature_block(3, lo, lo + n, Fo)
    step = adist.GnoShardStep(shard, Fi, Fo, d, H, dev, inputs=(x_l, up_l, theta, w, b, c_loc))
    nnz_local = int(shard.nnz)
    tt = torch.tensor([nnz_local], dtype=torch.int64, device=dev if (world > 1 and backend == "nccl") else "cpu")
    if world > 1:
        adist.first_contact(step, watch, torch.cuda.synchronize)
        with phase("all-reduce of the entry counts"):
            dist.all_reduce(tt)
    nnz_total = int(tt.item())

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ev = []
    with phase("warm-up steps"):
        for _ in range(args.warmup):
            step()
        barrier()
    with phase("timed loop", factor=max(1.0, args.steps / 10.0)):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(events=ev)
        barrier()
        dt = time.perf_counter() - t0
    if world > 1:
        t_ = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        with phase("max of the step time over the ranks"):
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
        dt = t_.item()
    ms = dt / args.steps * 1e3
    # roofline: the interior rows' forward aggregation launch of rank 0 (fp32 MFMA; SURVEY.md 8d's flop count)
    ni = shard.n_int
    int_nnz = int(step.g_fwd_int.nnz)
    fwd_ms = float(np.mean([a.elapsed_time(b_) for a, b_ in ev])) if ev else float("nan")
    flops = int_nnz * 2 * H * (Fi + d) + ni * 2 * Fo * (H + 1) * Fi
    tf = flops / (fwd_ms * 1e-3) / 1e12
    # parity: sampled rows of both blocks of `out` and of dx against the materialising oracle on the compact sub-problem
    from oracle import oracle
    r2 = np.random.default_rng(100 + rank)
    pick = [r2.choice(ni, min(100, ni), replace=False)] if ni else []
    if n - ni:
        pick.append(ni + r2.choice(n - ni, min(100, n - ni), replace=False))
    rows = np.unique(np.concatenate(pick))
    rsel = torch.from_numpy(rows).to(dev)

    def sub(backward):
        gi, gb = (step.g_bwd_int, step.g_bwd_bnd) if backward else (step.g_fwd_int, step.g_fwd_bnd)
        rp = np.concatenate([gi.export("rowptr"), gb.export("rowptr")[1:] + gi.nnz]).astype(np.int64)
        col = np.concatenate([gi.export("col"), gb.export("col")])
        eid = np.concatenate([gi.export("eid"), gb.export("eid")])
        ent = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in rows])
        cols, cinv = np.unique(col[ent], return_inverse=True)
        ecols, einv = np.unique(eid[ent], return_inverse=True)
        sia = np.concatenate([[1], 1 + np.cumsum(rp[rows + 1] - rp[rows])]).astype(np.int32)
        sja = np.zeros((2, ent.size), np.int32, order="F"); sja[0] = cinv + 1; sja[1] = einv + 1
        return cols, ecols, sia, sja

    torch.cuda.synchronize()
    if world > 1:
        # the checker needs dz of the remote neighbours of the sampled rows; the step itself may not have moved them (its
        # scatter-form reverse pass sends dx rows to their owners instead): one exchange of dz, for the check only
        with phase("halo exchange of dz for the parity check"):
            step.xchg_o.finish(step.xchg_o.start(step.g_ext))
            torch.cuda.synchronize()
    cols, ecols, sia, sja = sub(False)
    kap = oracle.gno_kernel_eval(c_loc[ecols], theta, H, Fo * Fi)
    nsq = max(rows.size, cols.size)
    xs = np.zeros((nsq, Fi), np.float32); xs[:cols.size] = step.x_ext[torch.from_numpy(cols).to(dev)].cpu().numpy()
    sia_sq = np.concatenate([sia, np.full(nsq - rows.size, sia[-1], np.int32)])
    m_ref = oracle.gno_aggregate(xs, kap, sia_sq, sja, Fo)[:rows.size]
    x_rows = step.x_ext[rsel].cpu().numpy()
    out_ref = oracle.add_bias_rows(m_ref + oracle.matmul(w, x_rows, Fo), b)
    res = {"out_rel": rel(step.out[rsel].cpu().numpy(), out_ref)}
    cols, ecols, sia, sja = sub(True)
    kap64 = oracle.gno_kernel_eval(c_loc[ecols], theta, H, Fo * Fi).reshape(-1, Fi, Fo).astype(np.float64)
    g_rows = step.g_ext if world > 1 else step.dz          # one rank: dz never went into the exchange buffer
    gs = g_rows[torch.from_numpy(cols).to(dev)].cpu().numpy().astype(np.float64)
    dref = np.zeros((rows.size, Fi))
    for k in range(rows.size):
        for w_ in range(sia[k] - 1, sia[k + 1] - 1):
            dref[k] += kap64[sja[1, w_] - 1] @ gs[sja[0, w_] - 1]
    dref += oracle.matmul_dx(w, g_rows[rsel].cpu().numpy(), Fi).astype(np.float64)
    res["dX_rel"] = rel(step.dX[rsel].cpu().numpy(), dref)
    # ... and independently of the pull-form graphs the step itself uses: dx of sampled INTERIOR vertices as the
    # reference forms it -- a SCATTER over the FORWARD graph, dx[u] += K_e^T g[v] for every entry (u, e) of every row v
    # (athena_diffstruc_extd_sub_nop.f90:419-458).  An interior vertex is referenced by local rows only, so the rank's own
    # forward blocks hold every entry that scatters into it; a symmetry bug in the backward blocks would show here.
    res["dX_scatter_rel"] = 0.0
    if ni:
        ucols = np.sort(r2.choice(ni, min(60, ni), replace=False))
        gi, gb = step.g_fwd_int, step.g_fwd_bnd
        rp = np.concatenate([gi.export("rowptr"), gb.export("rowptr")[1:] + gi.nnz]).astype(np.int64)
        col = np.concatenate([gi.export("col"), gb.export("col")])
        eid = np.concatenate([gi.export("eid"), gb.export("eid")])
        hit = np.flatnonzero(np.isin(col, ucols) & (eid >= 0))
        vrow = np.searchsorted(rp, hit, side="right") - 1                  # the row each such entry sits in
        ecs, einv = np.unique(eid[hit], return_inverse=True)
        kap_s = oracle.gno_kernel_eval(c_loc[ecs], theta, H, Fo * Fi).reshape(-1, Fi, Fo).astype(np.float64)
        gv = g_rows[torch.from_numpy(vrow).to(dev)].cpu().numpy().astype(np.float64)
        dsc = np.zeros((ucols.size, Fi))
        pos = np.searchsorted(ucols, col[hit])
        for k in range(hit.size):
            dsc[pos[k]] += kap_s[einv[k]] @ gv[k]
        usel = torch.from_numpy(ucols).to(dev)
        dsc += oracle.matmul_dx(w, g_rows[usel].cpu().numpy(), Fi).astype(np.float64)
        res["dX_scatter_rel"] = rel(step.dX[usel].cpu().numpy(), dsc)
    held = np.flatnonzero(shard.ext_ids >= 0)
    halo_ok = True
    if held.size:
        k = np.sort(r2.choice(held, min(300, held.size), replace=False))
        halo_ok = bool(np.array_equal(step.x_ext[torch.from_numpy(n + k).to(dev)].cpu().numpy(), synth.feature_rows(1, shard.ext_ids[k], Fi)))
    flags = torch.tensor([0.0 if halo_ok else 1.0, res["out_rel"], res["dX_rel"], res["dX_scatter_rel"]], dtype=torch.float64)
    if world > 1:
        if backend == "nccl":
            flags = flags.to(dev)
        with phase("parity of every rank against the oracle", factor=3.0):
            dist.all_reduce(flags, op=dist.ReduceOp.MAX)
    flags = flags.cpu().tolist()
    parity = {"against": f"materialising oracle on {rows.size} sampled rows of both blocks of every rank (max over ranks); halo rows "
                         "against the generator; the all-reduced gradients are held by tests/test_gpu_dist.py, not here",
              "halo_rows_bit_exact": flags[0] == 0.0, "out_rel": flags[1], "dX_rel": flags[2],
              "dX_rel_vs_scatter_over_the_forward_graph": flags[3], "tol": TOL}
    parity["ok"] = bool(flags[0] == 0.0 and max(flags[1:]) <= TOL)
    out = {"metric": "msgpass fwd+bwd edges/sec", "value": nnz_total * args.steps / dt, "unit": "edges/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
           "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"BASELINE configs[3] as a layer step: graph_nop_layer fwd+bwd, radius mesh {N} vertices / {nnz_total} CSR "
                                  f"entries in cell order, {Fi} features, d = 3, H = {H}, fp32"
                                  + (f", row-partitioned over {world} GPUs" if world > 1 else ""),
                      "vertices": N, "entries": nnz_total, "entries_per_gpu": nnz_local, "features": Fi,
                      "parallelism": f"row-partition x{world}" if world > 1 else "single GPU",
                      "halo_rows_per_gpu": int(shard.halo_ids.size), "halo_mode": shard.halo_mode,
                      "halo_fraction": round(float(shard.halo_fraction), 4), "interior_rows_per_gpu": int(ni),
                      "edge_columns_per_gpu": int(shard.n_edge_cols), "transport": shard.transport, "reverse_pass": step.reverse,
                      "S_kept_per_block": [t is not None for t in step._s]},
           "roofline": {"bound": "mfma", "kernel": "gno_pc_kernel<true> (gno_aggregate of rank 0's interior rows, S kept; the halo exchange in flight)",
                        "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS,
                        "traffic": None, "flops_per_launch": flops, "avg_launch_ms": fwd_ms},
           "parity": parity}
    ok = parity["ok"]
    if world > 1:
        with phase("breakdown (each part timed alone)", factor=2.0):
            out["breakdown"] = adist.measure_breakdown_gno(step)
        out["breakdown"]["note"] = "each part timed alone after the timed loop, rank 0 (events); in the step the exchanges run under the interior work"
        out["rccl"] = adist.c_comm_stats()      # rank 0's communicator: ranks RCCL itself counts, its version, bytes to every peer
        out["config"]["control_plane"] = control_plane_note(backend)
        err = transport_error(str(shard.transport), one_device)
        if err:
            out["ok"], out["error"], ok = False, err, False
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        with phase("final barrier"):
            dist.barrier()
        watch.close()
        adist.c_comm_destroy()
        dist.destroy_process_group()
    if not ok:
        sys.exit("bench.py: parity against the oracle FAILED (see the 'parity' object of the line above)")


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    # dev aid: ATHENA_MP_BENCH_ONE_DEVICE=1 runs the N > 1 code path with all ranks on
    # device 0 and host-staged transport (a dry run of the partition / halo / overlap logic on a 1-GPU box; the
    # numbers it prints mean nothing).  The driver's runs use one GPU per rank over RCCL.
    one_device = bool(os.environ.get("ATHENA_MP_BENCH_ONE_DEVICE"))
    if one_device:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        sys.exit(f"bench.py: {world} ranks but {torch.cuda.device_count()} GPU(s) visible "
                 "(dry run on one device: ATHENA_MP_BENCH_ONE_DEVICE=1)")
    if not torch.cuda.is_available():
        sys.exit("bench.py measures the HIP path; no GPU visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from athena_amd import DeviceGraph, _capi, ops, synth

    _capi.init(local_rank)
    if args.config == "c4-mesh":
        return main_gno(args, world, rank, dev, one_device)
    F = args.feat
    weak = args.config == "c2-weak-sbm"
    ev, ev_bnd, ev_dw, ev_bwd = [], [], [], []
    breakdown, seeds = None, None

    if world > 1:
        import torch.distributed as dist

        from athena_amd import dist as adist

        # every wait for a peer from here on has a deadline (ATHENA_MP_COLLECTIVE_TIMEOUT_S, default 120 s per phase): a
        # stalled rank prints {"ok": false, "error": "rank r stalled in <phase>"} and leaves with exit code 3
        os.environ.setdefault("ATHENA_MP_COLLECTIVE_TIMEOUT_S", "120")   # the bench's own deadline, for the python watchdog AND
        watch = adist.Watchdog(rank)                                     # comm.hip's monitor (library default: 1800 s)
        adist.set_watchdog(watch)
        with watch.phase("process group creation"):
            backend = init_control_plane(dev, one_device)
        _stall_for_test(rank, "setup")
        if weak:
            cut = None if args.cut < 0 else args.cut * (world - 1) / 7.0
            shard = adist.make_weak_scaling_shard(rank, world, args.nodes, args.pairs, F, cut=cut, device=dev)
            inputs = None
        else:
            shard = adist.make_global_shard(rank, world, args.nodes, args.pairs, device=dev, locality=args.locality)
            n = shard.n
            seeds = (1, 3)                                    # synth.kipf_inputs: X seed 1, W seed 2, dZ seed 3
            inputs = (synth.feature_block(1, rank * n, (rank + 1) * n, F), synth.feature_block(3, rank * n, (rank + 1) * n, F),
                      synth.kipf_weight(F))
        shard_step, nnz_local, info = adist.build_kipf_step(shard, F, dev, inputs=inputs)
        _stall_for_test(rank, "halo")
        adist.first_contact(shard_step, watch, torch.cuda.synchronize)
        tt = torch.tensor([nnz_local], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        with watch.phase("all-reduce of the entry counts"):
            dist.all_reduce(tt)
        nnz_total = int(tt.item())
        x = w = dz = ia = ja = None

        def step(record=False):
            shard_step(events=ev if record else None, events_bnd=ev_bnd if record else None)
    else:
        ia, ja = synth.random_graph_csr(args.nodes, args.pairs) if args.locality is None else _global_csr(synth, args)
        x, w, dz = synth.kipf_inputs(args.nodes, F)
        g = DeviceGraph(ia, ja, n_edge_cols=0, device=local_rank)
        nnz_local = nnz_total = int(ja.shape[1])
        xd, wd, dzd = (torch.from_numpy(t).to(dev) for t in (x, w, dz))
        N = args.nodes
        P = torch.empty((N, F), device=dev)
        Z = torch.empty((N, F), device=dev)
        dW = torch.empty(F * F, device=dev)
        dX = torch.empty((N, F), device=dev)

        def step(record=False):
            if record:
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
            ops.kipf_layer_fwd(g, xd, wd, F, P=P, Z=Z)
            if record:
                e1.record(); ev.append((e0, e1))
            ops.matmul_dw(P, dzd, out=dW)
            if record:
                e2 = torch.cuda.Event(enable_timing=True)
                e2.record(); ev_dw.append((e1, e2))
            ops.kipf_layer_bwd_x(g, dzd, wd, F, out=dX)
            if record:
                e3 = torch.cuda.Event(enable_timing=True)
                e3.record(); ev_bwd.append((e2, e3))
        info = {}
        watch = None

    import contextlib

    def phase(name, factor=1.0):
        return watch.phase(name, factor) if watch is not None else contextlib.nullcontext()

    def barrier():
        # drain this device first (compute AND the C ABI's communication stream), so that the process group's barrier
        # never runs beside a transfer of the other communicator, then meet the other ranks
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    with phase("warm-up steps"):
        for _ in range(args.warmup):
            step()
        barrier()
    with phase("timed loop", factor=max(1.0, args.steps / 50.0)):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(record=True)
        barrier()
        dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        with phase("max of the step time over the ranks"):
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()

    ms_per_step = dt / args.steps * 1e3
    value = nnz_total * args.steps / dt

    if args.custom:
        wl = "custom size: "
    else:
        wl = {"c2": "BASELINE configs[1]: ", "c2-weak-sbm": "BASELINE configs[1] per GPU (weak scaling, block model): ",
              "c5": "BASELINE configs[4]: ", "c5-local": "BASELINE configs[4], locality variant (SURVEY.md 8d): "}[args.config]
    n_total = args.nodes * world if weak else args.nodes
    wl += (f"Kipf GCN layer fwd+bwd, random graph {n_total} vertices / {nnz_total} CSR entries, {F} features, fp32"
           + (f", row-partitioned over {world} GPUs" if world > 1 else ""))
    out = {
        "metric": "msgpass fwd+bwd edges/sec", "value": value, "unit": "edges/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak" if (weak or world == 1) else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl, "vertices": n_total, "entries": nnz_total, "entries_per_gpu": nnz_local, "features": F,
                   "parallelism": f"row-partition x{world}" if world > 1 else "single GPU", **info},
    }
    # dominant kernel: the fused forward launch (CSR gather-aggregate + dense step; HBM bound), timed with HIP events
    # on its launch stream inside the timed loop.  algorithmic bytes = the aggregation's (SURVEY.md 8d:
    # nnz*(4F+8) + N*(4F+8), P written once) + the Z rows written; P is not re-read and W (64 KB) stays in LDS.
    # At N > 1 it is rank 0's larger forward launch: the interior rows of its shard (run while the halo is in flight)
    # or the boundary rows (run after it), whichever holds more entries.
    if world == 1:
        agg_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        k_rows, k_nnz, part = args.nodes, nnz_local, ""
    else:
        int_nnz = info["interior_entries_per_gpu"]
        use_int = int_nnz >= nnz_local - int_nnz and len(ev) > 0
        pairs = ev if use_int else ev_bnd
        agg_ms = float(np.mean([a.elapsed_time(b) for a, b in pairs])) if pairs else float("nan")
        k_rows = info["interior_rows_per_gpu"] if use_int else shard.n - info["interior_rows_per_gpu"]
        k_nnz = int_nnz if use_int else nnz_local - int_nnz
        part = (", interior rows of rank 0's shard (halo exchange in flight)" if use_int
                else ", boundary rows of rank 0's shard (after the halo exchange)")
    alg_bytes = k_nnz * (4 * F + 8) + k_rows * (4 * F + 8) + k_rows * 4 * F
    achieved = alg_bytes / (agg_ms * 1e-3) / 1e9
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if world == 1 and os.path.exists(tpath) and not args.custom and args.config == "c2":
        try:
            tj = json.load(open(tpath))
            traffic = tj.get("agg_gemm_fwd_bytes_per_launch")
            traffic_source = ("profiles/traffic_latest.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an EARLIER run of "
                              "this command (2*FETCH_SIZE + WRITE_SIZE, gfx950 correction); not measured by this run -- "
                              + str(tj.get("source", "")))
        except Exception:
            traffic = None
    fused = F in (64, 128, 256)
    kname = ((f"agg_gemm_kernel<{F},coef>" if F != 256 else "agg_gemm256_kernel<coef> (W in registers)")
             + " (fused kipf_propagate + matmul fwd)" if fused else
             "kipf_layer_fwd = csr_gather_agg + dense step (two launches at this width)") + part
    out["roofline"] = {"bound": "hbm", "kernel": kname,
                       "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                       "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": agg_ms}
    # the dense contraction of the step that is a launch of its own: dW = dZ . P^T (2 N F^2 flops on the fp32 matrix
    # cores; the other two dense steps ride inside the fused gather launches).  HIP events in the timed loop at N = 1.
    if world == 1 and ev_dw:
        dw_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_dw]))
        flops = 2.0 * args.nodes * F * F
        dw_kernel = (f"gemm_dw_full_kernel<{F},{F}> + slab_reduce_kernel" if F in (64, 128) else
                     f"gemm_dw_full_kernel<128,128,blocked> on {(F // 128) ** 2} blocks of dW + slab_reduce_kernel" if F % 128 == 0 and F <= 512 else
                     "gemm_atb_tiled (dW)")
        out["roofline"]["dense"] = {"bound": "mfma", "kernel": dw_kernel + " (matmul reverse wrt the weights, dW = dZ . P^T)",
                                    "achieved": flops / (dw_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": flops / (dw_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, "flops_per_launch": flops,
                                    "avg_launch_ms": dw_ms, "dtype": "f32 (v_mfma_f32_32x32x2_f32, exact fp32 products)"}
    if world == 1 and ev_bwd:
        # the reverse launch (get_partial_kipf_propagate_left_val + the dense step's reverse in one pull over the transposed
        # CSR): SURVEY.md 8d's per-entry bytes nnz*(4F+4) + N*(4F+4), + the dX rows it writes
        bw_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_bwd]))
        bw_bytes = nnz_local * (4 * F + 4) + args.nodes * (4 * F + 4) + args.nodes * 4 * F
        out["roofline"]["reverse"] = {"bound": "hbm", "kernel": "agg_gemm_kernel<F,plain> (fused pull: dX = (A^T dZ) . W)",
                                      "achieved": bw_bytes / (bw_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": bw_bytes / (bw_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": bw_bytes,
                                      "avg_launch_ms": bw_ms,
                                      "note": "the gathered dZ rows (512 MB) partly live in the 256 MiB Infinity Cache: algorithmic bytes, not HBM bytes"}
    ok = True
    if world == 1:
        try:
            ceil = measure_copy_ceiling(dev)
            out["roofline"]["measured_copy_ceiling_GBps"] = round(ceil, 1)
            out["roofline"]["frac_of_measured_copy_ceiling"] = round(achieved / ceil, 4)
            out["roofline"]["measured_copy_ceiling_note"] = ("1 GiB device copy on this box in this run (athena_mp_device_copy, one 16-byte "
                                                             "element per thread, read + written bytes / mean of 10 launches); the gather runs "
                                                             "above it because part of X lives in the 256 MiB Infinity Cache")
        except Exception as exc:
            out["roofline"]["measured_copy_ceiling_GBps"] = None
            out["roofline"]["measured_copy_ceiling_note"] = f"not measured: {type(exc).__name__}: {exc}"[:200]
        if not args.no_cpu_baseline:
            out["cpu_baseline"], ref = cpu_baseline(ia, ja, x, w, dz, F, args.cpu_sample_rows, args.cpu_runs)
            out["parity"] = parity_single(ref, P, Z, dW, dX, dz)
            ok = out["parity"]["ok"]
        else:
            out["parity"] = None
    else:
        with phase("parity of every rank against the oracle", factor=3.0):
            out["parity"] = parity_sharded(shard_step, shard, seeds, dev) if not shard_step.transform_first else None
        ok = out["parity"] is None or out["parity"]["ok"]
        with phase("breakdown (each part timed alone)", factor=2.0):
            out["breakdown"] = adist.measure_breakdown(shard_step)
        out["breakdown"]["note"] = ("each part timed alone after the timed loop, rank 0 (events); in the step the exchanges "
                                    "run under the interior launches")
        if out["breakdown"].get("dw_ms"):
            flops = 2.0 * shard.n * F * shard_step.Fo
            tf = flops / (out["breakdown"]["dw_ms"] * 1e-3) / 1e12
            out["roofline"]["dense"] = {"bound": "mfma", "kernel": "dW = dZ . P^T on rank 0's rows (timed alone after the loop)",
                                        "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                        "frac": tf / MFMA_F32_PEAK_TFLOPS, "flops_per_launch": flops,
                                        "avg_launch_ms": out["breakdown"]["dw_ms"]}
        # what rank 0's communicator did: the ranks RCCL itself counts (ncclCommCount), its version, bytes to every peer
        out["rccl"] = adist.c_comm_stats()
        out["config"]["control_plane"] = control_plane_note(backend)
        # a line that "scaled" through the host-staged TEST transport is not a measurement of RCCL over xGMI: outside the
        # one-device dry run it is an error, stated in the line and in the exit code
        err = transport_error(str(info.get("transport", "")), one_device)
        if err:
            out["ok"], out["error"], ok = False, err, False
    if world == 1:
        # "activation = none for the headline; relu epilogue reported separately" -- the same step with the
        # layer's activation: relu in the forward launch's store (athena_mp_kipf_layer_fwd, act = relu), its reverse factor as one
        # elementwise launch in front of dW and the fused pull.  Timed after the headline's loop; never part of `value`.
        try:
            Y = torch.empty((N, F), device=dev)
            dzr = torch.empty((N, F), device=dev)

            def relu_step():
                ops.kipf_layer_fwd(g, xd, wd, F, act="relu", P=P, Z=Y)
                ops.activation_bwd("relu", Y, dzd, out=dzr)
                ops.matmul_dw(P, dzr, out=dW)
                ops.kipf_layer_bwd_x(g, dzr, wd, F, out=dX)

            for _ in range(3):
                relu_step()
            torch.cuda.synchronize()
            k = max(5, args.steps // 5)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(k):
                relu_step()
            e1.record()
            torch.cuda.synchronize()
            relu_ms = e0.elapsed_time(e1) / k
            out["relu_epilogue"] = {"ms_per_step": relu_ms, "value": nnz_total / (relu_ms * 1e-3), "unit": "edges/s", "steps": k,
                                    "note": "the same step with activation relu: fused into the forward launch's store; reverse factor = one "
                                            "elementwise launch (reads y and dz, writes dz') in front of dW and the fused pull"}
            if not args.no_cpu_baseline and "parity" in out and out["parity"] and ref["full"]:
                from oracle import oracle as _o
                r_ = min(ref["rows"], 20000)
                out["relu_epilogue"]["Z_rel_first_rows"] = rel(Y[:r_].cpu().numpy(), _o.activation("relu", ref["z"][:r_]))
            del Y, dzr
        except Exception as exc:      # reported beside the headline, never instead of it
            out["relu_epilogue"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    if world == 1 and not args.no_secondary and not args.custom and args.config == "c2":
        # free the headline's tensors first: configs[3] keeps 33 GB of S and configs[4] holds 50 GB in the children
        P = Z = dW = dX = xd = dzd = g = None
        torch.cuda.empty_cache()
        out["secondary"] = run_secondary([k.strip() for k in args.secondary.split(",") if k.strip()])
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        with phase("final barrier"):
            dist.barrier()
        watch.close()
        adist.c_comm_destroy()
        dist.destroy_process_group()
    if not ok:
        sys.exit("bench.py: parity against the oracle FAILED (see the 'parity' object of the line above)")


def _global_csr(synth, args):
    """whole-graph CSR of the locality variant on one GPU (Fortran-convention arrays)"""
    import numpy as np
    ia, cols = synth.random_graph_csr_rows(args.nodes, args.pairs, 0, args.nodes, locality=args.locality)
    ja = np.zeros((2, cols.size), np.int32, order="F")
    ja[0] = cols + 1
    return ia.astype(np.int32), ja


if __name__ == "__main__":
    main()
