// Multi-GPU side of libathena_mp: communicator, row-partition shard and halo exchange behind the C ABI
// (SURVEY.md 8b "multi-GPU: athena_mp_comm_create / graph_partition / halo_exchange", 8e).  The reference has nothing
// distributed: this is new, one process per GPU.
//
//   transport      RCCL over xGMI: grouped ncclSend / ncclRecv to every peer (one transfer per xGMI link, not a ring) on a
//                  communication stream of its own, event-ordered against the compute stream (amp::stream()); dW by
//                  ncclAllReduce.  librccl is dlopen'ed on first use, so single-GPU users never load it and a process
//                  that already carries a copy (PyTorch) shares it.
//   test transport ATHENA_MP_COMM_TRANSPORT=shm loads the plugin libathena_mp_testcomm.so (test_transport.cpp; NOT part of this
//                  library): the same calls with the bytes staged through files under /dev/shm -- for boxes with ONE GPU
//                  (RCCL refuses two ranks on one device); it exercises everything here except the nccl* calls themselves.
//   shard          rank r owns a contiguous block of vertex rows.  Columns are renumbered [local | halo], local vertices
//                  INTERIOR FIRST (rows that reference no remote vertex), so rows [0, n_int) run while the halo is in
//                  flight and only [n_int, n) wait.  The backward graph lists each row's entries by ascending global
//                  source id -- the order the reference's scatter accumulates in
//                  (athena_diffstruc_extd_sub_kipf.f90:101-109; athena's graphs are undirected, so the transposed row of
//                  a vertex is its own row).  Degrees of halo columns come from their owners (the Kipf coefficient
//                  needs deg of the remote endpoint, :39-42).
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "common.h"
#include "transport.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// transports (the interface: transport.h)
// ---------------------------------------------------------------------------------------------------------------------
using amp_comm::Transport;

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;   // optional (diagnostics)
    ncclResult_t (*GetVersion)(int *) = nullptr;                     // optional
};
RcclApi g_rccl;

int rccl_load()
{
    if (g_rccl.lib) return 0;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) {
        amp::set_error("comm: cannot load librccl (%s)", dlerror());
        return 1;
    }
#define AMP_SYM(field, sym)                                                   \
    *(void **)(&g_rccl.field) = dlsym(h, sym);                                \
    if (!g_rccl.field) {                                                      \
        amp::set_error("comm: librccl lacks %s", sym);                        \
        return 1;                                                             \
    }
    AMP_SYM(GetUniqueId, "ncclGetUniqueId")
    AMP_SYM(CommInitRank, "ncclCommInitRank")
    AMP_SYM(CommDestroy, "ncclCommDestroy")
    AMP_SYM(GetErrorString, "ncclGetErrorString")
    AMP_SYM(AllReduce, "ncclAllReduce")
    AMP_SYM(AllGather, "ncclAllGather")
    AMP_SYM(Send, "ncclSend")
    AMP_SYM(Recv, "ncclRecv")
    AMP_SYM(GroupStart, "ncclGroupStart")
    AMP_SYM(GroupEnd, "ncclGroupEnd")
#undef AMP_SYM
    *(void **)(&g_rccl.CommCount) = dlsym(h, "ncclCommCount");
    *(void **)(&g_rccl.GetVersion) = dlsym(h, "ncclGetVersion");
    g_rccl.lib = h;
    return 0;
}

#define AMP_NCCL(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t _r = (expr);                                                                        \
        if (_r != ncclSuccess) {                                                                         \
            amp::set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
            return 1;                                                                                    \
        }                                                                                                \
    } while (0)

struct RcclTransport : Transport {
    ncclComm_t comm = nullptr;
    const char *name() const override { return "rccl"; }
    int ranks_seen() const override
    {
        int n = 0;
        return (g_rccl.CommCount && comm && g_rccl.CommCount(comm, &n) == ncclSuccess) ? n : -1;
    }
    int version() const override
    {
        int v = 0;
        return (g_rccl.GetVersion && g_rccl.GetVersion(&v) == ncclSuccess) ? v : -1;
    }
    ~RcclTransport() override
    {
        if (comm) g_rccl.CommDestroy(comm);
    }
    int exchange(const void *const *sendp, const size_t *sendb, void *const *recvp, const size_t *recvb,
                 hipStream_t s) override
    {
        AMP_NCCL(g_rccl.GroupStart());
        ncclResult_t bad = ncclSuccess;
        int bad_peer = -1;
        for (int p = 0; p < world && bad == ncclSuccess; ++p) {
            if (p == rank) continue;
            if (sendb[p]) bad = g_rccl.Send(sendp[p], sendb[p], ncclInt8, p, comm, s);
            if (bad == ncclSuccess && recvb[p]) bad = g_rccl.Recv(recvp[p], recvb[p], ncclInt8, p, comm, s);
            if (bad != ncclSuccess) bad_peer = p;
        }
        const ncclResult_t end = g_rccl.GroupEnd();   // the group is closed whatever happened inside it
        if (bad != ncclSuccess) {
            amp::set_error("grouped ncclSend / ncclRecv with rank %d failed: %s", bad_peer, g_rccl.GetErrorString(bad));
            return 1;
        }
        AMP_NCCL(end);
        return 0;
    }
    int allreduce_f32(float *buf, size_t count, hipStream_t s) override
    {
        if (count) AMP_NCCL(g_rccl.AllReduce(buf, buf, count, ncclFloat32, ncclSum, comm, s));
        return 0;
    }
    int allgather(const void *send, void *recv, size_t bytes, hipStream_t s) override
    {
        if (bytes) AMP_NCCL(g_rccl.AllGather(send, recv, bytes, ncclInt8, comm, s));
        return 0;
    }
};

// The TEST transport (bytes staged through /dev/shm files: several ranks on ONE GPU, where RCCL refuses) is not in this library:
// it is the plugin libathena_mp_testcomm.so (test_transport.cpp), looked for beside this library and loaded only when
// ATHENA_MP_COMM_TRANSPORT=shm asks for it.
struct TestPlugin {
    void *lib = nullptr;
    amp_comm::test_create_fn create = nullptr;
    amp_comm::test_unique_id_fn unique_id = nullptr;
};
TestPlugin g_test;
void test_plugin_error(const char *m) { amp::set_error("%s", m); }
int test_plugin_load()
{
    if (g_test.lib) return 0;
    std::string path = "libathena_mp_testcomm.so";
    Dl_info info;
    if (dladdr((const void *)&test_plugin_error, &info) && info.dli_fname) {
        const std::string self(info.dli_fname);
        const size_t slash = self.rfind('/');
        if (slash != std::string::npos) path = self.substr(0, slash + 1) + path;
    }
    void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        amp::set_error("comm: ATHENA_MP_COMM_TRANSPORT=shm asks for the TEST transport, but its plugin %s cannot be loaded (%s)",
                       path.c_str(), dlerror());
        return 1;
    }
    *(void **)(&g_test.create) = dlsym(h, "athena_mp_test_transport_create");
    *(void **)(&g_test.unique_id) = dlsym(h, "athena_mp_test_transport_unique_id");
    if (!g_test.create || !g_test.unique_id) {
        amp::set_error("comm: %s lacks the test transport's entry points", path.c_str());
        return 1;
    }
    g_test.lib = h;
    return 0;
}

bool want_shm()
{
    const char *t = getenv("ATHENA_MP_COMM_TRANSPORT");
    return t && strcmp(t, "shm") == 0;
}

} // namespace

// ---- a deadline for every transfer this library puts on a communication stream -------------------------------------------
// No RCCL collective has a completion deadline: two ranks that disagree about a transfer wait for each other until whoever
// launched them gives up.  The host never blocks in athena_mp_halo_start / _allreduce_start (the stall would surface in the
// caller's next synchronize, far from its cause), so a monitor thread watches the completion EVENT of every transfer that
// was started: one that is still pending ATHENA_MP_COLLECTIVE_TIMEOUT_S seconds later (default 1800 -- long enough for one rank to
// write a checkpoint, run an evaluation pass or sit in a debugger; bench.py sets 120 for itself; 0 = no monitor) ends the
// process -- a message naming the rank and the transfer on stderr and in athena_mp_last_error, the host's stall handler if one is
// registered (athena_mp_set_stall_handler: its chance to flush state), then _exit(3): no retry, no cleanup that could block in the
// same communicator, nothing written on the host program's stdout.  Fortran drivers get the protection bench.py's python
// watchdog gives the bench.  ATHENA_MP_COMM_TEST_DELAY_MS (tests only) parks a bounded spin kernel in front of the completion event.
typedef void (*athena_mp_stall_handler_fn)(int32_t rank, const char *what, double seconds);
namespace {
std::atomic<athena_mp_stall_handler_fn> g_stall_handler{nullptr};
struct Watched {
    hipEvent_t ev, begin;   // begin (may be null): recorded on the communication stream just in front of the transfer -- the clock
    bool started;           // starts when IT has completed, not at enqueue time (a host that runs many steps ahead of the device
    double t0, limit;       // must not be mistaken for a stall)
    int rank, device;
    char what[96];
};
// (heap-allocated and never destroyed: the detached monitor thread may still be polling while the process runs its static
// destructors at exit)
std::mutex &g_watch_mu = *new std::mutex;
std::vector<Watched> &g_watch = *new std::vector<Watched>;
std::atomic<bool> g_watch_started{false};
double watch_now()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
double watch_timeout()
{
    static const double t = [] {
        const char *e = getenv("ATHENA_MP_COLLECTIVE_TIMEOUT_S");
        const double v = e ? atof(e) : 1800.0;   // a library default must outlast a checkpoint, an evaluation pass or a debugger
        return v > 0.0 ? v : 0.0;                // stop on ONE rank; bench.py sets its own 120 s
    }();
    return t;
}
void watch_loop()
{
    int cur_dev = -1;
    for (;;) {
        usleep(100000);
        std::lock_guard<std::mutex> lock(g_watch_mu);
        const double now = watch_now();
        for (size_t i = 0; i < g_watch.size();) {
            Watched &w = g_watch[i];
            if (w.device != cur_dev) {
                (void)hipSetDevice(w.device);
                cur_dev = w.device;
            }
            const hipError_t q = hipEventQuery(w.ev);
            if (q == hipSuccess) {
                g_watch[i] = g_watch.back();
                g_watch.pop_back();
                continue;
            }
            (void)hipGetLastError();
            if (!w.started) {
                if (hipEventQuery(w.begin) == hipSuccess) {
                    w.started = true;
                    w.t0 = now;
                } else {
                    (void)hipGetLastError();
                }
            }
            if (q == hipErrorNotReady && w.started && now - w.t0 > w.limit) {
                // stderr and athena_mp_last_error only: the host program's stdout is not this library's to write on
                amp::set_error("rank %d stalled in %s for more than %.0f s (ATHENA_MP_COLLECTIVE_TIMEOUT_S): a peer is missing, late or "
                               "in another collective", w.rank, w.what, w.limit);
                fprintf(stderr, "[athena_mp] %s\n", athena_mp_last_error());
                fflush(stderr);
                // the host's chance to flush its own state (checkpoint, log): called on THIS monitor thread, once; the process
                // ends when it returns -- a stalled communicator cannot be recovered, and a retry would hang in it again
                if (athena_mp_stall_handler_fn h = g_stall_handler.load()) h(w.rank, w.what, w.limit);
                _exit(3);
            }
            ++i;
        }
    }
}
// the transfer whose completion `ev` marks has just been enqueued
void watch_arm(hipEvent_t ev, int rank, int device, const char *what, double factor = 1.0, hipEvent_t begin = nullptr)
{
    if (watch_timeout() <= 0.0) return;
    {
        std::lock_guard<std::mutex> lock(g_watch_mu);
        bool found = false;
        for (Watched &w : g_watch)
            if (w.ev == ev) {   // the event was re-recorded: the newest transfer is the one it now stands for
                w.t0 = watch_now();
                w.limit = watch_timeout() * factor;
                w.begin = begin;
                w.started = begin == nullptr;
                snprintf(w.what, sizeof(w.what), "%s", what);
                found = true;
            }
        if (!found) {
            Watched w;
            w.ev = ev; w.begin = begin; w.started = begin == nullptr;
            w.t0 = watch_now(); w.limit = watch_timeout() * factor; w.rank = rank; w.device = device;
            snprintf(w.what, sizeof(w.what), "%s", what);
            g_watch.push_back(w);
        }
    }
    bool expected = false;
    if (g_watch_started.compare_exchange_strong(expected, true)) {
        try {
            std::thread(watch_loop).detach();
        } catch (...) {   // no thread to be had: the transfers run unwatched, as before round 4
            fprintf(stderr, "[athena_mp] the deadline monitor could not be started; collectives run without a deadline\n");
        }
    }
}
void watch_forget(hipEvent_t ev)   // before the event is destroyed
{
    std::lock_guard<std::mutex> lock(g_watch_mu);
    for (size_t i = 0; i < g_watch.size();)
        if (g_watch[i].ev == ev) {
            g_watch[i] = g_watch.back();
            g_watch.pop_back();
        } else {
            ++i;
        }
}
__global__ void comm_test_delay_kernel(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
void test_delay(hipStream_t s)   // tests only: a BOUNDED wait in front of a completion event (100 MHz wall clock)
{
    static const long long ms = [] {
        const char *e = getenv("ATHENA_MP_COMM_TEST_DELAY_MS");
        return e ? atoll(e) : 0ll;
    }();
    if (ms > 0) hipLaunchKernelGGL(comm_test_delay_kernel, dim3(1), dim3(1), 0, s, std::min(ms, 20000ll) * 100000ll);
}
}   // namespace

struct athena_mp_comm {
    Transport *t = nullptr;
    hipStream_t cs = nullptr;       // communication stream
    hipEvent_t ev_ready = nullptr;  // compute -> comm
    hipEvent_t ev_done = nullptr;   // comm -> compute (all-reduce)
    hipEvent_t ev_begin = nullptr;  // in front of the all-reduce on the communication stream (the deadline's clock)
    int device = 0;
};

struct athena_mp_shard {
    athena_mp_comm *comm = nullptr;
    int32_t n = 0, n_int = 0, n_halo = 0;
    int64_t nnz = 0, row_offset = 0, n_total = 0;
    std::vector<int64_t> row_off;          // [world+1] first global row of every rank
    std::vector<int32_t> order;            // [n] original local id of the vertex now called k
    std::vector<int64_t> halo_ids;         // [n_halo] global ids, ascending (grouped by owner)
    std::vector<int64_t> recv_counts, send_counts, roff, soff;   // rows per peer
    std::vector<int32_t> send_idx_h, col_deg, row_deg;
    // how the halo travels: 0 = grouped send / recv of the rows peers asked for (packed by a gather kernel), 1 = one
    // all-gather of every rank's whole block (no pack, no send lists): x_ext = [max_n local slots | world x max_n]
    int mode = 0;
    int64_t max_n = 0;                     // largest block (all-gather slots)
    int32_t ext_rows = 0;                  // rows of x_ext beyond the n local ones (== n_halo in mode 0)
    double halo_fraction = 0.0, tau = 0.7; // (sum of distinct halo rows over ranks) / ((world - 1) * n_total), threshold
    std::vector<int64_t> ext_ids;          // [ext_rows] global id held by each row beyond the local ones, -1 = padding
    int32_t *send_idx = nullptr;           // device [n_send]: rows (new numbering) peers asked for, grouped by peer
    int64_t n_send = 0;
    athena_mp_graph *g[4] = {};            // fwd interior, fwd boundary, bwd interior, bwd boundary
    // shards that keep adj_ja(2,.) (Duvenaud / GNO on ONE partitioned graph): the rank's edge-feature columns are the
    // distinct global edge ids its rows reference, renumbered 1 .. n_edge_cols in ascending global order
    int32_t with_edges = 0, n_edge_cols = 0;
    std::vector<int64_t> edge_ids;         // [n_edge_cols] global edge id (0-based) of each local edge column
    // edge columns CUT by the partition (an entry of this rank's rows points at a vertex of peer p): local column ids grouped
    // by peer, ascending global id inside a group -- the peer holds the same columns in the same order (symmetry, checked)
    std::vector<int64_t> eoff;             // [world+1]
    std::vector<int32_t> eshare_h;         // [eoff[world]]
    int32_t *eshare = nullptr;             // device copy
    float *ered_buf = nullptr;             // [2][eoff[world] * F] pack + receive buffer of athena_mp_shard_edge_reduce
    size_t ered_cap = 0;
    hipEvent_t ev_halo[2] = {};            // comm -> compute, one per slot
    hipEvent_t ev_halo_b[2] = {};          // in front of each slot's transfer (the deadline's clock)
    float *send_buf[2] = {};
    size_t send_cap[2] = {};
    float *red_buf[2] = {};                // halo REDUCE: what the peers computed for this rank's rows, grouped by peer
    size_t red_cap[2] = {};
    int32_t red_F[2] = {};
};

using namespace amp;

namespace {

// host metadata through the transport (device staging: RCCL moves device memory)
int exchange_host(athena_mp_comm *c, const std::vector<const void *> &sp, const std::vector<size_t> &sb,
                  const std::vector<void *> &rp, const std::vector<size_t> &rb)
{
    const int W = c->t->world;
    size_t tot = 0;
    for (int p = 0; p < W; ++p) tot += ((sb[p] + 255) & ~(size_t)255) + ((rb[p] + 255) & ~(size_t)255);
    char *dev = nullptr;
    AMP_HIP(hipMalloc((void **)&dev, tot ? tot : 256));
    std::vector<const void *> dsp(W, nullptr);
    std::vector<void *> drp(W, nullptr);
    size_t off = 0;
    for (int p = 0; p < W; ++p) {
        dsp[p] = dev + off;
        if (sb[p] && p != c->t->rank) AMP_HIP(hipMemcpy(dev + off, sp[p], sb[p], hipMemcpyHostToDevice));
        off += (sb[p] + 255) & ~(size_t)255;
        drp[p] = dev + off;
        off += (rb[p] + 255) & ~(size_t)255;
    }
    int rc = c->t->counted_exchange(dsp.data(), sb.data(), drp.data(), rb.data(), c->cs);
    if (rc == 0 && hipEventRecord(c->ev_done, c->cs) == hipSuccess)
        watch_arm(c->ev_done, c->t->rank, c->device, "a metadata exchange of the shard build (athena_mp_shard_create)", 5.0);
    if (rc == 0 && hipStreamSynchronize(c->cs) != hipSuccess) {
        set_error("comm: metadata exchange failed on the communication stream");
        rc = 1;
    }
    if (rc == 0)
        for (int p = 0; p < W; ++p)
            if (rb[p] && p != c->t->rank && hipMemcpy(rp[p], drp[p], rb[p], hipMemcpyDeviceToHost) != hipSuccess) rc = 1;
    (void)hipFree(dev);
    if (rc && athena_mp_last_error()[0] == 0) set_error("comm: metadata exchange failed");
    return rc;
}

// every rank contributes `bytes`; out [world * bytes].  One in-place all-gather of a device staging buffer.
int allgather_host(athena_mp_comm *c, const void *mine, size_t bytes, void *out)
{
    const int W = c->t->world, r = c->t->rank;
    memcpy((char *)out + (size_t)r * bytes, mine, bytes);
    if (W == 1 || bytes == 0) return 0;
    char *dev = nullptr;
    AMP_HIP(hipMalloc((void **)&dev, (size_t)W * bytes));
    int rc = 0;
    if (hipMemcpy(dev + (size_t)r * bytes, mine, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = 1;
    if (rc == 0) rc = c->t->counted_allgather(dev + (size_t)r * bytes, dev, bytes, c->cs);
    if (rc == 0 && hipEventRecord(c->ev_done, c->cs) == hipSuccess)
        watch_arm(c->ev_done, c->t->rank, c->device, "a metadata all-gather of the shard build (athena_mp_shard_create)", 5.0);
    if (rc == 0 && hipStreamSynchronize(c->cs) != hipSuccess) rc = 1;
    if (rc == 0 && hipMemcpy(out, dev, (size_t)W * bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = 1;
    (void)hipFree(dev);
    if (rc && athena_mp_last_error()[0] == 0) set_error("comm: metadata all-gather failed");
    return rc;
}

// every rank reports a status word; returns 0 only when all are 0 -- so that a rank whose arguments are bad takes
// every rank out of a collective call together instead of leaving its peers blocked in the next exchange
int agree_ok(athena_mp_comm *c, int32_t mine, const char *what)
{
    const int W = c->t->world;
    std::vector<int32_t> all(W, 0);
    if (allgather_host(c, &mine, 4, all.data())) return 1;
    for (int p = 0; p < W; ++p)
        if (all[p] != 0) {
            if (mine == 0) set_error("%s: rank %d rejected its arguments (all ranks return together)", what, p);
            return 2;
        }
    return 0;
}

// how the halo travels: grouped send / recv of the distinct rows each peer needs, or -- when the ranks
// together need more than `tau` of all remote rows anyway -- one all-gather of whole blocks (no pack kernel, no send
// lists).  ATHENA_MP_HALO_MODE = auto (default) | p2p | allgather;  ATHENA_MP_HALO_ALLGATHER_FRACTION = tau (0.7).
double halo_tau()
{
    const char *e = getenv("ATHENA_MP_HALO_ALLGATHER_FRACTION");
    const double t = e ? atof(e) : 0.7;
    return t > 0.0 ? t : 0.7;
}
int halo_mode_env()   // -1 auto, 0 p2p, 1 allgather
{
    const char *e = getenv("ATHENA_MP_HALO_MODE");
    if (!e || !e[0] || strcmp(e, "auto") == 0) return -1;
    if (strcmp(e, "allgather") == 0) return 1;
    return 0;
}

// eid: local edge column (1-based, 0 = none) of every entry, or null for a shard without edge features
int make_graph(const std::vector<int32_t> &ia_all, const std::vector<int32_t> &col, const std::vector<int32_t> *eid, int32_t n_edge_cols,
               int32_t r0, int32_t r1, int32_t n_cols, const std::vector<int32_t> &row_deg, const std::vector<int32_t> &col_deg,
               athena_mp_graph **out)
{
    const int32_t e0 = ia_all[r0], e1 = ia_all[r1];
    std::vector<int32_t> ia(r1 - r0 + 1), ja(2 * (size_t)(e1 - e0));
    for (int32_t r = r0; r <= r1; ++r) ia[r - r0] = ia_all[r] - e0 + 1;
    for (int32_t w = e0; w < e1; ++w) {
        ja[2 * (size_t)(w - e0)] = col[w] + 1;
        ja[2 * (size_t)(w - e0) + 1] = eid ? (*eid)[w] : 0;
    }
    static const int32_t dummy[2] = {0, 0};
    return athena_mp_graph_create(r1 - r0, n_cols, e1 - e0, ia.data(), ja.empty() ? dummy : ja.data(), eid ? n_edge_cols : 0,
                                  row_deg.data() + r0, col_deg.data(), out);
}

} // namespace

// halo REDUCE, the transpose of the halo exchange: rows a rank computed FOR remote vertices travel to their owners and are added
// there.  One launch per peer in peer order (a row may be owed by several peers: their shares are added one after the other,
// the same order on every run), 16-byte lanes over F floats.
// edge reduce: rows of any width F (coordinate gradients: F = d), one thread per value
__global__ void edge_pack_kernel(const float *__restrict__ e, const int32_t *__restrict__ idx, int64_t n, int F, float *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * F) return;
    const int64_t r = t / F;
    out[t] = e[(int64_t)idx[r] * F + (t - r * F)];
}
__global__ void edge_add_kernel(const float *__restrict__ recv, const int32_t *__restrict__ idx, int64_t n, int F, float *__restrict__ e)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * F) return;
    const int64_t r = t / F;
    float *dst = e + (int64_t)idx[r] * F + (t - r * F);   // a cut column is shared with exactly ONE peer: no two rows collide
    *dst = *dst + recv[t];
}

__global__ void halo_add_rows_kernel(const float *__restrict__ recv, const int32_t *__restrict__ idx, int64_t n, int F, float *__restrict__ y)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = F / 4;
    if (t >= n * q) return;
    const int64_t r = t / q;
    const int c = (int)(t - r * q);
    const int64_t dst = idx ? idx[r] : r;
    typedef float v4 __attribute__((ext_vector_type(4)));
    v4 a = *reinterpret_cast<const v4 *>(recv + r * F + 4 * c);
    v4 *o = reinterpret_cast<v4 *>(y + dst * F + 4 * c);
    *o = *o + a;
}

extern "C" {

int athena_mp_comm_unique_id(void *id128)
{
    AMP_REQUIRE(id128 != nullptr, "comm_unique_id: null buffer");
    memset(id128, 0, 128);
    if (want_shm()) {
        if (test_plugin_load()) return 1;
        AMP_REQUIRE(g_test.unique_id(id128) == 0, "comm_unique_id: /dev/urandom unreadable");
        return 0;
    }
    if (rccl_load()) return 1;
    ncclUniqueId id;
    AMP_NCCL(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, 128);
    return 0;
}

int athena_mp_comm_create(int32_t rank, int32_t world, const void *id128, athena_mp_comm **out)
{
    AMP_REQUIRE(out != nullptr, "comm_create: null out pointer");
    *out = nullptr;
    AMP_REQUIRE(world >= 1 && rank >= 0 && rank < world && id128 != nullptr, "comm_create: bad rank %d / world %d", rank, world);
    athena_mp_comm *c = new athena_mp_comm();
    c->device = amp::device();
    if (want_shm()) {
        Transport *t = test_plugin_load() ? nullptr : g_test.create(id128, test_plugin_error);
        if (!t) {
            delete c;
            return 1;
        }
        c->t = t;
    } else {
        if (rccl_load()) {
            delete c;
            return 1;
        }
        RcclTransport *t = new RcclTransport();
        ncclUniqueId id;
        memcpy(&id, id128, 128);
        ncclResult_t r = g_rccl.CommInitRank(&t->comm, world, id, rank);
        if (r != ncclSuccess) {
            set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r));
            t->comm = nullptr;
            delete t;
            delete c;
            return 1;
        }
        c->t = t;
    }
    c->t->rank = rank;
    c->t->world = world;
    if (hipStreamCreateWithFlags(&c->cs, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_begin, hipEventDisableTiming) != hipSuccess) {
        set_error("comm_create: cannot create the communication stream / events");
        delete c->t;
        delete c;
        return 1;
    }
    *out = c;
    return 0;
}

/* MPI-free bootstrap for a Fortran host.  A file left behind by a run that crashed must never be taken for this
 * launch's (ncclCommInitRank with mismatched ids does not return), so the rendezvous proves liveness both ways:
 *   rank r > 0  draws a random nonce and keeps re-writing `path`.hello.r (same nonce, fresh file: a heartbeat) until
 *               `path` carries that nonce in slot r; a `path` without it -- a stale one -- is ignored.
 *   rank 0      removes any stale `path`, waits until it has seen every hello file CHANGE (a stale hello never does),
 *               then publishes id + the nonces it read (write + rename).  It removes `path` after every rank has joined.
 * Every wait is bounded (ATHENA_MP_COLLECTIVE_TIMEOUT_S when set, 300 s otherwise). */
namespace {
double boot_now()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
double boot_timeout()
{
    // the deadline of every other wait for a peer, when the caller set one; 300 s otherwise (ranks of a job start seconds apart,
    // not the half hour the library allows a running collective by default)
    const char *e = getenv("ATHENA_MP_COLLECTIVE_TIMEOUT_S");
    const double t = e ? atof(e) : 300.0;
    return t > 0 ? t : 300.0;
}
bool boot_write(const std::string &file, const void *data, size_t bytes)
{
    const std::string tmp = file + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const size_t w = fwrite(data, 1, bytes, f);
    fclose(f);
    return w == bytes && rename(tmp.c_str(), file.c_str()) == 0;
}
// reads exactly `bytes`; *stamp identifies this version of the file (inode + mtime)
bool boot_read(const std::string &file, void *data, size_t bytes, uint64_t *stamp)
{
    int fd = open(file.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    bool ok = fstat(fd, &st) == 0 && (size_t)st.st_size == bytes && read(fd, data, bytes) == (ssize_t)bytes;
    close(fd);
    if (ok && stamp) *stamp = (uint64_t)st.st_ino * 0x9e3779b97f4a7c15ull ^ ((uint64_t)st.st_mtim.tv_sec * 1000000000ull + (uint64_t)st.st_mtim.tv_nsec);
    return ok;
}
}   // namespace

int athena_mp_comm_create_from_file(int32_t rank, int32_t world, const char *path, athena_mp_comm **out)
{
    AMP_REQUIRE(out != nullptr && path != nullptr, "comm_create_from_file: null argument");
    *out = nullptr;
    AMP_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_create_from_file: bad rank %d / world %d", rank, world);
    const size_t rec = 128 + 8 * (size_t)world;
    std::vector<char> buf(rec, 0);
    const double t0 = boot_now(), limit = boot_timeout();
    auto hello = [&](int r) { return std::string(path) + ".hello." + std::to_string(r); };
    if (rank == 0) {
        unlink(path);   // whatever is there now is not ours
        std::vector<uint64_t> nonce(world, 0), first(world, 0);
        std::vector<char> seen(world, 0), live(world, 0);
        int n_live = 1;
        while (n_live < world) {
            for (int r = 1; r < world; ++r) {
                if (live[r]) continue;
                uint64_t v = 0, st = 0;
                if (!boot_read(hello(r), &v, 8, &st)) continue;
                if (!seen[r]) {
                    seen[r] = 1;
                    first[r] = st;
                } else if (st != first[r]) {   // the file moved on: its writer is alive, this is its nonce
                    live[r] = 1;
                    nonce[r] = v;
                    ++n_live;
                }
            }
            AMP_REQUIRE(boot_now() - t0 < limit, "comm_create_from_file: %d of %d ranks never announced themselves at %s.hello.*",
                        world - n_live, world, path);
            if (n_live < world) usleep(5000);
        }
        if (athena_mp_comm_unique_id(buf.data())) return 1;
        memcpy(buf.data() + 128, nonce.data(), 8 * (size_t)world);
        AMP_REQUIRE(boot_write(path, buf.data(), rec), "comm_create_from_file: cannot publish %s", path);
    } else {
        uint64_t mine = 0;
        {
            int fd = open("/dev/urandom", O_RDONLY);
            AMP_REQUIRE(fd >= 0 && read(fd, &mine, 8) == 8, "comm_create_from_file: /dev/urandom unreadable");
            close(fd);
            mine |= 1;   // never 0
        }
        for (;;) {
            AMP_REQUIRE(boot_write(hello(rank), &mine, 8), "comm_create_from_file: cannot write %s", hello(rank).c_str());
            uint64_t got = 0;
            if (boot_read(path, buf.data(), rec, nullptr)) {
                memcpy(&got, buf.data() + 128 + 8 * (size_t)rank, 8);
                if (got == mine) break;   // published for THIS launch
            }
            if (boot_now() - t0 >= limit) {
                unlink(hello(rank).c_str());
                AMP_REQUIRE(false, "comm_create_from_file: %s never carried this launch's id (is rank 0 running? stale file?)", path);
            }
            usleep(20000);
        }
        unlink(hello(rank).c_str());
    }
    int rc = athena_mp_comm_create(rank, world, buf.data(), out);
    if (rc) return rc;
    rc = athena_mp_comm_barrier(*out);
    if (rank == 0) unlink(path);
    return rc;
}

int athena_mp_comm_destroy(athena_mp_comm *c)
{
    if (!c) return 0;
    if (c->cs) (void)hipStreamSynchronize(c->cs);
    delete c->t;   // (the test transport removes its directory in its destructor: succeeds for the last rank out)
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    if (c->ev_done) {
        watch_forget(c->ev_done);
        (void)hipEventDestroy(c->ev_done);
    }
    if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
    if (c->cs) (void)hipStreamDestroy(c->cs);
    delete c;
    return 0;
}

int athena_mp_comm_info(const athena_mp_comm *c, int32_t *rank, int32_t *world, char *transport, int32_t transport_len)
{
    AMP_REQUIRE(c != nullptr, "comm_info: null communicator");
    if (rank) *rank = c->t->rank;
    if (world) *world = c->t->world;
    if (transport && transport_len > 0) {
        strncpy(transport, c->t->name(), (size_t)transport_len - 1);
        transport[transport_len - 1] = 0;
    }
    return 0;
}

int athena_mp_comm_stats(const athena_mp_comm *c, int32_t *ranks_seen, int32_t *library_version, int64_t *transfers,
                         int64_t *allreduce_bytes, int64_t *sent_bytes_per_peer, int32_t n_peers)
{
    AMP_REQUIRE(c != nullptr, "comm_stats: null communicator");
    AMP_REQUIRE(n_peers >= 0 && (n_peers == 0 || sent_bytes_per_peer), "comm_stats: bad peer array");
    if (ranks_seen) *ranks_seen = c->t->ranks_seen();
    if (library_version) *library_version = c->t->version();
    if (transfers) *transfers = c->t->transfers;
    if (allreduce_bytes) *allreduce_bytes = c->t->allreduce_bytes;
    for (int p = 0; p < n_peers; ++p) sent_bytes_per_peer[p] = p < (int)c->t->sent.size() ? c->t->sent[(size_t)p] : 0;
    return 0;
}

int athena_mp_set_stall_handler(void (*handler)(int32_t rank, const char *what, double seconds))
{
    g_stall_handler.store(handler);
    return 0;
}

/* sum over ranks, in place, float32.  _start enqueues it on the communication stream behind everything the compute
 * stream has queued so far; _finish makes the compute stream wait for it (host never blocks). */
int athena_mp_allreduce_start(athena_mp_comm *c, float *buf_dev, int64_t count)
{
    AMP_REQUIRE(c && count >= 0 && (buf_dev || count == 0), "allreduce: bad arguments");
    if (c->t->world == 1 || count == 0) return 0;
    AMP_HIP(hipEventRecord(c->ev_ready, amp::stream()));
    AMP_HIP(hipStreamWaitEvent(c->cs, c->ev_ready, 0));
    AMP_HIP(hipEventRecord(c->ev_begin, c->cs));
    if (c->t->counted_allreduce_f32(buf_dev, (size_t)count, c->cs)) return 1;
    test_delay(c->cs);
    AMP_HIP(hipEventRecord(c->ev_done, c->cs));
    watch_arm(c->ev_done, c->t->rank, c->device, "the all-reduce of the parameter gradients (athena_mp_allreduce_start)", 1.0, c->ev_begin);
    return 0;
}
int athena_mp_allreduce_finish(athena_mp_comm *c)
{
    AMP_REQUIRE(c != nullptr, "allreduce_finish: null communicator");
    if (c->t->world == 1) return 0;
    AMP_HIP(hipStreamWaitEvent(amp::stream(), c->ev_done, 0));
    return 0;
}
int athena_mp_allreduce(athena_mp_comm *c, float *buf_dev, int64_t count)
{
    int rc = athena_mp_allreduce_start(c, buf_dev, count);
    return rc ? rc : athena_mp_allreduce_finish(c);
}

int athena_mp_comm_barrier(athena_mp_comm *c)
{
    AMP_REQUIRE(c != nullptr, "comm_barrier: null communicator");
    float *one = nullptr;
    if (amp::named_buffer("comm.barrier_word", 256, true, (void **)&one)) return 1;
    AMP_HIP(hipStreamSynchronize(amp::stream()));
    if (c->t->world > 1 && c->t->counted_allreduce_f32(one, 1, c->cs)) return 1;
    if (c->t->world > 1) {
        AMP_HIP(hipEventRecord(c->ev_done, c->cs));
        watch_arm(c->ev_done, c->t->rank, c->device, "athena_mp_comm_barrier", 5.0);
    }
    AMP_HIP(hipStreamSynchronize(c->cs));
    AMP_HIP(hipMemsetAsync(one, 0, 4, c->cs));
    AMP_HIP(hipStreamSynchronize(c->cs));
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------------ */
int athena_mp_shard_destroy(athena_mp_shard *s)
{
    if (!s) return 0;
    for (auto &g : s->g)
        if (g) athena_mp_graph_destroy(g);
    if (s->send_idx) (void)hipFree(s->send_idx);
    for (int k = 0; k < 2; ++k) {
        if (s->send_buf[k]) (void)hipFree(s->send_buf[k]);
        if (s->red_buf[k]) (void)hipFree(s->red_buf[k]);
        if (k == 0 && s->eshare) (void)hipFree(s->eshare);
        if (k == 0 && s->ered_buf) (void)hipFree(s->ered_buf);
        if (s->ev_halo[k]) {
            watch_forget(s->ev_halo[k]);
            (void)hipEventDestroy(s->ev_halo[k]);
        }
        if (s->ev_halo_b[k]) (void)hipEventDestroy(s->ev_halo_b[k]);
    }
    delete s;
    return 0;
}

/* Build this rank's shard from its rows of the global graph.  adj_ia [n_local + 1] (1-based), adj_ja [2, nnz]
 * column-major with adj_ja(1, w) = GLOBAL vertex id (1-based) of the neighbour.  with_edges = 0: adj_ja(2, .) is ignored
 * (Kipf).  with_edges = 1 (athena_mp_shard_create_edges): adj_ja(2, w) = GLOBAL edge id (1-based, 0 = none) -- the
 * rank's edge-feature columns become the distinct ids its rows reference.
 * Collective: every rank of the communicator calls it. */
static int shard_create_impl(athena_mp_comm *c, int32_t n_local, int64_t nnz, const int32_t *adj_ia, const int32_t *adj_ja,
                             int32_t with_edges, athena_mp_shard **out)
{
    AMP_REQUIRE(out != nullptr, "shard_create: null out pointer");
    *out = nullptr;
    AMP_REQUIRE(c && n_local >= 0 && nnz >= 0 && adj_ia && (adj_ja || nnz == 0), "shard_create: bad arguments");
    AMP_REQUIRE(adj_ia[0] == 1 && (int64_t)adj_ia[n_local] - 1 == nnz, "shard_create: adj_ia does not span nnz = %lld",
                (long long)nnz);
    AMP_REQUIRE(nnz < (int64_t)INT32_MAX, "shard_create: nnz %lld exceeds the int32 CSR of one shard", (long long)nnz);
    const int W = c->t->world, rank = c->t->rank;
    athena_mp_shard *s = new athena_mp_shard();
    s->comm = c;
    s->n = n_local;
    s->nnz = nnz;
    const int32_t n = n_local;
#define SH_FAIL(...)                \
    do {                            \
        set_error(__VA_ARGS__);     \
        athena_mp_shard_destroy(s); \
        return 2;                   \
    } while (0)
#define SH_RC(expr)                     \
    do {                                \
        if ((expr) != 0) {              \
            athena_mp_shard_destroy(s); \
            return 1;                   \
        }                               \
    } while (0)
    // 1. row blocks of every rank; local validation first, agreed on before any data collective (a rank that bails
    //    out alone would leave its peers blocked in the next exchange)
    int32_t bad = 0;
    for (int32_t v = 0; v < n && !bad; ++v)
        if (adj_ia[v + 1] < adj_ia[v]) {
            set_error("shard_create: adj_ia not monotone at row %d", v + 1);
            bad = 1;
        }
    {
        const int arc = agree_ok(c, bad, "shard_create");
        if (arc) {
            athena_mp_shard_destroy(s);
            return arc;
        }
    }
    std::vector<int64_t> all_n(W);
    const int64_t my_n = n;
    SH_RC(allgather_host(c, &my_n, 8, all_n.data()));
    s->row_off.assign(W + 1, 0);
    for (int p = 0; p < W; ++p) s->row_off[p + 1] = s->row_off[p] + all_n[p];
    s->n_total = s->row_off[W];
    s->max_n = *std::max_element(all_n.begin(), all_n.end());
    const int64_t lo = s->row_off[rank], hi = lo + n;
    s->row_offset = lo;
    // 2. interior-first numbering
    std::vector<char> is_bnd(n, 0);
    for (int32_t v = 0; v < n && !bad; ++v) {
        for (int32_t w = adj_ia[v] - 1; w < adj_ia[v + 1] - 1; ++w) {
            const int64_t u = (int64_t)adj_ja[2 * (size_t)w] - 1;
            if (u < 0 || u >= s->n_total) {
                set_error("shard_create: adj_ja(1,%d) = %lld outside [1,%lld]", w + 1, (long long)u + 1, (long long)s->n_total);
                bad = 1;
                break;
            }
            if (u < lo || u >= hi) is_bnd[v] = 1;
        }
    }
    {
        const int arc = agree_ok(c, bad, "shard_create");
        if (arc) {
            athena_mp_shard_destroy(s);
            return arc;
        }
    }
    s->order.resize(n);
    {
        int32_t k = 0;
        for (int32_t v = 0; v < n; ++v)
            if (!is_bnd[v]) s->order[k++] = v;
        s->n_int = k;
        for (int32_t v = 0; v < n; ++v)
            if (is_bnd[v]) s->order[k++] = v;
    }
    std::vector<int32_t> new_of_old(n);
    for (int32_t k = 0; k < n; ++k) new_of_old[s->order[k]] = k;
    // 3. rows in the new order (entry order inside a row kept), halo ids
    std::vector<int32_t> ia(n + 1, 0);
    s->row_deg.resize(n);
    for (int32_t k = 0; k < n; ++k) {
        const int32_t v = s->order[k];
        s->row_deg[k] = adj_ia[v + 1] - adj_ia[v];
        ia[k + 1] = ia[k] + s->row_deg[k];
    }
    std::vector<int64_t> cg(nnz);   // global column of every entry, new row order
    for (int32_t k = 0; k < n; ++k) {
        const int32_t v = s->order[k];
        int32_t q = ia[k];
        for (int32_t w = adj_ia[v] - 1; w < adj_ia[v + 1] - 1; ++w) cg[q++] = (int64_t)adj_ja[2 * (size_t)w] - 1;
    }
    for (int64_t w = 0; w < nnz; ++w)
        if (cg[w] < lo || cg[w] >= hi) s->halo_ids.push_back(cg[w]);
    std::sort(s->halo_ids.begin(), s->halo_ids.end());
    s->halo_ids.erase(std::unique(s->halo_ids.begin(), s->halo_ids.end()), s->halo_ids.end());
    if (s->halo_ids.size() + (size_t)n >= (size_t)INT32_MAX) SH_FAIL("shard_create: local + halo rows exceed int32");
    s->n_halo = (int32_t)s->halo_ids.size();
    s->recv_counts.assign(W, 0);
    for (int64_t u : s->halo_ids) {
        const int p = (int)(std::upper_bound(s->row_off.begin(), s->row_off.end(), u) - s->row_off.begin()) - 1;
        s->recv_counts[p]++;
    }
    // 3b. edge-feature columns (shards of ONE graph with edge features: GNO on a mesh).  The reverse pass
    //     of such a shard is a PULL over the rank's own rows -- dx_v = sum_{(u,e) in row v} K_e^T g_u -- which equals the
    //     reference's scatter (athena_diffstruc_extd_sub_nop.f90:419-458) exactly when row u lists (v, e) whenever row v
    //     lists (u, e): athena's undirected graphs, both directions sharing one edge column (:369-376).  That property
    //     is CHECKED here over all ranks: every entry adds +h(v,u,e) or -h(u,v,e) to a 64-bit sum that must cancel.
    std::vector<int32_t> el(with_edges ? (size_t)nnz : 0), el_b(with_edges ? (size_t)nnz : 0);
    if (!with_edges) {
        // the Kipf shard's reverse pass is a pull over the rank's own rows too (get_partial_kipf_propagate_left_val,
        // athena_diffstruc_extd_sub_kipf.f90:85-111, read from the receiving side): the same check, without edge columns --
        // a directed graph is an error on every rank, not a silently different dX
        uint64_t signed_sum = 0;
        for (int64_t w = 0, k = 0; k < n; ++k)
            for (; w < ia[k + 1]; ++w) {
                const int64_t vg = lo + s->order[k], ug = cg[w];
                if (ug == vg) continue;
                uint64_t h = (uint64_t)std::min(ug, vg) * 0x9e3779b97f4a7c15ull;
                h = (h ^ (h >> 29)) + (uint64_t)std::max(ug, vg) * 0xbf58476d1ce4e5b9ull;
                h ^= h >> 30;
                signed_sum += vg < ug ? h : (uint64_t)0 - h;
            }
        std::vector<uint64_t> sums(W);
        SH_RC(allgather_host(c, &signed_sum, 8, sums.data()));
        uint64_t tot = 0;
        for (uint64_t x : sums) tot += x;
        if (tot != 0)
            SH_FAIL("shard_create: the graph is not undirected (row u must list v as often as row v lists u): the pull form of the "
                    "reverse pass would not equal the reference's scatter");
    }
    if (with_edges) {
        s->with_edges = 1;
        std::vector<int64_t> eg(nnz);   // global edge id (0-based, -1 = none) of every entry, new row order
        uint64_t signed_sum = 0;
        bad = 0;
        for (int32_t k = 0; k < n; ++k) {
            const int32_t v = s->order[k];
            const int64_t vg = lo + v;
            int32_t q = ia[k];
            for (int32_t w = adj_ia[v] - 1; w < adj_ia[v + 1] - 1; ++w, ++q) {
                const int64_t e = (int64_t)adj_ja[2 * (size_t)w + 1] - 1;
                if (e < -1) {
                    if (!bad) set_error("shard_create_edges: adj_ja(2,%d) = %lld is negative", w + 1, (long long)e + 1);
                    bad = 1;
                }
                eg[q] = e;
                const int64_t ug = cg[q];
                if (ug != vg) {
                    uint64_t h = (uint64_t)std::min(ug, vg) * 0x9e3779b97f4a7c15ull;
                    h = (h ^ (h >> 29)) + (uint64_t)std::max(ug, vg) * 0xbf58476d1ce4e5b9ull;
                    h = (h ^ (h >> 31)) + (uint64_t)(e + 1) * 0x94d049bb133111ebull;
                    h ^= h >> 30;
                    signed_sum += vg < ug ? h : (uint64_t)0 - h;
                }
            }
        }
        std::vector<uint64_t> sums(W);
        SH_RC(allgather_host(c, &signed_sum, 8, sums.data()));
        uint64_t tot = 0;
        for (uint64_t x : sums) tot += x;
        if (tot != 0 && !bad) {
            set_error("shard_create_edges: the graph is not undirected with shared edge columns (row u must list (v, e) whenever row v "
                      "lists (u, e)): the pull form of the reverse pass would not equal the reference's scatter");
            bad = 1;
        }
        {
            const int arc = agree_ok(c, bad, "shard_create_edges");
            if (arc) {
                athena_mp_shard_destroy(s);
                return arc;
            }
        }
        for (int64_t w = 0; w < nnz; ++w)
            if (eg[w] >= 0) s->edge_ids.push_back(eg[w]);
        std::sort(s->edge_ids.begin(), s->edge_ids.end());
        s->edge_ids.erase(std::unique(s->edge_ids.begin(), s->edge_ids.end()), s->edge_ids.end());
        if (s->edge_ids.size() >= (size_t)INT32_MAX) SH_FAIL("shard_create_edges: edge columns exceed int32");
        s->n_edge_cols = (int32_t)s->edge_ids.size();
        for (int64_t w = 0; w < nnz; ++w)
            el[w] = eg[w] < 0 ? 0 : 1 + (int32_t)(std::lower_bound(s->edge_ids.begin(), s->edge_ids.end(), eg[w]) - s->edge_ids.begin());
        // the columns the partition cuts, per peer (local ids ascend with the global ids: edge_ids is sorted)
        std::vector<std::vector<int32_t>> share(W);
        for (int64_t w = 0; w < nnz; ++w) {
            if (el[w] == 0 || (cg[w] >= lo && cg[w] < hi)) continue;
            const int p = (int)(std::upper_bound(s->row_off.begin(), s->row_off.end(), cg[w]) - s->row_off.begin()) - 1;
            share[p].push_back(el[w] - 1);
        }
        s->eoff.assign(W + 1, 0);
        for (int p = 0; p < W; ++p) {
            std::sort(share[p].begin(), share[p].end());
            share[p].erase(std::unique(share[p].begin(), share[p].end()), share[p].end());
            s->eoff[p + 1] = s->eoff[p] + (int64_t)share[p].size();
            s->eshare_h.insert(s->eshare_h.end(), share[p].begin(), share[p].end());
        }
        // athena_mp_shard_edge_reduce adds every peer's share in ONE launch, which is only race-free when a cut column is shared
        // with exactly one peer.  An adj_ja(2,:) that reuses one edge id for vertex pairs on different ranks is legal in the
        // reference (edge ids are just indices into the edge features) and would break that: refused here, on every rank.
        {
            std::vector<int32_t> all_cut(s->eshare_h);
            std::sort(all_cut.begin(), all_cut.end());
            const int dup = std::adjacent_find(all_cut.begin(), all_cut.end()) != all_cut.end();
            if (dup)
                set_error("shard_create_edges: an edge column of this rank is cut towards more than one peer (one edge id used for vertex "
                          "pairs on different ranks): the edge-column reduce would add racing shares");
            const int arc = agree_ok(c, dup, "shard_create_edges");
            if (arc) {
                athena_mp_shard_destroy(s);
                return arc;
            }
        }
        // both sides of a cut must list the same number of columns (they do when the symmetry check passed); agreed on
        // here so that a mismatch is an error on every rank and not a hang in the first edge reduce
        std::vector<int64_t> mine(W), theirs((size_t)W * W);
        for (int p = 0; p < W; ++p) mine[p] = s->eoff[p + 1] - s->eoff[p];
        SH_RC(allgather_host(c, mine.data(), 8 * (size_t)W, theirs.data()));
        for (int q = 0; q < W; ++q)
            for (int p = 0; p < W; ++p)
                if (theirs[(size_t)q * W + p] != theirs[(size_t)p * W + q])
                    SH_FAIL("shard_create_edges: ranks %d and %d disagree about the edge columns their cut shares (%lld / %lld)", q, p,
                            (long long)theirs[(size_t)q * W + p], (long long)theirs[(size_t)p * W + q]);
        if (!s->eshare_h.empty()) {
            if (hipMalloc((void **)&s->eshare, 4 * s->eshare_h.size()) != hipSuccess ||
                hipMemcpy(s->eshare, s->eshare_h.data(), 4 * s->eshare_h.size(), hipMemcpyHostToDevice) != hipSuccess)
                SH_FAIL("shard_create_edges: upload of the cut edge columns failed");
        }
    }
    // 4. who needs how much of whom, and from that the way the halo travels
    std::vector<int64_t> allc((size_t)W * W);
    SH_RC(allgather_host(c, s->recv_counts.data(), 8 * (size_t)W, allc.data()));   // allc[q*W + p] = rows q needs from p
    {
        int64_t need = 0;
        for (int64_t v : allc) need += v;
        const double remote = (double)(W - 1) * (double)s->n_total;
        s->halo_fraction = remote > 0 ? (double)need / remote : 0.0;
        s->tau = halo_tau();
        const int forced = halo_mode_env();
        s->mode = W > 1 && (forced >= 0 ? forced : (s->halo_fraction > s->tau ? 1 : 0));
        if (s->mode == 1 && (s->max_n * (int64_t)(W + 1) >= (int64_t)INT32_MAX)) s->mode = 0;   // int32 column ids
    }
    s->send_counts.assign(W, 0);
    s->roff.assign(W + 1, 0);
    s->soff.assign(W + 1, 0);
    std::vector<int32_t> col(nnz), col_b(nnz);
    if (s->mode == 0) {
        // 5a. grouped send / recv: columns [local | distinct halo rows], send lists learnt from the peers
        for (int64_t w = 0; w < nnz; ++w) {
            const int64_t u = cg[w];
            col[w] = (u >= lo && u < hi) ? new_of_old[u - lo]
                                         : n + (int32_t)(std::lower_bound(s->halo_ids.begin(), s->halo_ids.end(), u) - s->halo_ids.begin());
        }
        for (int q = 0; q < W; ++q)
            if (q != rank) s->send_counts[q] = allc[(size_t)q * W + rank];
        for (int p = 0; p < W; ++p) {
            s->roff[p + 1] = s->roff[p] + s->recv_counts[p];
            s->soff[p + 1] = s->soff[p] + s->send_counts[p];
        }
        s->n_send = s->soff[W];
        std::vector<int32_t> want(s->n_halo), asked(s->n_send);
        for (int32_t k = 0; k < s->n_halo; ++k) {
            const int p = (int)(std::upper_bound(s->row_off.begin(), s->row_off.end(), s->halo_ids[k]) - s->row_off.begin()) - 1;
            want[k] = (int32_t)(s->halo_ids[k] - s->row_off[p]);   // owner-local ORIGINAL id
        }
        {
            std::vector<const void *> sp(W);
            std::vector<void *> rp(W);
            std::vector<size_t> sb(W), rb(W);
            for (int p = 0; p < W; ++p) {
                sp[p] = want.data() + s->roff[p];
                sb[p] = 4 * (size_t)s->recv_counts[p];
                rp[p] = asked.data() + s->soff[p];
                rb[p] = 4 * (size_t)s->send_counts[p];
            }
            sb[rank] = rb[rank] = 0;
            SH_RC(exchange_host(c, sp, sb, rp, rb));
        }
        s->send_idx_h.resize(s->n_send);
        std::vector<int32_t> sdeg(s->n_send), hdeg(s->n_halo);
        bad = 0;
        for (int64_t i = 0; i < s->n_send; ++i) {
            if (asked[i] < 0 || asked[i] >= n) {
                set_error("shard_create: a peer asked for row %d of %d", asked[i], n);
                bad = 1;
                break;
            }
            s->send_idx_h[i] = new_of_old[asked[i]];
            sdeg[i] = s->row_deg[s->send_idx_h[i]];
        }
        {
            const int arc = agree_ok(c, bad, "shard_create");
            if (arc) {
                athena_mp_shard_destroy(s);
                return arc;
            }
        }
        {   // degrees of the halo columns from their owners
            std::vector<const void *> sp(W);
            std::vector<void *> rp(W);
            std::vector<size_t> sb(W), rb(W);
            for (int p = 0; p < W; ++p) {
                sp[p] = sdeg.data() + s->soff[p];
                sb[p] = 4 * (size_t)s->send_counts[p];
                rp[p] = hdeg.data() + s->roff[p];
                rb[p] = 4 * (size_t)s->recv_counts[p];
            }
            sb[rank] = rb[rank] = 0;
            SH_RC(exchange_host(c, sp, sb, rp, rb));
        }
        s->col_deg = s->row_deg;
        s->col_deg.insert(s->col_deg.end(), hdeg.begin(), hdeg.end());
        s->ext_rows = s->n_halo;
        s->ext_ids = s->halo_ids;
    } else {
        // 5b. all-gather of whole blocks: x_ext = [max_n local slots (n used) | world blocks of max_n slots, block p in
        //     rank p's OWN interior-first order] -- the halo rows are a view into the gathered blocks by block offset.
        //     Every rank publishes where its vertices went (new_of_old) and their degrees; nothing else is needed.
        const int64_t M = s->max_n;
        std::vector<int32_t> mine(2 * (size_t)M, -1), all(2 * (size_t)M * W);
        for (int32_t v = 0; v < n; ++v) {
            mine[v] = new_of_old[v];                        // [0, M): position of original local vertex v
            mine[(size_t)M + new_of_old[v]] = adj_ia[v + 1] - adj_ia[v];   // [M, 2M): degree of the vertex at each position
        }
        SH_RC(allgather_host(c, mine.data(), 8 * (size_t)M, all.data()));
        s->ext_rows = (int32_t)(M - n + M * W);
        s->col_deg.assign((size_t)n + s->ext_rows, 0);
        std::copy(s->row_deg.begin(), s->row_deg.end(), s->col_deg.begin());
        s->ext_ids.assign(s->ext_rows, -1);
        for (int p = 0; p < W; ++p) {
            const int32_t *pos = all.data() + 2 * (size_t)M * p, *deg = pos + M;
            const int64_t base = M + M * p;                 // first row of block p in x_ext
            for (int64_t i = 0; i < all_n[p]; ++i) {
                s->col_deg[base + i] = deg[i];
                s->ext_ids[base - n + pos[i]] = s->row_off[p] + i;
            }
        }
        for (int64_t w = 0; w < nnz; ++w) {
            const int64_t u = cg[w];
            if (u >= lo && u < hi) {
                col[w] = new_of_old[u - lo];
            } else {
                const int p = (int)(std::upper_bound(s->row_off.begin(), s->row_off.end(), u) - s->row_off.begin()) - 1;
                col[w] = (int32_t)(M + M * p + all[2 * (size_t)M * p + (u - s->row_off[p])]);
            }
        }
        for (int p = 0; p < W; ++p) {                       // what crosses a link per exchange: every other rank's block
            s->roff[p + 1] = s->roff[p] + (p == rank ? 0 : all_n[p]);
            s->soff[p + 1] = s->soff[p];
        }
        s->n_send = 0;
    }
    {   // backward rows: entries sorted by global source id (stable) -- the reference's scatter order
        std::vector<int32_t> perm;
        for (int32_t k = 0; k < n; ++k) {
            const int32_t b = ia[k], e = ia[k + 1];
            perm.resize(e - b);
            std::iota(perm.begin(), perm.end(), b);
            std::stable_sort(perm.begin(), perm.end(), [&](int32_t x, int32_t y) { return cg[x] < cg[y]; });
            for (int32_t i = 0; i < e - b; ++i) col_b[b + i] = col[perm[i]];
            if (with_edges)
                for (int32_t i = 0; i < e - b; ++i) el_b[b + i] = el[perm[i]];
        }
    }
    if (hipMalloc((void **)&s->send_idx, 4 * (size_t)std::max<int64_t>(s->n_send, 1)) != hipSuccess)
        SH_FAIL("shard_create: device allocation failed");
    if (s->n_send && hipMemcpy(s->send_idx, s->send_idx_h.data(), 4 * (size_t)s->n_send, hipMemcpyHostToDevice) != hipSuccess)
        SH_FAIL("shard_create: upload of the send list failed");
    // 6. the four row blocks as graph handles (rectangular: n + ext_rows columns, explicit degrees)
    const int32_t ncols = n + s->ext_rows;
    const std::vector<int32_t> *pe = with_edges ? &el : nullptr, *pe_b = with_edges ? &el_b : nullptr;
    SH_RC(make_graph(ia, col, pe, s->n_edge_cols, 0, s->n_int, ncols, s->row_deg, s->col_deg, &s->g[0]));
    SH_RC(make_graph(ia, col, pe, s->n_edge_cols, s->n_int, n, ncols, s->row_deg, s->col_deg, &s->g[1]));
    SH_RC(make_graph(ia, col_b, pe_b, s->n_edge_cols, 0, s->n_int, ncols, s->row_deg, s->col_deg, &s->g[2]));
    SH_RC(make_graph(ia, col_b, pe_b, s->n_edge_cols, s->n_int, n, ncols, s->row_deg, s->col_deg, &s->g[3]));
    for (int k = 0; k < 2; ++k)
        if (hipEventCreateWithFlags(&s->ev_halo[k], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->ev_halo_b[k], hipEventDisableTiming) != hipSuccess)
            SH_FAIL("shard_create: cannot create events");
#undef SH_FAIL
#undef SH_RC
    *out = s;
    return 0;
}

int athena_mp_shard_create(athena_mp_comm *c, int32_t n_local, int64_t nnz, const int32_t *adj_ia, const int32_t *adj_ja,
                           athena_mp_shard **out)
{
    return shard_create_impl(c, n_local, nnz, adj_ia, adj_ja, 0, out);
}

/* The shard of ONE graph with edge features (graph_nop_layer on a partitioned mesh, SURVEY.md 8e: "GNO: as Kipf plus
 * replicated theta and all-reduce of dtheta"; duvenaud_propagate's edge half takes the same handles).  adj_ja(2, w) =
 * global edge id (1-based; 0 = none) as graph_type%adj_ja carries it.  The four row-block handles then keep the edge
 * columns, renumbered to the rank's own set: the rank holds coords / edge features for exactly the columns
 * athena_mp_shard_export(7) lists (ascending global id), n_edge_cols of them (athena_mp_shard_edge_cols).  The graph must be
 * undirected with both directions of a pair sharing one column -- checked over all ranks, an error on every rank if not. */
int athena_mp_shard_create_edges(athena_mp_comm *c, int32_t n_local, int64_t nnz, const int32_t *adj_ia, const int32_t *adj_ja,
                                 athena_mp_shard **out)
{
    return shard_create_impl(c, n_local, nnz, adj_ia, adj_ja, 1, out);
}

int athena_mp_shard_edge_cols(const athena_mp_shard *s, int32_t *n_edge_cols)
{
    AMP_REQUIRE(s && n_edge_cols, "shard_edge_cols: null argument");
    *n_edge_cols = s->n_edge_cols;
    return 0;
}

int athena_mp_shard_dims(const athena_mp_shard *s, int32_t *n_local, int32_t *n_interior, int32_t *n_halo, int64_t *nnz,
                         int64_t *row_offset, int64_t *n_total)
{
    AMP_REQUIRE(s != nullptr, "shard_dims: null shard");
    if (n_local) *n_local = s->n;
    if (n_interior) *n_interior = s->n_int;
    if (n_halo) *n_halo = s->ext_rows;   // rows of x_ext beyond the local ones (the distinct halo rows in p2p mode)
    if (nnz) *nnz = s->nnz;
    if (row_offset) *row_offset = s->row_offset;
    if (n_total) *n_total = s->n_total;
    return 0;
}

/* how this shard's halo travels: *mode 0 = grouped send / recv of packed rows, 1 = all-gather of whole blocks;
 * *fraction = (distinct halo rows, summed over the ranks) / ((world - 1) * n_total); *tau = the threshold in force;
 * *recv_rows = rows that cross a link into this rank per exchange.  The choice is a property of the partition (it fixes
 * the column numbering of the shard's graphs), identical on every rank. */
int athena_mp_shard_info(const athena_mp_shard *s, int32_t *mode, double *fraction, double *tau, int64_t *recv_rows)
{
    AMP_REQUIRE(s != nullptr, "shard_info: null shard");
    if (mode) *mode = s->mode;
    if (fraction) *fraction = s->halo_fraction;
    if (tau) *tau = s->tau;
    if (recv_rows) *recv_rows = s->roff.empty() ? 0 : s->roff.back();
    return 0;
}

/* which: 0 fwd interior rows [0, n_int), 1 fwd boundary rows [n_int, n), 2 / 3 the same blocks of the backward (pull)
 * graph.  Columns index x_ext = [n local rows | n_halo halo rows].  Owned by the shard. */
int athena_mp_shard_graph(const athena_mp_shard *s, int32_t which, athena_mp_graph **g)
{
    AMP_REQUIRE(s && g && which >= 0 && which < 4, "shard_graph: bad arguments");
    *g = s->g[which];
    return 0;
}

/* arrays of the plan (tests, and hosts that keep their data in original order):
 * 0 order [n] int32 (original local id, 0-based, of new row k)   1 halo_ids [n_halo] int64 (global, 0-based)
 * 2 send_idx [n_send] int32   3 col_deg [n + n_halo] int32   4 send_counts [world] int64   5 recv_counts [world] int64
 * 6 ext_ids [n_halo of shard_dims] int64: global id (0-based) held by each row of x_ext beyond the local ones, -1 = a
 *   padding slot of the all-gather layout (== array 1 in p2p mode; array 1 is always the DISTINCT remote rows referenced)
 * 7 edge_ids [n_edge_cols] int64: global edge id (0-based) of each local edge column (athena_mp_shard_create_edges)
 * 8 cut edge columns (local ids, int32) grouped by peer   9 their offsets per peer [world+1] int64
 * count is in ELEMENTS; host_dst may be null for a size query. */
int athena_mp_shard_export(const athena_mp_shard *s, int32_t which, void *host_dst, int64_t capacity, int64_t *count)
{
    AMP_REQUIRE(s && count, "shard_export: null argument");
    const void *src = nullptr;
    int64_t n = 0;
    size_t el = 4;
    switch (which) {
    case 0: src = s->order.data(); n = (int64_t)s->order.size(); break;
    case 1: src = s->halo_ids.data(); n = (int64_t)s->halo_ids.size(); el = 8; break;
    case 2: src = s->send_idx_h.data(); n = (int64_t)s->send_idx_h.size(); break;
    case 3: src = s->col_deg.data(); n = (int64_t)s->col_deg.size(); break;
    case 4: src = s->send_counts.data(); n = (int64_t)s->send_counts.size(); el = 8; break;
    case 5: src = s->recv_counts.data(); n = (int64_t)s->recv_counts.size(); el = 8; break;
    case 6: src = s->ext_ids.data(); n = (int64_t)s->ext_ids.size(); el = 8; break;
    case 7: src = s->edge_ids.data(); n = (int64_t)s->edge_ids.size(); el = 8; break;
    case 8: src = s->eshare_h.data(); n = (int64_t)s->eshare_h.size(); break;
    case 9: src = s->eoff.data(); n = (int64_t)s->eoff.size(); el = 8; break;
    default: AMP_REQUIRE(false, "shard_export: unknown array id %d", which);
    }
    *count = n;
    if (!host_dst) return 0;
    AMP_REQUIRE(capacity >= n, "shard_export: buffer holds %lld elements, array has %lld", (long long)capacity, (long long)n);
    if (n) memcpy(host_dst, src, el * (size_t)n);
    return 0;
}

/* Halo exchange of x_ext [n + n_halo, F] (row-major; n_halo as athena_mp_shard_dims reports it).  p2p mode: packs the
 * rows the peers need (HIP gather on the compute stream), then -- on the communication stream, ordered behind the pack --
 * one grouped send/recv per peer straight into the halo rows of x_ext.  all-gather mode (athena_mp_shard_info): one
 * ncclAllGather of every rank's block into the rows behind the local slots; nothing is packed.  Returns at once; kernels enqueued before athena_mp_halo_finish (the interior rows) run
 * under the transfer.  slot 0 / 1: two exchanges may be outstanding (e.g. X and dZ). */
int athena_mp_halo_start(athena_mp_shard *s, int32_t slot, int32_t F, float *x_ext_dev)
{
    AMP_REQUIRE(s && (slot == 0 || slot == 1) && F > 0 && (x_ext_dev || s->n + s->ext_rows == 0), "halo_start: bad arguments");
    athena_mp_comm *c = s->comm;
    const int W = c->t->world, rank = c->t->rank;
    if (W == 1) return 0;
    if (s->mode == 1) {
        // whole blocks: the first max_n rows of x_ext (this rank's n rows + padding slots) go out as they lie, block p of
        // every rank lands at row max_n * (1 + p).  No pack kernel, no send list, one collective.
        AMP_HIP(hipEventRecord(c->ev_ready, amp::stream()));
        AMP_HIP(hipStreamWaitEvent(c->cs, c->ev_ready, 0));
        AMP_HIP(hipEventRecord(s->ev_halo_b[slot], c->cs));
        const size_t block = sizeof(float) * (size_t)s->max_n * F;
        if (c->t->counted_allgather(x_ext_dev, x_ext_dev + (size_t)s->max_n * F, block, c->cs)) return 1;
        test_delay(c->cs);
        AMP_HIP(hipEventRecord(s->ev_halo[slot], c->cs));
        watch_arm(s->ev_halo[slot], rank, c->device, slot ? "the halo exchange in slot 1 (all-gather of whole blocks)" : "the halo exchange in slot 0 (all-gather of whole blocks)",
                  1.0, s->ev_halo_b[slot]);
        return 0;
    }
    const size_t need = sizeof(float) * (size_t)std::max<int64_t>(s->n_send, 1) * F;
    if (s->send_cap[slot] < need) {
        if (s->send_buf[slot]) {
            AMP_HIP(hipStreamSynchronize(c->cs));
            AMP_HIP(hipFree(s->send_buf[slot]));
            s->send_buf[slot] = nullptr;
        }
        AMP_HIP(hipMalloc((void **)&s->send_buf[slot], need));
        s->send_cap[slot] = need;
    }
    if (s->n_send) {
        int rc = athena_mp_gather_rows(s->n_send, F, s->send_idx, x_ext_dev, s->send_buf[slot]);
        if (rc) return rc;
    }
    AMP_HIP(hipEventRecord(c->ev_ready, amp::stream()));      // the pack, and every earlier reader of the halo rows
    AMP_HIP(hipStreamWaitEvent(c->cs, c->ev_ready, 0));
    AMP_HIP(hipEventRecord(s->ev_halo_b[slot], c->cs));
    std::vector<const void *> sp(W, nullptr);
    std::vector<void *> rp(W, nullptr);
    std::vector<size_t> sb(W, 0), rb(W, 0);
    for (int p = 0; p < W; ++p) {
        if (p == rank) continue;
        sp[p] = s->send_buf[slot] + (size_t)s->soff[p] * F;
        sb[p] = sizeof(float) * (size_t)s->send_counts[p] * F;
        rp[p] = x_ext_dev + ((size_t)s->n + (size_t)s->roff[p]) * F;
        rb[p] = sizeof(float) * (size_t)s->recv_counts[p] * F;
    }
    if (c->t->counted_exchange(sp.data(), sb.data(), rp.data(), rb.data(), c->cs)) return 1;
    test_delay(c->cs);
    AMP_HIP(hipEventRecord(s->ev_halo[slot], c->cs));
    watch_arm(s->ev_halo[slot], rank, c->device, slot ? "the halo exchange in slot 1 (grouped send / recv)" : "the halo exchange in slot 0 (grouped send / recv)", 1.0,
              s->ev_halo_b[slot]);
    return 0;
}

/* Halo REDUCE (the transpose of the exchange): y_ext [n + n_halo, F] holds, in its rows beyond the local ones, what this rank
 * computed for REMOTE vertices (e.g. the feature gradient of a scatter-form reverse pass); _start sends every such row to its
 * owner (p2p mode: the contiguous segment of each owner; all-gather mode: the whole block of each owner) and receives what
 * the peers computed for this rank's rows; _finish adds the received rows into y_local [n, F] on the compute stream, peer by
 * peer in rank order (deterministic).  Same streams, events, deadline and slots as athena_mp_halo_start / _finish; a slot
 * carries either an exchange or a reduce at a time.  F must be a multiple of 4. */
int athena_mp_halo_reduce_start(athena_mp_shard *s, int32_t slot, int32_t F, const float *y_ext_dev)
{
    AMP_REQUIRE(s && (slot == 0 || slot == 1) && F > 0 && F % 4 == 0 && (y_ext_dev || s->n + s->ext_rows == 0), "halo_reduce_start: bad arguments");
    athena_mp_comm *c = s->comm;
    const int W = c->t->world, rank = c->t->rank;
    s->red_F[slot] = F;
    if (W == 1) return 0;
    const int64_t recv_rows = s->mode == 1 ? s->max_n * (int64_t)W : s->n_send;
    const size_t need = sizeof(float) * (size_t)std::max<int64_t>(recv_rows, 1) * F;
    if (s->red_cap[slot] < need) {
        if (s->red_buf[slot]) {
            AMP_HIP(hipStreamSynchronize(c->cs));
            AMP_HIP(hipStreamSynchronize(amp::stream()));
            AMP_HIP(hipFree(s->red_buf[slot]));
            s->red_buf[slot] = nullptr;
        }
        AMP_HIP(hipMalloc((void **)&s->red_buf[slot], need));
        s->red_cap[slot] = need;
    }
    AMP_HIP(hipEventRecord(c->ev_ready, amp::stream()));      // the kernels that produced y_ext, and the last reader of red_buf
    AMP_HIP(hipStreamWaitEvent(c->cs, c->ev_ready, 0));
    AMP_HIP(hipEventRecord(s->ev_halo_b[slot], c->cs));
    std::vector<const void *> sp(W, nullptr);
    std::vector<void *> rp(W, nullptr);
    std::vector<size_t> sb(W, 0), rb(W, 0);
    for (int p = 0; p < W; ++p) {
        if (p == rank) continue;
        if (s->mode == 1) {   // whole blocks: block p of x_ext goes to rank p, every peer's block for this rank comes back
            sp[p] = y_ext_dev + (size_t)s->max_n * (1 + p) * F;
            sb[p] = sizeof(float) * (size_t)s->max_n * F;
            rp[p] = s->red_buf[slot] + (size_t)s->max_n * p * F;
            rb[p] = sizeof(float) * (size_t)s->max_n * F;
        } else {              // the roles of the exchange's send and receive lists swapped
            sp[p] = y_ext_dev + ((size_t)s->n + (size_t)s->roff[p]) * F;
            sb[p] = sizeof(float) * (size_t)s->recv_counts[p] * F;
            rp[p] = s->red_buf[slot] + (size_t)s->soff[p] * F;
            rb[p] = sizeof(float) * (size_t)s->send_counts[p] * F;
        }
    }
    if (c->t->counted_exchange(sp.data(), sb.data(), rp.data(), rb.data(), c->cs)) return 1;
    test_delay(c->cs);
    AMP_HIP(hipEventRecord(s->ev_halo[slot], c->cs));
    watch_arm(s->ev_halo[slot], rank, c->device, slot ? "the halo reduce in slot 1" : "the halo reduce in slot 0", 1.0, s->ev_halo_b[slot]);
    return 0;
}

int athena_mp_halo_reduce_finish(athena_mp_shard *s, int32_t slot, float *y_local_dev)
{
    AMP_REQUIRE(s && (slot == 0 || slot == 1) && (y_local_dev || s->n == 0), "halo_reduce_finish: bad arguments");
    athena_mp_comm *c = s->comm;
    const int W = c->t->world, rank = c->t->rank, F = s->red_F[slot];
    if (W == 1) return 0;
    AMP_REQUIRE(F > 0, "halo_reduce_finish: no reduce was started in slot %d", slot);
    AMP_HIP(hipStreamWaitEvent(amp::stream(), s->ev_halo[slot], 0));
    for (int p = 0; p < W; ++p) {
        if (p == rank) continue;
        const int64_t rows = s->mode == 1 ? s->n : s->send_counts[p];
        if (rows == 0) continue;
        const float *src = s->red_buf[slot] + (size_t)(s->mode == 1 ? s->max_n * (int64_t)p : s->soff[p]) * F;
        const int32_t *idx = s->mode == 1 ? nullptr : s->send_idx + s->soff[p];
        const int64_t threads = rows * (F / 4);
        hipLaunchKernelGGL(halo_add_rows_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, amp::stream(), src, idx, rows, F,
                           y_local_dev);
        AMP_LAUNCH_CHECK();
    }
    return 0;
}

/* Sum over the partition of a per-edge-column quantity (the coordinate gradient of graph_nop_layer,
 * athena_diffstruc_extd_sub_nop.f90:137-216): e_dev [n_edge_cols, F] holds this rank's share -- the sum over ITS rows' entries;
 * an edge column the partition cuts receives a share on both sides.  Packs the cut columns per peer, exchanges them (grouped
 * send / recv on the communication stream), adds the peer's rows in: afterwards both ranks hold the full sum for the columns
 * they share (a + b on one side, b + a on the other: the same bits).  Stream-ordered on the compute stream; the host does
 * not block.  Shards built by athena_mp_shard_create_edges only. */
int athena_mp_shard_edge_reduce(athena_mp_shard *s, int32_t F, float *e_dev)
{
    AMP_REQUIRE(s && F > 0 && (e_dev || s->n_edge_cols == 0), "shard_edge_reduce: bad arguments");
    AMP_REQUIRE(s->with_edges, "shard_edge_reduce: the shard was built without edge columns (athena_mp_shard_create)");
    athena_mp_comm *c = s->comm;
    const int W = c->t->world, rank = c->t->rank;
    if (W == 1) return 0;
    const int64_t tot = s->eoff[W];
    const size_t need = sizeof(float) * 2 * (size_t)std::max<int64_t>(tot, 1) * F;
    if (s->ered_cap < need) {
        if (s->ered_buf) {
            AMP_HIP(hipStreamSynchronize(c->cs));
            AMP_HIP(hipStreamSynchronize(amp::stream()));
            AMP_HIP(hipFree(s->ered_buf));
            s->ered_buf = nullptr;
        }
        AMP_HIP(hipMalloc((void **)&s->ered_buf, need));
        s->ered_cap = need;
    }
    float *pack = s->ered_buf, *recv = s->ered_buf + (size_t)std::max<int64_t>(tot, 1) * F;
    if (tot) {
        hipLaunchKernelGGL(edge_pack_kernel, dim3((unsigned)((tot * F + 255) / 256)), dim3(256), 0, amp::stream(), e_dev, s->eshare, tot, F, pack);
        AMP_LAUNCH_CHECK();
    }
    AMP_HIP(hipEventRecord(c->ev_ready, amp::stream()));
    AMP_HIP(hipStreamWaitEvent(c->cs, c->ev_ready, 0));
    AMP_HIP(hipEventRecord(c->ev_begin, c->cs));
    std::vector<const void *> sp(W, nullptr);
    std::vector<void *> rp(W, nullptr);
    std::vector<size_t> sb(W, 0), rb(W, 0);
    for (int p = 0; p < W; ++p) {
        if (p == rank) continue;
        const size_t rows = (size_t)(s->eoff[p + 1] - s->eoff[p]);
        sp[p] = pack + (size_t)s->eoff[p] * F;
        rp[p] = recv + (size_t)s->eoff[p] * F;
        sb[p] = rb[p] = sizeof(float) * rows * F;
    }
    if (c->t->counted_exchange(sp.data(), sb.data(), rp.data(), rb.data(), c->cs)) return 1;
    AMP_HIP(hipEventRecord(c->ev_done, c->cs));
    watch_arm(c->ev_done, rank, c->device, "the edge-column reduce (athena_mp_shard_edge_reduce)", 1.0, c->ev_begin);
    AMP_HIP(hipStreamWaitEvent(amp::stream(), c->ev_done, 0));
    if (tot) {
        hipLaunchKernelGGL(edge_add_kernel, dim3((unsigned)((tot * F + 255) / 256)), dim3(256), 0, amp::stream(), recv, s->eshare, tot, F, e_dev);
        AMP_LAUNCH_CHECK();
    }
    return 0;
}

int athena_mp_halo_finish(athena_mp_shard *s, int32_t slot)
{
    AMP_REQUIRE(s && (slot == 0 || slot == 1), "halo_finish: bad arguments");
    if (s->comm->t->world == 1) return 0;
    AMP_HIP(hipStreamWaitEvent(amp::stream(), s->ev_halo[slot], 0));
    return 0;
}

} // extern "C"
