// Runtime + graph handle of libathena_mp.
//
// The graph handle replaces the reference's per-call CSR copies
// (athena_msgpass_layer_sub.f90:144-174, athena_diffstruc_extd_sub_kipf.f90:48-49):
// the Fortran 1-based CSR is validated and converted ONCE into the device layout the kernels
// stream (0-based forward CSR, transposed CSR for the pull form of every scatter, per-entry
// Kipf coefficients, and an edge-column index).
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace amp {

static thread_local char g_err[1024] = "";
static hipStream_t g_stream = nullptr;
struct NamedBuf {
    void *p = nullptr;
    size_t bytes = 0;
};
constexpr int kWsSlots = 20;
struct DevCtx {
    void *ws[kWsSlots] = {};
    size_t ws_bytes[kWsSlots] = {};
    hipStream_t aux = nullptr;            // second stream of the library (producer / consumer pipelines inside one call)
    hipEvent_t ev[6] = {};                // events of those pipelines
    std::map<std::string, NamedBuf> named;
    int cus = 0;
};
static DevCtx g_ctx[kMaxDevices];
static int g_device = 0;
static int g_inited = 0;   // athena_mp_init succeeded at least once
#define g_ws g_ctx[g_device].ws
#define g_ws_bytes g_ctx[g_device].ws_bytes

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
hipStream_t stream() { return g_stream; }
void swap_stream(hipStream_t s) { g_stream = s; }
int aux_stream(hipStream_t *s, hipEvent_t **events)
{
    DevCtx &c = g_ctx[g_device];
    if (!c.aux) {
        AMP_HIP(hipStreamCreateWithFlags(&c.aux, hipStreamNonBlocking));
        for (auto &e : c.ev) AMP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    *s = c.aux;
    *events = c.ev;
    return 0;
}
int device() { return g_device; }
int num_cus()
{
    DevCtx &c = g_ctx[g_device];
    if (c.cus <= 0) {
        hipDeviceProp_t p;
        c.cus = (hipGetDeviceProperties(&p, g_device) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    }
    return c.cus;
}

int named_buffer(const char *name, size_t bytes, bool zero_fill, void **ptr, bool *fresh)
{
    NamedBuf &b = g_ctx[g_device].named[name];
    if (fresh) *fresh = false;
    if (b.bytes < bytes || b.p == nullptr) {
        if (b.p) {
            AMP_HIP(hipStreamSynchronize(g_stream));
            AMP_HIP(hipFree(b.p));
            b.p = nullptr;
            b.bytes = 0;
        }
        AMP_HIP(hipMalloc(&b.p, bytes ? bytes : 4));
        b.bytes = bytes;
        if (zero_fill) AMP_HIP(hipMemset(b.p, 0, bytes ? bytes : 4));
        if (fresh) *fresh = true;
    }
    *ptr = b.p;
    return 0;
}

int workspace(void **ptr, size_t bytes, int slot)
{
    if (slot < 0 || slot >= kWsSlots) {
        set_error("workspace: slot %d out of range", slot);
        return 1;
    }
    if (bytes > g_ws_bytes[slot]) {
        if (g_ws[slot]) {
            AMP_HIP(hipStreamSynchronize(g_stream));
            AMP_HIP(hipFree(g_ws[slot]));
            g_ws[slot] = nullptr;
            g_ws_bytes[slot] = 0;
        }
        size_t want = bytes + (bytes >> 2) + 256;
        if (hipMalloc(&g_ws[slot], want) != hipSuccess) {   // no room for the 25 % headroom: exactly what was asked for
            (void)hipGetLastError();
            g_ws[slot] = nullptr;
            want = bytes;
            AMP_HIP(hipMalloc(&g_ws[slot], want));
        }
        g_ws_bytes[slot] = want;
    }
    *ptr = g_ws[slot];
    return 0;
}

int workspace_release(int slot)
{
    if (slot < 0 || slot >= kWsSlots || !g_ws[slot]) return 0;
    AMP_HIP(hipStreamSynchronize(g_stream));
    AMP_HIP(hipFree(g_ws[slot]));
    g_ws[slot] = nullptr;
    g_ws_bytes[slot] = 0;
    return 0;
}

} // namespace amp

using namespace amp;

extern "C" {

const char *athena_mp_last_error(void) { return g_err; }
int athena_mp_version(void) { return 100; }

int athena_mp_init(int device)
{
    int n = 0;
    AMP_HIP(hipGetDeviceCount(&n));
    AMP_REQUIRE(n > 0, "athena_mp_init: no HIP device visible");
    AMP_REQUIRE(device >= 0 && device < n, "athena_mp_init: device %d out of range [0,%d)", device, n);
    AMP_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    AMP_HIP(hipGetDeviceProperties(&prop, device));
    AMP_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
                "athena_mp_init: built for gfx950 (MI355X) only, device %d is %s", device,
                prop.gcnArchName);
    AMP_REQUIRE(device < kMaxDevices, "athena_mp_init: device %d beyond the %d this library tracks", device, kMaxDevices);
    if (device != g_device) g_stream = nullptr;   // the stream of the previous device does not carry over
    g_device = device;
    g_ctx[device].cus = prop.multiProcessorCount;
    g_inited = 1;
    return 0;
}

int athena_mp_initialized(void) { return g_inited ? g_device : -1; }

int athena_mp_finalize(void)
{
    amp::host_pool_release();
    amp::graph_cache_clear();
    for (int d = 0; d < kMaxDevices; ++d) {       // every device this process initialised
        DevCtx &c = g_ctx[d];
        for (int s = 0; s < kWsSlots; ++s) {
            if (c.ws[s]) {
                AMP_HIP(hipFree(c.ws[s]));
                c.ws[s] = nullptr;
                c.ws_bytes[s] = 0;
            }
        }
        for (auto &kv : c.named)
            if (kv.second.p) AMP_HIP(hipFree(kv.second.p));
        c.named.clear();
        if (c.aux) { (void)hipStreamDestroy(c.aux); c.aux = nullptr; }
        for (auto &e : c.ev)
            if (e) { (void)hipEventDestroy(e); e = nullptr; }
    }
    return 0;
}

int athena_mp_set_stream(void *hip_stream)
{
    g_stream = (hipStream_t)hip_stream;
    return 0;
}
int athena_mp_synchronize(void)
{
    AMP_HIP(hipStreamSynchronize(g_stream));
    return 0;
}
int athena_mp_malloc(void **p, uint64_t bytes)
{
    AMP_REQUIRE(p != nullptr, "athena_mp_malloc: null out pointer");
    *p = nullptr;
    const hipError_t e = hipMalloc(p, bytes ? bytes : 4);
    if (e != hipSuccess) {
        // callers with a fallback (the S buffer of a training-mode GNO layer) go on after a refusal: the error must not
        // stay behind as the thread's last error for the next AMP_LAUNCH_CHECK to find
        (void)hipGetLastError();
        *p = nullptr;
        amp::set_error("athena_mp_malloc: %llu bytes: %s", (unsigned long long)bytes, hipGetErrorString(e));
        return 1;
    }
    return 0;
}
int athena_mp_free(void *p)
{
    if (p) AMP_HIP(hipFree(p));
    return 0;
}
int athena_mp_memcpy_h2d(void *d, const void *h, uint64_t bytes)
{
    AMP_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, g_stream));
    AMP_HIP(hipStreamSynchronize(g_stream));
    return 0;
}
int athena_mp_memcpy_d2h(void *h, const void *d, uint64_t bytes)
{
    AMP_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, g_stream));
    AMP_HIP(hipStreamSynchronize(g_stream));
    return 0;
}
int athena_mp_memset_zero(void *d, uint64_t bytes)
{
    AMP_HIP(hipMemsetAsync(d, 0, bytes, g_stream));
    return 0;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------
// Kipf coefficient per CSR entry.  Reference: coeff = (deg_v * deg_u) ** (-0.5_real32)
// (athena_diffstruc_extd_sub_kipf.f90:39-42): integer product, converted to real32, then libm's
// powf.  It is evaluated here ON THE HOST with the same powf, once per graph, so every coefficient
// is bit-identical to what the reference computes on this machine (a device pow, or 1/sqrt through
// fp64, differs from glibc's powf by 1 ulp for a few products).  The coefficient depends only on the
// integer product, so a small direct-mapped memo makes the pass memory-bound.
// ---------------------------------------------------------------------------------------------
#include <math.h>
struct CoefMemo {
    static constexpr int kSize = 1 << 16;
    std::vector<int32_t> key;
    std::vector<float> val;
    CoefMemo() : key(kSize, -1), val(kSize, 0.0f) {}
    inline float get(int32_t prod)
    {
        const uint32_t slot = ((uint32_t)prod * 2654435761u) >> 16;
        if (key[slot] != prod) {
            key[slot] = prod;
            val[slot] = powf((float)prod, -0.5f);
        }
        return val[slot];
    }
};

template <typename T> static int upload(T **dst, const std::vector<T> &src)
{
    size_t bytes = sizeof(T) * (src.size() ? src.size() : 1);
    AMP_HIP(hipMalloc((void **)dst, bytes));
    if (!src.empty())
        AMP_HIP(hipMemcpyAsync(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice, stream()));
    return 0;
}

static int build_long_plan(const std::vector<int32_t> &rowptr, int32_t n_rows, LongPlan *lp)
{
    std::vector<int32_t> beg, end, rid, t0;
    for (int32_t r = 0; r < n_rows; ++r) {
        const int32_t len = rowptr[r + 1] - rowptr[r];
        if (len <= kLongRow) continue;
        rid.push_back(r);
        t0.push_back((int32_t)beg.size());
        for (int32_t b = rowptr[r]; b < rowptr[r + 1]; b += kLongRow) {
            beg.push_back(b);
            end.push_back(std::min(b + kLongRow, rowptr[r + 1]));
        }
    }
    t0.push_back((int32_t)beg.size());
    lp->n_tasks = (int32_t)beg.size();
    lp->n_long = (int32_t)rid.size();
    if (lp->n_long == 0) return 0;
    int rc = 0;
    rc |= upload(&lp->task_beg, beg);
    rc |= upload(&lp->task_end, end);
    rc |= upload(&lp->row_id, rid);
    rc |= upload(&lp->row_task0, t0);
    if (rc == 0) AMP_HIP(hipStreamSynchronize(stream()));   // beg / end / rid / t0 die at return: the copies must be done
    return rc;
}

extern "C" {

static int graph_free(athena_mp_graph *g);

int athena_mp_graph_destroy(athena_mp_graph *g)
{
    if (!g) return 0;
    if (g->cache_refs >= 0) return athena_mp_graph_release(g);   // a handle of the cache: one user less
    return graph_free(g);
}

static int graph_free(athena_mp_graph *g)
{
    void *ptrs[] = {g->rowptr, g->col,   g->eid,      g->coef,  g->t_rowptr, g->t_src,  g->t_eid,
                    g->t_coef, g->e_rowptr, g->e_row, g->e_col, g->deg_row,  g->deg_col};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (g->bucket_perm) (void)hipFree(g->bucket_perm);
    if (g->t_entry) (void)hipFree(g->t_entry);
    for (int32_t *lp : {g->len_perm_fwd, g->len_perm_bwd})
        if (lp) {
            amp::gno_forget_perm(lp);
            (void)hipFree(lp);
        }
    if (g->btile_start) (void)hipFree(g->btile_start);
    if (g->btile_info) (void)hipFree(g->btile_info);
    if (g->btile_rows) (void)hipFree(g->btile_rows);
    if (g->btile_off_dev) (void)hipFree(g->btile_off_dev);
    for (LongPlan *lp : {&g->lp_fwd, &g->lp_bwd})
        for (void *p : {(void *)lp->task_beg, (void *)lp->task_end, (void *)lp->row_id, (void *)lp->row_task0})
            if (p) (void)hipFree(p);
    delete g;
    return 0;
}

int athena_mp_graph_dims(const athena_mp_graph *g, int32_t *n_rows, int32_t *n_cols, int64_t *nnz,
                         int32_t *n_edge_cols)
{
    AMP_REQUIRE(g != nullptr, "athena_mp_graph_dims: null graph");
    if (n_rows) *n_rows = g->n_rows;
    if (n_cols) *n_cols = g->n_cols;
    if (nnz) *nnz = g->nnz;
    if (n_edge_cols) *n_edge_cols = g->n_edge_cols;
    return 0;
}

int athena_mp_graph_export(const athena_mp_graph *g, int32_t which, void *host_dst, int64_t capacity,
                           int64_t *count)
{
    AMP_REQUIRE(g && count, "graph_export: null argument");
    const void *src = nullptr;
    int64_t n = 0;
    switch (which) {
    case 0: src = g->rowptr; n = (int64_t)g->n_rows + 1; break;
    case 1: src = g->col; n = g->nnz; break;
    case 2: src = g->eid; n = g->nnz; break;
    case 3: src = g->coef; n = g->nnz; break;
    case 4: src = g->t_rowptr; n = (int64_t)g->n_cols + 1; break;
    case 5: src = g->t_src; n = g->nnz; break;
    case 6: src = g->t_eid; n = g->nnz; break;
    case 7: src = g->t_coef; n = g->nnz; break;
    case 8: src = g->e_rowptr; n = (int64_t)g->n_edge_cols + 1; break;
    case 9: src = g->e_row; n = g->n_with_edge; break;
    case 10: src = g->e_col; n = g->n_with_edge; break;
    case 11: src = g->deg_row; n = g->n_rows; break;
    case 12: src = g->deg_col; n = g->n_cols; break;
    default: AMP_REQUIRE(false, "graph_export: unknown array id %d", which);
    }
    *count = n;
    if (host_dst == nullptr) return 0;   // size query
    AMP_REQUIRE(capacity >= n, "graph_export: buffer holds %lld elements, array has %lld", (long long)capacity, (long long)n);
    if (n > 0) {
        AMP_HIP(hipMemcpyAsync(host_dst, src, 4 * (size_t)n, hipMemcpyDeviceToHost, g_stream));
        AMP_HIP(hipStreamSynchronize(g_stream));
    }
    return 0;
}

static int graph_create_impl(int32_t n_rows, int32_t n_cols, int64_t nnz, const int32_t *adj_ia,
                             const int32_t *adj_ja, const int32_t *adj_ja_dev, int32_t n_edge_cols,
                             const int32_t *row_deg, const int32_t *col_deg, athena_mp_graph **out);

int athena_mp_graph_create(int32_t n_rows, int32_t n_cols, int64_t nnz, const int32_t *adj_ia,
                           const int32_t *adj_ja, int32_t n_edge_cols, const int32_t *row_deg,
                           const int32_t *col_deg, athena_mp_graph **out)
{
    AMP_REQUIRE(out != nullptr, "graph_create: null out pointer");
    *out = nullptr;
    AMP_REQUIRE(nnz <= 0 || adj_ja != nullptr, "graph_create: null CSR arrays");
    return graph_create_impl(n_rows, n_cols, nnz, adj_ia, adj_ja, nullptr, n_edge_cols, row_deg, col_deg, out);
}

/* edge list -> CSR -> handle without the entries leaving HBM in between (graphstruc's generate_adjacency
 * [+ add_self_loops] followed by set_graph).  adj_ia_out (n_vertices + 1, host) is always filled; adj_ja_out
 * (2 x capacity, host) only when non-null -- a caller that keeps the graph_type needs it, the layers do not. */
int athena_mp_graph_create_from_edges(int32_t n_vertices, int64_t n_pairs, const int32_t *index_list,
                                      int32_t add_self_loops, int32_t with_edge_ids, int32_t *adj_ia_out,
                                      int32_t *adj_ja_out, int64_t capacity, int64_t *nnz_out, athena_mp_graph **out)
{
    AMP_REQUIRE(out != nullptr && nnz_out != nullptr && adj_ia_out != nullptr, "graph_create_from_edges: null output pointer");
    *out = nullptr;
    int32_t *ja_dev = nullptr;
    int rc = amp::csr_from_edges_core(n_vertices, n_pairs, index_list, add_self_loops, adj_ia_out, adj_ja_out, capacity,
                                      nnz_out, &ja_dev);
    if (rc) {
        if (ja_dev) (void)hipFree(ja_dev);
        return rc;
    }
    const int32_t n_edge_cols = with_edge_ids ? (int32_t)n_pairs : 0;
    if (!with_edge_ids && *nnz_out > 0) {   // a handle without edge features ignores the ids: zero them in place
        AMP_HIP(hipMemset2DAsync(ja_dev + 1, 2 * sizeof(int32_t), 0, sizeof(int32_t), (size_t)*nnz_out, stream()));
    }
    rc = graph_create_impl(n_vertices, n_vertices, *nnz_out, adj_ia_out, adj_ja_out && with_edge_ids ? adj_ja_out : nullptr,
                           ja_dev, n_edge_cols, nullptr, nullptr, out);
    (void)hipFree(ja_dev);
    return rc;
}

static int graph_create_body(int32_t n_rows, int32_t n_cols, int64_t nnz, const int32_t *adj_ia,
                             const int32_t *adj_ja, const int32_t *adj_ja_dev, int32_t n_edge_cols,
                             const int32_t *row_deg, const int32_t *col_deg, athena_mp_graph **out);

static int64_t g_graph_builds = 0;

static int graph_create_impl(int32_t n_rows, int32_t n_cols, int64_t nnz, const int32_t *adj_ia,
                             const int32_t *adj_ja, const int32_t *adj_ja_dev, int32_t n_edge_cols,
                             const int32_t *row_deg, const int32_t *col_deg, athena_mp_graph **out)
{
    int rc = graph_create_body(n_rows, n_cols, nnz, adj_ia, adj_ja, adj_ja_dev, n_edge_cols, row_deg, col_deg, out);
    if (rc == 0) ++g_graph_builds;
    return rc;
}

/* ---- content key of a CSR (what a cached handle is valid for) ------------------------------------------------- */
namespace {
struct KeyHash {   // MurmurHash3's 64-bit lane: multiply-rotate per 8-byte word, fmix64 at the end
    uint64_t h;
    explicit KeyHash(uint64_t seed) : h(seed) {}
    inline void word(uint64_t k)
    {
        k *= 0x87c37b91114253d5ull;
        k = (k << 31) | (k >> 33);
        k *= 0x4cf5ad432745937full;
        h ^= k;
        h = ((h << 27) | (h >> 37)) * 5 + 0x52dce729;
    }
    void ints(const int32_t *p, int64_t count)   // count int32 values (pairs packed into one word)
    {
        int64_t i = 0;
        for (; i + 1 < count; i += 2) {
            uint64_t k;
            memcpy(&k, p + i, 8);
            word(k);
        }
        if (i < count) word((uint64_t)(uint32_t)p[i] | 0xa5a5a5a500000000ull);
    }
    uint64_t digest() const
    {
        uint64_t k = h;
        k ^= k >> 33;
        k *= 0xff51afd7ed558ccdull;
        k ^= k >> 33;
        k *= 0xc4ceb9fe1a85ec53ull;
        k ^= k >> 33;
        return k;
    }
};
}   // namespace

/* Chunked, lane-parallel form of the hash for large arrays: the array is cut into fixed chunks of kKeyChunk int32
 * values, every chunk is hashed by four interleaved lanes (four independent multiply chains per core instead of one),
 * chunks are handed to up to 8 host threads (fewer when the process's affinity mask holds fewer cores) and the chunk digests are folded in
 * chunk order -- the key does not depend on the thread count.  EVERY word of adj_ia / adj_ja enters the key. */
namespace {
constexpr int64_t kKeyChunk = (int64_t)1 << 20;
uint64_t key_chunk(const int32_t *p, int64_t count, uint64_t seed)
{
    KeyHash l0(seed), l1(seed ^ 0x243f6a8885a308d3ull), l2(seed ^ 0x13198a2e03707344ull), l3(seed ^ 0xa4093822299f31d0ull);
    int64_t i = 0;
    for (; i + 8 <= count; i += 8) {
        uint64_t k[4];
        memcpy(k, p + i, 32);
        l0.word(k[0]);
        l1.word(k[1]);
        l2.word(k[2]);
        l3.word(k[3]);
    }
    l0.ints(p + i, count - i);
    l0.word(l1.digest());
    l0.word(l2.digest());
    l0.word(l3.digest());
    return l0.digest();
}
void key_array(KeyHash &h, const int32_t *p, int64_t count)
{
    if (count <= 0) return;
    const int64_t n_chunks = (count + kKeyChunk - 1) / kKeyChunk;
    std::vector<uint64_t> dig((size_t)n_chunks);
    static const int max_threads = [] {
        int t = 8;
        const int hw = (int)std::thread::hardware_concurrency();   // the cores this process may run on (honours its affinity mask)
        if (hw > 0) t = std::min(t, hw);
        return std::max(1, t);
    }();
    const int n_thr = (int)std::min<int64_t>(max_threads, n_chunks);
    auto work = [&](int t) {
        for (int64_t c = t; c < n_chunks; c += n_thr) {
            const int64_t b = c * kKeyChunk;
            dig[(size_t)c] = key_chunk(p + b, std::min(kKeyChunk, count - b), 0x9e3779b97f4a7c15ull + (uint64_t)c);
        }
    };
    if (n_thr == 1) {
        work(0);
    } else {
        // a thread that cannot be started (process limits) must not become an exception across the C ABI: the chunks it
        // would have taken are hashed here instead -- same digests, same key
        std::vector<std::thread> pool;
        std::vector<int> orphan;
        for (int t = 1; t < n_thr; ++t) {
            try {
                pool.emplace_back(work, t);
            } catch (...) {
                orphan.push_back(t);
            }
        }
        work(0);
        for (int t : orphan) work(t);
        for (auto &th : pool) th.join();
    }
    for (uint64_t d : dig) h.word(d);
}
}   // namespace

extern "C++" {
namespace amp {
// 64-bit hash of EVERY byte of a host array (the chunked, threaded hash of the graph key): the "same content" test of
// the pair slots in host.hip
uint64_t content_hash(const void *p, size_t bytes)
{
    KeyHash h(0x51ed270b1f2e4a13ull);
    h.word((uint64_t)bytes);
    key_array(h, (const int32_t *)p, (int64_t)(bytes / 4));
    const unsigned char *t = (const unsigned char *)p + (bytes & ~(size_t)3);
    for (size_t i = 0; i < (bytes & 3); ++i) h.word((uint64_t)t[i] | 0x7700ull);
    return h.digest();
}
}   // namespace amp
}   // extern "C++"

int athena_mp_graph_key(int32_t n_rows, int64_t nnz, const int32_t *adj_ia, const int32_t *adj_ja, uint64_t *key)
{
    AMP_REQUIRE(key != nullptr, "graph_key: null key pointer");
    AMP_REQUIRE(n_rows >= 0 && nnz >= 0, "graph_key: negative size");
    AMP_REQUIRE(adj_ia != nullptr && (nnz == 0 || adj_ja != nullptr), "graph_key: null CSR arrays");
    // The reference re-copies the CSR on every set_graph (athena_msgpass_layer_sub.f90:144-174) and is always current:
    // the DEFAULT key therefore covers every word.  ATHENA_MP_GRAPH_KEY_SAMPLED=1 opts into the cheap sampled key for
    // callers that promise to invalidate_graph() after an in-place edit.
    static const int sampled = [] {
        const char *e = getenv("ATHENA_MP_GRAPH_KEY_SAMPLED");
        return (e && e[0] && e[0] != '0') ? 1 : 0;
    }();
    KeyHash h(0x9e3779b97f4a7c15ull);
    h.word((uint64_t)(uint32_t)n_rows);
    h.word((uint64_t)nnz);
    const int64_t ni = (int64_t)n_rows + 1;
    if (!sampled || nnz < ((int64_t)1 << 18)) {
        key_array(h, adj_ia, ni);
        key_array(h, adj_ja, 2 * nnz);
    } else {
        h.word(0x5a3b1edull);   // a sampled key never equals a full one
        const int64_t head = 1024;
        // row pointers: head, tail, every (ni / 4096)-th
        h.ints(adj_ia, std::min(head, ni));
        h.ints(adj_ia + std::max<int64_t>(0, ni - head), std::min(head, ni));
        const int64_t st_i = std::max<int64_t>(1, ni / 4096);
        for (int64_t i = 0; i < ni; i += st_i) {   // cold lines: keep 16 misses in flight
            if (i + 16 * st_i < ni) __builtin_prefetch(adj_ia + i + 16 * st_i);
            h.word((uint64_t)(uint32_t)adj_ia[i] ^ ((uint64_t)i << 32));
        }
        // entries (neighbour, edge id) as one 8-byte column each
        h.ints(adj_ja, 2 * head);
        h.ints(adj_ja + 2 * (nnz - head), 2 * head);
        const int64_t st_j = std::max<int64_t>(1, nnz / 4096);
        for (int64_t w = 0; w < nnz; w += st_j) {
            if (w + 16 * st_j < nnz) __builtin_prefetch(adj_ja + 2 * (w + 16 * st_j));
            uint64_t k;
            memcpy(&k, adj_ja + 2 * w, 8);
            h.word(k);
        }
    }
    *key = h.digest();
    return 0;
}

/* ---- handle cache: set_graph before every forward costs a key, not a build -------------------------------------- */
namespace {
struct CacheId {
    uint64_t key;
    int64_t nnz;
    int32_t n, n_edge_cols, device;
    bool operator<(const CacheId &o) const
    {
        if (key != o.key) return key < o.key;
        if (nnz != o.nnz) return nnz < o.nnz;
        if (n != o.n) return n < o.n;
        if (n_edge_cols != o.n_edge_cols) return n_edge_cols < o.n_edge_cols;
        return device < o.device;
    }
};
std::map<CacheId, athena_mp_graph *> g_cache;
std::mutex g_cache_mu;   // acquire / release / evict / clear may come from different host threads (one per layer is legal)
uint64_t g_cache_tick = 0;
int64_t g_cache_hits = 0;

int64_t cache_idle_cap()   // entries (nnz + n) the idle handles may hold together; 0 = free on last release
{
    static const int64_t cap = [] {
        const char *e = getenv("ATHENA_MP_GRAPH_CACHE_ENTRIES");
        return e ? (int64_t)atoll(e) : ((int64_t)1 << 28);
    }();
    return cap;
}

void cache_unlink(athena_mp_graph *g)   // the map forgets g; g stays alive for the users it still has
{
    for (auto it = g_cache.begin(); it != g_cache.end(); ++it)
        if (it->second == g) {
            g_cache.erase(it);
            break;
        }
    g->cache_linked = false;
}

void cache_trim()   // least recently used idle handles go first
{
    for (;;) {
        int64_t idle = 0, n_idle = 0;
        auto victim = g_cache.end();
        for (auto it = g_cache.begin(); it != g_cache.end(); ++it) {
            if (it->second->cache_refs != 0) continue;
            idle += it->second->nnz + it->second->n_rows;
            ++n_idle;
            if (victim == g_cache.end() || it->second->cache_tick < victim->second->cache_tick) victim = it;
        }
        if (victim == g_cache.end() || (idle <= cache_idle_cap() && n_idle <= 4096)) return;
        athena_mp_graph *g = victim->second;
        g_cache.erase(victim);
        g->cache_refs = -1;
        (void)graph_free(g);
    }
}
}   // namespace

extern "C++" {
/* athena_mp_finalize: idle handles are freed; a handle somebody still holds is only unlinked -- its owner's
 * athena_mp_graph_destroy / _release frees it (it used to be freed here, under its owner's feet). */
void amp::graph_cache_clear()
{
    std::lock_guard<std::mutex> lock(g_cache_mu);
    for (auto &kv : g_cache) {
        athena_mp_graph *g = kv.second;
        if (g->cache_refs > 0) {
            g->cache_linked = false;
            continue;
        }
        g->cache_refs = -1;
        (void)graph_free(g);
    }
    g_cache.clear();
}
}

int athena_mp_graph_acquire(int32_t n, int64_t nnz, const int32_t *adj_ia, const int32_t *adj_ja, int32_t n_edge_cols,
                            athena_mp_graph **out)
{
    AMP_REQUIRE(out != nullptr, "graph_acquire: null out pointer");
    *out = nullptr;
    CacheId id;
    int rc = athena_mp_graph_key(n, nnz, adj_ia, adj_ja, &id.key);
    if (rc) return rc;
    id.nnz = nnz;
    id.n = n;
    id.n_edge_cols = n_edge_cols;
    id.device = amp::device();
    std::lock_guard<std::mutex> lock(g_cache_mu);
    auto it = g_cache.find(id);
    if (it != g_cache.end()) {
        it->second->cache_refs++;
        it->second->cache_tick = ++g_cache_tick;
        ++g_cache_hits;
        *out = it->second;
        return 0;
    }
    athena_mp_graph *g = nullptr;
    rc = athena_mp_graph_create(n, n, nnz, adj_ia, adj_ja, n_edge_cols, nullptr, nullptr, &g);
    if (rc) return rc;
    g->cache_refs = 1;
    g->cache_linked = true;
    g->cache_key = id.key;
    g->cache_tick = ++g_cache_tick;
    g->cache_device = id.device;
    g_cache[id] = g;
    *out = g;
    return 0;
}

static int graph_release_locked(athena_mp_graph *g)
{
    AMP_REQUIRE(g->cache_refs > 0, "graph_release: handle is not an acquired one (refs %d)", g->cache_refs);
    g->cache_refs--;
    g->cache_tick = ++g_cache_tick;
    if (g->cache_refs == 0) {
        if (!g->cache_linked) {   // evicted, or the cache was cleared, while this user held it: the last user frees
            g->cache_refs = -1;
            return graph_free(g);
        }
        cache_trim();
    }
    return 0;
}

int athena_mp_graph_release(athena_mp_graph *g)
{
    if (!g) return 0;
    std::lock_guard<std::mutex> lock(g_cache_mu);
    return graph_release_locked(g);
}

/* invalidate_graph() / graph_type%touch(): the caller edited adj_ia / adj_ja in place and says so.  The handle leaves
 * the cache NOW -- the next acquire of whatever key the arrays hash to builds from the arrays -- and the caller's
 * reference is dropped; other holders keep a valid (old-topology) handle until they release it. */
int athena_mp_graph_evict(athena_mp_graph *g)
{
    if (!g) return 0;
    if (g->cache_refs < 0) return graph_free(g);   // never cached: plain destroy
    std::lock_guard<std::mutex> lock(g_cache_mu);
    if (g->cache_linked) cache_unlink(g);
    return graph_release_locked(g);
}

int athena_mp_graph_cache_stats(int64_t *handles, int64_t *hits, int64_t *builds)
{
    std::lock_guard<std::mutex> lock(g_cache_mu);
    if (handles) *handles = (int64_t)g_cache.size();
    if (hits) *hits = g_cache_hits;
    if (builds) *builds = g_graph_builds;
    return 0;
}

static int graph_create_body(int32_t n_rows, int32_t n_cols, int64_t nnz, const int32_t *adj_ia,
                             const int32_t *adj_ja, const int32_t *adj_ja_dev, int32_t n_edge_cols,
                             const int32_t *row_deg, const int32_t *col_deg, athena_mp_graph **out)
{
    *out = nullptr;
    g_err[0] = 0;   // the "allocation failed" fallbacks below only fill in when no callee left a message in THIS call
    AMP_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0 && n_edge_cols >= 0, "graph_create: negative size");
    AMP_REQUIRE(nnz < (int64_t)INT32_MAX, "graph_create: nnz %lld exceeds int32 CSR", (long long)nnz);
    AMP_REQUIRE(adj_ia != nullptr && (nnz == 0 || adj_ja != nullptr || adj_ja_dev != nullptr), "graph_create: null CSR arrays");
    AMP_REQUIRE((row_deg == nullptr) == (col_deg == nullptr),
                "graph_create: row_deg and col_deg must be given together");
    AMP_REQUIRE(row_deg != nullptr || n_rows == n_cols,
                "graph_create: a rectangular block (%d x %d) needs explicit degrees", n_rows, n_cols);
    AMP_REQUIRE(adj_ia[0] == 1, "graph_create: adj_ia(1) must be 1 (Fortran 1-based), got %d", adj_ia[0]);
    AMP_REQUIRE((int64_t)adj_ia[n_rows] - 1 == nnz, "graph_create: adj_ia(n+1)-1 = %d != nnz = %lld",
                adj_ia[n_rows] - 1, (long long)nnz);

    std::vector<int32_t> rowptr(n_rows + 1), col(nnz), eid(nnz), degr(n_rows), degc(n_cols);
    int32_t max_row = 0;
    for (int32_t v = 0; v < n_rows; ++v) {
        AMP_REQUIRE(adj_ia[v + 1] >= adj_ia[v], "graph_create: adj_ia not monotone at row %d", v + 1);
        rowptr[v] = adj_ia[v] - 1;
        int32_t len = adj_ia[v + 1] - adj_ia[v];
        max_row = len > max_row ? len : max_row;
        degr[v] = row_deg ? row_deg[v] : len;
    }
    rowptr[n_rows] = (int32_t)nnz;
    for (int32_t u = 0; u < n_cols; ++u) degc[u] = col_deg ? col_deg[u] : degr[u];
    int32_t band = INT32_MAX;
    if (adj_ja && n_rows == n_cols) {   // one pass over the entries on the host (stops at the first wide one)
        band = 0;
        for (int32_t v = 0; v < n_rows && band <= 64; ++v)
            for (int32_t w = rowptr[v]; w < rowptr[v + 1]; ++w) {
                const int32_t dlt = adj_ja[2 * (size_t)w] - 1 - v;
                band = std::max(band, dlt < 0 ? -dlt : dlt);
            }
        if (band > 64) band = INT32_MAX;
    }

    // Large graphs are built on the device (graph_build.hip): one upload of the caller's arrays, radix sorts
    // in HBM.  Small ones (mini-batches: ~0.1 ms on the host) and graphs with too many distinct degrees for
    // the coefficient table take the host builder below.  ATHENA_MP_GRAPH_BUILD=host|device pins the choice
    // (tests build both and compare every array).
    {
        const char *mode = getenv("ATHENA_MP_GRAPH_BUILD");
        const bool want_device = mode ? (strcmp(mode, "device") == 0) : (nnz >= ((int64_t)1 << 18));
        if (want_device && !(mode && strcmp(mode, "host") == 0)) {
            athena_mp_graph *g = new athena_mp_graph();
            g->n_rows = n_rows;
            g->n_cols = n_cols;
            g->nnz = nnz;
            g->n_edge_cols = n_edge_cols;
            g->max_row_len = max_row;
            g->band = band;
            std::vector<int32_t> t_rowptr_h;
            int rc = amp::graph_build_device(g, adj_ja, rowptr, degr, degc, &t_rowptr_h, adj_ja_dev);
            if (rc == 0) {
                int32_t max_col = 0;
                for (int32_t u = 0; u < n_cols; ++u) max_col = std::max(max_col, t_rowptr_h[u + 1] - t_rowptr_h[u]);
                g->max_col_len = max_col;
                g->h_deg_row = degr;
                rc = build_long_plan(rowptr, n_rows, &g->lp_fwd) | build_long_plan(t_rowptr_h, n_cols, &g->lp_bwd);
                if (rc == 0 && hipStreamSynchronize(stream()) == hipSuccess) {
                    *out = g;
                    return 0;
                }
                if (g_err[0] == 0) set_error("graph_create: device allocation failed");
                athena_mp_graph_destroy(g);
                return 1;
            }
            athena_mp_graph_destroy(g);
            if (rc > 0) {
                if (g_err[0] == 0) set_error("graph_create: device build failed");
                return rc;
            }
            // rc < 0: not applicable -> host builder
        }
    }

    // host builder: needs the entries on the host
    std::vector<int32_t> ja_copy;
    if (adj_ja == nullptr && nnz > 0) {
        ja_copy.resize(2 * (size_t)nnz);
        AMP_HIP(hipMemcpy(ja_copy.data(), adj_ja_dev, sizeof(int32_t) * ja_copy.size(), hipMemcpyDeviceToHost));
        adj_ja = ja_copy.data();
    }
    // transposed CSR + edge-column index by stable counting sort (entry order w preserved, so each
    // transposed row lists its sources in ascending v: the reference's accumulation order,
    // athena_diffstruc_extd_sub_kipf.f90:101-109).
    std::vector<int32_t> t_rowptr(n_cols + 1, 0), e_rowptr(n_edge_cols + 1, 0);
    int64_t n_with_edge = 0;
    for (int64_t w = 0; w < nnz; ++w) {
        int32_t u = adj_ja[2 * w] - 1, e = adj_ja[2 * w + 1] - 1;
        AMP_REQUIRE(u >= 0 && u < n_cols, "graph_create: adj_ja(1,%lld) = %d outside [1,%d]",
                    (long long)w + 1, u + 1, n_cols);
        AMP_REQUIRE(e >= -1 && e < n_edge_cols, "graph_create: adj_ja(2,%lld) = %d outside [0,%d]",
                    (long long)w + 1, e + 1, n_edge_cols);
        col[w] = u;
        eid[w] = e;
        t_rowptr[u + 1]++;
        if (e >= 0) {
            e_rowptr[e + 1]++;
            n_with_edge++;
        }
    }
    int32_t max_col = 0;
    for (int32_t u = 0; u < n_cols; ++u) {
        max_col = t_rowptr[u + 1] > max_col ? t_rowptr[u + 1] : max_col;
        t_rowptr[u + 1] += t_rowptr[u];
    }
    for (int32_t e = 0; e < n_edge_cols; ++e) e_rowptr[e + 1] += e_rowptr[e];
    std::vector<int32_t> t_src(nnz), t_eid(nnz), e_row(n_with_edge), e_col(n_with_edge);
    {
        std::vector<int32_t> tpos(t_rowptr.begin(), t_rowptr.end() - 1);
        std::vector<int32_t> epos(e_rowptr.begin(), e_rowptr.end() - 1);
        for (int32_t v = 0; v < n_rows; ++v)
            for (int32_t w = rowptr[v]; w < rowptr[v + 1]; ++w) {
                int32_t p = tpos[col[w]]++;
                t_src[p] = v;
                t_eid[p] = eid[w];
                if (eid[w] >= 0) {
                    int32_t q = epos[eid[w]]++;
                    e_row[q] = v;
                    e_col[q] = w; // entry index
                }
            }
    }

    athena_mp_graph *g = new athena_mp_graph();
    g->n_rows = n_rows;
    g->n_cols = n_cols;
    g->nnz = nnz;
    g->n_edge_cols = n_edge_cols;
    g->max_row_len = max_row;
    g->max_col_len = max_col;
    g->band = band;
    g->n_with_edge = n_with_edge;
    g->h_deg_row = degr;
    int rc = 0;
    rc |= upload(&g->rowptr, rowptr);
    rc |= upload(&g->col, col);
    rc |= upload(&g->eid, eid);
    rc |= upload(&g->t_rowptr, t_rowptr);
    rc |= upload(&g->t_src, t_src);
    rc |= upload(&g->t_eid, t_eid);
    rc |= upload(&g->e_rowptr, e_rowptr);
    rc |= upload(&g->e_row, e_row);
    rc |= upload(&g->e_col, e_col);
    rc |= upload(&g->deg_row, degr);
    rc |= upload(&g->deg_col, degc);
    std::vector<float> coef(nnz), t_coef(nnz);
    {
        CoefMemo memo;
        for (int32_t v = 0; v < n_rows; ++v)
            for (int32_t w = rowptr[v]; w < rowptr[v + 1]; ++w) coef[w] = memo.get(degr[v] * degc[col[w]]);
        for (int32_t u = 0; u < n_cols; ++u)
            for (int32_t w = t_rowptr[u]; w < t_rowptr[u + 1]; ++w) t_coef[w] = memo.get(degr[t_src[w]] * degc[u]);
    }
    rc |= upload(&g->coef, coef);
    rc |= upload(&g->t_coef, t_coef);
    rc |= build_long_plan(rowptr, n_rows, &g->lp_fwd);
    rc |= build_long_plan(t_rowptr, n_cols, &g->lp_bwd);
    if (rc != 0) {
        if (g_err[0] == 0) set_error("graph_create: device allocation failed");
        athena_mp_graph_destroy(g);
        return 1;
    }
    hipError_t e = hipStreamSynchronize(stream()); // host vectors die at scope exit
    if (e != hipSuccess) {
        set_error("graph_create: %s", hipGetErrorString(e));
        athena_mp_graph_destroy(g);
        return 1;
    }
    *out = g;
    return 0;
}

} // extern "C"
