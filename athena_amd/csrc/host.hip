// Host-pointer variants of the op-level entry points ("phase 1" of SURVEY.md 7.4): a Fortran caller
// that only holds array_type%val passes host arrays; the shim stages them through HBM around the same
// device entry point.  Plumbing, never timed -- device-resident callers use the *_dev entry points.
#include <string.h>

#include <initializer_list>
#include <map>

#include "common.h"

using namespace amp;

namespace {

struct Stage {
    const void *in;   // host source (nullptr: output only)
    void *out;        // host destination (nullptr: input only)
    size_t bytes;
    bool through = false;   // output that is ALWAYS copied home, resident mode or not: the partials of the reverse callbacks, which
                            // diffstruc's grad_reverse accumulates with host arithmetic the moment the callback returns
    void *dev = nullptr;
};

// Staging buffers are pooled per argument position (grown on demand, kept until athena_mp_finalize): a
// hipMalloc / hipFree pair per argument costs more than the whole kernel for mini-batch sized inputs.
constexpr int kMaxStaged = 12;
void *g_pool[kMaxStaged] = {};
size_t g_pool_bytes[kMaxStaged] = {};

int pool_get(int slot, size_t bytes, void **out)
{
    if (bytes == 0) bytes = 4;
    if (g_pool_bytes[slot] < bytes) {
        if (g_pool[slot]) {
            AMP_HIP(hipStreamSynchronize(stream()));
            AMP_HIP(hipFree(g_pool[slot]));
            g_pool[slot] = nullptr;
            g_pool_bytes[slot] = 0;
        }
        const size_t want = bytes + bytes / 4;   // headroom so a slowly growing batch does not reallocate every call
        AMP_HIP(hipMalloc(&g_pool[slot], want));
        g_pool_bytes[slot] = want;
    }
    *out = g_pool[slot];
    return 0;
}

// ---- residency table (SURVEY.md 7.4 phase 2 for callers that only ever hold host arrays) ---------------------------------
// Opt-in (athena_mp_resident_mode(1)): the result of a *_host call stays in HBM, registered under the host array's
// address and byte length, and is NOT copied back; a later *_host call that is handed the same host array as an input
// takes the device copy.  The host array is materialised by athena_mp_resident_flush at the edge of the HIP island
// (athena's forward_generic2d reads layer%output, athena_network_sub.f90:2752,2761; the optimiser reads the gradients,
// :2856).  While a result lives only on the device its host array carries a 16-byte sentinel at both ends: if host code
// (or a re-allocated array at the same address) writes there, the next use sees the sentinel gone and uploads the host
// content instead of trusting the device copy; a flush (explicit, forced by an overlapping argument, or on leaving
// resident mode) sees it gone and leaves the host array alone.  The sentinel guards the two ENDS of the array: host code
// that rewrites only interior elements of a parked result is outside the contract (include/athena_mp.h, residency).  An argument that only OVERLAPS a registered array (a slice) makes that
// array materialise first.
struct Resident {
    void *dev = nullptr;
    size_t bytes = 0;
    bool dev_newer = false;   // the device copy is the only valid one (host holds the sentinel)
    bool trusted = false;     // caller promised (acquire with dirty_host = 0) that host code does not change the array
    uint64_t id = 0;
    uint64_t version = 0;     // bumped whenever the device copy's content changes (upload, result of a call, release dirty)
};
std::map<const char *, Resident> g_res;
bool g_res_on = false;
uint64_t g_res_next_id = 1;
uint64_t g_res_version = 1;
struct ResidentStats {
    int64_t h2d_bytes = 0, d2h_bytes = 0, reused_inputs = 0, lazy_outputs = 0;
    int64_t stale_dropped = 0;   // flushes skipped because the host array had been overwritten (sentinel gone)
} g_res_stats;

constexpr size_t kResidentMin = 64;   // smaller arguments are staged as before
void sentinel_words(const Resident &r, uint32_t w[4])
{
    w[0] = 0x7fc0a7e1u;   // quiet NaNs with a payload no computation produces
    w[1] = 0x7fc0a7e2u;
    w[2] = (uint32_t)r.id;
    w[3] = (uint32_t)(r.id >> 32) ^ 0x5a5a5a5au;
}
void sentinel_write(const Resident &r, void *host)
{
    uint32_t w[4];
    sentinel_words(r, w);
    memcpy(host, w, 16);
    memcpy((char *)host + r.bytes - 16, w, 16);
}
bool sentinel_intact(const Resident &r, const void *host)
{
    uint32_t w[4];
    sentinel_words(r, w);
    return memcmp(host, w, 16) == 0 && memcmp((const char *)host + r.bytes - 16, w, 16) == 0;
}
int resident_flush_one(const char *host, Resident &r)
{
    if (r.dev_newer && !sentinel_intact(r, host)) {
        // host code wrote the array (or the address now belongs to another allocation) after the result was parked on the
        // device: the HOST content is the newer one.  Copying r.bytes down would overwrite it -- and, for a re-allocated
        // smaller array, memory beyond it.  The device copy is stale: forget that it was ever newer.
        r.dev_newer = false;
        r.trusted = false;
        ++g_res_stats.stale_dropped;
        return 0;
    }
    if (r.dev_newer) {
        AMP_HIP(hipMemcpyAsync((void *)host, r.dev, r.bytes, hipMemcpyDeviceToHost, stream()));
        AMP_HIP(hipStreamSynchronize(stream()));
        g_res_stats.d2h_bytes += (int64_t)r.bytes;
        r.dev_newer = false;
    }
    return 0;
}
void resident_free(Resident &r)
{
    if (r.dev) (void)hipFree(r.dev);
    r.dev = nullptr;
}
// every registered array that overlaps [p, p + bytes) other than an exact match: materialise and forget it
int resident_evict_overlaps(const char *p, size_t bytes)
{
    for (auto it = g_res.begin(); it != g_res.end();) {
        const char *b = it->first, *e = b + it->second.bytes;
        const bool exact = b == p && it->second.bytes == bytes;
        if (!exact && b < p + bytes && p < e) {
            if (resident_flush_one(b, it->second)) return 1;
            resident_free(it->second);
            it = g_res.erase(it);
        } else {
            ++it;
        }
    }
    return 0;
}
// device pointer for a host array; *uploaded says whether host content went up now
int resident_get(const void *host, size_t bytes, bool as_input, void **dev)
{
    const char *p = (const char *)host;
    if (resident_evict_overlaps(p, bytes)) return 1;
    auto it = g_res.find(p);
    if (it == g_res.end()) {
        Resident r;
        r.bytes = bytes;
        r.id = g_res_next_id++;
        AMP_HIP(hipMalloc(&r.dev, bytes));
        it = g_res.emplace(p, r).first;
    } else if (as_input) {
        Resident &r = it->second;
        const bool device_valid = (r.dev_newer && sentinel_intact(r, host)) || (!r.dev_newer && r.trusted);
        if (device_valid) {
            ++g_res_stats.reused_inputs;
            *dev = r.dev;
            return 0;
        }
    }
    Resident &r = it->second;
    if (as_input) {   // first touch, or host code wrote the array since: the host content is the valid one
        AMP_HIP(hipMemcpyAsync(r.dev, host, bytes, hipMemcpyHostToDevice, stream()));
        g_res_stats.h2d_bytes += (int64_t)bytes;
        r.dev_newer = false;
        r.version = ++g_res_version;
    }
    *dev = r.dev;
    return 0;
}

// takes a device buffer per argument (pooled staging, or the residency table in resident mode), uploads inputs, runs
// body(dev pointers), downloads outputs (resident mode: leaves them in HBM behind a sentinel)
template <typename Body> int staged(std::initializer_list<Stage> args, Body &&body)
{
    std::vector<Stage> a(args);
    if ((int)a.size() > kMaxStaged) {
        set_error("host staging: too many arguments");
        return 1;
    }
    int rc = 0, slot = 0;
    std::vector<char> resident(a.size(), 0);
    for (size_t k = 0; k < a.size() && rc == 0; ++k) {
        Stage &s = a[k];
        const void *host = s.in ? s.in : s.out;
        if (g_res_on && host && s.bytes >= kResidentMin) {
            resident[k] = 1;
            ++slot;
            if (resident_get(host, s.bytes, s.in != nullptr, &s.dev)) rc = 1;
            continue;
        }
        if (pool_get(slot++, s.bytes, &s.dev)) {
            if (rc == 0) set_error("host staging: device allocation of %zu bytes failed", s.bytes);
            rc = 1;
            break;
        }
        if (s.in && s.bytes && hipMemcpyAsync(s.dev, s.in, s.bytes, hipMemcpyHostToDevice, stream()) != hipSuccess) {
            set_error("host staging: upload failed");
            rc = 1;
            break;
        }
    }
    if (rc == 0) {
        std::vector<void *> d;
        for (auto &s : a) d.push_back(s.dev);
        rc = body(d);
    }
    if (rc == 0) {
        for (size_t k = 0; k < a.size(); ++k) {
            Stage &s = a[k];
            if (!s.out || !s.bytes) continue;
            if (resident[k] && s.through) {
                // a reverse partial goes home at once: the tape ACCUMULATES into the host array it lands in (grad%val +=
                // partial), so the device copy cannot be trusted beyond this call -- dev_newer / trusted off: the next call
                // that names this array uploads it again.  The residency table saves nothing for reverse outputs; it is the
                // forward values (read several times by the reverse pass) that stay in HBM.
                Resident &r = g_res[(const char *)s.out];
                if (hipMemcpyAsync(s.out, s.dev, s.bytes, hipMemcpyDeviceToHost, stream()) != hipSuccess) {
                    set_error("host staging: download failed");
                    rc = 1;
                }
                r.dev_newer = false;
                r.trusted = false;
                r.version = ++g_res_version;
                g_res_stats.d2h_bytes += (int64_t)s.bytes;
            } else if (resident[k]) {   // stays in HBM; the host array is materialised by athena_mp_resident_flush
                Resident &r = g_res[(const char *)s.out];
                r.dev_newer = true;
                r.trusted = false;
                r.version = ++g_res_version;
                sentinel_write(r, s.out);
                ++g_res_stats.lazy_outputs;
            } else if (hipMemcpyAsync(s.out, s.dev, s.bytes, hipMemcpyDeviceToHost, stream()) != hipSuccess) {
                set_error("host staging: download failed");
                rc = 1;
            }
        }
        if (hipStreamSynchronize(stream()) != hipSuccess && rc == 0) {
            set_error("host staging: synchronize failed");
            rc = 1;
        }
    } else {
        (void)hipStreamSynchronize(stream());
    }
    return rc;
}

inline size_t fb(int64_t rows, int64_t cols) { return sizeof(float) * (size_t)rows * (size_t)cols; }

// ---- pair slots: both partials of a two-operand node from ONE device pass, across two stateless callbacks ----------------
// diffstruc's grad_reverse asks a node for its partials one at a time, each through a `pure` callback with an intent(in)
// node: the callback can keep nothing.  The fused reverse kernels produce both partials from one pass over the upstream
// gradient (athena_mp_duvenaud_update_bwd: 1.95 instead of 2.55 GB at BASELINE configs[2]; athena_mp_gno_aggregate_bwd: one
// G = g Vmat^T for dx and dtheta).  A *_pair_host entry point therefore computes BOTH on the first request, hands over the one
// asked for and parks the other on the device; the second request takes the parked one if -- and only if -- it names the
// same graph handle, the same shapes and operand arrays with the same CONTENT.  "Same content" is a stamp per operand: for an
// array that lives in the residency table behind an intact sentinel, the table entry's (id, version) -- the version moves
// whenever the device copy is rewritten; for any other array a 64-bit hash of every byte (amp::content_hash, the graph key's
// threaded hash).  A parked partial is handed over once, then the slot is empty; a request that matches no slot recomputes
// and parks.  kPairSlots slots per op kind, least recently parked replaced first: a tape that asks a node for its left partial,
// walks the WHOLE left subtree (other two-partial nodes included: every earlier time step of a Duvenaud chain) and only then
// comes back for the right partial finds it still parked, as does one that asks for both back to back -- the traversal order
// of diffstruc's grad_reverse is not something this library may assume.  What is parked in the usual order (left partial
// asked first) is the small one (dW, dtheta); a byte budget bounds the table when it is the other way round.
inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
uint64_t operand_stamp(const void *host, size_t bytes)
{
    if (g_res_on && bytes >= kResidentMin) {
        auto it = g_res.find((const char *)host);
        if (it != g_res.end() && it->second.bytes == bytes && it->second.dev_newer && sentinel_intact(it->second, host))
            // (id, version) of the table entry, both full 64-bit counters, folded into 63 bits by a mixing hash: neither can
            // overflow into the other however long the process runs
            return 0x8000000000000000ull | (mix64(mix64(it->second.id) ^ it->second.version) >> 1);
    }
    return content_hash(host, bytes) & 0x7fffffffffffffffull;
}
struct PairSlot {
    const athena_mp_graph *g = nullptr;
    uint64_t serial = 0;                          // the handle's serial: its address may be reused, this is not
    int32_t dims[4] = {0, 0, 0, 0};
    uint64_t stamp[4] = {0, 0, 0, 0};
    void *dev[3] = {nullptr, nullptr, nullptr};   // parked products
    size_t cap[3] = {0, 0, 0};
    bool ready[3] = {false, false, false};        // computed for (g, dims, stamp) and not handed over yet
    uint64_t parked_at = 0;                       // g_pair_clock when it was last parked: the oldest slot is replaced first
    bool any_ready() const { return ready[0] || ready[1] || ready[2]; }
    bool matches(const athena_mp_graph *g_, const int32_t (&d)[4], const uint64_t (&st)[4]) const
    {
        return g == g_ && serial == g_->serial && memcmp(dims, d, sizeof(dims)) == 0 && memcmp(stamp, st, sizeof(stamp)) == 0;
    }
    int reserve(int k, size_t bytes);
    size_t held() const { return cap[0] + cap[1] + cap[2]; }
    void clear()
    {
        g = nullptr;
        ready[0] = ready[1] = ready[2] = false;
    }
    void release()
    {
        for (int k = 0; k < 3; ++k) {
            if (dev[k]) (void)hipFree(dev[k]);
            dev[k] = nullptr;
            cap[k] = 0;
        }
        clear();
    }
};
constexpr int kPairSlots = 16;
constexpr size_t kPairBudget = (size_t)4 << 30;   // bytes parked over both tables before idle slots give theirs back
struct PairTable {
    PairSlot slot[kPairSlots];
    // the slot that holds `which` for (g, dims, stamps), or nullptr
    PairSlot *find(const athena_mp_graph *g, const int32_t (&d)[4], const uint64_t (&st)[4], int which)
    {
        for (auto &s : slot)
            if (s.ready[which] && s.matches(g, d, st)) return &s;
        return nullptr;
    }
    // where the next fused pass parks: an empty slot if there is one, else the one parked longest ago
    PairSlot *victim()
    {
        PairSlot *best = nullptr;
        for (auto &s : slot) {
            if (!s.any_ready()) return &s;
            if (!best || s.parked_at < best->parked_at) best = &s;
        }
        return best;
    }
    void release()
    {
        for (auto &s : slot) s.release();
    }
    size_t held() const
    {
        size_t b = 0;
        for (auto &s : slot) b += s.held();
        return b;
    }
    // give back the buffers of slots that hold nothing ready (all but `keep`)
    void trim(const PairSlot *keep)
    {
        for (auto &s : slot)
            if (&s != keep && !s.any_ready() && s.held()) s.release();
    }
};
PairTable g_pair_duv, g_pair_gno;
uint64_t g_pair_clock = 0;
int64_t g_pair_fused = 0, g_pair_handed = 0;

int PairSlot::reserve(int k, size_t bytes)
{
    if (cap[k] >= bytes) return 0;
    if (dev[k]) {
        AMP_HIP(hipStreamSynchronize(stream()));
        AMP_HIP(hipFree(dev[k]));
        dev[k] = nullptr;
        cap[k] = 0;
    }
    if (g_pair_duv.held() + g_pair_gno.held() + bytes > kPairBudget) {
        AMP_HIP(hipStreamSynchronize(stream()));
        g_pair_duv.trim(this);
        g_pair_gno.trim(this);
    }
    AMP_HIP(hipMalloc(&dev[k], bytes));
    cap[k] = bytes;
    return 0;
}

} // namespace

namespace amp {
void host_pool_release()
{
    for (auto &kv : g_res) resident_free(kv.second);   // finalize: no flush, the process is letting go of the device
    g_res.clear();
    g_res_on = false;
    g_pair_duv.release();
    g_pair_gno.release();
    for (int i = 0; i < kMaxStaged; ++i) {
        if (g_pool[i]) (void)hipFree(g_pool[i]);
        g_pool[i] = nullptr;
        g_pool_bytes[i] = 0;
    }
}
} // namespace amp

extern "C" {

/* ---- residency of host arrays (phase 2 for the *_host entry points) ---------------------------------------------------- */
int athena_mp_resident_mode(int32_t on)
{
    if (!on) {   // leaving the island: every result goes home, every device copy is released
        int rc = 0;
        for (auto &kv : g_res) rc |= resident_flush_one(kv.first, kv.second);
        for (auto &kv : g_res) resident_free(kv.second);
        g_res.clear();
        g_res_on = false;
        return rc;
    }
    g_res_on = true;
    return 0;
}
int athena_mp_resident_acquire(const void *host_ptr, uint64_t bytes, int32_t dirty_host, void **dev_ptr)
{
    AMP_REQUIRE(host_ptr && dev_ptr && bytes >= kResidentMin, "resident_acquire: null pointer or fewer than %zu bytes", kResidentMin);
    *dev_ptr = nullptr;
    const char *p = (const char *)host_ptr;
    if (resident_evict_overlaps(p, (size_t)bytes)) return 1;
    auto it = g_res.find(p);
    if (it == g_res.end()) {
        Resident r;
        r.bytes = (size_t)bytes;
        r.id = g_res_next_id++;
        AMP_HIP(hipMalloc(&r.dev, (size_t)bytes));
        it = g_res.emplace(p, r).first;
    }
    Resident &r = it->second;
    if (dirty_host) {
        AMP_HIP(hipMemcpyAsync(r.dev, host_ptr, (size_t)bytes, hipMemcpyHostToDevice, stream()));
        g_res_stats.h2d_bytes += (int64_t)bytes;
        r.dev_newer = false;
        r.version = ++g_res_version;
    }
    r.trusted = true;   // until a release says the device copy changed, or a *_host call writes it
    *dev_ptr = r.dev;
    return 0;
}
int athena_mp_resident_release(const void *host_ptr, int32_t dirty_dev)
{
    auto it = g_res.find((const char *)host_ptr);
    AMP_REQUIRE(it != g_res.end(), "resident_release: %p is not a registered host array", host_ptr);
    if (dirty_dev) {
        it->second.dev_newer = true;
        it->second.trusted = false;
        it->second.version = ++g_res_version;
        sentinel_write(it->second, (void *)host_ptr);
    }
    return 0;
}
int athena_mp_resident_flush(const void *host_ptr)
{
    if (!host_ptr) {
        int rc = 0;
        for (auto &kv : g_res) rc |= resident_flush_one(kv.first, kv.second);
        return rc;
    }
    auto it = g_res.find((const char *)host_ptr);
    if (it == g_res.end()) return 0;   // never left the host, or already home
    return resident_flush_one(it->first, it->second);
}
int athena_mp_resident_drop(const void *host_ptr)
{
    if (!host_ptr) {
        for (auto &kv : g_res) resident_free(kv.second);
        g_res.clear();
        return 0;
    }
    auto it = g_res.find((const char *)host_ptr);
    if (it != g_res.end()) {
        resident_free(it->second);
        g_res.erase(it);
    }
    return 0;
}
int athena_mp_resident_stats(int64_t *arrays, int64_t *h2d_bytes, int64_t *d2h_bytes, int64_t *reused_inputs,
                             int64_t *lazy_outputs)
{
    if (arrays) *arrays = (int64_t)g_res.size();
    if (h2d_bytes) *h2d_bytes = g_res_stats.h2d_bytes;
    if (d2h_bytes) *d2h_bytes = g_res_stats.d2h_bytes;
    if (reused_inputs) *reused_inputs = g_res_stats.reused_inputs;
    if (lazy_outputs) *lazy_outputs = g_res_stats.lazy_outputs;
    return 0;
}

int athena_mp_kipf_propagate_fwd_host(const athena_mp_graph *g, int32_t F, const float *xh, float *yh)
{
    AMP_REQUIRE(g && xh && yh && F > 0, "kipf_propagate_fwd_host: bad arguments");
    return staged({{xh, nullptr, fb(g->n_cols, F)}, {nullptr, yh, fb(g->n_rows, F)}}, [&](std::vector<void *> &d) {
        return athena_mp_kipf_propagate_fwd(g, F, (const float *)d[0], (float *)d[1]);
    });
}
int athena_mp_kipf_propagate_bwd_host(const athena_mp_graph *g, int32_t F, const float *gh, float *dxh, int32_t exact)
{
    AMP_REQUIRE(g && gh && dxh && F > 0, "kipf_propagate_bwd_host: bad arguments");
    return staged({{gh, nullptr, fb(g->n_rows, F)}, {nullptr, dxh, fb(g->n_cols, F), true}}, [&](std::vector<void *> &d) {
        return athena_mp_kipf_propagate_bwd(g, F, (const float *)d[0], (float *)d[1], exact);
    });
}
int athena_mp_gemm_fwd_host(int64_t N, int32_t Fi, int32_t Fo, const float *Ph, const float *Wh, const float *bh, int32_t act,
                            float *Zh)
{
    AMP_REQUIRE(N >= 0 && Fi > 0 && Fo > 0 && Ph && Wh && Zh, "gemm_fwd_host: bad arguments");
    return staged({{Ph, nullptr, fb(N, Fi)}, {Wh, nullptr, fb(Fi, Fo)}, {bh, nullptr, bh ? fb(Fo, 1) : 0}, {nullptr, Zh, fb(N, Fo)}},
                  [&](std::vector<void *> &d) {
                      return athena_mp_gemm_fwd(N, Fi, Fo, (const float *)d[0], (const float *)d[1], bh ? (const float *)d[2] : nullptr,
                                                act, (float *)d[3]);
                  });
}

int athena_mp_gemm_dw_host(int64_t N, int32_t Fi, int32_t Fo, const float *P, const float *dZ, float *dW)
{
    AMP_REQUIRE(N >= 0 && Fi > 0 && Fo > 0 && P && dZ && dW, "gemm_dw_host: bad arguments");
    return staged({{P, nullptr, fb(N, Fi)}, {dZ, nullptr, fb(N, Fo)}, {nullptr, dW, fb(Fi, Fo), true}}, [&](std::vector<void *> &d) {
        return athena_mp_gemm_dw(N, Fi, Fo, (float *)d[0], (float *)d[1], (float *)d[2]);
    });
}
int athena_mp_gemm_dx_host(int64_t N, int32_t Fi, int32_t Fo, const float *dZ, const float *W, float *dP)
{
    AMP_REQUIRE(N >= 0 && Fi > 0 && Fo > 0 && dZ && W && dP, "gemm_dx_host: bad arguments");
    return staged({{dZ, nullptr, fb(N, Fo)}, {W, nullptr, fb(Fi, Fo)}, {nullptr, dP, fb(N, Fi), true}}, [&](std::vector<void *> &d) {
        return athena_mp_gemm_dx(N, Fi, Fo, (float *)d[0], (float *)d[1], (float *)d[2]);
    });
}
int athena_mp_activation_fwd_host(int32_t act, int64_t n, const float *z, float *y)
{
    AMP_REQUIRE(n >= 0 && z && y, "activation_fwd_host: bad arguments");
    return staged({{z, nullptr, fb(n, 1)}, {nullptr, y, fb(n, 1)}}, [&](std::vector<void *> &d) {
        return athena_mp_activation_fwd(act, n, (float *)d[0], (float *)d[1]);
    });
}
int athena_mp_activation_bwd_host(int32_t act, int64_t n, const float *y, const float *g, float *dz)
{
    AMP_REQUIRE(n >= 0 && y && g && dz, "activation_bwd_host: bad arguments");
    return staged({{y, nullptr, fb(n, 1)}, {g, nullptr, fb(n, 1)}, {nullptr, dz, fb(n, 1), true}}, [&](std::vector<void *> &d) {
        return athena_mp_activation_bwd(act, n, (float *)d[0], (float *)d[1], (float *)d[2]);
    });
}

int athena_mp_duvenaud_propagate_fwd_host(const athena_mp_graph *g, int32_t Fv, int32_t Fe, const float *x,
                                          const float *e, float *c)
{
    AMP_REQUIRE(g && Fv > 0 && Fe >= 0 && x && c && (Fe == 0 || e), "duvenaud_propagate_fwd_host: bad arguments");
    return staged({{x, nullptr, fb(g->n_cols, Fv)}, {e, nullptr, fb(g->n_edge_cols, Fe)}, {nullptr, c, fb(g->n_rows, Fv + Fe)}},
                  [&](std::vector<void *> &d) {
                      return athena_mp_duvenaud_propagate_fwd(g, Fv, Fe, (float *)d[0], Fe ? (float *)d[1] : nullptr, (float *)d[2]);
                  });
}
int athena_mp_duvenaud_propagate_bwd_x_host(const athena_mp_graph *g, int32_t Fv, int32_t Fe, const float *grad, float *dx)
{
    AMP_REQUIRE(g && Fv > 0 && Fe >= 0 && grad && dx, "duvenaud_propagate_bwd_x_host: bad arguments");
    return staged({{grad, nullptr, fb(g->n_rows, Fv + Fe)}, {nullptr, dx, fb(g->n_cols, Fv), true}}, [&](std::vector<void *> &d) {
        return athena_mp_duvenaud_propagate_bwd_x(g, Fv, Fe, (float *)d[0], (float *)d[1]);
    });
}
int athena_mp_duvenaud_propagate_bwd_e_host(const athena_mp_graph *g, int32_t Fv, int32_t Fe, const float *grad, float *de)
{
    AMP_REQUIRE(g && Fv > 0 && Fe > 0 && grad && de, "duvenaud_propagate_bwd_e_host: bad arguments");
    return staged({{grad, nullptr, fb(g->n_rows, Fv + Fe)}, {nullptr, de, fb(g->n_edge_cols, Fe), true}}, [&](std::vector<void *> &d) {
        return athena_mp_duvenaud_propagate_bwd_e(g, Fv, Fe, (float *)d[0], (float *)d[1]);
    });
}
int athena_mp_duvenaud_update_fwd_host(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t mn, int32_t mx,
                                       const float *a, const float *w, float *c)
{
    AMP_REQUIRE(g && Fi > 0 && Fo > 0 && mx >= mn && a && w && c, "duvenaud_update_fwd_host: bad arguments");
    return staged({{a, nullptr, fb(g->n_rows, Fi)}, {w, nullptr, fb((int64_t)Fi * Fo, mx - mn + 1)}, {nullptr, c, fb(g->n_rows, Fo)}},
                  [&](std::vector<void *> &d) {
                      return athena_mp_duvenaud_update_fwd(g, Fi, Fo, mn, mx, (float *)d[0], (float *)d[1], (float *)d[2]);
                  });
}
int athena_mp_duvenaud_update_bwd_a_host(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t mn, int32_t mx,
                                         const float *grad, const float *w, float *da)
{
    AMP_REQUIRE(g && Fi > 0 && Fo > 0 && mx >= mn && grad && w && da, "duvenaud_update_bwd_a_host: bad arguments");
    return staged({{grad, nullptr, fb(g->n_rows, Fo)}, {w, nullptr, fb((int64_t)Fi * Fo, mx - mn + 1)}, {nullptr, da, fb(g->n_rows, Fi), true}},
                  [&](std::vector<void *> &d) {
                      return athena_mp_duvenaud_update_bwd_a(g, Fi, Fo, mn, mx, (float *)d[0], (float *)d[1], (float *)d[2]);
                  });
}
int athena_mp_duvenaud_update_bwd_w_host(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t mn, int32_t mx,
                                         const float *grad, const float *a, float *dw)
{
    AMP_REQUIRE(g && Fi > 0 && Fo > 0 && mx >= mn && grad && a && dw, "duvenaud_update_bwd_w_host: bad arguments");
    return staged({{grad, nullptr, fb(g->n_rows, Fo)}, {a, nullptr, fb(g->n_rows, Fi)}, {nullptr, dw, fb((int64_t)Fi * Fo, mx - mn + 1), true}},
                  [&](std::vector<void *> &d) {
                      return athena_mp_duvenaud_update_bwd_w(g, Fi, Fo, mn, mx, (float *)d[0], (float *)d[1], (float *)d[2]);
                  });
}
int athena_mp_softmax_segsum_fwd_host(int32_t O, int64_t N, int32_t S, const int32_t *seg, const float *logits,
                                      float *p, float *out, int32_t accumulate)
{
    AMP_REQUIRE(O > 0 && N >= 0 && S >= 0 && seg && logits && p && out, "softmax_segsum_fwd_host: bad arguments");
    return staged({{seg, nullptr, sizeof(int32_t) * (size_t)(S + 1)}, {logits, nullptr, fb(N, O)}, {nullptr, p, fb(N, O)},
                   {accumulate ? out : nullptr, out, fb(S, O)}},
                  [&](std::vector<void *> &d) {
                      return athena_mp_softmax_segsum_fwd(O, N, S, (int32_t *)d[0], (float *)d[1], (float *)d[2], (float *)d[3], accumulate);
                  });
}
int athena_mp_softmax_segsum_bwd_host(int32_t O, int64_t N, int32_t S, const int32_t *seg, const float *p,
                                      const float *gout, float *dlogits)
{
    AMP_REQUIRE(O > 0 && N >= 0 && S > 0 && seg && p && gout && dlogits, "softmax_segsum_bwd_host: bad arguments");
    return staged({{seg, nullptr, sizeof(int32_t) * (size_t)(S + 1)}, {p, nullptr, fb(N, O)}, {gout, nullptr, fb(S, O)},
                   {nullptr, dlogits, fb(N, O), true}},
                  [&](std::vector<void *> &d) {
                      return athena_mp_softmax_segsum_bwd(O, N, S, (int32_t *)d[0], (float *)d[1], (float *)d[2], (float *)d[3]);
                  });
}

static inline size_t gno_theta_floats(int d, int H, int Fi, int Fo) { return (size_t)H * d + H + (size_t)Fo * Fi * H + (size_t)Fo * Fi; }

int athena_mp_gno_aggregate_fwd_host(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                                     const float *theta, const float *coords, const float *x, float *m)
{
    AMP_REQUIRE(g && d > 0 && H > 0 && Fi > 0 && Fo > 0 && theta && coords && x && m, "gno_aggregate_fwd_host: bad arguments");
    return staged({{theta, nullptr, sizeof(float) * gno_theta_floats(d, H, Fi, Fo)}, {coords, nullptr, fb(g->n_edge_cols, d)},
                   {x, nullptr, fb(g->n_cols, Fi)}, {nullptr, m, fb(g->n_rows, Fo)}},
                  [&](std::vector<void *> &p) {
                      return athena_mp_gno_aggregate_fwd(g, d, H, Fi, Fo, (float *)p[0], (float *)p[1], (float *)p[2], (float *)p[3]);
                  });
}
int athena_mp_gno_aggregate_bwd_x_host(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                                       const float *theta, const float *coords, const float *grad, float *dx)
{
    AMP_REQUIRE(g && d > 0 && H > 0 && Fi > 0 && Fo > 0 && theta && coords && grad && dx, "gno_aggregate_bwd_x_host: bad arguments");
    return staged({{theta, nullptr, sizeof(float) * gno_theta_floats(d, H, Fi, Fo)}, {coords, nullptr, fb(g->n_edge_cols, d)},
                   {grad, nullptr, fb(g->n_rows, Fo)}, {nullptr, dx, fb(g->n_cols, Fi), true}},
                  [&](std::vector<void *> &p) {
                      return athena_mp_gno_aggregate_bwd_x(g, d, H, Fi, Fo, (float *)p[0], (float *)p[1], (float *)p[2], (float *)p[3]);
                  });
}
int athena_mp_gno_aggregate_bwd_theta_host(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                                           const float *theta, const float *coords, const float *x, const float *grad,
                                           float *dtheta)
{
    AMP_REQUIRE(g && d > 0 && H > 0 && Fi > 0 && Fo > 0 && theta && coords && x && grad && dtheta,
                "gno_aggregate_bwd_theta_host: bad arguments");
    const size_t tb = sizeof(float) * gno_theta_floats(d, H, Fi, Fo);
    return staged({{theta, nullptr, tb}, {coords, nullptr, fb(g->n_edge_cols, d)}, {x, nullptr, fb(g->n_cols, Fi)},
                   {grad, nullptr, fb(g->n_rows, Fo)}, {nullptr, dtheta, tb, true}},
                  [&](std::vector<void *> &p) {
                      return athena_mp_gno_aggregate_bwd_theta(g, d, H, Fi, Fo, (float *)p[0], (float *)p[1], (float *)p[2],
                                                               (float *)p[3], (float *)p[4]);
                  });
}
int athena_mp_gno_aggregate_bwd_coords_host(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                                            const float *theta, const float *coords, const float *x, const float *grad,
                                            float *dcoords)
{
    AMP_REQUIRE(g && d > 0 && H > 0 && Fi > 0 && Fo > 0 && theta && coords && x && grad && dcoords,
                "gno_aggregate_bwd_coords_host: bad arguments");
    return staged({{theta, nullptr, sizeof(float) * gno_theta_floats(d, H, Fi, Fo)}, {coords, nullptr, fb(g->n_edge_cols, d)},
                   {x, nullptr, fb(g->n_cols, Fi)}, {grad, nullptr, fb(g->n_rows, Fo)}, {nullptr, dcoords, fb(g->n_edge_cols, d), true}},
                  [&](std::vector<void *> &p) {
                      return athena_mp_gno_aggregate_bwd_coords(g, d, H, Fi, Fo, (float *)p[0], (float *)p[1], (float *)p[2],
                                                                (float *)p[3], (float *)p[4]);
                  });
}


// ---- composites and shaped activations ---------------------------------------------------------------
int athena_mp_duvenaud_update_act_fwd_host(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t mn, int32_t mx,
                                           const float *a, const float *w, int32_t act, float *z)
{
    AMP_REQUIRE(g && a && w && z && Fi > 0 && Fo > 0 && mx >= mn, "duvenaud_update_act_fwd_host: bad arguments");
    const int64_t N = g->n_rows;
    return staged({{a, nullptr, fb(N, Fi)}, {w, nullptr, fb((int64_t)Fi * Fo, mx - mn + 1)}, {nullptr, z, fb(N, Fo)}},
                  [&](std::vector<void *> &d) {
                      return athena_mp_duvenaud_update_act_fwd(g, Fi, Fo, mn, mx, (float *)d[0], (float *)d[1], act, (float *)d[2]);
                  });
}
int athena_mp_duvenaud_readout_fwd_host(int64_t N, int32_t Fv, int32_t O, int32_t S, const int32_t *seg, const float *z,
                                        const float *R, float *p, float *out, int32_t accumulate)
{
    AMP_REQUIRE(N >= 0 && Fv > 0 && O > 0 && S >= 0 && seg && z && R && p && out, "duvenaud_readout_fwd_host: bad arguments");
    return staged({{seg, nullptr, sizeof(int32_t) * (size_t)(S + 1)}, {z, nullptr, fb(N, Fv)}, {R, nullptr, fb(O, Fv)},
                   {nullptr, p, fb(N, O)}, {accumulate ? out : nullptr, out, fb(S, O)}},
                  [&](std::vector<void *> &d) {
                      return athena_mp_duvenaud_readout_fwd(N, Fv, O, S, (int32_t *)d[0], (float *)d[1], (float *)d[2],
                                                            (float *)d[3], (float *)d[4], accumulate);
                  });
}
int athena_mp_duvenaud_readout_bwd_host(int64_t N, int32_t Fv, int32_t O, int32_t S, const int32_t *seg, const float *z,
                                        const float *R, const float *p, const float *gout, const float *dz_next,
                                        int32_t act, float *dc, float *dR, int32_t accumulate)
{
    AMP_REQUIRE(N >= 0 && Fv > 0 && O > 0 && S > 0 && seg && z && R && p && gout && dc && dR,
                "duvenaud_readout_bwd_host: bad arguments");
    return staged({{seg, nullptr, sizeof(int32_t) * (size_t)(S + 1)}, {z, nullptr, fb(N, Fv)}, {R, nullptr, fb(O, Fv)},
                   {p, nullptr, fb(N, O)}, {gout, nullptr, fb(S, O)}, {dz_next, nullptr, dz_next ? fb(N, Fv) : 0},
                   {nullptr, dc, fb(N, Fv), true}, {accumulate ? dR : nullptr, dR, fb(O, Fv), true}},
                  [&](std::vector<void *> &d) {
                      return athena_mp_duvenaud_readout_bwd(N, Fv, O, S, (int32_t *)d[0], (float *)d[1], (float *)d[2],
                                                            (float *)d[3], (float *)d[4], dz_next ? (float *)d[5] : nullptr,
                                                            act, (float *)d[6], (float *)d[7], accumulate);
                  });
}
int athena_mp_softmax_fwd_host(int64_t N, int32_t F, const float *z, float *y)
{
    AMP_REQUIRE(N >= 0 && F > 0 && (N == 0 || (z && y)), "softmax_fwd_host: bad arguments");
    return staged({{z, nullptr, fb(N, F)}, {nullptr, y, fb(N, F)}},
                  [&](std::vector<void *> &d) { return athena_mp_softmax_fwd(N, F, (float *)d[0], (float *)d[1]); });
}
int athena_mp_softmax_bwd_host(int64_t N, int32_t F, const float *y, const float *g, float *dz)
{
    AMP_REQUIRE(N >= 0 && F > 0 && (N == 0 || (y && g && dz)), "softmax_bwd_host: bad arguments");
    return staged({{y, nullptr, fb(N, F)}, {g, nullptr, fb(N, F)}, {nullptr, dz, fb(N, F), true}}, [&](std::vector<void *> &d) {
        return athena_mp_softmax_bwd(N, F, (float *)d[0], (float *)d[1], (float *)d[2]);
    });
}
int athena_mp_swish_fwd_host(int64_t n, float beta, const float *x, float *y)
{
    AMP_REQUIRE(n >= 0 && (n == 0 || (x && y)), "swish_fwd_host: bad arguments");
    return staged({{x, nullptr, fb(n, 1)}, {nullptr, y, fb(n, 1)}},
                  [&](std::vector<void *> &d) { return athena_mp_swish_fwd(n, beta, (float *)d[0], (float *)d[1]); });
}
int athena_mp_swish_bwd_host(int64_t n, float beta, const float *x, const float *g, float *dx)
{
    AMP_REQUIRE(n >= 0 && (n == 0 || (x && g && dx)), "swish_bwd_host: bad arguments");
    return staged({{x, nullptr, fb(n, 1)}, {g, nullptr, fb(n, 1)}, {nullptr, dx, fb(n, 1), true}}, [&](std::vector<void *> &d) {
        return athena_mp_swish_bwd(n, beta, (float *)d[0], (float *)d[1], (float *)d[2]);
    });
}

int athena_mp_activation_param_fwd_host(int32_t kind, int64_t n, float scale, float p0, float p1, const float *x, float *y)
{
    AMP_REQUIRE(n >= 0 && (n == 0 || (x && y)), "activation_param_fwd_host: bad arguments");
    return staged({{x, nullptr, fb(n, 1)}, {nullptr, y, fb(n, 1)}}, [&](std::vector<void *> &d) {
        return athena_mp_activation_param_fwd(kind, n, scale, p0, p1, (float *)d[0], (float *)d[1]);
    });
}
int athena_mp_activation_param_bwd_host(int32_t kind, int64_t n, float scale, float p0, float p1, const float *x,
                                        const float *g, float *dx)
{
    AMP_REQUIRE(n >= 0 && (n == 0 || (x && g && dx)), "activation_param_bwd_host: bad arguments");
    return staged({{x, nullptr, fb(n, 1)}, {g, nullptr, fb(n, 1)}, {nullptr, dx, fb(n, 1), true}}, [&](std::vector<void *> &d) {
        return athena_mp_activation_param_bwd(kind, n, scale, p0, p1, (float *)d[0], (float *)d[1], (float *)d[2]);
    });
}

// update + activation + the readout's per-vertex softmax(R z) of the same time step (athena_mp_duvenaud_update_readout_fwd)
int athena_mp_duvenaud_update_readout_fwd_host(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t mn, int32_t mx,
                                               const float *a, const float *w, int32_t act, float *z, int32_t O, const float *R,
                                               float *p)
{
    AMP_REQUIRE(g && a && w && z && R && p && Fi > 0 && Fo > 0 && O > 0 && mx >= mn, "duvenaud_update_readout_fwd_host: bad arguments");
    const int64_t N = g->n_rows;
    return staged({{a, nullptr, fb(N, Fi)}, {w, nullptr, fb((int64_t)Fi * Fo, mx - mn + 1)}, {R, nullptr, fb(O, Fo)},
                   {nullptr, z, fb(N, Fo)}, {nullptr, p, fb(N, O)}},
                  [&](std::vector<void *> &d) {
                      return athena_mp_duvenaud_update_readout_fwd(g, Fi, Fo, mn, mx, (float *)d[0], (float *)d[1], act, (float *)d[3],
                                                                   O, (float *)d[2], (float *)d[4]);
                  });
}

// which = 0: da [n_rows, Fi] (get_partial_duvenaud_update_val, :284-324) | 1: dweight [Fo Fi D] (:326-368)
// act != ATHENA_MP_ACT_NONE: the node is update + message activation in one (its value z = act(duvenaud_update(a, w)), what
// athena_mp_duvenaud_update_act_fwd / _update_readout_fwd produce): grad is the gradient w.r.t. z and the activation's factor
// act'(z) is applied on the device first (relu / sigmoid / tanh differentiate at the OUTPUT); z_or_null = z [n_rows, Fo]
int athena_mp_duvenaud_update_bwd_pair_host(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t mn, int32_t mx, int32_t act,
                                            const float *z_or_null, const float *grad, const float *a, const float *w,
                                            int32_t which, float *out)
{
    AMP_REQUIRE(g && grad && a && w && out && Fi > 0 && Fo > 0 && mx >= mn && (which == 0 || which == 1),
                "duvenaud_update_bwd_pair_host: bad arguments");
    AMP_REQUIRE(act == ATHENA_MP_ACT_NONE || z_or_null, "duvenaud_update_bwd_pair_host: an activation needs the node's value z");
    const bool has_act = act != ATHENA_MP_ACT_NONE;
    const int64_t N = g->n_rows;
    const int32_t D = mx - mn + 1;
    const size_t bytes[2] = {fb(N, Fi), fb((int64_t)Fi * Fo, D)};
    const int32_t dims[4] = {Fi, Fo, mn | (act << 16), mx};
    const uint64_t st[4] = {operand_stamp(grad, fb(N, Fo)), operand_stamp(a, fb(N, Fi)), operand_stamp(w, bytes[1]),
                            has_act ? operand_stamp(z_or_null, fb(N, Fo)) : 0};
    if (PairSlot *hit = g_pair_duv.find(g, dims, st, which)) {
        hit->ready[which] = false;
        ++g_pair_handed;
        return staged({{nullptr, out, bytes[which], true}}, [&](std::vector<void *> &d) {
            AMP_HIP(hipMemcpyAsync(d[0], hit->dev[which], bytes[which], hipMemcpyDeviceToDevice, stream()));
            return 0;
        });
    }
    PairSlot &S = *g_pair_duv.victim();
    const int other = 1 - which;
    S.clear();
    if (S.reserve(other, bytes[other])) return 1;
    if (has_act && S.reserve(2, fb(N, Fo))) return 1;   // dc = act'(z) * grad
    const int rc = staged({{grad, nullptr, fb(N, Fo)}, {a, nullptr, fb(N, Fi)}, {w, nullptr, bytes[1]},
                           {has_act ? z_or_null : nullptr, nullptr, has_act ? fb(N, Fo) : 0}, {nullptr, out, bytes[which], true}},
                          [&](std::vector<void *> &d) {
                              const float *dc = (const float *)d[0];
                              if (has_act) {
                                  if (int e = athena_mp_activation_bwd(act, N * (int64_t)Fo, (const float *)d[3], (const float *)d[0], (float *)S.dev[2]))
                                      return e;
                                  dc = (const float *)S.dev[2];
                              }
                              float *da = which == 0 ? (float *)d[4] : (float *)S.dev[0];
                              float *dw = which == 1 ? (float *)d[4] : (float *)S.dev[1];
                              return athena_mp_duvenaud_update_bwd(g, Fi, Fo, mn, mx, dc, (float *)d[1], (float *)d[2], da, dw);
                          });
    if (rc) return rc;
    S.g = g;
    S.serial = g->serial;
    memcpy(S.dims, dims, sizeof(dims));
    memcpy(S.stamp, st, sizeof(st));
    S.ready[other] = true;
    S.parked_at = ++g_pair_clock;
    ++g_pair_fused;
    return 0;
}

// which = 0: dx [n_cols, Fi] (get_partial_gno_agg_features_val, athena_diffstruc_extd_sub_nop.f90:419-458)
//         1: dtheta (:480-526 composed with get_partial_gno_kernel_params_val :235-325)
//         2: dcoords [n_edge_cols, d] (:137-216) -- computed only when it is the one asked for
int athena_mp_gno_aggregate_bwd_pair_host(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                                          const float *theta, const float *coords, const float *x, const float *grad,
                                          int32_t which, float *out)
{
    AMP_REQUIRE(g && d > 0 && H > 0 && Fi > 0 && Fo > 0 && theta && coords && x && grad && out && which >= 0 && which <= 2,
                "gno_aggregate_bwd_pair_host: bad arguments");
    const size_t tb = sizeof(float) * gno_theta_floats(d, H, Fi, Fo);
    const size_t bytes[3] = {fb(g->n_cols, Fi), tb, fb(g->n_edge_cols, d)};
    const int32_t dims[4] = {d, H, Fi, Fo};
    const uint64_t st[4] = {operand_stamp(theta, tb), operand_stamp(coords, bytes[2]), operand_stamp(x, fb(g->n_cols, Fi)),
                            operand_stamp(grad, fb(g->n_rows, Fo))};
    if (PairSlot *hit = g_pair_gno.find(g, dims, st, which)) {
        hit->ready[which] = false;
        ++g_pair_handed;
        return staged({{nullptr, out, bytes[which], true}}, [&](std::vector<void *> &p) {
            AMP_HIP(hipMemcpyAsync(p[0], hit->dev[which], bytes[which], hipMemcpyDeviceToDevice, stream()));
            return 0;
        });
    }
    PairSlot &S = *g_pair_gno.victim();
    S.clear();
    for (int k = 0; k < 2; ++k)
        if (k != which && S.reserve(k, bytes[k])) return 1;
    const int rc = staged({{theta, nullptr, tb}, {coords, nullptr, bytes[2]}, {x, nullptr, fb(g->n_cols, Fi)},
                           {grad, nullptr, fb(g->n_rows, Fo)}, {nullptr, out, bytes[which], true}},
                          [&](std::vector<void *> &p) {
                              float *dx = which == 0 ? (float *)p[4] : (float *)S.dev[0];
                              float *dth = which == 1 ? (float *)p[4] : (float *)S.dev[1];
                              float *dc = which == 2 ? (float *)p[4] : nullptr;
                              return athena_mp_gno_aggregate_bwd(g, d, H, Fi, Fo, (float *)p[0], (float *)p[1], (float *)p[2], (float *)p[3],
                                                                 nullptr, dx, dth, dc, nullptr);
                          });
    if (rc) return rc;
    S.g = g;
    S.serial = g->serial;
    memcpy(S.dims, dims, sizeof(dims));
    memcpy(S.stamp, st, sizeof(st));
    S.ready[0] = which != 0;
    S.ready[1] = which != 1;
    S.parked_at = ++g_pair_clock;
    ++g_pair_fused;
    return 0;
}

int athena_mp_pair_stats(int64_t *fused_passes, int64_t *handed_over)
{
    if (fused_passes) *fused_passes = g_pair_fused;
    if (handed_over) *handed_over = g_pair_handed;
    return 0;
}

} // extern "C"
