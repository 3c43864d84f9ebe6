// Shared internals of libathena_mp (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <atomic>
#include <vector>

#include "../../include/athena_mp.h"

namespace amp {

// ---- error plumbing: C side never aborts, it returns a code + message ---------------------
void set_error(const char *fmt, ...);
hipStream_t stream();
// a second, library-owned stream of the current device and six events (pipelines inside one API call); swap_stream
// redirects the launchers of this library (they all launch on stream()) for the duration of such a pipeline
void swap_stream(hipStream_t s);
int aux_stream(hipStream_t *s, hipEvent_t **events);

#define AMP_HIP(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            amp::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                           __LINE__);                                                        \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

#define AMP_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            amp::set_error(__VA_ARGS__); \
            return 2;                   \
        }                               \
    } while (0)

#define AMP_LAUNCH_CHECK() AMP_HIP(hipGetLastError())

// workspace owned by the library (grown on demand, never shrunk; stream-ordered reuse); per device
int workspace(void **ptr, size_t bytes, int slot = 0);
// hand a slot's memory back to the device (after the stream drained); the next workspace() call of that slot allocates again
int workspace_release(int slot);

// Everything the library keeps on a device belongs to the device athena_mp_init selected: the workspaces above, the
// named buffers below (ticket ring, identity row pointer, zero bias ...) and the "attribute set" flags of kernels with
// more than 64 KB of LDS.  athena_mp_finalize releases the buffers of every device; a second athena_mp_init for
// another device gets buffers of its own.
constexpr int kMaxDevices = 32;
int device();        // the device athena_mp_init selected (0 before the first call)
int num_cus();       // its compute units (256 on MI355X)
// a device buffer of at least `bytes` under `name` on the current device.  *fresh is set when it was (re)allocated --
// its contents are then undefined (or zero with zero_fill) and the caller initialises them.
int named_buffer(const char *name, size_t bytes, bool zero_fill, void **ptr, bool *fresh = nullptr);
struct PerDeviceFlag {
    bool done[kMaxDevices] = {};
    bool &get() { return done[device()]; }
};

} // namespace amp

// Rows longer than kLongRow entries (hubs of heavy-tailed graphs) are cut into segments of kLongRow
// entries that independent lane groups sum in parallel; the segment sums are then added in segment
// order (deterministic).  Shorter rows keep the strictly sequential CSR-order sum.
constexpr int kLongRow = 512;
struct LongPlan {
    int32_t n_tasks = 0, n_long = 0;
    int32_t *task_beg = nullptr, *task_end = nullptr; // [n_tasks] entry ranges
    int32_t *row_id = nullptr;                        // [n_long] the long rows
    int32_t *row_task0 = nullptr;                     // [n_long+1] first task of each long row
};

// ---- graph handle: everything the kernels need, resident in HBM --------------------------------
inline uint64_t next_graph_serial()
{
    static std::atomic<uint64_t> n{0};
    return ++n;
}
struct athena_mp_graph {
    int32_t n_rows = 0, n_cols = 0, n_edge_cols = 0;
    int64_t nnz = 0;
    int32_t max_row_len = 0, max_col_len = 0;
    // max |column - row| over the entries of a square graph (INT32_MAX: not known): a batch of small graphs is block-diagonal, every
    // neighbour of a vertex sits within a few rows of it, and the gathers stage a block's rows in LDS instead of chasing them (agg.hip)
    int32_t band = INT32_MAX;
    // unique per handle for the life of the process (a freed handle's ADDRESS can come back; its serial never does)
    uint64_t serial = next_graph_serial();
    // handle cache (athena_mp_graph_acquire / _release): users of a cached handle, -1 = not cached
    int32_t cache_refs = -1;
    bool cache_linked = false;   // still findable by key (false after athena_mp_graph_evict / athena_mp_finalize)
    uint64_t cache_key = 0;
    uint64_t cache_tick = 0;
    int cache_device = 0;
    // forward CSR, 0-based
    int32_t *rowptr = nullptr;  // [n_rows+1]
    int32_t *col = nullptr;     // [nnz]  neighbour u
    int32_t *eid = nullptr;     // [nnz]  edge-feature column, -1 = none
    float *coef = nullptr;      // [nnz]  (deg_v*deg_u)^-1/2, the Kipf coefficient of each entry
    // transposed CSR (pull form of every scatter): row u lists the (v,w) with col[w]==u, v ascending
    int32_t *t_rowptr = nullptr; // [n_cols+1]
    int32_t *t_src = nullptr;    // [nnz] source row v
    int32_t *t_eid = nullptr;    // [nnz]
    float *t_coef = nullptr;     // [nnz]
    // edge-column index: column e lists the entries that carry it, in w order
    int32_t *e_rowptr = nullptr; // [n_edge_cols+1]
    int32_t *e_row = nullptr;    // [n_with_edge] row v of the entry
    int32_t *e_col = nullptr;    // [n_with_edge] CSR entry index w of the entry
    int32_t *deg_row = nullptr;  // [n_rows]
    int32_t *deg_col = nullptr;  // [n_cols]
    int64_t n_with_edge = 0;
    LongPlan lp_fwd, lp_bwd;        // hub rows of the forward / transposed CSR
    std::vector<int32_t> h_deg_row; // host copy (bucket planning)
    // Duvenaud degree buckets, built on first use for a (min_deg, max_deg) pair: vertices sorted by
    // bucket (stable), so each bucket is one contiguous run of a row-index array
    mutable int bucket_min = 0, bucket_max = -1;
    mutable int32_t *bucket_perm = nullptr;      // [n_rows] device
    mutable std::vector<int64_t> bucket_off;     // [n_buckets+1] host
    // rows ordered by length (longest first), built on first use by the fused GNO kernel: the 16 rows of a
    // tile then have (nearly) equal entry counts, so no wave waits at the tile barrier for a longer row
    mutable int32_t *len_perm_fwd = nullptr;     // [n_rows] device, forward CSR
    mutable int32_t n_long_fwd = 0;              // rows of the forward CSR with more than 32 entries (the first slots of len_perm_fwd)
    mutable int32_t n_mid_fwd = 0;               // ... with more than 16 entries (n_long_fwd included)
    mutable int32_t *len_perm_bwd = nullptr;     // [n_cols] device, transposed CSR
    // forward entry of every transposed entry (-1: no edge column), built on first use by athena_mp_gno_aggregate_bwd
    mutable int32_t *t_entry = nullptr;          // [nnz] device
    // the same runs cut into 16-vertex tiles (one MFMA column block each), bucket-major
    mutable int32_t n_btiles = 0;
    mutable int32_t *btile_start = nullptr;      // [n_btiles] device: first index into bucket_perm
    mutable int32_t *btile_rows = nullptr;       // [4][16*n_btiles] device: vertex of each tile slot (copies: see duvenaud_buckets); in copy 0 padding slots hold
                                                 //   ~(first vertex of the tile) (negative: load from it, never store)
    mutable int32_t *btile_info = nullptr;       // [n_btiles] device: bucket << 8 | vertices in the tile (1..16)
    mutable int32_t *btile_off_dev = nullptr;    // [n_buckets+1] device: first tile of each bucket
    mutable std::vector<int32_t> btile_off;      // [n_buckets+1] host
};

namespace amp {
// device-side construction of the handle arrays (graph_build.hip); -1 = use the host builder
int graph_build_device(athena_mp_graph *g, const int32_t *adj_ja, const std::vector<int32_t> &rowptr,
                       const std::vector<int32_t> &degr, const std::vector<int32_t> &degc,
                       std::vector<int32_t> *t_rowptr_host, const int32_t *adj_ja_dev = nullptr);
int csr_from_edges_core(int32_t n_vertices, int64_t n_pairs, const int32_t *index_list, int32_t add_self_loops,
                        int32_t *adj_ia_out, int32_t *adj_ja_out, int64_t capacity, int64_t *nnz_out,
                        int32_t **keep_ja_dev);
void gno_forget_perm(const int32_t *perm_dev);   // gno.hip: counts cached beside a row-length order
void graph_cache_clear(); // idle and live handles of athena_mp_graph_acquire (capi.hip)
void host_pool_release();   // staging buffers of the *_host entry points (host.hip)
uint64_t content_hash(const void *p, size_t bytes);   // every byte of a host array (capi.hip)
bool kipf_gather_is_banded(const athena_mp_graph *g, bool transposed, int F, const float *x, const float *y);   // agg.hip
// banded_fused.hip: the Kipf layer step on a banded graph in one launch at 64 -> 64 (0 done, -1 not its shape, > 0 error)
int banded_agg_gemm64(const athena_mp_graph *g, bool transposed, const float *coef, const float *x, const float *W, int b_nk,
                      const float *bias, int act, float *P, float *Z);
int agg_blocks_cap();
void set_agg_blocks_cap(int n);
// shared launchers (defined in agg.hip / gemm.hip)
int gather_agg(const int32_t *rowptr, const int32_t *idx, const float *coef, const float *x, int64_t ldx,
               float *y, int64_t ldy, int32_t n_rows, int32_t F, const LongPlan *lp = nullptr, int act = 0);
int gather_agg_dual(const int32_t *rowptr, const int32_t *idx, const float *coef, const float *x, float *y, float *y2,
                    int32_t n_rows, int32_t F, const LongPlan *lp);
// Z[M,N] = act(A[M,K] . B + bias); b_nk: B stored [N][K] instead of [K][N]
int gemm_dispatch(const float *A, const float *B, int b_nk, const float *bias, int act, float *Z, int64_t M,
                  int K, int N);
// dWt[Fi,Fo] (+)= sum_v P[v,:]^T dZ[v,:]
int gemm_dw_dispatch(int64_t N, int Fi, int Fo, const float *P, const float *dZ, float *dW, bool accumulate);
int slab_reduce(const float *slabs, int n_slabs, int n, float *out, bool accumulate);
// n_segs independent ranges [first[i], first[i]+count[i]) of the same slab array -> out + i*out_stride (count 0 => zeros)
int slab_reduce_segs(const float *slabs, int n, int n_segs, const int *first, const int *count, float *out,
                     int64_t out_stride, bool accumulate);

// reverse pass of a square Kipf step with dW folded in (fused_dw.hip)
bool fused_dw_shape(int Fi, int Fo);
int fused_dw_dispatch(const int32_t *t_rowptr, const int32_t *t_src, const float *t_coef, const float *dZ, const float *W,
                      const float *X, int exact, float *dX, float *dW, int64_t n_cols);

// shape-generic tiled MFMA contraction (gemm_tiled.hip)
struct TiledArgs {
    const float *A = nullptr; int64_t lda = 0; const int32_t *a_idx = nullptr; float a_div = 1.0f;
    const float *B = nullptr; int64_t ldb = 0; int b_nk = 0;
    const float *bias = nullptr; int act = 0; float c_div = 1.0f;
    float *C = nullptr; int64_t ldc = 0; const int32_t *c_idx = nullptr;
    int64_t M = 0; int N = 0; int K = 0;
};
int gemm_tiled(const TiledArgs &p);
int gemm_atb_tiled(const float *A, int64_t lda, const float *B, int64_t ldb, const int32_t *idx, float div, int64_t M,
                   int KI, int NO, float *C, bool accumulate);
// bucket-sorted vertex permutation of a graph for Duvenaud's degree buckets (duvenaud.hip)
int duvenaud_buckets(const athena_mp_graph *g, int min_deg, int max_deg);
// [a_x | a_e] -> a packed in the library's workspace (duvenaud.hip): split a for shapes outside the split kernels
int duv_pack_a(const athena_mp_graph *g, int32_t Fv, int32_t Fe, const float *a_x, const float *a_e, const float **packed);
// register-resident-weight MFMA kernels of the bucketed update (duv_mfma.hip); return -1 when the shape
// is outside what they cover (caller falls back to the tiled route)
int duv_mfma_fwd(const athena_mp_graph *g, int Fi, int Fo, const float *a, const float *w, int act, float *c);
int duv_mfma_bwd_a(const athena_mp_graph *g, int Fi, int Fo, const float *grad, const float *w, float *da);
int duv_mfma_bwd_w(const athena_mp_graph *g, int Fi, int Fo, const float *grad, const float *a, float *dw);
int duv_mfma_fwd_readout(const athena_mp_graph *g, int Fi, int Fo, const float *a, const float *w, int act, float *z,
                         const float *R, int O, float *p, const float *a_tail = nullptr);
int duv_mfma_bwd(const athena_mp_graph *g, int Fi, int Fo, const float *grad, const float *a, const float *w, float *da, float *dw,
                 float *da_tail = nullptr);
// readout reverse + update reverse in one launch (duv_mfma.hip, duv_bwd_ro_kernel); -1: shape outside it
int duv_mfma_bwd_readout(const athena_mp_graph *g, int Fi, int Fo, int O, int act, const float *a, const float *w, const float *z,
                         const float *dz_next, const float *p, const int32_t *tgid, const float *gout, const float *R, float *da,
                         float *da_tail, float *dw, float *dr_slabs, int *n_slabs, bool accumulate_tail,
                         const float *a_tail = nullptr);
} // namespace amp
