// CSR gather-aggregate kernels: the HBM-bound heart of the path.
//
//   y[r, 0:F] = sum_{w in row r} coef[w] * x[idx[w], 0:F]          (idx[w] < 0 contributes 0)
//
// One kernel shape serves
//   kipf_propagate fwd            (idx = col,   coef = (deg_v deg_u)^-1/2)   ..._sub_kipf.f90:29-46
//   kipf bwd (pull form)          (idx = t_src, coef = none | t_coef)        ..._sub_kipf.f90:100-109
//   duvenaud_propagate fwd x / e  (idx = col / eid, no coef, strided y)      ..._sub_duvenaud.f90:34-42
//   duvenaud bwd x / e            (idx = t_src / e_row, strided x)           ..._sub_duvenaud.f90:135-141,164-170
//
// Mapping (CDNA4, 64-wide waves): a GROUP of G lanes owns one output row; each lane owns VEC
// contiguous features, so one group-wide load instruction reads a G*VEC*4-byte run of a vertex row
// (F=128: G=32, VEC=4 -> a full 512 B row per half-wave as 16 B/lane loads; 2 rows per wave).
// The group first loads G consecutive (idx, coef) entries of its row with one coalesced load,
// then walks them IN CSR ORDER, broadcasting each entry with a wave shuffle and keeping UNROLL
// independent row loads in flight.  Walking in CSR order reproduces the reference's per-feature
// summation order exactly; with -ffp-contract=off (rounded multiply, rounded add) the result is
// bit-identical to the strict-fp32 CPU restatement.
//
// Scatters of the reference (conflicting `do concurrent` writes) become pulls over the
// transposed CSR / edge index built at graph_create: no atomics, deterministic order.
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "common.h"

namespace amp {
// > 0: aggregation launches use at most this many (persistent, grid-striding) workgroups
static int g_agg_cap = -1;
int agg_blocks_cap()
{
    if (g_agg_cap < 0) g_agg_cap = 0;   // no cap (the env switch of rounds 1-4 is gone: set_agg_blocks_cap is the one way in)
    return g_agg_cap;
}
void set_agg_blocks_cap(int n) { g_agg_cap = n; }
} // namespace amp

namespace {

template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<4> { using type = float4; };

template <int VEC> __device__ __forceinline__ void vzero(float (&a)[VEC])
{
#pragma unroll
    for (int i = 0; i < VEC; ++i) a[i] = 0.0f;
}

template <int VEC>
__device__ __forceinline__ void vload(float (&dst)[VEC], const float *__restrict__ p)
{
    using V = typename VecT<VEC>::type;
    V t = *reinterpret_cast<const V *>(p);
    if constexpr (VEC == 1) dst[0] = t;
    if constexpr (VEC == 2) { dst[0] = t.x; dst[1] = t.y; }
    if constexpr (VEC == 4) { dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w; }
}

template <int VEC> __device__ __forceinline__ void vstore(float *__restrict__ p, const float (&src)[VEC])
{
    using V = typename VecT<VEC>::type;
    V t;
    if constexpr (VEC == 1) t = src[0];
    if constexpr (VEC == 2) { t.x = src[0]; t.y = src[1]; }
    if constexpr (VEC == 4) { t.x = src[0]; t.y = src[1]; t.z = src[2]; t.w = src[3]; }
    *reinterpret_cast<V *>(p) = t;
}

constexpr int kBlock = 256;
constexpr int kUnroll = 4;

__device__ __forceinline__ float agg_act(float t, int act)
{
    switch (act) {
    case ATHENA_MP_ACT_RELU: return t > 0.0f ? t : 0.0f;
    case ATHENA_MP_ACT_SIGMOID: return 1.0f / (1.0f + expf(-t));
    case ATHENA_MP_ACT_TANH: return tanhf(t);
    default: return t;
    }
}

// G lanes per row, VEC floats per lane, COEF: multiply by coef[w]; DUAL (with COEF): the rows are gathered once
// and summed twice -- y = plain sum, y2 = coefficient-weighted sum (same leading dimension)
template <int G, int VEC, bool COEF, bool DUAL = false>
__global__ __launch_bounds__(kBlock) void csr_gather_agg(
    const int32_t *__restrict__ rowbeg, const int32_t *__restrict__ rowend, const int32_t *__restrict__ idx,
    const float *__restrict__ coef, const float *__restrict__ x, int64_t ldx, float *__restrict__ y,
    int64_t ldy, int32_t n_rows, int32_t F, int32_t long_thr, float *__restrict__ y2, int32_t act)
{
    constexpr int kRowsPerBlock = kBlock / G;
    const int gl = threadIdx.x & (G - 1);          // lane inside the group
    const int n_row_blocks = (n_rows + kRowsPerBlock - 1) / kRowsPerBlock;
    // grid-stride over row blocks: gridDim.x == n_row_blocks for a plain launch, or a fixed number of
    // resident workgroups per CU when the launch leaves room for a co-running MFMA kernel
    for (int rb = blockIdx.x; rb < n_row_blocks; rb += gridDim.x) {
    const int row = rb * kRowsPerBlock + (threadIdx.x / G);
    bool live = row < n_rows;

    int start = 0, len = 0;
    if (live) {
        start = rowbeg[row];
        len = rowend[row] - start;
        if (long_thr > 0 && len > long_thr) { // hub row: summed by the segment launch instead
            len = 0;
            live = false;
        }
    }
    // longest row among the groups of this wave: keeps every shuffle wave-uniform
    int maxlen = len;
#pragma unroll
    for (int o = G; o < 64; o <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o));

    for (int f0 = 0; f0 < F; f0 += G * VEC) {
        const int f = f0 + gl * VEC;
        const bool fin = f < F;
        float acc[VEC], acc2[VEC];
        vzero<VEC>(acc);
        vzero<VEC>(acc2);
        for (int off = 0; off < maxlen; off += G) {
            // one coalesced load of up to G entries of the row
            int my_idx = -1;
            float my_c = 0.0f;
            if (off + gl < len) {
                my_idx = idx[start + off + gl];
                if constexpr (COEF) my_c = coef[start + off + gl];
            }
            const int cnt = min(G, len - off);       // entries of MY row in this chunk (may be <= 0)
            const int cntmax = min(G, maxlen - off); // wave-uniform
            for (int j = 0; j < cntmax; j += kUnroll) {
                int u[kUnroll];
                float c[kUnroll];
                float v[kUnroll][VEC];
#pragma unroll
                for (int k = 0; k < kUnroll; ++k) {
                    u[k] = __shfl(my_idx, j + k, G);
                    if constexpr (COEF) c[k] = __shfl(my_c, j + k, G);
                }
#pragma unroll
                for (int k = 0; k < kUnroll; ++k) {
                    vzero<VEC>(v[k]);
                    if (j + k < cnt && u[k] >= 0 && fin) vload<VEC>(v[k], x + (int64_t)u[k] * ldx + f);
                }
#pragma unroll
                for (int k = 0; k < kUnroll; ++k) {
                    if (j + k < cnt && u[k] >= 0) {
#pragma unroll
                        for (int i = 0; i < VEC; ++i) {
                            if constexpr (DUAL) {
                                acc[i] = acc[i] + v[k][i];
                                acc2[i] = acc2[i] + c[k] * v[k][i];
                            } else if constexpr (COEF)
                                acc[i] = acc[i] + c[k] * v[k][i];
                            else
                                acc[i] = acc[i] + v[k][i];
                        }
                    }
                }
            }
        }
        if (live && fin) {
            if (act != 0) {   // activation of the aggregated row (the dense step ran BEFORE the aggregation)
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc[i] = agg_act(acc[i], act);
            }
            vstore<VEC>(y + (int64_t)row * ldy + f, acc);
            if constexpr (DUAL) vstore<VEC>(y2 + (int64_t)row * ldy + f, acc2);
        }
    }
    } // row blocks
}


// ---------------------------------------------------------------------------------------------------------------------
// Short rows (molecule batches: 3-5 entries per vertex; BASELINE configs[2]).  The general kernel above spends such a
// row on its set-up -- wave-wide length reductions, index broadcasts by shuffle, a second launch for the edge part that
// writes 32 B slivers into 288 B rows -- while the neighbour rows themselves come from L2 (a block-diagonal batch keeps a
// vertex and its neighbours in the same few cache lines).  Here a group of G = F/4 lanes owns a row with no cross-lane
// traffic at all: every lane reads the row's bounds and entries itself (same address across the group: one request),
// keeps UN independent 16 B row loads in flight and adds them in CSR order (bit-identical to the general kernel); the
// first FE4 lanes of the group also gather the edge-feature part, so duvenaud_propagate is ONE launch that writes whole
// rows.  Latency is covered by occupancy (about 30 registers: 8 waves per SIMD).  Measured on configs[2] (2.34 M vertices,
// 7.14 M entries, F_v = 64, F_e = 8): 0.384 ms against 0.454 ms for the two general launches.  Two things measured and
// dropped: a software pipeline over a grid-stride walk (bounds of row k+2, entries of k+1, rows of k in flight together)
// took 0.42 ms -- the kernel is not waiting on its dependency chain -- and the same kernel on the reverse gather over the
// transposed CSR (no edge part to merge) took 0.32 ms against 0.31 ms for the general kernel, which stays.
// ---------------------------------------------------------------------------------------------------------------------
template <int G, int UN>
__global__ __launch_bounds__(256) void csr_gather_short_rows(const int32_t *__restrict__ rowptr,
                                                             const int32_t *__restrict__ idx,
                                                             const int32_t *__restrict__ eidx,
                                                             const float *__restrict__ x, int64_t ldx,
                                                             const float *__restrict__ e, int64_t lde, int FE4,
                                                             float *__restrict__ y, int64_t ldy, int32_t n_rows)
{
    typedef float v4 __attribute__((ext_vector_type(4)));
    const int gl = threadIdx.x & (G - 1);
    const int64_t row = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G;
    if (row >= n_rows) return;
    const int w0 = rowptr[row], w1 = rowptr[row + 1];
    const bool edge_lane = gl < FE4;
    v4 acc = {0.0f, 0.0f, 0.0f, 0.0f}, acce = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int w = w0; w < w1; w += UN) {
        int u[UN], ee[UN];
#pragma unroll
        for (int k = 0; k < UN; ++k) {
            u[k] = ee[k] = -1;
            if (w + k < w1) {
                u[k] = idx[w + k];
                if (FE4 > 0) ee[k] = eidx[w + k];
            }
        }
        v4 v[UN], ve[UN];
#pragma unroll
        for (int k = 0; k < UN; ++k) {
            v[k] = (v4){0.0f, 0.0f, 0.0f, 0.0f};
            ve[k] = (v4){0.0f, 0.0f, 0.0f, 0.0f};
            if (u[k] >= 0) v[k] = *reinterpret_cast<const v4 *>(x + (int64_t)u[k] * ldx + 4 * gl);
            if (edge_lane && ee[k] >= 0) ve[k] = *reinterpret_cast<const v4 *>(e + (int64_t)ee[k] * lde + 4 * gl);
        }
#pragma unroll
        for (int k = 0; k < UN; ++k) {
            if (u[k] >= 0) acc = acc + v[k];
            if (edge_lane && ee[k] >= 0) acce = acce + ve[k];
        }
    }
    *reinterpret_cast<v4 *>(y + row * ldy + 4 * gl) = acc;
    if (edge_lane) *reinterpret_cast<v4 *>(y + row * ldy + 4 * G + 4 * gl) = acce;
}

// ---------------------------------------------------------------------------------------------------------------------
// Banded graphs (a batch of small graphs is block-diagonal: every neighbour of row r lies within `band` rows of r; athena_mp_graph::band).
// The two kernels above chase rows: bounds -> entry ids -> rows, three dependent memory latencies per wave.  Here a workgroup owns RB
// consecutive rows and STREAMS what they can touch -- rows r0 - band .. r0 + RB + band of x, 16 bytes per lane, in launch order, no
// address depending on a load -- into LDS beside the block's bounds and entry ids (one dependent pair of loads per workgroup), then
// gathers from LDS: 16 lanes own a row and add its entries in CSR order (bit-identical to the kernels above).  Workgroup b takes
// chunk (b % 8) * per + b / 8: the workgroups of one XCD walk neighbouring chunks, so the 2 band rows two chunks share are L2 hits.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kBandRows = 128, kBandMax = 32, kBandEntries = 8;   // rows per workgroup, widest band, entries per row
// COEF: every entry carries a coefficient (Kipf: y[v] = sum_w coef[w] x[col w], a rounded multiply and a rounded add per entry in CSR
// order -- the general kernel's arithmetic, athena_diffstruc_extd_sub_kipf.f90:29-46); the coefficients of the block ride into LDS
// beside the entry ids.  Rows wider than 64 floats are gathered 64 columns at a time (x / y pointers moved, pitches kept).
template <bool COEF>
__global__ __launch_bounds__(256) void csr_gather_banded64(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ idx,
                                                           const float *__restrict__ coef, const float *__restrict__ x, int32_t xp4,
                                                           float *__restrict__ y, int32_t yp4, int32_t n_rows, int32_t band, int32_t per)
{
    // xp4 / yp4: pitch of x / y in 16-byte slices (16 = dense rows of 64 floats; 18 = the vertex part of packed [N, 72] rows, ...)
    typedef float v4 __attribute__((ext_vector_type(4)));
    constexpr int RB = kBandRows, XR = RB + 2 * kBandMax, NL = XR * 16 / 256;   // 12 row slices of 16 bytes per thread
    __shared__ __attribute__((aligned(16))) float xs[XR * 64];
    __shared__ int32_t rp[RB + 1], es[RB * kBandEntries];
    __shared__ float cs[COEF ? RB * kBandEntries : 1];
    const int chunk = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    const int r0 = chunk * RB;
    if (r0 >= n_rows) return;
    const int r1 = min(r0 + RB, n_rows), c0 = max(0, r0 - band), c1 = min(n_rows, r1 + band);
    const int nx = (c1 - c0) * 16;                 // 16-byte slices of x this workgroup stages
    v4 xv[NL];
    const v4 *x4 = reinterpret_cast<const v4 *>(x) + (int64_t)c0 * xp4;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int t = threadIdx.x + 256 * i;
        xv[i] = t < nx ? x4[(t >> 4) * xp4 + (t & 15)] : (v4){0.0f, 0.0f, 0.0f, 0.0f};
    }
    if ((int)threadIdx.x <= r1 - r0) rp[threadIdx.x] = rowptr[r0 + threadIdx.x];
    __syncthreads();
    const int w0 = rp[0], nw = rp[r1 - r0] - w0;
    int32_t ev[RB * kBandEntries / 256];
    [[maybe_unused]] float cv[RB * kBandEntries / 256];
#pragma unroll
    for (int i = 0; i < RB * kBandEntries / 256; ++i) {
        const int t = threadIdx.x + 256 * i;
        ev[i] = t < nw ? idx[w0 + t] : 0;
        if constexpr (COEF) cv[i] = t < nw ? coef[w0 + t] : 0.0f;
    }
    v4 *xs4 = reinterpret_cast<v4 *>(xs);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int t = threadIdx.x + 256 * i;
        if (t < nx) xs4[t] = xv[i];
    }
#pragma unroll
    for (int i = 0; i < RB * kBandEntries / 256; ++i) {
        const int t = threadIdx.x + 256 * i;
        if (t < nw) {
            es[t] = ev[i];
            if constexpr (COEF) cs[t] = cv[i];
        }
    }
    __syncthreads();
    const int gl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    v4 *y4 = reinterpret_cast<v4 *>(y);
#pragma unroll 2
    for (int pass = 0; pass < RB / 16; ++pass) {
        const int r = 16 * pass + rl;
        if (r0 + r >= r1) break;
        const int a = rp[r] - w0, b = rp[r + 1] - w0;
        v4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int w = a; w < b; ++w) {
            const v4 v = xs4[(es[w] - c0) * 16 + gl];
            if constexpr (COEF) acc = acc + cs[w] * v;
            else acc = acc + v;
        }
        y4[(int64_t)(r0 + r) * yp4 + gl] = acc;
    }
}

// rows of F = 64 or 128 floats on both sides (pitches in whole 16-byte slices), every row at most kBandEntries entries, every
// neighbour within kBandMax rows
bool banded_ok(const athena_mp_graph *g, bool transposed, int F, int64_t ldx, int64_t ldy, const float *x, const float *y)
{
#ifdef AGG_NO_BANDED
    return false;
#endif
    return g->band <= kBandMax && g->n_rows == g->n_cols && (F == 64 || F == 128) && ldx >= F && ldx % 4 == 0 && ldx <= 1024 &&
           ldy >= F && ldy % 4 == 0 && ldy <= 1024 && (transposed ? g->max_col_len : g->max_row_len) <= kBandEntries &&
           (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0;
}
int gather_banded(const athena_mp_graph *g, const int32_t *rowptr, const int32_t *idx, const float *coef, const float *x, int64_t ldx,
                  float *y, int64_t ldy, int F)
{
    if (g->n_rows == 0) return 0;
    const int chunks = (g->n_rows + kBandRows - 1) / kBandRows, per = (chunks + 7) / 8;
    for (int f0 = 0; f0 < F; f0 += 64) {
        if (coef)
            hipLaunchKernelGGL(csr_gather_banded64<true>, dim3(8 * per), dim3(256), 0, amp::stream(), rowptr, idx, coef, x + f0,
                               (int32_t)(ldx / 4), y + f0, (int32_t)(ldy / 4), g->n_rows, g->band, per);
        else
            hipLaunchKernelGGL(csr_gather_banded64<false>, dim3(8 * per), dim3(256), 0, amp::stream(), rowptr, idx, coef, x + f0,
                               (int32_t)(ldx / 4), y + f0, (int32_t)(ldy / 4), g->n_rows, g->band, per);
    }
    AMP_LAUNCH_CHECK();
    return 0;
}

// rows of at most kShortRow entries, F a power-of-two multiple of 16 floats up to 256, 16 B aligned slices
constexpr int kShortRow = 32;
bool short_rows_ok(int32_t max_row_len, int F, int Fe, const float *x, int64_t ldx, const float *e, const float *y, int64_t ldy)
{
    // (A/B against the general gather: 0.454 -> 0.384 ms at configs[2]; the switch is gone)
    const int G = F / 4;
    return max_row_len <= kShortRow && F % 16 == 0 && (G & (G - 1)) == 0 && G >= 4 && G <= 64 && Fe % 4 == 0 &&
           Fe / 4 <= G && ldx % 4 == 0 && ldy % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0 &&
           (Fe == 0 || (uintptr_t)e % 16 == 0);
}
int gather_short_rows(const int32_t *rowptr, const int32_t *idx, const int32_t *eidx, const float *x, int64_t ldx,
                      const float *e, int Fe, float *y, int64_t ldy, int32_t n_rows, int F)
{
    if (n_rows == 0) return 0;
    const int G = F / 4;
    const unsigned nb = (unsigned)(((int64_t)n_rows + 256 / G - 1) / (256 / G));
#define AMP_SR(G_)                                                                                                 \
    case G_:                                                                                                       \
        hipLaunchKernelGGL((csr_gather_short_rows<G_, 4>), dim3(nb), dim3(256), 0, amp::stream(), rowptr, idx, eidx, x, ldx, \
                           e, (int64_t)Fe, Fe / 4, y, ldy, n_rows);                                                \
        break;
    switch (G) {
        AMP_SR(4) AMP_SR(8) AMP_SR(16) AMP_SR(32) AMP_SR(64)
    default: amp::set_error("gather_short_rows: F = %d", F); return 2;
    }
#undef AMP_SR
    AMP_LAUNCH_CHECK();
    return 0;
}

template <int G, int VEC>
int launch_dual_gv(const int32_t *rowbeg, const int32_t *rowend, const int32_t *idx, const float *coef, const float *x,
                   int64_t ldx, float *y, float *y2, int64_t ldy, int32_t n_rows, int32_t F)
{
    constexpr int rpb = kBlock / G;
    int nb = (n_rows + rpb - 1) / rpb;
    if (amp::agg_blocks_cap() > 0) nb = std::min(nb, amp::agg_blocks_cap());
    hipLaunchKernelGGL((csr_gather_agg<G, VEC, true, true>), dim3(nb), dim3(kBlock), 0, amp::stream(), rowbeg, rowend,
                       idx, coef, x, ldx, y, ldy, n_rows, F, 0, y2, 0);
    AMP_LAUNCH_CHECK();
    return 0;
}
template <int VEC>
int launch_dual_v(int G, const int32_t *rowbeg, const int32_t *rowend, const int32_t *idx, const float *coef,
                  const float *x, int64_t ldx, float *y, float *y2, int64_t ldy, int32_t n_rows, int32_t F)
{
    switch (G) {
    case 1: return launch_dual_gv<1, VEC>(rowbeg, rowend, idx, coef, x, ldx, y, y2, ldy, n_rows, F);
    case 2: return launch_dual_gv<2, VEC>(rowbeg, rowend, idx, coef, x, ldx, y, y2, ldy, n_rows, F);
    case 4: return launch_dual_gv<4, VEC>(rowbeg, rowend, idx, coef, x, ldx, y, y2, ldy, n_rows, F);
    case 8: return launch_dual_gv<8, VEC>(rowbeg, rowend, idx, coef, x, ldx, y, y2, ldy, n_rows, F);
    case 16: return launch_dual_gv<16, VEC>(rowbeg, rowend, idx, coef, x, ldx, y, y2, ldy, n_rows, F);
    case 32: return launch_dual_gv<32, VEC>(rowbeg, rowend, idx, coef, x, ldx, y, y2, ldy, n_rows, F);
    default: return launch_dual_gv<64, VEC>(rowbeg, rowend, idx, coef, x, ldx, y, y2, ldy, n_rows, F);
    }
}

template <int G, int VEC>
int launch_gv(bool has_coef, const int32_t *rowbeg, const int32_t *rowend, const int32_t *idx, const float *coef,
              const float *x, int64_t ldx, float *y, int64_t ldy, int32_t n_rows, int32_t F, int32_t long_thr, int act)
{
    constexpr int rpb = kBlock / G;
    int nb = (n_rows + rpb - 1) / rpb;
    if (amp::agg_blocks_cap() > 0) nb = std::min(nb, amp::agg_blocks_cap());
    dim3 grid(nb), block(kBlock);
    if (has_coef)
        hipLaunchKernelGGL((csr_gather_agg<G, VEC, true>), grid, block, 0, amp::stream(), rowbeg, rowend, idx, coef, x,
                           ldx, y, ldy, n_rows, F, long_thr, (float *)nullptr, act);
    else
        hipLaunchKernelGGL((csr_gather_agg<G, VEC, false>), grid, block, 0, amp::stream(), rowbeg, rowend, idx, coef,
                           x, ldx, y, ldy, n_rows, F, long_thr, (float *)nullptr, act);
    AMP_LAUNCH_CHECK();
    return 0;
}

template <int VEC>
int launch_v(int G, bool has_coef, const int32_t *rowbeg, const int32_t *rowend, const int32_t *idx,
             const float *coef, const float *x, int64_t ldx, float *y, int64_t ldy, int32_t n_rows, int32_t F,
             int32_t long_thr, int act)
{
    switch (G) {
    case 1: return launch_gv<1, VEC>(has_coef, rowbeg, rowend, idx, coef, x, ldx, y, ldy, n_rows, F, long_thr, act);
    case 2: return launch_gv<2, VEC>(has_coef, rowbeg, rowend, idx, coef, x, ldx, y, ldy, n_rows, F, long_thr, act);
    case 4: return launch_gv<4, VEC>(has_coef, rowbeg, rowend, idx, coef, x, ldx, y, ldy, n_rows, F, long_thr, act);
    case 8: return launch_gv<8, VEC>(has_coef, rowbeg, rowend, idx, coef, x, ldx, y, ldy, n_rows, F, long_thr, act);
    case 16: return launch_gv<16, VEC>(has_coef, rowbeg, rowend, idx, coef, x, ldx, y, ldy, n_rows, F, long_thr, act);
    case 32: return launch_gv<32, VEC>(has_coef, rowbeg, rowend, idx, coef, x, ldx, y, ldy, n_rows, F, long_thr, act);
    default: return launch_gv<64, VEC>(has_coef, rowbeg, rowend, idx, coef, x, ldx, y, ldy, n_rows, F, long_thr, act);
    }
}

// y[long row, :] = sum of its segment partials, in segment order
__global__ void long_combine_kernel(const int32_t *__restrict__ row_id, const int32_t *__restrict__ row_task0,
                                    const float *__restrict__ partial, int64_t ldp, int32_t n_long, int32_t F,
                                    float *__restrict__ y, int64_t ldy, int32_t act)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)n_long * F) return;
    const int r = (int)(t / F), f = (int)(t - (int64_t)r * F);
    float s = 0.0f;
    for (int k = row_task0[r]; k < row_task0[r + 1]; ++k) s = s + partial[(size_t)k * ldp + f];
    y[(int64_t)row_id[r] * ldy + f] = agg_act(s, act);
}

} // namespace

namespace amp {

// true where athena_mp_kipf_propagate_fwd / _bwd take the LDS-staged gather for this graph and width (dense rows of F floats)
bool kipf_gather_is_banded(const athena_mp_graph *g, bool transposed, int F, const float *x, const float *y)
{
    return banded_ok(g, transposed, F, F, F, x, y);
}

// Generic launcher used by every layer family.  x/y may be column slices of wider tensors
// (ldx/ldy = leading dimension in floats).
int gather_agg(const int32_t *rowptr, const int32_t *idx, const float *coef, const float *x, int64_t ldx,
               float *y, int64_t ldy, int32_t n_rows, int32_t F, const LongPlan *lp, int act)
{
    if (n_rows == 0 || F == 0) return 0;
    // widest vector the slices allow (16 B loads need 16 B aligned rows)
    int vec = 1;
    auto aligned = [&](int v) {
        return F % v == 0 && ldx % v == 0 && ldy % v == 0 && ((uintptr_t)x % (4 * v)) == 0 &&
               ((uintptr_t)y % (4 * v)) == 0;
    };
    if (aligned(4)) vec = 4;
    else if (aligned(2)) vec = 2;
    int lanes = (F + vec - 1) / vec;
    int G = 1;
    while (G < lanes && G < 64) G <<= 1;
    const bool has_coef = coef != nullptr;
    const bool hubs = lp && lp->n_long > 0;
    auto run = [&](const int32_t *rb, const int32_t *re, float *out, int64_t ldo, int32_t rows, int32_t thr, int a) {
        switch (vec) {
        case 4: return launch_v<4>(G, has_coef, rb, re, idx, coef, x, ldx, out, ldo, rows, F, thr, a);
        case 2: return launch_v<2>(G, has_coef, rb, re, idx, coef, x, ldx, out, ldo, rows, F, thr, a);
        default: return launch_v<1>(G, has_coef, rb, re, idx, coef, x, ldx, out, ldo, rows, F, thr, a);
        }
    };
    int rc = run(rowptr, rowptr + 1, y, ldy, n_rows, hubs ? kLongRow : 0, act);
    if (rc || !hubs) return rc;
    // hub rows: one lane group per kLongRow-entry segment into a partial buffer, then an ordered combine
    void *ws = nullptr;
    const int64_t Fp = (F + 3) & ~3; // keep partial rows 16 B aligned
    if (workspace(&ws, sizeof(float) * (size_t)lp->n_tasks * Fp, 7)) return 1;
    {
        // the partial buffer's leading dimension must satisfy the same vector alignment as y
        const int vec_saved = vec;
        if (Fp % vec != 0) vec = 1;
        rc = run(lp->task_beg, lp->task_end, (float *)ws, Fp, lp->n_tasks, 0, 0);   // partial sums: no activation yet
        vec = vec_saved;
        if (rc) return rc;
    }
    const int64_t tot = (int64_t)lp->n_long * F;
    hipLaunchKernelGGL(long_combine_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream(), lp->row_id,
                       lp->row_task0, (const float *)ws, Fp, lp->n_long, F, y, ldy, act);
    AMP_LAUNCH_CHECK();
    return 0;
}

// One gather, two sums: y = sum of the rows, y2 = coefficient-weighted sum (dense rows, ld = F).  Graphs with hub
// rows (segment plan) take two single passes instead -- the segment path is not duplicated for this.
int gather_agg_dual(const int32_t *rowptr, const int32_t *idx, const float *coef, const float *x, float *y, float *y2,
                    int32_t n_rows, int32_t F, const LongPlan *lp)
{
    if (n_rows == 0 || F == 0) return 0;
    if (lp && lp->n_long > 0) {
        if (int rc = gather_agg(rowptr, idx, nullptr, x, F, y, F, n_rows, F, lp)) return rc;
        return gather_agg(rowptr, idx, coef, x, F, y2, F, n_rows, F, lp);
    }
    auto aligned = [&](int v) {
        return F % v == 0 && ((uintptr_t)x % (4 * v)) == 0 && ((uintptr_t)y % (4 * v)) == 0 &&
               ((uintptr_t)y2 % (4 * v)) == 0;
    };
    const int vec = aligned(4) ? 4 : (aligned(2) ? 2 : 1);
    const int lanes = (F + vec - 1) / vec;
    int G = 1;
    while (G < lanes && G < 64) G <<= 1;
    switch (vec) {
    case 4: return launch_dual_v<4>(G, rowptr, rowptr + 1, idx, coef, x, F, y, y2, F, n_rows, F);
    case 2: return launch_dual_v<2>(G, rowptr, rowptr + 1, idx, coef, x, F, y, y2, F, n_rows, F);
    default: return launch_dual_v<1>(G, rowptr, rowptr + 1, idx, coef, x, F, y, y2, F, n_rows, F);
    }
}

} // namespace amp

using namespace amp;

extern "C" {

int athena_mp_kipf_propagate_fwd(const athena_mp_graph *g, int32_t F, const float *x, float *y)
{
    AMP_REQUIRE(g && F > 0, "kipf_propagate_fwd: bad arguments");
    if (g->n_rows == 0) return 0; // empty graph: nothing to do (pointers may be null)
    AMP_REQUIRE(x && y, "kipf_propagate_fwd: null tensor");
    // a block-diagonal batch of small graphs (onnx_gnn / msgpass_chemical-shaped Kipf batches): the rows a block can touch staged in LDS
    if (banded_ok(g, false, F, F, F, x, y)) return gather_banded(g, g->rowptr, g->col, g->coef, x, F, y, F, F);
    return gather_agg(g->rowptr, g->col, g->coef, x, F, y, F, g->n_rows, F, &g->lp_fwd);
}

/* y = act( kipf_propagate(x) ): the activation of a time step whose dense step ran BEFORE the aggregation
 * (Z = A^ (X W^T)) applied in the aggregation's store instead of a pass of its own */
int athena_mp_kipf_propagate_act_fwd(const athena_mp_graph *g, int32_t F, const float *x, int32_t act, float *y)
{
    AMP_REQUIRE(g && F > 0, "kipf_propagate_act_fwd: bad arguments");
    AMP_REQUIRE(act >= 0 && act <= ATHENA_MP_ACT_TANH, "kipf_propagate_act_fwd: unknown activation %d", act);
    if (g->n_rows == 0) return 0;
    AMP_REQUIRE(x && y, "kipf_propagate_act_fwd: null tensor");
    return gather_agg(g->rowptr, g->col, g->coef, x, F, y, F, g->n_rows, F, &g->lp_fwd, act);
}

int athena_mp_kipf_propagate_bwd(const athena_mp_graph *g, int32_t F, const float *grad, float *dx,
                                 int32_t exact)
{
    AMP_REQUIRE(g && F > 0, "kipf_propagate_bwd: bad arguments");
    if (g->n_cols == 0) return 0;
    AMP_REQUIRE(dx && (grad || g->nnz == 0), "kipf_propagate_bwd: null tensor");
    if (grad && banded_ok(g, true, F, F, F, grad, dx)) return gather_banded(g, g->t_rowptr, g->t_src, exact ? g->t_coef : nullptr, grad, F, dx, F, F);
    return gather_agg(g->t_rowptr, g->t_src, exact ? g->t_coef : nullptr, grad, F, dx, F, g->n_cols, F, &g->lp_bwd);
}

/* reverse_kipf_propagate and its two partial forms (athena_diffstruc_extd_sub_kipf.f90:116-205) */
int athena_mp_reverse_kipf_propagate_fwd(const athena_mp_graph *g, int32_t F, const float *a, float *c)
{
    return athena_mp_kipf_propagate_bwd(g, F, a, c, 0);
}
int athena_mp_reverse_kipf_propagate_partial(const athena_mp_graph *g, int32_t F, const float *upstream, float *out)
{
    return athena_mp_kipf_propagate_fwd(g, F, upstream, out);
}
int athena_mp_reverse_kipf_propagate_partial_val(const athena_mp_graph *g, int32_t F, const float *upstream, float *out)
{
    return athena_mp_kipf_propagate_bwd(g, F, upstream, out, 0);
}

/* the pull form of the same pair on a row shard (athena_amd/dist.py): over the FORWARD rows of g,
 * y_plain[v] = sum_w x[col w], y_coef[v] = sum_w coef_w x[col w] -- one gather of the (local + halo) rows */
int athena_mp_kipf_propagate_fwd_dual(const athena_mp_graph *g, int32_t F, const float *x, float *y_plain, float *y_coef)
{
    AMP_REQUIRE(g && F > 0, "kipf_propagate_fwd_dual: bad arguments");
    if (g->n_rows == 0) return 0;
    AMP_REQUIRE(y_plain && y_coef && (x || g->nnz == 0), "kipf_propagate_fwd_dual: null tensor");
    return gather_agg_dual(g->rowptr, g->col, g->coef, x, y_plain, y_coef, g->n_rows, F, &g->lp_fwd);
}

/* both reverse forms of kipf_propagate from one gather of the upstream rows: dx_plain = the reference's
 * coefficient-free scatter (get_partial_kipf_propagate_left_val :85-111), dx_coef = the adjoint of the forward */
int athena_mp_kipf_propagate_bwd_dual(const athena_mp_graph *g, int32_t F, const float *grad, float *dx_plain,
                                      float *dx_coef)
{
    AMP_REQUIRE(g && F > 0, "kipf_propagate_bwd_dual: bad arguments");
    if (g->n_cols == 0) return 0;
    AMP_REQUIRE(dx_plain && dx_coef && (grad || g->nnz == 0), "kipf_propagate_bwd_dual: null tensor");
    return gather_agg_dual(g->t_rowptr, g->t_src, g->t_coef, grad, dx_plain, dx_coef, g->n_cols, F, &g->lp_bwd);
}

int athena_mp_duvenaud_propagate_fwd(const athena_mp_graph *g, int32_t Fv, int32_t Fe, const float *x,
                                     const float *e, float *c)
{
    AMP_REQUIRE(g && c && Fv >= 0 && Fe >= 0 && Fv + Fe > 0 && (Fv == 0 || x), "duvenaud_propagate_fwd: bad arguments");
    AMP_REQUIRE(Fe == 0 || e, "duvenaud_propagate_fwd: null edge features");
    const int64_t Fc = (int64_t)Fv + Fe;
    if (Fe == 0 && banded_ok(g, false, Fv, Fv, Fv, x, c)) return gather_banded(g, g->rowptr, g->col, nullptr, x, Fv, c, Fv, Fv);
    if (Fv == 0)   // the edge part alone, [n_rows, Fe]: the same sums in the same order as columns Fv .. of the packed form
        return gather_agg(g->rowptr, g->eid, nullptr, e, Fe, c, Fe, g->n_rows, Fe, &g->lp_fwd);
    if (short_rows_ok(g->max_row_len, Fv, Fe, x, Fv, e, c, Fc))   // molecule-sized rows: one launch, whole rows written
        return gather_short_rows(g->rowptr, g->col, g->eid, x, Fv, e, Fe, c, Fc, g->n_rows, Fv);
    int rc = gather_agg(g->rowptr, g->col, nullptr, x, Fv, c, Fc, g->n_rows, Fv, &g->lp_fwd);
    if (rc == 0 && Fe > 0) rc = gather_agg(g->rowptr, g->eid, nullptr, e, Fe, c + Fv, Fc, g->n_rows, Fe, &g->lp_fwd);
    return rc;
}

int athena_mp_duvenaud_propagate_bwd_x(const athena_mp_graph *g, int32_t Fv, int32_t Fe, const float *grad,
                                       float *dx)
{
    AMP_REQUIRE(g && grad && dx && Fv > 0 && Fe >= 0, "duvenaud_propagate_bwd_x: bad arguments");
    // (the vertex part of packed rows [n, Fv + Fe] as well: the staging loads skip the edge part of every row)
    if (banded_ok(g, true, Fv, (int64_t)Fv + Fe, Fv, grad, dx)) return gather_banded(g, g->t_rowptr, g->t_src, nullptr, grad, (int64_t)Fv + Fe, dx, Fv, Fv);
    return gather_agg(g->t_rowptr, g->t_src, nullptr, grad, (int64_t)Fv + Fe, dx, Fv, g->n_cols, Fv, &g->lp_bwd);
}

int athena_mp_duvenaud_propagate_bwd_e(const athena_mp_graph *g, int32_t Fv, int32_t Fe, const float *grad,
                                       float *de)
{
    AMP_REQUIRE(g && grad && de && Fv >= 0 && Fe > 0, "duvenaud_propagate_bwd_e: bad arguments");   // Fv = 0: grad holds the edge part alone
    return gather_agg(g->e_rowptr, g->e_row, nullptr, grad + Fv, (int64_t)Fv + Fe, de, Fe, g->n_edge_cols, Fe);
}

int athena_mp_gather_rows(int64_t n, int32_t F, const int32_t *idx, const float *x, float *out)
{
    AMP_REQUIRE(n >= 0 && F > 0 && n < (int64_t)INT32_MAX, "gather_rows: bad arguments");
    if (n == 0) return 0;
    AMP_REQUIRE(idx && x && out, "gather_rows: null pointer");
    // identity row pointer [0,1,..,n]: every output row has exactly one entry
    static int64_t iota_n[amp::kMaxDevices] = {};      // elements of the per-device buffer "agg.iota"
    int32_t *iota = nullptr;
    int64_t &have = iota_n[amp::device()];
    const int64_t want = have < n + 1 ? n + 1 + (n >> 2) : have;
    bool fresh = false;
    if (amp::named_buffer("agg.iota", sizeof(int32_t) * (size_t)want, false, (void **)&iota, &fresh)) return 1;
    if (fresh) {
        std::vector<int32_t> h(want);
        for (int64_t i = 0; i < want; ++i) h[i] = (int32_t)i;
        AMP_HIP(hipMemcpy(iota, h.data(), sizeof(int32_t) * want, hipMemcpyHostToDevice));
        have = want;
    }
    return gather_agg(iota, idx, nullptr, x, F, out, F, (int32_t)n, F);
}

} // extern "C"
