// Reverse pass of a square Kipf layer step in ONE launch: input gradient AND weight gradient.
//
//   dX = (A^T dZ) . W          matmul reverse + get_partial_kipf_propagate_left_val (reference: no coefficient;
//                              athena_diffstruc_extd_sub_kipf.f90:85-111), as in athena_mp_kipf_layer_bwd_x
//   dW = dZ . P^T,  P = A^ X   matmul reverse wrt the weights (call site athena_kipf_msgpass_layer.f90:951), re-associated:
//        = sum_u Qc[u] (x) X[u]     with Qc = A^^T dZ (the coefficient-weighted pull over the transposed CSR)
//
// Why: the stock sequence writes P in the forward launch (0.5 GB at configs[1]) only so that a third kernel can read it
// back with dZ (1 GB) for dW.  Both reverse sums -- the plain one for dX and the coefficient-weighted one for dW -- are sums
// over the SAME gathered dZ rows, so one gather feeds both contractions; the forward launch no longer stores P and the
// dW launch disappears: -1.5 GB of the step's ~13 GB and one launch.
//
// Shape: an 8-wave workgroup per CU (two waves per SIMD, 256 registers per lane -- the 16-wave kernels of fused.hip sit at
// 122 of their 128 registers and have no room for a second accumulator set).  W stays in LDS as in fused.hip; 32-row
// chunks; every wave gathers two row pairs (half-wave per row, CSR-order sums, both pairs advancing together), then
//   dX block:  2 of the chunk's 16 [16 x 16] blocks of Q . W       (A = Q tile in LDS, B = W in LDS)
//   dW tiles:  8 of the 64 [16 x 16] tiles of X^T Qc, accumulated over ALL chunks the workgroup draws
//              (A = X rows of the chunk's vertices, read straight from global memory in fragment order -- each row is
//               read once chip-wide; B = Qc tile in LDS, pitch 144: conflict-free for the one-word-per-lane reads)
// The per-workgroup dW tiles go to a slab each and are summed in fixed order (amp::slab_reduce); chunks are walked with a
// fixed stride, so the result is the same bits on every launch.
//
// Measured at configs[1] (1 M vertices / 10 M entries, MI355X): this launch 1.20 ms against 0.88 (dX only, fused.hip) +
// 0.31 (gemm_dw_full_kernel) = 1.19 ms for the two it replaces; forward without the P store 0.93 against 0.99 in the
// step; whole step 2.12 ms either way.  The bytes go down by 1.5 GB, the time does not: per chunk the gather costs
// ~7 us of which ~3 us are vector instructions (index broadcasts, predication, the unfused multiply + add of the
// bit-exact sums), and the extra 32 MFMAs per wave land on the same issue ports instead of under the row loads.  What it
// does buy: no P tensor kept per layer (0.5 GB at configs[1]) and one launch fewer.  An 8-wave form (DW_WAVES=8, 256
// registers, 12-16 rows prefetched per row) measured 1.39-1.43 ms: with two waves per SIMD the matrix work and the row
// loads do not overlap at all.  Staggering the two jobs of a wave by wave number (waves 4-7, 12-15 gather first) took
// 1.26 -> 1.20 ms and is kept (DW_STAGGER).
#include <algorithm>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kN = 128;          // features (square step)
constexpr int kLD = kN + 4;      // pitch of W and of the Q tile (k-contiguous 16 B reads)
constexpr int kLQ = kN + 16;     // pitch of the Qc tile (one word per lane: rows 4s + g4, columns 16 t + l15)
constexpr int kCH = 32;          // rows per chunk
#define KFD 8
constexpr int kFd = KFD;         // row loads per row issued ahead of the matrix work
constexpr int kTd = 4;           // entries per row per round beyond them

#define DW_STAGGER 1
#define DW_WAVES 16
constexpr int kNW = DW_WAVES;                  // waves per workgroup: 16 (one row pair, one dX block, 4 dW tiles per wave;
                                               // 128 registers) or 8 (two pairs, two blocks, 8 tiles; 256 registers)
constexpr int kPairs = 16 / kNW;               // row pairs a wave gathers per 32-row chunk
constexpr int kDwT = 64 / kNW;                 // dW tiles per wave

template <bool EXACT>
__global__ __launch_bounds__(64 * kNW) void agg_gemm_dw_kernel(const int32_t *__restrict__ rowptr,
                                                          const int32_t *__restrict__ idx,
                                                          const float *__restrict__ coef,
                                                          const float *__restrict__ dz,
                                                          const float *__restrict__ W,     // [N = Fi][K = Fo]
                                                          const float *__restrict__ X,     // [n_rows][Fi]
                                                          float *__restrict__ dX,          // [n_rows][Fi] or null
                                                          float *__restrict__ slabs,       // [grid][Fi * Fo]
                                                          int64_t n_rows, unsigned long long *__restrict__ ticket,
                                                          uint32_t dz_bytes)
{
    constexpr int K = kN, N = kN, G = 32, KG = K / 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Bs = lds;                               // [N][kLD]
    float *Ts = Bs + N * kLD;                      // [2][kCH][kLD]   plain sums Q (dX)
    float *Qs = Ts + 2 * kCH * kLD;                // [2][kCH][kLQ]   coefficient-weighted sums Qc (dW)
    int64_t *s_ticket = reinterpret_cast<int64_t *>(Qs + 2 * kCH * kLQ);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, gl = lane & 31;       // half-wave (row of the pair) and lane inside it
    const int l15 = lane & 15, g4 = lane >> 4;

    {   // W resident in LDS as [n][K + 4] (stored [N][K]: rows contiguous)
        constexpr int NV = K * N / 4, PER = NV / (64 * kNW);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int t = i * 64 * kNW + tid;
            const v4f w4 = reinterpret_cast<const v4f *>(W)[t];
            const int n = t / (K / 4), q = t - n * (K / 4);
            *reinterpret_cast<v4f *>(Bs + n * kLD + 4 * q) = w4;
        }
    }
    const int64_t n_chunks = (n_rows + kCH - 1) / kCH;

    // ---- gather state: this half-wave's row of pair p (chunk rows 4 wave + 2 p + h) --------------------------------------
    int start[kPairs], len[kPairs], idx0[kPairs];
    float c0[kPairs];
    auto load_state = [&](int64_t chunk) {
#pragma unroll
        for (int p = 0; p < kPairs; ++p) {
            start[p] = 0; len[p] = 0; idx0[p] = -1; c0[p] = 0.0f;
            const int64_t row = chunk * kCH + 2 * kPairs * wave + 2 * p + h;
            if (chunk < n_chunks && row < n_rows) {
                start[p] = rowptr[row];
                len[p] = rowptr[row + 1] - start[p];
            }
            if (gl < len[p]) {
                idx0[p] = idx[start[p] + gl];
                c0[p] = coef[start[p] + gl];
            }
        }
    };
    // row loads: 32-bit byte offsets against a buffer descriptor when dZ is below 4 GB (fused.hip)
    __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)dz, 0, dz_bytes ? (int)dz_bytes : 0, 0x00020000);
    auto load_row = [&](int u) -> v4f {
        if (dz_bytes) {   // workgroup-uniform
            typedef int v4i_ __attribute__((ext_vector_type(4)));
            const v4i_ t = __builtin_amdgcn_raw_buffer_load_b128(zrsrc, (int)(((uint32_t)u << 9) + 16u * (uint32_t)gl), 0, 0);   // byte offset in uint32: tensors of 2 .. 4 GB
            return __builtin_bit_cast(v4f, t);
        }
        return *reinterpret_cast<const v4f *>(dz + (int64_t)u * K + 4 * gl);
    };
    v4f v[kPairs][kFd];
    auto issue_first = [&]() {
#pragma unroll
        for (int p = 0; p < kPairs; ++p)
#pragma unroll
            for (int k = 0; k < kFd; ++k) {
                const int u = __shfl(idx0[p], k, G);
                v[p][k] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
                if (k < len[p]) v[p][k] = load_row(u);
            }
    };
    auto fma4 = [](v4f &a, float c, const v4f &x) { a.x = a.x + c * x.x; a.y = a.y + c * x.y; a.z = a.z + c * x.z; a.w = a.w + c * x.w; };
    auto add4 = [](v4f &a, const v4f &x) { a.x = a.x + x.x; a.y = a.y + x.y; a.z = a.z + x.z; a.w = a.w + x.w; };
    auto store_rows = [&](int64_t chunk, int buf) {
        v4f acc[kPairs], accc[kPairs];       // plain and coefficient-weighted sums, both in CSR order
#pragma unroll
        for (int p = 0; p < kPairs; ++p) {
            acc[p] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
            accc[p] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < kFd; ++k) {
                const float c = __shfl(c0[p], k, G);
                if (k < len[p]) {
                    if constexpr (!EXACT) add4(acc[p], v[p][k]);
                    fma4(accc[p], c, v[p][k]);
                }
            }
        }
        // entries beyond the prefetched block: both pairs advance together; trip counts are wave-uniform (the longest
        // of the wave's four rows), shorter rows are predicated
        int maxlen = len[0];
#pragma unroll
        for (int p = 1; p < kPairs; ++p) maxlen = max(maxlen, len[p]);
        maxlen = max(maxlen, __shfl_xor(maxlen, 32));
        for (int off = 0; off < maxlen; off += G) {
            int my_idx[kPairs];
            float my_c[kPairs];
#pragma unroll
            for (int p = 0; p < kPairs; ++p) {
                my_idx[p] = idx0[p]; my_c[p] = c0[p];
                if (off > 0) {
                    my_idx[p] = -1; my_c[p] = 0.0f;
                    if (off + gl < len[p]) {
                        my_idx[p] = idx[start[p] + off + gl];
                        my_c[p] = coef[start[p] + off + gl];
                    }
                }
            }
            const int cntmax = min(G, maxlen - off);
            for (int j = (off == 0 ? kFd : 0); j < cntmax; j += kTd) {
                int u[kPairs][kTd];
                float c[kPairs][kTd];
                v4f w[kPairs][kTd];
#pragma unroll
                for (int p = 0; p < kPairs; ++p)
#pragma unroll
                    for (int k = 0; k < kTd; ++k) {
                        u[p][k] = __shfl(my_idx[p], j + k, G);
                        c[p][k] = __shfl(my_c[p], j + k, G);
                        if (off + j + k >= len[p]) u[p][k] = -1;
                    }
#pragma unroll
                for (int p = 0; p < kPairs; ++p)
#pragma unroll
                    for (int k = 0; k < kTd; ++k) {
                        w[p][k] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
                        if (u[p][k] >= 0) w[p][k] = load_row(u[p][k]);
                    }
#pragma unroll
                for (int p = 0; p < kPairs; ++p)
#pragma unroll
                    for (int k = 0; k < kTd; ++k)
                        if (u[p][k] >= 0) {
                            if constexpr (!EXACT) add4(acc[p], w[p][k]);
                            fma4(accc[p], c[p][k], w[p][k]);
                        }
            }
        }
#pragma unroll
        for (int p = 0; p < kPairs; ++p) {
            const int lrow = 2 * kPairs * wave + 2 * p + h;
            *reinterpret_cast<v4f *>(Ts + (buf * kCH + lrow) * kLD + 4 * gl) = EXACT ? accc[p] : acc[p];
            *reinterpret_cast<v4f *>(Qs + (buf * kCH + lrow) * kLQ + 4 * gl) = accc[p];
        }
    };

    // ---- matrix work of one chunk -----------------------------------------------------------------------------------------
    // dX: 16 blocks of [16 x 16] per chunk: rows 16 rb .. +15, column tiles ct0 .. ct0 + kNB - 1
    // dW: 64 tiles: row tile (of Fi) mt, column tiles (of Fo) nt0 .. nt0 + kDwT - 1
    constexpr int kNB = 16 / kNW;
    const int rb = wave & 1, ct0 = (wave >> 1) * kNB;
    const int mt = wave & 7, nt0 = (wave >> 3) * kDwT;
    f32x4 dw[kDwT];
#pragma unroll
    for (int t = 0; t < kDwT; ++t) dw[t] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    auto matrix_work = [&](int64_t chunk, int buf) {
        // A operand of dW: X[row0 + 4 s + g4][16 mt + l15], s = 0..7 -- issued first, consumed last
        float xa[8];
        const int64_t row0 = chunk * kCH;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int64_t r = row0 + 4 * s + g4;
            xa[s] = r < n_rows ? X[r * N + 16 * mt + l15] : 0.0f;
        }
        if (dX != nullptr) {
            const float *arow = Ts + (buf * kCH + 16 * rb + l15) * kLD + KG * g4;
            const float *b0 = Bs + (16 * ct0 + l15) * kLD + KG * g4;
            f32x4 cc[kNB];
#pragma unroll
            for (int j = 0; j < kNB; ++j) cc[j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int q = 0; q < KG / 4; ++q) {
                const v4f a4 = *reinterpret_cast<const v4f *>(arow + 4 * q);
                v4f b4[kNB];
#pragma unroll
                for (int j = 0; j < kNB; ++j) b4[j] = *reinterpret_cast<const v4f *>(b0 + 16 * j * kLD + 4 * q);
#pragma unroll
                for (int j = 0; j < kNB; ++j) cc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4[j].x, cc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < kNB; ++j) cc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4[j].y, cc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < kNB; ++j) cc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4[j].z, cc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < kNB; ++j) cc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4[j].w, cc[j], 0, 0, 0);
            }
            const int64_t r0 = row0 + 16 * rb + 4 * g4;     // C/D: col = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r0 + r < n_rows) {
#pragma unroll
                    for (int j = 0; j < kNB; ++j) dX[(r0 + r) * N + 16 * (ct0 + j) + l15] = cc[j][r];
                }
        }
        const float *qrow = Qs + (buf * kCH + g4) * kLQ + 16 * nt0 + l15;
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int t = 0; t < kDwT; ++t)
                dw[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], qrow[(4 * s) * kLQ + 16 * t], dw[t], 0, 0, 0);
    };

    // chunks are walked with a FIXED stride (workgroup b takes b, b + grid, ...), not drawn from a ticket counter as
    // in fused.hip: the dW tiles are sums over the workgroup's chunks, so who takes which chunk decides the order of an
    // fp32 sum -- with a fixed walk the result is the same bits on every launch
    (void)ticket; (void)s_ticket;
    int64_t k0 = blockIdx.x, k1 = k0 + gridDim.x, k2 = k1 + gridDim.x;
    load_state(k0);
    issue_first();
    store_rows(k0, 0);
    load_state(k1);
    __syncthreads();
    for (int it = 0; k0 < n_chunks; ++it) {
        const int buf = it & 1;
        issue_first();                                      // rows of k1 fly under the matrix work of k0
#if DW_STAGGER
        // Between two barriers a wave has two independent jobs (matrix work on tile `buf`, gather into `buf ^ 1`).
        // Waves 4-7 and 12-15 run them in the opposite order: each SIMD then holds two waves on the matrix pipe and two
        // waiting for rows, instead of all four queueing for the pipe and then all four waiting on memory.
        if ((wave >> 2) & 1) {
            store_rows(k1, buf ^ 1);
            matrix_work(k0, buf);
        } else
#endif
        {
            matrix_work(k0, buf);
            store_rows(k1, buf ^ 1);
        }
        load_state(k2);
        __syncthreads();
        k0 = k1;
        k1 = k2;
        k2 += gridDim.x;
    }
    // this workgroup's share of dW: tile (mt, nt0 + t): rows i = 16 mt + 4 g4 + r of Fi, columns o = 16 (nt0 + t) + l15 of Fo;
    // dW(Fo, Fi) column-major flat == row-major [Fi][Fo]
    float *slab = slabs + (size_t)blockIdx.x * N * K;
#pragma unroll
    for (int t = 0; t < kDwT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(16 * mt + 4 * g4 + r) * K + 16 * (nt0 + t) + l15] = dw[t][r];
}

} // namespace

namespace amp {

bool fused_dw_shape(int Fi, int Fo) { return Fi == kN && Fo == kN; }

// dX (may be null) and dW of a square step from one gather of dZ over the transposed CSR
int fused_dw_dispatch(const int32_t *t_rowptr, const int32_t *t_src, const float *t_coef, const float *dZ, const float *W,
                      const float *X, int exact, float *dX, float *dW, int64_t n_cols)
{
    const int grid = (int)std::min<int64_t>((n_cols + kCH - 1) / kCH, amp::num_cus());
    if (grid == 0) {
        AMP_HIP(hipMemsetAsync(dW, 0, sizeof(float) * kN * kN, amp::stream()));
        return 0;
    }
    constexpr size_t lds = sizeof(float) * ((size_t)kN * kLD + 2 * kCH * kLD + 2 * kCH * kLQ) + 4 * sizeof(int64_t);
    static amp::PerDeviceFlag attr;
    if (!attr.get()) {
        AMP_HIP(hipFuncSetAttribute((const void *)agg_gemm_dw_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        AMP_HIP(hipFuncSetAttribute((const void *)agg_gemm_dw_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.get() = true;
    }
    static int slot = 0;
    unsigned long long *ring = nullptr;
    if (amp::named_buffer("fused.ticket_ring", sizeof(unsigned long long) * 64, true, (void **)&ring)) return 1;
    unsigned long long *ticket = ring + (slot++ & 63);
    AMP_HIP(hipMemsetAsync(ticket, 0, sizeof(unsigned long long), amp::stream()));
    void *slabs = nullptr;
    if (amp::workspace(&slabs, sizeof(float) * (size_t)grid * kN * kN, 6)) return 1;
    const int64_t zb = n_cols * kN * 4;   // square graph: dZ has n_cols rows
    const uint32_t dz_bytes = zb < ((int64_t)1 << 32) - 4096 ? (uint32_t)zb : 0u;
    if (exact)
        hipLaunchKernelGGL(agg_gemm_dw_kernel<true>, dim3(grid), dim3(64 * kNW), lds, amp::stream(), t_rowptr, t_src, t_coef, dZ, W, X,
                           dX, (float *)slabs, n_cols, ticket, dz_bytes);
    else
        hipLaunchKernelGGL(agg_gemm_dw_kernel<false>, dim3(grid), dim3(64 * kNW), lds, amp::stream(), t_rowptr, t_src, t_coef, dZ, W, X,
                           dX, (float *)slabs, n_cols, ticket, dz_bytes);
    AMP_LAUNCH_CHECK();
    return amp::slab_reduce((const float *)slabs, grid, kN * kN, dW, false);
}

} // namespace amp
