// Shape-generic fp32 MFMA contractions (any M, N, K; optional row indirection).
//
// gemm.hip keeps the weight matrix resident in LDS, which only works for K, N <= 128.  These two
// kernels tile all three dimensions and serve everything else on the path:
//   gemm_tiled      C[c_idx[m], :] = act( (A[a_idx[m], :] / a_div) . B / c_div + bias )
//                   - Kipf dense step at F = 256 (BASELINE configs[4])
//                   - GNO contraction m = S . Vaug, K = (H+1) F_in = 4160  (gno.hip)
//                   - Duvenaud readout R z and its dx  (N or K = num_outputs = 10)
//                   - Duvenaud degree-bucketed update: one launch per bucket over a bucket-sorted
//                     vertex permutation, A rows divided by the bucket index on load exactly as the
//                     reference does (athena_diffstruc_extd_sub_duvenaud.f90:209-210)
//   gemm_atb_tiled  C[i, o] = sum_m A[idx[m], i] B[idx[m], o] / div      (reductions over vertices)
//                   - GNO dVaug = S^T g, Duvenaud dW_d, readout dR
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32).  Zero-fill handles every ragged edge.
#include <algorithm>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 64, BK = 32, LDT = BK + 4;

__device__ __forceinline__ float act_rt(float z, int act)
{
    switch (act) {
    case ATHENA_MP_ACT_RELU: return z > 0.0f ? z : 0.0f;
    case ATHENA_MP_ACT_SIGMOID: return 1.0f / (1.0f + expf(-z));
    case ATHENA_MP_ACT_TANH: return tanhf(z);
    default: return z;
    }
}

// 4 consecutive floats p[0..3] with element-wise bound `nvalid` (vector fast path when allowed)
__device__ __forceinline__ v4f load4(const float *__restrict__ p, int nvalid, bool vec_ok)
{
    v4f r = {0.0f, 0.0f, 0.0f, 0.0f};
    if (p == nullptr || nvalid <= 0) return r;
    if (vec_ok && nvalid >= 4) return *reinterpret_cast<const v4f *>(p);
    r.x = p[0];
    if (nvalid > 1) r.y = p[1];
    if (nvalid > 2) r.z = p[2];
    if (nvalid > 3) r.w = p[3];
    return r;
}

// WM = rows of C per wave (32 or 64); the workgroup tile is (4 WM) x 64.  WM = 64 halves the LDS reads,
// barriers and B re-reads per MFMA and is used when M gives enough workgroups to fill the chip.
template <int WM>
__global__ __launch_bounds__(256) void gemm_tiled_kernel(amp::TiledArgs p)
{
    constexpr int BMT = 4 * WM, NA = BMT * 8 / 256, RB = WM / 32;
    __shared__ __attribute__((aligned(16))) float As[BMT * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r31 = lane & 31, h = lane >> 5;
    // column block fastest: the workgroups that run together write neighbouring pieces of the SAME C rows (whole
    // DRAM pages instead of one 256 B piece per 4*N-byte row pitch) and share their A tile through L2
    const int ncb = (p.N + BN - 1) / BN;
    const int64_t m0 = (int64_t)(blockIdx.x / ncb) * BMT;
    const int n0 = (int)(blockIdx.x % ncb) * BN;
    const bool a_vec = (p.lda % 4 == 0) && ((uintptr_t)p.A % 16 == 0);
    const bool b_vec = (p.ldb % 4 == 0) && ((uintptr_t)p.B % 16 == 0);

    // this thread's NA A rows (row = t>>3, k-quad = t&7)
    const float *ap[NA];
#pragma unroll
    for (int it = 0; it < NA; ++it) {
        const int64_t m = m0 + ((it * 256 + tid) >> 3);
        ap[it] = nullptr;
        if (m < p.M) {
            const int64_t phys = p.a_idx ? (int64_t)p.a_idx[m] : m;
            ap[it] = p.A + phys * p.lda;
        }
    }
    v4f ra[NA], rb[2];
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int it = 0; it < NA; ++it) {
            const int q = (it * 256 + tid) & 7;
            v4f v = load4(ap[it] ? ap[it] + k0 + 4 * q : nullptr, p.K - (k0 + 4 * q), a_vec);
            if (p.a_div != 1.0f) { v.x = v.x / p.a_div; v.y = v.y / p.a_div; v.z = v.z / p.a_div; v.w = v.w / p.a_div; }
            ra[it] = v;
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int t = it * 256 + tid;
            if (p.b_nk) { // B stored [N][K]
                const int n = t >> 3, q = t & 7;
                const bool ok = n0 + n < p.N;
                rb[it] = load4(ok ? p.B + (int64_t)(n0 + n) * p.ldb + k0 + 4 * q : nullptr, p.K - (k0 + 4 * q), b_vec);
            } else {      // B stored [K][N]
                const int k = t >> 4, q4 = t & 15;
                const bool ok = k0 + k < p.K;
                rb[it] = load4(ok ? p.B + (int64_t)(k0 + k) * p.ldb + n0 + 4 * q4 : nullptr, p.N - (n0 + 4 * q4), b_vec);
            }
        }
    };
    f32x16 acc[RB][2];
#pragma unroll
    for (int a = 0; a < RB; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.0f;

    load_chunk(0);
    for (int k0 = 0; k0 < p.K; k0 += BK) {
        __syncthreads(); // previous chunk's fragment reads are done
#pragma unroll
        for (int it = 0; it < NA; ++it) {
            const int t = it * 256 + tid;
            *reinterpret_cast<v4f *>(As + (t >> 3) * LDT + 4 * (t & 7)) = ra[it];
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int t = it * 256 + tid;
            if (p.b_nk) {
                *reinterpret_cast<v4f *>(Bs + (t >> 3) * LDT + 4 * (t & 7)) = rb[it];
            } else {
                const int k = t >> 4, q4 = t & 15;
                Bs[(4 * q4 + 0) * LDT + k] = rb[it].x;
                Bs[(4 * q4 + 1) * LDT + k] = rb[it].y;
                Bs[(4 * q4 + 2) * LDT + k] = rb[it].z;
                Bs[(4 * q4 + 3) * LDT + k] = rb[it].w;
            }
        }
        __syncthreads();
        if (k0 + BK < p.K) load_chunk(k0 + BK); // in flight under the MFMAs
        const float *arow = As + (WM * wave + r31) * LDT + 16 * h;
        const float *brow = Bs + r31 * LDT + 16 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v4f a4[RB];
#pragma unroll
            for (int a = 0; a < RB; ++a) a4[a] = *reinterpret_cast<const v4f *>(arow + 32 * a * LDT + 4 * q);
            const v4f b0 = *reinterpret_cast<const v4f *>(brow + 4 * q);
            const v4f b1 = *reinterpret_cast<const v4f *>(brow + 32 * LDT + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int a = 0; a < RB; ++a) {
                    acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[a][e], b0[e], acc[a][0], 0, 0, 0);
                    acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[a][e], b1[e], acc[a][1], 0, 0, 0);
                }
        }
    }
    // epilogue straight from the accumulators (col = lane&31, row = (r&3) + 8*(r>>2) + 4*h)
#pragma unroll
    for (int a = 0; a < RB; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = n0 + 32 * c + r31;
            if (col >= p.N) continue;
            const float bv = p.bias ? p.bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + WM * wave + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < p.M) {
                    const int64_t phys = p.c_idx ? (int64_t)p.c_idx[m] : m;
                    float v = acc[a][c][r];
                    if (p.c_div != 1.0f) v = v / p.c_div;
                    p.C[phys * p.ldc + col] = act_rt(v + bv, p.act);
                }
            }
        }
}

// ---- short contraction, very wide output:  C[M, N] = A[M, 64] . B^T,  B stored [N][64],  N >> 64 ------------------
// (GNO: G = g . Vmat^T with N = H*F_in = 4096.)  The generic kernel above spends most of such a launch outside the
// matrix pipe -- two k-iterations per 256 x 64 tile, 32 k workgroups whose first load nothing hides.  Here a workgroup
// keeps its 128 rows of A in LDS and walks all column blocks: B blocks (16 KB, L2) are fetched one block ahead into
// registers, one barrier per block, and the workgroup writes whole C rows piece by piece (DRAM pages stay open).
constexpr int AR_K = 64, AR_LD = AR_K + 4, AR_BM = 128;

__global__ __launch_bounds__(256) void gemm_arow_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                        float *__restrict__ C, int64_t M, int N)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *As = sm;                       // [128][68]
    float *Bs = sm + AR_BM * AR_LD;       // [2][64][68]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r31 = lane & 31, h = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * AR_BM;
#pragma unroll
    for (int it = 0; it < 8; ++it) {      // A tile: 128 rows x 16 float4
        const int t = it * 256 + tid, row = t >> 4, q = t & 15;
        v4f v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (m0 + row < M) v = *reinterpret_cast<const v4f *>(A + (m0 + row) * AR_K + 4 * q);
        *reinterpret_cast<v4f *>(As + row * AR_LD + 4 * q) = v;
    }
    v4f rb[4];
    auto load_b = [&](int nb) {           // B block: 64 rows (n) x 16 float4
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int t = it * 256 + tid, n = t >> 4, q = t & 15;
            rb[it] = *reinterpret_cast<const v4f *>(B + ((size_t)nb * 64 + n) * AR_K + 4 * q);
        }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int t = it * 256 + tid;
            *reinterpret_cast<v4f *>(Bs + (buf * 64 + (t >> 4)) * AR_LD + 4 * (t & 15)) = rb[it];
        }
    };
    const int n_blocks = N / 64;
    load_b(0);
    store_b(0);
    __syncthreads();
    for (int nb = 0; nb < n_blocks; ++nb) {
        const int buf = nb & 1;
        if (nb + 1 < n_blocks) load_b(nb + 1);     // in flight under this block's MFMAs
        f32x16 acc[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
        const float *arow = As + (32 * wave + r31) * AR_LD + 16 * h;
        const float *brow = Bs + (buf * 64 + r31) * AR_LD + 16 * h;
#pragma unroll
        for (int kc = 0; kc < AR_K; kc += 32)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const v4f a4 = *reinterpret_cast<const v4f *>(arow + kc + 4 * q);
                const v4f b0 = *reinterpret_cast<const v4f *>(brow + kc + 4 * q);
                const v4f b1 = *reinterpret_cast<const v4f *>(brow + 32 * AR_LD + kc + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b0[e], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b1[e], acc[1], 0, 0, 0);
                }
            }
        if (nb + 1 < n_blocks) store_b(buf ^ 1);   // the other buffer was last read one barrier ago
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = nb * 64 + 32 * c + r31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < M) C[m * N + col] = acc[c][r];
            }
        }
        __syncthreads();
    }
}

// ---- C[i,o] = sum_m A[idx[m], i0+i] B[idx[m], o] / div : slab per (i-tile, split) ----------------
constexpr int BI = 128, BO = 64, BMK = 32;

__global__ __launch_bounds__(256) void gemm_atb_tiled_kernel(const float *__restrict__ A, int64_t lda,
                                                             const float *__restrict__ B, int64_t ldb,
                                                             const int32_t *__restrict__ idx, float div,
                                                             int64_t M, int KI, int NO, int64_t rows_per_split,
                                                             float *__restrict__ slabs)
{
    __shared__ __attribute__((aligned(16))) float As[BMK * BI];
    __shared__ __attribute__((aligned(16))) float Bs[BMK * BO];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r31 = lane & 31, h = lane >> 5;
    const int i0 = blockIdx.x * BI;
    const int o0 = blockIdx.z * BO;
    const int64_t ms = (int64_t)blockIdx.y * rows_per_split, me = min(M, ms + rows_per_split);
    const bool a_vec = (lda % 4 == 0) && ((uintptr_t)A % 16 == 0);
    const bool b_vec = (ldb % 4 == 0) && ((uintptr_t)B % 16 == 0);
    f32x16 acc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;

    v4f ra[4], rb[2];
    auto load_chunk = [&](int64_t mb) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {  // A chunk: 32 rows x 128 floats = 1024 v4f
            const int t = it * 256 + tid, mm = t >> 5, q = t & 31;
            const int64_t m = mb + mm;
            const float *row = nullptr;
            if (m < me) row = A + (idx ? (int64_t)idx[m] : m) * lda + i0 + 4 * q;
            ra[it] = load4(row, KI - (i0 + 4 * q), a_vec);
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {  // B chunk: 32 rows x 64 floats = 512 v4f
            const int t = it * 256 + tid, mm = t >> 4, q = t & 15;
            const int64_t m = mb + mm;
            const float *row = nullptr;
            if (m < me) row = B + (idx ? (int64_t)idx[m] : m) * ldb + o0 + 4 * q;
            rb[it] = load4(row, NO - (o0 + 4 * q), b_vec);
        }
    };
    if (ms < me) load_chunk(ms);
    for (int64_t mb = ms; mb < me; mb += BMK) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int t = it * 256 + tid;
            *reinterpret_cast<v4f *>(As + (t >> 5) * BI + 4 * (t & 31)) = ra[it];
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int t = it * 256 + tid;
            *reinterpret_cast<v4f *>(Bs + (t >> 4) * BO + 4 * (t & 15)) = rb[it];
        }
        __syncthreads();
        if (mb + BMK < me) load_chunk(mb + BMK);
        // MFMA step s consumes vertices 2s + h: A[i][k] = As[k][i], lanes along i -> conflict-free b32 reads
#pragma unroll
        for (int s = 0; s < BMK / 2; ++s) {
            const int k = 2 * s + h;
            const float a = As[k * BI + 32 * wave + r31];
            const float b0 = Bs[k * BO + r31], b1 = Bs[k * BO + 32 + r31];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
        }
    }
    float *slab = slabs + (size_t)blockIdx.y * KI * NO;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int o = o0 + 32 * c + r31;
        if (o >= NO) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (i < KI) slab[(size_t)i * NO + o] = div != 1.0f ? acc[c][r] / div : acc[c][r];
        }
    }
}

} // namespace

namespace amp {

int gemm_tiled(const TiledArgs &p)
{
    if (p.M <= 0 || p.N <= 0) return 0;
    if (p.K == AR_K && p.b_nk && p.N % 64 == 0 && p.N >= 512 && p.M >= 1024 && !p.a_idx && !p.c_idx && !p.bias &&
        p.act == ATHENA_MP_ACT_NONE && p.a_div == 1.0f && p.c_div == 1.0f && p.lda == AR_K && p.ldb == AR_K &&
        p.ldc == p.N && (uintptr_t)p.A % 16 == 0 && (uintptr_t)p.B % 16 == 0) {
        constexpr size_t lds = sizeof(float) * (size_t)(AR_BM + 128) * AR_LD;
        static amp::PerDeviceFlag attr;
        if (!attr.get()) {
            AMP_HIP(hipFuncSetAttribute((const void *)gemm_arow_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr.get() = true;
        }
        hipLaunchKernelGGL(gemm_arow_kernel, dim3((unsigned)((p.M + AR_BM - 1) / AR_BM)), dim3(256), lds, stream(), p.A, p.B,
                           p.C, p.M, p.N);
        AMP_LAUNCH_CHECK();
        return 0;
    }
    const int64_t ny = (p.N + BN - 1) / BN;
    // 256-row workgroup tiles once they still give >= 1.5 workgroups per CU; 128-row tiles otherwise
    if ((p.M + 255) / 256 * ny >= 384) {
        hipLaunchKernelGGL((gemm_tiled_kernel<64>), dim3((unsigned)((p.M + 255) / 256 * ny)), dim3(256), 0, stream(), p);
    } else {
        hipLaunchKernelGGL((gemm_tiled_kernel<32>), dim3((unsigned)((p.M + 127) / 128 * ny)), dim3(256), 0, stream(), p);
    }
    AMP_LAUNCH_CHECK();
    return 0;
}

int gemm_atb_tiled(const float *A, int64_t lda, const float *B, int64_t ldb, const int32_t *idx, float div, int64_t M,
                   int KI, int NO, float *C, bool accumulate)
{
    const int n = KI * NO;
    if (M <= 0) {
        if (!accumulate) AMP_HIP(hipMemsetAsync(C, 0, sizeof(float) * n, stream()));
        return 0;
    }
    const int n_it = (KI + BI - 1) / BI, n_ot = (NO + BO - 1) / BO;
    int splits = (int)std::max<int64_t>(1, std::min<int64_t>((M + 255) / 256, std::max(1, 1024 / (n_it * n_ot))));
    int64_t rps = (M + splits - 1) / splits;
    rps = (rps + 1) & ~(int64_t)1;
    splits = (int)((M + rps - 1) / rps);
    void *ws = nullptr;
    if (workspace(&ws, sizeof(float) * (size_t)splits * n, 6)) return 1;
    hipLaunchKernelGGL(gemm_atb_tiled_kernel, dim3(n_it, splits, n_ot), dim3(256), 0, stream(), A, lda, B, ldb, idx, div,
                       M, KI, NO, rps, (float *)ws);
    AMP_LAUNCH_CHECK();
    return slab_reduce((const float *)ws, splits, n, C, accumulate);
}

} // namespace amp
