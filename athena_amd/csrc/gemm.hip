// Dense contraction of the path on fp32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain).
//
// Replaces diffstruc's `matmul(params(t), ptr2)` and its reverse-mode products (call sites
// athena_kipf_msgpass_layer.f90:951, athena_duvenaud_msgpass_layer.f90:842, athena_graph_nop_layer.f90:761):
//   fwd : Z[M,N]  = A[M,K] . B[K,N]          A = P, B = Wt (row-major [Fi][Fo]), K=Fi, N=Fo
//   dx  : dP[M,N] = A[M,K] . B[K,N]          A = dZ, B[k=o][n=i] = Wt[i][o] (stored [N][K]), K=Fo, N=Fi
//   dw  : dWt[Fi,Fo] = sum_v P[v,:]^T dZ[v,:]
// M (vertices) is 10^6..10^7, K and N are 32..128: the weight matrix (<= 64 KB) stays resident in LDS
// for the lifetime of a persistent workgroup, each wave streams its own 32-row slabs of A through a
// private LDS region (no workgroup barrier in the main loop), and every byte of A and Z moves once.
//
// LDS images are [row][K+4] floats: the +4 pad makes the 16 B/lane fragment reads conflict free
// (MI355X_MICROARCH: ds_read_b128 is serviced in 16-lane groups over 64 banks; row pitch = 33 slots of
// 16 B puts the 16 rows of a group on 16 distinct slots).  The k index is permuted -- lane half h of
// MFMA step s consumes k = s + (K/2) h -- so four consecutive steps read one 16-byte run per lane.
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4))); // native vector: stays in VGPRs (HIP float4 struct copies became memcpy-to-scratch)

__device__ __forceinline__ float apply_act(float z, int act)
{
    switch (act) {
    case ATHENA_MP_ACT_RELU: return z > 0.0f ? z : 0.0f;
    case ATHENA_MP_ACT_SIGMOID: return 1.0f / (1.0f + expf(-z));
    case ATHENA_MP_ACT_TANH: return tanhf(z);
    default: return z;
    }
}

// ---------------------------------------------------------------------------------------------
// B-resident persistent GEMM.  b_nk: B is stored [N][K] (dx) instead of [K][N] (fwd).
// ---------------------------------------------------------------------------------------------
template <int ACT> __device__ __forceinline__ float act_ct(float z)
{
    if constexpr (ACT == ATHENA_MP_ACT_RELU) return z > 0.0f ? z : 0.0f;
    if constexpr (ACT == ATHENA_MP_ACT_SIGMOID) return 1.0f / (1.0f + expf(-z));
    if constexpr (ACT == ATHENA_MP_ACT_TANH) return tanhf(z);
    return z;
}

// Stage B into LDS as [n][K+4].  All global loads of a thread are issued before the first LDS write
// (a load->write loop serialised ~2 us of latency per iteration: 15 % of the kernel at 1M rows).
template <int K, int N, int THREADS>
__device__ __forceinline__ void stage_b(float *__restrict__ Bs, const float *__restrict__ B, int b_nk)
{
    constexpr int LD = K + 4;
    constexpr int NV = K * N / 4;                       // v4f in B
    constexpr int PER = (NV + THREADS - 1) / THREADS;   // v4f per thread
    v4f tmp[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        int t = i * THREADS + threadIdx.x;
        tmp[i] = reinterpret_cast<const v4f *>(B)[t < NV ? t : NV - 1];
    }
    if (b_nk) { // B stored [N][K]: straight copy of K-runs
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            int t = i * THREADS + threadIdx.x;
            if (t < NV) {
                int n = t / (K / 4), q = t - n * (K / 4);
                *reinterpret_cast<v4f *>(Bs + n * LD + 4 * q) = tmp[i];
            }
        }
    } else {    // B stored [K][N]: transpose while writing
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            int t = i * THREADS + threadIdx.x;
            if (t < NV) {
                int k = t / (N / 4), n4 = t - k * (N / 4);
                Bs[(4 * n4 + 0) * LD + k] = tmp[i].x;
                Bs[(4 * n4 + 1) * LD + k] = tmp[i].y;
                Bs[(4 * n4 + 2) * LD + k] = tmp[i].z;
                Bs[(4 * n4 + 3) * LD + k] = tmp[i].w;
            }
        }
    }
}

template <int K>
__device__ __forceinline__ void load_slab(v4f (&pre)[K / 8], const float *__restrict__ A, int64_t s, int64_t M,
                                          int lane)
{
    constexpr int RQ = K / 4;
    const int64_t r0 = s * 32;
#pragma unroll
    for (int it = 0; it < K / 8; ++it) {
        int t = it * 64 + lane;
        int row = t / RQ, q = t - row * RQ;
        int64_t gr = min(r0 + row, M - 1); // tail rows re-read the last row; never stored
        pre[it] = *reinterpret_cast<const v4f *>(A + gr * K + 4 * q);
    }
}

template <int K, int N, int ACT, bool BIAS>
__global__ __launch_bounds__(256, 1) void gemm_bres_kernel(const float *__restrict__ A,
                                                            const float *__restrict__ B, int b_nk,
                                                            const float *__restrict__ bias,
                                                            float *__restrict__ Z, int64_t M)
{
    constexpr int LD = K + 4;                 // floats per LDS row of the operand images
    constexpr int LDW = (K > N ? K : N) + 4;  // pitch of a wave's private region (A image, then C image)
    constexpr int NT = N / 32;                // 32-wide column tiles per wave
    constexpr int A4 = K / 8;                 // v4f per lane per 32-row slab of A
    constexpr int RQ = K / 4;                 // v4f per row of A
    constexpr int C4 = N / 8;                 // v4f per lane per 32-row slab of C
    constexpr int CQ = N / 4;                 // v4f per row of C
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Bs = lds;                                              // [N][LD]
    float *Ws = lds + N * LD + (threadIdx.x >> 6) * 32 * LDW;     // this wave's [32][LDW]

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r31 = lane & 31;
    const int h = lane >> 5;

    // ---- stage B once per workgroup --------------------------------------------------------
    stage_b<K, N, 256>(Bs, B, b_nk);
    __syncthreads();

    const int64_t n_slabs = (M + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t slab = (int64_t)blockIdx.x * 4 + wave;

    v4f pre[A4];
    if (slab < n_slabs) load_slab<K>(pre, A, slab, M, lane);

    for (; slab < n_slabs; slab += stride) {
        // registers -> this wave's LDS image of A
#pragma unroll
        for (int it = 0; it < A4; ++it) {
            int t = it * 64 + lane;
            int row = t / RQ, q = t - row * RQ;
            *reinterpret_cast<v4f *>(Ws + row * LDW + 4 * q) = pre[it];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // prefetch the next slab while the MFMAs run
        if (slab + stride < n_slabs) load_slab<K>(pre, A, slab + stride, M, lane);

        f32x16 acc[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;

        const float *arow = Ws + r31 * LDW + (K / 2) * h;
        const float *brow = Bs + r31 * LD + (K / 2) * h;
#pragma unroll 2
        for (int q = 0; q < K / 8; ++q) {
            v4f a4 = *reinterpret_cast<const v4f *>(arow + 4 * q);
            v4f b4[NT];
#pragma unroll
            for (int c = 0; c < NT; ++c) b4[c] = *reinterpret_cast<const v4f *>(brow + c * 32 * LD + 4 * q);
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4[c].x, acc[c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4[c].y, acc[c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4[c].z, acc[c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4[c].w, acc[c], 0, 0, 0);
        }
        // all LDS reads of the A image are done; the region now receives the C tile
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---- epilogue: C/D map of the 32x32 MFMA is col = lane&31, row = (r&3) + 8*(r>>2) + 4*h.
        // Transpose through LDS so that HBM sees whole 16 B/lane rows (2 full rows per instruction).
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Ws[((r & 3) + 8 * (r >> 2) + 4 * h) * LDW + c * 32 + r31] = acc[c][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int64_t r0 = slab * 32;
#pragma unroll
        for (int it = 0; it < C4; ++it) {
            int t = it * 64 + lane;
            int row = t / CQ, q = t - row * CQ;
            v4f v = *reinterpret_cast<const v4f *>(Ws + row * LDW + 4 * q);
            if constexpr (BIAS) {
                v4f bv = *reinterpret_cast<const v4f *>(bias + 4 * q);
                v += bv;
            }
            v.x = act_ct<ACT>(v.x); v.y = act_ct<ACT>(v.y); v.z = act_ct<ACT>(v.z); v.w = act_ct<ACT>(v.w);
            if (r0 + row < M) *reinterpret_cast<v4f *>(Z + (r0 + row) * N + 4 * q) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// 8-wave variant for K, N in {64,128}: two independent waves per SIMD, so one wave's LDS staging /
// epilogue overlaps the other's MFMAs.  To fit 8 private regions beside the resident B in 160 KB of
// LDS, A is staged in two K-halves and the C tile leaves in two N-halves through the same
// [32][max(K,N)/2 + 4] region.
// ---------------------------------------------------------------------------------------------
template <int K, int N, int ACT, bool BIAS>
__global__ __launch_bounds__(512, 2) void gemm_bres8_kernel(const float *__restrict__ A,
                                                             const float *__restrict__ B, int b_nk,
                                                             const float *__restrict__ bias,
                                                             float *__restrict__ Z, int64_t M)
{
    constexpr int KH = K / 2, NH = N / 2;
    constexpr int LD = K + 4;
    constexpr int LDW = (KH > NH ? KH : NH) + 4;
    constexpr int NT = N / 32, NTH = NT / 2;
    constexpr int A4 = K / 8, A4H = A4 / 2;   // v4f per lane per slab / per K-half
    constexpr int RQH = KH / 4;               // v4f per half row of A
    constexpr int C4H = NH / 8;               // v4f per lane per N-half of C
    constexpr int CQH = NH / 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Bs = lds;
    float *Ws = lds + N * LD + (threadIdx.x >> 6) * 32 * LDW;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r31 = lane & 31;
    const int h = lane >> 5;

    stage_b<K, N, 512>(Bs, B, b_nk);
    __syncthreads();

    const int64_t n_slabs = (M + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * 8;
    int64_t slab = (int64_t)blockIdx.x * 8 + wave;

    v4f pre[A4];
    auto gaddr = [&](int64_t s, int it) -> const v4f * {
        const int hk = it / A4H, j = it - hk * A4H;
        const int t = j * 64 + lane;
        const int row = t / RQH, q = t - row * RQH;
        const int64_t gr = min(s * 32 + row, M - 1);
        return reinterpret_cast<const v4f *>(A + gr * K + hk * KH + 4 * q);
    };
    if (slab < n_slabs) {
#pragma unroll
        for (int it = 0; it < A4; ++it) pre[it] = *gaddr(slab, it);
    }

    for (; slab < n_slabs; slab += stride) {
        f32x16 acc[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;

#pragma unroll
        for (int hk = 0; hk < 2; ++hk) {
#pragma unroll
            for (int j = 0; j < A4H; ++j) {
                const int t = j * 64 + lane;
                const int row = t / RQH, q = t - row * RQH;
                *reinterpret_cast<v4f *>(Ws + row * LDW + 4 * q) = pre[hk * A4H + j];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (hk == 1 && slab + stride < n_slabs) {
#pragma unroll
                for (int it = 0; it < A4; ++it) pre[it] = *gaddr(slab + stride, it);
            }
            const float *arow = Ws + r31 * LDW + (KH / 2) * h;
            const float *brow = Bs + r31 * LD + hk * KH + (KH / 2) * h;
#pragma unroll 2
            for (int q = 0; q < KH / 8; ++q) {
                v4f a4 = *reinterpret_cast<const v4f *>(arow + 4 * q);
                v4f b4[NT];
#pragma unroll
                for (int c = 0; c < NT; ++c) b4[c] = *reinterpret_cast<const v4f *>(brow + c * 32 * LD + 4 * q);
#pragma unroll
                for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4[c].x, acc[c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4[c].y, acc[c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4[c].z, acc[c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < NT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4[c].w, acc[c], 0, 0, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }

        const int64_t r0 = slab * 32;
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
#pragma unroll
            for (int c = 0; c < NTH; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Ws[((r & 3) + 8 * (r >> 2) + 4 * h) * LDW + c * 32 + r31] = acc[nh * NTH + c][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int it = 0; it < C4H; ++it) {
                const int t = it * 64 + lane;
                const int row = t / CQH, q = t - row * CQH;
                v4f v = *reinterpret_cast<const v4f *>(Ws + row * LDW + 4 * q);
                if constexpr (BIAS) v += *reinterpret_cast<const v4f *>(bias + nh * NH + 4 * q);
                v.x = act_ct<ACT>(v.x); v.y = act_ct<ACT>(v.y); v.z = act_ct<ACT>(v.z); v.w = act_ct<ACT>(v.w);
                if (r0 + row < M) *reinterpret_cast<v4f *>(Z + (r0 + row) * N + nh * NH + 4 * q) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int U, int TI, int TJ>
__device__ __forceinline__ void dw_mfma(f32x16 (&acc)[TI][TJ], const float (&a)[U][TI], const float (&b)[U][TJ])
{
#pragma unroll
    for (int s = 0; s < U; ++s)
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj)
                acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][ti], b[s][tj], acc[ti][tj], 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// dW, one WAVE per full [FI x FO] tile: every vertex row of P and dZ is read exactly once chip-wide
// with 16 B/lane loads (lane m holds features TI*m .. TI*m+TI-1: element t feeds MFMA row-tile t),
// TI*TJ MFMAs per vertex pair.  256 accumulator registers at 128x128, one wave per SIMD; the four
// waves of a workgroup add their tiles through LDS in wave order (deterministic), one slab per
// workgroup.
// ---------------------------------------------------------------------------------------------
template <int T> struct FragLoadV;
template <> struct FragLoadV<1> {
    static __device__ __forceinline__ void ld(float (&d)[1], const float *p) { d[0] = *p; }
};
template <> struct FragLoadV<2> {
    static __device__ __forceinline__ void ld(float (&d)[2], const float *p)
    {
        float2 t = *reinterpret_cast<const float2 *>(p);
        d[0] = t.x; d[1] = t.y;
    }
};
template <> struct FragLoadV<4> {
    static __device__ __forceinline__ void ld(float (&d)[4], const float *p)
    {
        v4f t = *reinterpret_cast<const v4f *>(p);
        d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
    }
};

template <bool MASK, int U, int TI, int TJ, int FI, int FO>
__device__ __forceinline__ void dwf_load(float (&a)[U][TI], float (&b)[U][TJ], const float *__restrict__ pa,
                                         const float *__restrict__ pb, int64_t v, int64_t v1, int h, int64_t ldp = FI,
                                         int64_t ldz = FO)
{
#pragma unroll
    for (int s = 0; s < U; ++s) {
        const int64_t vv = v + 2 * s + h;
        const bool ok = vv < v1;
        const int64_t vc = ok ? vv : v1 - 1;
        FragLoadV<TI>::ld(a[s], pa + vc * ldp);
        FragLoadV<TJ>::ld(b[s], pb + vc * ldz);
        if constexpr (MASK) {
            if (!ok) {
#pragma unroll
                for (int t = 0; t < TI; ++t) a[s][t] = 0.0f;
            }
        }
    }
}

// BLOCKED (widths that are multiples of 128 beyond 128: BASELINE configs[4]'s 256 x 256): dW is cut in nbi x nbo blocks of
// FI x FO = 128 x 128, one workgroup per (row range x, block y); P and dZ keep their own row pitch (ldp, ldz) and the block
// reads its 128 columns of each.  Workgroup id = x_lo + 8 (x_hi nb + y): the nb blocks of a row range sit on ONE XCD
// (workgroup b runs on XCD b % 8) next to each other in dispatch order, so each half row of P / dZ comes from HBM once.
template <int FI, int FO, bool BLOCKED = false>
__global__ __launch_bounds__(256, 1) void gemm_dw_full_kernel(const float *__restrict__ P,
                                                               const float *__restrict__ dZ,
                                                               float *__restrict__ slabs, int64_t M,
                                                               int64_t rows_per_wave, int64_t ldp = FI, int64_t ldz = FO,
                                                               int nbo = 1, int nb = 1)
{
    constexpr int TI = FI / 32, TJ = FO / 32;
    constexpr int U = 4;
    __shared__ __attribute__((aligned(16))) float red[FI * FO];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);   // in an SGPR for the epilogue (see there)
    const int m = lane & 31, h = lane >> 5;
    int64_t bx = blockIdx.x;          // the row range
    size_t slab_id = blockIdx.x;
    if constexpr (BLOCKED) {
        const int x_lo = blockIdx.x & 7, rest = blockIdx.x >> 3, y = rest % nb, x_hi = rest / nb;
        bx = x_hi * 8 + x_lo;
        P += (size_t)FI * (y / nbo);
        dZ += (size_t)FO * (y % nbo);
        slab_id = (size_t)y * (gridDim.x / nb) + bx;   // slabs of one block contiguous: [y][x]
    } else {
        ldp = FI; ldz = FO;
    }

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
        for (int b = 0; b < TJ; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const int64_t gw = bx * 4 + wave;
    const int64_t v0 = min(M, gw * rows_per_wave);
    const int64_t v1 = min(M, v0 + rows_per_wave);
    const float *pa = P + TI * m;
    const float *pb = dZ + TJ * m;

    if (v1 > v0) {
        float a0[U][TI], b0[U][TJ], a1[U][TI], b1[U][TJ];
        const int64_t n_pairs = (v1 - v0) / (4 * U);
        int64_t v = v0;
        if (n_pairs > 0) dwf_load<false, U, TI, TJ, FI, FO>(a0, b0, pa, pb, v, v1, h, ldp, ldz);
        for (int64_t p = 0; p < n_pairs; ++p, v += 4 * U) {
            dwf_load<false, U, TI, TJ, FI, FO>(a1, b1, pa, pb, v + 2 * U, v1, h, ldp, ldz);
            __builtin_amdgcn_sched_barrier(0);
            dw_mfma<U, TI, TJ>(acc, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            dwf_load<false, U, TI, TJ, FI, FO>(a0, b0, pa, pb, v + 4 * U, v1, h, ldp, ldz);
            __builtin_amdgcn_sched_barrier(0);
            dw_mfma<U, TI, TJ>(acc, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (; v < v1; v += 2 * U) {
            dwf_load<true, U, TI, TJ, FI, FO>(a0, b0, pa, pb, v, v1, h, ldp, ldz);
            dw_mfma<U, TI, TJ>(acc, a0, b0);
        }
    }
    // wave-ordered reduction of the four tiles; element (ti, r) of lane (m,h) is dWt[i][TJ*m .. +TJ-1]
    // with i = TI*((r&3) + 8*(r>>2) + 4*h) + ti  -> one TJ-wide vector store per (ti, r)
    // The lane index is derived AFRESH here (v_mbcnt through a volatile asm the compiler cannot merge with the value the main
    // loop used) and the wave index travels in an SGPR: at 128 x 128 every vector register is taken inside the loop, and a
    // thread index kept alive across it went to scratch (16 B per lane: scripts/isa_lint.py rule R1).
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int m_e = lane_e & 31, h_e = lane_e >> 5;
    for (int p = 0; p < 4; ++p) {
        if (wave_s == p) {
#pragma unroll
            for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = TI * ((r & 3) + 8 * (r >> 2) + 4 * h_e) + ti;
                    float *dst = red + i * FO + TJ * m_e;
#pragma unroll
                    for (int tj = 0; tj < TJ; ++tj) dst[tj] = (p == 0 ? 0.0f : dst[tj]) + acc[ti][tj][r];
                }
        }
        __syncthreads();
    }
    float *slab = slabs + slab_id * FI * FO;
    for (int t = wave_s * 64 + lane_e; t < FI * FO / 4; t += 256)
        reinterpret_cast<v4f *>(slab)[t] = reinterpret_cast<const v4f *>(red)[t];
}

// dW[(128 bi + i) Fo + 128 bo + o] (+)= blocks[bi nbo + bo][i][o]
__global__ void dw_blocks_place_kernel(const float *__restrict__ blocks, float *__restrict__ dW, int Fi, int Fo, int accumulate)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= Fi * Fo) return;
    const int i = t / Fo, o = t - i * Fo, nbo = Fo / 128;
    const float v = blocks[((size_t)((i >> 7) * nbo + (o >> 7)) * 128 + (i & 127)) * 128 + (o & 127)];
    dW[t] = accumulate ? dW[t] + v : v;
}

// out[seg][t] = sum_b slabs[first_seg + b][t]: b ascending inside each of 16 interleaved groups, the groups
// added in fixed order -- deterministic.  64 outputs x 16 slab groups per workgroup; blockIdx.y walks up
// to 16 independent slab ranges (the degree buckets of the Duvenaud weight gradient) in one launch.
struct SlabSegs {
    int first[16], count[16];
};
__global__ __launch_bounds__(1024) void slab_reduce_kernel(const float *__restrict__ slabs, SlabSegs segs, int n,
                                                           float *__restrict__ out, int64_t out_stride,
                                                           int accumulate)
{
    __shared__ float part[16][64];
    const int o = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + o;
    const int first = segs.first[blockIdx.y], count = segs.count[blockIdx.y];
    float s = 0.0f;
    if (t < n) {
        const float *src = slabs + (size_t)first * n + t;
#pragma unroll 4
        for (int b = grp; b < count; b += 16) s = s + src[(size_t)b * n];
    }
    part[grp][o] = s;
    __syncthreads();
    if (grp == 0 && t < n) {
        float r = part[0][o];
#pragma unroll
        for (int g = 1; g < 16; ++g) r = r + part[g][o];
        float *dst = out + (size_t)blockIdx.y * out_stride + t;
        *dst = accumulate ? *dst + r : r;
    }
}

// ---------------------------------------------------------------------------------------------
// Shape-generic VALU kernels (tiny / odd feature counts such as the 6-7-10 of msgpass_chemical).
// fmaf chain in k order == the MFMA's numerics.
// ---------------------------------------------------------------------------------------------
__global__ void gemm_small_kernel(const float *__restrict__ A, const float *__restrict__ B, int b_nk,
                                  const float *__restrict__ bias, int act, float *__restrict__ Z,
                                  int64_t M, int K, int N)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * N) return;
    int64_t v = t / N;
    int n = (int)(t - v * N);
    const float *a = A + v * K;
    float s = 0.0f;
    if (b_nk)
        for (int k = 0; k < K; ++k) s = fmaf(a[k], B[(size_t)n * K + k], s);
    else
        for (int k = 0; k < K; ++k) s = fmaf(a[k], B[(size_t)k * N + n], s);
    if (bias) s += bias[n];
    Z[t] = apply_act(s, act);
}

__global__ void gemm_dw_small_kernel(const float *__restrict__ P, const float *__restrict__ dZ,
                                     float *__restrict__ slabs, int64_t M, int FI, int FO,
                                     int64_t rows_per_block)
{
    const int64_t v0 = (int64_t)blockIdx.x * rows_per_block, v1 = min(M, v0 + rows_per_block);
    float *slab = slabs + (size_t)blockIdx.x * FI * FO;
    const int n = FI * FO;
    // few outputs (a bias gradient is FI = 1: the column sums of dZ): the block's 256 threads also split the ROWS, R row lanes
    // per output, combined through LDS in lane order (deterministic) -- one thread per output left 3/4 of the block idle and
    // walked its rows as one dependent chain (1.6 ms for the [2 M x 64] bias gradient of the configs[3] layer step)
    const int R = 2 * n <= (int)blockDim.x ? (int)blockDim.x / n : 1;
    if (R > 1) {
        __shared__ float part[256];
        const int r = threadIdx.x / n, t = threadIdx.x - r * n;
        float s = 0.0f;
        if (r < R) {
            const int i = t / FO, o = t - i * FO;
            float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;   // four rows in flight per thread
            int64_t v = v0 + r;
            for (; v + 3 * R < v1; v += 4 * R) {
                const float a0 = P[v * FI + i], a1 = P[(v + R) * FI + i], a2 = P[(v + 2 * R) * FI + i], a3 = P[(v + 3 * R) * FI + i];
                const float b0 = dZ[v * FO + o], b1 = dZ[(v + R) * FO + o], b2 = dZ[(v + 2 * R) * FO + o], b3 = dZ[(v + 3 * R) * FO + o];
                s = fmaf(a0, b0, s);
                s1 = fmaf(a1, b1, s1);
                s2 = fmaf(a2, b2, s2);
                s3 = fmaf(a3, b3, s3);
            }
            for (; v < v1; v += R) s = fmaf(P[v * FI + i], dZ[v * FO + o], s);
            s = (s + s1) + (s2 + s3);
        }
        part[threadIdx.x] = s;
        __syncthreads();
        if ((int)threadIdx.x < n) {
            float a = part[threadIdx.x];
            for (int q = 1; q < R; ++q) a += part[q * n + threadIdx.x];
            slab[threadIdx.x] = a;
        }
        return;
    }
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        int i = t / FO, o = t - i * FO;
        float s = 0.0f;
        for (int64_t v = v0; v < v1; ++v) s = fmaf(P[v * FI + i], dZ[v * FO + o], s);
        slab[t] = s;
    }
}

int num_cu() { return amp::num_cus(); }

template <int K, int N, int ACT, bool BIAS>
int launch_bres8(const float *A, const float *B, int b_nk, const float *bias, float *Z, int64_t M)
{
    constexpr int LDW = ((K > N ? K : N) / 2) + 4;
    constexpr size_t lds = sizeof(float) * ((size_t)N * (K + 4) + 8 * 32 * LDW);
    static amp::PerDeviceFlag attr_done;
    if (!attr_done.get()) {
        AMP_HIP(hipFuncSetAttribute((const void *)gemm_bres8_kernel<K, N, ACT, BIAS>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done.get() = true;
    }
    int64_t n_slabs = (M + 31) / 32;
    int grid = (int)std::min<int64_t>((n_slabs + 7) / 8, num_cu());
    hipLaunchKernelGGL((gemm_bres8_kernel<K, N, ACT, BIAS>), dim3(grid), dim3(512), lds, amp::stream(), A, B, b_nk,
                       bias, Z, M);
    AMP_LAUNCH_CHECK();
    return 0;
}

template <int K, int N, int ACT, bool BIAS>
int launch_bres2(const float *A, const float *B, int b_nk, const float *bias, float *Z, int64_t M)
{
    if constexpr (K >= 64 && N >= 64) {
        return launch_bres8<K, N, ACT, BIAS>(A, B, b_nk, bias, Z, M);   // eight waves: 0.72 against 0.55 of fp32 MFMA peak with four
    }
    constexpr size_t lds = sizeof(float) * ((size_t)N * (K + 4) + 4 * 32 * ((K > N ? K : N) + 4));
    static amp::PerDeviceFlag attr_done;
    if (!attr_done.get()) {
        AMP_HIP(hipFuncSetAttribute((const void *)gemm_bres_kernel<K, N, ACT, BIAS>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done.get() = true;
    }
    int64_t n_slabs = (M + 31) / 32;
    int grid = (int)std::min<int64_t>((n_slabs + 3) / 4, num_cu());
    hipLaunchKernelGGL((gemm_bres_kernel<K, N, ACT, BIAS>), dim3(grid), dim3(256), lds, amp::stream(), A, B, b_nk,
                       bias, Z, M);
    AMP_LAUNCH_CHECK();
    return 0;
}

template <int K, int N>
int launch_bres(const float *A, const float *B, int b_nk, const float *bias, int act, float *Z, int64_t M)
{
    if (!bias && act == ATHENA_MP_ACT_NONE) return launch_bres2<K, N, ATHENA_MP_ACT_NONE, false>(A, B, b_nk, bias, Z, M);
    if (!bias) {   // bias-free activations reuse the BIAS=true kernels
        float *zero_bias = nullptr;
        if (amp::named_buffer("gemm.zero_bias", sizeof(float) * 256, true, (void **)&zero_bias)) return 1;
        bias = zero_bias;
    }
    switch (act) {
    case ATHENA_MP_ACT_RELU: return launch_bres2<K, N, ATHENA_MP_ACT_RELU, true>(A, B, b_nk, bias, Z, M);
    case ATHENA_MP_ACT_SIGMOID: return launch_bres2<K, N, ATHENA_MP_ACT_SIGMOID, true>(A, B, b_nk, bias, Z, M);
    case ATHENA_MP_ACT_TANH: return launch_bres2<K, N, ATHENA_MP_ACT_TANH, true>(A, B, b_nk, bias, Z, M);
    default: return launch_bres2<K, N, ATHENA_MP_ACT_NONE, true>(A, B, b_nk, bias, Z, M);
    }
}

bool mfma_shape(int K, int N) { return (K == 32 || K == 64 || K == 128) && (N == 32 || N == 64 || N == 128); }

} // namespace

namespace amp {
int gemm_dispatch(const float *A, const float *B, int b_nk, const float *bias, int act, float *Z, int64_t M,
                  int K, int N)
{
    if (M == 0) return 0;
    if (mfma_shape(K, N) && ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)Z % 16 == 0) &&
        (!bias || (uintptr_t)bias % 16 == 0)) {
#define AMP_CASE(KK, NN) \
    if (K == KK && N == NN) return launch_bres<KK, NN>(A, B, b_nk, bias, act, Z, M);
        AMP_CASE(32, 32) AMP_CASE(32, 64) AMP_CASE(32, 128)
        AMP_CASE(64, 32) AMP_CASE(64, 64) AMP_CASE(64, 128)
        AMP_CASE(128, 32) AMP_CASE(128, 64) AMP_CASE(128, 128)
#undef AMP_CASE
    }
    if (K >= 8 && N >= 8 && M >= 128) { // anything MFMA-worthy that is not weight-resident
        TiledArgs t;
        t.A = A; t.lda = K; t.B = B; t.ldb = b_nk ? K : N; t.b_nk = b_nk; t.bias = bias; t.act = act;
        t.C = Z; t.ldc = N; t.M = M; t.N = N; t.K = K;
        return gemm_tiled(t);
    }
    int64_t total = M * N;
    hipLaunchKernelGGL(gemm_small_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, amp::stream(), A,
                       B, b_nk, bias, act, Z, M, K, N);
    AMP_LAUNCH_CHECK();
    return 0;
}
} // namespace amp

namespace amp {
int slab_reduce_segs(const float *slabs, int n, int n_segs, const int *first, const int *count, float *out,
                     int64_t out_stride, bool accumulate)
{
    for (int s0 = 0; s0 < n_segs; s0 += 16) {
        SlabSegs segs;
        const int ns = std::min(16, n_segs - s0);
        for (int i = 0; i < 16; ++i) segs.first[i] = i < ns ? first[s0 + i] : 0, segs.count[i] = i < ns ? count[s0 + i] : 0;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((n + 63) / 64, ns), dim3(1024), 0, stream(), slabs, segs, n,
                           out + (size_t)s0 * out_stride, out_stride, accumulate ? 1 : 0);
        AMP_LAUNCH_CHECK();
    }
    return 0;
}

int slab_reduce(const float *slabs, int n_slabs, int n, float *out, bool accumulate)
{
    const int first = 0;
    return slab_reduce_segs(slabs, n, 1, &first, &n_slabs, out, 0, accumulate);
}

int gemm_dw_dispatch(int64_t N, int Fi, int Fo, const float *P, const float *dZ, float *dW, bool accumulate)
{
    const int n = Fi * Fo;
    if (N == 0) {
        if (!accumulate) AMP_HIP(hipMemsetAsync(dW, 0, sizeof(float) * n, stream()));
        return 0;
    }
    bool mf = (Fi == 64 || Fi == 128) && (Fo == 64 || Fo == 128) && ((uintptr_t)P % 16 == 0) &&
              ((uintptr_t)dZ % 16 == 0);
    if (!mf && Fi % 128 == 0 && Fo % 128 == 0 && Fi <= 512 && Fo <= 512 && N >= 4096 &&
        (uintptr_t)P % 16 == 0 && (uintptr_t)dZ % 16 == 0) {
        // 128 x 128 blocks of dW on the register-resident kernel (BASELINE configs[4]: 256 x 256 = four blocks)
        const int nbi = Fi / 128, nbo = Fo / 128, nb = nbi * nbo;
        int nx = (int)std::min<int64_t>((N + 127) / 128, num_cu());
        int64_t rpw = (N + (int64_t)nx * 4 - 1) / ((int64_t)nx * 4);
        rpw = (rpw + 1) & ~(int64_t)1;
        nx = (int)((N + rpw * 4 - 1) / (rpw * 4));
        nx = (nx + 7) & ~7;                       // whole groups of eight row ranges (the workgroup order above)
        void *ws = nullptr, *blk = nullptr;
        if (workspace(&ws, sizeof(float) * (size_t)nx * nb * 16384, 2) || workspace(&blk, sizeof(float) * (size_t)nb * 16384, 10))
            return 1;
        hipLaunchKernelGGL((gemm_dw_full_kernel<128, 128, true>), dim3(nx * nb), dim3(256), 0, stream(), P, dZ, (float *)ws, N, rpw,
                           (int64_t)Fi, (int64_t)Fo, nbo, nb);
        AMP_LAUNCH_CHECK();
        std::vector<int> first(nb), count(nb, nx);
        for (int y = 0; y < nb; ++y) first[y] = y * nx;
        if (int rc = slab_reduce_segs((const float *)ws, 16384, nb, first.data(), count.data(), (float *)blk, 16384, false)) return rc;
        hipLaunchKernelGGL(dw_blocks_place_kernel, dim3((n + 255) / 256), dim3(256), 0, stream(), (const float *)blk, dW, Fi, Fo,
                           accumulate ? 1 : 0);
        AMP_LAUNCH_CHECK();
        return 0;
    }
    if (!mf && (int64_t)Fi * Fo >= 256 && N >= 256)
        return gemm_atb_tiled(P, Fi, dZ, Fo, nullptr, 1.0f, N, Fi, Fo, dW, accumulate);
    int nblk;
    int64_t rpb = 0, rpw = 0;
    if (mf) {
        nblk = (int)std::min<int64_t>((N + 127) / 128, num_cu());
        rpw = (N + (int64_t)nblk * 4 - 1) / ((int64_t)nblk * 4);
        rpw = (rpw + 1) & ~(int64_t)1;
        nblk = (int)((N + rpw * 4 - 1) / (rpw * 4));
    } else {
        nblk = (int)std::min<int64_t>((N + 255) / 256, (2 * n <= 256 ? 8 : 2) * num_cu());   // few outputs: small slabs, more of them
        rpb = (N + nblk - 1) / nblk;
        rpb = (rpb + 1) & ~(int64_t)1; // even: the MFMA form consumes vertex pairs
        nblk = (int)((N + rpb - 1) / rpb);
    }
    void *ws = nullptr;
    if (workspace(&ws, sizeof(float) * (size_t)nblk * n, 2)) return 1;
    float *slabs = (float *)ws;
    if (mf) {
#define AMP_CASE(A_, B_)                                                                                    \
    if (Fi == A_ && Fo == B_)                                                                               \
        hipLaunchKernelGGL((gemm_dw_full_kernel<A_, B_>), dim3(nblk), dim3(256), 0, stream(), P, dZ, slabs, N, rpw);
        AMP_CASE(64, 64) AMP_CASE(64, 128) AMP_CASE(128, 64) AMP_CASE(128, 128)
#undef AMP_CASE
    } else {
        hipLaunchKernelGGL(gemm_dw_small_kernel, dim3(nblk), dim3(256), 0, stream(), P, dZ, slabs, N, Fi, Fo, rpb);
    }
    AMP_LAUNCH_CHECK();
    return slab_reduce(slabs, nblk, n, dW, accumulate);
}
} // namespace amp

using namespace amp;

extern "C" {

int athena_mp_gemm_fwd(int64_t N, int32_t Fi, int32_t Fo, const float *P, const float *W, const float *bias,
                       int32_t act, float *Z)
{
    AMP_REQUIRE(N >= 0 && Fi > 0 && Fo > 0 && P && W && Z, "gemm_fwd: bad arguments");
    return gemm_dispatch(P, W, /*b_nk=*/0, bias, act, Z, N, Fi, Fo);
}

int athena_mp_gemm_dx(int64_t N, int32_t Fi, int32_t Fo, const float *dZ, const float *W, float *dP)
{
    AMP_REQUIRE(N >= 0 && Fi > 0 && Fo > 0 && dZ && W && dP, "gemm_dx: bad arguments");
    // dP[v,i] = sum_o dZ[v,o] Wt[i,o]:  K = Fo, N = Fi, B stored [N][K]
    return gemm_dispatch(dZ, W, /*b_nk=*/1, nullptr, ATHENA_MP_ACT_NONE, dP, N, Fo, Fi);
}

int athena_mp_gemm_dw(int64_t N, int32_t Fi, int32_t Fo, const float *P, const float *dZ, float *dW)
{
    AMP_REQUIRE(N >= 0 && Fi > 0 && Fo > 0 && P && dZ && dW, "gemm_dw: bad arguments");
    return gemm_dw_dispatch(N, Fi, Fo, P, dZ, dW, false);
}

} // extern "C"
