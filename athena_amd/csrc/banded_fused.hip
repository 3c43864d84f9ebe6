// Kipf layer step on a BANDED graph (a block-diagonal batch of small graphs) in ONE launch at 64 -> 64 features:
//   P = A^ X   gathered from LDS as csr_gather_banded64 does (agg.hip: the rows a 128-row block can touch staged once, CSR-order sums
//              with the per-entry coefficient -- bit-identical to the general kernel and to the oracle),
//   Z = act(P . Wt + b)   on the 128 x 64 tile while it is still in LDS (W resident in LDS, v_mfma_f32_32x32x2_f32 as gemm.hip's
//              weight-resident kernel; the C tile leaves through the same LDS rows as whole 256-byte rows).
// update_message_kipf, athena_kipf_msgpass_layer.f90:943-952; the reverse to x, dX = (A^T dZ) . W, is the same launch over the
// transposed CSR with W read as [N][K].  Against agg.hip's banded gather followed by gemm.hip's dense step this saves P's trip out of
// and back into HBM (the forward still WRITES P when the caller keeps it for dW): 1.9 instead of 2.5 GB per forward launch at 130 k
// molecule-sized graphs, 1.3 instead of 2.4 GB per reverse launch.  agg_gemm_kernel (fused.hip) is the one-launch form for graphs
// whose rows come from all over HBM; on banded graphs its row chase loses to both (profiles/r06_kipf_banded_ab.txt).
#include <algorithm>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kRB = 128, kBandMax = 32, kEnt = 8, kF = 64, kLD = kF + 4;
constexpr int kXR = kRB + 2 * kBandMax;

template <int ACT> __device__ __forceinline__ float act_of(float z)
{
    if constexpr (ACT == ATHENA_MP_ACT_RELU) return z > 0.0f ? z : 0.0f;
    if constexpr (ACT == ATHENA_MP_ACT_SIGMOID) return 1.0f / (1.0f + expf(-z));
    if constexpr (ACT == ATHENA_MP_ACT_TANH) return tanhf(z);
    return z;
}

// COEF: entries carry a coefficient; ACT: epilogue activation; BIAS; WP: P rows are also written to HBM (the forward's tape)
template <bool COEF, int ACT, bool BIAS, bool WP>
__global__ __launch_bounds__(256, 2) void banded_agg_gemm64_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ idx,
                                                                   const float *__restrict__ coef, const float *__restrict__ x,
                                                                   const float *__restrict__ W, int b_nk, const float *__restrict__ bias,
                                                                   float *__restrict__ P, float *__restrict__ Z, int32_t n_rows,
                                                                   int32_t band, int32_t per)
{
    constexpr int NL = kXR * 16 / 256;            // 12 row slices of 16 bytes per thread
    constexpr int NE = kRB * kEnt / 256;          // 4 entries per thread
    __shared__ __attribute__((aligned(16))) float r0s[kXR * kF];   // the staged rows, then the P tile [128][kLD] / the waves' C tiles
    __shared__ __attribute__((aligned(16))) float Bs[kF * kLD];    // W as [n][k]
    __shared__ int32_t rp[kRB + 1], es[kRB * kEnt];
    __shared__ float cs[COEF ? kRB * kEnt : 1];
    static_assert(kRB * kLD <= kXR * kF, "the P tile overlays the staged rows");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r31 = lane & 31, h = lane >> 5;
    const int gl = threadIdx.x & 15, rl = threadIdx.x >> 4;

    // ---- W once per workgroup: Bs[n][k] = Wt[k][n] (forward, W stored [K][N]) or W[n][k] (reverse, stored [N][K]) ----------------------
    {
        v4f tmp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) tmp[i] = reinterpret_cast<const v4f *>(W)[i * 256 + threadIdx.x];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = i * 256 + threadIdx.x;
            if (b_nk) {
                const int n = t >> 4, q = t & 15;
                *reinterpret_cast<v4f *>(Bs + n * kLD + 4 * q) = tmp[i];
            } else {
                const int k = t >> 4, n4 = t & 15;
                Bs[(4 * n4 + 0) * kLD + k] = tmp[i].x;
                Bs[(4 * n4 + 1) * kLD + k] = tmp[i].y;
                Bs[(4 * n4 + 2) * kLD + k] = tmp[i].z;
                Bs[(4 * n4 + 3) * kLD + k] = tmp[i].w;
            }
        }
    }

    const v4f *x4g = reinterpret_cast<const v4f *>(x);
    v4f *r0s4 = reinterpret_cast<v4f *>(r0s);
    // virtual block vb -> chunk (vb & 7) * per + (vb >> 3): the workgroups of one XCD walk neighbouring chunks (gridDim.x is a multiple of 8).
    // The rows of the NEXT chunk are requested (into registers) before the dense step of the current one: their latency runs under it.
    v4f xv[NL];
    auto request_rows = [&](int vb_) {
        const int r0_ = ((vb_ & 7) * per + (vb_ >> 3)) * kRB;
        if (vb_ >= 8 * per || r0_ >= n_rows) return;
        const int r1_ = min(r0_ + kRB, n_rows), c0_ = max(0, r0_ - band), c1_ = min(n_rows, r1_ + band);
        const int nx_ = (c1_ - c0_) * 16;
        const v4f *x4 = x4g + (int64_t)c0_ * 16;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int t = threadIdx.x + 256 * i;
            xv[i] = t < nx_ ? x4[t] : (v4f){0.0f, 0.0f, 0.0f, 0.0f};
        }
    };
    // (a workgroup's chunks with r0 >= n_rows all lie at the end of its walk: chunk grows with vb for a fixed vb & 7)
    request_rows(blockIdx.x);
    for (int vb = blockIdx.x; vb < 8 * per; vb += gridDim.x) {
        const int chunk = (vb & 7) * per + (vb >> 3);
        const int r0 = chunk * kRB;
        if (r0 >= n_rows) break;                  // (uniform per workgroup)
        const int r1 = min(r0 + kRB, n_rows), c0 = max(0, r0 - band), c1 = min(n_rows, r1 + band);
        const int nx = (c1 - c0) * 16;
        __syncthreads();                           // the previous chunk's tiles are done with (and W is staged)
        if ((int)threadIdx.x <= r1 - r0) rp[threadIdx.x] = rowptr[r0 + threadIdx.x];
        __syncthreads();
        const int w0 = rp[0], nw = rp[r1 - r0] - w0;
        int32_t ev[NE];
        [[maybe_unused]] float cv[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int t = threadIdx.x + 256 * i;
            ev[i] = t < nw ? idx[w0 + t] : 0;
            if constexpr (COEF) cv[i] = t < nw ? coef[w0 + t] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int t = threadIdx.x + 256 * i;
            if (t < nx) r0s4[t] = xv[i];
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int t = threadIdx.x + 256 * i;
            if (t < nw) {
                es[t] = ev[i];
                if constexpr (COEF) cs[t] = cv[i];
            }
        }
        __syncthreads();
        // ---- the gather: 16 lanes own a row, entries in CSR order (a rounded multiply and a rounded add per entry) -------------------
        v4f acc[kRB / 16];
#pragma unroll
        for (int pass = 0; pass < kRB / 16; ++pass) {
            const int r = 16 * pass + rl;
            acc[pass] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
            if (r0 + r < r1) {
                const int a = rp[r] - w0, b = rp[r + 1] - w0;
                for (int w = a; w < b; ++w) {
                    const v4f v = r0s4[(es[w] - c0) * 16 + gl];
                    if constexpr (COEF) acc[pass] = acc[pass] + cs[w] * v;
                    else acc[pass] = acc[pass] + v;
                }
            }
        }
        __syncthreads();                           // nobody reads the staged rows any more: the region becomes the P tile
#pragma unroll
        for (int pass = 0; pass < kRB / 16; ++pass) {
            const int r = 16 * pass + rl;
            *reinterpret_cast<v4f *>(r0s + r * kLD + 4 * gl) = acc[pass];
            if constexpr (WP)
                if (r0 + r < r1) reinterpret_cast<v4f *>(P)[(int64_t)(r0 + r) * 16 + gl] = acc[pass];
        }
        request_rows(vb + gridDim.x);              // the next chunk's rows: in flight under the dense step
        __syncthreads();
        // ---- the dense step: wave w takes rows 32 w .. 32 w + 31 of the tile (gemm.hip's weight-resident loop) ---------------------
        float *Ws = r0s + wave * 32 * kLD;
        f32x16 c[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) c[t][r] = 0.0f;
        const float *arow = Ws + r31 * kLD + (kF / 2) * h;
        const float *brow = Bs + r31 * kLD + (kF / 2) * h;
#pragma unroll 2
        for (int q = 0; q < kF / 8; ++q) {
            const v4f a4 = *reinterpret_cast<const v4f *>(arow + 4 * q);
            v4f b4[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) b4[t] = *reinterpret_cast<const v4f *>(brow + t * 32 * kLD + 4 * q);
#pragma unroll
            for (int t = 0; t < 2; ++t) c[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4[t].x, c[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t) c[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4[t].y, c[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t) c[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4[t].z, c[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t) c[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4[t].w, c[t], 0, 0, 0);
        }
        // this wave's rows of the tile have been read: they receive its C tile (32x32 C/D map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 h)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) Ws[((r & 3) + 8 * (r >> 2) + 4 * h) * kLD + t * 32 + r31] = c[t][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int t = it * 64 + lane, row = t >> 4, q = t & 15;
            v4f v = *reinterpret_cast<const v4f *>(Ws + row * kLD + 4 * q);
            if constexpr (BIAS) v += *reinterpret_cast<const v4f *>(bias + 4 * q);
            v.x = act_of<ACT>(v.x); v.y = act_of<ACT>(v.y); v.z = act_of<ACT>(v.z); v.w = act_of<ACT>(v.w);
            const int gr = r0 + wave * 32 + row;
            if (gr < r1) reinterpret_cast<v4f *>(Z)[(int64_t)gr * 16 + q] = v;
        }
    }
}

} // namespace

namespace amp {

// 0 done, -1 not this kernel's shape (the caller takes the two-launch route), > 0 error.  The caller has checked that the graph is
// banded for this direction (kipf_gather_is_banded).
int banded_agg_gemm64(const athena_mp_graph *g, bool transposed, const float *coef, const float *x, const float *W, int b_nk,
                      const float *bias, int act, float *P, float *Z)
{
    if (act < 0 || act > ATHENA_MP_ACT_TANH || g->n_rows != g->n_cols) return -1;
    if (((uintptr_t)x | (uintptr_t)W | (uintptr_t)Z | (uintptr_t)P | (uintptr_t)bias) % 16) return -1;
    const int32_t *rowptr = transposed ? g->t_rowptr : g->rowptr, *idx = transposed ? g->t_src : g->col;
    const int chunks = (g->n_rows + kRB - 1) / kRB, per = (chunks + 7) / 8;
    const int grid = std::min(8 * per, 512);      // two workgroups per CU, a multiple of 8
#define AMP_BF(COEF_, ACT_, BIAS_, WP_)                                                                                       \
    hipLaunchKernelGGL((banded_agg_gemm64_kernel<COEF_, ACT_, BIAS_, WP_>), dim3(grid), dim3(256), 0, stream(), rowptr, idx, coef, x, \
                       W, b_nk, bias, P, Z, g->n_rows, g->band, per)
#define AMP_BF_ACT(COEF_, BIAS_, WP_)                                                                                         \
    switch (act) {                                                                                                            \
    case ATHENA_MP_ACT_RELU: AMP_BF(COEF_, ATHENA_MP_ACT_RELU, BIAS_, WP_); break;                                            \
    case ATHENA_MP_ACT_SIGMOID: AMP_BF(COEF_, ATHENA_MP_ACT_SIGMOID, BIAS_, WP_); break;                                      \
    case ATHENA_MP_ACT_TANH: AMP_BF(COEF_, ATHENA_MP_ACT_TANH, BIAS_, WP_); break;                                            \
    default: AMP_BF(COEF_, ATHENA_MP_ACT_NONE, BIAS_, WP_); break;                                                            \
    }
    if (!transposed) {             // forward: always the coefficient; bias optional; P optional
        if (!coef) return -1;
        if (bias && P) { AMP_BF_ACT(true, true, true) }
        else if (bias) { AMP_BF_ACT(true, true, false) }
        else if (P) { AMP_BF_ACT(true, false, true) }
        else { AMP_BF_ACT(true, false, false) }
    } else {                       // reverse to x: no epilogue, no P; coefficient only for the exact adjoint
        if (bias || P || act != ATHENA_MP_ACT_NONE) return -1;
        if (coef) AMP_BF(true, ATHENA_MP_ACT_NONE, false, false);
        else AMP_BF(false, ATHENA_MP_ACT_NONE, false, false);
    }
#undef AMP_BF_ACT
#undef AMP_BF
    AMP_LAUNCH_CHECK();
    return 0;
}

} // namespace amp
