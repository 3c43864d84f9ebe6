// Degree-bucketed Duvenaud update on the matrix cores with the weights held in REGISTERS.
//
//   duvenaud_update                        athena_diffstruc_extd_sub_duvenaud.f90:176-228
//   get_partial_duvenaud_update_val        :284-324
//   get_partial_duvenaud_update_weight_val :326-368
//
// The graph handle keeps the vertices sorted by bucket and cut into 16-vertex tiles
// (amp::duvenaud_buckets).  Every 16x16x4 MFMA takes the 16 vertices of a tile on its COLUMN axis:
//   c^T[o, v] = sum_k W_d[o, k] a[v, k]       A operand = W_d fragment (registers, per bucket)
//                                             B operand = a[v, 16j + 4q .. +3], one 16 B load per lane
// so vertex rows are read and written 16 B per lane straight from/to HBM (no LDS staging of the
// streamed operand), and the accumulator of output tile `ot` is c[v, 16 ot + 4q .. +3] -- the same
// lane layout, ready for a 16 B store.  A wave owns a contiguous run of tiles and reloads the weight
// fragments (from L2) only when the bucket changes.
//
// The weight gradient contracts over VERTICES, so both streamed operands are turned once through a
// wave-private LDS tile; per-bucket partial sums go to slab (workgroup + bucket) -- a contiguous slab
// range per bucket, reduced in fixed order (deterministic, no atomics).
#include <algorithm>

#include "common.h"

#ifndef DUV_VARIANT
#define DUV_VARIANT 0   // 1 / 2: timing-only diagnostic builds (scripts/build_variants.sh), never shipped
#endif

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

// Epilogue arithmetic is kept off the critical path of the matrix pipe: the sigmoid uses the hardware
// exp2 / rcp (each 1 ulp; the result is within ~4e-7 relative of the libm form, inside the 1e-5 bound the
// MFMA route is tested to), and x/d for the bucket divisor d uses one Newton correction on x*(1/d):
// 3 instructions instead of the ~10 of the IEEE division sequence, equal to the IEEE quotient in all but
// ~4 of 1e5 operands and 1 ulp off otherwise (checked exhaustively-by-sampling for d = 1..64).
__device__ __forceinline__ float act_apply(float t, int act)
{
    switch (act) {
    case ATHENA_MP_ACT_RELU: return t > 0.0f ? t : 0.0f;
    case ATHENA_MP_ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t * -1.4426950408889634f));
    case ATHENA_MP_ACT_TANH: return tanhf(t);
    default: return t;
    }
}
__device__ __forceinline__ float div_by(float x, float d, float inv)
{
    const float q = x * inv;
    return fmaf(fmaf(-q, d, x), inv, q);
}

// Y[v, 0:NO] = f( sum_k Wd(o,k) X[v,k] ),  Wd(o,k) = W[b*wb + o*so + k*sk]
//   div_in  = 1: X is divided by d = b+1 before the product (forward: a/d, the reference's order)
//   div_in  = 0: the sum is divided by d afterwards (reverse w.r.t. a)
// Work split: every bucket gets a share of the waves (workgroups for the weight gradient) proportional to
// its tile count, and the waves of a bucket stride through that bucket's tiles.  All buckets therefore
// sweep the vertex array front to back at the same relative pace: the rows a DRAM page holds are asked for
// by the different buckets at about the same time (a bucket-after-bucket schedule touches every page once
// per bucket, 288 B at a time), and a wave never changes its weight fragments.
constexpr int kMaxBuckets = 32;
struct BucketSplit {
    int unit_off[kMaxBuckets + 1];   // first wave / workgroup of each bucket
    int tile_off[kMaxBuckets + 1];   // first tile of each bucket
    int n_buckets;
};

template <int KJ, int OT>
__global__ __launch_bounds__(256) void duv_rows_kernel(BucketSplit sp, const int32_t *__restrict__ trows,
                                                       const float *__restrict__ X, int K,
                                                       const float *__restrict__ W, int64_t wb, int so, int sk,
                                                       float *__restrict__ Y, int NO, int div_in, int act)
{
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= sp.unit_off[sp.n_buckets]) return;
    int b = 0;
    while (gw >= sp.unit_off[b + 1]) ++b;
    const int nw = sp.unit_off[b + 1] - sp.unit_off[b];
    const int t0 = sp.tile_off[b] + (gw - sp.unit_off[b]), t1 = sp.tile_off[b + 1];
    if (t0 >= t1) return;

    const float d = (float)(b + 1), inv = 1.0f / d;   // the bucket index is the divisor
    const bool pow2 = ((b + 1) & b) == 0;
    float Wf[OT][KJ][4];
    {
        const float *wd = W + (int64_t)b * wb;
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
#pragma unroll
            for (int j = 0; j < KJ; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int o = 16 * ot + n, k = 16 * j + 4 * q + c;
                    Wf[ot][j][c] = (o < NO && k < K) ? wd[(int64_t)o * so + (int64_t)k * sk] : 0.0f;
                }
    }

    // Three-stage software pipeline, rotated by unrolling (never by register moves, which would wait on
    // the load they move): while tile t is on the matrix cores, the rows of t+W and t+2W are in flight
    // and the vertex ids of t+3W are being fetched.  Issue order inside a step is ids first, rows second,
    // so waiting for an id never waits for the rows issued after it (vmcnt counts in order).
    struct Stage {
        v4f x[KJ];
        int r;
    };
    Stage S0, S1, S2;
    auto issue_r = [&](Stage &s, int t) {
        if (t < t1) s.r = trows[(int64_t)t * 16 + n];
    };
    auto issue_x = [&](Stage &s, int t) {
#if DUV_VARIANT == 2 || DUV_VARIANT == 3   // timing-only: no row loads
        if (t < t1) {
#pragma unroll
            for (int j = 0; j < KJ; ++j) s.x[j] = v4f{1.0f, 2.0f, 3.0f, (float)s.r};
            return;
        }
#endif
        if (t < t1) {
            const float *src = X + (int64_t)(s.r < 0 ? ~s.r : s.r) * K + 4 * q;
#pragma unroll
            for (int j = 0; j < KJ; ++j)
                s.x[j] = 16 * j + 4 * q < K ? *reinterpret_cast<const v4f *>(src + 16 * j) : v4f{0.0f, 0.0f, 0.0f, 0.0f};
        }
    };
    v4f keep = {0.0f, 0.0f, 0.0f, 0.0f};
    auto step = [&](Stage &cur, Stage &fill, int t) {
        // padding slots repeat the tile's first vertex: same loads, same arithmetic, same address, same value --
        // their stores are benign duplicates, so no store sits under a per-lane branch
        const int row = cur.r < 0 ? ~cur.r : cur.r;
        issue_r(cur, t + 3 * nw);     // cur's id slot is free (row decoded above); it is refilled first ...
        issue_x(fill, t + 2 * nw);    // ... then the rows of the tile two steps ahead
        v4f xf[KJ];
#pragma unroll
        for (int j = 0; j < KJ; ++j) xf[j] = cur.x[j];
        if (div_in) {
            if (pow2) {
#pragma unroll
                for (int j = 0; j < KJ; ++j) xf[j] = xf[j] * inv;   // exact for d = 1, 2, 4, 8, ...
            } else {
#pragma unroll
                for (int j = 0; j < KJ; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) xf[j][c] = div_by(xf[j][c], d, inv);
            }
        }
        float *dst = Y + (int64_t)row * NO + 4 * q;
        // OT independent accumulation chains, interleaved so consecutive MFMAs never wait on each other
        v4f accs[OT];
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) accs[ot] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
#if DUV_VARIANT == 1 || DUV_VARIANT == 4   // timing-only: no matrix work
#pragma unroll
        for (int j = 0; j < KJ; ++j)
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) accs[ot] = accs[ot] + xf[j] * Wf[ot][j][0];
#else
#pragma unroll
        for (int j = 0; j < KJ; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int ot = 0; ot < OT; ++ot)
                    accs[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wf[ot][j][c], xf[j][c], accs[ot], 0, 0, 0);
#endif
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
            v4f acc = accs[ot];
            if (!div_in) {
                if (pow2) acc = acc * inv;
                else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = div_by(acc[c], d, inv);
                }
            }
            if (act != ATHENA_MP_ACT_NONE) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = act_apply(acc[c], act);
            }
#if DUV_VARIANT == 3 || DUV_VARIANT == 4   // timing-only: no per-tile stores
            keep = keep + acc;
#else
            if (16 * ot + 4 * q < NO) *reinterpret_cast<v4f *>(dst + 16 * ot) = acc;
#endif
        }
    };
    S0.r = S1.r = S2.r = 0;
    issue_r(S0, t0);
    issue_r(S1, t0 + nw);
    issue_r(S2, t0 + 2 * nw);
    issue_x(S0, t0);
    issue_x(S1, t0 + nw);
    for (int t = t0; t < t1; t += 3 * nw) {
        step(S0, S2, t);
        if (t + nw < t1) step(S1, S0, t + nw);
        if (t + 2 * nw < t1) step(S2, S1, t + 2 * nw);
    }
#if DUV_VARIANT == 3 || DUV_VARIANT == 4
    if (keep[0] + keep[1] + keep[2] + keep[3] == 12345.678f) Y[gw] = keep[0];
#endif
}

// slab[workgroup][i*Fo + o] = sum over the workgroup's tiles (all of one bucket b) of (a[v,i]/d) g[v,o]
template <int IT, int OT>
__global__ __launch_bounds__(256) void duv_dw_kernel(BucketSplit sp, const int32_t *__restrict__ trows,
                                                     const float *__restrict__ A, int Fi,
                                                     const float *__restrict__ G, int Fo, float *__restrict__ slabs)
{
    constexpr int AP = 16 * IT + 4, GP = 16 * OT + 4, FOP = 16 * OT;
    constexpr int kTurn = 4 * 16 * (AP + GP), kRed = 16 * IT * FOP;
    __shared__ __attribute__((aligned(16))) float buf[kTurn > kRed ? kTurn : kRed];
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    float *al = buf + wave * 16 * (AP + GP), *gl = al + 16 * AP;
    int b = 0;
    while ((int)blockIdx.x >= sp.unit_off[b + 1]) ++b;
    const int nwg = sp.unit_off[b + 1] - sp.unit_off[b];
    const int stride = 4 * nwg;
    const int t0 = sp.tile_off[b] + 4 * ((int)blockIdx.x - sp.unit_off[b]) + wave, t1 = sp.tile_off[b + 1];
    const float d = (float)(b + 1), inv = 1.0f / d;
    const bool pow2 = ((b + 1) & b) == 0;

    v4f af[IT], gf[OT], an[IT], gn[OT];
    auto load = [&](v4f(&ad)[IT], v4f(&gd)[OT], int rr) {
        const bool ok = rr >= 0;
        const int r = ok ? rr : ~rr;
        const float *pa = A + (int64_t)r * Fi + 4 * q;
        const float *pg = G + (int64_t)r * Fo + 4 * q;
        const v4f zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < IT; ++j) ad[j] = 16 * j + 4 * q < Fi ? *reinterpret_cast<const v4f *>(pa + 16 * j) : zero;
#pragma unroll
        for (int j = 0; j < OT; ++j)
            gd[j] = (ok && 16 * j + 4 * q < Fo) ? *reinterpret_cast<const v4f *>(pg + 16 * j) : zero;
    };

    v4f acc[IT][OT];
#pragma unroll
    for (int i = 0; i < IT; ++i)
#pragma unroll
        for (int o = 0; o < OT; ++o) acc[i][o] = v4f{0.0f, 0.0f, 0.0f, 0.0f};

    int r_n = t0 < t1 ? trows[(int64_t)t0 * 16 + n] : 0;
    int r_nn = t0 + stride < t1 ? trows[(int64_t)(t0 + stride) * 16 + n] : 0;
    if (t0 < t1) load(an, gn, r_n);
    for (int t = t0; t < t1; t += stride) {
#pragma unroll
        for (int j = 0; j < IT; ++j) af[j] = an[j];
#pragma unroll
        for (int j = 0; j < OT; ++j) gf[j] = gn[j];
        r_n = r_nn;
        if (t + stride < t1) load(an, gn, r_n);
        if (t + 2 * stride < t1) r_nn = trows[(int64_t)(t + 2 * stride) * 16 + n];
        if (pow2) {
#pragma unroll
            for (int j = 0; j < IT; ++j) af[j] = af[j] * inv;
        } else {
#pragma unroll
            for (int j = 0; j < IT; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) af[j][c] = div_by(af[j][c], d, inv);
        }
#pragma unroll
        for (int j = 0; j < IT; ++j) *reinterpret_cast<v4f *>(al + n * AP + 16 * j + 4 * q) = af[j];
#pragma unroll
        for (int j = 0; j < OT; ++j) *reinterpret_cast<v4f *>(gl + n * GP + 16 * j + 4 * q) = gf[j];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float aa[IT], bb[OT];
#pragma unroll
            for (int i = 0; i < IT; ++i) aa[i] = al[(4 * q + r) * AP + 16 * i + n];
#pragma unroll
            for (int o = 0; o < OT; ++o) bb[o] = gl[(4 * q + r) * GP + 16 * o + n];
#pragma unroll
            for (int i = 0; i < IT; ++i)
#pragma unroll
                for (int o = 0; o < OT; ++o)
                    acc[i][o] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[i], bb[o], acc[i][o], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // acc[i][o][r] = dW(i = 16 i + 4q + r, o = 16 o + n); waves added in fixed order
    __syncthreads();
    for (int p = 0; p < 4; ++p) {
        if (wave == p) {
#pragma unroll
            for (int i = 0; i < IT; ++i)
#pragma unroll
                for (int o = 0; o < OT; ++o)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float *dst = buf + (16 * i + 4 * q + r) * FOP + 16 * o + n;
                        *dst = (p == 0 ? 0.0f : *dst) + acc[i][o][r];
                    }
        }
        __syncthreads();
    }
    float *slab = slabs + (size_t)blockIdx.x * Fi * Fo;
    for (int t = threadIdx.x; t < Fi * Fo; t += 256) {
        const int i = t / Fo, o = t - i * Fo;
        slab[t] = buf[i * FOP + o];
    }
}

// proportional split of `units` waves / workgroups over the buckets (every non-empty bucket gets >= 1)
BucketSplit make_split(const athena_mp_graph *g, int units, int tiles_per_unit_step)
{
    BucketSplit sp;
    const int nb = (int)g->btile_off.size() - 1, nt = g->n_btiles;
    sp.n_buckets = nb;
    sp.unit_off[0] = 0;
    for (int b = 0; b < nb; ++b) {
        const int tb = g->btile_off[b + 1] - g->btile_off[b];
        int u = 0;
        if (tb > 0) {
            const int steps = (tb + tiles_per_unit_step - 1) / tiles_per_unit_step;   // units that can be kept busy
            u = (int)std::min<int64_t>(steps, std::max<int64_t>(1, ((int64_t)units * tb + nt / 2) / nt));
        }
        sp.unit_off[b + 1] = sp.unit_off[b] + u;
        sp.tile_off[b] = g->btile_off[b];
    }
    sp.tile_off[nb] = g->btile_off[nb];
    for (int b = nb + 1; b <= kMaxBuckets; ++b) sp.unit_off[b] = sp.unit_off[nb], sp.tile_off[b] = sp.tile_off[nb];
    return sp;
}

inline int ceil16(int x) { return (x + 15) / 16; }
inline bool frag_shape(int kj, int ot) { return kj >= 1 && ot >= 1 && kj <= 6 && ot <= 6 && kj * ot <= 24; }

int launch_rows(const athena_mp_graph *g, const float *X, int K, const float *W, int64_t wb, int so, int sk, float *Y,
                int NO, int div_in, int act)
{
    const int kj = ceil16(K), ot = ceil16(NO);
    if ((K & 3) || (NO & 3) || !frag_shape(kj, ot)) return -1;
    const int nt = g->n_btiles;
    if (nt == 0) return 0;
    if ((int)g->btile_off.size() - 1 > kMaxBuckets) return -1;
    const BucketSplit sp = make_split(g, 256 * 4 * 2, 1);   // two resident waves per SIMD at ~220 VGPRs
    const dim3 grid((sp.unit_off[sp.n_buckets] + 3) / 4);
#define AMP_CASE(KJ_, OT_)                                                                                         \
    if (kj == KJ_ && ot == OT_) {                                                                                  \
        hipLaunchKernelGGL((duv_rows_kernel<KJ_, OT_>), grid, dim3(256), 0, amp::stream(), sp, g->btile_rows, X,   \
                           K, W, wb, so, sk, Y, NO, div_in, act);                                                  \
    }
#define AMP_ROW(KJ_) AMP_CASE(KJ_, 1) AMP_CASE(KJ_, 2) AMP_CASE(KJ_, 3) AMP_CASE(KJ_, 4)
    AMP_ROW(1) AMP_ROW(2) AMP_ROW(3) AMP_ROW(4) AMP_ROW(5) AMP_ROW(6)
    AMP_CASE(1, 5) AMP_CASE(2, 5) AMP_CASE(3, 5) AMP_CASE(4, 5) AMP_CASE(1, 6) AMP_CASE(2, 6) AMP_CASE(3, 6) AMP_CASE(4, 6)
#undef AMP_ROW
#undef AMP_CASE
    AMP_LAUNCH_CHECK();
    return 0;
}

} // namespace

namespace amp {

int duv_mfma_fwd(const athena_mp_graph *g, int Fi, int Fo, const float *a, const float *w, int act, float *c)
{
    // W_d(o,i) flat o + Fo*i:  output index o (stride 1), contraction index i (stride Fo)
    return launch_rows(g, a, Fi, w, (int64_t)Fo * Fi, 1, Fo, c, Fo, /*div_in=*/1, act);
}

int duv_mfma_bwd_a(const athena_mp_graph *g, int Fi, int Fo, const float *grad, const float *w, float *da)
{
    // da[v,i] = (sum_o g[v,o] W_d(o,i)) / d:  output index i (stride Fo), contraction index o (stride 1)
    return launch_rows(g, grad, Fo, w, (int64_t)Fo * Fi, Fo, 1, da, Fi, /*div_in=*/0, ATHENA_MP_ACT_NONE);
}

int duv_mfma_bwd_w(const athena_mp_graph *g, int Fi, int Fo, const float *grad, const float *a, float *dw)
{
    const int it = ceil16(Fi), ot = ceil16(Fo);
    if ((Fi & 3) || (Fo & 3) || !frag_shape(it, ot)) return -1;
    const int nt = g->n_btiles, nb = (int)g->btile_off.size() - 1, n = Fi * Fo;
    if (nt == 0) {
        AMP_HIP(hipMemsetAsync(dw, 0, sizeof(float) * (size_t)nb * n, stream()));
        return 0;
    }
    if (nb > kMaxBuckets) return -1;
    const BucketSplit sp = make_split(g, 512, 4);
    const int nwg = sp.unit_off[nb];
    void *slabs = nullptr;
    if (workspace(&slabs, sizeof(float) * (size_t)nwg * n, 2)) return 1;
#define AMP_CASE(IT_, OT_)                                                                                         \
    if (it == IT_ && ot == OT_) {                                                                                  \
        hipLaunchKernelGGL((duv_dw_kernel<IT_, OT_>), dim3(nwg), dim3(256), 0, stream(), sp, g->btile_rows, a, Fi, \
                           grad, Fo, (float *)slabs);                                                              \
    }
#define AMP_ROW(IT_) AMP_CASE(IT_, 1) AMP_CASE(IT_, 2) AMP_CASE(IT_, 3) AMP_CASE(IT_, 4)
    AMP_ROW(1) AMP_ROW(2) AMP_ROW(3) AMP_ROW(4) AMP_ROW(5) AMP_ROW(6)
    AMP_CASE(1, 5) AMP_CASE(2, 5) AMP_CASE(3, 5) AMP_CASE(4, 5) AMP_CASE(1, 6) AMP_CASE(2, 6) AMP_CASE(3, 6) AMP_CASE(4, 6)
#undef AMP_ROW
#undef AMP_CASE
    AMP_LAUNCH_CHECK();
    std::vector<int> first(nb, 0), count(nb, 0);
    for (int b = 0; b < nb; ++b) first[b] = sp.unit_off[b], count[b] = sp.unit_off[b + 1] - sp.unit_off[b];
    return slab_reduce_segs((const float *)slabs, n, nb, first.data(), count.data(), dw, n, false);
}

} // namespace amp
