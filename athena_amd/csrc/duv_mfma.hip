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

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_apply(float t, int act)
{
    switch (act) {
    case ATHENA_MP_ACT_RELU: return t > 0.0f ? t : 0.0f;
    case ATHENA_MP_ACT_SIGMOID: return 1.0f / (1.0f + expf(-t));
    case ATHENA_MP_ACT_TANH: return tanhf(t);
    default: return t;
    }
}

// Y[v, 0:NO] = f( sum_k Wd(o,k) X[v,k] ),  Wd(o,k) = W[b*wb + o*so + k*sk]
//   div_in  = 1: X is divided by d = b+1 before the product (forward: a/d, the reference's order)
//   div_in  = 0: the sum is divided by d afterwards (reverse w.r.t. a)
template <int KJ, int OT>
__global__ __launch_bounds__(256) void duv_rows_kernel(int n_tiles, const int32_t *__restrict__ trows,
                                                       const int32_t *__restrict__ tinfo, const float *__restrict__ X,
                                                       int K, const float *__restrict__ W, int64_t wb, int so, int sk,
                                                       float *__restrict__ Y, int NO, int div_in, int act,
                                                       int tiles_per_wave)
{
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
    // wave w takes tiles w, w + W, w + 2W, ...: neighbouring waves stream neighbouring vertices at the same
    // time, every wave sees the same mix of buckets (balanced), and buckets are long runs so the weight
    // fragments are still reloaded only a handful of times per wave
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const int t0 = gw, t1 = n_tiles;
    if (t0 >= t1) return;

    float Wf[OT][KJ][4];
    int cur_b = -1;
    float d = 1.0f, inv = 1.0f;
    bool pow2 = true;

    // software pipeline: vertex ids two tiles ahead, vertex rows one tile ahead (one dependent load per stage)
    v4f xf[KJ], xn[KJ];
    auto load = [&](v4f(&xd)[KJ], int r) {
        const float *src = X + (int64_t)(r < 0 ? ~r : r) * K + 4 * q;
#pragma unroll
        for (int j = 0; j < KJ; ++j)
            xd[j] = 16 * j + 4 * q < K ? *reinterpret_cast<const v4f *>(src + 16 * j) : v4f{0.0f, 0.0f, 0.0f, 0.0f};
    };
    int r_cur = 0, r_n = trows[(int64_t)t0 * 16 + n], r_nn = t0 + nw < t1 ? trows[(int64_t)(t0 + nw) * 16 + n] : 0;
    int info_n = tinfo[t0];   // kept raw: shifting it here would put a full vmcnt(0) wait right behind the prefetch
    load(xn, r_n);
    for (int t = t0; t < t1; t += nw) {
#pragma unroll
        for (int j = 0; j < KJ; ++j) xf[j] = xn[j];
        r_cur = r_n, r_n = r_nn;
        const int b = __builtin_amdgcn_readfirstlane(info_n >> 8);
        // padding slots repeat the tile's first vertex: same loads, same arithmetic, same address, same value --
        // their stores are benign duplicates, so no store sits under a per-lane branch
        const int row = r_cur < 0 ? ~r_cur : r_cur;
        if (b != cur_b) {
            cur_b = b;
            d = (float)(b + 1);
            pow2 = ((b + 1) & b) == 0;
            inv = 1.0f / d;
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
            // drain the fragment loads HERE: left to the compiler, their wait lands at the first MFMA as a
            // vmcnt(0) on the join of both paths, i.e. behind the prefetch that the MFMAs are meant to hide
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt/lgkmcnt untouched
        }
        // prefetch AFTER the (rare) weight reload: the reload's join point carries a full vmcnt(0) wait, which
        // must not sit between the prefetch and the MFMAs that are meant to hide it
        if (t + nw < t1) {
            load(xn, r_n);
            info_n = tinfo[t + nw];
        }
        if (t + 2 * nw < t1) r_nn = trows[(int64_t)(t + 2 * nw) * 16 + n];
        if (div_in) {
            if (pow2) {
#pragma unroll
                for (int j = 0; j < KJ; ++j) xf[j] = xf[j] * inv;   // exact for d = 1, 2, 4, 8, ...
            } else {
#pragma unroll
                for (int j = 0; j < KJ; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) xf[j][c] = xf[j][c] / d;
            }
        }
        float *dst = Y + (int64_t)row * NO + 4 * q;
        // OT independent accumulation chains, interleaved so consecutive MFMAs never wait on each other
        v4f accs[OT];
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) accs[ot] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < KJ; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int ot = 0; ot < OT; ++ot)
                    accs[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wf[ot][j][c], xf[j][c], accs[ot], 0, 0, 0);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
            v4f acc = accs[ot];
            if (!div_in) {
                if (pow2) acc = acc * inv;
                else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = acc[c] / d;
                }
            }
            if (act != ATHENA_MP_ACT_NONE) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = act_apply(acc[c], act);
            }
            if (16 * ot + 4 * q < NO) *reinterpret_cast<v4f *>(dst + 16 * ot) = acc;
        }
    }
}

// slab[blockIdx + b][i*Fo + o] = sum over this workgroup's tiles of bucket b of (a[v,i]/d) g[v,o]
template <int IT, int OT>
__global__ __launch_bounds__(256) void duv_dw_kernel(int n_tiles, const int32_t *__restrict__ trows,
                                                     const int32_t *__restrict__ tinfo,
                                                     const int32_t *__restrict__ toff, const float *__restrict__ A,
                                                     int Fi, const float *__restrict__ G, int Fo,
                                                     float *__restrict__ slabs, int tiles_per_wg)
{
    constexpr int AP = 16 * IT + 4, GP = 16 * OT + 4, FOP = 16 * OT;
    constexpr int kTurn = 4 * 16 * (AP + GP), kRed = 16 * IT * FOP;
    __shared__ __attribute__((aligned(16))) float buf[kTurn > kRed ? kTurn : kRed];
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    float *al = buf + wave * 16 * (AP + GP), *gl = al + 16 * AP;
    const int T0 = blockIdx.x * tiles_per_wg, T1 = min(n_tiles, T0 + tiles_per_wg);

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

    int seg = T0;
    while (seg < T1) {
        const int b = tinfo[seg] >> 8;
        const int send = min(T1, toff[b + 1]);
        const float d = (float)(b + 1), inv = 1.0f / d;
        const bool pow2 = ((b + 1) & b) == 0;
        v4f acc[IT][OT];
#pragma unroll
        for (int i = 0; i < IT; ++i)
#pragma unroll
            for (int o = 0; o < OT; ++o) acc[i][o] = v4f{0.0f, 0.0f, 0.0f, 0.0f};

        int r_n = seg + wave < send ? trows[(int64_t)(seg + wave) * 16 + n] : 0;
        int r_nn = seg + wave + 4 < send ? trows[(int64_t)(seg + wave + 4) * 16 + n] : 0;
        if (seg + wave < send) load(an, gn, r_n);
        for (int t = seg + wave; t < send; t += 4) {
#pragma unroll
            for (int j = 0; j < IT; ++j) af[j] = an[j];
#pragma unroll
            for (int j = 0; j < OT; ++j) gf[j] = gn[j];
            r_n = r_nn;
            if (t + 4 < send) load(an, gn, r_n);
            if (t + 8 < send) r_nn = trows[(int64_t)(t + 8) * 16 + n];
            if (pow2) {
#pragma unroll
                for (int j = 0; j < IT; ++j) af[j] = af[j] * inv;
            } else {
#pragma unroll
                for (int j = 0; j < IT; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) af[j][c] = af[j][c] / d;
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
        float *slab = slabs + (size_t)(blockIdx.x + b) * Fi * Fo;
        for (int t = threadIdx.x; t < Fi * Fo; t += 256) {
            const int i = t / Fo, o = t - i * Fo;
            slab[t] = buf[i * FOP + o];
        }
        __syncthreads();
        seg = send;
    }
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
    const int max_waves = 256 * 4 * 2;   // two resident waves per SIMD at ~200 VGPRs
    const int tpw = 0;
    const int waves = std::min(nt, max_waves);
    const dim3 grid((waves + 3) / 4);
#define AMP_CASE(KJ_, OT_)                                                                                         \
    if (kj == KJ_ && ot == OT_) {                                                                                  \
        hipLaunchKernelGGL((duv_rows_kernel<KJ_, OT_>), grid, dim3(256), 0, amp::stream(), nt, g->btile_rows,      \
                           g->btile_info, X, K, W, wb, so, sk, Y, NO, div_in, act, tpw);                           \
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
    const int max_wg = 512;
    const int tpw = std::max(4, (nt + max_wg - 1) / max_wg);
    const int nwg = (nt + tpw - 1) / tpw;
    void *slabs = nullptr;
    if (workspace(&slabs, sizeof(float) * (size_t)(nwg + nb) * n, 2)) return 1;
#define AMP_CASE(IT_, OT_)                                                                                         \
    if (it == IT_ && ot == OT_) {                                                                                  \
        hipLaunchKernelGGL((duv_dw_kernel<IT_, OT_>), dim3(nwg), dim3(256), 0, stream(), nt, g->btile_rows,        \
                           g->btile_info, g->btile_off_dev, a, Fi, grad, Fo, (float *)slabs, tpw);                 \
    }
#define AMP_ROW(IT_) AMP_CASE(IT_, 1) AMP_CASE(IT_, 2) AMP_CASE(IT_, 3) AMP_CASE(IT_, 4)
    AMP_ROW(1) AMP_ROW(2) AMP_ROW(3) AMP_ROW(4) AMP_ROW(5) AMP_ROW(6)
    AMP_CASE(1, 5) AMP_CASE(2, 5) AMP_CASE(3, 5) AMP_CASE(4, 5) AMP_CASE(1, 6) AMP_CASE(2, 6) AMP_CASE(3, 6) AMP_CASE(4, 6)
#undef AMP_ROW
#undef AMP_CASE
    AMP_LAUNCH_CHECK();
    std::vector<int> first(nb, 0), count(nb, 0);
    for (int b = 0; b < nb; ++b) {
        const int tb0 = g->btile_off[b], tb1 = g->btile_off[b + 1];
        if (tb1 == tb0) continue;                          // empty bucket: count 0 -> zeros
        const int w0 = tb0 / tpw, w1 = (tb1 - 1) / tpw;    // workgroups that saw bucket b: slabs w0+b .. w1+b
        first[b] = w0 + b, count[b] = w1 - w0 + 1;
    }
    return slab_reduce_segs((const float *)slabs, n, nb, first.data(), count.data(), dw, n, false);
}

} // namespace amp
