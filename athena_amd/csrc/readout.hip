// Duvenaud readout of one time step in one launch per direction.
//
//   forward   athena_duvenaud_msgpass_layer.f90:838-855
//       p[v,:] = softmax_over_outputs(R z[v,:]);   out[s,:] (+)= sum_{v in graph s} p[v,:]
//   reverse   the chain softmax (athena_diffstruc_extd_sub.f90:309-313) -> matmul (diffstruc) ->
//             message activation (athena_activation_*.f90 differentiate), composed:
//       dl[v,:] = p[v,:] (g_s - <g_s, p[v,:]>);  dR (+)= dl^T z;  dc = act'(z) (dl R + dz_next)
//
// The number of outputs O is small (10 in msgpass_chemical), so the contraction is written with the
// VERTEX index on the MFMA column (n) axis: every 16x16x4 MFMA handles 16 vertices, the logits of a
// vertex land in 4 lanes x 4 registers, and the softmax is two cross-lane steps.  z is read exactly
// once per direction (16 B per lane, row-contiguous), logits and dl never reach HBM.
#include <algorithm>

#include "common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

template <int ACT>
__device__ __forceinline__ float act_back(float y, float g)
{
    if constexpr (ACT == ATHENA_MP_ACT_RELU) return y > 0.0f ? g : 0.0f;
    if constexpr (ACT == ATHENA_MP_ACT_SIGMOID) return g * y * (1.0f - y);
    if constexpr (ACT == ATHENA_MP_ACT_TANH) return g * (1.0f - y * y);
    return g;
}

__device__ __forceinline__ float xor_add(float v)
{
    v = v + __shfl_xor(v, 16);
    return v + __shfl_xor(v, 32);
}
__device__ __forceinline__ float xor_max(float v)
{
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
}

// lane (n = lane&15, q = lane>>4):  z fragment j = z[row n][16j + 4q .. +3]  (B operand, k = q)
//                                   R fragment (j,c) = R[o = n][16j + 4q + c] (A operand, m = o)
// accumulator register r = logits[row n][o = 4q + r]
template <int J>
__global__ __launch_bounds__(256) void readout_fwd_kernel(int O, int64_t N, const float *__restrict__ z,
                                                          const float *__restrict__ R, float *__restrict__ p)
{
    constexpr int FV = 16 * J;
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
    const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    const int64_t tiles = (N + 15) >> 4;

    float Rf[J][4];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) Rf[j][c] = n < O ? R[(size_t)(16 * j + 4 * q + c) * O + n] : 0.0f;

    v4f zf[J], zn[J];
    auto load = [&](v4f(&dst)[J], int64_t tile) {
        const int64_t row = min(tile * 16 + n, N - 1);
        const float *src = z + row * FV + 4 * q;
#pragma unroll
        for (int j = 0; j < J; ++j) dst[j] = *reinterpret_cast<const v4f *>(src + 16 * j);
    };
    if (gw < tiles) load(zn, gw);
    for (int64_t tile = gw; tile < tiles; tile += nw) {
#pragma unroll
        for (int j = 0; j < J; ++j) zf[j] = zn[j];
        if (tile + nw < tiles) load(zn, tile + nw);
        v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Rf[j][c], zf[j][c], acc, 0, 0, 0);
        float m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * q + r < O) m = fmaxf(m, acc[r]);
        m = xor_max(m);
        float e[4], s = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            e[r] = 4 * q + r < O ? expf(acc[r] - m) : 0.0f;
            s = s + e[r];
        }
        s = xor_add(s);
        // rows past the end repeat row N-1 (same loads, same values): duplicate stores instead of a branch
        float *dst = p + min(tile * 16 + n, N - 1) * O + 4 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * q + r < O) dst[r] = e[r] / s;
    }
}

// one thread per (graph, output): sequential over the graph's vertices (the reference's order)
__global__ void readout_segsum_kernel(int O, int S, const int32_t *__restrict__ seg, const float *__restrict__ p,
                                      float *__restrict__ out, int accumulate)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S * O) return;
    int s = t / O, o = t - s * O;
    float acc = 0.0f;
    // eight loads in flight, added in vertex order (the same sum: the loop was one memory latency per vertex)
    const int v1 = seg[s + 1];
    int v = seg[s];
    for (; v + 8 <= v1; v += 8) {
        float q[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = p[(size_t)(v + k) * O + o];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = acc + q[k];
    }
    for (; v < v1; ++v) acc = acc + p[(size_t)v * O + o];
    out[t] = accumulate ? out[t] + acc : acc;
}

// graph of every vertex: one thread per graph fills its vertex range
__global__ void readout_gid_kernel(int S, const int32_t *__restrict__ seg, int32_t *__restrict__ gid)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    for (int v = seg[s]; v < seg[s + 1]; ++v) gid[v] = s;
}

// dp[v, :] = gout[graph of v, :]
__global__ void readout_bcast_kernel(int O, int64_t N, const int32_t *__restrict__ gid, const float *__restrict__ gout,
                                     float *__restrict__ dp)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * O) return;
    const int64_t v = t / O;
    dp[t] = gout[(size_t)gid[v] * O + (t - v * O)];
}

// Reverse pass.  dl stays in registers as the B operand (k = q <-> o = 4q + r) of the dz product,
// whose accumulator comes out in the z-fragment layout, so dc = act'(z)(dz + dz_next) is formed in
// place and stored 16 B per lane.  dR contracts over VERTICES, so z and dl are turned once through a
// wave-private LDS tile to put the vertex on the k axis.
// DIN: a dz_next arrives from step t + 1; its rows are prefetched one tile ahead beside z's (loaded where they are used they
// cost a latency per tile: 0.443 ms per launch at configs[2]; prefetched 0.365 ms; 0.272 ms without dz_next)
template <int J, int ACT, bool DIN>
__global__ __launch_bounds__(256) void readout_bwd_kernel(int O, int64_t N, const float *__restrict__ z,
                                                          const float *__restrict__ R, const float *__restrict__ p,
                                                          const int32_t *__restrict__ gid,
                                                          const float *__restrict__ gout,
                                                          const float *__restrict__ dz_in, float *__restrict__ dc,
                                                          float *__restrict__ slabs)
{
    constexpr int FV = 16 * J, ZP = FV + 4, DP = 20;
    __shared__ __attribute__((aligned(16))) float zl_all[4][16 * ZP];
    __shared__ float dl_all[4][16 * DP];
    __shared__ float red[4][FV * 16];
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    float *zl = zl_all[wave], *dll = dl_all[wave];
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave, nw = (int64_t)gridDim.x * 4;
    const int64_t tiles = (N + 15) >> 4;

    // A operand of the dz product: (ft, r) -> R[o = 4q' + r][f = 16 ft + m] with m = n, k = q' = q
    float Ra[J][4];
#pragma unroll
    for (int ft = 0; ft < J; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ra[ft][r] = 4 * q + r < O ? R[(size_t)(16 * ft + n) * O + 4 * q + r] : 0.0f;

    v4f accR[J];
#pragma unroll
    for (int ft = 0; ft < J; ++ft) accR[ft] = v4f{0.0f, 0.0f, 0.0f, 0.0f};

    v4f zf[J], zn[J];
    [[maybe_unused]] v4f df[J], dn[J];
    float pf[4], pn[4], gf[4], gn[4];
    // software pipeline: graph ids two tiles ahead, rows one tile ahead, so no load waits on a load of its own stage
    auto load = [&](v4f(&zd)[J], float(&pd)[4], float(&gd)[4], int64_t tile, int g) {
        const int64_t row = min(tile * 16 + n, N - 1);
        const float *src = z + row * FV + 4 * q;
#pragma unroll
        for (int j = 0; j < J; ++j) zd[j] = *reinterpret_cast<const v4f *>(src + 16 * j);
        if constexpr (DIN) {
            const float *dsrc = dz_in + row * FV + 4 * q;
#pragma unroll
            for (int j = 0; j < J; ++j) dn[j] = *reinterpret_cast<const v4f *>(dsrc + 16 * j);
        }
        const float *ps = p + row * O + 4 * q;
        const float *gs = gout + (size_t)g * O + 4 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = 4 * q + r < O;
            pd[r] = ok ? ps[r] : 0.0f;
            gd[r] = ok ? gs[r] : 0.0f;
        }
    };
    auto gid_of = [&](int64_t tile) { return tile < tiles ? gid[min(tile * 16 + n, N - 1)] : 0; };
    int g_n = gid_of(gw), g_nn = gid_of(gw + nw);
    if (gw < tiles) load(zn, pn, gn, gw, g_n);
    for (int64_t tile = gw; tile < tiles; tile += nw) {
#pragma unroll
        for (int j = 0; j < J; ++j) zf[j] = zn[j];
        if constexpr (DIN) {
#pragma unroll
            for (int j = 0; j < J; ++j) df[j] = dn[j];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) pf[r] = pn[r], gf[r] = gn[r];
        g_n = g_nn;
        if (tile + nw < tiles) load(zn, pn, gn, tile + nw, g_n);
        g_nn = gid_of(tile + 2 * nw);
        const int64_t row = tile * 16 + n;
        const bool ok = row < N;

        float dot = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) dot = dot + gf[r] * pf[r];
        dot = xor_add(dot);
        float dl[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) dl[r] = pf[r] * (gf[r] - dot);

        // stage the tile for the vertex-contracted product before the registers are consumed
#pragma unroll
        for (int j = 0; j < J; ++j) *reinterpret_cast<v4f *>(zl + n * ZP + 16 * j + 4 * q) = zf[j];
#pragma unroll
        for (int r = 0; r < 4; ++r) dll[n * DP + 4 * q + r] = ok ? dl[r] : 0.0f;   // repeated rows count once in dR

        // dz[row n][16 ft + 4q .. +3] = sum_o dl[row][o] R[o][f]
        // rows past the end repeat row N-1 (same loads, same values): duplicate stores instead of a branch
        float *dst = dc + min(row, N - 1) * FV + 4 * q;
#pragma unroll
        for (int ft = 0; ft < J; ++ft) {
            v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Ra[ft][r], dl[r], acc, 0, 0, 0);
            if constexpr (DIN) acc = acc + df[ft];
            v4f o4;
#pragma unroll
            for (int c = 0; c < 4; ++c) o4[c] = act_back<ACT>(zf[ft][c], acc[c]);
            *reinterpret_cast<v4f *>(dst + 16 * ft) = o4;
        }

        // dR[f][o] += sum_rows z[row][f] dl[row][o]:  A[m = f][k = row], B[k = row][n = o]
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float db[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) db[r] = dll[(4 * q + r) * DP + n];
#pragma unroll
        for (int ft = 0; ft < J; ++ft)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                accR[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(zl[(4 * q + r) * ZP + 16 * ft + n], db[r], accR[ft], 0, 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // accR[ft][r] = dR[f = 16 ft + 4q + r][o = n]; waves added in fixed order, one slab per workgroup
#pragma unroll
    for (int ft = 0; ft < J; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(16 * ft + 4 * q + r) * 16 + n] = accR[ft][r];
    __syncthreads();
    float *slab = slabs + (size_t)blockIdx.x * FV * O;
    for (int t = threadIdx.x; t < FV * O; t += 256) {
        const int f = t / O, o = t - f * O;
        slab[t] = ((red[0][f * 16 + o] + red[1][f * 16 + o]) + red[2][f * 16 + o]) + red[3][f * 16 + o];
    }
}

// graph of every tile slot of the degree buckets (copy 1 of the tile ids: padding slots repeat the tile's first vertex)
__global__ void readout_tile_gid_kernel(int64_t n_slots, const int32_t *__restrict__ trows_abs, const int32_t *__restrict__ gid,
                                        int32_t *__restrict__ tgid)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_slots) tgid[t] = gid[trows_abs[t]];
}

bool fused_readout_shape(int Fv, int O) { return O <= 16 && (Fv == 16 || Fv == 32 || Fv == 64 || Fv == 128); }

int readout_grid(int64_t N, int wg_per_cu)
{
    const int64_t tiles = (N + 15) / 16;
    return (int)std::max<int64_t>(1, std::min<int64_t>((tiles + 3) / 4, 256 * wg_per_cu));
}

} // namespace

using namespace amp;

extern "C" {

int athena_mp_duvenaud_readout_fwd(int64_t N, int32_t Fv, int32_t O, int32_t S, const int32_t *seg,
                                   const float *z, const float *R, float *p, float *out, int32_t accumulate)
{
    AMP_REQUIRE(N >= 0 && Fv > 0 && O > 0 && S >= 0 && seg && R && out && (N == 0 || (z && p)),
                "duvenaud_readout_fwd: bad arguments");
    if (N > 0) {
        if (fused_readout_shape(Fv, O)) {
            const dim3 grid(readout_grid(N, 7));   // 66 VGPRs at Fv = 64: seven waves per SIMD
#define AMP_CASE(J_)                                                                                               \
    if (Fv == 16 * J_) hipLaunchKernelGGL((readout_fwd_kernel<J_>), grid, dim3(256), 0, stream(), O, N, z, R, p);
            AMP_CASE(1) AMP_CASE(2) AMP_CASE(4) AMP_CASE(8)
#undef AMP_CASE
            AMP_LAUNCH_CHECK();
        } else {
            void *lg = nullptr;
            if (workspace(&lg, sizeof(float) * (size_t)N * O, 5)) return 1;
            int rc = athena_mp_gemm_fwd(N, Fv, O, z, R, nullptr, ATHENA_MP_ACT_NONE, (float *)lg);
            if (rc) return rc;
            return athena_mp_softmax_segsum_fwd(O, N, S, seg, (const float *)lg, p, out, accumulate);
        }
    }
    if (S > 0) {
        hipLaunchKernelGGL(readout_segsum_kernel, dim3((unsigned)(((int64_t)S * O + 255) / 256)), dim3(256), 0,
                           stream(), O, S, seg, p, out, accumulate);
        AMP_LAUNCH_CHECK();
    }
    return 0;
}

int athena_mp_duvenaud_readout_bwd(int64_t N, int32_t Fv, int32_t O, int32_t S, const int32_t *seg,
                                   const float *z, const float *R, const float *p, const float *gout,
                                   const float *dz_next, int32_t act, float *dc, float *dR, int32_t accumulate)
{
    AMP_REQUIRE(N >= 0 && Fv > 0 && O > 0 && S > 0 && seg && R && gout && dR && (N == 0 || (z && p && dc)),
                "duvenaud_readout_bwd: bad arguments");
    AMP_REQUIRE(act >= 0 && act <= ATHENA_MP_ACT_TANH, "duvenaud_readout_bwd: unknown activation %d", act);
    const int n = Fv * O;
    if (N == 0) {
        if (!accumulate) AMP_HIP(hipMemsetAsync(dR, 0, sizeof(float) * n, stream()));
        return 0;
    }
    if (!fused_readout_shape(Fv, O)) {
        void *dl = nullptr;
        if (workspace(&dl, sizeof(float) * (size_t)N * O, 5)) return 1;
        int rc = athena_mp_softmax_segsum_bwd(O, N, S, seg, p, gout, (float *)dl);
        if (!rc) rc = gemm_dw_dispatch(N, Fv, O, z, (const float *)dl, dR, accumulate != 0);
        if (!rc) rc = athena_mp_gemm_dx(N, Fv, O, (const float *)dl, R, dc);
        if (!rc && dz_next) rc = athena_mp_axpy((int64_t)N * Fv, 1.0f, dz_next, dc);
        if (!rc && act != ATHENA_MP_ACT_NONE) rc = athena_mp_activation_bwd(act, (int64_t)N * Fv, z, dc, dc);
        return rc;
    }
    void *gid = nullptr, *slabs = nullptr;
    const int nblk = readout_grid(N, 4);      // 132 / 168 VGPRs at Fv = 64 (without / with dz_next): three waves per SIMD; one slab
                                              // per workgroup (three workgroups per CU measured the same as four)
    if (workspace(&gid, sizeof(int32_t) * (size_t)N, 4) || workspace(&slabs, sizeof(float) * (size_t)nblk * n, 3))
        return 1;
    hipLaunchKernelGGL(readout_gid_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, stream(), S, seg,
                       (int32_t *)gid);
    AMP_LAUNCH_CHECK();
#define AMP_CASE(J_, A_)                                                                                           \
    if (Fv == 16 * J_ && act == A_) {                                                                              \
        if (dz_next)                                                                                               \
            hipLaunchKernelGGL((readout_bwd_kernel<J_, A_, true>), dim3(nblk), dim3(256), 0, stream(), O, N, z, R, p, \
                               (const int32_t *)gid, gout, dz_next, dc, (float *)slabs);                           \
        else                                                                                                       \
            hipLaunchKernelGGL((readout_bwd_kernel<J_, A_, false>), dim3(nblk), dim3(256), 0, stream(), O, N, z, R, p, \
                               (const int32_t *)gid, gout, dz_next, dc, (float *)slabs);                           \
    }
#define AMP_ACTS(J_) AMP_CASE(J_, 0) AMP_CASE(J_, 1) AMP_CASE(J_, 2) AMP_CASE(J_, 3)
    AMP_ACTS(1) AMP_ACTS(2) AMP_ACTS(4) AMP_ACTS(8)
#undef AMP_ACTS
#undef AMP_CASE
    AMP_LAUNCH_CHECK();
    return slab_reduce((const float *)slabs, nblk, n, dR, accumulate != 0);
}


/* One time step of the Duvenaud layer's reverse pass in one call (update_readout_duvenaud's and update_message_duvenaud's
 * reverse, athena_duvenaud_msgpass_layer.f90:755-859 through grad_reverse): the readout's reverse (softmax -> matmul(R, z) ->
 * message activation) and the update's reverse (da split as athena_mp_duvenaud_update_bwd_split, dW) -- ONE launch where the
 * shape allows it (F_v = 64, F_v + F_e <= 96, O <= 16: dc never reaches HBM), the two launches through a workspace otherwise.
 * Same results as athena_mp_duvenaud_readout_bwd followed by athena_mp_duvenaud_update_bwd_split up to the order of the dR sum.
 * accumulate_da_e: da_e is added to (the edge-feature gradient is linear in da_e: a layer sums da_e over its time steps and
 * calls athena_mp_duvenaud_propagate_bwd_e once).  a_e != NULL: a arrives split, a = a_x [n_rows, Fv] and a_e [n_rows, Fe]
 * (athena_mp_duvenaud_update_readout_fwd_split). */
int athena_mp_duvenaud_readout_update_bwd(const athena_mp_graph *g, int32_t Fv, int32_t Fe, int32_t min_deg, int32_t max_deg,
                                          int32_t O, int32_t S, const int32_t *seg, const float *z, const float *R, const float *p,
                                          const float *gout, const float *dz_next, int32_t act, const float *a,
                                          const float *weight, float *da_x, float *da_e, float *dweight, float *dR,
                                          int32_t accumulate_dR, int32_t accumulate_da_e, const float *a_e)
{
    AMP_REQUIRE(g && Fv > 0 && Fe > 0 && O > 0 && S > 0 && seg && z && R && p && gout && a && weight && da_x && da_e && dweight && dR &&
                    max_deg >= min_deg, "duvenaud_readout_update_bwd: bad arguments");
    AMP_REQUIRE(act >= 0 && act <= ATHENA_MP_ACT_TANH, "duvenaud_readout_update_bwd: unknown activation %d", act);
    const int64_t N = g->n_rows;
    const int32_t Fi = Fv + Fe;
    if (N >= 1024 && Fv == 64 && Fi <= 96 && (Fi & 3) == 0 && O <= 16 && max_deg - min_deg + 1 <= 32) {
        if (duvenaud_buckets(g, min_deg, max_deg)) return 1;
        const int64_t n_slots = (int64_t)16 * g->n_btiles;
        void *gid = nullptr, *tgid = nullptr, *rslabs = nullptr;
        if (workspace(&gid, sizeof(int32_t) * (size_t)N, 4) || workspace(&tgid, sizeof(int32_t) * (size_t)n_slots, 12) ||
            workspace(&rslabs, sizeof(float) * (size_t)256 * Fv * O, 3))
            return 1;
        hipLaunchKernelGGL(readout_gid_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, stream(), S, seg, (int32_t *)gid);
        hipLaunchKernelGGL(readout_tile_gid_kernel, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, stream(), n_slots,
                           g->btile_rows + (size_t)n_slots, (const int32_t *)gid, (int32_t *)tgid);
        AMP_LAUNCH_CHECK();
        int n_slabs = 0;
        const int rc = duv_mfma_bwd_readout(g, Fi, Fv, O, act, a, weight, z, dz_next, p, (const int32_t *)tgid, gout, R, da_x, da_e,
                                            dweight, (float *)rslabs, &n_slabs, accumulate_da_e != 0, a_e);
        if (rc > 0) return rc;
        if (rc == 0) return slab_reduce((const float *)rslabs, n_slabs, Fv * O, dR, accumulate_dR != 0);
    }
    if (a_e) {   // a arrived split (a = a_x [n_rows, Fv]): the two launches take it packed
        if (duv_pack_a(g, Fv, Fe, a, a_e, &a)) return 1;
    }
    void *dc = nullptr;
    if (workspace(&dc, sizeof(float) * (size_t)std::max<int64_t>(N, 1) * Fv, 13)) return 1;
    if (const int rc = athena_mp_duvenaud_readout_bwd(N, Fv, O, S, seg, z, R, p, gout, dz_next, act, (float *)dc, dR, accumulate_dR)) return rc;
    if (!accumulate_da_e)
        return athena_mp_duvenaud_update_bwd_split(g, Fv, Fe, Fv, min_deg, max_deg, (const float *)dc, a, weight, da_x, da_e, dweight);
    void *et = nullptr;
    if (workspace(&et, sizeof(float) * (size_t)std::max<int64_t>(N, 1) * Fe, 8)) return 1;
    if (const int rc = athena_mp_duvenaud_update_bwd_split(g, Fv, Fe, Fv, min_deg, max_deg, (const float *)dc, a, weight, da_x, (float *)et, dweight))
        return rc;
    return athena_mp_axpy(N * Fe, 1.0f, (const float *)et, da_e);
}

/* per-graph sum of already-activated per-vertex values, and its reverse (a broadcast): the readout of
 * athena_duvenaud_msgpass_layer.f90:838-855 for a readout activation other than softmax */
int athena_mp_segment_sum(int32_t O, int64_t N, int32_t S, const int32_t *seg, const float *p, float *out,
                          int32_t accumulate)
{
    AMP_REQUIRE(O > 0 && N >= 0 && S >= 0 && seg && out && (N == 0 || p), "segment_sum: bad arguments");
    if (S > 0) {
        hipLaunchKernelGGL(readout_segsum_kernel, dim3((unsigned)(((int64_t)S * O + 255) / 256)), dim3(256), 0,
                           stream(), O, S, seg, p, out, accumulate);
        AMP_LAUNCH_CHECK();
    }
    return 0;
}

int athena_mp_segment_sum_bwd(int32_t O, int64_t N, int32_t S, const int32_t *seg, const float *gout, float *dp)
{
    AMP_REQUIRE(O > 0 && N >= 0 && S > 0 && seg && gout && (N == 0 || dp), "segment_sum_bwd: bad arguments");
    if (N == 0) return 0;
    void *gid = nullptr;
    if (workspace(&gid, sizeof(int32_t) * (size_t)N, 4)) return 1;
    hipLaunchKernelGGL(readout_gid_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, stream(), S, seg,
                       (int32_t *)gid);
    hipLaunchKernelGGL(readout_bcast_kernel, dim3((unsigned)((N * O + 255) / 256)), dim3(256), 0, stream(), O, N,
                       (const int32_t *)gid, gout, dp);
    AMP_LAUNCH_CHECK();
    return 0;
}

} // extern "C"
