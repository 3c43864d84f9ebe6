// Duvenaud degree-conditioned update and graph readout.
//
//   duvenaud_update                       athena_diffstruc_extd_sub_duvenaud.f90:176-228
//       d_v = clamp(deg_v, min, max) - min + 1;  c[v,:] = W_d (a[v,:] / real(d_v))
//       (the BUCKET INDEX is the divisor -- SURVEY.md F8; reproduced exactly)
//   get_partial_duvenaud_update_val        :284-324   da[v,:] = (g[v,:]^T W_d) / real(d_v)
//   get_partial_duvenaud_update_weight_val :326-368   dW_d[o,i] += g[v,o] a[v,i] / real(d_v)
//   readout                                athena_duvenaud_msgpass_layer.f90:838-855
//       out[s,:] (+)= sum_{v in graph s} softmax_over_outputs(logits[v,:])
//       softmax: athena_diffstruc_extd_sub.f90:309-313 (max-subtracted, per vertex)
//
// fwd / bwd_a keep the reference's operation order (divide first, k-ordered multiply-add, no
// contraction) so they agree with the strict-fp32 restatement bit for bit; bwd_w reduces vertex
// chunks into per-bucket slabs that are summed in fixed order (deterministic, no atomics).
#include <algorithm>

#include "common.h"

namespace {

__device__ __forceinline__ int bucket_of(int deg, int min_deg, int max_deg)
{
    return max(min_deg, min(deg, max_deg)) - min_deg + 1; // 1-based, as the reference's d
}

__global__ void duv_update_fwd_kernel(const int32_t *__restrict__ deg, int min_deg, int max_deg, int Fi, int Fo,
                                      int64_t N, const float *__restrict__ a, const float *__restrict__ weight,
                                      float *__restrict__ c)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * Fo) return;
    int64_t v = t / Fo;
    int o = (int)(t - v * Fo);
    int d = bucket_of(deg[v], min_deg, max_deg);
    const float *Wd = weight + (size_t)(d - 1) * Fo * Fi;
    const float rd = (float)d;
    const float *av = a + v * Fi;
    float s = 0.0f;
    for (int i = 0; i < Fi; ++i) s = s + Wd[o + (size_t)Fo * i] * (av[i] / rd);
    c[t] = s;
}

__global__ void duv_update_bwd_a_kernel(const int32_t *__restrict__ deg, int min_deg, int max_deg, int Fi,
                                        int Fo, int64_t N, const float *__restrict__ g,
                                        const float *__restrict__ weight, float *__restrict__ da)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * Fi) return;
    int64_t v = t / Fi;
    int i = (int)(t - v * Fi);
    int d = bucket_of(deg[v], min_deg, max_deg);
    const float *Wd = weight + (size_t)(d - 1) * Fo * Fi + (size_t)Fo * i;
    const float *gv = g + v * Fo;
    float s = 0.0f;
    for (int o = 0; o < Fo; ++o) s = s + gv[o] * Wd[o];
    da[t] = s / (float)d;
}

// grid = (chunks, buckets).  Block (chunk, d) walks its vertex chunk and accumulates only the
// vertices of bucket d, so every g/a row is read by exactly one block.
__global__ void duv_update_bwd_w_kernel(const int32_t *__restrict__ deg, int min_deg, int max_deg, int Fi,
                                        int Fo, int64_t N, int64_t rows_per_chunk,
                                        const float *__restrict__ g, const float *__restrict__ a,
                                        float *__restrict__ slabs)
{
    const int d = blockIdx.y + 1;
    const int nb = gridDim.y;
    const int64_t v0 = (int64_t)blockIdx.x * rows_per_chunk, v1 = min(N, v0 + rows_per_chunk);
    const float rd = (float)d;
    float *slab = slabs + ((size_t)blockIdx.x * nb + (d - 1)) * Fo * Fi;
    for (int t = threadIdx.x; t < Fo * Fi; t += blockDim.x) {
        int i = t / Fo, o = t - i * Fo; // flat index o + Fo*i == t
        float s = 0.0f;
        for (int64_t v = v0; v < v1; ++v)
            if (bucket_of(deg[v], min_deg, max_deg) == d) s = s + g[v * Fo + o] * a[v * Fi + i] / rd;
        slab[t] = s;
    }
}

__global__ void slab_reduce_kernel2(const float *__restrict__ slabs, int n_slabs, int n, float *__restrict__ out)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float s = 0.0f;
    for (int b = 0; b < n_slabs; ++b) s = s + slabs[(size_t)b * n + t];
    out[t] = s;
}

// one thread per vertex: O (number of outputs) is small (10 in msgpass_chemical)
__global__ void softmax_rows_kernel(int O, int64_t N, const float *__restrict__ z, float *__restrict__ p)
{
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    const float *zv = z + v * O;
    float *pv = p + v * O;
    float m = zv[0];
    for (int o = 1; o < O; ++o) m = fmaxf(m, zv[o]);
    float s = 0.0f;
    for (int o = 0; o < O; ++o) {
        float e = expf(zv[o] - m);
        pv[o] = e;
        s = s + e;
    }
    for (int o = 0; o < O; ++o) pv[o] = pv[o] / s;
}

// one thread per (graph, output): sequential over the graph's vertices (the reference's order)
__global__ void segsum_kernel(int O, int S, const int32_t *__restrict__ seg, const float *__restrict__ p,
                              float *__restrict__ out, int accumulate)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S * O) return;
    int s = t / O, o = t - s * O;
    float acc = 0.0f;
    for (int v = seg[s]; v < seg[s + 1]; ++v) acc = acc + p[(size_t)v * O + o];
    out[t] = accumulate ? out[t] + acc : acc;
}

// one wave-lane per vertex, segment found by binary search
__global__ void softmax_segsum_bwd_kernel(int O, int64_t N, int S, const int32_t *__restrict__ seg,
                                          const float *__restrict__ p, const float *__restrict__ gout,
                                          float *__restrict__ dz)
{
    int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    int lo = 0, hi = S; // seg[lo] <= v < seg[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (seg[mid] <= v) lo = mid; else hi = mid;
    }
    const float *gs = gout + (size_t)lo * O;
    const float *pv = p + v * O;
    float dot = 0.0f;
    for (int o = 0; o < O; ++o) dot = dot + gs[o] * pv[o];
    for (int o = 0; o < O; ++o) dz[v * O + o] = pv[o] * (gs[o] - dot);
}

} // namespace

namespace amp {
// stable sort of the vertices by degree bucket (host, once per (min,max)); each bucket becomes one
// contiguous run of g->bucket_perm so the bucketed update is <= D dense contractions with row indirection
int duvenaud_buckets(const athena_mp_graph *g, int min_deg, int max_deg)
{
    if (g->bucket_perm && g->bucket_min == min_deg && g->bucket_max == max_deg) return 0;
    const int nb = max_deg - min_deg + 1;
    const int32_t n = g->n_rows;
    std::vector<int64_t> off(nb + 1, 0);
    std::vector<int32_t> bucket(n);
    for (int32_t v = 0; v < n; ++v) {
        int d = std::max(min_deg, std::min(g->h_deg_row[v], max_deg)) - min_deg; // 0-based
        bucket[v] = d;
        off[d + 1]++;
    }
    for (int b = 0; b < nb; ++b) off[b + 1] += off[b];
    std::vector<int32_t> perm(n);
    {
        std::vector<int64_t> pos(off.begin(), off.end() - 1);
        for (int32_t v = 0; v < n; ++v) perm[pos[bucket[v]]++] = v;
    }
    std::vector<int32_t> tstart, tinfo, toff(nb + 1, 0);
    for (int b = 0; b < nb; ++b) {
        for (int64_t i = off[b]; i < off[b + 1]; i += 16) {
            tstart.push_back((int32_t)i);
            tinfo.push_back((b << 8) | (int32_t)std::min<int64_t>(16, off[b + 1] - i));
        }
        toff[b + 1] = (int32_t)tstart.size();
    }
    if (g->bucket_perm) {
        AMP_HIP(hipStreamSynchronize(stream()));
        AMP_HIP(hipFree(g->bucket_perm));
        AMP_HIP(hipFree(g->btile_start));
        AMP_HIP(hipFree(g->btile_info));
        AMP_HIP(hipFree(g->btile_rows));
        AMP_HIP(hipFree(g->btile_off_dev));
        g->bucket_perm = g->btile_start = g->btile_info = g->btile_rows = g->btile_off_dev = nullptr;
    }
    const size_t nt = tstart.size();
    // four copies back to back, [16 nt] each:
    //   0  padding slots as ~(first vertex of the tile): the weight-gradient kernel zeroes their gradient rows
    //   1  padding slots as the first vertex itself: the row kernels load and store them as benign duplicates
    //   2  copy 1 transposed 4 x 4 inside each tile (slot 4 i + r at position 4 r + i): a lane that serves rows r, 4 + r,
    //      8 + r, 12 + r of a tile in four coalesced loads fetches its four ids with one 16-byte load
    //   3  copy 0 transposed the same way
    std::vector<int32_t> trows(64 * nt);
    for (size_t t = 0; t < nt; ++t) {
        const int cnt = tinfo[t] & 255;
        for (int i = 0; i < 16; ++i) {
            const int32_t v = i < cnt ? perm[tstart[t] + i] : perm[tstart[t]];
            const int32_t sv = i < cnt ? v : ~v;
            const int tp = 4 * (i & 3) + (i >> 2);
            trows[16 * t + i] = sv;
            trows[16 * (nt + t) + i] = v;
            trows[16 * (2 * nt + t) + tp] = v;
            trows[16 * (3 * nt + t) + tp] = sv;
        }
    }
    AMP_HIP(hipMalloc((void **)&g->btile_rows, sizeof(int32_t) * (nt ? 64 * nt : 1)));
    if (nt) AMP_HIP(hipMemcpy(g->btile_rows, trows.data(), sizeof(int32_t) * 64 * nt, hipMemcpyHostToDevice));
    AMP_HIP(hipMalloc((void **)&g->bucket_perm, sizeof(int32_t) * (n ? n : 1)));
    AMP_HIP(hipMalloc((void **)&g->btile_start, sizeof(int32_t) * (nt ? nt : 1)));
    AMP_HIP(hipMalloc((void **)&g->btile_info, sizeof(int32_t) * (nt ? nt : 1)));
    AMP_HIP(hipMalloc((void **)&g->btile_off_dev, sizeof(int32_t) * (nb + 1)));
    if (n) AMP_HIP(hipMemcpy(g->bucket_perm, perm.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
    if (nt) {
        AMP_HIP(hipMemcpy(g->btile_start, tstart.data(), sizeof(int32_t) * nt, hipMemcpyHostToDevice));
        AMP_HIP(hipMemcpy(g->btile_info, tinfo.data(), sizeof(int32_t) * nt, hipMemcpyHostToDevice));
    }
    AMP_HIP(hipMemcpy(g->btile_off_dev, toff.data(), sizeof(int32_t) * (nb + 1), hipMemcpyHostToDevice));
    g->n_btiles = (int32_t)nt;
    g->btile_off = toff;
    g->bucket_off = off;
    g->bucket_min = min_deg;
    g->bucket_max = max_deg;
    return 0;
}
// a = [a_x | a_e] packed into the library's workspace (slot 16): the way shapes outside the split kernels take a split a
int duv_pack_a(const athena_mp_graph *g, int32_t Fv, int32_t Fe, const float *a_x, const float *a_e, const float **packed)
{
    const int32_t Fi = Fv + Fe;
    void *tmp = nullptr;
    if (workspace(&tmp, sizeof(float) * (size_t)std::max<int64_t>(g->n_rows, 1) * Fi, 16)) return 1;
    if (g->n_rows > 0) {
        AMP_HIP(hipMemcpy2DAsync(tmp, sizeof(float) * Fi, a_x, sizeof(float) * Fv, sizeof(float) * Fv, (size_t)g->n_rows,
                                 hipMemcpyDeviceToDevice, stream()));
        AMP_HIP(hipMemcpy2DAsync((float *)tmp + Fv, sizeof(float) * Fi, a_e, sizeof(float) * Fe, sizeof(float) * Fe, (size_t)g->n_rows,
                                 hipMemcpyDeviceToDevice, stream()));
    }
    *packed = (const float *)tmp;
    return 0;
}
} // namespace amp

using namespace amp;

// MFMA route when both feature counts are matrix-sized; tiny layers (6-7-10 of msgpass_chemical) keep
// the bit-exact VALU kernels above
static inline bool duv_use_mfma(int Fi, int Fo, int64_t n) { return Fi >= 16 && Fo >= 16 && n >= 1024; }

extern "C" {

int athena_mp_duvenaud_update_act_fwd(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t min_deg,
                                      int32_t max_deg, const float *a, const float *weight, int32_t act, float *z)
{
    AMP_REQUIRE(g && a && weight && z && Fi > 0 && Fo > 0 && max_deg >= min_deg, "duvenaud_update_act_fwd: bad arguments");
    AMP_REQUIRE(act >= 0 && act <= ATHENA_MP_ACT_TANH, "duvenaud_update_act_fwd: unknown activation %d", act);
    if ((int64_t)g->n_rows * Fo == 0) return 0;
    if (duv_use_mfma(Fi, Fo, g->n_rows)) {
        if (duvenaud_buckets(g, min_deg, max_deg)) return 1;
        const int rc = duv_mfma_fwd(g, Fi, Fo, a, weight, act, z);
        if (rc >= 0) return rc;
    }
    if (int rc = athena_mp_duvenaud_update_fwd(g, Fi, Fo, min_deg, max_deg, a, weight, z)) return rc;
    return act == ATHENA_MP_ACT_NONE ? 0 : athena_mp_activation_fwd(act, (int64_t)g->n_rows * Fo, z, z);
}

int athena_mp_duvenaud_update_fwd(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t min_deg,
                                  int32_t max_deg, const float *a, const float *weight, float *c)
{
    AMP_REQUIRE(g && a && weight && c && Fi > 0 && Fo > 0 && max_deg >= min_deg, "duvenaud_update_fwd: bad arguments");
    int64_t total = (int64_t)g->n_rows * Fo;
    if (total == 0) return 0;
    if (duv_use_mfma(Fi, Fo, g->n_rows)) {
        if (duvenaud_buckets(g, min_deg, max_deg)) return 1;
        if (const int rc = duv_mfma_fwd(g, Fi, Fo, a, weight, ATHENA_MP_ACT_NONE, c); rc >= 0) return rc;
        for (int b = 0; b <= max_deg - min_deg; ++b) {
            const int64_t cnt = g->bucket_off[b + 1] - g->bucket_off[b];
            if (cnt == 0) continue;
            TiledArgs t;   // c[v,:] = W_d (a[v,:] / d): the divide happens on the A operand, as in the reference
            t.A = a; t.lda = Fi; t.a_idx = g->bucket_perm + g->bucket_off[b]; t.a_div = (float)(b + 1);
            t.B = weight + (size_t)b * Fo * Fi; t.ldb = Fo; t.b_nk = 0;   // W_d(o,i) flat o + Fo*i == [K=i][N=o]
            t.C = c; t.ldc = Fo; t.c_idx = t.a_idx; t.M = cnt; t.N = Fo; t.K = Fi;
            if (int rc = gemm_tiled(t)) return rc;
        }
        return 0;
    }
    hipLaunchKernelGGL(duv_update_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream(),
                       g->deg_row, min_deg, max_deg, Fi, Fo, (int64_t)g->n_rows, a, weight, c);
    AMP_LAUNCH_CHECK();
    return 0;
}

int athena_mp_duvenaud_update_bwd_a(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t min_deg,
                                    int32_t max_deg, const float *grad, const float *weight, float *da)
{
    AMP_REQUIRE(g && grad && weight && da && Fi > 0 && Fo > 0 && max_deg >= min_deg,
                "duvenaud_update_bwd_a: bad arguments");
    int64_t total = (int64_t)g->n_rows * Fi;
    if (total == 0) return 0;
    if (duv_use_mfma(Fi, Fo, g->n_rows)) {
        if (duvenaud_buckets(g, min_deg, max_deg)) return 1;
        if (const int rc = duv_mfma_bwd_a(g, Fi, Fo, grad, weight, da); rc >= 0) return rc;
        for (int b = 0; b <= max_deg - min_deg; ++b) {
            const int64_t cnt = g->bucket_off[b + 1] - g->bucket_off[b];
            if (cnt == 0) continue;
            TiledArgs t;   // da[v,i] = (sum_o g[v,o] W_d(o,i)) / d
            t.A = grad; t.lda = Fo; t.a_idx = g->bucket_perm + g->bucket_off[b];
            t.B = weight + (size_t)b * Fo * Fi; t.ldb = Fo; t.b_nk = 1;   // [N=i][K=o]
            t.c_div = (float)(b + 1);
            t.C = da; t.ldc = Fi; t.c_idx = t.a_idx; t.M = cnt; t.N = Fi; t.K = Fo;
            if (int rc = gemm_tiled(t)) return rc;
        }
        return 0;
    }
    hipLaunchKernelGGL(duv_update_bwd_a_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream(),
                       g->deg_row, min_deg, max_deg, Fi, Fo, (int64_t)g->n_rows, grad, weight, da);
    AMP_LAUNCH_CHECK();
    return 0;
}

int athena_mp_duvenaud_update_bwd_w(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t min_deg,
                                    int32_t max_deg, const float *grad, const float *a, float *dweight)
{
    AMP_REQUIRE(g && grad && a && dweight && Fi > 0 && Fo > 0 && max_deg >= min_deg,
                "duvenaud_update_bwd_w: bad arguments");
    const int nb = max_deg - min_deg + 1;
    const int n = nb * Fi * Fo;
    const int64_t N = g->n_rows;
    if (N == 0) {
        AMP_HIP(hipMemsetAsync(dweight, 0, sizeof(float) * n, stream()));
        return 0;
    }
    if (duv_use_mfma(Fi, Fo, N)) {
        if (duvenaud_buckets(g, min_deg, max_deg)) return 1;
        if (const int rc = duv_mfma_bwd_w(g, Fi, Fo, grad, a, dweight); rc >= 0) return rc;
        for (int b = 0; b < nb; ++b) {   // dW_d(o,i) = sum_{v in bucket} g[v,o] a[v,i] / d  -> [Fi][Fo] row-major
            const int64_t cnt = g->bucket_off[b + 1] - g->bucket_off[b];
            if (int rc = gemm_atb_tiled(a, Fi, grad, Fo, g->bucket_perm + g->bucket_off[b], (float)(b + 1), cnt, Fi, Fo,
                                        dweight + (size_t)b * Fo * Fi, false))
                return rc;
        }
        return 0;
    }
    int chunks = (int)std::min<int64_t>((N + 1023) / 1024, 512);
    int64_t rpc = (N + chunks - 1) / chunks;
    chunks = (int)((N + rpc - 1) / rpc);
    void *ws = nullptr;
    if (workspace(&ws, sizeof(float) * (size_t)chunks * n, 2)) return 1;
    hipLaunchKernelGGL(duv_update_bwd_w_kernel, dim3(chunks, nb), dim3(256), 0, stream(), g->deg_row, min_deg,
                       max_deg, Fi, Fo, N, rpc, grad, a, (float *)ws);
    AMP_LAUNCH_CHECK();
    hipLaunchKernelGGL(slab_reduce_kernel2, dim3((n + 255) / 256), dim3(256), 0, stream(), (const float *)ws,
                       chunks, n, dweight);
    AMP_LAUNCH_CHECK();
    return 0;
}

/* z = act(duvenaud_update(a)) AND the readout's per-vertex p = softmax_over_outputs(R z) (update_readout_duvenaud,
 * athena_duvenaud_msgpass_layer.f90:838-855, default softmax readout) from one launch: the activated rows feed the logits
 * product straight from the accumulators, so the readout does not read z back.  The per-graph sums stay with
 * athena_mp_segment_sum (p is [n_rows, O]).  Shapes outside the fused kernel run update_act_fwd and the readout launch. */
int athena_mp_duvenaud_update_readout_fwd(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t min_deg, int32_t max_deg,
                                          const float *a, const float *weight, int32_t act, float *z, int32_t O,
                                          const float *R, float *p)
{
    AMP_REQUIRE(g && a && weight && z && R && p && Fi > 0 && Fo > 0 && O > 0 && max_deg >= min_deg,
                "duvenaud_update_readout_fwd: bad arguments");
    AMP_REQUIRE(act >= 0 && act <= ATHENA_MP_ACT_TANH, "duvenaud_update_readout_fwd: unknown activation %d", act);
    if (g->n_rows == 0) return 0;
    if (duv_use_mfma(Fi, Fo, g->n_rows) && max_deg - min_deg + 1 <= 32) {
        if (duvenaud_buckets(g, min_deg, max_deg)) return 1;
        if (const int rc = duv_mfma_fwd_readout(g, Fi, Fo, a, weight, act, z, R, O, p); rc >= 0) return rc;
    }
    if (int rc = athena_mp_duvenaud_update_act_fwd(g, Fi, Fo, min_deg, max_deg, a, weight, act, z)) return rc;
    // S = 0: logits + softmax only (the segment pointer is not read)
    return athena_mp_duvenaud_readout_fwd(g->n_rows, Fo, O, 0, (const int32_t *)g->rowptr, z, R, p, p, 0);
}

/* athena_mp_duvenaud_update_readout_fwd with a SPLIT: a_x [n_rows, Fv] (the neighbour sums of the vertex features,
 * athena_mp_duvenaud_propagate_fwd with Fe = 0) and a_e [n_rows, Fe] (those of the edge features, athena_mp_duvenaud_propagate_fwd
 * with Fv = 0 -- the same at every time step of a layer, so a layer gathers it once).  One launch at F_v = F_o = 64, F_e <= 16,
 * O <= 16 (duv_rows_wide_kernel<5, 4, readout, split>); other shapes pack a into a workspace first. */
int athena_mp_duvenaud_update_readout_fwd_split(const athena_mp_graph *g, int32_t Fv, int32_t Fe, int32_t Fo, int32_t min_deg,
                                                int32_t max_deg, const float *a_x, const float *a_e, const float *weight,
                                                int32_t act, float *z, int32_t O, const float *R, float *p)
{
    AMP_REQUIRE(g && a_x && a_e && weight && z && R && p && Fv > 0 && Fe > 0 && Fo > 0 && O > 0 && max_deg >= min_deg,
                "duvenaud_update_readout_fwd_split: bad arguments");
    AMP_REQUIRE(act >= 0 && act <= ATHENA_MP_ACT_TANH, "duvenaud_update_readout_fwd_split: unknown activation %d", act);
    if (g->n_rows == 0) return 0;
    const int32_t Fi = Fv + Fe;
    if (Fv == 64 && duv_use_mfma(Fi, Fo, g->n_rows) && max_deg - min_deg + 1 <= 32) {
        if (duvenaud_buckets(g, min_deg, max_deg)) return 1;
        if (const int rc = duv_mfma_fwd_readout(g, Fi, Fo, a_x, weight, act, z, R, O, p, a_e); rc >= 0) return rc;
    }
    const float *a = nullptr;
    if (duv_pack_a(g, Fv, Fe, a_x, a_e, &a)) return 1;
    return athena_mp_duvenaud_update_readout_fwd(g, Fi, Fo, min_deg, max_deg, a, weight, act, z, O, R, p);
}

/* both reverse products of duvenaud_update from one pass over the upstream gradient: da (w.r.t. the aggregated features)
 * and dweight.  Same results as the two entry points above; where the fused kernel does not cover the shape they run. */
int athena_mp_duvenaud_update_bwd(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t min_deg, int32_t max_deg,
                                  const float *grad, const float *a, const float *weight, float *da, float *dweight)
{
    AMP_REQUIRE(g && grad && a && weight && da && dweight && Fi > 0 && Fo > 0 && max_deg >= min_deg,
                "duvenaud_update_bwd: bad arguments");
    if (g->n_rows > 0 && duv_use_mfma(Fi, Fo, g->n_rows) && max_deg - min_deg + 1 <= 32) {
        if (duvenaud_buckets(g, min_deg, max_deg)) return 1;
        if (const int rc = duv_mfma_bwd(g, Fi, Fo, grad, a, weight, da, dweight); rc >= 0) return rc;
    }
    if (int rc = athena_mp_duvenaud_update_bwd_w(g, Fi, Fo, min_deg, max_deg, grad, a, dweight)) return rc;
    return athena_mp_duvenaud_update_bwd_a(g, Fi, Fo, min_deg, max_deg, grad, weight, da);
}

/* the same pair with da SPLIT where it is written: da_x [n_rows, Fv] (the part athena_mp_duvenaud_propagate_bwd_x gathers, with
 * Fe = 0) and da_e [n_rows, Fe] (the part _bwd_e gathers, with Fv = 0).  One launch for F_v = 64 and the widths of the fused
 * kernel; any other shape computes the packed da into a workspace and copies the two parts out. */
int athena_mp_duvenaud_update_bwd_split(const athena_mp_graph *g, int32_t Fv, int32_t Fe, int32_t Fo, int32_t min_deg, int32_t max_deg,
                                        const float *grad, const float *a, const float *weight, float *da_x, float *da_e,
                                        float *dweight)
{
    AMP_REQUIRE(g && grad && a && weight && da_x && da_e && dweight && Fv > 0 && Fe > 0 && Fo > 0 && max_deg >= min_deg,
                "duvenaud_update_bwd_split: bad arguments");
    const int32_t Fi = Fv + Fe;
    if (g->n_rows > 0 && Fv == 64 && duv_use_mfma(Fi, Fo, g->n_rows) && max_deg - min_deg + 1 <= 32) {
        if (duvenaud_buckets(g, min_deg, max_deg)) return 1;
        if (const int rc = duv_mfma_bwd(g, Fi, Fo, grad, a, weight, da_x, dweight, da_e); rc >= 0) return rc;
    }
    void *tmp = nullptr;
    if (workspace(&tmp, sizeof(float) * (size_t)std::max<int64_t>(g->n_rows, 1) * Fi, 11)) return 1;
    if (int rc = athena_mp_duvenaud_update_bwd(g, Fi, Fo, min_deg, max_deg, grad, a, weight, (float *)tmp, dweight)) return rc;
    if (g->n_rows > 0) {
        AMP_HIP(hipMemcpy2DAsync(da_x, sizeof(float) * Fv, tmp, sizeof(float) * Fi, sizeof(float) * Fv, (size_t)g->n_rows,
                                 hipMemcpyDeviceToDevice, stream()));
        AMP_HIP(hipMemcpy2DAsync(da_e, sizeof(float) * Fe, (const float *)tmp + Fv, sizeof(float) * Fi, sizeof(float) * Fe,
                                 (size_t)g->n_rows, hipMemcpyDeviceToDevice, stream()));
    }
    return 0;
}

int athena_mp_softmax_segsum_fwd(int32_t O, int64_t N, int32_t S, const int32_t *seg, const float *logits,
                                 float *p, float *out, int32_t accumulate)
{
    AMP_REQUIRE(O > 0 && N >= 0 && S >= 0 && seg && logits && p && out, "softmax_segsum_fwd: bad arguments");
    if (N > 0) {
        hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream(), O, N,
                           logits, p);
        AMP_LAUNCH_CHECK();
    }
    if (S > 0) {
        hipLaunchKernelGGL(segsum_kernel, dim3((unsigned)(((int64_t)S * O + 255) / 256)), dim3(256), 0, stream(),
                           O, S, seg, p, out, accumulate);
        AMP_LAUNCH_CHECK();
    }
    return 0;
}

int athena_mp_softmax_segsum_bwd(int32_t O, int64_t N, int32_t S, const int32_t *seg, const float *p,
                                 const float *gout, float *dlogits)
{
    AMP_REQUIRE(O > 0 && N >= 0 && S > 0 && seg && p && gout && dlogits, "softmax_segsum_bwd: bad arguments");
    if (N == 0) return 0;
    hipLaunchKernelGGL(softmax_segsum_bwd_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream(), O, N,
                       S, seg, p, gout, dlogits);
    AMP_LAUNCH_CHECK();
    return 0;
}

} // extern "C"
