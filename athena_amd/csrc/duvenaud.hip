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

using namespace amp;

extern "C" {

int athena_mp_duvenaud_update_fwd(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t min_deg,
                                  int32_t max_deg, const float *a, const float *weight, float *c)
{
    AMP_REQUIRE(g && a && weight && c && Fi > 0 && Fo > 0 && max_deg >= min_deg, "duvenaud_update_fwd: bad arguments");
    int64_t total = (int64_t)g->n_rows * Fo;
    if (total == 0) return 0;
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
