// Element-wise brackets of the path: activations (athena_activation_{none,relu,sigmoid,tanh}.f90
// `apply`), their reverse-mode factors, and y += alpha*x (diffstruc operator(+) at
// athena_graph_nop_layer.f90:764 / athena_duvenaud_msgpass_layer.f90:849-852).  Pure HBM streams.
#include <algorithm>

#include "common.h"

namespace {

__device__ __forceinline__ float act_f(float t, int act)
{
    switch (act) {
    case ATHENA_MP_ACT_RELU: return t > 0.0f ? t : 0.0f;
    case ATHENA_MP_ACT_SIGMOID: return 1.0f / (1.0f + expf(-t));
    case ATHENA_MP_ACT_TANH: return tanhf(t);
    default: return t;
    }
}
__device__ __forceinline__ float act_b(float y, float g, int act)
{
    switch (act) {
    case ATHENA_MP_ACT_RELU: return y > 0.0f ? g : 0.0f;
    case ATHENA_MP_ACT_SIGMOID: return g * y * (1.0f - y);
    case ATHENA_MP_ACT_TANH: return g * (1.0f - y * y);
    default: return g;
    }
}

__global__ void act_fwd_kernel(int act, int64_t n, const float *__restrict__ z, float *__restrict__ y)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = act_f(z[i], act);
}
__global__ void act_bwd_kernel(int act, int64_t n, const float *__restrict__ y, const float *__restrict__ g,
                               float *__restrict__ dz)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dz[i] = act_b(y[i], g[i], act);
}
__global__ void axpy_kernel(int64_t n, float alpha, const float *__restrict__ x, float *__restrict__ y)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = y[i] + alpha * x[i];
}

__global__ void swish_fwd_kernel(int64_t n, float beta, const float *__restrict__ x, float *__restrict__ y)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = x[i] * (1.0f / (1.0f + expf(-beta * x[i])));
}
__global__ void swish_bwd_kernel(int64_t n, float beta, const float *__restrict__ x, const float *__restrict__ g,
                                 float *__restrict__ dx)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float e = expf(beta * x[i]);
        dx[i] = g[i] * e * (beta * x[i] + e + 1.0f) / ((e + 1.0f) * (e + 1.0f));
    }
}

// parameterised activations (athena_activation_{linear,relu,sigmoid,tanh,leaky_relu,selu,gaussian,piecewise}.f90
// `apply` with their scale / threshold / alpha / lambda / sigma / mu / gradient / limit attributes).  The reverse
// factor is evaluated at the INPUT x, as the autodiff nodes of these ops do.
__device__ __forceinline__ float actp_f(int kind, float x, float p0, float p1)
{
    switch (kind) {
    case ATHENA_MP_ACTP_RELU: return fmaxf(x, p0);                       // max(val, threshold)
    case ATHENA_MP_ACTP_SIGMOID: return 1.0f / (1.0f + expf(-x));
    case ATHENA_MP_ACTP_TANH: return tanhf(x);
    case ATHENA_MP_ACTP_LEAKY_RELU: return fmaxf(x * p0, x);             // max(val*alpha, val)
    case ATHENA_MP_ACTP_SELU: return x > 0.0f ? x * p1 : (expf(x) - 1.0f) * p0 * p1;
    case ATHENA_MP_ACTP_GAUSSIAN: {
        const float t = (x - p1) / p0;
        return 1.0f / (sqrtf(6.283185307179586f) * p0) * expf(-0.5f * t * t);
    }
    case ATHENA_MP_ACTP_PIECEWISE:                                       // piecewise_array :216-254
        return x >= p1 ? p0 * (x - p1) + p1 : (x <= -p1 ? p0 * (x + p1) - p1 : x);
    default: return x;
    }
}
__device__ __forceinline__ float actp_b(int kind, float x, float g, float p0, float p1)
{
    switch (kind) {
    case ATHENA_MP_ACTP_RELU: return x > p0 ? g : 0.0f;
    case ATHENA_MP_ACTP_SIGMOID: {
        const float y = 1.0f / (1.0f + expf(-x));
        return g * y * (1.0f - y);
    }
    case ATHENA_MP_ACTP_TANH: {
        const float y = tanhf(x);
        return g * (1.0f - y * y);
    }
    case ATHENA_MP_ACTP_LEAKY_RELU: return x * p0 > x ? g * p0 : g;
    case ATHENA_MP_ACTP_SELU: return x > 0.0f ? g * p1 : g * (expf(x) * p0 * p1);
    case ATHENA_MP_ACTP_GAUSSIAN: {
        const float t = (x - p1) / p0;
        const float f = 1.0f / (sqrtf(6.283185307179586f) * p0) * expf(-0.5f * t * t);
        return g * (-(x - p1) / (p0 * p0) * f);
    }
    case ATHENA_MP_ACTP_PIECEWISE:                                       // get_partial_piecewise_val :275-290, as written
        return (x <= p1 || x >= -p1) ? g : g * p0;
    default: return g;
    }
}
__global__ void actp_fwd_kernel(int kind, int64_t n, float scale, float p0, float p1, const float *__restrict__ x,
                                float *__restrict__ y)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = actp_f(kind, x[i], p0, p1) * scale;
}
__global__ void actp_bwd_kernel(int kind, int64_t n, float scale, float p0, float p1, const float *__restrict__ x,
                                const float *__restrict__ g, float *__restrict__ dx)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dx[i] = actp_b(kind, x[i], g[i] * scale, p0, p1);
}

// softmax over the features of each vertex, G lanes per vertex (G = 4, 16 or 64 by feature count)
template <int G>
__device__ __forceinline__ float group_max(float v)
{
#pragma unroll
    for (int w = G / 2; w > 0; w >>= 1) v = fmaxf(v, __shfl_xor(v, w));
    return v;
}
template <int G>
__device__ __forceinline__ float group_sum(float v)
{
#pragma unroll
    for (int w = G / 2; w > 0; w >>= 1) v = v + __shfl_xor(v, w);
    return v;
}
template <int G>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(int64_t N, int F, const float *__restrict__ z,
                                                          float *__restrict__ y)
{
    const int l = threadIdx.x % G;
    // a vertex is owned by one G-lane group, so the loop exit is uniform across the lanes that shuffle together
    for (int64_t v = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G; v < N; v += (int64_t)gridDim.x * (256 / G)) {
        constexpr bool ok = true;
        const float *zv = z + (ok ? v : 0) * F;
        float m = -INFINITY;
        for (int i = l; i < F; i += G) m = fmaxf(m, zv[i]);
        m = group_max<G>(m);
        float s = 0.0f;
        for (int i = l; i < F; i += G) s = s + expf(zv[i] - m);
        s = group_sum<G>(s);
        if (ok)
            for (int i = l; i < F; i += G) y[v * F + i] = expf(zv[i] - m) / s;
    }
}
template <int G>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(int64_t N, int F, const float *__restrict__ y,
                                                          const float *__restrict__ g, float *__restrict__ dz)
{
    const int l = threadIdx.x % G;
    // a vertex is owned by one G-lane group, so the loop exit is uniform across the lanes that shuffle together
    for (int64_t v = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G; v < N; v += (int64_t)gridDim.x * (256 / G)) {
        constexpr bool ok = true;
        const float *yv = y + (ok ? v : 0) * F, *gv = g + (ok ? v : 0) * F;
        float dot = 0.0f;
        for (int i = l; i < F; i += G) dot = dot + yv[i] * gv[i];
        dot = group_sum<G>(dot);
        if (ok)
            for (int i = l; i < F; i += G) dz[v * F + i] = yv[i] * gv[i] - yv[i] * dot;   // the reference's form
    }
}

__global__ void concat_fwd_kernel(int64_t N, int Fa, int Fb, const float *__restrict__ a, const float *__restrict__ b,
                                  float *__restrict__ out)
{
    const int F = Fa + Fb;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N * F; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = t / F;
        const int i = (int)(t - v * F);
        out[t] = i < Fa ? a[v * Fa + i] : b[v * Fb + (i - Fa)];
    }
}
__global__ void concat_bwd_kernel(int64_t N, int Fa, int Fb, const float *__restrict__ gout, float *__restrict__ da,
                                  float *__restrict__ db)
{
    const int F = Fa + Fb;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N * F; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = t / F;
        const int i = (int)(t - v * F);
        if (i < Fa) {
            if (da) da[v * Fa + i] = gout[t];
        } else if (db) {
            db[v * Fb + (i - Fa)] = gout[t];
        }
    }
}

inline dim3 grid_for(int64_t n) { return dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048 * 4)); }

// one 16-byte element per thread in launch order: the access pattern that reaches this HBM's streaming ceiling (6.2 TB/s
// on the pool's boxes against 4.4 - 5.0 for persistent grid-stride loops; profiles/r03_ubench_stream_rows.txt)
typedef float copy_v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy16_kernel(const copy_v4f *__restrict__ src, copy_v4f *__restrict__ dst, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void copy1_kernel(const char *__restrict__ src, char *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

} // namespace

using namespace amp;

extern "C" {

int athena_mp_device_copy(void *dst, const void *src, uint64_t bytes)
{
    AMP_REQUIRE(bytes == 0 || (dst && src), "device_copy: null pointer");
    AMP_REQUIRE((((uintptr_t)dst | (uintptr_t)src) & 15) == 0, "device_copy: pointers must be 16-byte aligned");
    const size_t n16 = (size_t)(bytes / 16), tail = (size_t)(bytes % 16);
    AMP_REQUIRE((n16 + 255) / 256 < ((size_t)1 << 31), "device_copy: %llu bytes exceed one launch", (unsigned long long)bytes);
    if (n16) {
        hipLaunchKernelGGL(copy16_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, stream(), (const copy_v4f *)src,
                           (copy_v4f *)dst, n16);
        AMP_LAUNCH_CHECK();
    }
    if (tail) {
        hipLaunchKernelGGL(copy1_kernel, dim3(1), dim3(256), 0, stream(), (const char *)src + 16 * n16, (char *)dst + 16 * n16, tail);
        AMP_LAUNCH_CHECK();
    }
    return 0;
}

int athena_mp_activation_fwd(int32_t act, int64_t n, const float *z, float *y)
{
    AMP_REQUIRE(n >= 0 && (n == 0 || (z && y)), "activation_fwd: bad arguments");
    AMP_REQUIRE(act >= 0 && act <= ATHENA_MP_ACT_TANH, "activation_fwd: unknown activation %d", act);
    if (n == 0) return 0;
    hipLaunchKernelGGL(act_fwd_kernel, grid_for(n), dim3(256), 0, stream(), act, n, z, y);
    AMP_LAUNCH_CHECK();
    return 0;
}
int athena_mp_activation_bwd(int32_t act, int64_t n, const float *y, const float *g, float *dz)
{
    AMP_REQUIRE(n >= 0 && (n == 0 || (y && g && dz)), "activation_bwd: bad arguments");
    AMP_REQUIRE(act >= 0 && act <= ATHENA_MP_ACT_TANH, "activation_bwd: unknown activation %d", act);
    if (n == 0) return 0;
    hipLaunchKernelGGL(act_bwd_kernel, grid_for(n), dim3(256), 0, stream(), act, n, y, g, dz);
    AMP_LAUNCH_CHECK();
    return 0;
}
int athena_mp_axpy(int64_t n, float alpha, const float *x, float *y)
{
    AMP_REQUIRE(n >= 0 && (n == 0 || (x && y)), "axpy: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(axpy_kernel, grid_for(n), dim3(256), 0, stream(), n, alpha, x, y);
    AMP_LAUNCH_CHECK();
    return 0;
}


/* swish_array / get_partial_swish_val, athena_diffstruc_extd_sub.f90:424-492 (reverse pass takes the INPUT) */
int athena_mp_swish_fwd(int64_t n, float beta, const float *x, float *y)
{
    AMP_REQUIRE(n >= 0 && (n == 0 || (x && y)), "swish_fwd: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(swish_fwd_kernel, grid_for(n), dim3(256), 0, stream(), n, beta, x, y);
    AMP_LAUNCH_CHECK();
    return 0;
}
int athena_mp_swish_bwd(int64_t n, float beta, const float *x, const float *g, float *dx)
{
    AMP_REQUIRE(n >= 0 && (n == 0 || (x && g && dx)), "swish_bwd: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(swish_bwd_kernel, grid_for(n), dim3(256), 0, stream(), n, beta, x, g, dx);
    AMP_LAUNCH_CHECK();
    return 0;
}

/* parameterised activations, athena_activation_*.f90 `apply` (scale applied to the output) */
int athena_mp_activation_param_fwd(int32_t kind, int64_t n, float scale, float p0, float p1, const float *x, float *y)
{
    AMP_REQUIRE(n >= 0 && (n == 0 || (x && y)), "activation_param_fwd: bad arguments");
    AMP_REQUIRE(kind >= 0 && kind <= ATHENA_MP_ACTP_PIECEWISE, "activation_param_fwd: unknown activation %d", kind);
    if (n == 0) return 0;
    hipLaunchKernelGGL(actp_fwd_kernel, grid_for(n), dim3(256), 0, stream(), kind, n, scale, p0, p1, x, y);
    AMP_LAUNCH_CHECK();
    return 0;
}
int athena_mp_activation_param_bwd(int32_t kind, int64_t n, float scale, float p0, float p1, const float *x,
                                   const float *g, float *dx)
{
    AMP_REQUIRE(n >= 0 && (n == 0 || (x && g && dx)), "activation_param_bwd: bad arguments");
    AMP_REQUIRE(kind >= 0 && kind <= ATHENA_MP_ACTP_PIECEWISE, "activation_param_bwd: unknown activation %d", kind);
    if (n == 0) return 0;
    hipLaunchKernelGGL(actp_bwd_kernel, grid_for(n), dim3(256), 0, stream(), kind, n, scale, p0, p1, x, g, dx);
    AMP_LAUNCH_CHECK();
    return 0;
}

/* softmax activation (apply_softmax -> softmax(val, dim=2), athena_activation_softmax.f90:183-203;
 * athena_diffstruc_extd_sub.f90:295-379): over the F features of each of the N vertices */
int athena_mp_softmax_fwd(int64_t N, int32_t F, const float *z, float *y)
{
    AMP_REQUIRE(N >= 0 && F > 0 && (N == 0 || (z && y)), "softmax_fwd: bad arguments");
    if (N == 0) return 0;
#define AMP_SM(G_)                                                                                             \
    hipLaunchKernelGGL((softmax_fwd_kernel<G_>), grid_for(N *G_), dim3(256), 0, stream(), N, F, z, y)
    if (F <= 8) AMP_SM(4);
    else if (F <= 48) AMP_SM(16);
    else AMP_SM(64);
#undef AMP_SM
    AMP_LAUNCH_CHECK();
    return 0;
}
int athena_mp_softmax_bwd(int64_t N, int32_t F, const float *y, const float *g, float *dz)
{
    AMP_REQUIRE(N >= 0 && F > 0 && (N == 0 || (y && g && dz)), "softmax_bwd: bad arguments");
    if (N == 0) return 0;
#define AMP_SM(G_)                                                                                             \
    hipLaunchKernelGGL((softmax_bwd_kernel<G_>), grid_for(N *G_), dim3(256), 0, stream(), N, F, y, g, dz)
    if (F <= 8) AMP_SM(4);
    else if (F <= 48) AMP_SM(16);
    else AMP_SM(64);
#undef AMP_SM
    AMP_LAUNCH_CHECK();
    return 0;
}

/* 'concatenate' merge of two layer inputs (network%add(..., operator='concatenate')): per-vertex feature
 * stacking, a first; the reverse pass splits the gradient (either output may be NULL) */
int athena_mp_concat_fwd(int64_t N, int32_t Fa, int32_t Fb, const float *a, const float *b, float *out)
{
    AMP_REQUIRE(N >= 0 && Fa > 0 && Fb > 0 && (N == 0 || (a && b && out)), "concat_fwd: bad arguments");
    if (N == 0) return 0;
    hipLaunchKernelGGL(concat_fwd_kernel, grid_for(N * (Fa + Fb)), dim3(256), 0, stream(), N, Fa, Fb, a, b, out);
    AMP_LAUNCH_CHECK();
    return 0;
}
int athena_mp_concat_bwd(int64_t N, int32_t Fa, int32_t Fb, const float *gout, float *da, float *db)
{
    AMP_REQUIRE(N >= 0 && Fa > 0 && Fb > 0 && (N == 0 || gout), "concat_bwd: bad arguments");
    if (N == 0) return 0;
    hipLaunchKernelGGL(concat_bwd_kernel, grid_for(N * (Fa + Fb)), dim3(256), 0, stream(), N, Fa, Fb, gout, da, db);
    AMP_LAUNCH_CHECK();
    return 0;
}

} // extern "C"
