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

inline dim3 grid_for(int64_t n) { return dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048 * 4)); }

} // namespace

using namespace amp;

extern "C" {

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

} // extern "C"
