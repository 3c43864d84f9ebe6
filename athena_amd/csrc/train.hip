// Device-resident tail of a train step: loss, gradient clipping and the flat-parameter
// optimiser update, so that parameters, gradients and optimiser state never leave HBM between steps.
//
//   compute_mse       athena_loss.f90:393-430        mean((p-e)^2)/2 and its gradient
//   apply_clip        athena_clipper.f90:165-210     min/max clamp, then global L2-norm scaling
//   regularise_*      athena_regulariser.f90:85-138  folded into the optimiser kernels
//   minimise_sgd      athena_optimiser.f90:634-673
//   minimise_adam     athena_optimiser.f90:1027-1091
//   network%update    athena_network_sub.f90:2816-2929 (the order: clip, then minimise on flat vectors)
//
// Element-wise kernels keep the reference's operation order (no contraction: build flag) and IEEE
// sqrt / divide, so they agree bit for bit with the strict-fp32 restatement; the two reductions
// (sum of squares) are two-stage with a fixed order -- deterministic, equal to the sequential sum
// within rounding.
#include <algorithm>

#include "common.h"

namespace {

constexpr int kRedBlocks = 1024;

template <bool DIFF>
__global__ __launch_bounds__(256) void sumsq_partial_kernel(int64_t n, const float *__restrict__ a,
                                                            const float *__restrict__ b, float *__restrict__ part,
                                                            float *__restrict__ dout, float dscale)
{
    __shared__ float red[256];
    float s = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float d = a[i];
        if constexpr (DIFF) {
            d = d - b[i];
            if (dout) dout[i] = d / dscale;
        }
        s = s + d * d;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// mode 0: out = sum;  mode 1 (mse): out = sum / n / 2;  mode 2 (clip): out = min(1, norm / sqrt(sum))
__global__ __launch_bounds__(256) void sumsq_final_kernel(int n_part, const float *__restrict__ part, int mode,
                                                          float arg, float *__restrict__ out)
{
    __shared__ float red[256];
    float s = 0.0f;
    for (int i = threadIdx.x; i < n_part; i += 256) s = s + part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float r = red[0];
        if (mode == 1) r = r / arg / 2.0f;
        else if (mode == 2) r = fminf(1.0f, arg / sqrtf(r + 0.0f));
        out[0] = r;
    }
}

__global__ void clamp_kernel(int64_t n, float *__restrict__ g, float mn, float mx)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) g[i] = fmaxf(mn, fminf(mx, g[i]));
}

__global__ void scale_if_kernel(int64_t n, float *__restrict__ g, const float *__restrict__ scale)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float s = scale[0];
    if (i < n && s < 1.0f) g[i] = g[i] * s;
}

__device__ __forceinline__ float reg_term(int kind, float l1, float l2, float p, float lr)
{
    const float sg = signbit(p) ? -1.0f : 1.0f;
    switch (kind) {
    case 1: return lr * l1 * sg;
    case 2: return lr * 2.0f * l2 * p;
    case 3: return lr * (l1 * sg + 2.0f * l2 * p);
    default: return 0.0f;
    }
}

__global__ void sgd_kernel(int64_t n, float lr, float momentum, int nesterov, int reg_kind, float l1, float l2,
                           float *__restrict__ param, float *__restrict__ grad, float *__restrict__ velocity)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float g = grad[i], p = param[i];
    if (reg_kind) g = g + reg_term(reg_kind, l1, l2, p, lr);
    g = -lr * g;
    float v;
    if (momentum > 1.0e-8f) {
        v = momentum * velocity[i] + g;
        p = nesterov ? p + momentum * v + g : p + v;
    } else {
        v = g;
        p = p + v;
    }
    velocity[i] = v;
    param[i] = p;
    grad[i] = g;
}

__global__ void adam_kernel(int64_t n, float lr, float beta1, float beta2, float epsilon, float bc1, float bc2,
                            int reg_kind, float l1, float l2, int decoupled, float *__restrict__ param,
                            float *__restrict__ grad, float *__restrict__ m, float *__restrict__ v)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float g = grad[i], p = param[i];
    if (reg_kind) g = g + reg_term(reg_kind, l1, l2, p, lr);
    grad[i] = g;
    const float mi = beta1 * m[i] + (1.0f - beta1) * g;
    const float vi = beta2 * v[i] + (1.0f - beta2) * g * g;
    m[i] = mi;
    v[i] = vi;
    const float m_hat = mi / bc1, v_hat = vi / bc2;
    if (reg_kind == 2 && decoupled) {
        p = p - lr * l2 * p;
        p = p - lr * (m_hat / (sqrtf(v_hat) + epsilon));
    } else if (reg_kind == 2) {
        p = p - lr * ((m_hat + l2 * p) / (sqrtf(v_hat) + epsilon));
    } else {
        p = p - lr * (m_hat / (sqrtf(v_hat) + epsilon));
    }
    param[i] = p;
}

// real ** integer as compilers lower it: binary powering by repeated multiplication
float powi(float x, int n)
{
    float r = (n & 1) ? x : 1.0f;
    for (n >>= 1; n; n >>= 1) {
        x = x * x;
        if (n & 1) r = r * x;
    }
    return r;
}

inline dim3 grid1(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

} // namespace

using namespace amp;

extern "C" {

int athena_mp_mse_loss(int64_t n, const float *pred, const float *expected, float *loss_dev, float *dpred)
{
    AMP_REQUIRE(n > 0 && pred && expected && loss_dev, "mse_loss: bad arguments");
    void *part = nullptr;
    if (workspace(&part, sizeof(float) * kRedBlocks, 2)) return 1;
    const int nb = (int)std::min<int64_t>(kRedBlocks, (n + 255) / 256);
    hipLaunchKernelGGL((sumsq_partial_kernel<true>), dim3(nb), dim3(256), 0, stream(), n, pred, expected,
                       (float *)part, dpred, (float)n);
    AMP_LAUNCH_CHECK();
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, stream(), nb, (const float *)part, 1, (float)n,
                       loss_dev);
    AMP_LAUNCH_CHECK();
    return 0;
}

int athena_mp_clip(int64_t n, float *grad, int32_t l_min_max, float clip_min, float clip_max, int32_t l_norm,
                   float clip_norm)
{
    AMP_REQUIRE(n >= 0 && (n == 0 || grad), "clip: bad arguments");
    if (n == 0) return 0;
    if (l_min_max) {
        hipLaunchKernelGGL(clamp_kernel, grid1(n), dim3(256), 0, stream(), n, grad, clip_min, clip_max);
        AMP_LAUNCH_CHECK();
    }
    if (l_norm) {
        void *part = nullptr;
        if (workspace(&part, sizeof(float) * (kRedBlocks + 1), 2)) return 1;
        const int nb = (int)std::min<int64_t>(kRedBlocks, (n + 255) / 256);
        float *scale = (float *)part + kRedBlocks;
        hipLaunchKernelGGL((sumsq_partial_kernel<false>), dim3(nb), dim3(256), 0, stream(), n, (const float *)grad,
                           (const float *)nullptr, (float *)part, (float *)nullptr, 1.0f);
        AMP_LAUNCH_CHECK();
        hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, stream(), nb, (const float *)part, 2,
                           clip_norm, scale);
        AMP_LAUNCH_CHECK();
        hipLaunchKernelGGL(scale_if_kernel, grid1(n), dim3(256), 0, stream(), n, grad, (const float *)scale);
        AMP_LAUNCH_CHECK();
    }
    return 0;
}

int athena_mp_sgd_step(int64_t n, float lr, float momentum, int32_t nesterov, int32_t reg_kind, float l1, float l2,
                       float *param, float *grad, float *velocity)
{
    AMP_REQUIRE(n >= 0 && (n == 0 || (param && grad && velocity)), "sgd_step: bad arguments");
    AMP_REQUIRE(reg_kind >= 0 && reg_kind <= 3, "sgd_step: unknown regulariser %d", reg_kind);
    if (n == 0) return 0;
    hipLaunchKernelGGL(sgd_kernel, grid1(n), dim3(256), 0, stream(), n, lr, momentum, nesterov, reg_kind, l1, l2,
                       param, grad, velocity);
    AMP_LAUNCH_CHECK();
    return 0;
}

int athena_mp_adam_step(int64_t n, float lr, float beta1, float beta2, float epsilon, int32_t iter,
                        int32_t reg_kind, float l1, float l2, int32_t decoupled, float *param, float *grad,
                        float *m, float *v)
{
    AMP_REQUIRE(n >= 0 && (n == 0 || (param && grad && m && v)), "adam_step: bad arguments");
    AMP_REQUIRE(reg_kind >= 0 && reg_kind <= 3, "adam_step: unknown regulariser %d", reg_kind);
    AMP_REQUIRE(iter >= 1, "adam_step: iteration counter must be >= 1 (bias correction divides by 1 - beta**iter)");
    if (n == 0) return 0;
    const float bc1 = 1.0f - powi(beta1, iter), bc2 = 1.0f - powi(beta2, iter);
    hipLaunchKernelGGL(adam_kernel, grid1(n), dim3(256), 0, stream(), n, lr, beta1, beta2, epsilon, bc1, bc2,
                       reg_kind, l1, l2, decoupled, param, grad, m, v);
    AMP_LAUNCH_CHECK();
    return 0;
}

} // extern "C"
