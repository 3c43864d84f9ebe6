// Micro-benchmark: what streaming rates does this box's HBM sustain?  1 GiB vectors (beyond the 256 MiB Infinity Cache).
//   copy (read + write), read-only reduction, write-only fill; plain / unrolled x4 / non-temporal forms; grid sizes.
// Build: hipcc --offload-arch=gfx950 -O3 stream_copy.hip -o stream_copy
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int UN, bool NT>
__global__ __launch_bounds__(256) void copy_k(const v4f *__restrict__ x, v4f *__restrict__ y, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UN - 1) * stride < n4; i += UN * stride) {
        v4f v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) v[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], y + i + u * stride);
            else y[i + u * stride] = v[u];
        }
    }
    for (; i < n4; i += stride) y[i] = x[i];
}
// non-persistent form: workgroup b copies the contiguous chunk [b * 256 * E, (b + 1) * 256 * E) and exits
template <int E>
__global__ __launch_bounds__(256) void copy_chunk_k(const v4f *__restrict__ x, v4f *__restrict__ y, size_t n4)
{
    const size_t base = (size_t)blockIdx.x * 256 * E + threadIdx.x;
    v4f v[E > 8 ? 8 : E];
    for (int e0 = 0; e0 < E; e0 += 8) {
#pragma unroll
        for (int e = 0; e < (E > 8 ? 8 : E); ++e) v[e] = x[base + (size_t)(e0 + e) * 256];
#pragma unroll
        for (int e = 0; e < (E > 8 ? 8 : E); ++e) y[base + (size_t)(e0 + e) * 256] = v[e];
    }
}
template <int UN>
__global__ __launch_bounds__(256) void read_k(const v4f *__restrict__ x, float *out, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    v4f acc = {0, 0, 0, 0};
    for (; i + (UN - 1) * stride < n4; i += UN * stride) {
        v4f v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < UN; ++u) acc = acc + v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.6789f) out[0] = acc[0];
}
template <bool NT>
__global__ __launch_bounds__(256) void fill_k(v4f *__restrict__ y, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * 256;
    const v4f v = {1.0f, 2.0f, 3.0f, 4.0f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        if (NT) __builtin_nontemporal_store(v, y + i);
        else y[i] = v;
    }
}
template <typename F> float timeit(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 20;
}
int main()
{
    const size_t n = (size_t)1 << 28, n4 = n / 4;   // 1 GiB per vector
    v4f *x, *y; float *o;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&o, 4);
    hipMemset(x, 0, n * 4); hipMemset(y, 0, n * 4);
    for (int grid : {2048, 8192, 32768, 262144}) {
        float ms;
        ms = timeit([&] { hipLaunchKernelGGL((copy_k<1, false>), dim3(grid), dim3(256), 0, 0, x, y, n4); });
        printf("grid %6d copy   plain    : %.3f ms  %.2f TB/s\n", grid, ms, 2.0 * n * 4 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((copy_k<4, false>), dim3(grid), dim3(256), 0, 0, x, y, n4); });
        printf("grid %6d copy   x4       : %.3f ms  %.2f TB/s\n", grid, ms, 2.0 * n * 4 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((copy_k<4, true>), dim3(grid), dim3(256), 0, 0, x, y, n4); });
        printf("grid %6d copy   x4 nt    : %.3f ms  %.2f TB/s\n", grid, ms, 2.0 * n * 4 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((read_k<4>), dim3(grid), dim3(256), 0, 0, x, o, n4); });
        printf("grid %6d read   x4       : %.3f ms  %.2f TB/s\n", grid, ms, 1.0 * n * 4 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((fill_k<false>), dim3(grid), dim3(256), 0, 0, y, n4); });
        printf("grid %6d fill   plain    : %.3f ms  %.2f TB/s\n", grid, ms, 1.0 * n * 4 / ms / 1e9);
        ms = timeit([&] { hipLaunchKernelGGL((fill_k<true>), dim3(grid), dim3(256), 0, 0, y, n4); });
        printf("grid %6d fill   nt       : %.3f ms  %.2f TB/s\n", grid, ms, 1.0 * n * 4 / ms / 1e9);
    }
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL((copy_chunk_k<4>), dim3(n4 / (256 * 4)), dim3(256), 0, 0, x, y, n4); });
    printf("chunk  4 el/thread (%zu workgroups): %.3f ms  %.2f TB/s\n", n4 / (256 * 4), ms, 2.0 * n * 4 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL((copy_chunk_k<16>), dim3(n4 / (256 * 16)), dim3(256), 0, 0, x, y, n4); });
    printf("chunk 16 el/thread (%zu workgroups): %.3f ms  %.2f TB/s\n", n4 / (256 * 16), ms, 2.0 * n * 4 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL((copy_chunk_k<64>), dim3(n4 / (256 * 64)), dim3(256), 0, 0, x, y, n4); });
    printf("chunk 64 el/thread (%zu workgroups): %.3f ms  %.2f TB/s\n", n4 / (256 * 64), ms, 2.0 * n * 4 / ms / 1e9);
    return 0;
}
