// Micro-benchmark: what FETCH_SIZE (rocprofv3 --pmc) tallies for a gather of 64-BYTE segments -- the access the GNO producer
// waves make (16 lanes x 4 B = one quarter of a 256-byte feature row per entry) -- against a streaming read.
//   k_half    : every 16-lane group reads ONE random 64-byte half of a random 128-byte line          (N x 64 B asked for)
//   k_both    : the same lines, both halves by two consecutive load instructions of the same wave     (N x 128 B asked for)
//   k_stream  : contiguous 16 B per lane                                                              (bytes of the table)
// If the L2 brings in whole 128-byte lines, k_both's second half hits and both kernels fetch the same bytes; if it
// fetches 64-byte sectors, k_both fetches twice k_half's.  Table 4 GB (beyond L2 and the 256 MB Infinity Cache).
// Build: hipcc --offload-arch=gfx950 -O3 gather64.hip -o gather64 ; run under rocprofv3 --pmc FETCH_SIZE (and TCC_HIT_sum TCC_MISS_sum)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
template <int BOTH>
__global__ __launch_bounds__(256) void k_gather(const float *__restrict__ tab, uint32_t n_lines, int per_wave, float *out)
{
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    float acc = 0.0f;
    for (int i = 0; i < per_wave; ++i) {
        const uint32_t line = mix((wave * (uint32_t)per_wave + i) * 4u + g) % n_lines;   // 4 segments per load instruction
        const uint32_t half = BOTH ? 0u : (mix(line) & 1u);
        const float *p = tab + (size_t)line * 32 + half * 16 + n;
        acc += p[0];
        if (BOTH) acc += p[16];
    }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_stream(const v4f *__restrict__ tab, size_t n16, float *out)
{
    v4f acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) acc = acc + tab[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}
template <typename F> float timeit(F f)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}
int main()
{
    const size_t bytes = 4ull << 30;
    float *tab, *out;
    if (hipMalloc(&tab, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    (void)hipMemset(tab, 0, bytes);
    const uint32_t n_lines = (uint32_t)(bytes / 128);
    const int waves = 2048 * 4, per_wave = 2048;                 // 8192 waves x 2048 loads x 4 segments = 67.1 M segments
    const double segs = (double)waves * per_wave * 4;
    float ms = timeit([&] { hipLaunchKernelGGL(k_gather<0>, dim3(2048), dim3(256), 0, 0, tab, n_lines, per_wave, out); });
    printf("k_half  : %.0f M segments of 64 B = %.2f GB asked for, %.3f ms, %.2f TB/s of asked bytes\n", segs / 1e6, segs * 64 / 1e9, ms, segs * 64 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(k_gather<1>, dim3(2048), dim3(256), 0, 0, tab, n_lines, per_wave, out); });
    printf("k_both  : %.0f M lines of 128 B = %.2f GB asked for, %.3f ms, %.2f TB/s of asked bytes\n", segs / 1e6, segs * 128 / 1e9, ms, segs * 128 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, (const v4f *)tab, bytes / 16, out); });
    printf("k_stream: %.2f GB, %.3f ms, %.2f TB/s\n", bytes / 1e9, ms, bytes / ms / 1e9);
    return 0;
}
