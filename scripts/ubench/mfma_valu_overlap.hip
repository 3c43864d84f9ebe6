// Micro-benchmark: do fp32-input MFMAs and plain VALU instructions of OTHER waves of the same SIMD overlap on gfx950?
// One workgroup per CU, 8 waves = 2 per SIMD: waves 0-3 run MFMAs, waves 4-7 run independent v_fma chains.
//   mode 1: MFMA waves only   mode 2: VALU waves only   mode 3: both
// Build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(512) void k(int mode, int iters, float *out)
{
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (!(mode & 1)) return;
        v4f a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float x = threadIdx.x * 1e-3f, y = 1.0f + x;
        for (int i = 0; i < iters; ++i) {
            if (KIND == 0) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
            } else {
                typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
                bf8 p, q;
                for (int j = 0; j < 8; ++j) { p[j] = (__bf16)x; q[j] = (__bf16)y; }
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, q, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, q, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, q, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, q, a3, 0, 0, 0);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    } else {
        if (!(mode & 2)) return;
        float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
        const float m = 0.999f, c = 0.001f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f0 = __builtin_fmaf(f0, m, c); f1 = __builtin_fmaf(f1, m, c); f2 = __builtin_fmaf(f2, m, c); f3 = __builtin_fmaf(f3, m, c);
                f4 = __builtin_fmaf(f4, m, c); f5 = __builtin_fmaf(f5, m, c); f6 = __builtin_fmaf(f6, m, c); f7 = __builtin_fmaf(f7, m, c);
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    }
}
template <int KIND> void run(const char *name, float *out)
{
    const int iters = 20000;
    for (int mode = 1; mode <= 3; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, mode, iters, out);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, mode, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: 4 MFMAs x iters (one MFMA wave) and 32 v_fma x iters (one VALU wave)
        printf("%s mode %d (%s): %.3f ms  -> %.1f ns per iteration (4 MFMA | 32 v_fma)\n", name, mode,
               mode == 1 ? "MFMA only" : mode == 2 ? "VALU only" : "both", ms, ms * 1e6 / iters);
    }
}
int main()
{
    float *out; hipMalloc(&out, 256 * 512 * 4);
    run<0>("f32  16x16x4 ", out);
    run<1>("bf16 16x16x32", out);
    return 0;
}
