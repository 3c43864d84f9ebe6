// Micro-benchmark / reproducer: is the data of a buffer_store_dwordx4 safe when a VALU instruction rewrites its first data
// register straight behind the store?  LLVM's hazard recogniser (GCNHazardRecognizer::createsVALUHazard) inserts the wait
// states only when the store's soffset is NOT an SGPR; gno_dh_pc_kernel<PX> (round 4) lost the first dword of its last lanes'
// partials on gfx950 with a register soffset.  Three exact instruction sequences through inline asm, 2048 workgroups x 12
// waves each, every lane's four dwords checked:
//   A  store (soffset = SGPR) ; v_mov  v4, 0x7b          -- what the compiler emitted
//   B  store (soffset = 0)    ; v_mov  v4, 0x7b          -- the case LLVM knows (and pads when IT schedules)
//   C  store (soffset = SGPR) ; s_nop 1 ; v_mov v4, 0x7b -- two wait states
// Build: hipcc --offload-arch=gfx950 -O3 store_soffset_hazard.hip -o store_soffset_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(768) void k(unsigned *out, unsigned bytes, int soff)
{
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long p = (unsigned long long)out;
    v4i rs = {(int)(unsigned)p, (int)(unsigned)(p >> 32), (int)bytes, 0x00020000};
    rs[0] = __builtin_amdgcn_readfirstlane(rs[0]);
    rs[1] = __builtin_amdgcn_readfirstlane(rs[1]);
    rs[2] = __builtin_amdgcn_readfirstlane(rs[2]);
    rs[3] = __builtin_amdgcn_readfirstlane(rs[3]);
    v4u d = {4 * gid + 1000u, 4 * gid + 1001u, 4 * gid + 1002u, 4 * gid + 1003u};
    const unsigned voff = 16u * gid + 64u - (MODE == 1 ? 0u : (unsigned)soff);   // lane g's 16 bytes land at byte 64 + 16 g in every mode
    // some matrix work in front, as in the kernel where it was found (the store's data comes out of an MFMA there)
    float a = (float)gid, acc = 0.0f;
    for (int i = 0; i < 8; ++i) acc += a * 1.0001f;
    if (acc == 12345.678f) d[1] = 7;
    if (MODE == 0)
        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\tv_mov_b32 v4, 0x7b" : "+{v[4:7]}"(d) : "v"(voff), "s"(rs), "s"(soff) : "memory");
    else if (MODE == 1)
        asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\tv_mov_b32 v4, 0x7b" : "+{v[4:7]}"(d) : "v"(voff), "s"(rs) : "memory");
    else
        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1\n\tv_mov_b32 v4, 0x7b" : "+{v[4:7]}"(d) : "v"(voff), "s"(rs), "s"(soff) : "memory");
    if (d[0] != 0x7b) out[0] = 0;   // keep the rewritten register alive
}

template <int MODE> long run(const char *name)
{
    const int wg = 2048, th = 768;
    const size_t n = (size_t)wg * th;
    unsigned *dev;
    hipMalloc(&dev, n * 16 + 64);
    long bad = 0, bad_first = 0, bad_last16 = 0;
    std::vector<unsigned> h(n * 4 + 16);
    for (int rep = 0; rep < 20; ++rep) {
        hipMemset(dev, 0xff, n * 16 + 64);
        hipLaunchKernelGGL(k<MODE>, dim3(wg), dim3(th), 0, 0, dev, (unsigned)(n * 16 + 64), 64);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), dev, n * 16 + 64, hipMemcpyDeviceToHost);
        for (size_t g = 0; g < n; ++g)
            for (int j = 0; j < 4; ++j)
                if (h[16 + 4 * g + j] != 4 * (unsigned)g + 1000u + j) {
                    ++bad;
                    if (j == 0) ++bad_first;
                    if ((g & 63) >= 48) ++bad_last16;
                }
    }
    printf("%-44s wrong dwords %8ld of %ld   (first dword of a lane: %ld, lanes 48..63: %ld)\n", name, bad, (long)(20 * n * 4), bad_first, bad_last16);
    hipFree(dev);
    return bad;
}

int main()
{
    run<0>("A  store, soffset SGPR ; v_mov v4");
    run<1>("B  store, soffset 0    ; v_mov v4");
    run<2>("C  store, soffset SGPR ; s_nop 1 ; v_mov v4");
    return 0;
}
