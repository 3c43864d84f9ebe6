// Fixture for tests/test_isa_lint.py: three kernels that MUST trip scripts/isa_lint.py, named into the timed families so that
// the family rules apply.  Compiled to a shared library by the test (hipcc cross-compiles gfx950 without a GPU), never run.
//   agg_gemm_fixture_indexed_array   a per-lane indexed private array inside a loop -> scratch accesses inside a loop (R2)
//                                    (the shape of round 4's accident: GnoProd::by_group written as a[g])
//   gno_pc_fixture_hazard            buffer_store_dwordx4 with an SGPR soffset, VALU write of its first data register behind it (R3)
//   gno_pc_fixture_padded            the same with two wait states between: must NOT be reported
#include <hip/hip_runtime.h>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void agg_gemm_fixture_indexed_array(const int *idx, float *out, int n)
{
    float a[160];
    for (int i = 0; i < 160; ++i) a[i] = out[i * 64 + threadIdx.x];
    float s = 0.0f;
    for (int k = 0; k < n; ++k) {
        const int j = (unsigned)idx[k * 64 + threadIdx.x] % 160u;
        s += a[j];
        a[(j + 7) % 160] = s;
    }
    out[threadIdx.x] = s;
}

template <int PAD>
__device__ __forceinline__ void store_then_write(unsigned *out, unsigned bytes, int soff)
{
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long p = (unsigned long long)out;
    v4i rs = {(int)(unsigned)p, (int)(unsigned)(p >> 32), (int)bytes, 0x00020000};
    rs[0] = __builtin_amdgcn_readfirstlane(rs[0]);
    rs[1] = __builtin_amdgcn_readfirstlane(rs[1]);
    rs[2] = __builtin_amdgcn_readfirstlane(rs[2]);
    rs[3] = __builtin_amdgcn_readfirstlane(rs[3]);
    v4u d = {4 * gid, 4 * gid + 1u, 4 * gid + 2u, 4 * gid + 3u};
    const unsigned voff = 16u * gid;
    if (PAD)
        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1\n\tv_mov_b32 v4, 0x7b" : "+{v[4:7]}"(d) : "v"(voff), "s"(rs), "s"(soff) : "memory");
    else
        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\tv_mov_b32 v4, 0x7b" : "+{v[4:7]}"(d) : "v"(voff), "s"(rs), "s"(soff) : "memory");
    if (d[0] != 0x7b) out[0] = 0;
}
__global__ void gno_pc_fixture_hazard(unsigned *out, unsigned bytes, int soff) { store_then_write<0>(out, bytes, soff); }
__global__ void gno_pc_fixture_padded(unsigned *out, unsigned bytes, int soff) { store_then_write<1>(out, bytes, soff); }
