// Fused Kipf layer kernels: CSR gather-aggregate + dense contraction in ONE launch.
//
//   forward :  P = A^ X  (kept for dW),  Z = act(P . Wt + b)      athena_kipf_msgpass_layer.f90:943-952
//   backward:  dX = (A^T dZ) . W  ==  A^T (dZ . W)                  matmul reverse + ..._sub_kipf.f90:85-111
//              (the contraction is linear, so aggregating first is the same map; rounding differs
//               from the unfused order by fp32 re-association only)
//
// Why: the unfused step is HBM bound as a whole; the dense kernels are matrix-pipe bound
// and the aggregation is fabric bound, but run back to back they cannot overlap and P / dP make a
// round trip through HBM.  Here a 16-wave workgroup per CU keeps W resident in LDS (K, N = 128: 67.6 KB)
// and software-pipelines 64-row chunks through two LDS tile buffers:
//   gather (every wave: one row pair, same half-wave-per-row mapping and CSR-order summation as
//           csr_gather_agg: P is bit-identical to the unfused kernel) -> LDS tile [32][K+4] (+ P rows
//           to HBM, forward)
//   MFMA   (every wave: one 16x16 block of the chunk's 32x128 output, v_mfma_f32_16x16x4_f32 against
//           the resident W) while the first 16 row loads of the NEXT chunk are already in flight.
// One __syncthreads per chunk; row pointers and the first index block are prefetched a chunk ahead, so
// only the row gathers themselves sit on the latency chain.  (Two earlier forms -- dedicated MFMA waves
// with 3-pair gather waves, and 64-row chunks with two pairs per wave -- measured 1.40 / 1.03 ms
// against 0.98 ms for this one and 1.12 ms unfused.)
#include <algorithm>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int ACT> __device__ __forceinline__ float act_f(float z)
{
    if constexpr (ACT == ATHENA_MP_ACT_RELU) return z > 0.0f ? z : 0.0f;
    if constexpr (ACT == ATHENA_MP_ACT_SIGMOID) return 1.0f / (1.0f + expf(-z));
    if constexpr (ACT == ATHENA_MP_ACT_TANH) return tanhf(z);
    return z;
}

constexpr int kUn = 4;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// 32-row chunks, one row pair per wave; the first kFirst row loads of the NEXT chunk are issued before
// the matrix work of the CURRENT chunk and consumed after it.
#define FUSED_STAGGER 0   // measured at configs[1]: 0.936 / 0.884 ms with, 0.930 / 0.890 ms without (forward / reverse): no gain here
#define KFIRST 16
constexpr int kFirst = KFIRST;

#define KIPF_NT 13   // bit mask, A/B in profiles/r04_kipf_nt_ab.txt: nontemporal stores of P (1), of Z (2; 4 = agg_gemm256_kernel's forward launch), nontemporal loads of agg_gemm_kernel's forward-launch entry ids + coefficients (8; the same in agg_gemm256_kernel measured slower, not in the tree)
template <int N, bool COEF, int ACT, bool BUF>
__global__ __launch_bounds__(1024) void agg_gemm_kernel(const int32_t *__restrict__ rowptr,
                                                        const int32_t *__restrict__ idx,
                                                        const float *__restrict__ coef,
                                                        const float *__restrict__ x,
                                                        const float *__restrict__ B, int b_nk,
                                                        const float *__restrict__ bias,
                                                        float *__restrict__ P, float *__restrict__ Z,
                                                        int64_t n_rows, unsigned long long *__restrict__ ticket,
                                                        uint32_t x_bytes)
{
    // square layers, N = K in {64, 128}: a row is K/4 lanes of float4, a wave holds 64/(K/4) rows, a chunk is
    // 16 waves' worth of rows, and its CH x N output is exactly 16 blocks of 16x16 -- one per wave
    static_assert(N == 64 || N == 128, "fused kernel: 64- or 128-wide square layers");
    constexpr int K = N;
    constexpr int LD = K + 4;
    constexpr int G = K / 4;              // lanes per row
    constexpr int RPW = 64 / G;           // rows per wave: 2 (K=128) or 4 (K=64)
    constexpr int CH = 16 * RPW;          // rows per chunk
    constexpr int KG = K / 4;             // k range of one MFMA lane group (k = KG*g4 + s)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Bs = lds;                      // [N][LD]
    float *Ts = lds + N * LD;             // [2 buffers][CH rows][LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane / G, gl = lane & (G - 1);

    {   // W resident in LDS as [n][K+4]
        constexpr int NV = K * N / 4, PER = (NV + 1023) / 1024;
        v4f tmp[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            int t = i * 1024 + tid;
            tmp[i] = reinterpret_cast<const v4f *>(B)[t < NV ? t : NV - 1];
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            int t = i * 1024 + tid;
            if (t < NV) {
                if (b_nk) {
                    int n = t / (K / 4), q = t - n * (K / 4);
                    *reinterpret_cast<v4f *>(Bs + n * LD + 4 * q) = tmp[i];
                } else {
                    int k = t / (N / 4), n4 = t - k * (N / 4);
                    Bs[(4 * n4 + 0) * LD + k] = tmp[i].x;
                    Bs[(4 * n4 + 1) * LD + k] = tmp[i].y;
                    Bs[(4 * n4 + 2) * LD + k] = tmp[i].z;
                    Bs[(4 * n4 + 3) * LD + k] = tmp[i].w;
                }
            }
        }
    }
    const int64_t n_chunks = (n_rows + CH - 1) / CH;
    // matrix task: rows 16*rb .. +15 of the chunk, columns 16*ct .. +15
    const int rb = wave % RPW, ct = wave / RPW;
    const int l15 = lane & 15, g4 = lane >> 4;
    const float bv = bias ? bias[16 * ct + l15] : 0.0f;
    const int lrow = RPW * wave + h;      // this lane group's row inside a chunk

    // per-chunk gather state of this half-wave's row
    int start = 0, len = 0, idx0 = -1;
    float c0 = 0.0f;
    auto load_state = [&](int64_t chunk) {
        start = 0; len = 0; idx0 = -1; c0 = 0.0f;
        const int64_t row = chunk * CH + lrow;
        if (chunk < n_chunks && row < n_rows) {
            start = rowptr[row];
            len = rowptr[row + 1] - start;
        }
        if (gl < len) {
            // (the forward launch reads its 80 MB of entry ids and coefficients nontemporal -- they are used once and would
            // push rows of X out of the caches: + 1.8 % on it; the coefficient-free reverse launch lost 0.8 % with the same)
            if constexpr (COEF && (KIPF_NT & 8)) {
                idx0 = __builtin_nontemporal_load(idx + start + gl);
                c0 = __builtin_nontemporal_load(coef + start + gl);
            } else {
                idx0 = idx[start + gl];
                if constexpr (COEF) c0 = coef[start + gl];
            }
        }
    };
    // Row loads.  BUF: the gathered tensor is below 4 GB, so a row is addressed by a 32-bit byte offset against a
    // buffer descriptor held in scalar registers -- one v_lshl_add per load instead of a 64-bit multiply-add pair
    // (the gather is bound by its instruction stream before it is bound by the fabric); offsets past
    // x_bytes read as zero.
    __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, BUF ? (int)x_bytes : 0, 0x00020000);
    auto load_row = [&](int u) -> v4f {
        if constexpr (BUF) {
            typedef int v4i_ __attribute__((ext_vector_type(4)));
            const v4i_ t = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(((uint32_t)u << (K == 128 ? 9 : 8)) + 16u * (uint32_t)gl), 0, 0);   // byte offset in uint32: tensors of 2 .. 4 GB
            return __builtin_bit_cast(v4f, t);
        } else {
            return *reinterpret_cast<const v4f *>(x + (int64_t)u * K + 4 * gl);
        }
    };
    constexpr int kF = kFirst < G ? kFirst : G;   // entries whose loads fly under the matrix work
    v4f v[kF];
    // Slots are predicated (a slot beyond the row's length issues nothing when neither row of the wave reaches it).
    // Measured and dropped: unpredicated slots, with the byte offset 0xFFFFFFFF for a slot beyond the row so that the
    // buffer bounds check returns zeros and the sum adds every slot unconditionally -- 616 -> 422 instructions per chunk
    // iteration in the ISA, yet 2.06 ms per step against 2.05: the empty slots (rows average 10 of the 16) then cost a
    // trip through the address unit each, which is more than the scalar exec-mask bookkeeping they replace.
    auto issue_first = [&]() {            // row loads of entries 0 .. kF-1 (no waits)
#pragma unroll
        for (int k = 0; k < kF; ++k) {
            const int u = __shfl(idx0, k, G);
            v[k] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
            if (k < len) v[k] = load_row(u);      // the Kipf CSR carries no negative ids: no `u >= 0` test, and the sum
                                                  // below needs no second broadcast of the index
        }
    };
    auto finish = [&]() -> v4f {          // CSR-order accumulation: first block from registers, rest streamed
        v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < kF; ++k) {
            const float c = COEF ? __shfl(c0, k, G) : 1.0f;
            if (k < len) {
                if constexpr (COEF) { acc.x = acc.x + c * v[k].x; acc.y = acc.y + c * v[k].y; acc.z = acc.z + c * v[k].z; acc.w = acc.w + c * v[k].w; }
                else { acc.x = acc.x + v[k].x; acc.y = acc.y + v[k].y; acc.z = acc.z + v[k].z; acc.w = acc.w + v[k].w; }
            }
        }
        int maxlen = len;
#pragma unroll
        for (int o = G; o < 64; o <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o));
        for (int off = 0; off < maxlen; off += G) {            // entries kFirst.. of long rows
            int my_idx = idx0;
            float my_c = c0;
            if (off > 0) {
                my_idx = -1; my_c = 0.0f;
                if (off + gl < len) {
                    if constexpr (COEF && (KIPF_NT & 8)) {
                        my_idx = __builtin_nontemporal_load(idx + start + off + gl);
                        my_c = __builtin_nontemporal_load(coef + start + off + gl);
                    } else {
                        my_idx = idx[start + off + gl];
                        if constexpr (COEF) my_c = coef[start + off + gl];
                    }
                }
            }
            const int cntmax = min(G, maxlen - off);
            for (int j = (off == 0 ? kF : 0); j < cntmax; j += kUn) {
                int u[kUn];
                float c[kUn];
                v4f w[kUn];
#pragma unroll
                for (int k = 0; k < kUn; ++k) {
                    u[k] = __shfl(my_idx, j + k, G);
                    if constexpr (COEF) c[k] = __shfl(my_c, j + k, G);
                    if (off + j + k >= len) u[k] = -1;
                }
#pragma unroll
                for (int k = 0; k < kUn; ++k) {
                    w[k] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
                    if (u[k] >= 0) w[k] = load_row(u[k]);
                }
#pragma unroll
                for (int k = 0; k < kUn; ++k) {
                    if (u[k] >= 0) {
                        if constexpr (COEF) { acc.x = acc.x + c[k] * w[k].x; acc.y = acc.y + c[k] * w[k].y; acc.z = acc.z + c[k] * w[k].z; acc.w = acc.w + c[k] * w[k].w; }
                        else { acc.x = acc.x + w[k].x; acc.y = acc.y + w[k].y; acc.z = acc.z + w[k].z; acc.w = acc.w + w[k].w; }
                    }
                }
            }
        }
        return acc;
    };
    auto store_row = [&](int64_t chunk, int buf, const v4f &acc) {
        const int64_t row = chunk * CH + lrow;
        *reinterpret_cast<v4f *>(Ts + (buf * CH + lrow) * LD + 4 * gl) = acc;
        if (P != nullptr && chunk < n_chunks && row < n_rows) {
#if KIPF_NT & 1
            __builtin_nontemporal_store(acc, reinterpret_cast<v4f *>(P + row * K + 4 * gl));
#else
            *reinterpret_cast<v4f *>(P + row * K + 4 * gl) = acc;
#endif
        }
    };

    // (Measured and dropped: walking the rows longest first so that the rows of a chunk have equal lengths -- 6 % slower,
    // every level of the walk turns into a random access; walking only the 2 % of rows longer than the prefetched block
    // last, in chunks of their own -- no change: the exposed round of loads such a row costs its workgroup is not what a
    // chunk waits for.)
    // Pipeline, one barrier per chunk.  Chunks are handed out by a device-wide ticket counter (zeroed by
    // the launcher) so a workgroup that starts late -- e.g. because a communication kernel holds its CU --
    // or that meets heavier rows simply takes fewer chunks; results do not depend on who processes what.
    //   prologue: tickets k0, k1, k2; gather k0 -> buffer 0; prefetch state of k1; barrier
    //   iteration: issue first row loads of k1 | MFMA k0 (buffer b) | finish k1 -> buffer b^1 |
    //              prefetch state of k2 | draw the next ticket | barrier | (k0,k1,k2) <- (k1,k2,new)
    // Buffer b is rewritten one iteration later, after the barrier: all MFMA reads of it are done.
    __shared__ int64_t s_ticket[4];
    auto draw = [&]() -> int64_t { return (int64_t)atomicAdd(ticket, 1ull); };
    if (tid == 0) { s_ticket[0] = draw(); s_ticket[1] = draw(); s_ticket[2] = draw(); }
    __syncthreads();
    int64_t k0 = s_ticket[0], k1 = s_ticket[1], k2 = s_ticket[2];
    load_state(k0);
    issue_first();
    store_row(k0, 0, finish());
    load_state(k1);
    __syncthreads();
    for (int it = 0; k0 < n_chunks; ++it) {
        const int64_t chunk = k0;
        const int buf = it & 1;
        issue_first();                                      // rows of k1
        auto matrix_work = [&]() {
            const float *arow = Ts + (buf * CH + 16 * rb + l15) * LD + KG * g4;
            const float *brow = Bs + (16 * ct + l15) * LD + KG * g4;
            f32x4 c = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int q = 0; q < KG / 4; ++q) {
                const v4f a4 = *reinterpret_cast<const v4f *>(arow + 4 * q);
                const v4f b4 = *reinterpret_cast<const v4f *>(brow + 4 * q);
                c = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, c, 0, 0, 0);
            }
            const int64_t row0 = chunk * CH + 16 * rb + 4 * g4;   // C/D: col = lane&15, row = 4*(lane>>4) + reg
            const int col = 16 * ct + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (row0 + r < n_rows) {
#if KIPF_NT & 2
                    __builtin_nontemporal_store(act_f<ACT>(c[r] + bv), &Z[(row0 + r) * N + col]);
#else
                    Z[(row0 + r) * N + col] = act_f<ACT>(c[r] + bv);
#endif
                }
        };
#if FUSED_STAGGER
        // Between two barriers a wave has two independent jobs: the matrix work of chunk k0 (reads tile `buf`) and the
        // gather of its rows of chunk k1 (writes tile `buf ^ 1`).  Waves 4-7 and 12-15 run them in the opposite order, so
        // every SIMD holds two waves on the matrix pipe and two waiting for rows instead of four doing the same thing.
        if ((wave >> 2) & 1) {
            store_row(k1, buf ^ 1, finish());
            matrix_work();
        } else
#endif
        {
            matrix_work();
            store_row(k1, buf ^ 1, finish());
        }
        load_state(k2);
        if (tid == 0) s_ticket[it & 1] = draw();
        __syncthreads();
        k0 = k1;
        k1 = k2;
        k2 = s_ticket[it & 1];
    }
}

template <int N, bool COEF, int ACT, bool BUF>
int launch_fused_b(const int32_t *rowptr, const int32_t *idx, const float *coef, const float *x, const float *B,
                   int b_nk, const float *bias, float *P, float *Z, int64_t n_rows, int grid, uint32_t x_bytes)
{
    constexpr int CHr = 16 * (64 / (N / 4));
    constexpr size_t lds = sizeof(float) * ((size_t)N * (N + 4) + 2 * CHr * (N + 4));
    static amp::PerDeviceFlag attr;
    if (!attr.get()) {
        AMP_HIP(hipFuncSetAttribute((const void *)agg_gemm_kernel<N, COEF, ACT, BUF>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.get() = true;
    }
    // chunk ticket counter: one 8-byte word per launch from a small per-device ring, zeroed on the stream
    static int slot = 0;
    unsigned long long *ring = nullptr;
    if (amp::named_buffer("fused.ticket_ring", sizeof(unsigned long long) * 64, true, (void **)&ring)) return 1;
    unsigned long long *ticket = ring + (slot++ & 63);
    AMP_HIP(hipMemsetAsync(ticket, 0, sizeof(unsigned long long), amp::stream()));
    hipLaunchKernelGGL((agg_gemm_kernel<N, COEF, ACT, BUF>), dim3(grid), dim3(1024), lds, amp::stream(), rowptr, idx, coef, x,
                       B, b_nk, bias, P, Z, n_rows, ticket, x_bytes);
    AMP_LAUNCH_CHECK();
    return 0;
}

// x_rows: rows of the gathered tensor (0 = unknown).  Below 4 GB the rows are addressed through a buffer descriptor.
template <int N, bool COEF, int ACT>
int launch_fused(const int32_t *rowptr, const int32_t *idx, const float *coef, const float *x, const float *B,
                 int b_nk, const float *bias, float *P, float *Z, int64_t n_rows, int grid, int64_t x_rows)
{
    // buffer-descriptor addressing wherever the tensor is below 4 GB (A/B in DESIGN.md 3.1b-bis; the switch is gone)
    const int64_t bytes = x_rows * N * 4;
    if (x_rows > 0 && bytes < ((int64_t)1 << 32) - 4096)
        return launch_fused_b<N, COEF, ACT, true>(rowptr, idx, coef, x, B, b_nk, bias, P, Z, n_rows, grid, (uint32_t)bytes);
    return launch_fused_b<N, COEF, ACT, false>(rowptr, idx, coef, x, B, b_nk, bias, P, Z, n_rows, grid, 0u);
}


// ---------------------------------------------------------------------------------------------------------------------
// 256 -> 256 layers (BASELINE configs[4]'s width).  W is 256 KB: it does not fit the 160 KB of LDS, so it lives in
// REGISTERS instead -- an 8-wave workgroup (two waves per SIMD, 256 registers per lane), wave w owning output columns
// [32w, 32w+32): Wr[j][s] = B[k = 64 g4 + s][n = 32 w + 16 j + (lane & 15)] is exactly the B operand of MFMA step s of
// column block j (128 registers), and only the gathered rows pass through LDS (two [16][260] tiles, 33 KB; pitch 260 with
// k = 64 g4 + s is conflict-free for ds_read_b128).  A row is one whole wave (64 lanes x 16 B = 1 KB), a chunk is 16 rows =
// two per wave, its [16 x 256] output is 16 column blocks = two per wave, 128 MFMAs (16x16x4) per wave per chunk, two
// independent accumulators.  Same software pipeline as the LDS-resident kernel above: the first 8 row loads of each of the
// wave's two NEXT rows are issued before the matrix work of the current chunk and consumed after it; rows are summed in CSR
// order, so P is bit-identical to csr_gather_agg.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kFirst256 = 8;
constexpr int kTail256 = 4;        // entries per row per round beyond the prefetched block (both rows together)

template <bool COEF, int ACT, bool BUF>
__global__ __launch_bounds__(512) void agg_gemm256_kernel(const int32_t *__restrict__ rowptr,
                                                          const int32_t *__restrict__ idx,
                                                          const float *__restrict__ coef,
                                                          const float *__restrict__ x,
                                                          const float *__restrict__ B, int b_nk,
                                                          const float *__restrict__ bias,
                                                          float *__restrict__ P, float *__restrict__ Z,
                                                          int64_t n_rows, unsigned long long *__restrict__ ticket,
                                                          uint32_t x_bytes)
{
    constexpr int K = 256, N = 256, LD = 260, CH = 16, KG = 64, kF = kFirst256;
    __shared__ __attribute__((aligned(16))) float Ts[2 * CH * LD];
    __shared__ int64_t s_ticket[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int n0 = 32 * wave;

    float Wr[2][64];
    if (b_nk) {   // B stored [N][K]
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float *src = B + (int64_t)(n0 + 16 * j + l15) * K + KG * g4;
#pragma unroll
            for (int s4 = 0; s4 < 16; ++s4) {
                const v4f t = *reinterpret_cast<const v4f *>(src + 4 * s4);
                Wr[j][4 * s4 + 0] = t.x; Wr[j][4 * s4 + 1] = t.y; Wr[j][4 * s4 + 2] = t.z; Wr[j][4 * s4 + 3] = t.w;
            }
        }
    } else {      // B stored [K][N]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 64; ++s) Wr[j][s] = B[(int64_t)(KG * g4 + s) * N + n0 + 16 * j + l15];
    }
    float bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bv[j] = bias ? bias[n0 + 16 * j + l15] : 0.0f;

    const int64_t n_chunks = (n_rows + CH - 1) / CH;
    int start[2], len[2], idx0[2];
    float c0[2];
    auto load_state = [&](int64_t chunk) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            start[r] = 0; len[r] = 0; idx0[r] = -1; c0[r] = 0.0f;
            const int64_t row = chunk * CH + 2 * wave + r;
            if (chunk < n_chunks && row < n_rows) {
                start[r] = rowptr[row];
                len[r] = rowptr[row + 1] - start[r];
            }
            if (lane < len[r]) {
                idx0[r] = idx[start[r] + lane];
                if constexpr (COEF) c0[r] = coef[start[r] + lane];
            }
        }
    };
    // row loads: 32-bit byte offsets against a buffer descriptor when the gathered tensor is below 4 GB (see the
    // 64/128-wide kernel above)
    __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, BUF ? (int)x_bytes : 0, 0x00020000);
    auto load_row = [&](int u) -> v4f {
        if constexpr (BUF) {
            typedef int v4i_ __attribute__((ext_vector_type(4)));
            const v4i_ t = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (int)(((uint32_t)u << 10) + 16u * (uint32_t)lane), 0, 0);   // byte offset in uint32: tensors of 2 .. 4 GB
            return __builtin_bit_cast(v4f, t);
        } else {
            return *reinterpret_cast<const v4f *>(x + (int64_t)u * K + 4 * lane);
        }
    };
    v4f v[2][kF];
    auto issue_first = [&]() {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < kF; ++k) {
                const int u = __shfl(idx0[r], k, 64);
                v[r][k] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
                if (k < len[r]) v[r][k] = load_row(u);
            }
    };
    // both rows of the wave advance TOGETHER through the entries beyond the prefetched block (two rows' loads in flight
    // per round instead of one); each row still adds its entries in CSR order
    auto store_rows = [&](int64_t chunk, int buf) {
        v4f acc[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            acc[r] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < kF; ++k) {
                const float c = COEF ? __shfl(c0[r], k, 64) : 1.0f;
                if (k < len[r]) {
                    if constexpr (COEF) { acc[r].x = acc[r].x + c * v[r][k].x; acc[r].y = acc[r].y + c * v[r][k].y; acc[r].z = acc[r].z + c * v[r][k].z; acc[r].w = acc[r].w + c * v[r][k].w; }
                    else { acc[r].x = acc[r].x + v[r][k].x; acc[r].y = acc[r].y + v[r][k].y; acc[r].z = acc[r].z + v[r][k].z; acc[r].w = acc[r].w + v[r][k].w; }
                }
            }
        }
        const int nmax = max(len[0], len[1]);                // wave-uniform
        for (int off = 0; off < nmax; off += 64) {
            int my_idx[2];
            float my_c[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                my_idx[r] = idx0[r]; my_c[r] = c0[r];
                if (off > 0) {
                    my_idx[r] = -1; my_c[r] = 0.0f;
                    if (off + lane < len[r]) {
                        my_idx[r] = idx[start[r] + off + lane];
                        if constexpr (COEF) my_c[r] = coef[start[r] + off + lane];
                    }
                }
            }
            const int cnt = min(64, nmax - off);
            for (int j = (off == 0 ? kF : 0); j < cnt; j += kTail256) {
                int u[2][kTail256];
                float c[2][kTail256];
                v4f w[2][kTail256];
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int k = 0; k < kTail256; ++k) {
                        u[r][k] = __shfl(my_idx[r], j + k, 64);
                        if constexpr (COEF) c[r][k] = __shfl(my_c[r], j + k, 64);
                        if (off + j + k >= len[r]) u[r][k] = -1;
                    }
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int k = 0; k < kTail256; ++k) {
                        w[r][k] = (v4f){0.0f, 0.0f, 0.0f, 0.0f};
                        if (u[r][k] >= 0) w[r][k] = load_row(u[r][k]);
                    }
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int k = 0; k < kTail256; ++k) {
                        if (u[r][k] >= 0) {
                            if constexpr (COEF) { acc[r].x = acc[r].x + c[r][k] * w[r][k].x; acc[r].y = acc[r].y + c[r][k] * w[r][k].y; acc[r].z = acc[r].z + c[r][k] * w[r][k].z; acc[r].w = acc[r].w + c[r][k] * w[r][k].w; }
                            else { acc[r].x = acc[r].x + w[r][k].x; acc[r].y = acc[r].y + w[r][k].y; acc[r].z = acc[r].z + w[r][k].z; acc[r].w = acc[r].w + w[r][k].w; }
                        }
                    }
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int lrow = 2 * wave + r;
            const int64_t row = chunk * CH + lrow;
            *reinterpret_cast<v4f *>(Ts + (buf * CH + lrow) * LD + 4 * lane) = acc[r];
            if (P != nullptr && chunk < n_chunks && row < n_rows) {
#if KIPF_NT & 1
                __builtin_nontemporal_store(acc[r], reinterpret_cast<v4f *>(P + row * K + 4 * lane));
#else
                *reinterpret_cast<v4f *>(P + row * K + 4 * lane) = acc[r];
#endif
            }
        }
    };
    auto matrix_work = [&](int64_t chunk, int buf) {     // this wave's two 16x16 blocks of the chunk's [16 x 256] output
        const float *arow = Ts + (buf * CH + l15) * LD + KG * g4;
        f32x4 ca = {0.0f, 0.0f, 0.0f, 0.0f}, cb = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const v4f a4 = *reinterpret_cast<const v4f *>(arow + 4 * q);
            ca = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, Wr[0][4 * q + 0], ca, 0, 0, 0);
            cb = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, Wr[1][4 * q + 0], cb, 0, 0, 0);
            ca = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, Wr[0][4 * q + 1], ca, 0, 0, 0);
            cb = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, Wr[1][4 * q + 1], cb, 0, 0, 0);
            ca = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, Wr[0][4 * q + 2], ca, 0, 0, 0);
            cb = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, Wr[1][4 * q + 2], cb, 0, 0, 0);
            ca = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, Wr[0][4 * q + 3], ca, 0, 0, 0);
            cb = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, Wr[1][4 * q + 3], cb, 0, 0, 0);
        }
        const int64_t row0 = chunk * CH + 4 * g4;           // C/D: col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (row0 + r < n_rows) {
                // (the forward launch -- the one that also keeps P -- streams Z out nontemporal: - 1.4 % on it at configs[4];
                // the reverse launch, whose output the next layer's reverse reads, does not: + 0.7 % there)
                if ((KIPF_NT & 4) && P != nullptr) {
                    __builtin_nontemporal_store(act_f<ACT>(ca[r] + bv[0]), &Z[(row0 + r) * N + n0 + l15]);
                    __builtin_nontemporal_store(act_f<ACT>(cb[r] + bv[1]), &Z[(row0 + r) * N + n0 + 16 + l15]);
                } else {
                    Z[(row0 + r) * N + n0 + l15] = act_f<ACT>(ca[r] + bv[0]);
                    Z[(row0 + r) * N + n0 + 16 + l15] = act_f<ACT>(cb[r] + bv[1]);
                }
            }
    };

    auto draw = [&]() -> int64_t { return (int64_t)atomicAdd(ticket, 1ull); };
    if (tid == 0) { s_ticket[0] = draw(); s_ticket[1] = draw(); s_ticket[2] = draw(); }
    __syncthreads();
    int64_t k0 = s_ticket[0], k1 = s_ticket[1], k2 = s_ticket[2];
    load_state(k0);
    issue_first();
    store_rows(k0, 0);
    load_state(k1);
    __syncthreads();
    for (int it = 0; k0 < n_chunks; ++it) {
        const int buf = it & 1;
        issue_first();                                      // first kF row loads of both rows of k1 ...
        matrix_work(k0, buf);                               // ... fly under the matrix work of k0
        store_rows(k1, buf ^ 1);
        load_state(k2);
        if (tid == 0) s_ticket[it & 1] = draw();
        __syncthreads();
        k0 = k1;
        k1 = k2;
        k2 = s_ticket[it & 1];
    }
}

template <bool COEF, int ACT>
int launch_fused256(const int32_t *rowptr, const int32_t *idx, const float *coef, const float *x, const float *B, int b_nk,
                    const float *bias, float *P, float *Z, int64_t n_rows, int grid, int64_t x_rows)
{
    const int64_t bytes = x_rows * 1024;
    const bool buf = x_rows > 0 && bytes < ((int64_t)1 << 32) - 4096;
    static int slot = 0;
    unsigned long long *ring = nullptr;
    if (amp::named_buffer("fused.ticket_ring", sizeof(unsigned long long) * 64, true, (void **)&ring)) return 1;
    unsigned long long *ticket = ring + (slot++ & 63);
    AMP_HIP(hipMemsetAsync(ticket, 0, sizeof(unsigned long long), amp::stream()));
    if (buf)
        hipLaunchKernelGGL((agg_gemm256_kernel<COEF, ACT, true>), dim3(grid), dim3(512), 0, amp::stream(), rowptr, idx, coef, x, B,
                           b_nk, bias, P, Z, n_rows, ticket, (uint32_t)bytes);
    else
        hipLaunchKernelGGL((agg_gemm256_kernel<COEF, ACT, false>), dim3(grid), dim3(512), 0, amp::stream(), rowptr, idx, coef, x, B,
                           b_nk, bias, P, Z, n_rows, ticket, 0u);
    AMP_LAUNCH_CHECK();
    return 0;
}

int fused256_dispatch(const int32_t *rowptr, const int32_t *idx, const float *coef, const float *x, const float *B, int b_nk,
                      const float *bias, int act, float *P, float *Z, int64_t n_rows, int64_t x_rows)
{
    const int grid = (int)std::min<int64_t>((n_rows + 15) / 16, amp::num_cus());
    if (grid == 0) return 0;
#define AMP_G(ACT_)                                                                                        \
    return coef ? launch_fused256<true, ACT_>(rowptr, idx, coef, x, B, b_nk, bias, P, Z, n_rows, grid, x_rows)    \
                : launch_fused256<false, ACT_>(rowptr, idx, coef, x, B, b_nk, bias, P, Z, n_rows, grid, x_rows)
    switch (act) {
    case ATHENA_MP_ACT_RELU: AMP_G(ATHENA_MP_ACT_RELU);
    case ATHENA_MP_ACT_SIGMOID: AMP_G(ATHENA_MP_ACT_SIGMOID);
    case ATHENA_MP_ACT_TANH: AMP_G(ATHENA_MP_ACT_TANH);
    default: AMP_G(ATHENA_MP_ACT_NONE);
    }
#undef AMP_G
}

bool fused_shape(int K, int N) { return K == N && (K == 64 || K == 128 || K == 256); }

int fused_dispatch(const int32_t *rowptr, const int32_t *idx, const float *coef, const float *x, int K, int N,
                   const float *B, int b_nk, const float *bias, int act, float *P, float *Z, int64_t n_rows,
                   int64_t x_rows = 0)
{
    if (!fused_shape(K, N)) {
        amp::set_error("fused Kipf layer kernel: built for 64 -> 64, 128 -> 128 and 256 -> 256 features, got %d -> %d", K, N);
        return 2;
    }
    if (K == 256) return fused256_dispatch(rowptr, idx, coef, x, B, b_nk, bias, act, P, Z, n_rows, x_rows);
    const int cus = amp::num_cus();
    const int ch = 16 * (64 / (N / 4));
    const int grid = (int)std::min<int64_t>((n_rows + ch - 1) / ch, cus);
    if (grid == 0) return 0;
    const bool c = coef != nullptr;
#define AMP_F2(NN_, ACT_)                                                                                    \
    return c ? launch_fused<NN_, true, ACT_>(rowptr, idx, coef, x, B, b_nk, bias, P, Z, n_rows, grid, x_rows)  \
             : launch_fused<NN_, false, ACT_>(rowptr, idx, coef, x, B, b_nk, bias, P, Z, n_rows, grid, x_rows)
#define AMP_F(ACT_)                    \
    if (N == 64) { AMP_F2(64, ACT_); } \
    AMP_F2(128, ACT_)
    switch (act) {
    case ATHENA_MP_ACT_RELU: AMP_F(ATHENA_MP_ACT_RELU);
    case ATHENA_MP_ACT_SIGMOID: AMP_F(ATHENA_MP_ACT_SIGMOID);
    case ATHENA_MP_ACT_TANH: AMP_F(ATHENA_MP_ACT_TANH);
    default: AMP_F(ATHENA_MP_ACT_NONE);
    }
#undef AMP_F
#undef AMP_F2
}

} // namespace

using namespace amp;

extern "C" {

int athena_mp_kipf_layer_fwd(const athena_mp_graph *g, int32_t Fi, int32_t Fo, const float *x, const float *W,
                             const float *bias, int32_t act, float *P, float *Z)
{
    AMP_REQUIRE(g && Fi > 0 && Fo > 0, "kipf_layer_fwd: bad arguments");
    if (g->n_rows == 0) return 0;
    AMP_REQUIRE(x && W && Z, "kipf_layer_fwd: null tensor");
    // hub rows (> kLongRow entries) would stall a whole workgroup at the chunk barrier: such graphs take
    // the two-kernel route, whose aggregation splits them into parallel segments
#ifndef KIPF_LAYER_BANDED
#define KIPF_LAYER_BANDED 1   // A/B builds: 0 = block-diagonal batches through the one-launch kernel like every other graph
#endif
    // a block-diagonal batch of small graphs: the aggregation gathers from LDS (agg.hip, csr_gather_banded64) and the dense step
    // follows as its own launch -- faster than the one launch that chases rows through HBM (profiles/r06_kipf_banded_ab.txt)
    const bool banded = KIPF_LAYER_BANDED && (Fi == 64 || Fi == 128) && kipf_gather_is_banded(g, false, Fi, x, P ? P : x);
#ifndef KIPF_BANDED_FUSED
#define KIPF_BANDED_FUSED 1   // A/B builds: 0 = banded aggregation and dense step as two launches also at 64 -> 64
#endif
    if (KIPF_BANDED_FUSED && banded && Fi == 64 && Fo == 64) {   // ... in ONE launch: P goes from LDS into the dense step (banded_fused.hip)
        const int rc = banded_agg_gemm64(g, false, g->coef, x, W, 0, bias, act, P, Z);
        if (rc >= 0) return rc;
    }
    if (!banded && fused_shape(Fi, Fo) && g->lp_fwd.n_long == 0)
        return fused_dispatch(g->rowptr, g->col, g->coef, x, Fi, Fo, W, 0, bias, act, P, Z, g->n_rows, g->n_cols);
    if (P == nullptr) {   // the caller keeps no tape of P (its reverse pass is athena_mp_kipf_layer_bwd)
        void *ws = nullptr;
        if (workspace(&ws, sizeof(float) * (size_t)g->n_rows * Fi, 5)) return 1;
        P = (float *)ws;
    }
    int rc = athena_mp_kipf_propagate_fwd(g, Fi, x, P);   // other widths: the two kernels back to back
    if (rc) return rc;
    return athena_mp_gemm_fwd(g->n_rows, Fi, Fo, P, W, bias, act, Z);
}

int athena_mp_kipf_layer_bwd_x(const athena_mp_graph *g, int32_t Fi, int32_t Fo, const float *dZ, const float *W,
                               int32_t exact, float *dX)
{
    AMP_REQUIRE(g && Fi > 0 && Fo > 0, "kipf_layer_bwd_x: bad arguments");
    if (g->n_cols == 0) return 0;
    AMP_REQUIRE(dZ && W && dX, "kipf_layer_bwd_x: null tensor");
    const bool banded = KIPF_LAYER_BANDED && (Fo == 64 || Fo == 128) && Fo <= Fi && kipf_gather_is_banded(g, true, Fo, dZ, dZ);
    if (KIPF_BANDED_FUSED && banded && Fi == 64 && Fo == 64) {
        const int rc = banded_agg_gemm64(g, true, exact ? g->t_coef : nullptr, dZ, W, 1, nullptr, ATHENA_MP_ACT_NONE, nullptr, dX);
        if (rc >= 0) return rc;
    }
    if (!banded && fused_shape(Fi, Fo) && g->lp_bwd.n_long == 0) // dX = (A^T dZ) . W : aggregate, then contract with B [N=Fi][K=Fo]
        return fused_dispatch(g->t_rowptr, g->t_src, exact ? g->t_coef : nullptr, dZ, Fo, Fi, W, 1, nullptr,
                              ATHENA_MP_ACT_NONE, nullptr, dX, g->n_cols, g->n_rows);
    void *ws = nullptr;
    if (Fo < Fi || banded) { // (A^T dZ) . W : the scatter moves the narrower rows
        if (workspace(&ws, sizeof(float) * (size_t)g->n_cols * Fo, 5)) return 1;
        int rc = athena_mp_kipf_propagate_bwd(g, Fo, dZ, (float *)ws, exact);
        if (rc) return rc;
        return athena_mp_gemm_dx(g->n_cols, Fi, Fo, (const float *)ws, W, dX);
    }
    if (workspace(&ws, sizeof(float) * (size_t)g->n_rows * Fi, 5)) return 1;
    int rc = athena_mp_gemm_dx(g->n_rows, Fi, Fo, dZ, W, (float *)ws);
    if (rc) return rc;
    return athena_mp_kipf_propagate_bwd(g, Fi, (const float *)ws, dX, exact);
}

/* Whole reverse pass of one Kipf step from the step's INPUT X instead of a stored P:
 *   dX = (A^T dZ) . W   (exact = 0: the reference's coefficient-free scatter; may be null: first layer of a network)
 *   dW = dZ . P^T with P = A^ X, evaluated as sum_u (A^^T dZ)[u] (x) X[u]
 * 128 -> 128 without hub rows: ONE launch (fused_dw.hip); otherwise one dual gather + two contractions. */
int athena_mp_kipf_layer_bwd(const athena_mp_graph *g, int32_t Fi, int32_t Fo, const float *dZ, const float *W,
                             const float *X, int32_t exact, float *dX, float *dW)
{
    AMP_REQUIRE(g && Fi > 0 && Fo > 0, "kipf_layer_bwd: bad arguments");
    AMP_REQUIRE(g->n_rows == g->n_cols, "kipf_layer_bwd: needs a square graph (%d x %d); shards use athena_mp_pull_gemm", g->n_rows,
                g->n_cols);
    AMP_REQUIRE(dW != nullptr, "kipf_layer_bwd: null dW");
    if (g->n_cols == 0) {
        AMP_HIP(hipMemsetAsync(dW, 0, sizeof(float) * (size_t)Fi * Fo, stream()));
        return 0;
    }
    AMP_REQUIRE(dZ && W && X, "kipf_layer_bwd: null tensor");
    if (amp::fused_dw_shape(Fi, Fo) && g->lp_bwd.n_long == 0 && (uintptr_t)dZ % 16 == 0 && (uintptr_t)W % 16 == 0)
        return amp::fused_dw_dispatch(g->t_rowptr, g->t_src, g->t_coef, dZ, W, X, exact, dX, dW, g->n_cols);
    void *ws = nullptr;
    const size_t rows = (size_t)g->n_cols * Fo;
    if (workspace(&ws, sizeof(float) * rows * (exact ? 1 : 2), 5)) return 1;
    float *qc = (float *)ws, *qp = exact ? qc : qc + rows;
    int rc = exact ? athena_mp_kipf_propagate_bwd(g, Fo, dZ, qc, 1) : athena_mp_kipf_propagate_bwd_dual(g, Fo, dZ, qp, qc);
    if (rc) return rc;
    rc = athena_mp_gemm_dw(g->n_cols, Fi, Fo, X, qc, dW);            // dW(Fo,Fi) = sum_u Qc[u] (x) X[u]
    if (rc || dX == nullptr) return rc;
    return athena_mp_gemm_dx(g->n_cols, Fi, Fo, qp, W, dX);           // dX = Qp . W
}

/* pull form for a row shard whose FORWARD rows list the sources (athena_amd/dist.py g_bwd):
 *   dX[v,:] = ( sum_{w in row v} [coef] dZ[col[w],:] ) . W                                        */
int athena_mp_pull_gemm(const athena_mp_graph *g, int32_t Fi, int32_t Fo, const float *dZ, const float *W,
                        int32_t exact, float *dX)
{
    AMP_REQUIRE(g && Fi > 0 && Fo > 0, "pull_gemm: bad arguments");
    if (g->n_rows == 0) return 0;
    AMP_REQUIRE(dZ && W && dX, "pull_gemm: null tensor");
    if (fused_shape(Fi, Fo) && g->lp_fwd.n_long == 0)
        return fused_dispatch(g->rowptr, g->col, exact ? g->coef : nullptr, dZ, Fo, Fi, W, 1, nullptr,
                              ATHENA_MP_ACT_NONE, nullptr, dX, g->n_rows, g->n_cols);
    void *ws = nullptr;
    if (Fi < Fo) { // A (dZ . W) : contract every source row first, gather the narrower result
        if (workspace(&ws, sizeof(float) * (size_t)g->n_cols * Fi, 5)) return 1;
        int rc = athena_mp_gemm_dx(g->n_cols, Fi, Fo, dZ, W, (float *)ws);
        if (rc) return rc;
        return gather_agg(g->rowptr, g->col, exact ? g->coef : nullptr, (const float *)ws, Fi, dX, Fi, g->n_rows, Fi,
                          &g->lp_fwd);
    }
    if (workspace(&ws, sizeof(float) * (size_t)g->n_rows * Fo, 5)) return 1;
    int rc = gather_agg(g->rowptr, g->col, exact ? g->coef : nullptr, dZ, Fo, (float *)ws, Fo, g->n_rows, Fo, &g->lp_fwd);
    if (rc) return rc;
    return athena_mp_gemm_dx(g->n_rows, Fi, Fo, (const float *)ws, W, dX);
}

} // extern "C"
