// Graph neural operator (athena's graph_nop layer) without the per-edge kernel tensor.
//
// Reference (athena_diffstruc_extd_sub_nop.f90): gno_kernel_eval materialises
//   kappa[:, e] = V relu(U dx_e + b_u) + b_v            ([F_out*F_in, E], :88-93)
// and gno_aggregate contracts it per CSR entry           (m_i += reshape(kappa_e) x_j, :367-378).
// At BASELINE config 4 that tensor is 246 GB.  Here the sum is re-associated:
//   h_e = [relu(U dx_e + b_u) ; 1]                                     (H+1 values per edge column)
//   S_i[k, q] = sum_{(j,e) in row i} h_e[k] x_j[q]                     ((H+1) x F_in per vertex)
//   m_i[o]    = sum_{k,q} Vaug[o + F_out*(q + F_in*k)] S_i[k, q]       (Vaug = [V | b_v], contiguous in theta)
// i.e. a gather + rank-1 accumulation per entry followed by ONE dense contraction with K = (H+1) F_in
// whose B operand is theta's own memory viewed row-major [(k,q)][o].  Same maths, different rounding
// (checked against the materialising oracle at 1e-5).  The backward passes reuse the two pieces:
//   dx_j  : T_j[k,o] = sum_{(i,e) in column j} h_e[k] g_i[o]  (pull over the transposed CSR), dx = T . B2
//   dVaug : S^T g   (dense reduction over vertices)            -> dV and db_v   (:291-300)
//   dU,db_u,dcoords : per entry dh[k] = sum_q G_i[k,q] x_j[q], G = g . Vmat^T, masked by relu' (:303-318, :195-210)
// S / T / G are produced per super-tile of vertices into a bounded HBM workspace.
#include <algorithm>
#include <iterator>
#include <map>
#include <type_traits>

#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace {

constexpr int kEB = 8; // entries staged per batch

// S[r][k*Fy + q] = sum_{w in row r} h_{eidx[w]}[k] * y[idx[w]][q],  k in [0,H] (h[H] = 1)
template <int EPT>
__global__ __launch_bounds__(256) void gno_outer_kernel(const int32_t *__restrict__ rowptr,
                                                        const int32_t *__restrict__ idx,
                                                        const int32_t *__restrict__ eidx,
                                                        const float *__restrict__ y, int Fy,
                                                        const float *__restrict__ coords,
                                                        const float *__restrict__ theta, int d, int H,
                                                        int r0, float *__restrict__ S)
{
    extern __shared__ float sm[];
    float *hb = sm;                    // [kEB][H+1]
    float *yb = sm + kEB * (H + 1);    // [kEB][Fy]
    const int row = r0 + blockIdx.x;
    const int R = (H + 1) * Fy;
    const float *U = theta, *bu = theta + (size_t)H * d;
    float acc[EPT];
    int kk[EPT], qq[EPT];
#pragma unroll
    for (int n = 0; n < EPT; ++n) {
        int e = threadIdx.x + 256 * n;
        acc[n] = 0.0f;
        kk[n] = e < R ? e / Fy : 0;
        qq[n] = e < R ? e - kk[n] * Fy : 0;
    }
    const int w0 = rowptr[row], w1 = rowptr[row + 1];
    for (int wb = w0; wb < w1; wb += kEB) {
        const int nb = min(kEB, w1 - wb);
        __syncthreads();
        for (int t = threadIdx.x; t < nb * (H + 1); t += 256) {
            int b = t / (H + 1), k = t - b * (H + 1);
            int e = eidx[wb + b];
            float hv = 0.0f;
            if (e >= 0) {
                if (k == H) hv = 1.0f;
                else {
                    const float *dx = coords + (size_t)e * d;
                    float s = 0.0f;
                    for (int j = 0; j < d; ++j) s = s + U[k + (size_t)H * j] * dx[j];
                    s = s + bu[k];
                    hv = s > 0.0f ? s : 0.0f;
                }
            }
            hb[b * (H + 1) + k] = hv;
        }
        for (int t = threadIdx.x; t < nb * Fy; t += 256) {
            int b = t / Fy, q = t - b * Fy;
            yb[b * Fy + q] = y[(size_t)idx[wb + b] * Fy + q];
        }
        __syncthreads();
        for (int b = 0; b < nb; ++b) {
#pragma unroll
            for (int n = 0; n < EPT; ++n) acc[n] = fmaf(hb[b * (H + 1) + kk[n]], yb[b * Fy + qq[n]], acc[n]);
        }
    }
    float *out = S + (size_t)blockIdx.x * R;
#pragma unroll
    for (int n = 0; n < EPT; ++n) {
        int e = threadIdx.x + 256 * n;
        if (e < R) out[e] = acc[n];
    }
}

// ---- MFMA form of the outer-product accumulation (H, F multiples of 32, <= 64; d <= 4) -----------
// One wave per vertex.  S_i = Hm^T X with the row's entries as the contraction index:
//   v_mfma_f32_32x32x2_f32: A[k][e] = h_e[k] (computed on the fly by the lane that owns k),
//                           B[e][q] = y_j[q] (128 B coalesced loads, two entries per step).
// The bias row S_i[H,:] = sum_j y_j is accumulated on the VALU from the same loads.
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int HT, int QT>
__global__ __launch_bounds__(256) void gno_outer_mfma_kernel(const int32_t *__restrict__ rowptr,
                                                             const int32_t *__restrict__ idx,
                                                             const int32_t *__restrict__ eidx,
                                                             const float *__restrict__ y,
                                                             const float *__restrict__ coords,
                                                             const float *__restrict__ theta, int d, int r0,
                                                             int n_rows_tile, float *__restrict__ S)
{
    constexpr int H = 32 * HT, Fy = 32 * QT, R = (H + 1) * Fy;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r31 = lane & 31, h = lane >> 5;
    const int lr = blockIdx.x * 4 + wave;
    if (lr >= n_rows_tile) return;
    const int row = r0 + lr;
    // this lane's rows of U and b_u (k = 32 t + r31)
    float Uk[HT][4], bk[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) {
        bk[t] = theta[(size_t)H * d + 32 * t + r31];
#pragma unroll
        for (int j = 0; j < 4; ++j) Uk[t][j] = j < d ? theta[(32 * t + r31) + (size_t)H * j] : 0.0f;
    }
    f32x16 acc[HT][QT];
#pragma unroll
    for (int a = 0; a < HT; ++a)
#pragma unroll
        for (int b = 0; b < QT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    float sb[QT];
#pragma unroll
    for (int b = 0; b < QT; ++b) sb[b] = 0.0f;

    const int w0 = rowptr[row], w1 = rowptr[row + 1];
    for (int wb = w0; wb < w1; wb += 64) {
        const int nb = min(64, w1 - wb);
        int my_j = -1, my_e = -1;
        if (lane < nb) { my_j = idx[wb + lane]; my_e = eidx[wb + lane]; }
        constexpr int U4 = 4;
        for (int s0 = 0; s0 < (nb + 1) / 2; s0 += U4) {
            float yv[U4][QT], dx[U4][4];
            bool ok[U4];
#pragma unroll
            for (int u = 0; u < U4; ++u) {
                const int ent = 2 * (s0 + u) + h;
                const int j = __shfl(my_j, ent), e = __shfl(my_e, ent);
                ok[u] = ent < nb && e >= 0;
#pragma unroll
                for (int b = 0; b < QT; ++b) yv[u][b] = ok[u] ? y[(size_t)j * Fy + 32 * b + r31] : 0.0f;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) dx[u][jj] = (ok[u] && jj < d) ? coords[(size_t)e * d + jj] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < U4; ++u) {
                float hv[HT];
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    float sacc = 0.0f;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) sacc = sacc + Uk[t][jj] * dx[u][jj];   // j-ordered, as :88-90
                    sacc = sacc + bk[t];
                    hv[t] = (ok[u] && sacc > 0.0f) ? sacc : 0.0f;
                }
#pragma unroll
                for (int b = 0; b < QT; ++b) sb[b] = sb[b] + yv[u][b];
#pragma unroll
                for (int t = 0; t < HT; ++t)
#pragma unroll
                    for (int b = 0; b < QT; ++b)
                        acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(hv[t], yv[u][b], acc[t][b], 0, 0, 0);
            }
        }
    }
    float *out = S + (size_t)lr * R;
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int b = 0; b < QT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
                out[(size_t)k * Fy + 32 * b + r31] = acc[t][b][r];
            }
#pragma unroll
    for (int b = 0; b < QT; ++b) {
        const float tot = sb[b] + __shfl_xor(sb[b], 32);
        if (h == 0) out[(size_t)H * Fy + 32 * b + r31] = tot;
    }
}

// ---- fused aggregate for H = 64, gathered width 64, output width 64 (BASELINE configs[3]) ---------------
// out[r,:] = S_r . Vaug without S ever reaching HBM.  A 16-wave workgroup owns 16 vertices; the hidden index
// is processed in two halves so that the 16 x (33 x 64) half of S fits LDS (135 KB):
//   phase 1  wave w builds S_half of vertex w exactly like gno_outer_mfma_kernel (32x32x2 MFMA over the
//            row's entries, h_e on the fly, neighbour rows re-gathered per half: 2 x 7.6 GB, L2-hot the second
//            time, instead of the 2 x 33 GB S round trip) and parks it in LDS; the workgroup is persistent and
//            fetches the next tile's row pointers and ids under the last contraction;
//   phase 2  out^T[o, r] += sum_kq Vaug[kq, o] S[r, kq] on 16x16x4 MFMAs with the VERTEX on the column axis:
//            B = S[r, 4s + g] (one LDS word per lane), A = Vaug[4s + g, 4n .. 4n+3] -- one 16 B load from L2
//            feeds four MFMAs whose output tiles interleave o = 4m + c, so a lane ends up with 16 CONSECUTIVE
//            outputs of its vertex; the 4 s-steps of a wave's slice are split over the 16 waves.
// The per-wave partial sums are added through LDS in wave order (deterministic).
typedef float v4f_g __attribute__((ext_vector_type(4)));
typedef unsigned int v4u_g __attribute__((ext_vector_type(4)));
// how gno_pc_kernel<true> writes the S it keeps: 0 nontemporal global stores, 1 plain stores, 2 buffer stores with the cache
// bits GNO_SAVE_AUX (1 sc0, 2 nt, 16 sc1), 3 none (timing only).  configs[3], ms per launch, two runs each on one box
// (scripts/gpu_keeps_ab.sh): plain 14.65 / 14.74, nt global 14.21 / 14.28, buffer nt 14.05 / 14.08, buffer sc0 sc1 nt 14.07 / 14.09,
// buffer sc0 sc1 14.39 / 14.40, none 11.88 / 11.89 (= the kernel without the copy: the LDS reads of the copy cost nothing,
// the 33 GB of writes make the launch HBM bound: 63 GB in 14.05 ms = 4.5 TB/s)
#define GNO_SAVE_MODE 2
#define GNO_PX_NT_LOAD 1   // gno_px_gather_kernel reads the partials (15 GB, read once) with nontemporal loads: A/B in profiles/r04_c4_px_store_ab.txt
#define GNO_PX_RB_AUX 0   // cache bits of the read-back of the kh = 0 partial (2 = nt: A/B in profiles/r04_c4_px_one_array_ab.txt)
#define GNO_PX_AUX 0   // cache bits of the per-entry partials' stores (1 sc0, 2 nt, 16 sc1): none -- A/B in profiles/r04_c4_px_one_array_ab.txt
#define GNO_SAVE_AUX 2
#ifndef GNO_FV
#define GNO_FV 0   // timing-only variants (scripts/build_variants.sh), bit mask: 1 no sparse loop, 2 no V loads, 4 no S reads, 8 no contraction, 16 no cross-wave reduction (gno_fused_kernel); 32 idle producers, 64 idle consumers, 8192 half the gathers (gno_pc_kernel); 16384 gno_px_gather_kernel reads the partials as one stream
#endif
constexpr int kGF = 64, kGH = 64, kGRows = 16, kGSP = 33 * kGF + 4;   // LDS row pitch of S_half

__global__ __launch_bounds__(1024) void gno_fused_kernel(const int32_t *__restrict__ rowptr,
                                                         const int32_t *__restrict__ idx,
                                                         const int32_t *__restrict__ eidx,
                                                         const float *__restrict__ y,
                                                         const float *__restrict__ coords,
                                                         const float *__restrict__ theta, int d,
                                                         const float *__restrict__ Vaug, int n_rows,
                                                         const int32_t *__restrict__ perm,
                                                         float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float Sh[];   // [16][kGSP]; reused for the final reduction
    // (Measured and dropped in round 2: the sparse phase's row and coordinate loads through buffer descriptors with
    // 32-bit offsets and dead slots pointed past the buffer -- 18.3 ms against 17.5: as in fused.hip, unpredicated loads
    // of dead slots cost more than the address arithmetic they save.)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r31 = lane & 31, h = lane >> 5;
    const int n = lane & 15, g = lane >> 4;
    const int n_tiles = (n_rows + kGRows - 1) / kGRows;

    // persistent over tiles; the next tile's row pointers and first 64 (neighbour, edge) ids are fetched while
    // the current tile is on the matrix cores, so phase 1 starts with one dependent load instead of three
    int tile = blockIdx.x;
    int w0 = 0, w1 = 0, my_j = -1, my_e = -1;
    auto fetch_ids = [&](int tl) {
        const int slot = tl * kGRows + wave;
        w0 = w1 = 0;
        my_j = my_e = -1;
        if (tl < n_tiles && slot < n_rows) {
            const int row = perm[slot];          // slots walk the vertices longest row first
            w0 = rowptr[row];
            w1 = rowptr[row + 1];
            if (lane < w1 - w0) { my_j = idx[w0 + lane]; my_e = eidx[w0 + lane]; }
        }
    };
    fetch_ids(tile);
    for (; tile < n_tiles; tile += gridDim.x) {
        const int r0 = tile * kGRows;
        v4f_g om[4];                                     // phase-2 accumulators: tile c, register r -> o = 16g + 4r + c
#pragma unroll
        for (int c = 0; c < 4; ++c) om[c] = v4f_g{0.0f, 0.0f, 0.0f, 0.0f};
        float *srow = Sh + wave * kGSP;
        const int cw0 = w0, cw1 = w1, cj = my_j, ce = my_e;   // this tile's row (ids of its first 64 entries)
        for (int half = 0; half < 2; ++half) {
            // ---------------- phase 1: S_half of vertex `wave` (hidden units 32*half + r31) ----------------
            // the neighbour rows are re-read for the second half (they are L2-hot); holding both halves in
            // registers does not fit the 128 registers a 16-wave workgroup leaves per lane
            float Uk[4], bk;   // this lane's hidden unit of this half: row of U and b_u (L1-hot reload per half)
            {
                const int k = 32 * half + r31;
                bk = theta[(size_t)kGH * d + k];
#pragma unroll
                for (int j = 0; j < 4; ++j) Uk[j] = j < d ? theta[k + (size_t)kGH * j] : 0.0f;
            }
            f32x16 acc[2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;
            float sb[2] = {0.0f, 0.0f};
            for (int wb = cw0; wb < ((GNO_FV & 1) ? cw0 : cw1); wb += 64) {
                const int nb = min(64, cw1 - wb);
                int bj = cj, be = ce;
                if (wb != cw0) {   // rows longer than 64 entries: later blocks are fetched here
                    bj = be = -1;
                    if (lane < nb) { bj = idx[wb + lane]; be = eidx[wb + lane]; }
                }
                constexpr int U4 = 4;
                for (int s0 = 0; s0 < (nb + 1) / 2; s0 += U4) {
                    float yv[U4][2], dx[U4][4];
                    bool ok[U4];
#pragma unroll
                    for (int u = 0; u < U4; ++u) {
                        const int ent = 2 * (s0 + u) + h;
                        const int j = __shfl(bj, ent), e = __shfl(be, ent);
                        ok[u] = ent < nb && e >= 0;
#pragma unroll
                        for (int b = 0; b < 2; ++b) yv[u][b] = ok[u] ? y[(size_t)j * kGF + 32 * b + r31] : 0.0f;
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) dx[u][jj] = (ok[u] && jj < d) ? coords[(size_t)e * d + jj] : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < U4; ++u) {
                        float sacc = 0.0f;
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) sacc = sacc + Uk[jj] * dx[u][jj];   // j-ordered, as :88-90
                        sacc = sacc + bk;
                        const float hv = (ok[u] && sacc > 0.0f) ? sacc : 0.0f;
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            sb[b] = sb[b] + yv[u][b];
                            acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(hv, yv[u][b], acc[b], 0, 0, 0);
                        }
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kl = (r & 3) + 8 * (r >> 2) + 4 * h;
                    srow[kl * kGF + 32 * b + r31] = acc[b][r];
                }
            if (half == 1) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float tot = sb[b] + __shfl_xor(sb[b], 32);
                    if (h == 0) srow[32 * kGF + 32 * b + r31] = tot;     // bias row k = H
                }
                fetch_ids(tile + gridDim.x);   // next tile's ids fly under the last contraction
            }
            __syncthreads();
            // ---------------- phase 2: contraction of this half ----------------
            const int n_steps = (half == 0 ? 32 * kGF : 33 * kGF) / 4;    // 512 / 528 k-steps of 4
            const int per_wave = (n_steps + 15) / 16;
            const int s_beg = wave * per_wave, s_end = min(n_steps, s_beg + per_wave);
            const float *vbase = Vaug + (size_t)half * 32 * kGF * kGF;     // rows (kq) of this half
            const float *sl = Sh + n * kGSP + g;
            // (Measured and dropped in round 2: a software pipeline over two register sets, the V fragments and S words of
            // round r+1 in flight under the MFMAs of round r -- 18.1-18.8 ms against 17.5 ms for this loop, with 4-11
            // spilled registers at 3-4 steps per round; the contraction is not waiting for its operands.)
            // PMC (profiles/r02_c4_gno_pmc.txt): 3.5e9 vector instructions against 6.7e8 MFMAs per launch -- the kernel is
            // bound by instruction ISSUE (matrix pipe busy 56 %, vector issue ~37 % of the SIMD cycles), so the loop below
            // carries no per-step arithmetic: every wave has exactly 32 steps (+1 in the second half: 528 = 16 x 33), the
            // operand addresses of a round are one base pointer plus compile-time offsets (1 KB apart in V, 16 B apart in
            // S), no clamps, no per-step conditions.
            constexpr int UN = 8;
            const float *ap = vbase + (size_t)(4 * s_beg + g) * kGF + 4 * n;
            const float *bp = sl + 4 * s_beg;
#pragma unroll 1
            for (int r = 0; r < ((GNO_FV & 8) ? 0 : 4); ++r) {
                v4f_g a[UN];
                float b[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
#if GNO_FV & 2
                    a[u] = v4f_g{(float)u, (float)r, (float)lane, 1.0f};
                    asm volatile("" : "+v"(a[u]));
#else
                    a[u] = *reinterpret_cast<const v4f_g *>(ap + (size_t)u * 4 * kGF);
#endif
#if GNO_FV & 4
                    b[u] = (float)(u + r);
                    asm volatile("" : "+v"(b[u]));
#else
                    b[u] = bp[4 * u];
#endif
                }
#pragma unroll
                for (int u = 0; u < UN; ++u)
#pragma unroll
                    for (int c = 0; c < 4; ++c) om[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][c], b[u], om[c], 0, 0, 0);
                ap += (size_t)UN * 4 * kGF;
                bp += 4 * UN;
            }
            if (s_end - s_beg > 32) {   // second half: the 33rd step (bias row)
                const v4f_g a1 = *reinterpret_cast<const v4f_g *>(ap);
                const float b1 = bp[0];
#pragma unroll
                for (int c = 0; c < 4; ++c) om[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c], b1, om[c], 0, 0, 0);
            }
            __syncthreads();
        }
        // ---------------- cross-wave reduction (fixed order) and store ----------------
        // lane (n = vertex, g): om[c][r] = out[vertex][16g + 4r + c]
        float *red = Sh;   // [16 waves][16 vertices][64]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const v4f_g v = {om[0][r], om[1][r], om[2][r], om[3][r]};
            *reinterpret_cast<v4f_g *>(red + ((size_t)wave * kGRows + n) * kGF + 16 * g + 4 * r) = v;
        }
        __syncthreads();
        {
            const int t = threadIdx.x;                 // 1024 threads = 16 vertices x 64 outputs
            float sum = red[t];
#pragma unroll
            for (int w = 1; w < ((GNO_FV & 16) ? 1 : 16); ++w) sum = sum + red[(size_t)w * kGRows * kGF + t];
            const int v = t >> 6;
            if (r0 + v < n_rows) out[(size_t)perm[r0 + v] * kGF + (t & 63)] = sum;
        }
        __syncthreads();   // red is S of the next tile
    }
}

bool gno_fused_shape(int H, int Fy, int Fout, int d)
{
    return H == kGH && Fy == kGF && Fout == kGF && d <= 4;
}

// vertices ordered by row length, longest first (stable counting sort on the host, once per graph and CSR).  The permutation
// and the two class counts that go with it (rows of more than 32 / more than 16 entries: the first slots of the permutation)
// are ONE cached unit: a caller that asks for the counts gets them whether or not it was the one that built the permutation.
struct LenCounts {
    int32_t n_long, n_mid;
};
std::map<const int32_t *, LenCounts> g_len_counts;   // keyed by the cached device permutation

int length_order(const int32_t *rowptr_dev, int n_rows, int32_t **perm_dev, int32_t *n_long = nullptr, int32_t *n_mid = nullptr)
{
    if (*perm_dev) {
        const auto it = g_len_counts.find(*perm_dev);
        if (it != g_len_counts.end()) {
            if (n_long) *n_long = it->second.n_long;
            if (n_mid) *n_mid = it->second.n_mid;
            return 0;
        }
        (void)hipFree(*perm_dev);   // a permutation without its counts (never left by this function): rebuild both
        *perm_dev = nullptr;
    }
    std::vector<int32_t> rp((size_t)n_rows + 1);
    AMP_HIP(hipMemcpyAsync(rp.data(), rowptr_dev, sizeof(int32_t) * rp.size(), hipMemcpyDeviceToHost, amp::stream()));
    AMP_HIP(hipStreamSynchronize(amp::stream()));
    int32_t mx = 0;
    int32_t longer = 0, mid = 0;
    for (int i = 0; i < n_rows; ++i) {
        mx = std::max(mx, rp[i + 1] - rp[i]);
        longer += rp[i + 1] - rp[i] > 32;
        mid += rp[i + 1] - rp[i] > 16;
    }
    std::vector<int32_t> start((size_t)mx + 2, 0), perm((size_t)std::max(n_rows, 1));
    for (int i = 0; i < n_rows; ++i) start[(size_t)(mx - (rp[i + 1] - rp[i])) + 1]++;
    for (int l = 0; l <= mx; ++l) start[(size_t)l + 1] += start[l];
    for (int i = 0; i < n_rows; ++i) perm[start[(size_t)(mx - (rp[i + 1] - rp[i]))]++] = i;
    int32_t *dev = nullptr;
    AMP_HIP(hipMalloc((void **)&dev, sizeof(int32_t) * perm.size()));
    if (hipMemcpy(dev, perm.data(), sizeof(int32_t) * perm.size(), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(dev);         // nothing half-built stays behind
        amp::set_error("gno: upload of the row-length order failed");
        return 1;
    }
    for (auto it = g_len_counts.begin(); it != g_len_counts.end();)   // a freed permutation's address may come back
        it = it->first == dev ? g_len_counts.erase(it) : std::next(it);
    g_len_counts[dev] = LenCounts{longer, mid};
    *perm_dev = dev;
    if (n_long) *n_long = longer;
    if (n_mid) *n_mid = mid;
    return 0;
}

int launch_gno_pc(const int32_t *rowptr, const int32_t *idx, const int32_t *eidx, const float *y, const float *coords,
                  const float *theta, int d, const float *Vaug, int n_rows, const int32_t *perm, float *out, size_t y_bytes,
                  size_t c_bytes, size_t id_bytes, float *save = nullptr);

// the producer / consumer kernels address their gathers through buffer descriptors (32-bit byte offsets, d <= 3)
bool gno_pc_route(int d, int y_rows, int n_edge_cols, int64_t nnz)
{
    const size_t y_bytes = sizeof(float) * kGF * (size_t)y_rows, c_bytes = sizeof(float) * (size_t)d * n_edge_cols,
                 id_bytes = sizeof(int32_t) * (size_t)nnz;
    const size_t lim = 0xFFFFE000ull;   // below the kernel's dead-slot offset
    return d <= 3 && y_bytes < lim && c_bytes < lim && id_bytes < lim;   // (17.2 ms one phase at a time against 11.9: DESIGN.md 3.5)
}

int launch_gno_fused(const int32_t *rowptr, const int32_t *idx, const int32_t *eidx, const float *y,
                     const float *coords, const float *theta, int d, const float *Vaug, int n_rows,
                     int32_t **perm_cache, float *out, int y_rows, int n_edge_cols, int64_t nnz, int32_t *n_long = nullptr,
                     int32_t *n_mid = nullptr, float *save = nullptr)
{
    if (n_rows > 0 && length_order(rowptr, n_rows, perm_cache, n_long, n_mid)) return 1;
    const size_t y_bytes = sizeof(float) * kGF * (size_t)y_rows, c_bytes = sizeof(float) * (size_t)d * n_edge_cols,
                 id_bytes = sizeof(int32_t) * (size_t)nnz;
    if (gno_pc_route(d, y_rows, n_edge_cols, nnz))
        return launch_gno_pc(rowptr, idx, eidx, y, coords, theta, d, Vaug, n_rows, *perm_cache, out, y_bytes, c_bytes, id_bytes,
                             save);
    if (save) {
        amp::set_error("gno_aggregate_fwd_save: this shape does not take the kernel that keeps S (athena_mp_gno_saved_bytes says so)");
        return 2;
    }
    constexpr size_t lds = sizeof(float) * (size_t)kGRows * kGSP;
    static amp::PerDeviceFlag attr;
    if (!attr.get()) {
        AMP_HIP(hipFuncSetAttribute((const void *)gno_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.get() = true;
    }
    if (n_rows <= 0) return 0;
    const int n_tiles = (n_rows + kGRows - 1) / kGRows;
    hipLaunchKernelGGL(gno_fused_kernel, dim3(std::min(n_tiles, 256)), dim3(1024), lds, amp::stream(), rowptr, idx,
                       eidx, y, coords, theta, d, Vaug, n_rows, (const int32_t *)*perm_cache, out);
    AMP_LAUNCH_CHECK();
    return 0;
}

// ---- producer / consumer form of the fused aggregate (same shapes; d <= 3) -------------------------------------
// Timing splits of gno_fused_kernel at C4 (GNO_FV builds, profiles/r02_c4_gno_variants.txt): its sparse loop costs
// 5.2 ms, its contraction 8.1 ms on the matrix pipe alone (+1.8 ms of exposed V loads), the skeleton 2 ms -- and they ADD
// (17.2 ms), because every wave of the one resident workgroup is in the same phase.  Here the two phases run side by
// side in one 16-wave workgroup that owns 32 vertices per tile:
//   waves 0-7   PRODUCERS, four vertices each.  S is built in eight PIECES per tile, piece (c, kh) = hidden units
//               32 kh .. +31  x  gathered features 16 c .. +15 (512 words per vertex), on 16x16x4 MFMAs only:
//                 h^T[slot][hid] = [dx_e ; 1] . [U ; b_u]      one MFMA per 16 entries and 16 hidden units (K = d + 1);
//                                                              relu on its four result registers;
//                 S[hid][q]     += h[hid][e] x_j[q]            the result registers ARE the A operand: register r of
//                                                              lane group g is the entry in slot 4g + r, so MFMA step r
//                                                              contracts entries 4r .. 4r+3 (slot i holds entry
//                                                              4 (i & 3) + (i >> 2)) and a row of nb entries takes
//                                                              ceil(nb / 4) steps per 16 hidden units.
//               Per entry and piece the vector pipe sees one 4-byte load and a share of a shuffle -- the per-lane
//               h arithmetic of gno_fused_kernel (20 vector instructions per entry and pass) is gone.
//   waves 8-11  CONSUMERS, one per SIMD, wave = output tile ot of 16: out^T[o, v] += V[kq, o] S[v, kq] for both groups
//               of 16 vertices per 16-byte V load, so V streams from L2 once per 32 vertices (1.06 MB per tile: half
//               the L2 traffic per vertex of gno_fused_kernel); S comes from LDS as one ds_read_b128 per four steps
//               (row pitch 520 words: conflict-free for the lane groups of b128 reads).  V is re-laid once per call in
//               exactly the order the waves stream it (gno_vrelay_kernel), 1 KB per load instruction.
// Pieces are double-buffered in LDS (2 x 65 KB) and handed over by ONE workgroup barrier per piece; the bias row
// (sum of x_j) is a ninth, 64-word piece.  Twelve waves, not sixteen: 168 registers per lane hold a producer's four
// operand sets without spilling (a spill reload is a vector-memory operation: its wait drains every prefetch).
constexpr int kPV = 32, kPPitch = 520, kPBPitch = 72;
constexpr int kPcLdsFloats = 2 * kPV * kPPitch + 2 * kPV * kPBPitch;
constexpr int kPcThreads = 768;   // 8 producer + 4 consumer waves: three per SIMD, 168 registers each
constexpr int kVpFloats = 64 * 64 * 64 + 64 * 64 + 1024;   // pieces, bias piece, slack for the look-ahead loads

// Vp[pc][ot][gi][lane][s] = Vin[kq][o]:  o = 16 ot + lane % 16, L = 16 gi + 4 (lane / 16) + s (gi < 32) the position
// inside the piece as the producers lay it down (row = L / 16 = 16 t + 4 r + g <-> hidden unit 32 kh + 16 t + 4 g + r).
__global__ void gno_vrelay_kernel(const float *__restrict__ Vin, float *__restrict__ Vp)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 65 * 64 * 64) return;
    if (t < 64 * 64 * 64) {
        const int s = t & 3, lane = (t >> 2) & 63, gi = (t >> 8) & 31, ot = (t >> 13) & 3, pc = t >> 15;
        const int o = 16 * ot + (lane & 15), g = lane >> 4;
        const int L = 16 * gi + 4 * g + s;
        const int row = L >> 4, q = L & 15;
        const int tp = row >> 4, r = (row >> 2) & 3, gg = row & 3;
        const int k = 32 * (pc >> 2) + 16 * tp + 4 * gg + r;   // pieces in the order (kh, c): pc = 4 kh + c
        const int qg = 16 * (pc & 3) + q;
        Vp[t] = Vin[(size_t)(k * 64 + qg) * 64 + o];
    } else {
        const int u = t - 64 * 64 * 64;   // [ot][gi (4)][lane][s]
        const int s = u & 3, lane = (u >> 2) & 63, gi = (u >> 8) & 3, ot = u >> 10;
        const int o = 16 * ot + (lane & 15), g = lane >> 4;
        const int qg = 16 * gi + 4 * g + s;
        Vp[t] = Vin[(size_t)(64 * 64 + qg) * 64 + o];
    }
}

struct GnoIds {   // one tile's rows as a producer wave holds them
    int J0, J1, E0, E1;                       // first 32 (neighbour, edge column) ids: lane = (vertex lane / 16, entry lane % 16)
    int row[4], w0[4], len[4];                // wave-uniform
    bool ok[4];
};
struct GnoLoads {   // one vertex's operands for one piece: features of up to 32 entries
    float x[8];
};

// What a producer wave of the fused GNO kernels knows and does (gno_pc_kernel: aggregate / dx; gno_stg_kernel: S^T g).
// Every gather below is UNCONDITIONAL (buffer loads; a dead slot's offset lies beyond the buffer and reads 0): a load
// inside a branch makes the compiler's wait-count bookkeeping give up and drain the queue at every use (first version:
// 42 x s_waitcnt vmcnt(0), producers alone 9.9 ms), which is the latency these kernels exist to hide.
struct GnoProd {
    static constexpr uint32_t kDead = 0xFFFFF000u;   // beyond every buffer, and still beyond with a lane's few bytes added
    int lane, n, g, p, d, n_rows, n_tiles, vpw;   // vpw: vertices of a tile per producer wave (4; 2 in gno_dh_pc_kernel<2, 2>)
    const int32_t *perm, *rowptr, *idx, *eidx;
    __amdgpu_buffer_rsrc_t yrs, crs, jrs, ers;
    // h MFMA per 16 hidden units: A lane (slot n, K index g) = coordinate g of the slot's edge (1 at g = d), B lane
    // (hid = n, K index g) = U[hid][g] (b_u[hid] at g = d, 0 beyond).  A dead slot's h is relu(b_u): finite, and it
    // meets x = 0.
    float Ub[4];
    bool g_is_d;
    int eslot;            // the entry that sits in slot n of a 16-entry block
    uint32_t n4, g4;

    __device__ __forceinline__ void init(int wave, int lane_, const int32_t *rowptr_, const int32_t *idx_, const int32_t *eidx_,
                                         const float *y, const float *coords, const float *theta, int d_, int n_rows_,
                                         const int32_t *perm_, uint32_t y_bytes, uint32_t c_bytes, uint32_t id_bytes, int vpw_ = 4)
    {
        lane = lane_; n = lane & 15; g = lane >> 4; d = d_; n_rows = n_rows_; vpw = vpw_; n_tiles = (n_rows + 8 * vpw - 1) / (8 * vpw);
        perm = perm_; rowptr = rowptr_; idx = idx_; eidx = eidx_;
        p = __builtin_amdgcn_readfirstlane(wave);
        yrs = __builtin_amdgcn_make_buffer_rsrc((void *)y, 0, (int)y_bytes, 0x00020000);
        crs = __builtin_amdgcn_make_buffer_rsrc((void *)coords, 0, (int)c_bytes, 0x00020000);
        jrs = __builtin_amdgcn_make_buffer_rsrc((void *)idx, 0, (int)id_bytes, 0x00020000);
        ers = __builtin_amdgcn_make_buffer_rsrc((void *)eidx, 0, (int)id_bytes, 0x00020000);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int hid = 16 * t + n;
            Ub[t] = g < d ? theta[hid + kGH * g] : (g == d ? theta[(size_t)kGH * d + hid] : 0.0f);
        }
        g_is_d = g == d;
        eslot = 4 * (n & 3) + (n >> 2);
        n4 = 4 * n; g4 = 4 * g;
    }
    // a tile's ids arrive in three dependent steps; row numbers and row pointers are wave-uniform (scalar loads), the
    // entries lane-parallel: lane (vertex g of the four, entry n).  What is kept of an entry is its two BYTE OFFSETS
    // (feature row, coordinate row), kDead for a slot beyond the row or an entry without an edge column (oracle: e < 0
    // contributes nothing).
    __device__ __forceinline__ void ids_rows(int tl, GnoIds &I) const
    {
#pragma unroll
        for (int vi = 0; vi < 4; ++vi) {
            const int slot = tl * (8 * vpw) + vpw * p + vi;
            I.ok[vi] = vi < vpw && tl < n_tiles && slot < n_rows;
            I.row[vi] = perm[I.ok[vi] ? slot : 0];
        }
    }
    __device__ __forceinline__ void ids_ptrs(GnoIds &I) const
    {
#pragma unroll
        for (int vi = 0; vi < 4; ++vi) {
            I.w0[vi] = rowptr[I.row[vi]];
            I.len[vi] = I.ok[vi] ? rowptr[I.row[vi] + 1] - I.w0[vi] : 0;
        }
    }
    // lane group g's element of a wave-uniform array of four.  Written as three selects with the compiler kept from seeing
    // them as ONE indexed read: it otherwise parks the array in scratch memory and reads it back with a per-lane index --
    // 2.7 GB of scratch writes per launch of gno_pc_kernel at configs[3] (WRITE_SIZE 3.1e6 KiB for 0.5 GB of output).
    __device__ __forceinline__ int by_group(const int (&a)[4]) const
    {
        int v = a[0];
        v = g >= 1 ? a[1] : v;
        asm volatile("" : "+v"(v));
        v = g >= 2 ? a[2] : v;
        asm volatile("" : "+v"(v));
        v = g >= 3 ? a[3] : v;
        return v;
    }
    __device__ __forceinline__ void ids_entries(GnoIds &I) const
    {
        const int w0 = by_group(I.w0), len = by_group(I.len);
        const uint32_t o0 = n < len ? 4u * (uint32_t)(w0 + n) : kDead, o1 = n + 16 < len ? 4u * (uint32_t)(w0 + n + 16) : kDead;
        I.J0 = __builtin_amdgcn_raw_buffer_load_b32(jrs, (int)o0, 0, 0);
        I.E0 = __builtin_amdgcn_raw_buffer_load_b32(ers, (int)o0, 0, 0);
        I.J1 = __builtin_amdgcn_raw_buffer_load_b32(jrs, (int)o1, 0, 0);
        I.E1 = __builtin_amdgcn_raw_buffer_load_b32(ers, (int)o1, 0, 0);
    }
    __device__ __forceinline__ void to_offsets(int &J, int &E, bool inrow) const
    {
        const bool alive = inrow && E >= 0;
        J = (int)(alive ? (uint32_t)J * (4u * kGF) : kDead);
        E = (int)(alive ? (uint32_t)E * (4u * (uint32_t)d) : kDead);
    }
    __device__ __forceinline__ void ids_finish(GnoIds &I) const
    {
        const int len = by_group(I.len);
        to_offsets(I.J0, I.E0, n < len);
        to_offsets(I.J1, I.E1, n + 16 < len);
    }
    // operand loads of one vertex for the feature quarter c; (J0, J1) hold the offsets of its 32 entries in lane group
    // srcg.  Per load: one shuffle, one add.
    __device__ __forceinline__ void issue(GnoLoads &L, int J0, int J1, int srcg, int c, bool second) const
    {
        const int src = 16 * srcg;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t j = (uint32_t)__shfl(J0, src + 4 * r + g) + n4;
            L.x[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, (int)j, 64 * c, 0));
        }
        if (second) {   // entries 16 .. 31: only tiles whose rows are that long ask for them
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t j = (uint32_t)__shfl(J1, src + 4 * r + g) + n4;
                L.x[4 + r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, (int)j, 64 * c, 0));
            }
        }
    }
    // the edge coordinates of a vertex's 32 slots (they do not depend on the piece)
    __device__ __forceinline__ void load_cv(float (&cv)[2], int E0, int E1, int srcg) const
    {
        const int src = 16 * srcg;
        const uint32_t es0 = (uint32_t)__shfl(E0, src + eslot) + g4, es1 = (uint32_t)__shfl(E1, src + eslot) + g4;
        cv[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(crs, (int)es0, 0, 0));
        cv[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(crs, (int)es1, 0, 0));
    }
    // one instruction per value, as an INTEGER max (a float with its sign bit set is a negative integer): fmaxf and
    // fmed3 cost two (they quiet their operand first), and an inline-asm v_max is invisible to the hazard recogniser --
    // no wait states between the MFMA and the read of its result (wrong rows at C4 size)
    static __device__ __forceinline__ void relu4(v4f_g &h)
    {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t = h[r];   // (__builtin_bit_cast of a vector ELEMENT reads element 0 whatever r is)
            h[r] = __int_as_float(max(__float_as_int(t), 0));   // v_max_i32
        }
    }
    // S piece of one vertex from its operands: acc[t][r2] = S[hid = 16 t + 4 g + r2][q = n], NST groups of four entries.
    // NST is a compile-time constant: branches around single steps (or a switch that falls through them) make the
    // compiler copy the accumulators between register sets at every step.
    // HC: the relu'd h of the first 16 entries depends on kh only -- 0: computed here; 1: computed here and kept in hc;
    //     2: taken from hc (the pieces of one kh follow each other in gno_pc_kernel)
    template <int NST, int HC = 0>
    __device__ __forceinline__ void compute(const GnoLoads &L, const float (&cvs)[2], float ub0, float ub1, v4f_g (&acc)[2],
                                            float &bs, v4f_g *hc = nullptr) const
    {
        const v4f_g z = {0.0f, 0.0f, 0.0f, 0.0f};
        {
            v4f_g h0, h1;
            if constexpr (HC == 2) {
                h0 = hc[0];
                h1 = hc[1];
            } else {
                const float cv = g_is_d ? 1.0f : cvs[0];
                h0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cv, ub0, z, 0, 0, 0);
                h1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cv, ub1, z, 0, 0, 0);
                relu4(h0);
                relu4(h1);
                if constexpr (HC == 1) {
                    hc[0] = h0;
                    hc[1] = h1;
                }
            }
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(h0[0], L.x[0], z, 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(h1[0], L.x[0], z, 0, 0, 0);
#pragma unroll
            for (int r = 1; r < (NST < 4 ? NST : 4); ++r) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(h0[r], L.x[r], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(h1[r], L.x[r], acc[1], 0, 0, 0);
            }
            bs = (L.x[0] + L.x[1]) + (L.x[2] + L.x[3]);
        }
        if constexpr (NST > 4) {
            const float cv = g_is_d ? 1.0f : cvs[1];
            v4f_g h0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cv, ub0, z, 0, 0, 0);
            v4f_g h1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cv, ub1, z, 0, 0, 0);
            relu4(h0);
            relu4(h1);
#pragma unroll
            for (int r = 0; r < NST - 4; ++r) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(h0[r], L.x[4 + r], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(h1[r], L.x[4 + r], acc[1], 0, 0, 0);
            }
            bs = bs + ((L.x[4] + L.x[5]) + (L.x[6] + L.x[7]));
        }
    }
    // rows longer than 32 entries: one further block of 32 (entries e0 ..) of a row, plain loads, nothing prefetched
    __device__ __forceinline__ void extra_block(int w0, int len, int e0, int c, float ub0, float ub1, v4f_g (&acc)[2], float &bs) const
    {
        int J0 = 0, E0 = -1, J1 = 0, E1 = -1;
        if (e0 + n < len) { J0 = idx[w0 + e0 + n]; E0 = eidx[w0 + e0 + n]; }
        if (e0 + 16 + n < len) { J1 = idx[w0 + e0 + 16 + n]; E1 = eidx[w0 + e0 + 16 + n]; }
        to_offsets(J0, E0, true);
        to_offsets(J1, E1, true);
        GnoLoads Lx;
        float cvx[2];
        issue(Lx, J0, J1, g, c, true);   // every lane group holds the same 32 entries
        load_cv(cvx, E0, E1, g);
        compute<8>(Lx, cvx, ub0, ub1, acc, bs);
    }
};


// SAVE (training-mode forward, athena_mp_gno_aggregate_fwd_save): the consumers also copy every piece of S, as it lies in
// LDS, to `save` -- [tile][piece][vertex slot 32][512] + bias rows [tile][32][64] behind the pieces -- so that the reverse
// pass's S^T g needs no producers (gno_stg_kernel<true>).  33 GB at BASELINE configs[3]: HBM capacity bought back as time.
// Wave ot copies slots 8 ot .. 8 ot + 7 of the piece in the last two rounds (one contiguous KB per store); the MFMA
// rounds themselves are untouched, so `out` has the same bits as without the copy.
constexpr size_t kSavePiece = 32 * 512, kSaveTile = 8 * kSavePiece + 32 * 64;   // words
template <bool SAVE>
__global__ __launch_bounds__(kPcThreads) void gno_pc_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ idx,
                                                      const int32_t *__restrict__ eidx, const float *__restrict__ y,
                                                      const float *__restrict__ coords, const float *__restrict__ theta,
                                                      int d, const float *__restrict__ Vp, int n_rows,
                                                      const int32_t *__restrict__ perm, float *__restrict__ out,
                                                      uint32_t y_bytes, uint32_t c_bytes, uint32_t id_bytes,
                                                      float *__restrict__ save)
{
    extern __shared__ __attribute__((aligned(16))) float Sh[];
    float *Sbuf = Sh;                                   // [2][32][520]
    float *Bbuf = Sh + 2 * kPV * kPPitch;               // [2][32][72]   bias rows, by tile parity
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int n_tiles = (n_rows + kPV - 1) / kPV;
    const int nt = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles of this workgroup (>= 1)

    if (wave < 8) {
        // ======================================= producer =======================================
        __builtin_amdgcn_s_setprio(3);   // the producers' instructions first (worth 1 %)
        GnoProd P;
        P.init(wave, lane, rowptr, idx, eidx, y, coords, theta, d, n_rows, perm, y_bytes, c_bytes, id_bytes);
        const int p = P.p;
        const float(&Ub)[4] = P.Ub;

        GnoIds cur, nxt;
        P.ids_rows(blockIdx.x, cur);
        P.ids_ptrs(cur);
        P.ids_entries(cur);
        P.ids_finish(cur);
        nxt = cur;
        int pJ0 = cur.J0, pJ1 = cur.J1;                                  // the offsets the refills read: this tile's, from a tile's
                                                                     // last piece on the next tile's
        GnoLoads LS[4];   // one set per vertex: refilled for the NEXT piece right after this piece's MFMAs have read it,
                          // so a load has a whole piece interval to land
#pragma unroll
        for (int vi = 0; vi < 4; ++vi) P.issue(LS[vi], pJ0, pJ1, vi, 0, true);
        float CV[4][2];   // coordinates of the tile's slots; refilled for the next tile after the last piece has read them
        // h (first 16 entries) of a vertex kept over the four pieces of a kh: 8 registers per vertex.  Tiles whose rows have more
        // than 16 entries keep it for two of the wave's four vertices (the second operand block fills the register file);
        // the tiles after them -- a workgroup's tiles come in falling order of length -- for all four, in a loop of their own
        // in which the second block's registers are dead.
        v4f_g HCc[4][2];
#pragma unroll
        for (int vi = 0; vi < 4; ++vi) P.load_cv(CV[vi], cur.E0, cur.E1, vi);
        auto run_tiles = [&](auto SHORT_, int &ti) {
        constexpr bool SHORT = decltype(SHORT_)::value;
        constexpr int kHCache = SHORT ? 4 : 2;
        for (; ti < nt; ++ti) {
            const int tile = blockIdx.x + ti * gridDim.x;
            const bool more = ti + 1 < nt;
            const int maxlen = max(max(cur.len[0], cur.len[1]), max(cur.len[2], cur.len[3]));
            const int nstT = min(8, (maxlen + 3) >> 2);
            if (!SHORT && nstT <= 4) break;   // the rest of the workgroup's tiles: the loop that keeps h for all four vertices
            int nstN = nstT;   // the next tile's (known from its third piece on; its rows are not longer than this tile's)
#pragma unroll 1
            for (int pc = 0; pc < 8; ++pc) {
                const int c = pc & 3, kh = pc >> 2;   // the four feature quarters of a kh follow each other: h is kept
                float *buf = Sbuf + (size_t)((ti * 8 + pc) & 1) * kPV * kPPitch;
                float *bb = Bbuf + (size_t)(ti & 1) * kPV * kPBPitch;
                const float ub0 = kh ? Ub[2] : Ub[0], ub1 = kh ? Ub[3] : Ub[1];
                // the next tile's ids, one dependent step at a time
                if (more) {
                    if (pc == 0) P.ids_rows(tile + gridDim.x, nxt);
                    if (pc == 2) {
                        P.ids_ptrs(nxt);
                        nstN = min(8, (max(max(nxt.len[0], nxt.len[1]), max(nxt.len[2], nxt.len[3])) + 3) >> 2);
                    }
                    if (pc == 4) P.ids_entries(nxt);
                    if (pc == 6) P.ids_finish(nxt);
                }
                // the refill is unconditional: from a tile's last piece on it reads the next tile's rows (the last tile of
                // all re-reads its own: harmless, nothing consumes them)
                const int cn = (pc + 1) & 3;                              // feature quarter of the next piece
                const bool last = pc == 7;
                if (last) { pJ0 = nxt.J0; pJ1 = nxt.J1; }
                const bool second = !SHORT && (last ? nstN : nstT) > 4;   // does the piece being requested read entries 16 .. 31
                // (asking for them unconditionally in this loop -- no branch, no register copies around it -- measured 11.81
                // against 11.85 ms: not worth a second code path)
                // the four vertices of the wave with the step count of the longest of them as a compile-time constant
                // (tiles hold vertices of nearly equal length, so the shorter rows' extra steps -- on zeros -- are few)
                auto four = [&](auto K, auto FILL) {
#pragma unroll
                    for (int vi = 0; vi < ((GNO_FV & 32) ? 0 : 4); ++vi) {
                        v4f_g acc[2];
                        float bs;
                        if (vi < kHCache) P.compute<decltype(K)::value, decltype(FILL)::value ? 1 : 2>(LS[vi], CV[vi], ub0, ub1, acc, bs, HCc[vi]);
                        else P.compute<decltype(K)::value>(LS[vi], CV[vi], ub0, ub1, acc, bs);
                        // (GNO_FV & 8192, timing only: no gathers during pieces 3 .. 6 -- what the launch costs when a feature quarter
                        // is gathered once per tile instead of once per kh; the results are wrong)
                        if (!(GNO_FV & 8192) || pc < 3 || pc == 7) P.issue(LS[vi], pJ0, pJ1, vi, cn, second);
                        if (last) P.load_cv(CV[vi], nxt.E0, nxt.E1, vi);
                        const int v = 4 * p + vi;
                        float *srow = buf + (size_t)v * kPPitch;
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r2 = 0; r2 < 4; ++r2) srow[(16 * t + 4 * r2 + g) * 16 + n] = acc[t][r2];
                        if (kh == 0) {
                            bs = bs + __shfl_xor(bs, 16);
                            bs = bs + __shfl_xor(bs, 32);
                            if (g == 0) bb[v * kPBPitch + 16 * c + n] = bs;
                        }
                    }
                };
#define GNO_FOUR(FILL_)                                                             \
    if (SHORT) {                                                                   \
        switch (nstT) {                                                            \
        case 0:                                                                    \
        case 1: four(std::integral_constant<int, 1>{}, std::integral_constant<bool, FILL_>{}); break; \
        case 2: four(std::integral_constant<int, 2>{}, std::integral_constant<bool, FILL_>{}); break; \
        case 3: four(std::integral_constant<int, 3>{}, std::integral_constant<bool, FILL_>{}); break; \
        default: four(std::integral_constant<int, 4>{}, std::integral_constant<bool, FILL_>{}); break; \
        }                                                                          \
    } else {                                                                       \
        switch (nstT) {                                                            \
        case 5: four(std::integral_constant<int, 5>{}, std::integral_constant<bool, FILL_>{}); break; \
        case 6: four(std::integral_constant<int, 6>{}, std::integral_constant<bool, FILL_>{}); break; \
        case 7: four(std::integral_constant<int, 7>{}, std::integral_constant<bool, FILL_>{}); break; \
        default: four(std::integral_constant<int, 8>{}, std::integral_constant<bool, FILL_>{}); break; \
        }                                                                          \
    }
                if (c == 0) { GNO_FOUR(true) } else { GNO_FOUR(false) }
#undef GNO_FOUR
                // rows longer than 32 entries (none at BASELINE configs[3]): the remaining blocks are added to the vertex's
                // own LDS row; plain loads, nothing prefetched
                if (maxlen > 32) {
#pragma unroll 1
                    for (int vi = 0; vi < 4; ++vi) {
                        const int len = vi == 0 ? cur.len[0] : vi == 1 ? cur.len[1] : vi == 2 ? cur.len[2] : cur.len[3];
                        const int w0 = vi == 0 ? cur.w0[0] : vi == 1 ? cur.w0[1] : vi == 2 ? cur.w0[2] : cur.w0[3];
                        const int v = 4 * p + vi;
                        float *srow = buf + (size_t)v * kPPitch;
                        for (int e0 = 32; e0 < len; e0 += 32) {
                            v4f_g acc[2];
                            float bs;
                            P.extra_block(w0, len, e0, c, ub0, ub1, acc, bs);
#pragma unroll
                            for (int t = 0; t < 2; ++t)
#pragma unroll
                                for (int r2 = 0; r2 < 4; ++r2) srow[(16 * t + 4 * r2 + g) * 16 + n] += acc[t][r2];
                            if (kh == 0) {
                                bs = bs + __shfl_xor(bs, 16);
                                bs = bs + __shfl_xor(bs, 32);
                                if (g == 0) bb[v * kPBPitch + 16 * c + n] += bs;
                            }
                        }
                    }
                }
                if (last) cur = nxt;
                __syncthreads();
            }
        }
        };
        int ti = 0;
        run_tiles(std::integral_constant<bool, false>{}, ti);
        run_tiles(std::integral_constant<bool, true>{}, ti);
        __syncthreads();   // the consumers' last piece
    } else {
        // ======================================= consumer =======================================
        // wave = output tile ot (16 outputs) over the whole K of a piece, both groups of 16 vertices: one wave per SIMD
        // issues the contraction's MFMAs back to back (two independent accumulators), nothing to add across waves
        const int ot = wave - 8;
        const v4f_g z = {0.0f, 0.0f, 0.0f, 0.0f};
        v4f_g acc0 = z, acc1 = z;
        // this wave's stream of V: [pc][ot][gi][lane][4]; the bias piece behind the eight pieces
        const float *vw = Vp + ((size_t)ot * 32) * 256 + lane * 4;
        const float *vbias = Vp + 64 * 64 * 64 + ((size_t)ot * 4) * 256 + lane * 4;
        auto vload = [&](const float *p) { return *reinterpret_cast<const v4f_g *>(p); };
        constexpr int kPiece = 4 * 32 * 256;   // words of V per piece
        v4f_g a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = vload(vw + (size_t)u * 256);
        __syncthreads();
        for (int ti = 0; ti < nt; ++ti) {
            const int tile = blockIdx.x + ti * gridDim.x;
            const int sa = tile * kPV + n, sb = sa + 16;
            const int ra = perm[min(sa, n_rows - 1)], rb = perm[min(sb, n_rows - 1)];   // unconditional loads (see above)
            acc0 = acc1 = z;
#pragma unroll 1
            for (int pc = 0; pc < 8; ++pc) {
                const float *sb0 = Sbuf + (size_t)((ti * 8 + pc) & 1) * kPV * kPPitch + (size_t)n * kPPitch + 4 * g;
                const float *sb1 = sb0 + 16 * kPPitch;
                const float *vp = vw + (size_t)pc * kPiece;
                const float *vnext = pc < 7 ? vp + kPiece : vbias;   // the piece after this one
#pragma unroll
                for (int rd = 0; rd < ((GNO_FV & 64) ? 0 : 8); ++rd) {
                    v4f_g b0[4], b1[4], an[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        b0[u] = *reinterpret_cast<const v4f_g *>(sb0 + 16 * (4 * rd + u));
                        b1[u] = *reinterpret_cast<const v4f_g *>(sb1 + 16 * (4 * rd + u));
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        an[u] = vload(rd < 7 ? vp + (size_t)(4 * (rd + 1) + u) * 256 : vnext + (size_t)u * 256);
                    __builtin_amdgcn_sched_barrier(0);   // the scheduler otherwise sinks these loads to just before their use
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][s], b0[u][s], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][s], b1[u][s], acc1, 0, 0, 0);
                        }
#pragma unroll
                    for (int u = 0; u < 4; ++u) a[u] = an[u];
                    if constexpr (SAVE) {
                        if (rd >= 6) {   // this wave's quarter of the piece: slots 8 ot + 4 (rd - 6) .. + 3, 2 KB each
                            const float *src = Sbuf + (size_t)((ti * 8 + pc) & 1) * kPV * kPPitch + 4 * lane;
                            float *dst = save + (size_t)tile * kSaveTile + (size_t)pc * kSavePiece + 4 * lane;
                            // (a descriptor per tile: the 33 GB lie beyond what one descriptor addresses)
                            [[maybe_unused]] const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
                                (void *)(save + (size_t)tile * kSaveTile), 0, (int)(kSaveTile * 4), 0x00020000);
                            v4f_g cp[8];
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const int v = 8 * ot + 4 * (rd - 6) + (i >> 1);
                                cp[i] = *reinterpret_cast<const v4f_g *>(src + (size_t)v * kPPitch + 256 * (i & 1));
                            }
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const int v = 8 * ot + 4 * (rd - 6) + (i >> 1);
#if GNO_SAVE_MODE == 0
                                __builtin_nontemporal_store(cp[i], reinterpret_cast<v4f_g *>(dst + (size_t)v * 512 + 256 * (i & 1)));
#elif GNO_SAVE_MODE == 1
                                *reinterpret_cast<v4f_g *>(dst + (size_t)v * 512 + 256 * (i & 1)) = cp[i];
#elif GNO_SAVE_MODE == 3
                                (void)dst;
                                asm volatile("" ::"v"(cp[i]));
#else
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_g, cp[i]), srs,
                                    (int)(((size_t)pc * kSavePiece + (size_t)v * 512 + 256 * (i & 1) + 4 * lane) * 4), 0, GNO_SAVE_AUX);
#endif
                            }
                        }
                    }
                }
                if (pc == 7) {
                    // the bias piece: 64 words = four groups of 16; a[0..3] hold its V rows
                    const float *bb0 = Bbuf + (size_t)(ti & 1) * kPV * kPBPitch + (size_t)n * kPBPitch + 4 * g;
                    const float *bb1 = bb0 + 16 * kPBPitch;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const v4f_g b0 = *reinterpret_cast<const v4f_g *>(bb0 + 16 * u);
                        const v4f_g b1 = *reinterpret_cast<const v4f_g *>(bb1 + 16 * u);
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][s], b0[s], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][s], b1[s], acc1, 0, 0, 0);
                        }
                    }
                    if constexpr (SAVE) {   // the tile's bias rows: slots 8 ot .. + 7, 256 B each
                        const float *src = Bbuf + (size_t)(ti & 1) * kPV * kPBPitch + 4 * n;
                        float *dst = save + (size_t)tile * kSaveTile + 8 * kSavePiece + 4 * n;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const int v = 8 * ot + 4 * i + g;
                            const v4f_g t = *reinterpret_cast<const v4f_g *>(src + (size_t)v * kPBPitch);
                            __builtin_nontemporal_store(t, reinterpret_cast<v4f_g *>(dst + (size_t)v * 64));
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) a[u] = vload(vw + (size_t)u * 256);   // first round of the next tile
                    // lane (vertex n of its group, g): acc[r] = out[vertex][16 ot + 4 g + r]
                    if (sa < n_rows) *reinterpret_cast<v4f_g *>(out + (size_t)ra * kGF + 16 * ot + 4 * g) = acc0;
                    if (sb < n_rows) *reinterpret_cast<v4f_g *>(out + (size_t)rb * kGF + 16 * ot + 4 * g) = acc1;
                }
                __syncthreads();
            }
        }
    }
}

// ---- dVaug = S^T g with S never in HBM (H = 64, widths 64, d <= 3) ------------------------------------------------
// The same producers, the contraction turned round: out[kq][o] = sum_v S[v][kq] g[v][o] contracts over the VERTICES, so
// its 4160 x 64 accumulators must stay put while the vertices stream by.  A workgroup therefore owns ONE piece (c, kh) of
// S for its whole life -- 512 x 64 sums = 128 registers per lane of its four consumer waves -- and every 32nd tile:
// workgroup b: piece b % 8, tiles b / 8, b / 8 + nsub, ... (nsub = 32 tile classes on a full-size graph)  Per tile its producers build just that piece (the gathers of a
// tile are shared out over the eight workgroups that visit it: each reads its own 64-byte quarter of the feature rows), and
// put the tile's 32 gradient rows beside it in LDS.  The bias row (sum of x_j) meets g on the producers' own MFMAs
// (K = the four vertices of a wave).  256 partial slabs (33 MB) are summed in a fixed order by gno_stg_reduce_kernel,
// which also undoes the producers' row order.  Replaces: outer product -> 33 GB of S through HBM -> contraction.
constexpr int kGPitch = 72;
constexpr int kStgLdsFloats = 2 * kPV * kPPitch + 2 * kPV * kGPitch;
constexpr int kStgGrid = 256;   // 8 pieces x 32 tile classes

// SAVED: S was kept by the forward pass (gno_pc_kernel<true>); the producers only copy the workgroup's piece of each tile
// (64 KB, one contiguous 2 KB row per slot) and the tile's gradient rows into LDS one tile ahead -- no ids, no gathers, no
// MFMAs of their own except the bias row's.  The consumers are the same code, so dV has the same bits either way.
template <bool SAVED>
__global__ __launch_bounds__(kPcThreads) void gno_stg_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ idx,
                                                       const int32_t *__restrict__ eidx, const float *__restrict__ y,
                                                       const float *__restrict__ coords, const float *__restrict__ theta,
                                                       int d, const float *__restrict__ grad, int n_rows,
                                                       const int32_t *__restrict__ perm, float *__restrict__ slab,
                                                       float *__restrict__ slabB, uint32_t y_bytes, uint32_t c_bytes,
                                                       uint32_t id_bytes, uint32_t g_bytes, int nsub, int grouped,
                                                       const float *__restrict__ save)
{
    extern __shared__ __attribute__((aligned(16))) float Sh[];
    float *Sbuf = Sh;                                   // [2][32][520]
    float *Gbuf = Sh + 2 * kPV * kPPitch;               // [2][32][72]   gradient rows of the tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    // Where the eight workgroups (pieces) that visit the same tiles sit.  Workgroup b runs on XCD b % 8.  Spread order (round 2:
    // piece = b % 8): the eight are on eight different XCDs, every 64-byte quarter row costs each of them its own 128-byte line
    // from the fabric -- 57 GB per launch at configs[3] (2 x FETCH_SIZE; L2 hit rate 2 %) against 8.7 GB algorithmic.  Grouped
    // order (round 3, the product path whenever the tile classes divide by 8): piece = b / nsub, so the eight workgroups of a
    // tile class share ONE XCD's L2 -- 19.5 GB, L2 hit rate 66 % (profiles/r03_c4_gno_pmc_traffic.txt; 10.9 GB / 81 % in round 2's
    // A/B build: the eight drift apart over a launch, by how much varies).  The kernel is bound by its matrix and vector work,
    // not by either figure; the grouped order leaves the fabric to whatever
    // runs beside it.  (Round 2's spread mapping was an A/B switch until round 5; docs/history/DESIGN_r01-r04.md.)
    const int pc = grouped ? blockIdx.x / nsub : blockIdx.x & 7, sub = grouped ? blockIdx.x % nsub : blockIdx.x >> 3;
    const int c = pc >> 1, kh = pc & 1;
    const int n_tiles = (n_rows + kPV - 1) / kPV;
    const int nt = sub < n_tiles ? (n_tiles - sub + nsub - 1) / nsub : 0;   // tiles of this workgroup: sub, sub + nsub, ...

    if (wave < 8) {
        // ======================================= producer =======================================
        __builtin_amdgcn_s_setprio(3);
        GnoProd P;
        P.init(wave, lane, rowptr, idx, eidx, y, coords, theta, d, n_rows, perm, y_bytes, c_bytes, id_bytes);
        const int p = P.p;
        const float ub0 = kh ? P.Ub[2] : P.Ub[0], ub1 = kh ? P.Ub[3] : P.Ub[1];
        __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)grad, 0, (int)g_bytes, 0x00020000);
        // gradient rows of the wave's four vertices as the bias MFMA's A operand: lane (o = 16 ot + n, vertex g)
        auto load_g = [&](float (&GV)[4], const GnoIds &I) {
            const int row = g == 0 ? I.row[0] : g == 1 ? I.row[1] : g == 2 ? I.row[2] : I.row[3];
            const bool ok = g == 0 ? I.ok[0] : g == 1 ? I.ok[1] : g == 2 ? I.ok[2] : I.ok[3];
            const uint32_t off = ok ? (uint32_t)row * (4u * kGF) + P.n4 : GnoProd::kDead;
#pragma unroll
            for (int ot = 0; ot < 4; ++ot)
                GV[ot] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, (int)off, 64 * ot, 0));
        };
        const v4f_g z = {0.0f, 0.0f, 0.0f, 0.0f};
        v4f_g accB[4] = {z, z, z, z};   // bias rows: [o = 16 ot + 4 g + r][q = 16 c + n], kh = 0 workgroups only
        if constexpr (SAVED) {
            const int pcF = 4 * kh + c;   // the piece's number in the forward kernel's order
            GnoIds T0, T1, T2;            // row numbers only (for the gradient rows), two tiles ahead
            P.ids_rows(sub, T0);
            P.ids_rows(sub + nsub, T1);
            T2 = T1;
            // (one tile ahead is enough: the same copy two tiles ahead, in two register sets, measured 21.4 ms per dtheta against
            // 20.8 -- three A/B pairs on one box)
            v4f_g R[8];
            float bsel, GV[4];
            auto load_tile = [&](int tl) {   // beyond the last tile: the last tile again (never consumed)
                const float *base = save + (size_t)min(tl, n_tiles - 1) * kSaveTile;
                const float *src = base + (size_t)pcF * kSavePiece + 4 * lane;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    R[i] = __builtin_nontemporal_load(reinterpret_cast<const v4f_g *>(src + (size_t)(4 * p + (i >> 1)) * 512 + 256 * (i & 1)));
                bsel = base[8 * kSavePiece + (size_t)(4 * p + g) * 64 + 16 * c + n];
            };
            load_tile(sub);
            load_g(GV, T0);
#pragma unroll 1
            for (int j = 0; j < nt; ++j) {
                P.ids_rows(sub + nsub * (j + 2), T2);
                float *buf = Sbuf + (size_t)(j & 1) * kPV * kPPitch + 4 * lane;
                float *gb = Gbuf + (size_t)(j & 1) * kPV * kGPitch;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    *reinterpret_cast<v4f_g *>(buf + (size_t)(4 * p + (i >> 1)) * kPPitch + 256 * (i & 1)) = R[i];
#pragma unroll
                for (int ot = 0; ot < 4; ++ot) gb[(4 * p + g) * kGPitch + 16 * ot + n] = GV[ot];
                if (kh == 0) {
#pragma unroll
                    for (int ot = 0; ot < 4; ++ot) accB[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(GV[ot], bsel, accB[ot], 0, 0, 0);
                }
                load_tile(sub + nsub * (j + 1));
                load_g(GV, T1);
                __syncthreads();
                T0 = T1; T1 = T2;
            }
        } else {
        // the ids of the workgroup's tiles: a four-deep queue, one dependent step per tile interval
        //   T4 row numbers (issued now) | T3 row pointers | T2 entries | T1 offsets: the tile whose operands are requested
        //   during this interval | T0 the tile being built
        GnoIds T0, T1, T2, T3, T4;
        P.ids_rows(sub, T0); P.ids_ptrs(T0); P.ids_entries(T0); P.ids_finish(T0);
        P.ids_rows(sub + nsub, T1); P.ids_ptrs(T1); P.ids_entries(T1);
        P.ids_rows(sub + 2 * nsub, T2); P.ids_ptrs(T2);
        P.ids_rows(sub + 3 * nsub, T3);
        T4 = T3;
        GnoLoads LS[4];
        float CV[4][2], GV[4];
#pragma unroll
        for (int vi = 0; vi < 4; ++vi) {
            P.issue(LS[vi], T0.J0, T0.J1, vi, c, true);
            P.load_cv(CV[vi], T0.E0, T0.E1, vi);
        }
        load_g(GV, T0);
#pragma unroll 1
        for (int j = 0; j < nt; ++j) {
            P.ids_rows(sub + nsub * (j + 4), T4);
            P.ids_ptrs(T3);
            P.ids_entries(T2);
            P.ids_finish(T1);
            const int maxlen = max(max(T0.len[0], T0.len[1]), max(T0.len[2], T0.len[3]));
            const int nstT = min(8, (maxlen + 3) >> 2);
            const bool second = max(max(T1.len[0], T1.len[1]), max(T1.len[2], T1.len[3])) > 16;
            float *buf = Sbuf + (size_t)(j & 1) * kPV * kPPitch;
            float *gb = Gbuf + (size_t)(j & 1) * kPV * kGPitch;
            float bsv[4];
            auto four = [&](auto K) {
#pragma unroll
                for (int vi = 0; vi < 4; ++vi) {
                    v4f_g acc[2];
                    float bs;
                    P.compute<decltype(K)::value>(LS[vi], CV[vi], ub0, ub1, acc, bs);
                    P.issue(LS[vi], T1.J0, T1.J1, vi, c, second);
                    P.load_cv(CV[vi], T1.E0, T1.E1, vi);
                    float *srow = buf + (size_t)(4 * p + vi) * kPPitch;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r2 = 0; r2 < 4; ++r2) srow[(16 * t + 4 * r2 + g) * 16 + n] = acc[t][r2];
                    bsv[vi] = bs;
                }
            };
            switch (nstT) {
            case 0:
            case 1: four(std::integral_constant<int, 1>{}); break;
            case 2: four(std::integral_constant<int, 2>{}); break;
            case 3: four(std::integral_constant<int, 3>{}); break;
            case 4: four(std::integral_constant<int, 4>{}); break;
            case 5: four(std::integral_constant<int, 5>{}); break;
            case 6: four(std::integral_constant<int, 6>{}); break;
            case 7: four(std::integral_constant<int, 7>{}); break;
            default: four(std::integral_constant<int, 8>{}); break;
            }
            // column sums of the feature quarter (the bias row of S): every lane (n, any g) ends with the total.  A row's first
            // 32 entries and each further block are reduced across the lane groups BEFORE they are added up -- the order of
            // gno_pc_kernel, so that the S it keeps and the S built here give the same bits
            if (kh == 0) {
#pragma unroll
                for (int vi = 0; vi < 4; ++vi) {
                    bsv[vi] = bsv[vi] + __shfl_xor(bsv[vi], 16);
                    bsv[vi] = bsv[vi] + __shfl_xor(bsv[vi], 32);
                }
            }
            if (maxlen > 32) {   // rows longer than 32 entries: the remaining blocks are added to the vertex's own LDS row
#pragma unroll 1
                for (int vi = 0; vi < 4; ++vi) {
                    const int len = vi == 0 ? T0.len[0] : vi == 1 ? T0.len[1] : vi == 2 ? T0.len[2] : T0.len[3];
                    const int w0 = vi == 0 ? T0.w0[0] : vi == 1 ? T0.w0[1] : vi == 2 ? T0.w0[2] : T0.w0[3];
                    float *srow = buf + (size_t)(4 * p + vi) * kPPitch;
                    for (int e0 = 32; e0 < len; e0 += 32) {
                        v4f_g acc[2];
                        float bs;
                        P.extra_block(w0, len, e0, c, ub0, ub1, acc, bs);
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r2 = 0; r2 < 4; ++r2) srow[(16 * t + 4 * r2 + g) * 16 + n] += acc[t][r2];
                        bs = bs + __shfl_xor(bs, 16);
                        bs = bs + __shfl_xor(bs, 32);
                        if (vi == 0) bsv[0] += bs; else if (vi == 1) bsv[1] += bs; else if (vi == 2) bsv[2] += bs; else bsv[3] += bs;
                    }
                }
            }
            // the tile's gradient rows: beside S in LDS for the consumers, and against the bias sums here
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) gb[(4 * p + g) * kGPitch + 16 * ot + n] = GV[ot];
            if (kh == 0) {
                const float bsel = g == 0 ? bsv[0] : g == 1 ? bsv[1] : g == 2 ? bsv[2] : bsv[3];
#pragma unroll
                for (int ot = 0; ot < 4; ++ot) accB[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(GV[ot], bsel, accB[ot], 0, 0, 0);
            }
            load_g(GV, T1);
            __syncthreads();
            T0 = T1; T1 = T2; T2 = T3; T3 = T4;
        }
        }
        __syncthreads();   // the consumers' last tile
        if (kh == 0) {     // bias rows: the eight waves' parts meet in LDS (S is done with), waves 0-3 add them up
            float *sc = Sh + (size_t)p * 1024;
#pragma unroll
            for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[(16 * ot + 4 * g + r) * 16 + n] = accB[ot][r];
        }
        __syncthreads();
        if (kh == 0 && p < 4) {
            const int t0 = p * 256 + lane * 4;
            v4f_g sum = *reinterpret_cast<const v4f_g *>(Sh + t0);
#pragma unroll
            for (int w = 1; w < 8; ++w) {
                const v4f_g v = *reinterpret_cast<const v4f_g *>(Sh + (size_t)w * 1024 + t0);
#pragma unroll
                for (int r = 0; r < 4; ++r) sum[r] = sum[r] + v[r];
            }
            *reinterpret_cast<v4f_g *>(slabB + (size_t)(sub * 8 + pc) * 1024 + t0) = sum;
        }
    } else {
        // ======================================= consumer =======================================
        // wave w: positions L = 128 w .. + 127 of the piece x all 64 outputs = 32 tiles of 16 x 16.  Per four vertices:
        // A = S[v][L0 + 4 n .. + 3] (one 16-byte LDS read feeds four tiles, L = L0 + 4 m + j), B = g[v][4 n .. + 3].
        const int w = wave - 8;
        const v4f_g z = {0.0f, 0.0f, 0.0f, 0.0f};
        v4f_g acc[2][4][4];
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[sp][a][b] = z;
        __syncthreads();
#pragma unroll 1
        for (int j = 0; j < nt; ++j) {
            const float *sb = Sbuf + (size_t)(j & 1) * kPV * kPPitch + (size_t)g * kPPitch + 128 * w + 4 * n;
            const float *gb = Gbuf + (size_t)(j & 1) * kPV * kGPitch + (size_t)g * kGPitch + 4 * n;
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {
                const v4f_g a0 = *reinterpret_cast<const v4f_g *>(sb + (size_t)(4 * s4) * kPPitch);
                const v4f_g a1 = *reinterpret_cast<const v4f_g *>(sb + (size_t)(4 * s4) * kPPitch + 64);
                const v4f_g bv = *reinterpret_cast<const v4f_g *>(gb + (size_t)(4 * s4) * kGPitch);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        acc[0][a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[a], bv[b], acc[0][a][b], 0, 0, 0);
                        acc[1][a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[a], bv[b], acc[1][a][b], 0, 0, 0);
                    }
            }
            __syncthreads();
        }
        // lane (o = 4 n + b, g): acc[sp][a][b][r] = out[L = 128 w + 64 sp + 4 (4 g + r) + a][o]
        float *sl = slab + (size_t)(sub * 8 + pc) * 512 * kGF;
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int L = 128 * w + 64 * sp + 4 * (4 * g + r) + a;
                    const v4f_g v = {acc[sp][a][0][r], acc[sp][a][1][r], acc[sp][a][2][r], acc[sp][a][3][r]};
                    *reinterpret_cast<v4f_g *>(sl + (size_t)L * kGF + 4 * n) = v;
                }
        __syncthreads();   // the producers' bias rows
    }
}

// dVaug[kq][o] = the nsub slabs of kq's piece, in workgroup order; kq = k * 64 + q sits at position L of piece (q / 16, k / 32)
__global__ void gno_stg_reduce_kernel(const float *__restrict__ slab, const float *__restrict__ slabB, float *__restrict__ dV, int nsub)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 65 * 64 * 64) return;
    const int kq = t >> 6, o = t & 63;
    float sum = 0.0f;
    if (kq < 64 * 64) {
        const int k = kq >> 6, q = kq & 63, c = q >> 4, q16 = q & 15, kh = k >> 5, kl = k & 31;
        const int tp = kl >> 4, gg = (kl >> 2) & 3, r = kl & 3;
        const int L = (16 * tp + 4 * r + gg) * 16 + q16, pc = 2 * c + kh;
        for (int sub = 0; sub < nsub; ++sub) sum = sum + slab[((size_t)(sub * 8 + pc) * 512 + L) * kGF + o];
    } else {
        const int q = kq - 64 * 64, c = q >> 4, q16 = q & 15;
        for (int sub = 0; sub < nsub; ++sub) sum = sum + slabB[((size_t)(sub * 8 + 2 * c) * 64 + o) * 16 + q16];
    }
    dV[t] = sum;
}

bool gno_stg_shape(int H, int Fi, int Fo, int d)
{
    return H == kGH && Fi == kGF && Fo == kGF && d <= 3;
}

int launch_gno_stg(const athena_mp_graph *g, const float *x, const float *coords, const float *theta, int d, const float *grad,
                   float *dV, const float *save = nullptr)
{
    const size_t y_bytes = sizeof(float) * kGF * (size_t)g->n_cols, c_bytes = sizeof(float) * (size_t)d * g->n_edge_cols,
                 id_bytes = sizeof(int32_t) * (size_t)g->nnz, g_bytes = sizeof(float) * kGF * (size_t)g->n_rows;
    const size_t lim = 0xFFFFE000ull;
    if (!(y_bytes < lim && c_bytes < lim && id_bytes < lim && g_bytes < lim)) return -1;   // caller takes the other route
    if (length_order(g->rowptr, g->n_rows, &g->len_perm_fwd, &g->n_long_fwd, &g->n_mid_fwd)) return 1;
    constexpr size_t lds = sizeof(float) * (size_t)kStgLdsFloats;
    static amp::PerDeviceFlag attr;
    if (!attr.get()) {
        AMP_HIP(hipFuncSetAttribute((const void *)gno_stg_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        AMP_HIP(hipFuncSetAttribute((const void *)gno_stg_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.get() = true;
    }
    void *slab = nullptr, *slabB = nullptr;
    if (amp::workspace(&slab, sizeof(float) * (size_t)kStgGrid * 512 * kGF, 0) ||
        amp::workspace(&slabB, sizeof(float) * (size_t)kStgGrid * 1024, 8))
        return 1;
    // 8 pieces x nsub tile classes: 32 classes fill the chip's 256 CUs; a small graph takes one class per tile
    const int nsub = std::max(1, std::min(kStgGrid / 8, (g->n_rows + kPV - 1) / kPV));
    const int grouped = nsub % 8 == 0 ? 1 : 0;   // the eight pieces of a tile class on one XCD (one L2): profiles/r03_c4_stg_order_ab.txt
    if (save)
        hipLaunchKernelGGL(gno_stg_kernel<true>, dim3(8 * nsub), dim3(kPcThreads), lds, amp::stream(), g->rowptr, g->col, g->eid, x,
                           coords, theta, d, grad, g->n_rows, (const int32_t *)g->len_perm_fwd, (float *)slab, (float *)slabB,
                           (uint32_t)y_bytes, (uint32_t)c_bytes, (uint32_t)id_bytes, (uint32_t)g_bytes, nsub, grouped, save);
    else
        hipLaunchKernelGGL(gno_stg_kernel<false>, dim3(8 * nsub), dim3(kPcThreads), lds, amp::stream(), g->rowptr, g->col, g->eid, x,
                           coords, theta, d, grad, g->n_rows, (const int32_t *)g->len_perm_fwd, (float *)slab, (float *)slabB,
                           (uint32_t)y_bytes, (uint32_t)c_bytes, (uint32_t)id_bytes, (uint32_t)g_bytes, nsub, grouped,
                           (const float *)nullptr);
    AMP_LAUNCH_CHECK();
    hipLaunchKernelGGL(gno_stg_reduce_kernel, dim3((65 * 64 * 64 + 255) / 256), dim3(256), 0, amp::stream(), (const float *)slab,
                       (const float *)slabB, dV, nsub);
    AMP_LAUNCH_CHECK();
    return 0;
}

int launch_gno_pc(const int32_t *rowptr, const int32_t *idx, const int32_t *eidx, const float *y, const float *coords,
                  const float *theta, int d, const float *Vaug, int n_rows, const int32_t *perm, float *out, size_t y_bytes,
                  size_t c_bytes, size_t id_bytes, float *save)
{
    constexpr size_t lds = sizeof(float) * (size_t)kPcLdsFloats;
    static amp::PerDeviceFlag attr;
    if (!attr.get()) {
        AMP_HIP(hipFuncSetAttribute((const void *)gno_pc_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        AMP_HIP(hipFuncSetAttribute((const void *)gno_pc_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr.get() = true;
    }
    if (n_rows <= 0) return 0;
    void *vp = nullptr;
    if (amp::workspace(&vp, sizeof(float) * kVpFloats, 9)) return 1;
    hipLaunchKernelGGL(gno_vrelay_kernel, dim3((65 * 64 * 64 + 255) / 256), dim3(256), 0, amp::stream(), Vaug, (float *)vp);
    AMP_LAUNCH_CHECK();
    const int n_tiles = (n_rows + kPV - 1) / kPV;
    if (save)
        hipLaunchKernelGGL(gno_pc_kernel<true>, dim3(std::min(n_tiles, amp::num_cus())), dim3(kPcThreads), lds, amp::stream(), rowptr,
                           idx, eidx, y, coords, theta, d, (const float *)vp, n_rows, perm, out, (uint32_t)y_bytes,
                           (uint32_t)c_bytes, (uint32_t)id_bytes, save);
    else
        hipLaunchKernelGGL(gno_pc_kernel<false>, dim3(std::min(n_tiles, amp::num_cus())), dim3(kPcThreads), lds, amp::stream(), rowptr,
                           idx, eidx, y, coords, theta, d, (const float *)vp, n_rows, perm, out, (uint32_t)y_bytes,
                           (uint32_t)c_bytes, (uint32_t)id_bytes, (float *)nullptr);
    AMP_LAUNCH_CHECK();
    return 0;
}

// B2[(k*Fo + o)*Fi + q] = Vaug[F*k + o + Fo*q]
__global__ void gno_perm_kernel(const float *__restrict__ vaug, int H1, int Fi, int Fo, float *__restrict__ B2)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int n = H1 * Fo * Fi;
    if (t >= n) return;
    int q = t % Fi, ko = t / Fi, o = ko % Fo, k = ko / Fo;
    B2[t] = vaug[(size_t)Fo * Fi * k + o + Fo * q];
}

// Per CSR entry: dh[k] = sum_q G_i[k,q] x_j[q]; gh = relu'(pre_e[k]) dh.  Accumulates dU / db_u partials
// per workgroup (chunk of rows) and optionally stores gh per entry for the coordinate gradient.
__global__ __launch_bounds__(256) void gno_dh_kernel(const int32_t *__restrict__ rowptr,
                                                     const int32_t *__restrict__ col,
                                                     const int32_t *__restrict__ eid,
                                                     const float *__restrict__ x, int Fi,
                                                     const float *__restrict__ coords,
                                                     const float *__restrict__ theta, int d, int H,
                                                     const float *__restrict__ G, int r0, int n_rows_tile,
                                                     int rows_per_block, float *__restrict__ slabs,
                                                     float *__restrict__ ghbuf)
{
    extern __shared__ float sm[];
    const int GP = Fi + 1;                 // padded pitch: column reads of G by consecutive k
    float *Gs = sm;                        // [H][GP]
    float *xs = Gs + H * GP;               // [kEB][Fi]
    float *ghs = xs + kEB * Fi;            // [kEB][H]
    float *part = ghs + kEB * H;           // [H*d + H] this block's dU | db_u partial
    const float *U = theta, *bu = theta + (size_t)H * d;
    const int np = H * d + H;
    for (int t = threadIdx.x; t < np; t += 256) part[t] = 0.0f;
    const int lr0 = blockIdx.x * rows_per_block;
    const int lr1 = min(n_rows_tile, lr0 + rows_per_block);
    for (int lr = lr0; lr < lr1; ++lr) {
        const int row = r0 + lr;
        __syncthreads();
        for (int t = threadIdx.x; t < H * Fi; t += 256) {
            int k = t / Fi, q = t - k * Fi;
            Gs[k * GP + q] = G[(size_t)lr * H * Fi + t];
        }
        const int w0 = rowptr[row], w1 = rowptr[row + 1];
        for (int wb = w0; wb < w1; wb += kEB) {
            const int nb = min(kEB, w1 - wb);
            __syncthreads();
            for (int t = threadIdx.x; t < nb * Fi; t += 256) {
                int b = t / Fi, q = t - b * Fi;
                xs[b * Fi + q] = x[(size_t)col[wb + b] * Fi + q];
            }
            __syncthreads();
            for (int t = threadIdx.x; t < nb * H; t += 256) {
                int b = t / H, k = t - b * H;
                int e = eid[wb + b];
                float gh = 0.0f;
                if (e >= 0) {
                    const float *dx = coords + (size_t)e * d;
                    float s = 0.0f;
                    for (int j = 0; j < d; ++j) s = s + U[k + (size_t)H * j] * dx[j];
                    s = s + bu[k];
                    if (s > 0.0f) {
                        float dh = 0.0f;
                        for (int q = 0; q < Fi; ++q) dh = fmaf(Gs[k * GP + q], xs[b * Fi + q], dh);
                        gh = dh;
                    }
                }
                ghs[b * H + k] = gh;
                if (ghbuf) ghbuf[(size_t)(wb + b) * H + k] = gh;
            }
            __syncthreads();
            // dU[k + H*j] += gh[k] dx_e[j];  db_u[k] += gh[k]   (entries of the batch in order)
            for (int t = threadIdx.x; t < np; t += 256) {
                float s = part[t];
                if (t < H * d) {
                    int j = t / H, k = t - j * H;
                    for (int b = 0; b < nb; ++b) {
                        int e = eid[wb + b];
                        if (e >= 0) s = fmaf(ghs[b * H + k], coords[(size_t)e * d + j], s);
                    }
                } else {
                    int k = t - H * d;
                    for (int b = 0; b < nb; ++b) s = s + ghs[b * H + k];
                }
                part[t] = s;
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < np; t += 256) slabs[(size_t)blockIdx.x * np + t] = part[t];
}

// ---- MFMA form of the kernel-MLP backward (H, F_in multiples of 32, <= 64; d <= 3) -----------------
// One wave per vertex i, entries in tiles of 32:
//   DH^T[e][k] = sum_q x_{j(e)}[q] G_i[k][q]                 (A = gathered neighbour rows, B = G_i^T)
//   GH = relu'(U dx_e + b_u) . DH^T                          (mask applied in the C layout)
//   dU^T[j][k] += sum_e dx_e[j] GH[e][k],  db_u[k] += sum_e GH[e][k]
//        -> second MFMA that takes the GH accumulator tile directly as its B operand (register r of
//           lane half h is row e = (r&3) + 8(r>>2) + 4h), A[j][e] = dx_e[j] for j < d and 1 for j = d.
// dU^T / db_u stay in accumulators across all vertices of the wave: one slab per wave, no atomics.
typedef float v4f_g __attribute__((ext_vector_type(4)));

template <int HT, int QT>
__global__ __launch_bounds__(256) void gno_dh_mfma_kernel(const int32_t *__restrict__ rowptr,
                                                          const int32_t *__restrict__ col,
                                                          const int32_t *__restrict__ eid,
                                                          const float *__restrict__ x,
                                                          const float *__restrict__ coords,
                                                          const float *__restrict__ theta, int d,
                                                          const float *__restrict__ G, int r0, int n_rows_tile,
                                                          float *__restrict__ slabs, float *__restrict__ ghbuf)
{
    constexpr int H = 32 * HT, Fi = 32 * QT, HF = Fi / 2;   // lane half h owns q in [HF*h, HF*h + HF)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r31 = lane & 31, h = lane >> 5;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    float Uk[HT][3], bk[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) {
        bk[t] = theta[(size_t)H * d + 32 * t + r31];
#pragma unroll
        for (int j = 0; j < 3; ++j) Uk[t][j] = j < d ? theta[(32 * t + r31) + (size_t)H * j] : 0.0f;
    }
    f32x16 du[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) du[t][r] = 0.0f;

    for (int lr = gw; lr < n_rows_tile; lr += nw) {
        const int row = r0 + lr;
        const int w0 = rowptr[row], w1 = rowptr[row + 1];
        if (w0 == w1) continue;
        // B operand: G_i[k = 32t + r31][HF*h + s], s = 0..HF-1
        float gf[HT][HF];
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int m = 0; m < HF / 4; ++m) {
                const v4f_g v = *reinterpret_cast<const v4f_g *>(G + ((size_t)lr * H + 32 * t + r31) * Fi + HF * h + 4 * m);
                gf[t][4 * m] = v.x; gf[t][4 * m + 1] = v.y; gf[t][4 * m + 2] = v.z; gf[t][4 * m + 3] = v.w;
            }
        for (int wb = w0; wb < w1; wb += 32) {
            const int nb = min(32, w1 - wb);
            const int my_j = r31 < nb ? col[wb + r31] : -1;
            const int my_e = r31 < nb ? eid[wb + r31] : -1;
            // A operand: x_{j(e = r31)}[HF*h + s]
            float xf[HF];
#pragma unroll
            for (int m = 0; m < HF / 4; ++m) {
                v4f_g v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (my_j >= 0 && my_e >= 0) v = *reinterpret_cast<const v4f_g *>(x + (size_t)my_j * Fi + HF * h + 4 * m);
                xf[4 * m] = v.x; xf[4 * m + 1] = v.y; xf[4 * m + 2] = v.z; xf[4 * m + 3] = v.w;
            }
            f32x16 dh[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) dh[t][r] = 0.0f;
#pragma unroll
                for (int s2 = 0; s2 < HF; ++s2) dh[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xf[s2], gf[t][s2], dh[t], 0, 0, 0);
            }
            // mask + the A operand of the second product, register by register
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int e_loc = (r & 3) + 8 * (r >> 2) + 4 * h;          // entry of this register's row
                const int ee = __shfl(my_e, e_loc);                        // lanes 0..31 hold the tile's entries
                const bool ok = e_loc < nb && ee >= 0;
                float dx[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (ok && j < d) dx[j] = coords[(size_t)ee * d + j];
                float amat = 0.0f;                                         // A[j = r31][e] : dx_e[j] | 1 | 0
                if (ok) amat = r31 < d ? (r31 == 0 ? dx[0] : (r31 == 1 ? dx[1] : dx[2])) : (r31 == d ? 1.0f : 0.0f);
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    float pre = 0.0f;
#pragma unroll
                    for (int j = 0; j < 3; ++j) pre = pre + Uk[t][j] * dx[j];
                    pre = pre + bk[t];
                    const float gh = (ok && pre > 0.0f) ? dh[t][r] : 0.0f;
                    if (ghbuf && e_loc < nb) ghbuf[(size_t)(wb + e_loc) * H + 32 * t + r31] = gh;
                    du[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(amat, gh, du[t], 0, 0, 0);
                }
            }
        }
    }
    // slab[j*H + k]: rows j = 0..d of the accumulator (registers 0..3 of lane half 0)
    float *slab = slabs + (size_t)gw * (H * d + H);
    if (h == 0) {
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r <= d) slab[(size_t)r * H + 32 * t + r31] = du[t][r];
    }
}

// dcoords[e, j] = sum_{entries w carrying e} sum_k U[k + H*j] gh[w, k]
__global__ void gno_dcoords_kernel(const int32_t *__restrict__ e_rowptr, const int32_t *__restrict__ e_ent,
                                   const float *__restrict__ ghbuf, const float *__restrict__ theta, int d, int H,
                                   int E, float *__restrict__ dcoords)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * d) return;
    int e = t / d, j = t - e * d;
    float s = 0.0f;
    for (int p = e_rowptr[e]; p < e_rowptr[e + 1]; ++p) {
        const float *gh = ghbuf + (size_t)e_ent[p] * H;
        for (int k = 0; k < H; ++k) s = fmaf(theta[k + (size_t)H * j], gh[k], s);
    }
    dcoords[t] = s;
}

int launch_outer(const int32_t *rowptr, const int32_t *idx, const int32_t *eidx, const float *y, int Fy,
                 const float *coords, const float *theta, int d, int H, int r0, int rows, float *S)
{
    const int R = (H + 1) * Fy;
    if (H % 32 == 0 && Fy % 32 == 0 && H <= 64 && Fy <= 64 && d <= 4) {
        dim3 g4((rows + 3) / 4), b4(256);
#define AMP_OM(A_, B_)                                                                                   \
    if (H == 32 * A_ && Fy == 32 * B_)                                                                   \
        hipLaunchKernelGGL((gno_outer_mfma_kernel<A_, B_>), g4, b4, 0, amp::stream(), rowptr, idx, eidx, y, coords, \
                           theta, d, r0, rows, S);
        AMP_OM(1, 1) AMP_OM(1, 2) AMP_OM(2, 1) AMP_OM(2, 2)
#undef AMP_OM
        AMP_LAUNCH_CHECK();
        return 0;
    }
    const size_t lds = sizeof(float) * kEB * (size_t)(H + 1 + Fy);
    dim3 grid(rows), block(256);
#define AMP_OUT(E_)                                                                                      \
    hipLaunchKernelGGL((gno_outer_kernel<E_>), grid, block, lds, amp::stream(), rowptr, idx, eidx, y, Fy, \
                       coords, theta, d, H, r0, S)
    if (R <= 256) AMP_OUT(1);
    else if (R <= 1024) AMP_OUT(4);
    else if (R <= 5120) AMP_OUT(20);
    else if (R <= 17408) AMP_OUT(68);
    else {
        amp::set_error("gno: (H+1)*F = %d exceeds the supported 17408", R);
        return 2;
    }
#undef AMP_OUT
    AMP_LAUNCH_CHECK();
    return 0;
}

// ---- fused kernel-MLP backward for H = 64, F_in = F_out = 64 (BASELINE configs[3]) -------------------------
// dh_e = x_{j(e)} . G_i with G_i[k][q] = sum_o g[i,o] V[o + Fo q + F k] -- G (16 KB per vertex, 32 GB at C4) never
// reaches HBM.  Same shape as gno_fused_kernel: a persistent 16-wave workgroup owns 16 vertices and walks the hidden
// index in two halves so that the 16 x (32 x 64) half of G fits LDS (140 KB):
//   phase A  G_half^T[(kl,q), v] = sum_o Vp[o][k][q] g[v,o] on 16x16x4 MFMAs with the VERTEX on the column axis:
//            B = g[v, 4s + gq] (16 registers per lane for the whole tile), A = Vp[4s + gq][32 half + kl][4m .. 4m+3]
//            (Vp = V re-laid [o][k][q] once per call) -- one 16 B load from L2 feeds four MFMAs whose output tiles
//            interleave q = 4m + c, so a lane ends up with 16 consecutive q of its vertex and parks them in LDS with
//            16 B stores; wave w produces hidden units kl = 2w, 2w + 1 of the half;
//   phase B  wave w owns vertex w exactly like gno_dh_mfma_kernel: DH^T[e][kl] = sum_q x_{j(e)}[q] G[kl][q] on
//            32x32x2 MFMAs (B operand read from LDS as it is needed), relu' mask in the C layout, then the
//            second MFMA that accumulates dU^T / db_u over all entries of all of the wave's vertices.
// One slab of (H d + H) partial sums per wave, reduced in fixed order by slab_reduce (no atomics).
#ifndef GDH_VARIANT
#define GDH_VARIANT 0   // 1: no phase A, 2: no phase B, 3: phase A without its LDS stores -- timing-only builds, never shipped
#endif
constexpr int kDRow = 68;                       // LDS pitch of one (vertex, kl) row of 64 q
constexpr int kDVtx = 32 * kDRow + 4;           // LDS pitch of one vertex' half

__global__ void gno_vperm_okq_kernel(const float *__restrict__ V, float *__restrict__ Vp)
{
    // Vp[(o*64 + k)*64 + q] = V[o + 64 q + 4096 k]
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 64 * 64 * 64) return;
    const int q = t & 63, k = (t >> 6) & 63, o = t >> 12;
    Vp[t] = V[o + 64 * q + 4096 * k];
}

template <bool WRITE_GH>
__global__ __launch_bounds__(1024) void gno_gdh_kernel(const int32_t *__restrict__ rowptr,
                                                       const int32_t *__restrict__ col,
                                                       const int32_t *__restrict__ eid,
                                                       const float *__restrict__ x,
                                                       const float *__restrict__ coords,
                                                       const float *__restrict__ theta, int d,
                                                       const float *__restrict__ Vp,
                                                       const float *__restrict__ grad, int n_rows,
                                                       const int32_t *__restrict__ perm,
                                                       float *__restrict__ slabs, float *__restrict__ ghbuf)
{
    extern __shared__ __attribute__((aligned(16))) float Gs[];   // [16 vertices][32 kl][kDRow] (+4 per vertex)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r31 = lane & 31, h = lane >> 5;
    const int m = lane & 15, gq = lane >> 4;
    const int n_tiles = (n_rows + kGRows - 1) / kGRows;
    // dU^T[j][k] (j < d) and db_u[k] (j = 3) for this lane's hidden unit k = 32 half + r31, summed over the entry
    // rows this lane half sees; the two lane halves are combined at the end
    float du[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) du[t][jj] = 0.0f;
    float *cbuf = Gs + (size_t)kGRows * kDVtx + wave * 128;   // per wave: coordinates of the 32 entries in flight

    // the next tile's row pointers and first 32 (neighbour, edge) ids are fetched while the current tile is on the
    // matrix cores, so phase B starts with its gathers instead of two dependent index loads
    int nw0 = 0, nw1 = 0, nj = -1, ne = -1;
    auto fetch_ids = [&](int tl) {
        const int slot = tl * kGRows + wave;
        nw0 = nw1 = 0;
        nj = ne = -1;
        if (tl < n_tiles && slot < n_rows) {
            const int row = perm[slot];
            nw0 = rowptr[row];
            nw1 = rowptr[row + 1];
            if (r31 < nw1 - nw0) { nj = col[nw0 + r31]; ne = eid[nw0 + r31]; }
        }
    };
    fetch_ids(blockIdx.x);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int r0 = tile * kGRows;
        const int vrow = (r0 + m) < n_rows ? perm[r0 + m] : -1;   // phase-A vertex of this lane
        const int w0 = nw0, w1 = nw1, cj = nj, ce = ne;           // phase-B row of this wave
        for (int half = 0; half < 2; ++half) {
            // ---------------- phase A: hidden units kl = 2 wave, 2 wave + 1 of this half ----------------
            // B operand: g[vertex m of the tile][o = 4s + gq], s = 0..15 (re-read per half: L1-hot, and it keeps
            // 16 registers free during phase B)
            float gb[16];
            {
                const float *gr = grad + (size_t)max(vrow, 0) * 64 + gq;     // clamped address, value selected after
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const float v = gr[4 * s];
                    gb[s] = vrow >= 0 ? v : 0.0f;
                }
            }
#pragma unroll 1
            for (int b = 0; b < (GDH_VARIANT == 1 ? 0 : 2); ++b) {
                const int kl = 2 * wave + b;
                v4f_g om[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) om[c] = v4f_g{0.0f, 0.0f, 0.0f, 0.0f};
                const float *vb = Vp + ((size_t)gq * 64 + 32 * half + kl) * 64 + 4 * m;   // + s * (4 * 64 * 64)
#pragma unroll
                for (int s0 = 0; s0 < 16; s0 += 4) {
                    v4f_g a[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const v4f_g *>(vb + (size_t)(s0 + u) * (4 * 64 * 64));
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            om[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][c], gb[s0 + u], om[c], 0, 0, 0);
                }
                // lane (vertex m, gq): om[c][r] = G[vertex][kl][q = 16 gq + 4 r + c]
                float *dst = Gs + (size_t)m * kDVtx + kl * kDRow + 16 * gq;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (GDH_VARIANT != 3 || om[0][r] == 12345.0f)
                        *reinterpret_cast<v4f_g *>(dst + 4 * r) = v4f_g{om[0][r], om[1][r], om[2][r], om[3][r]};
            }
            if (half == 1) fetch_ids(tile + gridDim.x);   // flies under the last phase B and the next phase A
            __syncthreads();
            // ---------------- phase B: entries of vertex `wave`, hidden units 32 half + r31 ----------------
            {
                float Uk[3], bk;
                {
                    const int k = 32 * half + r31;
                    bk = theta[(size_t)kGH * d + k];
#pragma unroll
                    for (int j = 0; j < 3; ++j) Uk[j] = j < d ? theta[k + (size_t)kGH * j] : 0.0f;
                }
                const float *grow = Gs + (size_t)wave * kDVtx + r31 * kDRow + 32 * h;   // G[k = r31][q = 32 h + s]
                for (int wb = w0; wb < (GDH_VARIANT == 2 ? w0 : w1); wb += 32) {
                    const int nb = min(32, w1 - wb);
                    int my_j = cj, my_e = ce;
                    if (wb != w0) {   // rows longer than 32 entries: later blocks are fetched here
                        my_j = r31 < nb ? col[wb + r31] : -1;
                        my_e = r31 < nb ? eid[wb + r31] : -1;
                    }
                    // the entry's coordinate difference goes through LDS: one gather per entry (lane half 0) next
                    // to the neighbour-row gathers, instead of a dependent load per accumulator row afterwards
                    if (h == 0) {
                        v4f_g cv = {0.0f, 0.0f, 0.0f, 0.0f};
                        if (my_e >= 0) {
                            cv.x = coords[(size_t)my_e * d];
                            if (d > 1) cv.y = coords[(size_t)my_e * d + 1];
                            if (d > 2) cv.z = coords[(size_t)my_e * d + 2];
                        }
                        *reinterpret_cast<v4f_g *>(cbuf + 4 * r31) = cv;
                    }
                    f32x16 dh;
#pragma unroll
                    for (int r = 0; r < 16; ++r) dh[r] = 0.0f;
                    {
                        // all eight 16 B pieces of the neighbour row at once (clamped address, value selected after)
                        const bool live = my_j >= 0 && my_e >= 0;
                        const float *xr = x + (size_t)max(my_j, 0) * 64 + 32 * h;
                        v4f_g xv[8];
#pragma unroll
                        for (int mm = 0; mm < 8; ++mm) xv[mm] = *reinterpret_cast<const v4f_g *>(xr + 4 * mm);
#pragma unroll
                        for (int mm = 0; mm < 8; ++mm) {
                            const v4f_g gv = *reinterpret_cast<const v4f_g *>(grow + 4 * mm);
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                dh = __builtin_amdgcn_mfma_f32_32x32x2f32(live ? xv[mm][c] : 0.0f, gv[c], dh, 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    float *ghp = nullptr;
                    if constexpr (WRITE_GH) ghp = ghbuf + (size_t)(wb + 4 * h) * kGH + 32 * half + r31;
                    // relu' mask in the C layout (register r of lane half h is entry (r&3) + 8(r>>2) + 4h), then the
                    // rank-1 updates of dU^T / db_u on the VALU
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int e_loc = (r & 3) + 8 * (r >> 2) + 4 * h;
                        const int ee = __shfl(my_e, e_loc);
                        const bool ok = e_loc < nb && ee >= 0;
                        const v4f_g cv = *reinterpret_cast<const v4f_g *>(cbuf + 4 * e_loc);
                        float pre = 0.0f;
                        pre = pre + Uk[0] * cv.x;
                        pre = pre + Uk[1] * cv.y;
                        pre = pre + Uk[2] * cv.z;
                        pre = pre + bk;
                        const float gh = (ok && pre > 0.0f) ? dh[r] : 0.0f;
                        if constexpr (WRITE_GH) {
                            if (e_loc < nb) ghp[((r & 3) + 8 * (r >> 2)) * kGH] = gh;
                        }
                        if (half == 0) {
                            du[0][0] = du[0][0] + cv.x * gh; du[0][1] = du[0][1] + cv.y * gh;
                            du[0][2] = du[0][2] + cv.z * gh; du[0][3] = du[0][3] + gh;
                        } else {
                            du[1][0] = du[1][0] + cv.x * gh; du[1][1] = du[1][1] + cv.y * gh;
                            du[1][2] = du[1][2] + cv.z * gh; du[1][3] = du[1][3] + gh;
                        }
                        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // four rows of LDS reads in flight, not sixteen
                    }
                    __builtin_amdgcn_wave_barrier();   // cbuf is rewritten by the next block of entries
                }
            }
            __syncthreads();   // the next half / tile overwrites Gs
        }
    }
    // slab[j*H + k], j = 0..d-1: dU^T; slab[d*H + k]: db_u -- lane halves combined, lane half 0 writes
    float *slab = slabs + ((size_t)blockIdx.x * 16 + wave) * (kGH * d + kGH);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const float tot = du[t][jj] + __shfl_xor(du[t][jj], 32);
            if (h == 0) {
                if (jj < 3 && jj < d) slab[(size_t)jj * kGH + 32 * t + r31] = tot;
                if (jj == 3) slab[(size_t)d * kGH + 32 * t + r31] = tot;
            }
        }
}

bool gno_gdh_shape(int H, int Fi, int Fo, int d)
{
    return H == kGH && Fi == kGF && Fo == kGF && d <= 3;
}

int tile_rows_for(int64_t n_rows, int64_t floats_per_row)
{
    constexpr int64_t budget = (int64_t)1024 << 20;   // bytes of S / T / G per super-tile: 1 GiB, two slots in flight (bwd_theta pipeline);
                                                      // 43.8 ms at C4 against 44.4 at 2 GiB
    int64_t t = budget / (4 * std::max<int64_t>(floats_per_row, 1));
    t = std::max<int64_t>(1, std::min<int64_t>(t, n_rows));
    return (int)t;
}

bool gno_args_ok(const athena_mp_graph *g, int d, int H, int Fi, int Fo)
{
    if (!g || d <= 0 || H <= 0 || Fi <= 0 || Fo <= 0) {
        amp::set_error("gno: bad arguments (d=%d H=%d F_in=%d F_out=%d)", d, H, Fi, Fo);
        return false;
    }
    if ((size_t)kEB * (H + 1 + std::max(Fi, Fo)) * 4 > 60000 || ((size_t)H * (Fi + 1) + kEB * (Fi + H) + H * d + H) * 4 > 150000) {
        amp::set_error("gno: H=%d, F=%d exceed the LDS staging of this build", H, std::max(Fi, Fo));
        return false;
    }
    return true;
}

// shared backward of the kernel MLP: dU, db_u (want_theta) and/or dcoords (want_coords)
// ---- kernel-MLP backward, dense and sparse halves side by side (H = 64, widths 64, d <= 3, rows of <= 32 entries) ---
// dh[e][k] = sum_q G_i[k][q] x_j[q] with G_i = g_i . Vmat^T (4096 values per vertex) is where dU, db_u and dcoords come
// from.  gno_gdh_kernel builds G half by half and then walks the entries, every wave in the same phase (22.6 ms at C4:
// 15.9 + 8.2 alone).  Here, as in gno_pc_kernel with the roles swapped, G is cut in eight pieces per 32-vertex tile --
// piece (kh, c) = hidden units 32 kh .. +31 x features 16 c .. +15 -- double-buffered in LDS, and
//   waves 8-11  (one per SIMD) are DENSE: G[v][kq] = sum_o g[v][o] V[kq][o] on 16x16x4 MFMAs, V streamed from L2 in
//               the order gno_vrelay_dense_kernel lays down (one 16-byte load per eight MFMAs), the tile's gradient
//               rows in registers; a finished 16 x 16 block is four hidden units x four features per lane group, so
//               it goes to LDS as one 16-byte store per vertex;
//   waves 0-7   are SPARSE, four vertices each: dh^T[e][k] += x_j[e][16 c ..] . G[..][k] on MFMAs (A = one 16-byte load
//               of the neighbour's feature quarter, B = one 16-byte LDS read of G), accumulated in registers over the
//               four c of a kh; then the relu' mask from the h MFMA of the forward pass (same layout), the entry's
//               masked dh to HBM when dcoords wants it, and [dU | db_u] += dh^T . [dx_e ; 1] on one more MFMA whose A
//               operand IS the masked accumulator (register r of lane group g = entry 4 g + r).
// The few rows longer than 32 entries (they head the length-ordered vertex list) stay with gno_gdh_kernel.
constexpr int kDhLdsFloats = 2 * kPV * kPPitch;
constexpr int kDhCStrip = 2 * 4 * kGF;   // PX: per sparse wave, two tiles' worth of its vertices' b_v^T g rows (4 x 64 floats each)

// Vd[pc'][w][mt][sg][lane][i] = Vmat[kq][o]: pc' = 4 kh + c; block (w, mt) = hidden units 4 (2 w + mt / 4) .. +3 x
// features 4 (mt % 4) .. +3 of the piece; lane (m, ok): row m = (hid m / 4, feature m % 4), o = 16 sg + 4 ok + i
__global__ void gno_vrelay_dense_kernel(const float *__restrict__ Vin, float *__restrict__ Vd)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 64 * 64 * 64) return;
    const int i = t & 3, lane = (t >> 2) & 63, sg = (t >> 8) & 3, mt = (t >> 10) & 7, w = (t >> 13) & 3, pcp = t >> 15;
    const int m = lane & 15, ok = lane >> 4, kh = pcp >> 2, c = pcp & 3;
    const int hidl = 4 * (2 * w + (mt >> 2)) + (m >> 2), q = 16 * c + 4 * (mt & 3) + (m & 3);
    const int o = 16 * sg + 4 * ok + i;
    Vd[t] = Vin[(size_t)((32 * kh + hidl) * 64 + q) * 64 + o];
}

// position of G[hid][4 chunk ..] inside a vertex's row of a piece: chunks of the hidden units 8 .. 15 (mod 16) swapped
// pairwise, so that the sparse waves' 16-byte reads (lane = hidden unit, lane group = chunk) fall on 16 different banks
__device__ __forceinline__ int gno_gpos(int hidl, int chunk) { return hidl * 16 + ((chunk ^ ((hidl >> 2) & 2)) << 2); }

// VPW vertices of a tile per sparse wave (tile = 8 VPW vertices), NB blocks of 16 entries per row: <4, 1> for rows of at most
// 16 entries, <2, 2> for rows of 17 .. 32 -- the register file holds 4 x 1 or 2 x 2 sets of dh accumulators, not 4 x 2
// PX (athena_mp_gno_aggregate_bwd: dx AND dtheta from ONE G = g . Vmat^T): while a piece of G_i lies in LDS the sparse waves
// also take the feature gradient's per-entry partial from it,
//     px[w][16 c + q] = (b_v^T g_i)[16 c + q] + sum_k h_e[k] G_i[k][16 c + q]   (the kh = 0 pieces store the first 32 hidden
//                       units' share, the kh = 1 pieces read it back -- two vertices ahead, from the L2 / Infinity Cache -- as
//                       the start value of their accumulators and store the finished partial over it),
// i.e. entry w = (i -> j, e)'s contribution to dx_j = sum K_e^T g_i (athena_diffstruc_extd_sub_nop.f90:419-458) -- eight
// more 16x16x4 MFMAs per 16 entries and piece (K = hidden units; A = G^T read from LDS one word per lane, B = h^T from an
// h MFMA with its operands swapped, whose result registers ARE the B operand: register r of lane (entry n, g) is hidden
// unit pi(4 g + r), pi chosen so that the A reads of a half wave fall on 32 different banks).  The partials go to HBM
// ([nnz][64], stored through one buffer descriptor per vertex: lanes beyond the row's length fall outside it and are
// dropped by the bounds check, so the store is unconditional) and gno_px_gather_kernel sums them over the transposed CSR:
// the second 1.07 TFLOP contraction of the reverse pass (T . B2 in the dx launch) is gone.
template <bool WRITE_GH, int VPW, int NB, bool PX = false>
__global__ __launch_bounds__(kPcThreads) void gno_dh_pc_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ idx,
                                                         const int32_t *__restrict__ eidx, const float *__restrict__ y,
                                                         const float *__restrict__ coords, const float *__restrict__ theta,
                                                         int d, const float *__restrict__ Vd, const float *__restrict__ grad,
                                                         int n_rows, const int32_t *__restrict__ perm, float *__restrict__ slabs,
                                                         float *__restrict__ ghbuf, uint32_t y_bytes, uint32_t c_bytes,
                                                         uint32_t id_bytes, uint32_t g_bytes, float *__restrict__ px = nullptr,
                                                         size_t px_half = 0, const float *__restrict__ cvec = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) float Sh[];
    constexpr int TV = 8 * VPW, NG = VPW / 2;          // vertices per tile, groups of 16 of them
    float *Gbuf = Sh;                                   // [2][TV][520]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int n_tiles = (n_rows + TV - 1) / TV;
    const int nt = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles of this workgroup (>= 1)

    if (wave >= 8) {
        // ======================================= dense =======================================
        const int w = wave - 8;
        __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)grad, 0, (int)g_bytes, 0x00020000);
        // the tile's gradient rows as B operands: lane (vertex n [+ 16], ok = g) holds g[v][16 j + 4 ok .. + 3], j = 0 .. 3
        auto load_gt = [&](v4f_g (&G0)[4], v4f_g (&G1)[4], int tile) {
            const int sa = tile * TV + n, sb = sa + 16;
            const bool oka = tile < n_tiles && sa < n_rows, okb = NG > 1 && tile < n_tiles && sb < n_rows;
            const int ra = perm[oka ? sa : 0], rb = perm[okb ? sb : 0];
            const uint32_t oa = oka ? (uint32_t)ra * (4u * kGF) + 16u * g : GnoProd::kDead;
            const uint32_t ob = okb ? (uint32_t)rb * (4u * kGF) + 16u * g : GnoProd::kDead;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                G0[j] = __builtin_bit_cast(v4f_g, __builtin_amdgcn_raw_buffer_load_b128(grs, (int)oa, 64 * j, 0));
                G1[j] = __builtin_bit_cast(v4f_g, __builtin_amdgcn_raw_buffer_load_b128(grs, (int)ob, 64 * j, 0));
            }
        };
        v4f_g G0[4], G1[4], G0n[4], G1n[4];
        load_gt(G0, G1, blockIdx.x);
        const float *vw = Vd + ((size_t)w * 8) * 1024 + lane * 4;      // + pc' * 32768 + mt * 1024 + sg * 256
        auto vload = [&](const float *p) { return *reinterpret_cast<const v4f_g *>(p); };
        v4f_g a[4];
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) a[sg] = vload(vw + (size_t)sg * 256);
        const v4f_g z = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int ti = 0; ti < nt; ++ti) {
            const int tile = blockIdx.x + ti * gridDim.x;
#pragma unroll 1
            for (int pcp = 0; pcp < 8; ++pcp) {
                float *buf = Gbuf + (size_t)((ti * 8 + pcp) & 1) * TV * kPPitch;
                const float *vp = vw + (size_t)pcp * 32768;
                const float *vnext = vw + (size_t)((pcp + 1) & 7) * 32768;
                if (pcp == 0) load_gt(G0n, G1n, tile + gridDim.x);   // the next tile's rows: seven pieces to land
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) {
                    v4f_g an[4];
#pragma unroll
                    for (int sg = 0; sg < 4; ++sg) an[sg] = vload(mt < 7 ? vp + (size_t)(mt + 1) * 1024 + (size_t)sg * 256 : vnext + (size_t)sg * 256);
                    __builtin_amdgcn_sched_barrier(0);
                    v4f_g c0 = z, c1 = z;
#pragma unroll
                    for (int sg = 0; sg < 4; ++sg)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[sg][i], G0[sg][i], c0, 0, 0, 0);
                            if (NG > 1) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[sg][i], G1[sg][i], c1, 0, 0, 0);
                        }
                    // lane (vertex n, g): c[r] = G[v][hid = 4 (2 w + mt / 4) + g][feature 4 (mt % 4) + r]
                    const int pos = gno_gpos(4 * (2 * w + (mt >> 2)) + g, mt & 3);
                    *reinterpret_cast<v4f_g *>(buf + (size_t)n * kPPitch + pos) = c0;
                    if (NG > 1) *reinterpret_cast<v4f_g *>(buf + (size_t)(n + 16) * kPPitch + pos) = c1;
#pragma unroll
                    for (int sg = 0; sg < 4; ++sg) a[sg] = an[sg];
                }
                if (pcp == 7) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { G0[j] = G0n[j]; G1[j] = G1n[j]; }
                }
                __syncthreads();
            }
        }
        __syncthreads();   // the sparse waves' last piece
    } else {
        // ======================================= sparse =======================================
        __builtin_amdgcn_s_setprio(3);
        GnoProd P;
        P.init(wave, lane, rowptr, idx, eidx, y, coords, theta, d, n_rows, perm, y_bytes, c_bytes, id_bytes, VPW);
        const int p = P.p;
        GnoIds cur, nxt;
        P.ids_rows(blockIdx.x, cur); P.ids_ptrs(cur); P.ids_entries(cur); P.ids_finish(cur);
        nxt = cur;
        // per vertex and block of 16 entries, for the whole tile: the byte offset of entry n's feature chunk g, and the
        // h MFMA's A operand (coordinate g of entry n's edge; 1 at g = d)
        uint32_t xoff[VPW][NB];
        float cvv[VPW][NB];
        auto derive = [&](int vi, const GnoIds &I) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                xoff[vi][b] = (uint32_t)__shfl(b ? I.J1 : I.J0, 16 * vi + n) + 16u * g;
                const uint32_t e = (uint32_t)__shfl(b ? I.E1 : I.E0, 16 * vi + n) + P.g4;
                cvv[vi][b] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(P.crs, (int)e, 0, 0));
            }
        };
        v4f_g XL[VPW][NB];   // the neighbours' feature chunks of the piece being consumed next
        auto xload = [&](int vi, int c) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
                XL[vi][b] = __builtin_bit_cast(v4f_g, __builtin_amdgcn_raw_buffer_load_b128(P.yrs, (int)xoff[vi][b], 64 * c, 0));
        };
#pragma unroll
        for (int vi = 0; vi < VPW; ++vi) {
            derive(vi, cur);
            xload(vi, 0);
        }
        const v4f_g z = {0.0f, 0.0f, 0.0f, 0.0f};
        // PX: pi(m) = (m & ~3) | ((m & 3) ^ ((m >> 2) & 1)) -- lane (n, g) of the swapped h MFMA carries [U ; b_u] of hidden
        // unit pi(n); its result register r is then hidden unit pi(4 g + r), whose parity alternates with g: the four
        // G rows a half wave reads per step lie in both halves of the 32 banks
        const int pi_src = 16 * g + ((n & ~3) | ((n & 3) ^ ((n >> 2) & 1)));
        int gq_off[4];            // word offset of G[hid pi(4 g + r)][feature n] inside a vertex's row of the piece
#pragma unroll
        for (int r = 0; r < 4; ++r) gq_off[r] = gno_gpos(4 * g + (r ^ (g & 1)), n >> 2) + (n & 3);
        v4f_g dacc[VPW][NB][2];   // [vertex][block][16 hidden units]: dh^T[entry 4 g + r][hid n], summed over the four c of a kh
        v4f_g accU[4] = {z, z, z, z};   // [16 hidden units]: lane (coordinate n, g): [dU | db_u][hid 4 g + r][n]
#pragma unroll
        for (int vi = 0; vi < VPW; ++vi)
#pragma unroll
            for (int b = 0; b < NB; ++b) dacc[vi][b][0] = dacc[vi][b][1] = z;
        // PX: c = b_v^T g of the wave's vertices (cvec[row][64], the rows of grad's own order): lane (vertex lane / 16, chunk
        // lane % 16) takes 16 bytes of its vertex's row -- one load per lane per tile, fetched a tile ahead (the row ids are
        // known by then), parked in a private LDS strip (no load sits in front of the partial MFMAs: a load there would make
        // their wait drain every prefetch).  A slot without a vertex reads some row; nothing of it is ever stored.
        float *cstrip = Sh + kDhLdsFloats + p * kDhCStrip;
        __amdgpu_buffer_rsrc_t cvrs = __builtin_amdgcn_make_buffer_rsrc((void *)cvec, 0, PX ? (int)g_bytes : 0, 0x00020000);
        auto c_off = [&](const GnoIds &I) { return (uint32_t)P.by_group(I.row) * (4u * kGF) + 16u * (uint32_t)n; };
        v4f_g cnext = z;
        // the kh = 0 partials of the kAhead vertices whose kh = 1 pieces come next (a ring: VPW is a multiple of kAhead)
#define GNO_PX_AHEAD 2
        constexpr int kAhead = GNO_PX_AHEAD < VPW ? GNO_PX_AHEAD : VPW;
        static_assert(VPW % kAhead == 0, "the ring of read-back partials");
        v4f_g pprev[kAhead][NB];
#pragma unroll
        for (int a = 0; a < kAhead; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) pprev[a][b] = z;
        if constexpr (PX) {
            const v4f_g c0 = __builtin_bit_cast(v4f_g, __builtin_amdgcn_raw_buffer_load_b128(cvrs, (int)c_off(cur), 0, 0));
            *reinterpret_cast<v4f_g *>(cstrip + 4 * lane) = c0;
        }
        __syncthreads();
        for (int ti = 0; ti < nt; ++ti) {
            const int tile = blockIdx.x + ti * gridDim.x;
            const bool more = ti + 1 < nt;
#pragma unroll 1
            for (int pcp = 0; pcp < 8; ++pcp) {
                const int kh = pcp >> 2, c = pcp & 3;
                const float *buf = Gbuf + (size_t)((ti * 8 + pcp) & 1) * TV * kPPitch;
                if (more) {
                    if (pcp == 0) P.ids_rows(tile + gridDim.x, nxt);
                    if (pcp == 2) P.ids_ptrs(nxt);
                    if (pcp == 4) P.ids_entries(nxt);
                    if (pcp == 6) P.ids_finish(nxt);
                }
                const bool last = pcp == 7;
                const float ub0 = kh ? P.Ub[2] : P.Ub[0], ub1 = kh ? P.Ub[3] : P.Ub[1];
                float ubp0 = 0.0f, ubp1 = 0.0f;
                if constexpr (PX) {
                    ubp0 = __shfl(ub0, pi_src);
                    ubp1 = __shfl(ub1, pi_src);
                    if (pcp == 4)     // the next tile's c rows (its row ids arrived with piece 0)
                        cnext = __builtin_bit_cast(v4f_g, __builtin_amdgcn_raw_buffer_load_b128(cvrs, (int)(more ? c_off(nxt) : GnoProd::kDead), 0, 0));
                }
                const float *cs = cstrip + (ti & 1) * (4 * kGF);
                {
#pragma unroll
                    for (int vi = 0; vi < VPW; ++vi) {
                        // G of the vertex: lane (hid n [+ 16], chunk g)
                        const float *grow = buf + (size_t)(VPW * p + vi) * kPPitch;
                        const v4f_g b0 = *reinterpret_cast<const v4f_g *>(grow + gno_gpos(n, g));
                        const v4f_g b1 = *reinterpret_cast<const v4f_g *>(grow + gno_gpos(16 + n, g));
#pragma unroll
                        for (int b = 0; b < NB; ++b)
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4) {
                                dacc[vi][b][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(XL[vi][b][s4], b0[s4], dacc[vi][b][0], 0, 0, 0);
                                dacc[vi][b][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(XL[vi][b][s4], b1[s4], dacc[vi][b][1], 0, 0, 0);
                            }
                        if (c == 3) {
                            // the kh's 32 hidden units are complete for this vertex: mask, store, fold into dU / db_u
                            const int w0 = cur.w0[vi], len = cur.len[vi];
#pragma unroll
                            for (int b = 0; b < NB; ++b) {
                                const float cv = P.g_is_d ? 1.0f : cvv[vi][b];
                                const v4f_g h0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cv, ub0, z, 0, 0, 0);
                                const v4f_g h1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cv, ub1, z, 0, 0, 0);
                                // [dx_e ; 1] as the B operand of the dU MFMA: lane (coordinate n, g), step r = entry 4 g + r
                                float dxT[4];
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const uint32_t eo = (uint32_t)__shfl(b ? cur.E1 : cur.E0, 16 * vi + 4 * g + r) + P.n4;
                                    dxT[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(P.crs, (int)(n < d ? eo : GnoProd::kDead), 0, 0));
                                }
                                v4f_g m0, m1;
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    m0[r] = h0[r] > 0.0f ? dacc[vi][b][0][r] : 0.0f;
                                    m1[r] = h1[r] > 0.0f ? dacc[vi][b][1][r] : 0.0f;
                                    if (WRITE_GH) {
                                        const int e = 16 * b + 4 * g + r;
                                        if (e < len) {
                                            float *gp = ghbuf + (size_t)(w0 + e) * kGH + 32 * kh + n;
                                            gp[0] = m0[r];
                                            gp[16] = m1[r];
                                        }
                                    }
                                }
                                if (kh == 0) {   // (static register names: a run-time index would put accU in scratch)
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        const float bx = n == d ? 1.0f : dxT[r];
                                        accU[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(m0[r], bx, accU[0], 0, 0, 0);
                                        accU[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(m1[r], bx, accU[1], 0, 0, 0);
                                    }
                                } else {
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        const float bx = n == d ? 1.0f : dxT[r];
                                        accU[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(m0[r], bx, accU[2], 0, 0, 0);
                                        accU[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(m1[r], bx, accU[3], 0, 0, 0);
                                    }
                                }
                                dacc[vi][b][0] = dacc[vi][b][1] = z;
                            }
                        }
                        if constexpr (PX) {
                            // G^T of the vertex as the A operand: lane (feature n, g), step (t, r) = hidden unit 16 t + pi(4 g + r)
                            float gq[2][4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                gq[0][r] = grow[gq_off[r]];
                                gq[1][r] = grow[gq_off[r] + 256];
                            }
                            // (b_v^T g_i)[16 c + 4 g ..] rides in the kh = 0 partial as the accumulator's start value; the kh = 1
                            // pieces start from the kh = 0 partial itself, read back a vertex ahead (pprev), so that ONE array
                            // [nnz][64] holds the finished partial
                            const v4f_g cq = *reinterpret_cast<const v4f_g *>(cs + vi * kGF + 16 * c + 4 * g);
                            // one descriptor per vertex: its len rows of px -- a lane beyond the row's length is out of range
                            __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
                                (void *)(px + (size_t)cur.w0[vi] * kGF), 0, cur.len[vi] * (4 * kGF), 0x00020000);
#pragma unroll
                            for (int b = 0; b < NB; ++b) {
                                const float cv = P.g_is_d ? 1.0f : cvv[vi][b];
                                v4f_g hT0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ubp0, cv, z, 0, 0, 0);
                                v4f_g hT1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ubp1, cv, z, 0, 0, 0);
                                GnoProd::relu4(hT0);
                                GnoProd::relu4(hT1);
                                v4f_g acc = kh ? pprev[vi % kAhead][b] : cq;
#pragma unroll
                                for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(gq[0][r], hT0[r], acc, 0, 0, 0);
#pragma unroll
                                for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(gq[1][r], hT1[r], acc, 0, 0, 0);
                                // lane (entry n, g): acc[r] = partial of entry 16 b + n, feature 16 c + 4 g + r
                                // (the piece's 64 c bytes ride in the VECTOR offset, the scalar offset stays an immediate 0: with a
                                // register there the compiler's hazard recogniser assumes the store's data registers may be rewritten
                                // at once -- on gfx950 a VALU write straight behind the store then replaced the first dword of the
                                // last 16 lanes' data: wrong partials for entries 12 .. 15 of some rows, found with the oracle)
                                // (plain stores: the kh = 0 partial is read back four pieces on.  The finished partial alone
                                // nontemporal -- a scalar branch on kh around two stores -- measured 0.4 ms SLOWER per reverse pass)
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_g, acc), prs,
                                                                       (16 * b + n) * (4 * kGF) + 16 * g + 64 * c, 0, GNO_PX_AUX);
                            }
                            // the kh = 0 partial of the vertex that comes TWO vertices on (in this piece or the next), if that one
                            // is a kh = 1 piece: an UNCONDITIONAL load -- a dead offset otherwise -- issued in front of this
                            // vertex's feature prefetch, so that the wait for it leaves those in flight.  Two vertices of matrix
                            // work cover its trip to the L2 / Infinity Cache, where the partial written four pieces ago still is.
                            {
                                const int vn = (vi + kAhead) % VPW;
                                const int pn = pcp + (vi + kAhead) / VPW;           // the piece that vertex belongs to
                                const bool want = pn >= 4 && pn < 8;
                                __amdgpu_buffer_rsrc_t nrs = __builtin_amdgcn_make_buffer_rsrc(
                                    (void *)(px + (size_t)cur.w0[vn] * kGF), 0, cur.len[vn] * (4 * kGF), 0x00020000);
#pragma unroll
                                for (int b = 0; b < NB; ++b) {
                                    const uint32_t o = (uint32_t)((16 * b + n) * (4 * kGF) + 16 * g + 64 * (pn & 3));
                                    pprev[vi % kAhead][b] = __builtin_bit_cast(v4f_g, __builtin_amdgcn_raw_buffer_load_b128(nrs, (int)(want ? o : GnoProd::kDead), 0, GNO_PX_RB_AUX));
                                }
                            }
                        }
                        // the next piece's feature chunks (from a tile's last piece on: the next tile's)
                        if (last) derive(vi, nxt);
                        xload(vi, (c + 1) & 3);
                    }
                }
                if (last) cur = nxt;
                if constexpr (PX) {
                    if (last) *reinterpret_cast<v4f_g *>(cstrip + ((ti + 1) & 1) * (4 * kGF) + 4 * lane) = cnext;
                }
                __syncthreads();
            }
        }
        // lane (coordinate n, g): accU[t][r] = [dU | db_u][hid 16 t + 4 g + r][n]; theta keeps U as [k + 64 j], b_u behind it
        float *sl = slabs + (size_t)(blockIdx.x * 8 + p) * (kGH * d + kGH);
        if (n <= d) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) sl[16 * t + 4 * g + r + kGH * n] = accU[t][r];
        }
    }
}

// ---- athena_mp_gno_aggregate_bwd: the pieces around gno_dh_pc_kernel<.., PX = true> ------------------------------------------
// rows of more than 32 entries (they head the length-ordered list; a handful on a mesh): one workgroup per row builds
// G_i = g_i . Vmat^T in LDS and walks the row's entries -- px[0][w] = h_e^T G_i + b_v^T g_i, px[1][w] = 0.  Plain VALU.
__global__ __launch_bounds__(256) void gno_px_long_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ eidx,
                                                          const float *__restrict__ coords, const float *__restrict__ theta, int d,
                                                          const float *__restrict__ grad, int n_long, const int32_t *__restrict__ perm,
                                                          float *__restrict__ px, size_t px_half, const float *__restrict__ cvec)
{
    __shared__ float G[kGH][kGF + 1];
    __shared__ float gi[kGF], hs[4][kGH];
    const int row = perm[blockIdx.x], t = threadIdx.x;
    const float *V = theta + (size_t)kGH * d + kGH;
    if (t < kGF) gi[t] = grad[(size_t)row * kGF + t];
    __syncthreads();
    for (int kq = t; kq < kGH * kGF; kq += 256) {            // G[k][q] = sum_o V[o + 64 q + 4096 k] g[o]
        const float *v = V + (size_t)kq * kGF;
        float sacc = 0.0f;
        for (int o = 0; o < kGF; ++o) sacc = fmaf(v[o], gi[o], sacc);
        G[kq >> 6][kq & 63] = sacc;
    }
    __syncthreads();
    const int w0 = rowptr[row], len = rowptr[row + 1] - w0;
    const int slot = t >> 6, k = t & 63;
    for (int e0 = 0; e0 < len; e0 += 4) {
        const int w = w0 + e0 + slot;
        const bool in = e0 + slot < len;
        const int e = in ? eidx[w] : -1;
        float h = theta[(size_t)kGH * d + k];
        if (e >= 0)
            for (int j = 0; j < d; ++j) h = fmaf(theta[k + (size_t)kGH * j], coords[(size_t)e * d + j], h);
        hs[slot][k] = h > 0.0f ? h : 0.0f;
        __syncthreads();
        if (in) {
            float sacc = cvec[(size_t)row * kGF + k];   // k doubles as q
            for (int kk = 0; kk < kGH; ++kk) sacc = fmaf(hs[slot][kk], G[kk][k], sacc);
            px[(size_t)w * kGF + k] = e >= 0 ? sacc : 0.0f;
        }
        __syncthreads();
    }
}

// where the forward entry w = (v -> u) sits in the transposed CSR: t_entry[t] = w, or -1 when the entry carries no edge
// column (it contributes nothing, oracle: e < 0).  Deterministic: position of v in column u's ascending source list plus
// the rank of w among the row's earlier entries with the same neighbour.  One thread per row.
__global__ void gno_t_entry_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const int32_t *__restrict__ eid,
                                   const int32_t *__restrict__ t_rowptr, const int32_t *__restrict__ t_src, int n_rows,
                                   int32_t *__restrict__ t_entry)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_rows) return;
    const int b = rowptr[v], e = rowptr[v + 1];
    for (int w = b; w < e; ++w) {
        const int u = col[w];
        int lo = t_rowptr[u], hi = t_rowptr[u + 1];
        while (lo < hi) {                                     // first t with t_src[t] >= v
            const int mid = (lo + hi) >> 1;
            if (t_src[mid] < v) lo = mid + 1;
            else hi = mid;
        }
        int rank = 0;
        if (lo + 1 < t_rowptr[u + 1] && t_src[lo + 1] == v)   // the pair (v, u) occurs more than once (multigraph): only then scan
            for (int w2 = b; w2 < w; ++w2) rank += col[w2] == u;
        t_entry[lo + rank] = eid[w] >= 0 ? w : -1;
    }
}

static __device__ __forceinline__ v4f_g px_load(const float *p)
{
#if GNO_PX_NT_LOAD
    return __builtin_nontemporal_load(reinterpret_cast<const v4f_g *>(p));
#else
    return *reinterpret_cast<const v4f_g *>(p);
#endif
}

// dx[u,:] = sum over the transposed row of u of px[w] (64 floats each): 16 lanes x 16 bytes per column,
// sources ascending (the reference's accumulation order, athena_diffstruc_extd_sub_nop.f90:441-452)
__global__ __launch_bounds__(256) void gno_px_gather_kernel(const int32_t *__restrict__ t_rowptr, const int32_t *__restrict__ t_entry,
                                                            const float *__restrict__ px, int n_cols, float *__restrict__ dx)
{
    const int l = threadIdx.x & 15;
    const int u = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (u >= n_cols) return;
    const int b = t_rowptr[u], e = t_rowptr[u + 1];
    const v4f_g z = {0.0f, 0.0f, 0.0f, 0.0f};
    v4f_g acc = z;
    int t = b;
    for (; t + 7 < e; t += 8) {   // eight 16-byte loads in flight per lane; the sums stay in source order
        int w[8];
        v4f_g pv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = t_entry[t + i];
#if GNO_FV & 16384   // timing only: the partials read in the order the transposed CSR lists them (what a segment-ordered emission
                     // would give the gather: one stream); the sums are wrong
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = w[i] < 0 ? w[i] : t + i;
#endif
#pragma unroll
        for (int i = 0; i < 8; ++i) pv[i] = px_load(px + (size_t)(w[i] < 0 ? 0 : w[i]) * kGF + 4 * l);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (w[i] >= 0) acc = acc + pv[i];
    }
    for (; t + 1 < e; t += 2) {
        const int wa = t_entry[t], wb = t_entry[t + 1];
        const v4f_g pa = px_load(px + (size_t)(wa < 0 ? 0 : wa) * kGF + 4 * l), pb = px_load(px + (size_t)(wb < 0 ? 0 : wb) * kGF + 4 * l);
        if (wa >= 0) acc = acc + pa;
        if (wb >= 0) acc = acc + pb;
    }
    if (t < e) {
        const int wa = t_entry[t];
        if (wa >= 0) acc = acc + px_load(px + (size_t)wa * kGF + 4 * l);
    }
    *reinterpret_cast<v4f_g *>(dx + (size_t)u * kGF + 4 * l) = acc;
}

int gno_mlp_backward(const athena_mp_graph *g, int d, int H, int Fi, int Fo, const float *theta,
                     const float *coords, const float *x, const float *grad, float *dtheta, float *dcoords,
                     float *px = nullptr, size_t px_half = 0, const float *cvec = nullptr)
{
    const size_t off_V = (size_t)H * d + H;
    const int np = H * d + H;
    const int HF = H * Fi;
    float *ghbuf = nullptr;
    if (dcoords) {
        void *p = nullptr;
        if (amp::workspace(&p, sizeof(float) * (size_t)std::max<int64_t>(g->nnz, 1) * H, 4)) return 1;
        ghbuf = (float *)p;
    }
    if (gno_gdh_shape(H, Fi, Fo, d) && g->n_rows > 0) {
        // G is produced and consumed inside the workgroup.  Rows of at most 32 entries (all but a handful) go through the
        // dense / sparse kernel, the longer ones -- they head the length-ordered list -- through gno_gdh_kernel.
        if (length_order(g->rowptr, g->n_rows, &g->len_perm_fwd, &g->n_long_fwd, &g->n_mid_fwd)) return 1;
        const size_t y_bytes = sizeof(float) * kGF * (size_t)g->n_cols, c_bytes = sizeof(float) * (size_t)d * g->n_edge_cols,
                     id_bytes = sizeof(int32_t) * (size_t)g->nnz, g_bytes = sizeof(float) * kGF * (size_t)g->n_rows;
        const size_t lim = 0xFFFFE000ull;
        const bool pc_ok = d <= 3 && y_bytes < lim && c_bytes < lim && id_bytes < lim && g_bytes < lim;
        if (px && !pc_ok) {
            amp::set_error("gno_aggregate_bwd: this size does not take the producer / consumer kernels (caller must check gno_bwd_fused_ok)");
            return 2;
        }
        // three classes of the length-ordered list: > 32 entries | 17 .. 32 (2 vertices per sparse wave, 2 blocks) | <= 16 (4, 1)
        const int n_old = pc_ok ? g->n_long_fwd : g->n_rows, n_mid = pc_ok ? g->n_mid_fwd - g->n_long_fwd : 0,
                  n_short = g->n_rows - n_old - n_mid;
        void *vp = nullptr, *vd = nullptr, *sl = nullptr;
        const int nwg_old = std::min((n_old + kGRows - 1) / kGRows, 256), nwg_mid = std::min((n_mid + 15) / 16, amp::num_cus()),
                  nwg_short = std::min((n_short + 31) / 32, amp::num_cus());
        const int n_slabs = nwg_old * 16 + (nwg_mid + nwg_short) * 8;
        if (amp::workspace(&sl, sizeof(float) * (size_t)n_slabs * np, 3)) return 1;
        const int32_t *perm = (const int32_t *)g->len_perm_fwd;
        if (n_old > 0) {
            if (amp::workspace(&vp, sizeof(float) * 64 * 64 * 64, 1)) return 1;
            hipLaunchKernelGGL(gno_vperm_okq_kernel, dim3(64 * 64 * 64 / 256), dim3(256), 0, amp::stream(), theta + off_V, (float *)vp);
            AMP_LAUNCH_CHECK();
            constexpr size_t glds = sizeof(float) * ((size_t)kGRows * kDVtx + 16 * 128);   // G half + per-wave coordinates
            static amp::PerDeviceFlag gattr;
            if (!gattr.get()) {
                AMP_HIP(hipFuncSetAttribute((const void *)gno_gdh_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds));
                AMP_HIP(hipFuncSetAttribute((const void *)gno_gdh_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)glds));
                gattr.get() = true;
            }
            if (ghbuf)
                hipLaunchKernelGGL(gno_gdh_kernel<true>, dim3(nwg_old), dim3(1024), glds, amp::stream(), g->rowptr, g->col, g->eid, x,
                                   coords, theta, d, (const float *)vp, grad, n_old, perm, (float *)sl, ghbuf);
            else
                hipLaunchKernelGGL(gno_gdh_kernel<false>, dim3(nwg_old), dim3(1024), glds, amp::stream(), g->rowptr, g->col, g->eid, x,
                                   coords, theta, d, (const float *)vp, grad, n_old, perm, (float *)sl, ghbuf);
            AMP_LAUNCH_CHECK();
            if (px) {   // their share of the feature gradient's partials
                hipLaunchKernelGGL(gno_px_long_kernel, dim3(n_old), dim3(256), 0, amp::stream(), g->rowptr, g->eid, coords, theta, d, grad,
                                   n_old, perm, px, px_half, cvec);
                AMP_LAUNCH_CHECK();
            }
        }
        if (n_mid + n_short > 0) {
            if (amp::workspace(&vd, sizeof(float) * 64 * 64 * 64, 9)) return 1;
            hipLaunchKernelGGL(gno_vrelay_dense_kernel, dim3(64 * 64 * 64 / 256), dim3(256), 0, amp::stream(), theta + off_V, (float *)vd);
            AMP_LAUNCH_CHECK();
            constexpr size_t dlds = sizeof(float) * ((size_t)kDhLdsFloats + 8 * kDhCStrip);
            static amp::PerDeviceFlag dattr;
            if (!dattr.get()) {
                AMP_HIP(hipFuncSetAttribute((const void *)gno_dh_pc_kernel<false, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds));
                AMP_HIP(hipFuncSetAttribute((const void *)gno_dh_pc_kernel<true, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds));
                AMP_HIP(hipFuncSetAttribute((const void *)gno_dh_pc_kernel<false, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds));
                AMP_HIP(hipFuncSetAttribute((const void *)gno_dh_pc_kernel<true, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds));
                AMP_HIP(hipFuncSetAttribute((const void *)gno_dh_pc_kernel<false, 4, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds));
                AMP_HIP(hipFuncSetAttribute((const void *)gno_dh_pc_kernel<true, 4, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds));
                AMP_HIP(hipFuncSetAttribute((const void *)gno_dh_pc_kernel<false, 2, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds));
                AMP_HIP(hipFuncSetAttribute((const void *)gno_dh_pc_kernel<true, 2, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds));
                dattr.get() = true;
            }
#define AMP_DHPC_(GH_, VPW_, NB_, PX_, NWG_, NROWS_, PERM_, SL_)                                                                    \
    hipLaunchKernelGGL((gno_dh_pc_kernel<GH_, VPW_, NB_, PX_>), dim3(NWG_), dim3(kPcThreads), dlds, amp::stream(), g->rowptr, g->col, g->eid, \
                       x, coords, theta, d, (const float *)vd, grad, NROWS_, PERM_, SL_, ghbuf, (uint32_t)y_bytes, (uint32_t)c_bytes, \
                       (uint32_t)id_bytes, (uint32_t)g_bytes, px, px_half, cvec)
#define AMP_DHPC(GH_, VPW_, NB_, NWG_, NROWS_, PERM_, SL_)                \
    do {                                                                  \
        if (px) AMP_DHPC_(GH_, VPW_, NB_, true, NWG_, NROWS_, PERM_, SL_); \
        else AMP_DHPC_(GH_, VPW_, NB_, false, NWG_, NROWS_, PERM_, SL_);   \
    } while (0)
            float *sl_mid = (float *)sl + (size_t)nwg_old * 16 * np, *sl_short = sl_mid + (size_t)nwg_mid * 8 * np;
            if (n_mid > 0) {
                if (ghbuf) AMP_DHPC(true, 2, 2, nwg_mid, n_mid, perm + n_old, sl_mid);
                else AMP_DHPC(false, 2, 2, nwg_mid, n_mid, perm + n_old, sl_mid);
                AMP_LAUNCH_CHECK();
            }
            if (n_short > 0) {
                if (ghbuf) AMP_DHPC(true, 4, 1, nwg_short, n_short, perm + n_old + n_mid, sl_short);
                else AMP_DHPC(false, 4, 1, nwg_short, n_short, perm + n_old + n_mid, sl_short);
                AMP_LAUNCH_CHECK();
            }
#undef AMP_DHPC
#undef AMP_DHPC_
        }
        const int nwg = n_slabs / 16;   // (slab_reduce below takes the slab count)
        (void)nwg;
        if (dtheta) {
            if (int rc2 = amp::slab_reduce((const float *)sl, n_slabs, np, dtheta, false)) return rc2;
        }
        if (dcoords && g->n_edge_cols > 0) {
            int64_t n = (int64_t)g->n_edge_cols * d;
            hipLaunchKernelGGL(gno_dcoords_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, amp::stream(),
                               g->e_rowptr, g->e_col, ghbuf, theta, d, H, g->n_edge_cols, dcoords);
            AMP_LAUNCH_CHECK();
        }
        return 0;
    }
    const int tile = tile_rows_for(g->n_rows, HF);
    const size_t lds = sizeof(float) * ((size_t)H * (Fi + 1) + (size_t)kEB * (Fi + H) + np);
    static amp::PerDeviceFlag attr;
    if (!attr.get()) {
        AMP_HIP(hipFuncSetAttribute((const void *)gno_dh_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr.get() = true;
    }
    bool first = true;
    for (int r0 = 0; r0 < g->n_rows; r0 += tile) {
        const int rows = std::min(tile, g->n_rows - r0);
        void *gw = nullptr;
        if (amp::workspace(&gw, sizeof(float) * (size_t)rows * HF, 1)) return 1;
        // G[r, (k,q)] = sum_o g[r,o] V[o + Fo*q + F*k]: B stored [N=(k,q)][K=o] = theta + off_V
        int rc = amp::gemm_dispatch(grad + (size_t)r0 * Fo, theta + off_V, 1, nullptr, ATHENA_MP_ACT_NONE,
                                    (float *)gw, rows, Fo, HF);
        if (rc) return rc;
        int nblk;
        void *sl = nullptr;
        const bool mf = H % 32 == 0 && Fi % 32 == 0 && H <= 64 && Fi <= 64 && d <= 3;
        if (mf) {
            nblk = std::min((rows + 3) / 4, 512);      // 4 waves per workgroup, one slab per wave
            if (amp::workspace(&sl, sizeof(float) * (size_t)nblk * 4 * np, 3)) return 1;
#define AMP_DH(A_, B_)                                                                                     \
    if (H == 32 * A_ && Fi == 32 * B_)                                                                     \
        hipLaunchKernelGGL((gno_dh_mfma_kernel<A_, B_>), dim3(nblk), dim3(256), 0, amp::stream(), g->rowptr, g->col, \
                           g->eid, x, coords, theta, d, (const float *)gw, r0, rows, (float *)sl, ghbuf);
            AMP_DH(1, 1) AMP_DH(1, 2) AMP_DH(2, 1) AMP_DH(2, 2)
#undef AMP_DH
            nblk *= 4;
        } else {
            nblk = std::min(rows, 512);
            int rpb = (rows + nblk - 1) / nblk;
            nblk = (rows + rpb - 1) / rpb;
            if (amp::workspace(&sl, sizeof(float) * (size_t)nblk * np, 3)) return 1;
            hipLaunchKernelGGL(gno_dh_kernel, dim3(nblk), dim3(256), lds, amp::stream(), g->rowptr, g->col, g->eid, x, Fi,
                               coords, theta, d, H, (const float *)gw, r0, rows, rpb, (float *)sl, ghbuf);
        }
        AMP_LAUNCH_CHECK();
        if (dtheta) {
            if (int rc2 = amp::slab_reduce((const float *)sl, nblk, np, dtheta, !first)) return rc2;
        }
        first = false;
    }
    if (dtheta && g->n_rows == 0) AMP_HIP(hipMemsetAsync(dtheta, 0, sizeof(float) * np, amp::stream()));
    if (dcoords && g->n_edge_cols > 0) {
        int64_t n = (int64_t)g->n_edge_cols * d;
        hipLaunchKernelGGL(gno_dcoords_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, amp::stream(),
                           g->e_rowptr, g->e_col, ghbuf, theta, d, H, g->n_edge_cols, dcoords);
        AMP_LAUNCH_CHECK();
    }
    return 0;
}

} // namespace

using namespace amp;

extern "C" {

int athena_mp_gno_aggregate_fwd(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                                const float *theta, const float *coords, const float *x, float *m)
{
    if (!gno_args_ok(g, d, H, Fi, Fo)) return 2;
    AMP_REQUIRE(theta && coords && x && m, "gno_aggregate_fwd: null pointer");
    const size_t off_V = (size_t)H * d + H;
    const int R = (H + 1) * Fi;
    if (gno_fused_shape(H, Fi, Fo, d))
        return launch_gno_fused(g->rowptr, g->col, g->eid, x, coords, theta, d, theta + off_V, g->n_rows, &g->len_perm_fwd, m,
                                g->n_cols, g->n_edge_cols, g->nnz, &g->n_long_fwd, &g->n_mid_fwd);
    const int tile = tile_rows_for(g->n_rows, R);
    for (int r0 = 0; r0 < g->n_rows; r0 += tile) {
        const int rows = std::min(tile, g->n_rows - r0);
        void *ws = nullptr;
        if (workspace(&ws, sizeof(float) * (size_t)rows * R, 0)) return 1;
        int rc = launch_outer(g->rowptr, g->col, g->eid, x, Fi, coords, theta, d, H, r0, rows, (float *)ws);
        if (rc) return rc;
        // m[r,o] = sum_{(k,q)} S[r,(k,q)] Vaug[o + Fo*(q + Fi*k)]: B = theta+off_V viewed [R][Fo] row-major
        rc = gemm_dispatch((const float *)ws, theta + off_V, 0, nullptr, ATHENA_MP_ACT_NONE, m + (size_t)r0 * Fo,
                           rows, R, Fo);
        if (rc) return rc;
    }
    return 0;
}

// ---- training-mode pair: the forward pass keeps S for the reverse pass's S^T g ---------------------------------------
int athena_mp_gno_saved_bytes(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo, int64_t *bytes)
{
    if (!gno_args_ok(g, d, H, Fi, Fo)) return 2;
    AMP_REQUIRE(bytes, "gno_saved_bytes: null pointer");
    const bool ok = g->n_rows > 0 && gno_fused_shape(H, Fi, Fo, d) && gno_stg_shape(H, Fi, Fo, d) &&
                    gno_pc_route(d, g->n_cols, g->n_edge_cols, g->nnz) &&
                    sizeof(float) * kGF * (size_t)g->n_rows < 0xFFFFE000ull;
    const int64_t n_tiles = (g->n_rows + kPV - 1) / kPV;
    *bytes = ok ? (int64_t)sizeof(float) * n_tiles * (int64_t)kSaveTile : 0;
    return 0;
}

int athena_mp_gno_aggregate_fwd_save(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                                     const float *theta, const float *coords, const float *x, float *m, float *s_save)
{
    if (!gno_args_ok(g, d, H, Fi, Fo)) return 2;
    AMP_REQUIRE(theta && coords && x && m && s_save, "gno_aggregate_fwd_save: null pointer");
    int64_t bytes = 0;
    if (int rc = athena_mp_gno_saved_bytes(g, d, H, Fi, Fo, &bytes)) return rc;
    AMP_REQUIRE(bytes > 0, "gno_aggregate_fwd_save: this shape does not keep S (athena_mp_gno_saved_bytes returned 0)");
    const size_t off_V = (size_t)H * d + H;
    return launch_gno_fused(g->rowptr, g->col, g->eid, x, coords, theta, d, theta + off_V, g->n_rows, &g->len_perm_fwd, m,
                            g->n_cols, g->n_edge_cols, g->nnz, &g->n_long_fwd, &g->n_mid_fwd, s_save);
}

int athena_mp_gno_aggregate_bwd_theta_saved(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                                            const float *theta, const float *coords, const float *x, const float *grad,
                                            const float *s_save, float *dtheta)
{
    if (!gno_args_ok(g, d, H, Fi, Fo)) return 2;
    AMP_REQUIRE(theta && coords && x && grad && s_save && dtheta, "gno_aggregate_bwd_theta_saved: null pointer");
    int64_t bytes = 0;
    if (int rc = athena_mp_gno_saved_bytes(g, d, H, Fi, Fo, &bytes)) return rc;
    AMP_REQUIRE(bytes > 0, "gno_aggregate_bwd_theta_saved: this shape does not keep S (athena_mp_gno_saved_bytes returned 0)");
    const size_t off_V = (size_t)H * d + H;
    const int rc = launch_gno_stg(g, x, coords, theta, d, grad, dtheta + off_V, s_save);
    if (rc) {
        if (rc < 0) set_error("gno_aggregate_bwd_theta_saved: tensors beyond the 4 GB a buffer descriptor addresses");
        return rc < 0 ? 2 : rc;
    }
    return gno_mlp_backward(g, d, H, Fi, Fo, theta, coords, x, grad, dtheta, nullptr);
}

int athena_mp_gno_aggregate_bwd_x(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                                  const float *theta, const float *coords, const float *grad, float *dx)
{
    if (!gno_args_ok(g, d, H, Fi, Fo)) return 2;
    AMP_REQUIRE(theta && coords && grad && dx, "gno_aggregate_bwd_x: null pointer");
    const size_t off_V = (size_t)H * d + H;
    const int R2 = (H + 1) * Fo;
    void *b2 = nullptr;
    if (workspace(&b2, sizeof(float) * (size_t)R2 * Fi, 3)) return 1;
    {
        int n = R2 * Fi;
        hipLaunchKernelGGL(gno_perm_kernel, dim3((n + 255) / 256), dim3(256), 0, stream(), theta + off_V, H + 1, Fi,
                           Fo, (float *)b2);
        AMP_LAUNCH_CHECK();
    }
    if (gno_fused_shape(H, Fo, Fi, d))
        return launch_gno_fused(g->t_rowptr, g->t_src, g->t_eid, grad, coords, theta, d, (const float *)b2, g->n_cols,
                                &g->len_perm_bwd, dx, g->n_rows, g->n_edge_cols, g->nnz);
    const int tile = tile_rows_for(g->n_cols, R2);
    for (int r0 = 0; r0 < g->n_cols; r0 += tile) {
        const int rows = std::min(tile, g->n_cols - r0);
        void *ws = nullptr;
        if (workspace(&ws, sizeof(float) * (size_t)rows * R2, 0)) return 1;
        int rc = launch_outer(g->t_rowptr, g->t_src, g->t_eid, grad, Fo, coords, theta, d, H, r0, rows, (float *)ws);
        if (rc) return rc;
        rc = gemm_dispatch((const float *)ws, (const float *)b2, 0, nullptr, ATHENA_MP_ACT_NONE,
                           dx + (size_t)r0 * Fi, rows, R2, Fi);
        if (rc) return rc;
    }
    return 0;
}

/* The reverse pass of gno_aggregate towards the features in PULL form over the graph's OWN rows:
 *     dx[v,:] = sum_{w in row v} K_{eid[w]}^T grad_ext[col[w],:]          v < n_rows, grad_ext [n_cols, Fo]
 * For an undirected graph whose two directions share one edge column (athena's graphs: row u lists (v, e) whenever row v
 * lists (u, e), athena_diffstruc_extd_sub_nop.f90:369-376) this is the reference's scatter dx_j += K_e^T g_i
 * (:419-458) read from the receiving side.  It is what a row block of a partitioned graph can evaluate: the block holds
 * its own rows, and the gradient rows of the remote neighbours arrive by the same halo exchange as the features
 * (athena_mp_shard_create_edges checks the symmetry).  Same kernels as athena_mp_gno_aggregate_bwd_x, walking
 * rowptr / col / eid instead of the transposed arrays. */
int athena_mp_gno_aggregate_bwd_x_pull(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                                       const float *theta, const float *coords, const float *grad_ext, float *dx)
{
    if (!gno_args_ok(g, d, H, Fi, Fo)) return 2;
    AMP_REQUIRE(theta && coords && grad_ext && dx, "gno_aggregate_bwd_x_pull: null pointer");
    const size_t off_V = (size_t)H * d + H;
    const int R2 = (H + 1) * Fo;
    void *b2 = nullptr;
    if (workspace(&b2, sizeof(float) * (size_t)R2 * Fi, 3)) return 1;
    {
        int n = R2 * Fi;
        hipLaunchKernelGGL(gno_perm_kernel, dim3((n + 255) / 256), dim3(256), 0, stream(), theta + off_V, H + 1, Fi,
                           Fo, (float *)b2);
        AMP_LAUNCH_CHECK();
    }
    if (gno_fused_shape(H, Fo, Fi, d))
        return launch_gno_fused(g->rowptr, g->col, g->eid, grad_ext, coords, theta, d, (const float *)b2, g->n_rows,
                                &g->len_perm_fwd, dx, g->n_cols, g->n_edge_cols, g->nnz, &g->n_long_fwd, &g->n_mid_fwd);
    const int tile = tile_rows_for(g->n_rows, R2);
    for (int r0 = 0; r0 < g->n_rows; r0 += tile) {
        const int rows = std::min(tile, g->n_rows - r0);
        void *ws = nullptr;
        if (workspace(&ws, sizeof(float) * (size_t)rows * R2, 0)) return 1;
        int rc = launch_outer(g->rowptr, g->col, g->eid, grad_ext, Fo, coords, theta, d, H, r0, rows, (float *)ws);
        if (rc) return rc;
        rc = gemm_dispatch((const float *)ws, (const float *)b2, 0, nullptr, ATHENA_MP_ACT_NONE,
                           dx + (size_t)r0 * Fi, rows, R2, Fi);
        if (rc) return rc;
    }
    return 0;
}

int athena_mp_gno_aggregate_bwd_theta(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                                      const float *theta, const float *coords, const float *x, const float *grad,
                                      float *dtheta)
{
    if (!gno_args_ok(g, d, H, Fi, Fo)) return 2;
    AMP_REQUIRE(theta && coords && x && grad && dtheta, "gno_aggregate_bwd_theta: null pointer");
    const size_t off_V = (size_t)H * d + H;
    const int R = (H + 1) * Fi;
    // dVaug = S^T g  (dV and db_v in one contraction; the bias row of S is s_i = sum_j x_j)
    // S goes through HBM one super-tile at a time.  The two kernels of a super-tile lean on different resources -- the
    // outer product is bound by its 2 GiB of writes, the contraction by the matrix pipe and reads -- so they are
    // pipelined over two workspace slots: the outer product of tile t+1 runs on the library's second stream while the
    // contraction of tile t runs on the caller's.
    const int tile = tile_rows_for(g->n_rows, R);
    if (g->n_rows == 0) AMP_HIP(hipMemsetAsync(dtheta + off_V, 0, sizeof(float) * (size_t)R * Fo, stream()));
    if (g->n_rows > 0 && gno_stg_shape(H, Fi, Fo, d)) {   // S stays on chip (gno_stg_kernel)
        const int rc = launch_gno_stg(g, x, coords, theta, d, grad, dtheta + off_V);
        if (rc == 0) return gno_mlp_backward(g, d, H, Fi, Fo, theta, coords, x, grad, dtheta, nullptr);
        if (rc > 0) return rc;
    }
    const int n_tiles = g->n_rows > 0 ? (g->n_rows + tile - 1) / tile : 0;
    if (n_tiles < 2) {
        for (int r0 = 0; r0 < g->n_rows; r0 += tile) {
            const int rows = std::min(tile, g->n_rows - r0);
            void *ws = nullptr;
            if (workspace(&ws, sizeof(float) * (size_t)rows * R, 0)) return 1;
            int rc = launch_outer(g->rowptr, g->col, g->eid, x, Fi, coords, theta, d, H, r0, rows, (float *)ws);
            if (rc) return rc;
            rc = gemm_dw_dispatch(rows, R, Fo, (const float *)ws, grad + (size_t)r0 * Fo, dtheta + off_V, r0 > 0);
            if (rc) return rc;
        }
    } else {
        hipStream_t main_s = stream(), aux = nullptr;
        hipEvent_t *ev = nullptr;
        if (amp::aux_stream(&aux, &ev)) return 1;
        void *wsl[2] = {nullptr, nullptr};
        if (workspace(&wsl[0], sizeof(float) * (size_t)tile * R, 0) || workspace(&wsl[1], sizeof(float) * (size_t)tile * R, 8)) return 1;
        // ev[0]: inputs ready (main -> aux); ev[1 + slot]: S of the slot written (aux -> main); ev[3 + slot]: consumed
        AMP_HIP(hipEventRecord(ev[0], main_s));
        AMP_HIP(hipStreamWaitEvent(aux, ev[0], 0));
        int rc = 0;
        for (int t = 0; t < n_tiles && rc == 0; ++t) {
            const int slot = t & 1, r0 = t * tile, rows = std::min(tile, g->n_rows - r0);
            if (t >= 2) rc = hipStreamWaitEvent(aux, ev[3 + slot], 0) == hipSuccess ? 0 : 1;
            amp::swap_stream(aux);
            if (rc == 0) rc = launch_outer(g->rowptr, g->col, g->eid, x, Fi, coords, theta, d, H, r0, rows, (float *)wsl[slot]);
            amp::swap_stream(main_s);
            if (rc == 0 && hipEventRecord(ev[1 + slot], aux) != hipSuccess) rc = 1;
            if (rc == 0 && hipStreamWaitEvent(main_s, ev[1 + slot], 0) != hipSuccess) rc = 1;
            if (rc == 0) rc = gemm_dw_dispatch(rows, R, Fo, (const float *)wsl[slot], grad + (size_t)r0 * Fo, dtheta + off_V, t > 0);
            if (rc == 0 && hipEventRecord(ev[3 + slot], main_s) != hipSuccess) rc = 1;
        }
        amp::swap_stream(main_s);
        if (rc) {
            if (athena_mp_last_error()[0] == 0) set_error("gno_aggregate_bwd_theta: stream pipeline failed");
            return rc;
        }
    }
    return gno_mlp_backward(g, d, H, Fi, Fo, theta, coords, x, grad, dtheta, nullptr);
}

/* The whole reverse pass of gno_aggregate from ONE G = g . Vmat^T (athena_diffstruc_extd_sub_nop.f90:419-458 features,
 * :235-325 kernel parameters, :137-216 coordinates): any of dx / dtheta / dcoords may be NULL.  Shapes and sizes that take
 * the producer / consumer kernels (H = F_in = F_out = 64, d <= 3, tensors below 4 GB, at most 1 row in 64 longer than 32
 * entries) run:  S^T g (streamed from s_save when given, rebuilt otherwise) -> dVaug;  gno_dh_pc_kernel<PX> -> dU, db_u,
 * the per-entry partials of dx [nnz][64] (and the per-entry dh for dcoords);  gno_px_gather_kernel -> dx.  Everything
 * else is the three separate entry points, one after the other.  *fused (may be NULL) says which it was. */
int athena_mp_gno_aggregate_bwd(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo, const float *theta,
                                const float *coords, const float *x, const float *grad, const float *s_save, float *dx,
                                float *dtheta, float *dcoords, int32_t *fused)
{
    if (!gno_args_ok(g, d, H, Fi, Fo)) return 2;
    AMP_REQUIRE(theta && coords && x && grad, "gno_aggregate_bwd: null pointer");
    if (fused) *fused = 0;
    bool ok = dx && g->n_rows > 0 && gno_gdh_shape(H, Fi, Fo, d) && gno_stg_shape(H, Fi, Fo, d) &&
              gno_pc_route(d, g->n_cols, g->n_edge_cols, g->nnz) && sizeof(float) * kGF * (size_t)g->n_rows < 0xFFFFE000ull &&
              sizeof(float) * kGF * (size_t)g->n_cols < 0xFFFFE000ull;
    if (ok) {
        if (length_order(g->rowptr, g->n_rows, &g->len_perm_fwd, &g->n_long_fwd, &g->n_mid_fwd)) return 1;
        ok = (int64_t)g->n_long_fwd * 64 <= (int64_t)g->n_rows;   // the long rows' partials come from a plain VALU kernel
    }
    auto separate = [&]() -> int {   // the three entry points one after the other: no workspace of nnz * 256 bytes
        if (dtheta) {
            const int rc = s_save ? athena_mp_gno_aggregate_bwd_theta_saved(g, d, H, Fi, Fo, theta, coords, x, grad, s_save, dtheta)
                                  : athena_mp_gno_aggregate_bwd_theta(g, d, H, Fi, Fo, theta, coords, x, grad, dtheta);
            if (rc) return rc;
        }
        if (dx)
            if (const int rc = athena_mp_gno_aggregate_bwd_x(g, d, H, Fi, Fo, theta, coords, grad, dx)) return rc;
        if (dcoords)
            if (const int rc = athena_mp_gno_aggregate_bwd_coords(g, d, H, Fi, Fo, theta, coords, x, grad, dcoords)) return rc;
        return 0;
    };
    if (!ok) return separate();
    const size_t off_V = (size_t)H * d + H;
    const size_t px_half = (size_t)std::max<int64_t>(g->nnz, 1) * kGF;
    void *pxp = nullptr, *cvp = nullptr, *dth_tmp = nullptr;
    // The fused route keeps nnz * 256 bytes of per-entry partials (7.6 GB at BASELINE configs[3]) and nnz int32 in the handle.
    // When the device cannot give them -- a larger graph, a smaller device, a layer that also keeps S -- the call does what it
    // did before the fused route existed: the separate entry points, which run in the memory there is.
    // Whichever of the three allocations fails, what the earlier ones took goes back first: the separate entry points must find
    // at least the memory they had before the fused route existed.
    auto give_up = [&]() -> int {
        (void)hipGetLastError();   // the failed allocation must not surface in the next launch check
        if (workspace_release(13) || workspace_release(14)) return 1;
        return separate();
    };
    if (workspace(&pxp, sizeof(float) * px_half, 13) || workspace(&cvp, sizeof(float) * kGF * (size_t)g->n_rows, 14)) return give_up();
    if (!g->t_entry && g->nnz > 0) {
        int32_t *te = nullptr;
        if (hipMalloc((void **)&te, sizeof(int32_t) * (size_t)g->nnz) != hipSuccess) return give_up();
        hipLaunchKernelGGL(gno_t_entry_kernel, dim3((g->n_rows + 255) / 256), dim3(256), 0, stream(), g->rowptr, g->col, g->eid,
                           g->t_rowptr, g->t_src, g->n_rows, te);
        if (hipGetLastError() != hipSuccess) {   // the handle only ever holds a map that was built
            (void)hipFree(te);
            set_error("gno_aggregate_bwd: launch of the transposed-entry map failed");
            return 1;
        }
        g->t_entry = te;
    }
    // c_i = b_v^T g_i for every row (b_v viewed [q][o]): one weight-resident GEMM launch
    if (int rc = gemm_dispatch(grad, theta + off_V + (size_t)Fo * Fi * H, /*b_nk=*/1, nullptr, ATHENA_MP_ACT_NONE, (float *)cvp,
                               g->n_rows, Fo, Fi))
        return rc;
    float *dth = dtheta;
    if (!dth) {   // dx alone still runs the fused kernel; its parameter sums go to a scratch vector
        if (workspace(&dth_tmp, sizeof(float) * (off_V + (size_t)Fo * Fi * (H + 1)), 15)) return 1;
        dth = (float *)dth_tmp;
    }
    // Order: the kernel MLP's launches (they write the partials) -> the gather of the partials and S^T g SIDE BY SIDE.  The two
    // share nothing but grad: the gather is a stream of 16-byte loads without LDS in 56 registers, S^T g leaves exactly that
    // many free per SIMD (3 waves x 152) and is bound by the matrix pipe, so the gather runs on the library's second stream in
    // the slots S^T g cannot use (A/B against everything on the caller's stream: docs/history/DESIGN_r01-r04.md, round 4).
    if (int rc = gno_mlp_backward(g, d, H, Fi, Fo, theta, coords, x, grad, dth, dcoords, (float *)pxp, px_half, (const float *)cvp)) return rc;
    hipStream_t main_s = stream(), gs = main_s;
    hipEvent_t *ev = nullptr;
    if (dtheta) {   // (side by side 26.6 - 27.3 ms against 28.9 with every launch on one stream: profiles/r04_c4_config.jsonl)
        if (amp::aux_stream(&gs, &ev)) return 1;
        AMP_HIP(hipEventRecord(ev[0], main_s));
        AMP_HIP(hipStreamWaitEvent(gs, ev[0], 0));
    }
    hipLaunchKernelGGL(gno_px_gather_kernel, dim3((g->n_cols + 15) / 16), dim3(256), 0, gs, g->t_rowptr, g->t_entry,
                       (const float *)pxp, g->n_cols, dx);
    const bool launched = hipGetLastError() == hipSuccess;
    if (gs != main_s) AMP_HIP(hipEventRecord(ev[1], gs));
    int rc = 0;
    if (launched && dtheta) rc = launch_gno_stg(g, x, coords, theta, d, grad, dtheta + off_V, s_save);
    // joined whatever happened above: the caller's stream never leaves this call with work it cannot see
    if (gs != main_s) AMP_HIP(hipStreamWaitEvent(main_s, ev[1], 0));
    if (!launched) {
        set_error("gno_aggregate_bwd: launch of the partials' gather failed");
        return 1;
    }
    if (rc) {
        if (rc < 0) set_error("gno_aggregate_bwd: tensors beyond the 4 GB a buffer descriptor addresses");
        return rc < 0 ? 2 : rc;
    }
    if (fused) *fused = 1;
    return 0;
}

int athena_mp_gno_aggregate_bwd_coords(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                                       const float *theta, const float *coords, const float *x,
                                       const float *grad, float *dcoords)
{
    if (!gno_args_ok(g, d, H, Fi, Fo)) return 2;
    AMP_REQUIRE(theta && coords && x && grad && dcoords, "gno_aggregate_bwd_coords: null pointer");
    return gno_mlp_backward(g, d, H, Fi, Fo, theta, coords, x, grad, nullptr, dcoords);
}

} // extern "C"

namespace amp {
// a graph handle is being freed: its cached row-length orders take their counts with them (capi.hip)
void gno_forget_perm(const int32_t *perm_dev) { g_len_counts.erase(perm_dev); }
} // namespace amp
