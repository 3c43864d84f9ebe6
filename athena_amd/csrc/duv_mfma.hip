// Degree-bucketed Duvenaud update on the matrix cores with the weights held in REGISTERS.
//
//   duvenaud_update                        athena_diffstruc_extd_sub_duvenaud.f90:176-228
//   get_partial_duvenaud_update_val        :284-324
//   get_partial_duvenaud_update_weight_val :326-368
//
// The graph handle keeps the vertices sorted by bucket and cut into 16-vertex tiles
// (amp::duvenaud_buckets).  Every 16x16x4 MFMA takes the 16 vertices of a tile on its COLUMN axis:
//   c^T[o, v] = sum_k W_d[o, k] a[v, k]       A operand = W_d fragment (registers, per bucket)
//                                             B operand = a[v, 16j + 4q .. +3], one 16 B load per lane
// so vertex rows are read and written 16 B per lane straight from/to HBM (no LDS staging of the
// streamed operand), and the accumulator of output tile `ot` is c[v, 16 ot + 4q .. +3] -- the same
// lane layout, ready for a 16 B store.  A wave owns a contiguous run of tiles and reloads the weight
// fragments (from L2) only when the bucket changes.
//
// The weight gradient contracts over VERTICES, so both streamed operands are turned once through a
// wave-private LDS tile; per-bucket partial sums go to slab (workgroup + bucket) -- a contiguous slab
// range per bucket, reduced in fixed order (deterministic, no atomics).
#include <algorithm>
#include <map>

#include "common.h"

#ifndef DUV_VARIANT
#define DUV_VARIANT 0   // timing-only diagnostic builds (scripts/build_variants.sh), never shipped: bit mask
                        // 1 no matrix work | 2 no row loads | 4 no stores | 8 no activation / divisor
#endif

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

// Epilogue arithmetic is kept off the critical path of the matrix pipe: the sigmoid uses the hardware
// exp2 / rcp (each 1 ulp; the result is within ~4e-7 relative of the libm form, inside the 1e-5 bound the
// MFMA route is tested to), and x/d for the bucket divisor d uses one Newton correction on x*(1/d):
// 3 instructions instead of the ~10 of the IEEE division sequence, equal to the IEEE quotient in all but
// ~4 of 1e5 operands and 1 ulp off otherwise (checked exhaustively-by-sampling for d = 1..64).
__device__ __forceinline__ float act_apply(float t, int act)
{
    switch (act) {
    case ATHENA_MP_ACT_RELU: return t > 0.0f ? t : 0.0f;
    case ATHENA_MP_ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t * -1.4426950408889634f));
    case ATHENA_MP_ACT_TANH: return tanhf(t);
    default: return t;
    }
}
__device__ __forceinline__ float div_by(float x, float d, float inv)
{
    const float q = x * inv;
    return fmaf(fmaf(-q, d, x), inv, q);
}

// Y[v, 0:NO] = f( sum_k Wd(o,k) X[v,k] ),  Wd(o,k) = W[b*wb + o*so + k*sk]
//   div_in  = 1: X is divided by d = b+1 before the product (forward: a/d, the reference's order)
//   div_in  = 0: the sum is divided by d afterwards (reverse w.r.t. a)
// Work split: every bucket gets a share of the waves (workgroups for the weight gradient) proportional to
// its tile count, and the waves of a bucket stride through that bucket's tiles.  All buckets therefore
// sweep the vertex array front to back at the same relative pace: the rows a DRAM page holds are asked for
// by the different buckets at about the same time (a bucket-after-bucket schedule touches every page once
// per bucket, 288 B at a time), and a wave never changes its weight fragments.
constexpr int kMaxBuckets = 32;
struct BucketSplit {
    int unit_off[kMaxBuckets + 1];   // first wave / workgroup of each bucket
    int tile_off[kMaxBuckets + 1];   // first tile of each bucket
    int n_buckets;
};

typedef float v2f __attribute__((ext_vector_type(2)));

// x / d for four values on the packed fp32 pipe (v_pk_mul / v_pk_fma: two values per instruction): q = x * (1/d), one Newton
// correction -- the IEEE quotient in all but ~4 of 1e5 operands, 1 ulp off otherwise
__device__ __forceinline__ v4f div4(v4f x, float d, float inv)
{
    const v2f dd = {d, d}, ii = {inv, inv};
    v2f lo = {x[0], x[1]}, hi = {x[2], x[3]};
    const v2f ql = lo * ii, qh = hi * ii;
    lo = __builtin_elementwise_fma(__builtin_elementwise_fma(-ql, dd, lo), ii, ql);
    hi = __builtin_elementwise_fma(__builtin_elementwise_fma(-qh, dd, hi), ii, qh);
    return v4f{lo[0], lo[1], hi[0], hi[1]};
}

// Round 3 -- what bounded round 2's kernels, measured with timing-only builds (DUV_VARIANT): memory path alone (row
// loads + epilogue + stores, no matrix work) 0.29 ms, matrix work without the row loads 0.28 ms, together 0.39 ms.  The
// memory path was slow because every lane read and wrote its vertex row 16 bytes at a time in the MFMA operand layout
// (lane (v, q) <-> a[v, 16 j + 4 q ..]): the 16 lanes the address unit takes together touched 16 different rows, so
// one 1 KB load or store became 64 partial-line requests.  Now the rows travel in a COALESCED lane mapping -- the
// 16 * K/4 sixteen-byte chunks of a tile are numbered row-major and lane l of load i takes chunk 64 i + l, so consecutive
// lanes read consecutive bytes of one row -- and are turned into the operand layout through a wave-private LDS tile
// (LDS operations do not use the vector ALU, which fp32 MFMAs share); results go back the same way.  The hot loop is
// branch-free (round 2 predicated loads and stores per lane: ~1100 basic blocks and `s_waitcnt vmcnt(0)` at three joins):
// a tile index past the wave's last tile is clamped to it, chunk numbers past the end of a tile re-read chunk 0, padding
// slots of a bucket's last tile repeat its first vertex (same loads, same values, same stores), K and NO tails are zero
// columns of the LDS tiles that no load writes and no store reads.  The activation is a wave-uniform switch around the
// epilogue and the divisor is applied to the 16 OUTPUT values on the packed pipe (folded into the exponent's constant
// for the sigmoid): W_d (a / d) = (W_d a) / d up to rounding, inside the 1e-5 this route is held to.
//
// Y[v, 0:NO] = act( (sum_k Wd(o,k) X[v,k]) / d ),  Wd(o,k) = W[b*wb + o*so + k*sk],  d = b + 1
template <int KJ, int OT>
__global__ __launch_bounds__(256) void duv_rows_any_kernel(BucketSplit sp, const int32_t *__restrict__ trows,
                                                       const float *__restrict__ X, int K,
                                                       const float *__restrict__ W, int64_t wb, int so, int sk,
                                                       float *__restrict__ Y, int NO, int act)
{
    constexpr int PI = 16 * KJ + 4, PO = 16 * OT + 4;   // pitches: 16-byte reads with the vertex on the lane index are conflict-free
    __shared__ __attribute__((aligned(16))) float lds[4 * 16 * (PI + PO)];
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    float *tin = lds + wave * 16 * (PI + PO), *tout = tin + 16 * PI;
    const int gw = blockIdx.x * 4 + wave;
    if (gw >= sp.unit_off[sp.n_buckets]) return;
    int b = 0;
    while (gw >= sp.unit_off[b + 1]) ++b;
    const int nw = sp.unit_off[b + 1] - sp.unit_off[b];
    const int t0 = sp.tile_off[b] + (gw - sp.unit_off[b]), t1 = sp.tile_off[b + 1];
    if (t0 >= t1) return;
    const int cnt = (t1 - t0 + nw - 1) / nw;          // tiles of this wave: t0, t0 + nw, ...

    const float d = (float)(b + 1), inv = 1.0f / d;   // the bucket index is the divisor
    float Wf[OT][KJ][4];
    {
        const float *wd = W + (int64_t)b * wb;
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
#pragma unroll
            for (int j = 0; j < KJ; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int o = 16 * ot + n, k = 16 * j + 4 * q + c;
                    const bool ok = o < NO && k < K;
                    const float v = wd[(int64_t)(ok ? o : 0) * so + (int64_t)(ok ? k : 0) * sk];   // never out of range
                    Wf[ot][j][c] = ok ? v : 0.0f;
                }
    }
    // the coalesced chunk numbering, fixed per lane: chunk s = 64 i + lane of a tile is row s / (K/4), columns 4 (s % (K/4)) ..
    int in_row[KJ], in_col[KJ], out_row[OT], out_col[OT];
    {
        const int ci = K >> 2, co = NO >> 2;
#pragma unroll
        for (int i = 0; i < KJ; ++i) {
            const int sl = 64 * i + lane, ok = sl < 16 * ci;
            in_row[i] = ok ? sl / ci : 0;
            in_col[i] = ok ? 4 * (sl - in_row[i] * ci) : 0;
        }
#pragma unroll
        for (int i = 0; i < OT; ++i) {
            const int sl = 64 * i + lane, ok = sl < 16 * co;
            out_row[i] = ok ? sl / co : 0;
            out_col[i] = ok ? 4 * (sl - out_row[i] * co) : 0;
        }
    }
    for (int e = lane; e < 16 * PI; e += 64) tin[e] = 0.0f;   // the K tail columns stay zero for the whole launch

    // Pipeline per wave (tile i on the matrix cores): ids of tile i+3 and rows of tile i+2 in flight, tile i+1 waiting in
    // registers for its turn through LDS, the operands of tile i in registers.  Two register sets each, rotated by
    // unrolling.  Issue order inside a step is ids before rows, so waiting for an id never waits for rows issued after it.
    struct Stage {
        v4f x[KJ];
    };
    struct Ids {
        int r[KJ];
    };
    Stage S0, S1;
    Ids I0, I1;
    auto tile_ids = [&](int i) { return trows + (int64_t)(t0 + (i < cnt ? i : cnt - 1) * nw) * 16; };   // past the end: the last tile again
    auto issue_ids = [&](Ids &id, int i) {
        const int32_t *tr = tile_ids(i);
#pragma unroll
        for (int k = 0; k < KJ; ++k) id.r[k] = tr[in_row[k]];
    };
    auto issue_rows = [&](Stage &s, const Ids &id) {
#pragma unroll
        for (int k = 0; k < KJ; ++k) {
#if DUV_VARIANT & 2   // timing-only: no row loads
            s.x[k] = v4f{1.0f, 2.0f, 3.0f, (float)id.r[k]};
#else
            s.x[k] = *reinterpret_cast<const v4f *>(X + (int64_t)id.r[k] * K + in_col[k]);
#endif
        }
    };
    v4f xf[KJ];
    auto turn_in = [&](const Stage &s) {     // coalesced registers -> LDS tile -> operand layout (lane (v, q): a[v, 16 j + 4 q ..])
#pragma unroll
        for (int k = 0; k < KJ; ++k) *reinterpret_cast<v4f *>(tin + in_row[k] * PI + in_col[k]) = s.x[k];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < KJ; ++j) xf[j] = *reinterpret_cast<const v4f *>(tin + n * PI + 16 * j + 4 * q);
        asm volatile("" ::: "memory");
    };
    auto step = [&](Stage &next, Stage &fill, Ids &id_fill, Ids &id_next, int i) {
        // the vertices of THIS tile in the store numbering, then the ids of tile i+3, then the rows of tile i+2
        int orow[OT];
        {
            const int32_t *tr = tile_ids(i);
#pragma unroll
            for (int k = 0; k < OT; ++k) orow[k] = tr[out_row[k]];
        }
        issue_ids(id_next, i + 3);
        issue_rows(fill, id_fill);
        // OT independent accumulation chains, interleaved so consecutive MFMAs never wait on each other
        v4f accs[OT];
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) accs[ot] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
#if DUV_VARIANT & 1   // timing-only: no matrix work
#pragma unroll
        for (int j = 0; j < KJ; ++j)
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) accs[ot] = accs[ot] + xf[j];
#else
#pragma unroll
        for (int j = 0; j < KJ; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int ot = 0; ot < OT; ++ot)
                    accs[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wf[ot][j][c], xf[j][c], accs[ot], 0, 0, 0);
#endif
        switch ((DUV_VARIANT & 8) ? 99 : act) {   // wave-uniform: one scalar branch per tile, each arm straight-line
        case 99: break;
        case ATHENA_MP_ACT_SIGMOID: {
            const float cs = -1.4426950408889634f * inv;     // 1 / (1 + 2^(-log2(e) x / d)): the divisor rides in the constant
#pragma unroll
            for (int ot = 0; ot < OT; ++ot)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    accs[ot][c] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(accs[ot][c] * cs));
            break;
        }
        case ATHENA_MP_ACT_RELU:
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                accs[ot] = div4(accs[ot], d, inv);
#pragma unroll
                for (int c = 0; c < 4; ++c) accs[ot][c] = accs[ot][c] > 0.0f ? accs[ot][c] : 0.0f;
            }
            break;
        case ATHENA_MP_ACT_TANH:
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                accs[ot] = div4(accs[ot], d, inv);
#pragma unroll
                for (int c = 0; c < 4; ++c) accs[ot][c] = tanhf(accs[ot][c]);
            }
            break;
        default:
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) accs[ot] = div4(accs[ot], d, inv);
            break;
        }
        // results: operand layout -> LDS tile -> coalesced rows (lane l of store i: chunk 64 i + l)
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) *reinterpret_cast<v4f *>(tout + n * PO + 16 * ot + 4 * q) = accs[ot];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < OT; ++k) {
            const v4f y = *reinterpret_cast<const v4f *>(tout + out_row[k] * PO + out_col[k]);
#if DUV_VARIANT & 4   // timing-only: no stores (every value stays live)
            if (y[0] + y[1] + y[2] + y[3] == 12345.678f) Y[gw] = y[0];
#else
            *reinterpret_cast<v4f *>(Y + (int64_t)orow[k] * NO + out_col[k]) = y;
#endif
        }
        asm volatile("" ::: "memory");
        turn_in(next);                // tile i+1 (its rows were issued one step ago) takes the operand registers
    };
    issue_ids(I0, 0);
    issue_ids(I1, 1);
    issue_rows(S0, I0);               // tile 0
    issue_ids(I0, 2);
    issue_rows(S1, I1);               // tile 1
    turn_in(S0);
    // step i: operands of tile i in xf, `next` holds tile i+1, rows of tile i+2 go into `fill` (ids in id_fill), the ids
    // of tile i+3 into id_next
    int i = 0;
    for (; i + 2 <= cnt; i += 2) {    // the body: no per-lane control flow
        step(S1, S0, I0, I1, i);
        step(S0, S1, I1, I0, i + 1);
    }
    if (i < cnt) step(S1, S0, I0, I1, i);
}

// slab[workgroup][i*Fo + o] = (sum over the workgroup's tiles (all of one bucket b) of a[v,i] g[v,o]) / d.
// Round 3: same cure -- rows arrive in the coalesced chunk numbering and are written straight to their place in the
// wave's LDS tile (the turn the contraction over vertices needs anyway), no load sits under a lane branch (tiles past
// the end are clamped and contribute zero gradient rows, like the padding slots of a bucket's last tile), the wavefront
// fences around the LDS turn are gone (a wave's LDS operations execute in order; the fences made the compiler drain the
// row prefetch with vmcnt(0) every tile) and the divisor is applied ONCE to the finished sums.
template <int IT, int OT>
__global__ __launch_bounds__(256) void duv_dw_any_kernel(BucketSplit sp, const int32_t *__restrict__ trows,
                                                     const float *__restrict__ A, int Fi,
                                                     const float *__restrict__ G, int Fo, float *__restrict__ slabs)
{
    constexpr int AP = 16 * IT + 4, GP = 16 * OT + 4, FOP = 16 * OT;
    constexpr int kTurn = 4 * 16 * (AP + GP), kRed = 16 * IT * FOP;
    __shared__ __attribute__((aligned(16))) float buf[kTurn > kRed ? kTurn : kRed];
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    float *al = buf + wave * 16 * (AP + GP), *gl = al + 16 * AP;
    int b = 0;
    while ((int)blockIdx.x >= sp.unit_off[b + 1]) ++b;
    const int nwg = sp.unit_off[b + 1] - sp.unit_off[b];
    const int stride = 4 * nwg;
    const int t0 = sp.tile_off[b] + 4 * ((int)blockIdx.x - sp.unit_off[b]) + wave, t1 = sp.tile_off[b + 1];
    const int cnt = t0 < t1 ? (t1 - t0 + stride - 1) / stride : 0;
    const float d = (float)(b + 1);
    int a_row[IT], a_col[IT], g_row[OT], g_col[OT];
    {
        const int ca = Fi >> 2, cg = Fo >> 2;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int sl = 64 * i + lane, ok = sl < 16 * ca;
            a_row[i] = ok ? sl / ca : 0;
            a_col[i] = ok ? 4 * (sl - a_row[i] * ca) : 0;
        }
#pragma unroll
        for (int i = 0; i < OT; ++i) {
            const int sl = 64 * i + lane, ok = sl < 16 * cg;
            g_row[i] = ok ? sl / cg : 0;
            g_col[i] = ok ? 4 * (sl - g_row[i] * cg) : 0;
        }
    }
    for (int e = lane; e < 16 * (AP + GP); e += 64) al[e] = 0.0f;    // tail columns stay zero

    v4f acc[IT][OT];
#pragma unroll
    for (int i = 0; i < IT; ++i)
#pragma unroll
        for (int o = 0; o < OT; ++o) acc[i][o] = v4f{0.0f, 0.0f, 0.0f, 0.0f};

    // ids two tiles ahead, rows one tile ahead
    struct Ids {
        int a[IT], g[OT];
    };
    auto issue_ids = [&](Ids &id, int i) {
        const int32_t *tr = trows + (int64_t)(t0 + (i < cnt ? i : cnt - 1) * stride) * 16;
#pragma unroll
        for (int k = 0; k < IT; ++k) id.a[k] = tr[a_row[k]];
#pragma unroll
        for (int k = 0; k < OT; ++k) id.g[k] = tr[g_row[k]];
    };
    v4f an[IT], gn[OT];
    auto issue_rows = [&](const Ids &id) {
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int r = id.a[k] ^ (id.a[k] >> 31);
            an[k] = *reinterpret_cast<const v4f *>(A + (int64_t)r * Fi + a_col[k]);
        }
#pragma unroll
        for (int k = 0; k < OT; ++k) {
            const int r = id.g[k] ^ (id.g[k] >> 31);
            gn[k] = *reinterpret_cast<const v4f *>(G + (int64_t)r * Fo + g_col[k]);
        }
    };
    if (cnt > 0) {
        Ids I0, I1;
        issue_ids(I0, 0);
        issue_ids(I1, 1);
        issue_rows(I0);
        auto body = [&](Ids &cur, Ids &nxt, int i) {
            // this tile's rows leave the prefetch registers for their place in the LDS tile; a padding slot (id < 0)
            // contributes a zero gradient row
            const v4f zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < IT; ++k) *reinterpret_cast<v4f *>(al + a_row[k] * AP + a_col[k]) = an[k];
#pragma unroll
            for (int k = 0; k < OT; ++k) *reinterpret_cast<v4f *>(gl + g_row[k] * GP + g_col[k]) = cur.g[k] >= 0 ? gn[k] : zero;
            issue_ids(cur, i + 2);             // ids of tile i+2 ...
            issue_rows(nxt);                   // ... and the rows of tile i+1 (tile cnt-1 again at the end: loaded, never used)
            asm volatile("" ::: "memory");     // compiler: keep the LDS writes above the reads (the hardware does)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float aa[IT], bb[OT];
#pragma unroll
                for (int ii = 0; ii < IT; ++ii) aa[ii] = al[(4 * q + r) * AP + 16 * ii + n];
#pragma unroll
                for (int o = 0; o < OT; ++o) bb[o] = gl[(4 * q + r) * GP + 16 * o + n];
#pragma unroll
                for (int ii = 0; ii < IT; ++ii)
#pragma unroll
                    for (int o = 0; o < OT; ++o)
                        acc[ii][o] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[ii], bb[o], acc[ii][o], 0, 0, 0);
            }
            asm volatile("" ::: "memory");     // and the next tile's writes below these reads
        };
        int i = 0;
        for (; i + 2 <= cnt; i += 2) {
            body(I0, I1, i);
            body(I1, I0, i + 1);
        }
        if (i < cnt) body(I0, I1, i);
    }
    // acc[i][o][r] = dW(i = 16 i + 4q + r, o = 16 o + n); waves added in fixed order
    __syncthreads();
    for (int p = 0; p < 4; ++p) {
        if (wave == p) {
#pragma unroll
            for (int i = 0; i < IT; ++i)
#pragma unroll
                for (int o = 0; o < OT; ++o)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float *dst = buf + (16 * i + 4 * q + r) * FOP + 16 * o + n;
                        *dst = (p == 0 ? 0.0f : *dst) + acc[i][o][r];
                    }
        }
        __syncthreads();
    }
    float *slab = slabs + (size_t)blockIdx.x * Fi * Fo;
    for (int t = threadIdx.x; t < Fi * Fo; t += 256) {
        const int i = t / Fo, o = t - i * Fo;
        slab[t] = buf[i * FOP + o] / d;        // the bucket index as divisor, once per sum (IEEE division)
    }
}

// ---- the same two kernels for rows of 64 .. 96 floats on both sides (BASELINE configs[2]: 72 -> 64) ---------------------
// The chunk numbering above costs ~45 registers of per-lane bookkeeping (row, column and LDS address of every load and
// store), which at 80 weight fragments pushes a wave past 256 registers -- one wave per SIMD.  With at least 16 chunks
// per row the numbering has structure the compiler can fold into immediates: load i (i < 4) of a tile takes rows
// 4 i + (lane >> 4), columns 4 (lane & 15) .. +3 -- sixteen lanes read 256 contiguous bytes of one row -- and the 0 .. 8
// chunks beyond column 64 go in one or two tail loads numbered as before.  A lane's four row ids come as ONE 16-byte load
// from the per-tile id block stored transposed (duvenaud_buckets, copies 2 / 3).
template <int F>   // fragments of 16 columns: 4, 5 or 6
struct WideTail {
    static constexpr int T = F - 4;               // tail loads
    int row[T > 0 ? T : 1], col[T > 0 ? T : 1];   // per lane: tile row and first column of its chunk in tail load u
    __device__ __forceinline__ void init(int lane, int width)
    {
        const int ct = (width >> 2) - 16;         // chunks per row beyond column 64
#pragma unroll
        for (int u = 0; u < T; ++u) {
            const int sl = 64 * u + lane;
            const bool ok = sl < 16 * ct;
            row[u] = ok ? sl / ct : 0;
            col[u] = ok ? 64 + 4 * (sl - row[u] * ct) : 0;   // past the end: chunk 0 of row 0 again (same value, same place)
        }
    }
};
typedef int v4i __attribute__((ext_vector_type(4)));

// RO: the Duvenaud readout of the same time step rides in the epilogue (athena_duvenaud_msgpass_layer.f90:838-855):
//     p[v,:] = softmax_over_outputs(R z[v,:])  with z = the activated rows this kernel has just produced.
// The activated tile sits in the accumulators in exactly the B-operand layout of the logits product (lane (v, q) holds
// z[v, 16 ot + 4 q + c], i.e. k = 16 ot + 4 q + c), so logits^T[o, v] costs OT x 4 more MFMAs per tile with R's fragments read
// from LDS, the softmax is two cross-lane steps (readout.hip), and z is not read back from HBM by a readout launch
// (600 MB at configs[2]).  p rows are 4 O bytes: written with bounds-checked buffer stores, lanes beyond O pointed past it.
// (Two waves per SIMD wherever 256 registers hold the shape.  96 OUTPUTS (24 accumulators) get one wave per SIMD and the whole
// register file instead of spills inside the tile loop (scripts/isa_lint.py R2); 96 INPUTS with the readout epilogue keep two
// waves and their 44 B of spills -- measured faster that way, 0.488 against 0.545 ms, while the fused reverse kernel of the same
// shape gains 26 % from the whole file, 0.961 -> 0.707 ms: profiles/r05_duv_96wide_ab.txt.)
// TC: MFMAs of the LAST 16-column fragment of the input.  4 = the plain mapping k = 16 j + 4 q + c.  A tail of 4 TC < 16 columns
// (K = 72: TC = 2) takes k = 16 (KJ - 1) + TC q + c, c < TC instead -- every k slot of its TC MFMAs is a real column, where the plain
// mapping spends 4 MFMAs on the tail with the slots of q >= TC zero (K = 72: 18 k-steps per output fragment instead of 20).
template <int KJ, int OT, bool RO, bool XS = false, int TC = 4>
__global__ __launch_bounds__(256, (OT > 5 ? 1 : 2)) void duv_rows_wide_kernel(BucketSplit sp, const int32_t *__restrict__ trows,
                                                            const int32_t *__restrict__ trows_t,
                                                            const float *__restrict__ X, int K,
                                                            const float *__restrict__ W, int64_t wb, int so, int sk,
                                                            float *__restrict__ Y, int NO, int act,
                                                            const float *__restrict__ R, int O, float *__restrict__ P,
                                                            uint32_t p_bytes, const float *__restrict__ XT)
{
    // XS: the input rows arrive SPLIT -- columns 0 .. 63 from X as rows of 64 floats, the columns beyond from XT as rows of
    // K - 64 floats (the Duvenaud layer's a = [neighbour sum of x | neighbour sum of e]: the edge part is the same at every time
    // step of a layer and is gathered once; round 5).  An instantiation of its own: the packed kernels keep their registers.
    constexpr int PI = 16 * KJ + 4, PO = 16 * OT + 4, TI = KJ - 4, TO = OT - 4;
    const int64_t xpm = XS ? 64 : K, xpt = XS ? K - 64 : K;
    const float *xt = XS ? XT - 64 : X;
    __shared__ __attribute__((aligned(16))) float lds[4 * 16 * (PI + PO) + (RO ? 64 * 4 * OT : 4)];
    float *rl = lds + 4 * 16 * (PI + PO);           // R fragments: [ot][lane][c] = R(o = lane & 15, k = 16 ot + 4 (lane >> 4) + c)
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    float *tin = lds + wave * 16 * (PI + PO), *tout = tin + 16 * PI;
    const int gw = blockIdx.x * 4 + wave;
    if (gw >= sp.unit_off[sp.n_buckets]) return;
    int b = 0;
    while (gw >= sp.unit_off[b + 1]) ++b;
    const int nw = sp.unit_off[b + 1] - sp.unit_off[b];
    const int t0 = sp.tile_off[b] + (gw - sp.unit_off[b]), t1 = sp.tile_off[b + 1];
    if (t0 >= t1) return;
    const int cnt = (t1 - t0 + nw - 1) / nw;

    const float d = (float)(b + 1), inv = 1.0f / d;
    float Wf[OT][KJ][4];
    {
        const float *wd = W + (int64_t)b * wb;
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
#pragma unroll
            for (int j = 0; j < KJ; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int o = 16 * ot + n, k = (TC < 4 && j == KJ - 1) ? 16 * j + TC * q + c : 16 * j + 4 * q + c;
                    const bool ok = o < NO && k < K && (TC == 4 || j < KJ - 1 || c < TC);
                    const float v = wd[(int64_t)(ok ? o : 0) * so + (int64_t)(ok ? k : 0) * sk];
                    Wf[ot][j][c] = ok ? v : 0.0f;
                }
    }
    WideTail<KJ> ti;
    WideTail<OT> to;
    ti.init(lane, K);
    to.init(lane, NO);
    for (int e = lane; e < 16 * PI; e += 64) tin[e] = 0.0f;   // the K tail columns stay zero for the whole launch
    __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void *)P, 0, RO ? (int)p_bytes : 0, 0x00020000);
    if constexpr (RO) {   // params(T + t)%val(:,1) = R(O, F_v) column-major: flat o + O k.  Every wave fills the same values.
        for (int e = lane; e < 64 * OT; e += 64) {
            const int ot = e >> 6;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = 16 * ot + 4 * q + c;
                rl[(ot * 64 + lane) * 4 + c] = (n < O && k < NO) ? R[(int64_t)k * O + n] : 0.0f;
            }
        }
    }

    struct Stage {
        v4f f[4], t[TI > 0 ? TI : 1];
    };
    struct Ids {
        v4i f;
        int t[TI > 0 ? TI : 1];
    };
    Stage S0, S1;
    Ids I0, I1;
    auto tile_of = [&](int i) { return (int64_t)(t0 + (i < cnt ? i : cnt - 1) * nw) * 16; };   // past the end: the last tile again
    auto issue_ids = [&](Ids &id, int i) {
        const int64_t tb = tile_of(i);
        id.f = *reinterpret_cast<const v4i *>(trows_t + tb + 4 * q);
#pragma unroll
        for (int u = 0; u < TI; ++u) id.t[u] = trows[tb + ti.row[u]];
    };
    auto issue_rows = [&](Stage &s, const Ids &id) {
#if DUV_VARIANT & 2   // timing-only: no row loads
#pragma unroll
        for (int i = 0; i < 4; ++i) s.f[i] = v4f{1.0f, 2.0f, 3.0f, (float)id.f[i]};
#pragma unroll
        for (int u = 0; u < TI; ++u) s.t[u] = v4f{1.0f, 2.0f, 3.0f, (float)id.t[u]};
#else
#pragma unroll
        for (int i = 0; i < 4; ++i) s.f[i] = *reinterpret_cast<const v4f *>(X + (int64_t)id.f[i] * xpm + 4 * n);
#pragma unroll
        for (int u = 0; u < TI; ++u)   // (a lane without a tail chunk repeats chunk 0 of row 0 -- column 0: the main part)
            s.t[u] = XS ? *reinterpret_cast<const v4f *>((ti.col[u] >= 64 ? xt + (int64_t)id.t[u] * xpt : X + (int64_t)id.t[u] * xpm) + ti.col[u])
                        : *reinterpret_cast<const v4f *>(X + (int64_t)id.t[u] * K + ti.col[u]);
#endif
    };
    v4f xf[KJ];
    auto turn_in = [&](const Stage &s) {     // coalesced registers -> LDS tile -> operand layout (lane (v, q): a[v, 16 j + 4 q ..])
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<v4f *>(tin + (4 * i + q) * PI + 4 * n) = s.f[i];
#pragma unroll
        for (int u = 0; u < TI; ++u) *reinterpret_cast<v4f *>(tin + ti.row[u] * PI + ti.col[u]) = s.t[u];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            if (TC < 4 && j == KJ - 1) {          // the tail fragment: TC consecutive columns per lane
                const float *tp = tin + n * PI + 16 * j + TC * q;
                if constexpr (TC == 2) {
                    const v2f t2 = *reinterpret_cast<const v2f *>(tp);
                    xf[j] = v4f{t2[0], t2[1], 0.0f, 0.0f};
                } else {
                    xf[j] = v4f{tp[0], TC > 1 ? tp[1] : 0.0f, TC > 2 ? tp[2] : 0.0f, 0.0f};
                }
            } else {
                xf[j] = *reinterpret_cast<const v4f *>(tin + n * PI + 16 * j + 4 * q);
            }
        }
        asm volatile("" ::: "memory");
    };
    auto step = [&](Stage &next, Stage &fill, Ids &id_fill, Ids &id_next, int i) {
        // the vertices of THIS tile for the stores, then the ids of tile i+3, then the rows of tile i+2
        const int64_t tb = tile_of(i);
        const v4i orow = *reinterpret_cast<const v4i *>(trows_t + tb + 4 * q);
        int otail[TO > 0 ? TO : 1];
#pragma unroll
        for (int u = 0; u < TO; ++u) otail[u] = trows[tb + to.row[u]];
        issue_ids(id_next, i + 3);
        issue_rows(fill, id_fill);
        v4f accs[OT];
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) accs[ot] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
#if DUV_VARIANT & 1   // timing-only: no matrix work
#pragma unroll
        for (int j = 0; j < KJ; ++j)
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) accs[ot] = accs[ot] + xf[j];
#else
#pragma unroll
        for (int j = 0; j < KJ; ++j)
#pragma unroll
            for (int c = 0; c < ((TC < 4 && j == KJ - 1) ? TC : 4); ++c)
#pragma unroll
                for (int ot = 0; ot < OT; ++ot)
                    accs[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wf[ot][j][c], xf[j][c], accs[ot], 0, 0, 0);
#endif
        switch ((DUV_VARIANT & 8) ? 99 : act) {   // wave-uniform: one scalar branch per tile, each arm straight-line
        case 99: break;
        case ATHENA_MP_ACT_SIGMOID: {
            const float cs = -1.4426950408889634f * inv;     // 1 / (1 + 2^(-log2(e) x / d)): the divisor rides in the constant
#pragma unroll
            for (int ot = 0; ot < OT; ++ot)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    accs[ot][c] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(accs[ot][c] * cs));
            break;
        }
        case ATHENA_MP_ACT_RELU:
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                accs[ot] = div4(accs[ot], d, inv);
#pragma unroll
                for (int c = 0; c < 4; ++c) accs[ot][c] = accs[ot][c] > 0.0f ? accs[ot][c] : 0.0f;
            }
            break;
        case ATHENA_MP_ACT_TANH:
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                accs[ot] = div4(accs[ot], d, inv);
#pragma unroll
                for (int c = 0; c < 4; ++c) accs[ot][c] = tanhf(accs[ot][c]);
            }
            break;
        default:
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) accs[ot] = div4(accs[ot], d, inv);
            break;
        }
        if constexpr (RO) {
            v4f lg = {0.0f, 0.0f, 0.0f, 0.0f};           // lane (v, q): logits[v, 4 q + r]
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                const v4f rf = *reinterpret_cast<const v4f *>(rl + (ot * 64 + lane) * 4);
#pragma unroll
                for (int c = 0; c < 4; ++c) lg = __builtin_amdgcn_mfma_f32_16x16x4f32(rf[c], accs[ot][c], lg, 0, 0, 0);
            }
            // softmax over the O outputs of a vertex: hardware exp2 / rcp (1 ulp each; the route is held to 1e-5), two
            // cross-lane steps over the four lanes of a vertex
            const int nv = O - 4 * q;                    // valid outputs in this lane's four: <= 0 none, >= 4 all
            float m = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, r < nv ? lg[r] : -INFINITY);
            m = fmaxf(m, __shfl_xor(m, 16));
            m = fmaxf(m, __shfl_xor(m, 32));
            float e[4], sum = 0.0f;                      // scalars: __builtin_bit_cast of a vector ELEMENT reads element 0
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e[r] = r < nv ? __builtin_amdgcn_exp2f((lg[r] - m) * 1.4426950408889634f) : 0.0f;
                sum = sum + e[r];
            }
            sum = sum + __shfl_xor(sum, 16);
            sum = sum + __shfl_xor(sum, 32);
            const float rs = __builtin_amdgcn_rcpf(sum);
            const int p0 = __builtin_bit_cast(int, e[0] * rs), p1 = __builtin_bit_cast(int, e[1] * rs),
                      p2 = __builtin_bit_cast(int, e[2] * rs), p3 = __builtin_bit_cast(int, e[3] * rs);
            // p[v, 4 q .. 4 q + nv): as few, as wide stores as the count allows -- 16 bytes where all four outputs exist,
            // then 8, then 4; a lane with nothing to write in a pass points past the buffer (the store is dropped)
            const uint32_t base = ((uint32_t)trows[tb + n] * (uint32_t)O + 4u * (uint32_t)q) * 4u, dead = 0xFFFFFFF0u;
            typedef int v2i_ __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b128(v4i{p0, p1, p2, p3}, prs, (int)(nv >= 4 ? base : dead), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(v2i_{p0, p1}, prs, (int)((nv == 2 || nv == 3) ? base : dead), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(nv == 3 ? p2 : p0, prs, (int)(nv == 3 ? base + 8u : nv == 1 ? base : dead), 0, 0);
        }
        // results: operand layout -> LDS tile -> coalesced rows
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) *reinterpret_cast<v4f *>(tout + n * PO + 16 * ot + 4 * q) = accs[ot];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 4 + TO; ++k) {
            const v4f y = k < 4 ? *reinterpret_cast<const v4f *>(tout + (4 * k + q) * PO + 4 * n)
                                : *reinterpret_cast<const v4f *>(tout + to.row[k < 4 ? 0 : k - 4] * PO + to.col[k < 4 ? 0 : k - 4]);
            float *dst = k < 4 ? Y + (int64_t)orow[k < 4 ? k : 0] * NO + 4 * n
                               : Y + (int64_t)otail[k < 4 ? 0 : k - 4] * NO + to.col[k < 4 ? 0 : k - 4];
#if DUV_VARIANT & 4   // timing-only: no stores (every value stays live)
            if (y[0] + y[1] + y[2] + y[3] == 12345.678f) Y[gw] = y[0];
#else
            *reinterpret_cast<v4f *>(dst) = y;
#endif
        }
        asm volatile("" ::: "memory");
        turn_in(next);                // tile i+1 (its rows were issued one step ago) takes the operand registers
    };
    issue_ids(I0, 0);
    issue_ids(I1, 1);
    issue_rows(S0, I0);               // tile 0
    issue_ids(I0, 2);
    issue_rows(S1, I1);               // tile 1
    turn_in(S0);
    int i = 0;
    for (; i + 2 <= cnt; i += 2) {    // the body: no per-lane control flow
        step(S1, S0, I0, I1, i);
        step(S0, S1, I1, I0, i + 1);
    }
    if (i < cnt) step(S1, S0, I0, I1, i);
}

template <int IT, int OT>
__global__ __launch_bounds__(256, 2) void duv_dw_wide_kernel(BucketSplit sp, const int32_t *__restrict__ trows,
                                                          const int32_t *__restrict__ trows_t,
                                                          const float *__restrict__ A, int Fi,
                                                          const float *__restrict__ G, int Fo, float *__restrict__ slabs)
{
    constexpr int AP = 16 * IT + 4, GP = 16 * OT + 4, FOP = 16 * OT, TA = IT - 4, TG = OT - 4;
    constexpr int kTurn = 4 * 16 * (AP + GP), kRed = 16 * IT * FOP;
    __shared__ __attribute__((aligned(16))) float buf[kTurn > kRed ? kTurn : kRed];
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    float *al = buf + wave * 16 * (AP + GP), *gl = al + 16 * AP;
    int b = 0;
    while ((int)blockIdx.x >= sp.unit_off[b + 1]) ++b;
    const int nwg = sp.unit_off[b + 1] - sp.unit_off[b];
    const int stride = 4 * nwg;
    const int t0 = sp.tile_off[b] + 4 * ((int)blockIdx.x - sp.unit_off[b]) + wave, t1 = sp.tile_off[b + 1];
    const int cnt = t0 < t1 ? (t1 - t0 + stride - 1) / stride : 0;
    const float d = (float)(b + 1);
    WideTail<IT> ta;
    WideTail<OT> tg;
    ta.init(lane, Fi);
    tg.init(lane, Fo);
    for (int e = lane; e < 16 * (AP + GP); e += 64) al[e] = 0.0f;    // tail columns stay zero

    v4f acc[IT][OT];
#pragma unroll
    for (int i = 0; i < IT; ++i)
#pragma unroll
        for (int o = 0; o < OT; ++o) acc[i][o] = v4f{0.0f, 0.0f, 0.0f, 0.0f};

    struct Ids {
        v4i f;                                          // rows 4 i + q of the tile, padding slots negative
        int a[TA > 0 ? TA : 1], g[TG > 0 ? TG : 1];
    };
    auto issue_ids = [&](Ids &id, int i) {
        const int64_t tb = (int64_t)(t0 + (i < cnt ? i : cnt - 1) * stride) * 16;
        id.f = *reinterpret_cast<const v4i *>(trows_t + tb + 4 * q);
#pragma unroll
        for (int u = 0; u < TA; ++u) id.a[u] = trows[tb + ta.row[u]];
#pragma unroll
        for (int u = 0; u < TG; ++u) id.g[u] = trows[tb + tg.row[u]];
    };
    v4f an[4 + (TA > 0 ? TA : 0) + 1], gn[4 + (TG > 0 ? TG : 0) + 1];
    auto issue_rows = [&](const Ids &id) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = id.f[i] ^ (id.f[i] >> 31);
            an[i] = *reinterpret_cast<const v4f *>(A + (int64_t)r * Fi + 4 * n);
            gn[i] = *reinterpret_cast<const v4f *>(G + (int64_t)r * Fo + 4 * n);
        }
#pragma unroll
        for (int u = 0; u < TA; ++u) {
            const int r = id.a[u] ^ (id.a[u] >> 31);
            an[4 + u] = *reinterpret_cast<const v4f *>(A + (int64_t)r * Fi + ta.col[u]);
        }
#pragma unroll
        for (int u = 0; u < TG; ++u) {
            const int r = id.g[u] ^ (id.g[u] >> 31);
            gn[4 + u] = *reinterpret_cast<const v4f *>(G + (int64_t)r * Fo + tg.col[u]);
        }
    };
    if (cnt > 0) {
        Ids I0, I1;
        issue_ids(I0, 0);
        issue_ids(I1, 1);
        issue_rows(I0);
        auto body = [&](Ids &cur, Ids &nxt, int i) {
            // this tile's rows leave the prefetch registers for their place in the LDS tile; a padding slot (id < 0)
            // contributes a zero gradient row
            const v4f zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                *reinterpret_cast<v4f *>(al + (4 * k + q) * AP + 4 * n) = an[k];
                *reinterpret_cast<v4f *>(gl + (4 * k + q) * GP + 4 * n) = cur.f[k] >= 0 ? gn[k] : zero;
            }
#pragma unroll
            for (int u = 0; u < TA; ++u) *reinterpret_cast<v4f *>(al + ta.row[u] * AP + ta.col[u]) = an[4 + u];
#pragma unroll
            for (int u = 0; u < TG; ++u) *reinterpret_cast<v4f *>(gl + tg.row[u] * GP + tg.col[u]) = cur.g[u] >= 0 ? gn[4 + u] : zero;
            issue_ids(cur, i + 2);             // ids of tile i+2 ...
            issue_rows(nxt);                   // ... and the rows of tile i+1 (tile cnt-1 again at the end: loaded, never used)
            asm volatile("" ::: "memory");     // compiler: keep the LDS writes above the reads (the hardware does)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float aa[IT], bb[OT];
#pragma unroll
                for (int ii = 0; ii < IT; ++ii) aa[ii] = al[(4 * q + r) * AP + 16 * ii + n];
#pragma unroll
                for (int o = 0; o < OT; ++o) bb[o] = gl[(4 * q + r) * GP + 16 * o + n];
#pragma unroll
                for (int ii = 0; ii < IT; ++ii)
#pragma unroll
                    for (int o = 0; o < OT; ++o)
                        acc[ii][o] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[ii], bb[o], acc[ii][o], 0, 0, 0);
            }
            asm volatile("" ::: "memory");     // and the next tile's writes below these reads
        };
        int i = 0;
        for (; i + 2 <= cnt; i += 2) {
            body(I0, I1, i);
            body(I1, I0, i + 1);
        }
        if (i < cnt) body(I0, I1, i);
    }
    __syncthreads();
    for (int p = 0; p < 4; ++p) {
        if (wave == p) {
#pragma unroll
            for (int i = 0; i < IT; ++i)
#pragma unroll
                for (int o = 0; o < OT; ++o)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float *dst = buf + (16 * i + 4 * q + r) * FOP + 16 * o + n;
                        *dst = (p == 0 ? 0.0f : *dst) + acc[i][o][r];
                    }
        }
        __syncthreads();
    }
    float *slab = slabs + (size_t)blockIdx.x * Fi * Fo;
    for (int t = threadIdx.x; t < Fi * Fo; t += 256) {
        const int i = t / Fo, o = t - i * Fo;
        slab[t] = buf[i * FOP + o] / d;        // the bucket index as divisor, once per sum (IEEE division)
    }
}

// ---- both reverse products of the update from ONE pass over the gradient rows --------------------------------------------
//   da[v,i] = (sum_o g[v,o] W_d(o,i)) / d          get_partial_duvenaud_update_val        (..._sub_duvenaud.f90:284-324)
//   dW_d(o,i) += g[v,o] a[v,i] / d                 get_partial_duvenaud_update_weight_val (:326-368)
// The two launches above each read g (600 MB at configs[2]); the pair is bound by the bytes it moves, so one launch that
// reads g and a once and writes da does 1.95 GB instead of 2.55 GB.  A workgroup serves one bucket: W_d sits in LDS (the A
// operand of the da product is one 16-byte LDS read per four MFMAs -- 80 weight fragments in registers beside 80 weight-
// gradient accumulators would leave one wave per SIMD), the rows of a tile arrive in the coalesced numbering and are turned
// in the wave's LDS tiles as in duv_dw_wide_kernel, and da leaves through the tile that held a.  Padding slots repeat the
// tile's first vertex: their a row is zeroed (no weight gradient), their g row is not, so their da row is that vertex's
// own -- a benign duplicate store.
template <int IT, int OT>
__global__ __launch_bounds__(256, (IT * OT > 20 ? 1 : 2)) void duv_bwd_wide_kernel(BucketSplit sp, const int32_t *__restrict__ trows,
                                                              const int32_t *__restrict__ trows_t,
                                                              const float *__restrict__ A, int Fi,
                                                              const float *__restrict__ G, int Fo,
                                                              const float *__restrict__ W, float *__restrict__ DA,
                                                              float *__restrict__ slabs, float *__restrict__ DAT)
{
    // DAT != null: da leaves SPLIT (athena_mp_duvenaud_update_bwd_split) -- columns 0 .. 63 to DA as rows of 64 floats (256-byte
    // rows: whole cache lines whatever the order the buckets write them in), the columns beyond to DAT as rows of Fi - 64
    // floats: the propagate reverse then gathers the edge part from a dense [N, F_e] array instead of 32-byte slivers of
    // 288-byte rows (profiles/r05_c3_split_da_ab.txt: 0.075 -> 0.037 ms, the three reverse launches 0.936 -> 0.872 ms)
    constexpr int AP = 16 * IT + 4, GP = 16 * OT + 4, FOP = 16 * OT, WP = 16 * OT + 4, TA = IT - 4, TG = OT - 4;
    constexpr int kW = 16 * IT * WP, kTurn = 4 * 16 * (AP + GP), kRed = 16 * IT * FOP;
    // (probe, round 5: padded to ONE workgroup per CU = one wave per SIMD this kernel takes 0.589 instead of 0.532 ms at configs[2] --
    // it is bound by what a SIMD issues per tile (~8 700 cycles, 5 120 of them MFMA), not by latency; folding the readout's reverse
    // in would add ~1 500 issue cycles per tile at one wave per SIMD: ~0.70 ms against 0.53 + 0.29 today -- sized, not built)
    __shared__ __attribute__((aligned(16))) float buf[kW + (kTurn > kRed ? kTurn : kRed)];
    float *wl = buf, *red = buf + kW;
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    float *al = red + wave * 16 * (AP + GP), *gl = al + 16 * AP;
    int b = 0;
    while ((int)blockIdx.x >= sp.unit_off[b + 1]) ++b;
    const int nwg = sp.unit_off[b + 1] - sp.unit_off[b];
    const int stride = 4 * nwg;
    const int t0 = sp.tile_off[b] + 4 * ((int)blockIdx.x - sp.unit_off[b]) + wave, t1 = sp.tile_off[b + 1];
    const int cnt = t0 < t1 ? (t1 - t0 + stride - 1) / stride : 0;
    const float d = (float)(b + 1), inv = 1.0f / d;
    // W_d(o,i) flat o + Fo i == [i][o] rows of Fo floats -> LDS rows of pitch WP, zero beyond Fi / Fo
    {
        // every load first, then every LDS write: written as load-then-store per element the loop took one memory latency per
        // trip (10.9 us of a 530 us launch with all 512 workgroups in it at once, profiles/r05_c3_bwd_timeline.txt)
        const float *wd = W + (int64_t)b * Fi * Fo;
        constexpr int kTrips = (16 * IT * WP + 255) / 256;
        float wv[kTrips];
#pragma unroll
        for (int t = 0; t < kTrips; ++t) {
            const int e = threadIdx.x + 256 * t, i = e / WP, o = e - i * WP;
            wv[t] = (i < Fi && o < Fo) ? wd[(int64_t)i * Fo + o] : 0.0f;
        }
#pragma unroll
        for (int t = 0; t < kTrips; ++t) {
            const int e = threadIdx.x + 256 * t;
            if (e < 16 * IT * WP) wl[e] = wv[t];
        }
    }
    WideTail<IT> ta;
    WideTail<OT> tg;
    ta.init(lane, Fi);
    tg.init(lane, Fo);
    for (int e = lane; e < 16 * (AP + GP); e += 64) al[e] = 0.0f;    // tail columns stay zero
    __syncthreads();

    v4f acc[IT][OT];
#pragma unroll
    for (int i = 0; i < IT; ++i)
#pragma unroll
        for (int o = 0; o < OT; ++o) acc[i][o] = v4f{0.0f, 0.0f, 0.0f, 0.0f};

    struct Ids {
        v4i f;                                          // rows 4 i + q of the tile, padding slots negative
        int a[TA > 0 ? TA : 1], g[TG > 0 ? TG : 1];
    };
    auto issue_ids = [&](Ids &id, int i) {
        const int64_t tb = (int64_t)(t0 + (i < cnt ? i : cnt - 1) * stride) * 16;
        id.f = *reinterpret_cast<const v4i *>(trows_t + tb + 4 * q);
#pragma unroll
        for (int u = 0; u < TA; ++u) id.a[u] = trows[tb + ta.row[u]];
#pragma unroll
        for (int u = 0; u < TG; ++u) id.g[u] = trows[tb + tg.row[u]];
    };
    v4f an[4 + (TA > 0 ? TA : 0) + 1], gn[4 + (TG > 0 ? TG : 0) + 1];
    auto issue_rows = [&](const Ids &id) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = id.f[i] ^ (id.f[i] >> 31);
            an[i] = *reinterpret_cast<const v4f *>(A + (int64_t)r * Fi + 4 * n);
            gn[i] = *reinterpret_cast<const v4f *>(G + (int64_t)r * Fo + 4 * n);
        }
#pragma unroll
        for (int u = 0; u < TA; ++u) {
            const int r = id.a[u] ^ (id.a[u] >> 31);
            an[4 + u] = *reinterpret_cast<const v4f *>(A + (int64_t)r * Fi + ta.col[u]);
        }
#pragma unroll
        for (int u = 0; u < TG; ++u) {
            const int r = id.g[u] ^ (id.g[u] >> 31);
            gn[4 + u] = *reinterpret_cast<const v4f *>(G + (int64_t)r * Fo + tg.col[u]);
        }
    };
    if (cnt > 0) {
        Ids I0, I1;
        issue_ids(I0, 0);
        issue_ids(I1, 1);
        issue_rows(I0);
        auto body = [&](Ids &cur, Ids &nxt, int i) {
            const v4f zero = {0.0f, 0.0f, 0.0f, 0.0f};
            // store addresses of this tile's da rows (the rows of a it was loaded from), before the id registers move on
            int srow[4 + (TA > 0 ? TA : 1)];
#pragma unroll
            for (int k = 0; k < 4; ++k) srow[k] = cur.f[k] ^ (cur.f[k] >> 31);
#pragma unroll
            for (int u = 0; u < TA; ++u) srow[4 + u] = cur.a[u] ^ (cur.a[u] >> 31);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                *reinterpret_cast<v4f *>(al + (4 * k + q) * AP + 4 * n) = cur.f[k] >= 0 ? an[k] : zero;
                *reinterpret_cast<v4f *>(gl + (4 * k + q) * GP + 4 * n) = gn[k];
            }
#pragma unroll
            for (int u = 0; u < TA; ++u) *reinterpret_cast<v4f *>(al + ta.row[u] * AP + ta.col[u]) = cur.a[u] >= 0 ? an[4 + u] : zero;
#pragma unroll
            for (int u = 0; u < TG; ++u) *reinterpret_cast<v4f *>(gl + tg.row[u] * GP + tg.col[u]) = gn[4 + u];
            issue_ids(cur, i + 2);             // ids of tile i+2 ...
            issue_rows(nxt);                   // ... and the rows of tile i+1
            asm volatile("" ::: "memory");
            // weight gradient: contraction over the tile's vertices (both operands read turned)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float aa[IT], bb[OT];
#pragma unroll
                for (int ii = 0; ii < IT; ++ii) aa[ii] = al[(4 * q + r) * AP + 16 * ii + n];
#pragma unroll
                for (int o = 0; o < OT; ++o) bb[o] = gl[(4 * q + r) * GP + 16 * o + n];
#pragma unroll
                for (int ii = 0; ii < IT; ++ii)
#pragma unroll
                    for (int o = 0; o < OT; ++o)
                        acc[ii][o] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[ii], bb[o], acc[ii][o], 0, 0, 0);
            }
            // da^T[i, v] = sum_o W_d(o, i) g[v, o]: the vertex on the column axis, W fragments from LDS
            v4f xg[OT], da[IT];
#pragma unroll
            for (int j = 0; j < OT; ++j) xg[j] = *reinterpret_cast<const v4f *>(gl + n * GP + 16 * j + 4 * q);
#pragma unroll
            for (int it = 0; it < IT; ++it) da[it] = zero;
#pragma unroll
            for (int j = 0; j < OT; ++j) {
                v4f wf[IT];
#pragma unroll
                for (int it = 0; it < IT; ++it) wf[it] = *reinterpret_cast<const v4f *>(wl + (16 * it + n) * WP + 16 * j + 4 * q);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int it = 0; it < IT; ++it) da[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[it][c], xg[j][c], da[it], 0, 0, 0);
            }
            asm volatile("" ::: "memory");     // the turned reads of a are done: its tile now carries da out
#pragma unroll
            for (int it = 0; it < IT; ++it) *reinterpret_cast<v4f *>(al + n * AP + 16 * it + 4 * q) = div4(da[it], d, inv);
            asm volatile("" ::: "memory");
            const int64_t pm = DAT ? 64 : Fi, pt = DAT ? Fi - 64 : Fi;
            float *dat = DAT ? DAT - 64 : DA;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const v4f y = *reinterpret_cast<const v4f *>(al + (4 * k + q) * AP + 4 * n);
                *reinterpret_cast<v4f *>(DA + (int64_t)srow[k] * pm + 4 * n) = y;
            }
#pragma unroll
            for (int u = 0; u < TA; ++u) {
                const v4f y = *reinterpret_cast<const v4f *>(al + ta.row[u] * AP + ta.col[u]);
                // (a lane without a tail chunk repeats chunk 0 of row 0 -- column 0: that one belongs to the main part)
                float *dst = ta.col[u] >= 64 ? dat + (int64_t)srow[4 + u] * pt : DA + (int64_t)srow[4 + u] * pm;
                *reinterpret_cast<v4f *>(dst + ta.col[u]) = y;
            }
            asm volatile("" ::: "memory");
            // the tail columns of the a tile must read zero again for the next tile's weight gradient
            if (Fi < 16 * IT) {
#pragma unroll
                for (int it = IT - 1; it < IT; ++it)
                    if (16 * it + 4 * q >= Fi) *reinterpret_cast<v4f *>(al + n * AP + 16 * it + 4 * q) = zero;
            }
        };
        int i = 0;
        for (; i + 2 <= cnt; i += 2) {
            body(I0, I1, i);
            body(I1, I0, i + 1);
        }
        if (i < cnt) body(I0, I1, i);
    }
    __syncthreads();
    for (int p = 0; p < 4; ++p) {
        if (wave == p) {
#pragma unroll
            for (int i = 0; i < IT; ++i)
#pragma unroll
                for (int o = 0; o < OT; ++o)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float *dst = red + (16 * i + 4 * q + r) * FOP + 16 * o + n;
                        *dst = (p == 0 ? 0.0f : *dst) + acc[i][o][r];
                    }
        }
        __syncthreads();
    }
    float *slab = slabs + (size_t)blockIdx.x * Fi * Fo;
    for (int t = threadIdx.x; t < Fi * Fo; t += 256) {
        const int i = t / Fo, o = t - i * Fo;
        slab[t] = red[i * FOP + o] / d;
    }
}

// ---- one time step of the layer's reverse pass in ONE launch: the readout's reverse folded into the update's ---------------
//   dl[v,:]  = p[v,:] (gout[graph(v),:] - <gout[graph(v),:], p[v,:]>)        softmax over the outputs (athena_diffstruc_extd_sub.f90:309-313)
//   dR(o,f) += dl[v,o] z[v,f]                                                the readout matmul's weight partial
//   dc[v,f]  = act'(z[v,f]) (sum_o dl[v,o] R(o,f) + dz_next[v,f])            its input partial + the next step's dx, through the activation
//   da, dW_d   from dc as in duv_bwd_wide_kernel                             (athena_diffstruc_extd_sub_duvenaud.f90:284-368)
// readout_bwd_kernel + duv_bwd_wide_kernel move dc out (600 MB at configs[2]) and back in; the second one is bound by the matrix
// pipe with HBM to spare, the first by its bytes.  Here a tile's z / dz_next / p rows arrive beside its a rows (bucket order, the
// coalesced numbering), dc is formed in the z-fragment layout (lane (v, q): columns 16 ft + 4 q ..) and written straight into the
// wave's LDS tile where the update's reverse expects the gradient rows.  One wave per SIMD (the two accumulator sets, 80 + 16
// registers, beside four row streams): one workgroup per CU.  F_o = F_v = 64 only (OT = 4, no tail chunks on the z side).
template <int ACT>
__device__ __forceinline__ float duv_act_back(float y, float g)
{
    if constexpr (ACT == ATHENA_MP_ACT_RELU) return y > 0.0f ? g : 0.0f;
    if constexpr (ACT == ATHENA_MP_ACT_SIGMOID) return g * y * (1.0f - y);
    if constexpr (ACT == ATHENA_MP_ACT_TANH) return g * (1.0f - y * y);
    return g;
}

#ifndef DUV_RO_ATTR
#define DUV_RO_ATTR
#endif
template <int IT, int ACT, bool DIN>
__global__ __launch_bounds__(256, 1) DUV_RO_ATTR void duv_bwd_ro_kernel(BucketSplit sp, const int32_t *__restrict__ trows,
                                                            const int32_t *__restrict__ trows_t, const int32_t *__restrict__ tgid,
                                                            const float *__restrict__ A, int Fi, const float *__restrict__ Z,
                                                            const float *__restrict__ DZ, const float *__restrict__ P,
                                                            const float *__restrict__ GOUT, const float *__restrict__ R, int O,
                                                            const float *__restrict__ W, float *__restrict__ DA,
                                                            float *__restrict__ DAT, float *__restrict__ slabs,
                                                            float *__restrict__ rslabs, int acc_e,
                                                            const float *__restrict__ AT)
{
    // AT != null: a arrives split as in duv_rows_wide_kernel (A rows of 64 floats, AT rows of F_i - 64)
    const int64_t apm = AT ? 64 : Fi, apt = AT ? Fi - 64 : Fi;
    const float *at = AT ? AT - 64 : A;
    // acc_e: the edge part of da is ADDED to what DAT holds (the layer sums da_e over its time steps and scatters the sum to the
    // edge features once, instead of one scatter + one axpy per time step); the old values travel with the tile's rows
    constexpr int OT = 4, Fo = 64;
    constexpr int AP = 16 * IT + 4, GP = 16 * OT + 4, ZP = 16 * OT + 4, DP = 20, FOP = 16 * OT, WP = 16 * OT + 4, TA = IT - 4;
    constexpr int kW = 16 * IT * WP, kWave = 16 * (AP + GP + ZP + DP), kRed = 16 * IT * FOP, kRedR = 4 * Fo * 16;
    static_assert(4 * kWave >= kRed + kRedR, "the reduction images overlay the waves' tiles");
    __shared__ __attribute__((aligned(16))) float buf[kW + 4 * kWave];
    float *wl = buf, *red = buf + kW, *redr = red + kRed;
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    float *al = red + wave * kWave, *gl = al + 16 * AP, *zt = gl + 16 * GP, *dll = zt + 16 * ZP;
    int b = 0;
    while ((int)blockIdx.x >= sp.unit_off[b + 1]) ++b;
    const int nwg = sp.unit_off[b + 1] - sp.unit_off[b];
    const int stride = 4 * nwg;
    const int t0 = sp.tile_off[b] + 4 * ((int)blockIdx.x - sp.unit_off[b]) + wave, t1 = sp.tile_off[b + 1];
    const int cnt = t0 < t1 ? (t1 - t0 + stride - 1) / stride : 0;
    const float d = (float)(b + 1), inv = 1.0f / d;
    {
        const float *wd = W + (int64_t)b * Fi * Fo;
        constexpr int kTrips = (16 * IT * WP + 255) / 256;
        float wv[kTrips];
#pragma unroll
        for (int t = 0; t < kTrips; ++t) {
            const int e = threadIdx.x + 256 * t, i = e / WP, o = e - i * WP;
            wv[t] = (i < Fi && o < Fo) ? wd[(int64_t)i * Fo + o] : 0.0f;
        }
#pragma unroll
        for (int t = 0; t < kTrips; ++t) {
            const int e = threadIdx.x + 256 * t;
            if (e < 16 * IT * WP) wl[e] = wv[t];
        }
    }
    // A operand of the dz product: (ft, r) -> R(o = 4 q + r, f = 16 ft + n), R flat o + O f (readout.hip)
    float Ra[OT][4];
#pragma unroll
    for (int ft = 0; ft < OT; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ra[ft][r] = 4 * q + r < O ? R[(size_t)(16 * ft + n) * O + 4 * q + r] : 0.0f;
    WideTail<IT> ta;
    ta.init(lane, Fi);
    for (int e = lane; e < kWave; e += 64) al[e] = 0.0f;    // tail columns stay zero
    __syncthreads();

    v4f acc[IT][OT], accR[OT];
#pragma unroll
    for (int i = 0; i < IT; ++i)
#pragma unroll
        for (int o = 0; o < OT; ++o) acc[i][o] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int o = 0; o < OT; ++o) accR[o] = v4f{0.0f, 0.0f, 0.0f, 0.0f};

    struct Ids {
        v4i f;                                          // rows 4 i + q of the tile, padding slots negative
        int a[TA > 0 ? TA : 1];
        int rn, gd;                                     // row n of the tile (padding negative) and its graph
    };
    auto issue_ids = [&](Ids &id, int i) {
        const int64_t tb = (int64_t)(t0 + (i < cnt ? i : cnt - 1) * stride) * 16;
        id.f = *reinterpret_cast<const v4i *>(trows_t + tb + 4 * q);
#pragma unroll
        for (int u = 0; u < TA; ++u) id.a[u] = trows[tb + ta.row[u]];
        id.rn = trows[tb + n];
        id.gd = tgid[tb + n];
    };
    v4f an[4 + (TA > 0 ? TA : 0) + 1], zn[4], en[TA > 0 ? TA : 1];
    [[maybe_unused]] v4f dn[4];
    float pn[4], gn[4];
    auto issue_rows = [&](const Ids &id) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = id.f[i] ^ (id.f[i] >> 31);
            an[i] = *reinterpret_cast<const v4f *>(A + (int64_t)r * apm + 4 * n);
            zn[i] = *reinterpret_cast<const v4f *>(Z + (int64_t)r * Fo + 4 * n);
            if constexpr (DIN) dn[i] = *reinterpret_cast<const v4f *>(DZ + (int64_t)r * Fo + 4 * n);
        }
#pragma unroll
        for (int u = 0; u < TA; ++u) {
            const int r = id.a[u] ^ (id.a[u] >> 31);
            an[4 + u] = *reinterpret_cast<const v4f *>((ta.col[u] >= 64 ? at + (int64_t)r * apt : A + (int64_t)r * apm) + ta.col[u]);
            en[u] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
            if (acc_e && ta.col[u] >= 64) en[u] = *reinterpret_cast<const v4f *>(DAT - 64 + (int64_t)r * (Fi - 64) + ta.col[u]);
        }
        const int rv = id.rn ^ (id.rn >> 31);
        const float *ps = P + (int64_t)rv * O + 4 * q, *gs = GOUT + (int64_t)id.gd * O + 4 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = 4 * q + r < O;
            pn[r] = ok ? ps[r] : 0.0f;
            gn[r] = ok ? gs[r] : 0.0f;
        }
    };
    if (cnt > 0) {
        Ids I0, I1;
        issue_ids(I0, 0);
        issue_ids(I1, 1);
        issue_rows(I0);
        auto body = [&](Ids &cur, Ids &nxt, int i) {
            const v4f zero = {0.0f, 0.0f, 0.0f, 0.0f};
            int srow[4 + (TA > 0 ? TA : 1)];
#pragma unroll
            for (int k = 0; k < 4; ++k) srow[k] = cur.f[k] ^ (cur.f[k] >> 31);
#pragma unroll
            for (int u = 0; u < TA; ++u) srow[4 + u] = cur.a[u] ^ (cur.a[u] >> 31);
            const bool live = cur.rn >= 0;             // row n of this tile is a vertex, not a padding slot
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                *reinterpret_cast<v4f *>(al + (4 * k + q) * AP + 4 * n) = cur.f[k] >= 0 ? an[k] : zero;
                *reinterpret_cast<v4f *>(zt + (4 * k + q) * ZP + 4 * n) = zn[k];
                if constexpr (DIN) *reinterpret_cast<v4f *>(gl + (4 * k + q) * GP + 4 * n) = dn[k];
            }
#pragma unroll
            for (int u = 0; u < TA; ++u) *reinterpret_cast<v4f *>(al + ta.row[u] * AP + ta.col[u]) = cur.a[u] >= 0 ? an[4 + u] : zero;
            float pf[4], gf[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) pf[r] = pn[r], gf[r] = gn[r];
            v4f ef[TA > 0 ? TA : 1];
#pragma unroll
            for (int u = 0; u < TA; ++u) ef[u] = en[u];
            issue_ids(cur, i + 2);             // ids of tile i+2 ...
            issue_rows(nxt);                   // ... and the rows of tile i+1
            asm volatile("" ::: "memory");
            // ---- the readout's reverse on this tile: lane (v = n, q) holds columns 16 ft + 4 q .. + 3 of vertex row v
            v4f zf[OT];
            [[maybe_unused]] v4f df[OT];
#pragma unroll
            for (int ft = 0; ft < OT; ++ft) {
                zf[ft] = *reinterpret_cast<const v4f *>(zt + n * ZP + 16 * ft + 4 * q);
                if constexpr (DIN) df[ft] = *reinterpret_cast<const v4f *>(gl + n * GP + 16 * ft + 4 * q);
            }
            float dot = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) dot = dot + gf[r] * pf[r];
            dot = dot + __shfl_xor(dot, 16);
            dot = dot + __shfl_xor(dot, 32);
            float dl[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) dl[r] = pf[r] * (gf[r] - dot);
#pragma unroll
            for (int r = 0; r < 4; ++r) dll[n * DP + 4 * q + r] = live ? dl[r] : 0.0f;   // a repeated row counts once in dR
#pragma unroll
            for (int ft = 0; ft < OT; ++ft) {
                v4f dz = zero;
#pragma unroll
                for (int r = 0; r < 4; ++r) dz = __builtin_amdgcn_mfma_f32_16x16x4f32(Ra[ft][r], dl[r], dz, 0, 0, 0);
                if constexpr (DIN) dz = dz + df[ft];
                v4f o4;
#pragma unroll
                for (int c = 0; c < 4; ++c) o4[c] = duv_act_back<ACT>(zf[ft][c], dz[c]);
                *reinterpret_cast<v4f *>(gl + n * GP + 16 * ft + 4 * q) = o4;   // the slot df came from: same lane, same address
            }
            asm volatile("" ::: "memory");
            // dR(f, o) += sum_v z[v, f] dl[v, o]: the vertex on the k axis, both operands read turned
            {
                float db[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) db[r] = dll[(4 * q + r) * DP + n];
#pragma unroll
                for (int ft = 0; ft < OT; ++ft)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        accR[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(zt[(4 * q + r) * ZP + 16 * ft + n], db[r], accR[ft], 0, 0, 0);
            }
            // ---- from here on duv_bwd_wide_kernel with the gradient tile already in LDS
            // (the compiler's own DS / MFMA interleave for this region: 0.650 -> 0.629 ms without a dz_next, 0.663 -> 0.651 with one, same bits)
            __builtin_amdgcn_iglp_opt(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float aa[IT], bb[OT];
#pragma unroll
                for (int ii = 0; ii < IT; ++ii) aa[ii] = al[(4 * q + r) * AP + 16 * ii + n];
#pragma unroll
                for (int o = 0; o < OT; ++o) bb[o] = gl[(4 * q + r) * GP + 16 * o + n];
#pragma unroll
                for (int ii = 0; ii < IT; ++ii)
#pragma unroll
                    for (int o = 0; o < OT; ++o)
                        acc[ii][o] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa[ii], bb[o], acc[ii][o], 0, 0, 0);
            }
            v4f xg[OT], da[IT];
#pragma unroll
            for (int j = 0; j < OT; ++j) xg[j] = *reinterpret_cast<const v4f *>(gl + n * GP + 16 * j + 4 * q);
#pragma unroll
            for (int it = 0; it < IT; ++it) da[it] = zero;
#pragma unroll
            for (int j = 0; j < OT; ++j) {
                v4f wf[IT];
#pragma unroll
                for (int it = 0; it < IT; ++it) wf[it] = *reinterpret_cast<const v4f *>(wl + (16 * it + n) * WP + 16 * j + 4 * q);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int it = 0; it < IT; ++it) da[it] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[it][c], xg[j][c], da[it], 0, 0, 0);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int it = 0; it < IT; ++it) *reinterpret_cast<v4f *>(al + n * AP + 16 * it + 4 * q) = div4(da[it], d, inv);
            asm volatile("" ::: "memory");
            const int64_t pm = DAT ? 64 : Fi, pt = DAT ? Fi - 64 : Fi;
            float *dat = DAT ? DAT - 64 : DA;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const v4f y = *reinterpret_cast<const v4f *>(al + (4 * k + q) * AP + 4 * n);
                *reinterpret_cast<v4f *>(DA + (int64_t)srow[k] * pm + 4 * n) = y;
            }
#pragma unroll
            for (int u = 0; u < TA; ++u) {
                v4f y = *reinterpret_cast<const v4f *>(al + ta.row[u] * AP + ta.col[u]);
                if (acc_e && ta.col[u] >= 64) y = y + ef[u];
                float *dst = ta.col[u] >= 64 ? dat + (int64_t)srow[4 + u] * pt : DA + (int64_t)srow[4 + u] * pm;
                *reinterpret_cast<v4f *>(dst + ta.col[u]) = y;
            }
            asm volatile("" ::: "memory");
            if (Fi < 16 * IT) {
#pragma unroll
                for (int it = IT - 1; it < IT; ++it)
                    if (16 * it + 4 * q >= Fi) *reinterpret_cast<v4f *>(al + n * AP + 16 * it + 4 * q) = zero;
            }
        };
        int i = 0;
        for (; i + 2 <= cnt; i += 2) {
            body(I0, I1, i);
            body(I1, I0, i + 1);
        }
        if (i < cnt) body(I0, I1, i);
    }
    __syncthreads();
    // accR[ft][r] = dR(f = 16 ft + 4 q + r, o = n): the four waves' images side by side, added in wave order below
#pragma unroll
    for (int ft = 0; ft < OT; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r) redr[wave * Fo * 16 + (16 * ft + 4 * q + r) * 16 + n] = accR[ft][r];
    for (int p = 0; p < 4; ++p) {
        if (wave == p) {
#pragma unroll
            for (int i = 0; i < IT; ++i)
#pragma unroll
                for (int o = 0; o < OT; ++o)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float *dst = red + (16 * i + 4 * q + r) * FOP + 16 * o + n;
                        *dst = (p == 0 ? 0.0f : *dst) + acc[i][o][r];
                    }
        }
        __syncthreads();
    }
    float *slab = slabs + (size_t)blockIdx.x * Fi * Fo;
    for (int t = threadIdx.x; t < Fi * Fo; t += 256) {
        const int i = t / Fo, o = t - i * Fo;
        slab[t] = red[i * FOP + o] / d;
    }
    float *rslab = rslabs + (size_t)blockIdx.x * Fo * O;
    for (int t = threadIdx.x; t < Fo * O; t += 256) {
        const int f = t / O, o = t - f * O;
        rslab[t] = ((redr[f * 16 + o] + redr[Fo * 16 + f * 16 + o]) + redr[2 * Fo * 16 + f * 16 + o]) + redr[3 * Fo * 16 + f * 16 + o];
    }
}

// split of at most `units` waves / workgroups over the buckets: every non-empty bucket gets >= 1, the rest go one by one to the
// bucket whose units carry the most tiles (min-max).  Never more than `units` in total: the launches size `units` to what is
// resident at once, and a workgroup beyond that starts when the first one ends.  (Round 5, per-block stamps of the fused reverse
// launch at configs[2], profiles/r05_c3_bwd_timeline.txt: rounding the proportional share to the nearest gave the bucket of 335 tiles
// one workgroup -- 84 tiles per wave against 72 -- and two workgroups beyond the 512 resident ones; the launch ended with them.)
BucketSplit make_split(const athena_mp_graph *g, int units, int tiles_per_unit_step)
{
    BucketSplit sp;
    const int nb = (int)g->btile_off.size() - 1;
    sp.n_buckets = nb;
    int u[kMaxBuckets + 1], steps[kMaxBuckets + 1], tb[kMaxBuckets + 1], used = 0;
    for (int b = 0; b < nb; ++b) {
        tb[b] = g->btile_off[b + 1] - g->btile_off[b];
        steps[b] = (tb[b] + tiles_per_unit_step - 1) / tiles_per_unit_step;   // units that can be kept busy
        u[b] = tb[b] > 0 ? 1 : 0;
        used += u[b];
    }
    // start from the floor of the proportional share, then hand out what is left
    const int nt = g->n_btiles;
    for (int b = 0; b < nb && nt > 0; ++b) {
        const int share = (int)std::min<int64_t>(steps[b], ((int64_t)(units - used) * tb[b]) / nt);
        if (share > 0) u[b] += std::min(share, steps[b] - u[b]);
    }
    used = 0;
    for (int b = 0; b < nb; ++b) used += u[b];
    while (used < units) {
        int best = -1;
        for (int b = 0; b < nb; ++b)
            if (u[b] > 0 && u[b] < steps[b] && (best < 0 || (int64_t)tb[b] * u[best] > (int64_t)tb[best] * u[b])) best = b;
        if (best < 0) break;
        ++u[best];
        ++used;
    }
    sp.unit_off[0] = 0;
    for (int b = 0; b < nb; ++b) {
        sp.unit_off[b + 1] = sp.unit_off[b] + u[b];
        sp.tile_off[b] = g->btile_off[b];
    }
    sp.tile_off[nb] = g->btile_off[nb];
    for (int b = nb + 1; b <= kMaxBuckets; ++b) sp.unit_off[b] = sp.unit_off[nb], sp.tile_off[b] = sp.tile_off[nb];
    return sp;
}

inline int ceil16(int x) { return (x + 15) / 16; }
inline bool frag_shape(int kj, int ot) { return kj >= 1 && ot >= 1 && kj <= 6 && ot <= 6 && kj * ot <= 24; }

struct ReadoutArgs {   // the readout of the same time step in the epilogue (duv_rows_wide_kernel<.., true>)
    const float *R;
    int O;
    float *P;
};

int launch_rows(const athena_mp_graph *g, const float *X, int K, const float *W, int64_t wb, int so, int sk, float *Y,
                int NO, int act, const ReadoutArgs *ro = nullptr, const float *XT = nullptr)
{
    if (ro && (ro->O < 1 || ro->O > 16 || (size_t)g->n_rows * ro->O * sizeof(float) >= ((size_t)1 << 32) - 4096)) return -1;
    const int kj = ceil16(K), ot = ceil16(NO);
    // split input rows: ONE instantiation, <5, 4, readout> (the 96-wide one spills inside its tile loop: scripts/isa_lint.py R2)
    if (XT && !(kj == 5 && ot == 4 && ro)) return -1;
    if ((K & 3) || (NO & 3) || !frag_shape(kj, ot)) return -1;
    const int nt = g->n_btiles;
    if (nt == 0) return 0;
    if ((int)g->btile_off.size() - 1 > kMaxBuckets) return -1;
    const BucketSplit sp = make_split(g, 256 * 4 * 2, 1);   // two resident waves per SIMD at ~220 VGPRs
    const int32_t *trows_abs = g->btile_rows + (size_t)16 * nt;   // padding slots decoded (duvenaud_buckets)
    const dim3 grid((sp.unit_off[sp.n_buckets] + 3) / 4);
    if (K >= 64 && NO >= 64) {   // 16+ chunks per row on both sides: the structured numbering (two waves per SIMD at 80 fragments)
        const int32_t *trows_t = g->btile_rows + (size_t)32 * nt;
        const uint32_t p_bytes = ro ? (uint32_t)((size_t)g->n_rows * ro->O * sizeof(float)) : 0u;
#ifdef DUV_PLAIN_TAIL   // A/B builds (scripts/build_variants.sh ... -DDUV_PLAIN_TAIL=1): the tail fragment in 4 half-empty MFMAs
        constexpr bool tail2 = false;
#else
        constexpr bool tail2 = true;
#endif
#define AMP_WIDE(KJ_, OT_)                                                                                            \
    if (kj == KJ_ && ot == OT_) {                                                                                     \
        if (KJ_ == 5 && OT_ == 4 && K == 72 && tail2) {   /* 72 -> 64 (configs[2]): the tail fragment in 2 MFMAs, all three forms */ \
            if constexpr (KJ_ == 5 && OT_ == 4) {                                                                       \
                if (ro && XT)                                                                                         \
                    hipLaunchKernelGGL((duv_rows_wide_kernel<5, 4, true, true, 2>), grid, dim3(256), 0, amp::stream(), sp, trows_abs, \
                                       trows_t, X, K, W, wb, so, sk, Y, NO, act, ro->R, ro->O, ro->P, p_bytes, XT);    \
                else if (ro)                                                                                          \
                    hipLaunchKernelGGL((duv_rows_wide_kernel<5, 4, true, false, 2>), grid, dim3(256), 0, amp::stream(), sp, trows_abs, \
                                       trows_t, X, K, W, wb, so, sk, Y, NO, act, ro->R, ro->O, ro->P, p_bytes, XT);    \
                else                                                                                                  \
                    hipLaunchKernelGGL((duv_rows_wide_kernel<5, 4, false, false, 2>), grid, dim3(256), 0, amp::stream(), sp, trows_abs, \
                                       trows_t, X, K, W, wb, so, sk, Y, NO, act, nullptr, 0, nullptr, 0u, XT);         \
            }                                                                                                         \
        } else if (ro && XT) {                                                                                        \
            if constexpr (KJ_ == 5 && OT_ == 4)   /* the split input: 65 .. 80 columns -> 64 only */                          \
                hipLaunchKernelGGL((duv_rows_wide_kernel<5, 4, true, true>), grid, dim3(256), 0, amp::stream(), sp, trows_abs, \
                                   trows_t, X, K, W, wb, so, sk, Y, NO, act, ro->R, ro->O, ro->P, p_bytes, XT);        \
        } else if (ro)                                                                                                  \
            hipLaunchKernelGGL((duv_rows_wide_kernel<KJ_, OT_, true>), grid, dim3(256), 0, amp::stream(), sp, trows_abs, \
                               trows_t, X, K, W, wb, so, sk, Y, NO, act, ro->R, ro->O, ro->P, p_bytes, XT);            \
        else                                                                                                          \
            hipLaunchKernelGGL((duv_rows_wide_kernel<KJ_, OT_, false>), grid, dim3(256), 0, amp::stream(), sp, trows_abs, \
                               trows_t, X, K, W, wb, so, sk, Y, NO, act, nullptr, 0, nullptr, 0u, XT);                 \
    }
        AMP_WIDE(4, 4) AMP_WIDE(4, 5) AMP_WIDE(5, 4) AMP_WIDE(4, 6) AMP_WIDE(6, 4)
#undef AMP_WIDE
        AMP_LAUNCH_CHECK();
        return 0;
    }
    if (ro || XT) return -1;   // the readout epilogue and split input rows exist in the wide kernel only
#define AMP_CASE(KJ_, OT_)                                                                                         \
    if (kj == KJ_ && ot == OT_) {                                                                                  \
        hipLaunchKernelGGL((duv_rows_any_kernel<KJ_, OT_>), grid, dim3(256), 0, amp::stream(), sp, trows_abs, X,      \
                           K, W, wb, so, sk, Y, NO, act);                                                  \
    }
#define AMP_ROW(KJ_) AMP_CASE(KJ_, 1) AMP_CASE(KJ_, 2) AMP_CASE(KJ_, 3) AMP_CASE(KJ_, 4)
    AMP_ROW(1) AMP_ROW(2) AMP_ROW(3) AMP_ROW(4) AMP_ROW(5) AMP_ROW(6)
    AMP_CASE(1, 5) AMP_CASE(2, 5) AMP_CASE(3, 5) AMP_CASE(4, 5) AMP_CASE(1, 6) AMP_CASE(2, 6) AMP_CASE(3, 6) AMP_CASE(4, 6)
#undef AMP_ROW
#undef AMP_CASE
    AMP_LAUNCH_CHECK();
    return 0;
}

} // namespace

namespace amp {

int duv_mfma_fwd(const athena_mp_graph *g, int Fi, int Fo, const float *a, const float *w, int act, float *c)
{
    // W_d(o,i) flat o + Fo*i:  output index o (stride 1), contraction index i (stride Fo)
    return launch_rows(g, a, Fi, w, (int64_t)Fo * Fi, 1, Fo, c, Fo, act);
}

int duv_mfma_bwd_a(const athena_mp_graph *g, int Fi, int Fo, const float *grad, const float *w, float *da)
{
    // da[v,i] = (sum_o g[v,o] W_d(o,i)) / d:  output index i (stride Fo), contraction index o (stride 1)
    return launch_rows(g, grad, Fo, w, (int64_t)Fo * Fi, Fo, 1, da, Fi, ATHENA_MP_ACT_NONE);
}

int duv_mfma_bwd_w(const athena_mp_graph *g, int Fi, int Fo, const float *grad, const float *a, float *dw)
{
    const int it = ceil16(Fi), ot = ceil16(Fo);
    if ((Fi & 3) || (Fo & 3) || !frag_shape(it, ot)) return -1;
    const int nt = g->n_btiles, nb = (int)g->btile_off.size() - 1, n = Fi * Fo;
    if (nt == 0) {
        AMP_HIP(hipMemsetAsync(dw, 0, sizeof(float) * (size_t)nb * n, stream()));
        return 0;
    }
    if (nb > kMaxBuckets) return -1;
    const BucketSplit sp = make_split(g, 512, 4);
    const int nwg = sp.unit_off[nb];
    void *slabs = nullptr;
    if (workspace(&slabs, sizeof(float) * (size_t)nwg * n, 2)) return 1;
    if (Fi >= 64 && Fo >= 64) {
#define AMP_WIDE(IT_, OT_)                                                                                            \
    if (it == IT_ && ot == OT_)                                                                                       \
        hipLaunchKernelGGL((duv_dw_wide_kernel<IT_, OT_>), dim3(nwg), dim3(256), 0, stream(), sp, g->btile_rows,      \
                           g->btile_rows + (size_t)48 * nt, a, Fi, grad, Fo, (float *)slabs);
        AMP_WIDE(4, 4) AMP_WIDE(4, 5) AMP_WIDE(5, 4) AMP_WIDE(4, 6) AMP_WIDE(6, 4)
#undef AMP_WIDE
    } else {
#define AMP_CASE(IT_, OT_)                                                                                         \
    if (it == IT_ && ot == OT_) {                                                                                  \
        hipLaunchKernelGGL((duv_dw_any_kernel<IT_, OT_>), dim3(nwg), dim3(256), 0, stream(), sp, g->btile_rows, a, Fi, \
                           grad, Fo, (float *)slabs);                                                              \
    }
#define AMP_ROW(IT_) AMP_CASE(IT_, 1) AMP_CASE(IT_, 2) AMP_CASE(IT_, 3) AMP_CASE(IT_, 4)
    AMP_ROW(1) AMP_ROW(2) AMP_ROW(3) AMP_ROW(4) AMP_ROW(5) AMP_ROW(6)
    AMP_CASE(1, 5) AMP_CASE(2, 5) AMP_CASE(3, 5) AMP_CASE(4, 5) AMP_CASE(1, 6) AMP_CASE(2, 6) AMP_CASE(3, 6) AMP_CASE(4, 6)
#undef AMP_ROW
#undef AMP_CASE
    }
    AMP_LAUNCH_CHECK();
    std::vector<int> first(nb, 0), count(nb, 0);
    for (int b = 0; b < nb; ++b) first[b] = sp.unit_off[b], count[b] = sp.unit_off[b + 1] - sp.unit_off[b];
    return slab_reduce_segs((const float *)slabs, n, nb, first.data(), count.data(), dw, n, false);
}

// update + activation + the readout's p = softmax(R z) in one launch; -1: shape outside the wide kernel
// a_tail != null: a split into a [n, 64] and a_tail [n, Fi - 64]
int duv_mfma_fwd_readout(const athena_mp_graph *g, int Fi, int Fo, const float *a, const float *w, int act, float *z,
                         const float *R, int O, float *p, const float *a_tail)
{
    const ReadoutArgs ro{R, O, p};
    return launch_rows(g, a, Fi, w, (int64_t)Fo * Fi, 1, Fo, z, Fo, act, &ro, a_tail);
}

// da and dW from one pass over grad (both widths 64 .. 96); -1: shape outside the fused kernel
// da_tail != null: da split into da [n, 64] and da_tail [n, Fi - 64] (Fi > 64 only)
int duv_mfma_bwd(const athena_mp_graph *g, int Fi, int Fo, const float *grad, const float *a, const float *w, float *da, float *dw,
                 float *da_tail)
{
    if (da_tail && Fi <= 64) return -1;
    const int it = ceil16(Fi), ot = ceil16(Fo);
    if ((Fi & 3) || (Fo & 3) || Fi < 64 || Fo < 64 || !frag_shape(it, ot)) return -1;
    const int nt = g->n_btiles, nb = (int)g->btile_off.size() - 1, n = Fi * Fo;
    if (nt == 0) {
        AMP_HIP(hipMemsetAsync(dw, 0, sizeof(float) * (size_t)nb * n, stream()));
        return 0;
    }
    if (nb > kMaxBuckets) return -1;
    const BucketSplit sp = make_split(g, 512, 4);
    const int nwg = sp.unit_off[nb];
    void *slabs = nullptr;
    if (workspace(&slabs, sizeof(float) * (size_t)nwg * n, 2)) return 1;
    bool launched = false;
#define AMP_WIDE(IT_, OT_)                                                                                            \
    if (it == IT_ && ot == OT_) {                                                                                     \
        hipLaunchKernelGGL((duv_bwd_wide_kernel<IT_, OT_>), dim3(nwg), dim3(256), 0, stream(), sp, g->btile_rows,     \
                           g->btile_rows + (size_t)48 * nt, a, Fi, grad, Fo, w, da, (float *)slabs, da_tail);         \
        launched = true;                                                                                              \
    }
    AMP_WIDE(4, 4) AMP_WIDE(5, 4) AMP_WIDE(4, 5) AMP_WIDE(6, 4) AMP_WIDE(4, 6)
#undef AMP_WIDE
    if (!launched) return -1;
    AMP_LAUNCH_CHECK();
    std::vector<int> first(nb, 0), count(nb, 0);
    for (int b = 0; b < nb; ++b) first[b] = sp.unit_off[b], count[b] = sp.unit_off[b + 1] - sp.unit_off[b];
    return slab_reduce_segs((const float *)slabs, n, nb, first.data(), count.data(), dw, n, false);
}

// the readout's reverse and the update's reverse in one launch (F_o = 64, F_i = 64 .. 96, O <= 16); -1: shape outside it.
// tgid: the graph of every tile slot, [16 tiles]; da_tail as in duv_mfma_bwd; dr_slabs: [workgroups][64 O]
int duv_mfma_bwd_readout(const athena_mp_graph *g, int Fi, int Fo, int O, int act, const float *a, const float *w, const float *z,
                         const float *dz_next, const float *p, const int32_t *tgid, const float *gout, const float *R, float *da,
                         float *da_tail, float *dw, float *dr_slabs, int *n_slabs, bool accumulate_tail, const float *a_tail)
{
    const int it = ceil16(Fi);
    if (!da_tail || (Fi & 3) || Fo != 64 || Fi <= 64 || it > 6 || O < 1 || O > 16 || act < 0 || act > ATHENA_MP_ACT_TANH) return -1;
    const int nt = g->n_btiles, nb = (int)g->btile_off.size() - 1, n = Fi * Fo;
    if (nt == 0 || nb > kMaxBuckets) return -1;
    const BucketSplit sp = make_split(g, 256, 4);          // one workgroup per CU
    const int nwg = sp.unit_off[nb];
    *n_slabs = nwg;
    void *slabs = nullptr;
    if (workspace(&slabs, sizeof(float) * (size_t)nwg * n, 2)) return 1;
    bool launched = false;
#define AMP_RO(IT_, A_)                                                                                                      \
    if (it == IT_ && act == A_) {                                                                                            \
        if (dz_next)                                                                                                         \
            hipLaunchKernelGGL((duv_bwd_ro_kernel<IT_, A_, true>), dim3(nwg), dim3(256), 0, stream(), sp, g->btile_rows,      \
                               g->btile_rows + (size_t)48 * nt, tgid, a, Fi, z, dz_next, p, gout, R, O, w, da, da_tail,       \
                               (float *)slabs, dr_slabs, accumulate_tail ? 1 : 0, a_tail);                                  \
        else                                                                                                                 \
            hipLaunchKernelGGL((duv_bwd_ro_kernel<IT_, A_, false>), dim3(nwg), dim3(256), 0, stream(), sp, g->btile_rows,     \
                               g->btile_rows + (size_t)48 * nt, tgid, a, Fi, z, dz_next, p, gout, R, O, w, da, da_tail,       \
                               (float *)slabs, dr_slabs, accumulate_tail ? 1 : 0, a_tail);                                  \
        launched = true;                                                                                                     \
    }
#define AMP_RO_ACTS(IT_) AMP_RO(IT_, 0) AMP_RO(IT_, 1) AMP_RO(IT_, 2) AMP_RO(IT_, 3)
    AMP_RO_ACTS(5) AMP_RO_ACTS(6)          // F_e > 0 with the split da: F_i = 68 .. 96
#undef AMP_RO_ACTS
#undef AMP_RO
    if (!launched) return -1;
    AMP_LAUNCH_CHECK();
    std::vector<int> first(nb, 0), count(nb, 0);
    for (int b = 0; b < nb; ++b) first[b] = sp.unit_off[b], count[b] = sp.unit_off[b + 1] - sp.unit_off[b];
    return slab_reduce_segs((const float *)slabs, n, nb, first.data(), count.data(), dw, n, false);
}


} // namespace amp

