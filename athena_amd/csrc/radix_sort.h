// Stable least-significant-digit radix sort of (key, int32 value) pairs for the graph builder (graph_build.hip) -- hand-written for
// gfx950's 64-wide wavefronts, no library behind it.  Every pass is a stable COUNTING sort on one 8-bit digit:
//   radix_hist_kernel     one 256-bin histogram per tile of kTile consecutive entries (LDS atomics), stored bin-major [256][tiles]
//   radix_scan_rows       per bin: exclusive scan over the tiles (where each tile's entries of that digit start inside the bin)
//   radix_scan_bins       exclusive scan of the 256 bin totals (where each bin starts)
//   radix_scatter_kernel  every tile again, 256 entries per round in index order: an entry's place = bin start + tile offset +
//                         entries of the same digit earlier in the tile.  "Earlier in the round" comes from wave-wide ballots
//                         (the lanes of a wave that hold the same digit, eight ballots) and a per-wave count table in LDS -- no
//                         atomics, so equal keys keep their input order, which is what makes the transposed CSR list its sources
//                         in ascending order (the reference's scatter order, athena_diffstruc_extd_sub_kipf.f90:101-109).
// Passes ping-pong between the output and one scratch pair; the input is never written.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace amp {
namespace radix {

constexpr int kThreads = 256;                  // 4 waves
constexpr int kRounds = 16;
constexpr int kTile = kThreads * kRounds;      // 4096 entries per tile
constexpr int kBins = 256;

template <typename K> __global__ __launch_bounds__(kThreads) void radix_hist_kernel(const K *__restrict__ keys, int64_t n, int shift,
                                                                                   uint32_t *__restrict__ hist, uint32_t tiles)
{
    __shared__ uint32_t h[kBins];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kTile;
#pragma unroll 4
    for (int r = 0; r < kRounds; ++r) {
        const int64_t i = base + (int64_t)r * kThreads + threadIdx.x;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * tiles + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of the 256 values one block holds (one per thread); returns the value's prefix, *total the sum
__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t *total)
{
    __shared__ uint32_t wsum[kThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(inc, d, 64);
        if (lane >= d) inc += up;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t off = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) {
        if (w < wave) off += wsum[w];
        all += wsum[w];
    }
    __syncthreads();
    *total = all;
    return off + inc - v;
}

// grid = 256 bins: the bin's row of per-tile counts becomes per-tile offsets inside the bin; bin_total[bin] = its size
__global__ __launch_bounds__(kThreads) void radix_scan_rows(uint32_t *__restrict__ hist, uint32_t tiles, uint32_t *__restrict__ bin_total)
{
    uint32_t *row = hist + (size_t)blockIdx.x * tiles;
    const uint32_t per = (tiles + kThreads - 1) / kThreads;
    const uint32_t lo = min(tiles, threadIdx.x * per), hi = min(tiles, lo + per);
    uint32_t s = 0;
    for (uint32_t k = lo; k < hi; ++k) s += row[k];
    uint32_t total;
    uint32_t run = block_exclusive_scan(s, &total);
    for (uint32_t k = lo; k < hi; ++k) {
        const uint32_t c = row[k];
        row[k] = run;
        run += c;
    }
    if (threadIdx.x == 0) bin_total[blockIdx.x] = total;
}

__global__ __launch_bounds__(kThreads) void radix_scan_bins(uint32_t *__restrict__ bin_total)
{
    uint32_t total;
    bin_total[threadIdx.x] = block_exclusive_scan(bin_total[threadIdx.x], &total);
}

template <typename K>
__global__ __launch_bounds__(kThreads) void radix_scatter_kernel(const K *__restrict__ keys, const int32_t *__restrict__ vals, int64_t n, int shift,
                                                                const uint32_t *__restrict__ hist, const uint32_t *__restrict__ bin_start,
                                                                uint32_t tiles, K *__restrict__ keys_out, int32_t *__restrict__ vals_out)
{
    __shared__ uint32_t place[kBins];              // where the tile's next entry of each digit goes
    __shared__ uint32_t wcnt[kThreads / 64][kBins]; // this round: entries of each digit per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    place[threadIdx.x] = bin_start[threadIdx.x] + hist[(size_t)threadIdx.x * tiles + blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * kTile;
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int r = 0; r < kRounds; ++r) {
        const int64_t i = base + (int64_t)r * kThreads + threadIdx.x;
        const bool valid = i < n;
        K key = 0;
        int32_t val = 0;
        if (valid) {
            key = keys[i];
            val = vals ? vals[i] : (int32_t)i;
        }
        const uint32_t d = (uint32_t)(key >> shift) & 255u;
        // the lanes of this wave that hold the same digit (invalid lanes sit at the tile's end and match nobody)
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            same &= bit ? bal : ~bal;
        }
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) wcnt[w][threadIdx.x] = 0;
        __syncthreads();
        const uint32_t before = (uint32_t)__popcll(same & below);
        if (valid && before == 0) wcnt[wave][d] = (uint32_t)__popcll(same);      // the group's first lane records its size
        __syncthreads();
        if (valid) {
            uint32_t off = before;
            for (int w = 0; w < wave; ++w) off += wcnt[w][d];
            const uint32_t pos = place[d] + off;
            keys_out[pos] = key;
            vals_out[pos] = val;
        }
        __syncthreads();
        uint32_t add = 0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) add += wcnt[w][threadIdx.x];
        place[threadIdx.x] += add;
        // (the next round's zeroing of wcnt happens after this thread's reads above; the barrier after it orders the rest)
        __syncthreads();
    }
}

inline uint32_t tiles_for(int64_t n) { return (uint32_t)((n + kTile - 1) / kTile); }
// bytes of the histogram scratch sort_pairs needs for n entries
inline size_t scratch_bytes(int64_t n) { return sizeof(uint32_t) * ((size_t)kBins * tiles_for(n) + kBins); }

// Stable sort of n pairs by the low `bits` bits of the key.  vals == nullptr: the values are the entry indices 0 .. n-1 (the sort
// then yields the permutation).  keys / vals are read only; the result lands in keys_out / vals_out; tmp_keys / tmp_vals [n] are
// the other half of the ping-pong (needed when bits > 8); scratch: scratch_bytes(n).  Launches on `st`, never synchronises.
template <typename K>
int sort_pairs(const K *keys, const int32_t *vals, int64_t n, int bits, K *keys_out, int32_t *vals_out, K *tmp_keys, int32_t *tmp_vals,
               void *scratch, hipStream_t st)
{
    if (n <= 0) return 0;
    const int passes = bits <= 8 ? 1 : (bits + 7) / 8;
    AMP_REQUIRE(passes == 1 || (tmp_keys && tmp_vals), "radix sort: more than one pass needs the scratch pair");
    const uint32_t tiles = tiles_for(n);
    uint32_t *hist = (uint32_t *)scratch, *bins = hist + (size_t)kBins * tiles;
    const K *src_k = keys;
    const int32_t *src_v = vals;
    for (int p = 0; p < passes; ++p) {
        const bool to_out = ((passes - 1 - p) & 1) == 0;          // the last pass writes the output
        K *dst_k = to_out ? keys_out : tmp_keys;
        int32_t *dst_v = to_out ? vals_out : tmp_vals;
        hipLaunchKernelGGL(radix_hist_kernel<K>, dim3(tiles), dim3(kThreads), 0, st, src_k, n, 8 * p, hist, tiles);
        hipLaunchKernelGGL(radix_scan_rows, dim3(kBins), dim3(kThreads), 0, st, hist, tiles, bins);
        hipLaunchKernelGGL(radix_scan_bins, dim3(1), dim3(kThreads), 0, st, bins);
        hipLaunchKernelGGL(radix_scatter_kernel<K>, dim3(tiles), dim3(kThreads), 0, st, src_k, src_v, n, 8 * p, (const uint32_t *)hist,
                           (const uint32_t *)bins, tiles, dst_k, dst_v);
        AMP_LAUNCH_CHECK();
        src_k = dst_k;
        src_v = dst_v;
    }
    return 0;
}

} // namespace radix
} // namespace amp
