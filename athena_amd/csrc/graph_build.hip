// Device-side construction of the graph handle: everything athena_mp_graph_create derives
// from the caller's CSR -- 0-based CSR, per-entry Kipf coefficient, transposed CSR, edge-column index --
// built in HBM from one upload of adj_ia / adj_ja, for callers whose graphs change between forward passes
// (set_graph runs before every forward, athena_network_sub.f90:2727-2730; radius graphs of moving points).
//
// It produces exactly the arrays of the host builder in capi.hip (tests compare them element for element):
//   * transposed rows list their sources in ascending row order = a STABLE sort of the entries by column
//     (radix_sort.h: counting-sort passes on 8-bit digits written for 64-wide wavefronts; no sort library), the
//     reference's scatter order (athena_diffstruc_extd_sub_kipf.f90:101-109);
//   * the edge-column index is the stable sort of the entries that carry an edge id by that id;
//   * the coefficient (deg_v*deg_u)**(-0.5) stays a HOST libm powf -- a device pow is 1 ulp off glibc for
//     ~0.06 % of the integer products -- but is evaluated once per pair of DISTINCT degrees into a small
//     table that the device indexes through degree ranks.
#include <math.h>

#include <algorithm>

#include "common.h"
#include "radix_sort.h"

namespace {

struct BadEntry {
    unsigned long long first_bad;   // smallest offending entry index, ~0 if none
};

__global__ void split_kernel(int64_t nnz, const int32_t *__restrict__ ja, int32_t n_cols, int32_t n_edge_cols,
                             int32_t *__restrict__ col, int32_t *__restrict__ eid, int32_t *__restrict__ ekey,
                             BadEntry *__restrict__ bad)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nnz) return;
    const int32_t u = ja[2 * w] - 1, e = ja[2 * w + 1] - 1;
    if (u < 0 || u >= n_cols || e < -1 || e >= n_edge_cols) atomicMin(&bad->first_bad, (unsigned long long)w);
    col[w] = u;
    eid[w] = e;
    if (ekey) ekey[w] = e + 1;   // 0 = carries no edge id: sorts in front
}

// row of every entry: binary search of the entry index in rowptr (upper bound - 1)
__global__ void row_of_kernel(int64_t nnz, int32_t n_rows, const int32_t *__restrict__ rowptr,
                              int32_t *__restrict__ row_of)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nnz) return;
    int lo = 0, hi = n_rows;   // rowptr[lo] <= w < rowptr[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (rowptr[mid] <= (int32_t)w) lo = mid; else hi = mid;
    }
    row_of[w] = lo;
}

__global__ void coef_kernel(int64_t nnz, const int32_t *__restrict__ row_of, const int32_t *__restrict__ col,
                            const int32_t *__restrict__ deg_row, const int32_t *__restrict__ deg_col,
                            const int32_t *__restrict__ rank_r, const int32_t *__restrict__ rank_c, int Kc,
                            const float *__restrict__ table, float *__restrict__ coef)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nnz) return;
    coef[w] = table[(size_t)rank_r[deg_row[row_of[w]]] * Kc + rank_c[deg_col[col[w]]]];
}

// transposed arrays from the stable (column, entry) sort
__global__ void gather_t_kernel(int64_t nnz, const int32_t *__restrict__ perm, const int32_t *__restrict__ row_of,
                                const int32_t *__restrict__ eid, const float *__restrict__ coef,
                                int32_t *__restrict__ t_src, int32_t *__restrict__ t_eid, float *__restrict__ t_coef)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    const int32_t w = perm[p];
    t_src[p] = row_of[w];
    t_eid[p] = eid[w];
    t_coef[p] = coef[w];
}

// ptr[k] = first position whose sorted key is >= k + key0  (k = 0 .. n), minus `shift`
__global__ void lower_bound_kernel(int32_t n, const int32_t *__restrict__ sorted, int64_t nnz, int32_t key0,
                                   int32_t shift, int32_t *__restrict__ ptr)
{
    const int32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n) return;
    const int32_t key = k + key0;
    int64_t lo = 0, hi = nnz;   // first index with sorted[idx] >= key
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (sorted[mid] < key) lo = mid + 1; else hi = mid;
    }
    ptr[k] = (int32_t)lo - shift;
}

__global__ void gather_e_kernel(int64_t n_with, int64_t n_none, const int32_t *__restrict__ perm,
                                const int32_t *__restrict__ row_of, int32_t *__restrict__ e_row,
                                int32_t *__restrict__ e_col)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_with) return;
    const int32_t w = perm[n_none + q];
    e_col[q] = w;
    e_row[q] = row_of[w];
}

inline unsigned blocks(int64_t n) { return (unsigned)((n + 255) / 256); }
inline int bits_for(int64_t max_value)
{
    int b = 1;
    while (b < 31 && ((int64_t)1 << b) <= max_value) ++b;
    return b;
}

template <typename T> int dev_alloc(T **p, size_t count)
{
    AMP_HIP(hipMalloc((void **)p, sizeof(T) * (count ? count : 1)));
    return 0;
}

struct Scratch {   // freed on every exit path
    std::vector<void *> ptrs;
    template <typename T> int get(T **p, size_t count)
    {
        if (dev_alloc(p, count)) return 1;
        ptrs.push_back(*p);
        return 0;
    }
    ~Scratch()
    {
        for (void *p : ptrs) (void)hipFree(p);
    }
};

} // namespace

namespace amp {

// returns 0 ok, -1 "use the host builder" (too many distinct degrees for the coefficient table), > 0 error.
// On success every device array of g except the long-row plans is filled; t_rowptr_host receives the
// transposed row pointers (the caller plans hub rows from them).
int graph_build_device(athena_mp_graph *g, const int32_t *adj_ja, const std::vector<int32_t> &rowptr,
                       const std::vector<int32_t> &degr, const std::vector<int32_t> &degc,
                       std::vector<int32_t> *t_rowptr_host, const int32_t *adj_ja_dev)
{
    // adj_ja_dev != nullptr: the CSR entries are already in HBM (athena_mp_graph_create_from_edges); adj_ja may
    // then be null
    const int32_t n_rows = g->n_rows, n_cols = g->n_cols, n_edge_cols = g->n_edge_cols;
    const int64_t nnz = g->nnz;
    hipStream_t st = stream();

    // ---- coefficient table over distinct degrees (host powf, bit-identical to the reference's) ----------
    auto distinct = [](const std::vector<int32_t> &d, std::vector<int32_t> *uniq, std::vector<int32_t> *rank) {
        int32_t mx = 0;
        for (int32_t v : d) mx = std::max(mx, v);
        std::vector<char> seen((size_t)mx + 1, 0);
        for (int32_t v : d) seen[v < 0 ? 0 : v] = 1;
        rank->assign((size_t)mx + 1, 0);
        for (int32_t v = 0; v <= mx; ++v)
            if (seen[v]) {
                (*rank)[v] = (int32_t)uniq->size();
                uniq->push_back(v);
            }
    };
    for (int32_t v : degr)
        if (v < 0) return -1;
    for (int32_t v : degc)
        if (v < 0) return -1;
    std::vector<int32_t> uniq_r, uniq_c, rank_r, rank_c;
    distinct(degr, &uniq_r, &rank_r);
    distinct(degc, &uniq_c, &rank_c);
    const size_t Kr = uniq_r.size(), Kc = uniq_c.size();
    if (Kr * Kc > ((size_t)1 << 22)) return -1;
    std::vector<float> table(Kr * Kc ? Kr * Kc : 1);
    for (size_t a = 0; a < Kr; ++a)
        for (size_t b = 0; b < Kc; ++b) table[a * Kc + b] = powf((float)(uniq_r[a] * uniq_c[b]), -0.5f);

    Scratch tmp;
    int32_t *d_ja = nullptr, *d_row_of = nullptr, *d_perm = nullptr, *d_keys_sorted = nullptr, *d_tmp_k = nullptr, *d_tmp_v = nullptr,
            *d_ekey = nullptr, *d_rank_r = nullptr, *d_rank_c = nullptr;
    float *d_table = nullptr;
    BadEntry *d_bad = nullptr;
    if ((adj_ja_dev == nullptr && tmp.get(&d_ja, 2 * (size_t)nnz)) || tmp.get(&d_row_of, nnz) || tmp.get(&d_perm, nnz) ||
        tmp.get(&d_keys_sorted, nnz) || tmp.get(&d_tmp_k, nnz) || tmp.get(&d_tmp_v, nnz) || tmp.get(&d_rank_r, rank_r.size()) || tmp.get(&d_rank_c, rank_c.size()) ||
        tmp.get(&d_table, table.size()) || tmp.get(&d_bad, 1))
        return 1;
    if (n_edge_cols > 0 && tmp.get(&d_ekey, nnz)) return 1;

    if (dev_alloc(&g->rowptr, rowptr.size()) || dev_alloc(&g->col, nnz) || dev_alloc(&g->eid, nnz) ||
        dev_alloc(&g->coef, nnz) || dev_alloc(&g->t_rowptr, (size_t)n_cols + 1) || dev_alloc(&g->t_src, nnz) ||
        dev_alloc(&g->t_eid, nnz) || dev_alloc(&g->t_coef, nnz) || dev_alloc(&g->e_rowptr, (size_t)n_edge_cols + 1) ||
        dev_alloc(&g->deg_row, degr.size()) || dev_alloc(&g->deg_col, degc.size()))
        return 1;

    const BadEntry none = {~0ull};
    AMP_HIP(hipMemcpyAsync(d_bad, &none, sizeof(none), hipMemcpyHostToDevice, st));
    if (adj_ja_dev == nullptr)
        AMP_HIP(hipMemcpyAsync(d_ja, adj_ja, sizeof(int32_t) * 2 * (size_t)nnz, hipMemcpyHostToDevice, st));
    else
        d_ja = const_cast<int32_t *>(adj_ja_dev);
    AMP_HIP(hipMemcpyAsync(g->rowptr, rowptr.data(), sizeof(int32_t) * rowptr.size(), hipMemcpyHostToDevice, st));
    if (!degr.empty()) AMP_HIP(hipMemcpyAsync(g->deg_row, degr.data(), sizeof(int32_t) * degr.size(), hipMemcpyHostToDevice, st));
    if (!degc.empty()) AMP_HIP(hipMemcpyAsync(g->deg_col, degc.data(), sizeof(int32_t) * degc.size(), hipMemcpyHostToDevice, st));
    AMP_HIP(hipMemcpyAsync(d_rank_r, rank_r.data(), sizeof(int32_t) * rank_r.size(), hipMemcpyHostToDevice, st));
    AMP_HIP(hipMemcpyAsync(d_rank_c, rank_c.data(), sizeof(int32_t) * rank_c.size(), hipMemcpyHostToDevice, st));
    AMP_HIP(hipMemcpyAsync(d_table, table.data(), sizeof(float) * table.size(), hipMemcpyHostToDevice, st));

    if (nnz > 0) {
        hipLaunchKernelGGL(split_kernel, dim3(blocks(nnz)), dim3(256), 0, st, nnz, (const int32_t *)d_ja, n_cols,
                           n_edge_cols, g->col, g->eid, d_ekey, d_bad);
        hipLaunchKernelGGL(row_of_kernel, dim3(blocks(nnz)), dim3(256), 0, st, nnz, n_rows, (const int32_t *)g->rowptr,
                           d_row_of);
        AMP_LAUNCH_CHECK();
        BadEntry bad;
        AMP_HIP(hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, st));
        AMP_HIP(hipStreamSynchronize(st));
        if (bad.first_bad != ~0ull) {   // same messages as the host builder
            const int64_t w = (int64_t)bad.first_bad;
            int32_t pair[2] = {0, 0};
            if (adj_ja) { pair[0] = adj_ja[2 * w]; pair[1] = adj_ja[2 * w + 1]; }
            else AMP_HIP(hipMemcpy(pair, d_ja + 2 * w, sizeof(pair), hipMemcpyDeviceToHost));
            const int32_t u = pair[0] - 1, e = pair[1] - 1;
            if (u < 0 || u >= n_cols)
                set_error("graph_create: adj_ja(1,%lld) = %d outside [1,%d]", (long long)w + 1, u + 1, n_cols);
            else
                set_error("graph_create: adj_ja(2,%lld) = %d outside [0,%d]", (long long)w + 1, e + 1, n_edge_cols);
            return 2;
        }
        hipLaunchKernelGGL(coef_kernel, dim3(blocks(nnz)), dim3(256), 0, st, nnz, (const int32_t *)d_row_of,
                           (const int32_t *)g->col, (const int32_t *)g->deg_row, (const int32_t *)g->deg_col,
                           (const int32_t *)d_rank_r, (const int32_t *)d_rank_c, (int)Kc, (const float *)d_table, g->coef);
        AMP_LAUNCH_CHECK();
    }

    // ---- transposed CSR: stable sort of the entries by column; the values are the entry indices ------------------
    void *d_temp = nullptr;
    const int col_bits = bits_for(std::max<int64_t>(n_cols - 1, 1));
    if (nnz > 0) {
        if (tmp.get((char **)&d_temp, radix::scratch_bytes(nnz))) return 1;
        if (int rc = radix::sort_pairs<uint32_t>((const uint32_t *)g->col, nullptr, nnz, col_bits, (uint32_t *)d_keys_sorted, d_perm,
                                                 (uint32_t *)d_tmp_k, d_tmp_v, d_temp, st))
            return rc;
        hipLaunchKernelGGL(gather_t_kernel, dim3(blocks(nnz)), dim3(256), 0, st, nnz, (const int32_t *)d_perm,
                           (const int32_t *)d_row_of, (const int32_t *)g->eid, (const float *)g->coef, g->t_src, g->t_eid,
                           g->t_coef);
        AMP_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(lower_bound_kernel, dim3(blocks((int64_t)n_cols + 1)), dim3(256), 0, st, n_cols,
                       (const int32_t *)d_keys_sorted, nnz, 0, 0, g->t_rowptr);
    AMP_LAUNCH_CHECK();
    t_rowptr_host->resize((size_t)n_cols + 1);
    AMP_HIP(hipMemcpyAsync(t_rowptr_host->data(), g->t_rowptr, sizeof(int32_t) * ((size_t)n_cols + 1),
                           hipMemcpyDeviceToHost, st));

    // ---- edge-column index: stable sort of the entries by edge id, entries without one in front --------
    int64_t n_with = 0;
    if (n_edge_cols > 0 && nnz > 0) {
        if (int rc = radix::sort_pairs<uint32_t>((const uint32_t *)d_ekey, nullptr, nnz, bits_for(n_edge_cols), (uint32_t *)d_keys_sorted,
                                                 d_perm, (uint32_t *)d_tmp_k, d_tmp_v, d_temp, st))
            return rc;
        // e_rowptr[e] = lower_bound(key >= e + 1) - n_none; computed in two steps: first the raw positions
        hipLaunchKernelGGL(lower_bound_kernel, dim3(blocks((int64_t)n_edge_cols + 1)), dim3(256), 0, st, n_edge_cols,
                           (const int32_t *)d_keys_sorted, nnz, 1, 0, g->e_rowptr);
        AMP_LAUNCH_CHECK();
        int32_t n_none = 0;
        AMP_HIP(hipMemcpyAsync(&n_none, g->e_rowptr, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        AMP_HIP(hipStreamSynchronize(st));
        n_with = nnz - n_none;
        hipLaunchKernelGGL(lower_bound_kernel, dim3(blocks((int64_t)n_edge_cols + 1)), dim3(256), 0, st, n_edge_cols,
                           (const int32_t *)d_keys_sorted, nnz, 1, n_none, g->e_rowptr);
        AMP_LAUNCH_CHECK();
        if (dev_alloc(&g->e_row, n_with) || dev_alloc(&g->e_col, n_with)) return 1;
        if (n_with > 0) {
            hipLaunchKernelGGL(gather_e_kernel, dim3(blocks(n_with)), dim3(256), 0, st, n_with, (int64_t)n_none,
                               (const int32_t *)d_perm, (const int32_t *)d_row_of, g->e_row, g->e_col);
            AMP_LAUNCH_CHECK();
        }
    } else {
        AMP_HIP(hipMemsetAsync(g->e_rowptr, 0, sizeof(int32_t) * ((size_t)n_edge_cols + 1), st));
        if (dev_alloc(&g->e_row, 0) || dev_alloc(&g->e_col, 0)) return 1;
    }
    g->n_with_edge = n_with;
    AMP_HIP(hipStreamSynchronize(st));   // host inputs and scratch die with this scope
    return 0;
}

} // namespace amp

// ---------------------------------------------------------------------------------------------------------------
// Edge list -> athena's CSR on the device: what graphstruc's generate_adjacency (+ add_self_loops)
// hands to the layers, for callers that hold pairs (radius graphs, mesh connectivity).  Conventions of the host
// mirror athena_amd/graph.py (which the tests compare against, element for element):
//   * edge e (1-based column of index_list) contributes (u -> v, id e) and (v -> u, id e); a self edge u == v one entry;
//   * add_self_loops adds (v, v, id 0) for every vertex without a self entry;
//   * inside a row the entries are ordered by edge id, self-loop entries without an id first.
// One stable sort of 64-bit keys  src * (E + 2) + (id + 1 for edges | 0 for an added loop)  does all of it (radix_sort.h, over
// the bits the largest key uses).
// ---------------------------------------------------------------------------------------------------------------
namespace {

__global__ void edge_entries_kernel(int64_t E, const int32_t *__restrict__ pairs /* [2,E] column-major, 1-based */,
                                    int32_t n, unsigned long long stride, unsigned long long *__restrict__ keys,
                                    int32_t *__restrict__ dst, int32_t *__restrict__ has_loop,
                                    unsigned long long *__restrict__ bad)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int32_t u = pairs[2 * e] - 1, v = pairs[2 * e + 1] - 1;
    if (u < 0 || u >= n || v < 0 || v >= n) {
        atomicMin(bad, (unsigned long long)e);
        keys[2 * e] = keys[2 * e + 1] = ~0ull;
        return;
    }
    keys[2 * e] = (unsigned long long)u * stride + (unsigned long long)(e + 2);
    dst[2 * e] = v;
    if (u == v) {
        keys[2 * e + 1] = ~0ull;          // a self edge is one entry
        has_loop[u] = 1;
    } else {
        keys[2 * e + 1] = (unsigned long long)v * stride + (unsigned long long)(e + 2);
        dst[2 * e + 1] = u;
    }
}

__global__ void loop_entries_kernel(int32_t n, int add, unsigned long long stride, const int32_t *__restrict__ has_loop,
                                    unsigned long long *__restrict__ keys, int32_t *__restrict__ dst)
{
    const int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const bool need = add && !has_loop[v];
    keys[v] = need ? (unsigned long long)v * stride + 1ull : ~0ull;   // key 1: before every edge id (>= 2)
    dst[v] = v;
}

// adj_ia (1-based) from the sorted keys; adj_ja[2, nnz] (1-based neighbour, edge id with 0 = none)
__global__ void csr_rows_kernel(int32_t n, const unsigned long long *__restrict__ sorted, int64_t total,
                                unsigned long long stride, int32_t *__restrict__ adj_ia)
{
    const int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v > n) return;
    const unsigned long long key = (unsigned long long)v * stride;
    int64_t lo = 0, hi = total;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (sorted[mid] < key) lo = mid + 1; else hi = mid;
    }
    adj_ia[v] = (int32_t)lo + 1;
}

__global__ void csr_entries_kernel(int64_t nnz, const unsigned long long *__restrict__ sorted,
                                   const int32_t *__restrict__ dst_sorted, unsigned long long stride,
                                   int32_t *__restrict__ adj_ja)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nnz) return;
    const unsigned long long low = sorted[w] % stride;      // 1 = added self loop (id 0), e + 2 for edge e (id e + 1)
    adj_ja[2 * w] = dst_sorted[w] + 1;
    adj_ja[2 * w + 1] = low <= 1 ? 0 : (int32_t)(low - 1);
}

} // namespace

namespace amp {
// the edge list -> CSR pass.  adj_ja_out: host copy of the entries (may be null); keep_ja_dev: when non-null the
// device copy of the entries is handed to the caller (hipFree it) instead of being released
int csr_from_edges_core(int32_t n_vertices, int64_t n_pairs, const int32_t *index_list, int32_t add_self_loops,
                        int32_t *adj_ia_out, int32_t *adj_ja_out, int64_t capacity, int64_t *nnz_out,
                        int32_t **keep_ja_dev)
{
    AMP_REQUIRE(n_vertices >= 0 && n_pairs >= 0 && nnz_out && adj_ia_out && (n_pairs == 0 || index_list),
                "csr_from_edges: bad arguments");
    AMP_REQUIRE(2 * n_pairs + n_vertices < (int64_t)INT32_MAX, "csr_from_edges: more than 2^31 CSR entries");
    hipStream_t st = stream();
    const int64_t total = 2 * n_pairs + n_vertices;
    const unsigned long long stride = (unsigned long long)n_pairs + 2;
    Scratch tmp;
    int32_t *d_pairs = nullptr, *d_dst = nullptr, *d_dst_s = nullptr, *d_dst_t = nullptr, *d_loop = nullptr, *d_ia = nullptr, *d_ja = nullptr;
    unsigned long long *d_keys = nullptr, *d_keys_s = nullptr, *d_keys_t = nullptr, *d_bad = nullptr;
    if (tmp.get(&d_pairs, 2 * (size_t)n_pairs) || tmp.get(&d_dst, total) || tmp.get(&d_dst_s, total) || tmp.get(&d_dst_t, total) ||
        tmp.get(&d_loop, n_vertices) || tmp.get(&d_ia, (size_t)n_vertices + 1) || tmp.get(&d_keys, total) ||
        tmp.get(&d_keys_s, total) || tmp.get(&d_keys_t, total) || tmp.get(&d_bad, 1))
        return 1;
    const unsigned long long none = ~0ull;
    AMP_HIP(hipMemcpyAsync(d_bad, &none, sizeof(none), hipMemcpyHostToDevice, st));
    AMP_HIP(hipMemsetAsync(d_loop, 0, sizeof(int32_t) * (size_t)std::max(n_vertices, 1), st));
    if (n_pairs > 0) {
        AMP_HIP(hipMemcpyAsync(d_pairs, index_list, sizeof(int32_t) * 2 * (size_t)n_pairs, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(edge_entries_kernel, dim3(blocks(n_pairs)), dim3(256), 0, st, n_pairs, (const int32_t *)d_pairs,
                           n_vertices, stride, d_keys, d_dst, d_loop, d_bad);
        AMP_LAUNCH_CHECK();
    }
    if (n_vertices > 0) {
        hipLaunchKernelGGL(loop_entries_kernel, dim3(blocks(n_vertices)), dim3(256), 0, st, n_vertices, add_self_loops,
                           stride, (const int32_t *)d_loop, d_keys + 2 * n_pairs, d_dst + 2 * n_pairs);
        AMP_LAUNCH_CHECK();
    }
    unsigned long long bad = none;
    AMP_HIP(hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, st));
    AMP_HIP(hipStreamSynchronize(st));
    if (bad != none) {
        set_error("csr_from_edges: index_list(:,%llu) = (%d, %d) outside [1,%d]", bad + 1, index_list[2 * bad],
                  index_list[2 * bad + 1], n_vertices);
        return 2;
    }
    int64_t nnz = 0;
    if (total > 0) {
        // invalid slots carry the all-ones key: every pass sees digit 255 of them, so they stay behind every row whatever the
        // number of key bits sorted (the bits of the largest VALID key, n_vertices * stride)
        void *d_temp = nullptr;
        int key_bits = 1;
        while (key_bits < 64 && ((unsigned long long)n_vertices * stride) >> key_bits) ++key_bits;
        if (tmp.get((char **)&d_temp, radix::scratch_bytes(total))) return 1;
        if (int rc = radix::sort_pairs<unsigned long long>((const unsigned long long *)d_keys, (const int32_t *)d_dst, total, key_bits, d_keys_s,
                                                           d_dst_s, d_keys_t, d_dst_t, d_temp, st))
            return rc;
        hipLaunchKernelGGL(csr_rows_kernel, dim3(blocks((int64_t)n_vertices + 1)), dim3(256), 0, st, n_vertices,
                           (const unsigned long long *)d_keys_s, total, stride, d_ia);
        AMP_LAUNCH_CHECK();
    } else {
        const int32_t one = 1;
        AMP_HIP(hipMemcpyAsync(d_ia, &one, sizeof(one), hipMemcpyHostToDevice, st));
    }
    AMP_HIP(hipMemcpyAsync(adj_ia_out, d_ia, sizeof(int32_t) * ((size_t)n_vertices + 1), hipMemcpyDeviceToHost, st));
    AMP_HIP(hipStreamSynchronize(st));
    nnz = (int64_t)adj_ia_out[n_vertices] - 1;     // invalid slots carry the all-ones key and sort behind every row
    *nnz_out = nnz;
    if (adj_ja_out == nullptr && keep_ja_dev == nullptr) return 0;             // size query
    if (adj_ja_out)
        AMP_REQUIRE(capacity >= nnz, "csr_from_edges: adj_ja buffer holds %lld entries, the graph has %lld",
                    (long long)capacity, (long long)nnz);
    if (keep_ja_dev) {
        if (dev_alloc(&d_ja, 2 * (size_t)nnz)) return 1;
        *keep_ja_dev = d_ja;
    } else if (tmp.get(&d_ja, 2 * (size_t)nnz))
        return 1;
    if (nnz > 0) {
        hipLaunchKernelGGL(csr_entries_kernel, dim3(blocks(nnz)), dim3(256), 0, st, nnz, (const unsigned long long *)d_keys_s,
                           (const int32_t *)d_dst_s, stride, d_ja);
        AMP_LAUNCH_CHECK();
        if (adj_ja_out) AMP_HIP(hipMemcpyAsync(adj_ja_out, d_ja, sizeof(int32_t) * 2 * (size_t)nnz, hipMemcpyDeviceToHost, st));
        AMP_HIP(hipStreamSynchronize(st));
    }
    return 0;
}
} // namespace amp

extern "C" int athena_mp_csr_from_edges(int32_t n_vertices, int64_t n_pairs, const int32_t *index_list,
                                        int32_t add_self_loops, int32_t *adj_ia_out, int32_t *adj_ja_out,
                                        int64_t capacity, int64_t *nnz_out)
{
    return amp::csr_from_edges_core(n_vertices, n_pairs, index_list, add_self_loops, adj_ia_out, adj_ja_out, capacity,
                                    nnz_out, nullptr);
}
