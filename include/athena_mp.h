/*
 * athena_mp.h -- C ABI of the MI355X (gfx950) message-passing engine for athena.
 *
 * This is the drop-in boundary.  The reference (nedtaylor/athena
 * v2.1.1) is pure Fortran and has no FFI for this path, so each entry point below
 * names the reference procedure whose arithmetic it replaces (paths relative to
 * src/athena/ of the reference).  The Fortran side binds them through
 * ISO_C_BINDING (athena_amd/fortran/athena_mp_c.f90; INTEGRATION.md shows the
 * layer-side stub).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; the message is
 *     available from athena_mp_last_error() (the Fortran wrapper turns it into
 *     coreutils' stop_program(msg), cf. athena_kipf_msgpass_layer.f90:271-274).
 *   - "dev" pointers are device (HBM) pointers; "host" pointers are host memory.
 *     The *_host variants stage through HBM for callers that only hold
 *     array_type%val (phase-1 plumbing); they are never timed.
 *   - feature tensors are athena's val(F, N) column-major == row-major [N][F]
 *     fp32: one vertex row contiguous.  Dense weights are params(t)%val(:,1):
 *     W(Fo,Fi) column-major flat == row-major Wt[Fi][Fo].
 *   - all kernels are enqueued on the stream set with athena_mp_set_stream
 *     (default: the null stream) and return without synchronising.
 *   - not thread-safe by design (the reference's layers are stateful and
 *     single-threaded, SURVEY.md 8b "Threading").
 *   - ONE stream at a time per device: scratch workspaces (reduction slabs, the GNO partials, packed halo rows) belong
 *     to the device, not to a stream, and are ordered by the stream the calls are enqueued on.  Switching streams with
 *     athena_mp_set_stream is fine once the work enqueued so far on the old stream is ordered before the new stream's
 *     (an event wait or a synchronize); two streams issuing calls for the same device side by side would share those
 *     workspaces.  The second, library-owned stream some calls use inside (athena_mp_gno_aggregate_bwd, the tiled
 *     athena_mp_gno_aggregate_bwd_theta) is forked from and joined to the caller's stream within the call; the
 *     communication stream is forked by the _start calls and joined by their _finish (athena_mp_halo_*, _allreduce_*).
 */
#ifndef ATHENA_MP_H
#define ATHENA_MP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct athena_mp_graph athena_mp_graph; /* opaque: device CSR + transposed CSR + coefficients */

/* activation kinds understood by the fused epilogues and athena_mp_activation_* */
enum {
    ATHENA_MP_ACT_NONE = 0,    /* athena_activation_none.f90    */
    ATHENA_MP_ACT_RELU = 1,    /* athena_activation_relu.f90    */
    ATHENA_MP_ACT_SIGMOID = 2, /* athena_activation_sigmoid.f90 */
    ATHENA_MP_ACT_TANH = 3     /* athena_activation_tanh.f90    */
};

/* ---- runtime ------------------------------------------------------------ */
int athena_mp_init(int device);            /* hipSetDevice + capability check (gfx950) */
/* the device athena_mp_init selected, -1 before the first successful call (after athena_mp_finalize the selection stands): lets a
 * layer that lives inside a host program it does not own initialise the library on first use WITHOUT overriding the host's choice --
 * nothing in the reference corresponds to it (athena has no device); the drop-in types' set_graph calls it (athena_hip_msgpass_layers.f90) */
int athena_mp_initialized(void);
int athena_mp_finalize(void);
const char *athena_mp_last_error(void);
int athena_mp_set_stream(void *hip_stream); /* hipStream_t; NULL = default stream */
int athena_mp_synchronize(void);
int athena_mp_version(void);

/* device memory helpers so a Fortran caller can keep tensors resident */
int athena_mp_malloc(void **dev_ptr, uint64_t bytes);
int athena_mp_free(void *dev_ptr);
int athena_mp_memcpy_h2d(void *dev_dst, const void *host_src, uint64_t bytes);
int athena_mp_memcpy_d2h(void *host_dst, const void *dev_src, uint64_t bytes);
int athena_mp_memset_zero(void *dev_ptr, uint64_t bytes);

/* ---- graph handle ------------------------------------------------------- *
 * Replaces the per-call CSR copies of set_graph_msgpass
 * (athena_msgpass_layer_sub.f90:144-174) and `c%indices = adj_ia; c%adj_ja =
 * adj_ja` (athena_diffstruc_extd_sub_kipf.f90:48-49): built once, reused.
 *   adj_ia  [n_rows+1]  1-based row pointers            (host)
 *   adj_ja  [2, nnz]    column-major; (1,w) neighbour 1-based in [1,n_cols],
 *                       (2,w) edge-feature column 1-based, 0 = none (host)
 *   row_deg / col_deg   optional global degrees for a row partition with halo
 *                       columns (NULL: degree = CSR row length, needs n_rows ==
 *                       n_cols -- exactly the reference's coefficient,
 *                       athena_diffstruc_extd_sub_kipf.f90:39-42)
 */
int athena_mp_graph_create(int32_t n_rows, int32_t n_cols, int64_t nnz, const int32_t *adj_ia,
                           const int32_t *adj_ja, int32_t n_edge_cols, const int32_t *row_deg,
                           const int32_t *col_deg, athena_mp_graph **out);
/* Edge list -> athena's CSR on the device (what graphstruc's generate_adjacency [+ add_self_loops] produces, in the
 * conventions of SURVEY.md Appendix C): index_list [2, n_pairs] column-major, 1-based vertex pairs, edge id = column.
 * Each pair gives (u -> v, id) and (v -> u, id); a self pair one entry; add_self_loops adds (v, v, id 0) where missing;
 * inside a row entries are ordered by edge id, id-less self loops first.  adj_ia_out [n_vertices+1] (1-based),
 * adj_ja_out [2, nnz] column-major; pass adj_ja_out = NULL to query *nnz_out first. */
int athena_mp_csr_from_edges(int32_t n_vertices, int64_t n_pairs, const int32_t *index_list_host,
                             int32_t add_self_loops, int32_t *adj_ia_out_host, int32_t *adj_ja_out_host,
                             int64_t capacity, int64_t *nnz_out);
/* Copies one array of the handle back to the host (4-byte elements; 0-based indices): which =
 * 0 rowptr, 1 col, 2 eid, 3 coef, 4 t_rowptr, 5 t_src, 6 t_eid, 7 t_coef (transposed CSR: the pull form of the
 * reference's scatters), 8 e_rowptr, 9 e_row, 10 e_entry (edge-column index), 11 deg_row, 12 deg_col.
 * host_dst NULL: size query only.  *count = number of elements. */
/* edge list -> CSR -> handle with the entries staying in HBM in between (generate_adjacency [+ add_self_loops]
 * followed by set_graph in one call).  adj_ia_out (n_vertices + 1, host) is always filled; adj_ja_out
 * (2 x capacity, host, column-major) only when non-null.  with_edge_ids = 0 builds a handle without edge-feature
 * columns (Kipf); 1 keeps the pair index as the edge id of both directed entries (Duvenaud / GNO). */
int athena_mp_graph_create_from_edges(int32_t n_vertices, int64_t n_pairs, const int32_t *index_list_host,
                                      int32_t add_self_loops, int32_t with_edge_ids, int32_t *adj_ia_out,
                                      int32_t *adj_ja_out, int64_t capacity, int64_t *nnz_out,
                                      athena_mp_graph **out);
int athena_mp_graph_export(const athena_mp_graph *g, int32_t which, void *host_dst, int64_t capacity,
                           int64_t *count);
int athena_mp_graph_destroy(athena_mp_graph *g);
int athena_mp_graph_dims(const athena_mp_graph *g, int32_t *n_rows, int32_t *n_cols, int64_t *nnz,
                         int32_t *n_edge_cols);
/* What a cached handle of (adj_ia, adj_ja) is valid for -- the key a layer's set_graph compares before it rebuilds
 * (SURVEY.md 8b "Ownership", F12; the reference re-copies the CSR in every set_graph_msgpass,
 * athena_msgpass_layer_sub.f90:144-174, called before every forward, athena_network_sub.f90:2727-2730).
 * 64-bit hash of the sizes and of EVERY word of both arrays (chunks of 2^20 values hashed by up to 8 host threads -- fewer
 * when the process's affinity mask holds fewer cores -- folded in chunk order: the key does not depend on the thread
 * count; about 2 ms for configs[1]'s 80 MB adj_ja on the GPU box's host): an in-place edit is always seen, which is what
 * the reference's copy-per-call guarantees.  Two different graphs of equal n and nnz get different keys (up to 2^-64).
 * ATHENA_MP_GRAPH_KEY_SAMPLED=1 opts into the cheap key for graphs of 2^18 entries and more (head, tail and 4096
 * strided samples of each array): then an in-place edit outside the samples is NOT seen and the caller must announce it
 * (invalidate_graph / graph_type.touch -> athena_mp_graph_evict).  Host-only, no device call, no library state: the
 * Fortran interface is declared `pure`. */
int athena_mp_graph_key(int32_t n_rows, int64_t nnz, const int32_t *adj_ia, const int32_t *adj_ja, uint64_t *key);
/* set_graph as a cache lookup: returns the handle of this CSR (square graph, degrees = row lengths -- what
 * set_graph_msgpass hands a layer), building it only when no handle with the same (device, n, nnz, n_edge_cols,
 * athena_mp_graph_key) is alive.  Handles are shared and reference counted: every layer of a network that is given
 * the same graph gets the same device arrays; a released handle stays cached (least recently used ones are freed once
 * the idle handles hold more than ATHENA_MP_GRAPH_CACHE_ENTRIES entries, default 2^28; 0 = free on the last release)
 * so the mini-batches of the next epoch find theirs.  athena_mp_graph_destroy on an acquired handle releases it.
 * The calling pattern it serves: athena_network_sub.f90:2727-2730 (set_graph before EVERY forward). */
int athena_mp_graph_acquire(int32_t n, int64_t nnz, const int32_t *adj_ia, const int32_t *adj_ja,
                            int32_t n_edge_cols, athena_mp_graph **out);
int athena_mp_graph_release(athena_mp_graph *g);
/* release, and the cache forgets the handle at once: the next athena_mp_graph_acquire of the same key BUILDS from the
 * arrays.  For callers that announce an in-place edit (layer%invalidate_graph, graph_type.touch) -- needed only under
 * the sampled key, harmless otherwise.  Other holders of the handle keep it (old topology) until they release it; the
 * last one frees it.  A handle that never came from the cache is destroyed. */
int athena_mp_graph_evict(athena_mp_graph *g);
/* cached handles alive or idle, lookups served from the cache, and device graph handles BUILT by this process so far
 * (graph_create / _from_edges / _acquire misses / shard blocks): what the cache tests count */
int athena_mp_graph_cache_stats(int64_t *handles, int64_t *hits, int64_t *builds);

/* ---- Kipf --------------------------------------------------------------- */
/* kipf_propagate, athena_diffstruc_extd_sub_kipf.f90:7-59
 *   y[v,:] = sum_w ((deg_v*deg_u)^-1/2) x[u,:]        x [n_cols,F], y [n_rows,F] */
int athena_mp_kipf_propagate_fwd(const athena_mp_graph *g, int32_t F, const float *x_dev, float *y_dev);
/* y = act(kipf_propagate(x)), act in {none, relu, sigmoid, tanh}: the activation of a time step whose dense step ran
 * before the aggregation, applied in the aggregation's store */
int athena_mp_kipf_propagate_act_fwd(const athena_mp_graph *g, int32_t F, const float *x_dev, int32_t act, float *y_dev);
/* get_partial_kipf_propagate_left_val, ..._sub_kipf.f90:85-111
 *   dx[u,:] = sum_{(v,w): ja(1,w)=u} grad[v,:]   (exact=0: the reference, NO coefficient;
 *   exact=1: multiplied by the coefficient, the mathematically exact adjoint) */
int athena_mp_kipf_propagate_bwd(const athena_mp_graph *g, int32_t F, const float *grad_dev,
                                 float *dx_dev, int32_t exact);
/* reverse_kipf_propagate, ..._sub_kipf.f90:116-158 (the node the higher-order path builds; grad_reverse does not
 * reach it):   c[u,:] = sum_{(v,w): ja(1,w)=u} a[v,:]   -- the coefficient-free scatter again, a [n_rows,F] -> c [n_cols,F].
 * Its partials: the FUNCTION form get_partial_left_reverse_kipf_propagate (:159-175) is kipf_propagate(upstream)
 * (with the coefficient), upstream [n_cols,F] -> out [n_rows,F]; the _val form (:176-205) is the same scatter as the
 * op itself.  Thin named entry points over the two kernels above. */
int athena_mp_reverse_kipf_propagate_fwd(const athena_mp_graph *g, int32_t F, const float *a_dev, float *c_dev);
int athena_mp_reverse_kipf_propagate_partial(const athena_mp_graph *g, int32_t F, const float *upstream_dev,
                                             float *out_dev);
int athena_mp_reverse_kipf_propagate_partial_val(const athena_mp_graph *g, int32_t F, const float *upstream_dev,
                                                 float *out_dev);
/* both reverse forms from ONE gather of the upstream rows: dx_plain = the reference's coefficient-free scatter
 * (exact = 0 above), dx_coef = the adjoint of the forward (exact = 1).  What a layer step needs when it applies
 * the dense step BEFORE the aggregation (F_out < F_in): dX = dx_plain . W,  dW = dx_coef^T . X */
int athena_mp_kipf_propagate_bwd_dual(const athena_mp_graph *g, int32_t F, const float *grad_dev,
                                      float *dx_plain_dev, float *dx_coef_dev);
/* the same pair in pull form over the FORWARD rows (a row shard of an undirected graph, whose transposed rows are
 * its own rows): y_plain[v] = sum_w x[col w], y_coef[v] = sum_w coef_w x[col w]; x has n_cols rows (local + halo) */
int athena_mp_kipf_propagate_fwd_dual(const athena_mp_graph *g, int32_t F, const float *x_dev,
                                      float *y_plain_dev, float *y_coef_dev);

/* out[r,:] = x[idx[r],:]  (r < n; idx 0-based device array).  Packs the halo rows a row partition
 * sends to its peers; same gather kernel as the aggregation. */
int athena_mp_gather_rows(int64_t n, int32_t F, const int32_t *idx_dev, const float *x_dev, float *out_dev);

/* ---- dense contraction (diffstruc matmul at athena_kipf_msgpass_layer.f90:951,
 *      athena_duvenaud_msgpass_layer.f90:842, athena_graph_nop_layer.f90:761) -- fp32 MFMA */
/* Z[N,Fo] = act( P[N,Fi] . Wt[Fi,Fo] (+ bias[Fo]) );  bias may be NULL */
int athena_mp_gemm_fwd(int64_t N, int32_t Fi, int32_t Fo, const float *P_dev, const float *W_dev,
                       const float *bias_dev, int32_t act, float *Z_dev);
/* dW(Fo,Fi) = dZ^T-contract: dWt[i,o] = sum_v P[v,i] dZ[v,o] */
int athena_mp_gemm_dw(int64_t N, int32_t Fi, int32_t Fo, const float *P_dev, const float *dZ_dev,
                      float *dW_dev);
/* dP[N,Fi] = dZ[N,Fo] . W  (dP[v,i] = sum_o dZ[v,o] Wt[i,o]) */
int athena_mp_gemm_dx(int64_t N, int32_t Fi, int32_t Fo, const float *dZ_dev, const float *W_dev,
                      float *dP_dev);

/* ---- fused Kipf layer step (one launch: aggregation + dense contraction, W resident in LDS) ------
 * update_message_kipf, athena_kipf_msgpass_layer.f90:943-952, one time step:
 *   P = kipf_propagate(X) (returned: the reverse pass needs it for dW; P_dev may be NULL when the reverse pass is
 *   athena_mp_kipf_layer_bwd, which works from X),  Z = act(P . Wt + bias)
 * One launch at 64 / 128 / 256 features on graphs without hub rows -- except on BANDED graphs (a block-diagonal batch of small
 * graphs: every neighbour within 32 rows, rows of at most 8 entries), where the aggregation gathers from LDS: one launch of its
 * own at 64 -> 64 (banded_fused.hip: the tile goes from LDS into the dense step), the gather and the dense step as two launches at
 * 128 features (profiles/r06_kipf_banded_ab.txt); same P bit for bit, Z within the order of the dense step's sums.  _bwd_x likewise. */
int athena_mp_kipf_layer_fwd(const athena_mp_graph *g, int32_t Fi, int32_t Fo, const float *x_dev,
                             const float *W_dev, const float *bias_dev, int32_t act, float *P_dev,
                             float *Z_dev);
/* reverse of the same step wrt its input: dX = A^T (dZ . W) evaluated as (A^T dZ) . W
 * (dZ = gradient at the pre-activation; exact as in athena_mp_kipf_propagate_bwd) */
int athena_mp_kipf_layer_bwd_x(const athena_mp_graph *g, int32_t Fi, int32_t Fo, const float *dZ_dev,
                               const float *W_dev, int32_t exact, float *dX_dev);

/* the whole reverse pass of one step from the step's INPUT X (no stored P): dX as above (may be NULL) and
 *   dW = dZ . P^T with P = A^ X (matmul reverse wrt params(t), athena_kipf_msgpass_layer.f90:951), evaluated as
 *   sum_u (A^^T dZ)[u] (x) X[u] -- both sums come from ONE gather of dZ over the transposed CSR.  With it the forward call
 *   may pass P_dev = NULL.  128 -> 128 without hub rows: one launch; otherwise a dual gather + two contractions. */
int athena_mp_kipf_layer_bwd(const athena_mp_graph *g, int32_t Fi, int32_t Fo, const float *dZ_dev,
                             const float *W_dev, const float *X_dev, int32_t exact, float *dX_dev, float *dW_dev);

/* same contraction over the FORWARD rows of a (rectangular) shard graph: dX[v,:] = (sum_w [coef] dZ[col[w],:]) . W
 * -- the backward of a row partition whose rows list their sources (athena_amd/dist.py) */
int athena_mp_pull_gemm(const athena_mp_graph *g, int32_t Fi, int32_t Fo, const float *dZ_dev,
                        const float *W_dev, int32_t exact, float *dX_dev);

/* ---- element-wise brackets of the path -------------- */
int athena_mp_activation_fwd(int32_t act, int64_t n, const float *z_dev, float *y_dev);
int athena_mp_activation_bwd(int32_t act, int64_t n, const float *y_dev, const float *g_dev, float *dz_dev);
int athena_mp_axpy(int64_t n, float alpha, const float *x_dev, float *y_dev); /* y += alpha x */
/* dst = src on the device, one 16-byte element per thread in launch order (stream-ordered; 16-byte aligned pointers): the
 * copy form that reaches the HBM's streaming ceiling -- bench.py times it on 1 GiB and prints the rate beside the roofline
 * as the measured ceiling of the box it ran on (SURVEY.md 8d "report the measured stream ceiling alongside") */
int athena_mp_device_copy(void *dst_dev, const void *src_dev, uint64_t bytes);

/* ---- Duvenaud ------------------------------------------------------------ */
/* duvenaud_propagate, athena_diffstruc_extd_sub_duvenaud.f90:7-59
 *   c[v,:] = sum_w [ x[u,:] ; e[ja(2,w),:] ]     c [n_rows, Fv+Fe]; edge id 0 => zero vector */
int athena_mp_duvenaud_propagate_fwd(const athena_mp_graph *g, int32_t Fv, int32_t Fe,
                                     const float *x_dev, const float *e_dev, float *c_dev);
/* get_partial_duvenaud_propagate_left_val :115-141 / _right_val :143-171.  grad is [n_rows, Fv + Fe]; _bwd_x with Fe = 0 takes
 * the vertex part alone [n_rows, Fv], _bwd_e with Fv = 0 the edge part alone [n_rows, Fe] (athena_mp_duvenaud_update_bwd_split) */
int athena_mp_duvenaud_propagate_bwd_x(const athena_mp_graph *g, int32_t Fv, int32_t Fe,
                                       const float *grad_dev, float *dx_dev);
int athena_mp_duvenaud_propagate_bwd_e(const athena_mp_graph *g, int32_t Fv, int32_t Fe,
                                       const float *grad_dev, float *de_dev);
/* duvenaud_update :176-228;  weight [Fo*Fi*(max_deg-min_deg+1)] packed per degree bucket */
int athena_mp_duvenaud_update_fwd(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t min_deg,
                                  int32_t max_deg, const float *a_dev, const float *weight_dev,
                                  float *c_dev);
/* the same with the message activation applied in the epilogue (athena_duvenaud_msgpass_layer.f90
 * :790-803: duvenaud_update then activation%apply): z = act(W_d (a/d)); only z is written */
int athena_mp_duvenaud_update_act_fwd(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t min_deg,
                                      int32_t max_deg, const float *a_dev, const float *weight_dev,
                                      int32_t act, float *z_dev);
/* get_partial_duvenaud_update_val :284-324 */
int athena_mp_duvenaud_update_bwd_a(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t min_deg,
                                    int32_t max_deg, const float *grad_dev, const float *weight_dev,
                                    float *da_dev);
/* get_partial_duvenaud_update_weight_val :326-368 */
int athena_mp_duvenaud_update_bwd_w(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t min_deg,
                                    int32_t max_deg, const float *grad_dev, const float *a_dev,
                                    float *dweight_dev);
/* update + activation + the per-vertex part of the readout, p = softmax_over_outputs(R z) (update_readout_duvenaud,
 * athena_duvenaud_msgpass_layer.f90:838-855), in one launch; z [n_rows, Fo], R = params(T+t)%val(:,1) = R(O, Fo) flat,
 * p [n_rows, O].  The per-graph sums: athena_mp_segment_sum(O, n_rows, S, seg, p, out, accumulate). */
int athena_mp_duvenaud_update_readout_fwd(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t min_deg, int32_t max_deg,
                                          const float *a_dev, const float *weight_dev, int32_t act, float *z_dev, int32_t O,
                                          const float *R_dev, float *p_dev);
/* both partials above from ONE pass over grad (the pair is bound by the bytes it moves: 1.95 GB instead of 2.55 GB at
 * configs[2]); da [n_rows, Fi], dweight [Fo*Fi*D].  Shapes outside the fused kernel run the two entry points above. */
int athena_mp_duvenaud_update_bwd(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t min_deg, int32_t max_deg,
                                  const float *grad_dev, const float *a_dev, const float *weight_dev, float *da_dev,
                                  float *dweight_dev);
/* the same pair with da SPLIT where it is written: da_x [n_rows, Fv] and da_e [n_rows, Fe] instead of [n_rows, Fv + Fe].  The
 * reverse kernel visits the vertices in bucket order; rows of 288 bytes (64 + 8 floats) written in that order are partial
 * cache lines, and the propagate reverse that follows gathers its edge part as 32-byte slivers of them.  Split, the vertex
 * part is 256-byte rows and the edge part a dense array: athena_mp_duvenaud_propagate_bwd_x(g, Fv, 0, da_x, dx) and
 * athena_mp_duvenaud_propagate_bwd_e(g, 0, Fe, da_e, de) take the two halves (same sums, same order: bit-identical dx / de).
 * One launch at F_v = 64 and the fused kernel's widths (BASELINE configs[2]); other shapes go through the packed form. */
int athena_mp_duvenaud_update_bwd_split(const athena_mp_graph *g, int32_t Fv, int32_t Fe, int32_t Fo, int32_t min_deg,
                                        int32_t max_deg, const float *grad_dev, const float *a_dev, const float *weight_dev,
                                        float *da_x_dev, float *da_e_dev, float *dweight_dev);

/* One time step of the Duvenaud layer's reverse pass in one call: the readout's reverse (softmax over the outputs ->
 * matmul(R, z) -> the message activation; athena_duvenaud_msgpass_layer.f90:838-855 through diffstruc's grad_reverse,
 * athena_diffstruc_extd_sub.f90:309-313) and the update's reverse (get_partial_duvenaud_update_val / _weight_val,
 * athena_diffstruc_extd_sub_duvenaud.f90:284-368) -- what athena_mp_duvenaud_readout_bwd followed by
 * athena_mp_duvenaud_update_bwd_split compute, without dc [n_rows, Fv] between them where the shape allows it (F_v = 64,
 * F_v + F_e <= 96, O <= 16: ONE launch); other shapes run the two through a workspace.  Device pointers.
 * z [n_rows, Fv] the ACTIVATED update output of this time step, p [n_rows, O] its softmax(R z), gout [S, O] the gradient of the
 * per-graph readout, dz_next [n_rows, Fv] or NULL the next time step's gradient with respect to z, a [n_rows, Fv + Fe],
 * weight as athena_mp_duvenaud_update_fwd; out: da_x [n_rows, Fv], da_e [n_rows, Fe], dweight, dR [O, Fv] flat o + O f
 * (accumulate_dR != 0: added to; accumulate_da_e != 0: da_e is added to as well -- get_partial_duvenaud_propagate_right_val,
 * athena_diffstruc_extd_sub_duvenaud.f90:143-171, is linear in its upstream, so a layer that owns its reverse pass sums da_e over
 * its time steps and scatters the sum to the edge features once).  a_e != NULL: a arrives split as
 * athena_mp_duvenaud_update_readout_fwd_split takes it -- a = a_x [n_rows, Fv], a_e [n_rows, Fe]. */
int athena_mp_duvenaud_readout_update_bwd(const athena_mp_graph *g, int32_t Fv, int32_t Fe, int32_t min_deg, int32_t max_deg,
                                          int32_t O, int32_t S, const int32_t *seg, const float *z, const float *R, const float *p,
                                          const float *gout, const float *dz_next, int32_t act, const float *a,
                                          const float *weight, float *da_x, float *da_e, float *dweight, float *dR,
                                          int32_t accumulate_dR, int32_t accumulate_da_e, const float *a_e);

/* athena_mp_duvenaud_update_readout_fwd with a split where duvenaud_propagate (athena_diffstruc_extd_sub_duvenaud.f90:7-59)
 * concatenates: a_x [n_rows, Fv] = the neighbour sums of the vertex features (athena_mp_duvenaud_propagate_fwd with Fe = 0),
 * a_e [n_rows, Fe] = those of the edge features (athena_mp_duvenaud_propagate_fwd with Fv = 0).  The edge features of a layer do
 * not change from time step to time step (update_message_duvenaud, athena_duvenaud_msgpass_layer.f90:755-836, passes the same
 * edge_features to every duvenaud_propagate), so a layer that owns its tape gathers a_e once.  One launch at F_v = F_o = 64,
 * F_e <= 16, O <= 16; other shapes pack a into a workspace and take athena_mp_duvenaud_update_readout_fwd.  Device pointers. */
int athena_mp_duvenaud_update_readout_fwd_split(const athena_mp_graph *g, int32_t Fv, int32_t Fe, int32_t Fo, int32_t min_deg,
                                                int32_t max_deg, const float *a_x, const float *a_e, const float *weight,
                                                int32_t act, float *z, int32_t O, const float *R, float *p);
/* readout, athena_duvenaud_msgpass_layer.f90:838-855 over a block-diagonal batch:
 *   p[v,:] = softmax_over_outputs(logits[v,:]); out[s,:] (+)= sum_{v in seg s} p[v,:]
 *   seg_dev [S+1] 0-based vertex offsets of the graphs */
int athena_mp_softmax_segsum_fwd(int32_t O, int64_t N, int32_t S, const int32_t *seg_dev,
                                 const float *logits_dev, float *p_dev, float *out_dev,
                                 int32_t accumulate);
/* dlogits[v,:] = p (g_s - <g_s,p>) for v in segment s */
int athena_mp_softmax_segsum_bwd(int32_t O, int64_t N, int32_t S, const int32_t *seg_dev,
                                 const float *p_dev, const float *gout_dev, float *dlogits_dev);

/* readout with a readout activation other than softmax: activate with athena_mp_activation_* / _swish_*, then
 * out[s,:] (+)= sum_{v in seg s} p[v,:];  reverse: dp[v,:] = gout[graph of v,:] */
int athena_mp_segment_sum(int32_t O, int64_t N, int32_t S, const int32_t *seg_dev, const float *p_dev,
                          float *out_dev, int32_t accumulate);
int athena_mp_segment_sum_bwd(int32_t O, int64_t N, int32_t S, const int32_t *seg_dev, const float *gout_dev,
                              float *dp_dev);
/* The same readout with the logits contraction fused in (one launch per direction, logits and
 * dlogits never reach HBM).  R is the readout matrix R(O,Fv) in Fortran order, z [N,Fv] the step's
 * vertex features:   p = softmax(z R^T) per vertex;  out[s,:] (+)= sum_{v in s} p[v,:]            */
int athena_mp_duvenaud_readout_fwd(int64_t N, int32_t Fv, int32_t O, int32_t S, const int32_t *seg_dev,
                                   const float *z_dev, const float *R_dev, float *p_dev, float *out_dev,
                                   int32_t accumulate);
/* reverse of readout + the message activation that produced z (athena_duvenaud_msgpass_layer.f90
 * :790-803 then :838-855, reversed):  dl = p (g_s - <g_s,p>);  dR (+)= dl^T z;
 *   dc = act'(z) * (dl R + dz_next)      dz_next: gradient arriving from step t+1, or NULL       */
int athena_mp_duvenaud_readout_bwd(int64_t N, int32_t Fv, int32_t O, int32_t S, const int32_t *seg_dev,
                                   const float *z_dev, const float *R_dev, const float *p_dev,
                                   const float *gout_dev, const float *dz_next_dev, int32_t act,
                                   float *dc_dev, float *dR_dev, int32_t accumulate);

/* ---- Graph neural operator ------------------------------------------------
 * gno_kernel_eval + gno_aggregate, athena_diffstruc_extd_sub_nop.f90:26-115, :330-397,
 * re-associated so the [Fo*Fi, E] edge-kernel tensor is never materialised:
 *   m[i,:] = sum_{(j,e) in row i} reshape(V relu(U dx_e + b_u) + b_v,[Fo,Fi]) x[j,:]
 * theta = [U(H,d) | b_u(H) | V(Fo*Fi,H) | b_v(Fo*Fi)]  (:74-82);  coords [E,d]. */
int athena_mp_gno_aggregate_fwd(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                                const float *theta_dev, const float *coords_dev, const float *x_dev,
                                float *m_dev);
/* gradients (:137-216, :235-325, :419-458, :480-526 composed through the same re-association) */
int athena_mp_gno_aggregate_bwd_x(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                                  const float *theta_dev, const float *coords_dev,
                                  const float *grad_dev, float *dx_dev);
/* get_partial_gno_agg_features_val (:419-458) in PULL form over the graph's own rows:
 *   dx[v,:] = sum_{w in row v} K_{eid[w]}^T grad_ext[col[w],:]      dx [n_rows, Fi], grad_ext [n_cols, Fo]
 * equal to the scatter of athena_mp_gno_aggregate_bwd_x on an undirected graph whose two directions share one edge column
 * (athena's graphs, :369-376) -- the form a row block of a partitioned graph evaluates after the halo exchange of grad */
int athena_mp_gno_aggregate_bwd_x_pull(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                                       const float *theta_dev, const float *coords_dev,
                                       const float *grad_ext_dev, float *dx_dev);
int athena_mp_gno_aggregate_bwd_theta(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi,
                                      int32_t Fo, const float *theta_dev, const float *coords_dev,
                                      const float *x_dev, const float *grad_dev, float *dtheta_dev);
int athena_mp_gno_aggregate_bwd_coords(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi,
                                       int32_t Fo, const float *theta_dev, const float *coords_dev,
                                       const float *x_dev, const float *grad_dev, float *dcoords_dev);
/* Training-mode pair (the reference keeps every forward intermediate on its tape for grad_reverse:
 * gno_aggregate's result node holds kappa [Fo*Fi, E], athena_diffstruc_extd_sub_nop.f90:330-397 -- 246 GB at
 * BASELINE configs[3]).  Here the forward pass may keep the re-associated S = sum_e [h_e;1] x_j^T per vertex
 * (n_tiles * 133120 floats: 33 GB at configs[3], sized for 288 GB of HBM) so that the reverse pass's
 * dVaug = S^T g streams it instead of rebuilding it.  Same m, same dtheta (bit for bit) as the pair above.
 *   _saved_bytes: *bytes = size of s_save_dev for this graph and shape, 0 if the shape does not take the
 *                 kernels that keep S (then use the pair above);
 *   _fwd_save:    athena_mp_gno_aggregate_fwd that also fills s_save_dev;
 *   _bwd_theta_saved: athena_mp_gno_aggregate_bwd_theta reading the s_save_dev of the SAME graph, theta, coords, x. */
int athena_mp_gno_saved_bytes(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                              int64_t *bytes);
int athena_mp_gno_aggregate_fwd_save(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                                     const float *theta_dev, const float *coords_dev, const float *x_dev,
                                     float *m_dev, float *s_save_dev);
int athena_mp_gno_aggregate_bwd_theta_saved(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi,
                                            int32_t Fo, const float *theta_dev, const float *coords_dev,
                                            const float *x_dev, const float *grad_dev,
                                            const float *s_save_dev, float *dtheta_dev);

/* The whole reverse pass of gno_aggregate in ONE call, from one G = g . Vmat^T: dx (get_partial_gno_agg_features_val,
 * athena_diffstruc_extd_sub_nop.f90:419-458), dtheta (:480-526 composed with get_partial_gno_kernel_params_val :235-325) and,
 * on request, dcoords (:137-216).  Any of the three outputs may be NULL; s_save_dev is the S of
 * athena_mp_gno_aggregate_fwd_save (NULL: S is rebuilt).  The separate entry points above compute G twice -- once inside
 * the dx launch (as T . B2), once for the kernel MLP's gradient; here the kernel that holds a piece of G_i in LDS also
 * emits every entry's partial h_e^T G_i of dx, and a gather over the transposed CSR sums them.  Shapes
 * outside the fused kernels run the separate entry points; *fused_out (may be NULL) reports which it was.  Same values
 * as the separate entry points to fp32 rounding (the sums associate differently).  The per-entry partials live in a
 * library workspace of nnz * 256 bytes (7.6 GB at BASELINE configs[3]; grown on demand, reused by every call, released
 * by athena_mp_finalize), the transposed-entry map (nnz int32) in the graph handle.
 * STREAMS: with dtheta requested, the gather of the partials runs on a second, library-owned stream beside the S^T g launch
 * (fork after the kernel MLP's launches, join before the call returns: the caller's stream waits for it, so the outputs are
 * ordered on the caller's stream like any other call's, and the pattern can be captured into a HIP graph). */
int athena_mp_gno_aggregate_bwd(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                                const float *theta_dev, const float *coords_dev, const float *x_dev,
                                const float *grad_dev, const float *s_save_dev, float *dx_dev, float *dtheta_dev,
                                float *dcoords_dev, int32_t *fused_out);

/* ---- activations with their own shape and the 'concatenate' merge (8f-2) -----
 * swish_array / get_partial_swish_val, athena_diffstruc_extd_sub.f90:424-492: y = x/(1+exp(-beta x));
 * the reverse pass differentiates at the INPUT x */
int athena_mp_swish_fwd(int64_t n, float beta, const float *x_dev, float *y_dev);
int athena_mp_swish_bwd(int64_t n, float beta, const float *x_dev, const float *grad_dev, float *dx_dev);
/* the activations that carry attributes (athena_activation_{linear,relu,sigmoid,tanh,leaky_relu,selu,gaussian,
 * piecewise}.f90 `apply`): y = f(x; p0, p1) * scale; the reverse pass differentiates at the INPUT x.
 *   kind        p0          p1       f
 *   LINEAR      -           -        x                                            (linear.f90:194)
 *   RELU        threshold   -        max(x, threshold)                            (relu.f90:201)
 *   SIGMOID     -           -        1/(1+exp(-x))
 *   TANH        -           -        tanh(x)
 *   LEAKY_RELU  alpha       -        max(x*alpha, x)                              (leaky_relu.f90:204)
 *   SELU        alpha       lambda   x>0 ? lambda x : alpha lambda (exp(x)-1)     (selu.f90:227-229)
 *   GAUSSIAN    sigma       mu       exp(-((x-mu)/sigma)^2/2) / (sqrt(2 pi) sigma) (gaussian.f90:8,221)
 *   PIECEWISE   gradient    limit    piecewise_array, athena_diffstruc_extd_sub.f90:216-254; its reverse factor
 *                                    follows get_partial_piecewise_val :275-290 AS WRITTEN (the test
 *                                    `x <= limit .or. x >= -limit` holds everywhere for limit >= 0, so the
 *                                    gradient passes through unchanged -- DESIGN.md 3.7) */
typedef enum {
    ATHENA_MP_ACTP_LINEAR = 0,
    ATHENA_MP_ACTP_RELU = 1,
    ATHENA_MP_ACTP_SIGMOID = 2,
    ATHENA_MP_ACTP_TANH = 3,
    ATHENA_MP_ACTP_LEAKY_RELU = 4,
    ATHENA_MP_ACTP_SELU = 5,
    ATHENA_MP_ACTP_GAUSSIAN = 6,
    ATHENA_MP_ACTP_PIECEWISE = 7
} athena_mp_activation_param;
int athena_mp_activation_param_fwd(int32_t kind, int64_t n, float scale, float p0, float p1, const float *x_dev,
                                   float *y_dev);
int athena_mp_activation_param_bwd(int32_t kind, int64_t n, float scale, float p0, float p1, const float *x_dev,
                                   const float *grad_dev, float *dx_dev);
/* softmax(val, dim=2): over the F features of each vertex (athena_activation_softmax.f90:183-203,
 * athena_diffstruc_extd_sub.f90:295-379); reverse: dz = y*g - y*sum(y*g) */
int athena_mp_softmax_fwd(int64_t N, int32_t F, const float *z_dev, float *y_dev);
int athena_mp_softmax_bwd(int64_t N, int32_t F, const float *y_dev, const float *grad_dev, float *dz_dev);
/* network%add(layer, input_list=[..], operator='concatenate') (example/msgpass_euler/src/main.f90:192-255):
 * out[v,:] = [a[v,:], b[v,:]]; the reverse pass splits grad (da_dev or db_dev may be NULL) */
int athena_mp_concat_fwd(int64_t N, int32_t Fa, int32_t Fb, const float *a_dev, const float *b_dev,
                         float *out_dev);
int athena_mp_concat_bwd(int64_t N, int32_t Fa, int32_t Fb, const float *grad_dev, float *da_dev,
                         float *db_dev);

/* ---- tail of a train step, device resident -----------------
 * compute_mse, athena_loss.f90:393-430:  *loss_dev = mean((p-e)^2)/2 ; dpred = (p-e)/n (may be NULL) */
int athena_mp_mse_loss(int64_t n, const float *pred_dev, const float *expected_dev, float *loss_dev,
                       float *dpred_dev);
/* apply_clip, athena_clipper.f90:165-210 on the flat gradient vector (athena_network_sub.f90:2903) */
int athena_mp_clip(int64_t n, float *grad_dev, int32_t l_min_max, float clip_min, float clip_max,
                   int32_t l_norm, float clip_norm);
/* minimise_sgd, athena_optimiser.f90:634-673.  reg_kind 0 none / 1 l1 / 2 l2 / 3 l1l2
 * (athena_regulariser.f90:85-138); grad is overwritten with -lr*grad as the reference does */
int athena_mp_sgd_step(int64_t n, float lr, float momentum, int32_t nesterov, int32_t reg_kind, float l1,
                       float l2, float *param_dev, float *grad_dev, float *velocity_dev);
/* minimise_adam, athena_optimiser.f90:1027-1091; iter >= 1 is optimiser%iter AFTER the increment of
 * network%update; decoupled selects the AdamW branch of the l2 regulariser */
int athena_mp_adam_step(int64_t n, float lr, float beta1, float beta2, float epsilon, int32_t iter,
                        int32_t reg_kind, float l1, float l2, int32_t decoupled, float *param_dev,
                        float *grad_dev, float *m_dev, float *v_dev);

/* ---- host-pointer staging variants (phase-1 Fortran callbacks) ------------- */
int athena_mp_kipf_propagate_fwd_host(const athena_mp_graph *g, int32_t F, const float *x_host, float *y_host);
int athena_mp_kipf_propagate_bwd_host(const athena_mp_graph *g, int32_t F, const float *grad_host,
                                      float *dx_host, int32_t exact);
int athena_mp_gemm_fwd_host(int64_t N, int32_t Fi, int32_t Fo, const float *P_host, const float *W_host,
                            const float *bias_host, int32_t act, float *Z_host);

/* more host-pointer staging variants (same semantics as the device entry points above; host.hip) */
int athena_mp_gemm_dw_host(int64_t N, int32_t Fi, int32_t Fo, const float *P, const float *dZ, float *dW);
int athena_mp_gemm_dx_host(int64_t N, int32_t Fi, int32_t Fo, const float *dZ, const float *W, float *dP);
int athena_mp_activation_fwd_host(int32_t act, int64_t n, const float *z, float *y);
int athena_mp_activation_bwd_host(int32_t act, int64_t n, const float *y, const float *g, float *dz);
int athena_mp_duvenaud_propagate_fwd_host(const athena_mp_graph *g, int32_t Fv, int32_t Fe, const float *x, const float *e, float *c);
int athena_mp_duvenaud_propagate_bwd_x_host(const athena_mp_graph *g, int32_t Fv, int32_t Fe, const float *grad, float *dx);
int athena_mp_duvenaud_propagate_bwd_e_host(const athena_mp_graph *g, int32_t Fv, int32_t Fe, const float *grad, float *de);
int athena_mp_duvenaud_update_fwd_host(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t mn, int32_t mx, const float *a, const float *w, float *c);
int athena_mp_duvenaud_update_bwd_a_host(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t mn, int32_t mx, const float *grad, const float *w, float *da);
int athena_mp_duvenaud_update_bwd_w_host(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t mn, int32_t mx, const float *grad, const float *a, float *dw);
int athena_mp_softmax_segsum_fwd_host(int32_t O, int64_t N, int32_t S, const int32_t *seg, const float *logits, float *p, float *out, int32_t accumulate);
int athena_mp_softmax_segsum_bwd_host(int32_t O, int64_t N, int32_t S, const int32_t *seg, const float *p, const float *gout, float *dlogits);
int athena_mp_gno_aggregate_fwd_host(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo, const float *theta, const float *coords, const float *x, float *m);
int athena_mp_gno_aggregate_bwd_x_host(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo, const float *theta, const float *coords, const float *grad, float *dx);
int athena_mp_gno_aggregate_bwd_theta_host(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo, const float *theta, const float *coords, const float *x, const float *grad, float *dtheta);
int athena_mp_gno_aggregate_bwd_coords_host(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo, const float *theta, const float *coords, const float *x, const float *grad, float *dcoords);

/* host-pointer variants of the composites and shaped activations */
int athena_mp_duvenaud_update_act_fwd_host(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t mn, int32_t mx, const float *a, const float *w, int32_t act, float *z);
int athena_mp_duvenaud_readout_fwd_host(int64_t N, int32_t Fv, int32_t O, int32_t S, const int32_t *seg, const float *z, const float *R, float *p, float *out, int32_t accumulate);
int athena_mp_duvenaud_readout_bwd_host(int64_t N, int32_t Fv, int32_t O, int32_t S, const int32_t *seg, const float *z, const float *R, const float *p, const float *gout, const float *dz_next, int32_t act, float *dc, float *dR, int32_t accumulate);
int athena_mp_softmax_fwd_host(int64_t N, int32_t F, const float *z, float *y);
int athena_mp_softmax_bwd_host(int64_t N, int32_t F, const float *y, const float *g, float *dz);
int athena_mp_swish_fwd_host(int64_t n, float beta, const float *x, float *y);
int athena_mp_swish_bwd_host(int64_t n, float beta, const float *x, const float *g, float *dx);
int athena_mp_activation_param_fwd_host(int32_t kind, int64_t n, float scale, float p0, float p1, const float *x,
                                        float *y);
int athena_mp_activation_param_bwd_host(int32_t kind, int64_t n, float scale, float p0, float p1, const float *x,
                                        const float *g, float *dx);

/* ---- the FUSED entry points behind diffstruc's one-partial-at-a-time callbacks (host pointers) ------------------------
 * update_message_duvenaud / update_readout_duvenaud (athena_duvenaud_msgpass_layer.f90:790-803, 838-855) in one launch:
 * z = act(duvenaud_update(a, w)) and the readout's per-vertex p = softmax(R z) -- host form of
 * athena_mp_duvenaud_update_readout_fwd; z becomes the value of the activation node, p the value of the readout's softmax node. */
int athena_mp_duvenaud_update_readout_fwd_host(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t mn, int32_t mx,
                                               const float *a, const float *w, int32_t act, float *z, int32_t O,
                                               const float *R, float *p);
/* grad_reverse asks a two-operand node for its partials ONE AT A TIME, each through a `pure` callback with an intent(in)
 * node (get_partial_left_val / get_partial_right_val: athena_diffstruc_extd_sub_duvenaud.f90:284-368,
 * athena_diffstruc_extd_sub_nop.f90:419-526) -- the callback can keep nothing, while the fused reverse kernels produce BOTH
 * partials from one pass over the upstream gradient.  A *_pair_host entry point computes both on the first request, returns
 * the one asked for (`which`) and parks the other on the device; the second request gets the parked one if and only if it
 * names the same graph handle, the same shapes and operand arrays with the same CONTENT (a 64-bit hash of every byte of each
 * operand, or the residency table's (id, version) for an array that lives on the device).  A request that does not match
 * recomputes.  A parked partial is handed over once.
 *   athena_mp_duvenaud_update_bwd_pair_host   which 0: da [n_rows, Fi] | 1: dweight [Fo Fi D]     (athena_mp_duvenaud_update_bwd)
 *                                             act != ATHENA_MP_ACT_NONE: the node is update + message activation in one (value
 *                                             z = act(duvenaud_update(a, w))); grad is then the gradient w.r.t. z and act'(z)
 *                                             is applied on the device first (z_or_null = z [n_rows, Fo])
 *   athena_mp_gno_aggregate_bwd_pair_host     which 0: dx [n_cols, Fi] | 1: dtheta | 2: dcoords [n_edge_cols, d]
 *                                             (athena_mp_gno_aggregate_bwd; dcoords is computed only when asked for)
 *   athena_mp_pair_stats                      fused passes run / partials handed over from a slot (tests, diagnostics) */
int athena_mp_duvenaud_update_bwd_pair_host(const athena_mp_graph *g, int32_t Fi, int32_t Fo, int32_t mn, int32_t mx,
                                            int32_t act, const float *z_or_null, const float *grad, const float *a,
                                            const float *w, int32_t which, float *out);
int athena_mp_gno_aggregate_bwd_pair_host(const athena_mp_graph *g, int32_t d, int32_t H, int32_t Fi, int32_t Fo,
                                          const float *theta, const float *coords, const float *x, const float *grad,
                                          int32_t which, float *out);
int athena_mp_pair_stats(int64_t *fused_passes, int64_t *handed_over);

/* ---- multi-GPU: communicator, row-partition shard, halo exchange (comm.hip) -----------------------------------------
 * The reference has no distributed code; these are the entry points SURVEY.md 8b/8e ask the
 * boundary to export (athena_mp_comm_create / graph_partition / halo_exchange) so that a Fortran host reaches the
 * 8-GPU Kipf step through ISO_C_BINDING, one process per GPU.  Transport: RCCL over xGMI -- grouped ncclSend /
 * ncclRecv to every peer on a communication stream, event-ordered against the compute stream; ncclAllReduce for dW.
 * (ATHENA_MP_COMM_TRANSPORT=shm swaps in a host-staged TEST transport for one-GPU boxes; never the default.)
 * What the sharded step replaces: the per-sample loop of update_message_kipf, athena_kipf_msgpass_layer.f90:940-957,
 * run on the rank's rows. */
typedef struct athena_mp_comm athena_mp_comm;
typedef struct athena_mp_shard athena_mp_shard;

/* 128 opaque bytes (an ncclUniqueId) drawn by ONE rank and handed to all (MPI_Bcast, a file, a torch store ...) */
int athena_mp_comm_unique_id(void *id128);
/* collective; uses the device athena_mp_init selected */
int athena_mp_comm_create(int32_t rank, int32_t world, const void *id128, athena_mp_comm **out);
/* MPI-free bootstrap: rank 0 publishes the id in `path`, the others wait for it.  Safe against files a crashed run left
 * behind: ranks > 0 announce themselves with a random nonce in `path`.hello.<rank> (re-written as a heartbeat), rank 0
 * accepts only hello files it has seen change and publishes id + nonces, a rank accepts `path` only when it carries its
 * own nonce.  Every wait is bounded by ATHENA_MP_COLLECTIVE_TIMEOUT_S when set, 300 s otherwise. */
int athena_mp_comm_create_from_file(int32_t rank, int32_t world, const char *path, athena_mp_comm **out);
int athena_mp_comm_destroy(athena_mp_comm *c);
int athena_mp_comm_info(const athena_mp_comm *c, int32_t *rank, int32_t *world, char *transport, int32_t transport_len);
/* What this rank's communicator has actually done, so that a multi-GPU line proves what it used: ranks_seen = the
 * transport's OWN count of ranks (RCCL: ncclCommCount; -1 unavailable), library_version (RCCL: ncclGetVersion, e.g. 22606),
 * transfers started, all-reduce payload bytes, and the bytes put on the wire to every peer (grouped ncclSend rows and this
 * rank's block of every whole-block all-gather); sent_bytes_per_peer [n_peers] may be NULL with n_peers 0. */
int athena_mp_comm_stats(const athena_mp_comm *c, int32_t *ranks_seen, int32_t *library_version, int64_t *transfers,
                         int64_t *allreduce_bytes, int64_t *sent_bytes_per_peer, int32_t n_peers);
/* Every transfer these entry points start has a deadline (ATHENA_MP_COLLECTIVE_TIMEOUT_S seconds from the moment it actually
 * starts on the device; default 1800, 0 = none): a rank whose peer is missing ends with a message on stderr / in
 * athena_mp_last_error and exit code 3 instead of hanging in its next synchronize.  A handler registered here is called first,
 * once, on the library's monitor thread -- the host's chance to flush its own state; the process ends when it returns.  Nothing
 * is ever written on the host program's stdout. */
int athena_mp_set_stall_handler(void (*handler)(int32_t rank, const char *what, double seconds));
int athena_mp_comm_barrier(athena_mp_comm *c);
/* in-place float32 sum over ranks (dW, d theta): _start enqueues it on the communication stream behind the work the
 * compute stream holds so far, _finish makes the compute stream wait for it; the host never blocks */
int athena_mp_allreduce_start(athena_mp_comm *c, float *buf_dev, int64_t count);
int athena_mp_allreduce_finish(athena_mp_comm *c);
int athena_mp_allreduce(athena_mp_comm *c, float *buf_dev, int64_t count);

/* the rank's rows of the global graph: adj_ia [n_local+1] 1-based, adj_ja [2,nnz] column-major with adj_ja(1,w) = GLOBAL
 * neighbour id (1-based; graph_type%adj_ja of the whole graph restricted to the rank's rows), adj_ja(2,.) ignored.
 * Rows are contiguous blocks in rank order.  Builds the [local | halo] renumbering (interior rows first), the halo
 * degrees and the four row blocks as graph handles, and decides HOW the halo travels:
 *   p2p        grouped ncclSend / ncclRecv of the distinct rows each peer needs, packed by a gather kernel;
 *              x_ext = [n local | n_halo distinct remote rows, grouped by owner]
 *   all-gather when the ranks together need more than tau (default 0.7, ATHENA_MP_HALO_ALLGATHER_FRACTION) of all remote
 *              rows anyway: ONE ncclAllGather of whole blocks, no pack kernel, no send lists;
 *              x_ext = [max_n local slots, n used | world blocks of max_n slots, block p in rank p's own row order] --
 *              a halo row is a view into its owner's block
 *   ATHENA_MP_HALO_MODE = auto | p2p | allgather overrides the choice.  Either way a caller allocates n_local + n_halo
 *   rows (athena_mp_shard_dims) and uses the shard's graphs; nothing else differs on its side.
 * Collective: argument errors are agreed on before any data moves, so all ranks return the error together.  The reverse
 * pass of a shard is a pull over the rank's own rows (exact for undirected graphs: row u lists v as often as row v lists
 * u) -- checked across all ranks with one 64-bit signed hash sum; a directed graph is an error on every rank. */
int athena_mp_shard_create(athena_mp_comm *c, int32_t n_local, int64_t nnz, const int32_t *adj_ia, const int32_t *adj_ja,
                           athena_mp_shard **out);
/* The same for ONE graph WITH edge features cut by rows (graph_nop_layer on a partitioned mesh -- SURVEY.md 8e "GNO: as
 * Kipf plus replicated theta and all-reduce of dtheta"; BASELINE configs[3] is one 2 M-vertex mesh, which sharding by whole
 * graphs cannot split).  adj_ja(2,w) = GLOBAL edge id (1-based, 0 = none) as graph_type%adj_ja carries it
 * (athena_diffstruc_extd_sub_nop.f90:367-378 reads it as the kernel column).  The four row-block handles keep the edge
 * columns, renumbered to the rank's own set: the distinct ids its rows reference, ascending -- athena_mp_shard_export(7)
 * lists them, athena_mp_shard_edge_cols counts them, and the rank holds coords [n_edge_cols, d] for exactly those.
 * Forward:  halo exchange of x, athena_mp_gno_aggregate_fwd on blocks 0 / 1 (interior rows under the transfer).
 * Reverse:  halo exchange of g = dL/dm, athena_mp_gno_aggregate_bwd_x_pull on blocks 2 / 3 (the pull over the rank's own
 *           rows -- requires row u to list (v, e) whenever row v lists (u, e), which is CHECKED here across all ranks and
 *           is an error on every rank otherwise); athena_mp_gno_aggregate_bwd_theta on blocks 0 / 1 needs local g rows
 *           only; theta is replicated and d theta (+ dW, db) all-reduced (athena_mp_allreduce). */
int athena_mp_shard_create_edges(athena_mp_comm *c, int32_t n_local, int64_t nnz, const int32_t *adj_ia,
                                 const int32_t *adj_ja, athena_mp_shard **out);
int athena_mp_shard_edge_cols(const athena_mp_shard *s, int32_t *n_edge_cols);
int athena_mp_shard_destroy(athena_mp_shard *s);
int athena_mp_shard_dims(const athena_mp_shard *s, int32_t *n_local, int32_t *n_interior, int32_t *n_halo, int64_t *nnz,
                         int64_t *row_offset, int64_t *n_total);
/* *mode 0 = p2p, 1 = all-gather; *fraction = distinct halo rows summed over the ranks / ((world-1) * n_total); *tau = the
 * threshold in force; *recv_rows = rows that cross a link into this rank per exchange */
int athena_mp_shard_info(const athena_mp_shard *s, int32_t *mode, double *fraction, double *tau, int64_t *recv_rows);
/* which: 0 / 1 = interior rows [0,n_int) / boundary rows [n_int,n) of the forward graph, 2 / 3 = the same blocks of the
 * backward (pull) graph; columns index x_ext = [n local rows | n_halo halo rows]; owned by the shard */
int athena_mp_shard_graph(const athena_mp_shard *s, int32_t which, athena_mp_graph **g);
/* 0 order [n] int32 | 1 halo_ids [distinct remote rows referenced] int64 | 2 send_idx [n_send] int32 |
 * 3 col_deg [n+n_halo] int32 | 4 send_counts [world] int64 | 5 recv_counts [world] int64 |
 * 6 ext_ids [n_halo] int64: global id held by each row of x_ext beyond the local ones, -1 = padding slot (== 1 in p2p mode);
 * 7 edge_ids [n_edge_cols] int64: global edge id (0-based) of each local edge column (shards built by _create_edges);
 * 8 cut edge columns [..] int32 (local ids grouped by peer) | 9 their offsets per peer [world+1] int64;
 * host_dst NULL = size query (count in elements) */
int athena_mp_shard_export(const athena_mp_shard *s, int32_t which, void *host_dst, int64_t capacity, int64_t *count);
/* x_ext [n + n_halo, F]: p2p mode packs + posts the grouped send/recv into the halo rows, all-gather mode posts one
 * ncclAllGather of the blocks; returns at once (kernels enqueued before _finish run under the transfer); slot 0 / 1 = two
 * exchanges may be outstanding */
int athena_mp_halo_start(athena_mp_shard *s, int32_t slot, int32_t F, float *x_ext_dev);
int athena_mp_halo_finish(athena_mp_shard *s, int32_t slot);
/* Halo REDUCE, the transpose of the exchange: y_ext [n + n_halo, F] (F a multiple of 4) holds in its rows beyond the local
 * ones what this rank computed FOR remote vertices -- the feature gradient of a scatter-form reverse pass
 * (athena_mp_gno_aggregate_bwd on a forward row block: get_partial_gno_agg_features_val, athena_diffstruc_extd_sub_nop.f90:419-458,
 * writes dx of every neighbour, local or not).  _start sends each such row to its owner (p2p mode: the contiguous segment of
 * every owner; all-gather mode: the whole block of every owner) and receives what the peers computed for this rank's rows;
 * _finish adds them into y_local [n, F] on the compute stream, peer by peer in rank order (deterministic).  Streams, events,
 * deadline and slots as athena_mp_halo_start / _finish; a slot carries one exchange OR one reduce at a time.  With it the
 * reverse pass of a partitioned graph needs no exchange of the upstream gradient and no symmetry of the graph. */
int athena_mp_halo_reduce_start(athena_mp_shard *s, int32_t slot, int32_t F, const float *y_ext_dev);
int athena_mp_halo_reduce_finish(athena_mp_shard *s, int32_t slot, float *y_local_dev);
/* Sum over the partition of a per-edge-column quantity -- the coordinate gradient of graph_nop_layer
 * (get_partial_gno_kernel_coords_val, athena_diffstruc_extd_sub_nop.f90:137-216): e_dev [n_edge_cols, F] holds this rank's share
 * (the sum over its own rows' entries); an edge column CUT by the partition has a share on both sides.  The cut columns are
 * exchanged with the one peer that shares them and added in: afterwards both ranks hold the full sum (the same bits).
 * Stream-ordered, the host does not block.  Shards of athena_mp_shard_create_edges only; athena_mp_shard_export(8 / 9) lists
 * the cut columns per peer. */
int athena_mp_shard_edge_reduce(athena_mp_shard *s, int32_t F, float *e_dev);
/* DEADLINES.  No RCCL collective has a completion deadline of its own, and the host never blocks in _halo_start /
 * _allreduce_start, so a rank whose peer is missing would hang in its next synchronize, far from the cause.  Every transfer
 * this library starts (halo exchange, gradient all-reduce, the metadata collectives of athena_mp_shard_create,
 * athena_mp_comm_barrier) is therefore watched through its completion event by a monitor thread: still pending
 * ATHENA_MP_COLLECTIVE_TIMEOUT_S seconds (default 1800 -- one rank may legitimately write a checkpoint, run an evaluation pass or
 * sit in a debugger; bench.py sets 120 for itself; x5 for the metadata collectives and the barrier; 0 = no monitor) after it
 * actually STARTED (an event just in front of it on the communication stream has completed: a host that enqueues many steps
 * ahead of the device is not mistaken for a stall) the PROCESS ends -- "[athena_mp] rank r stalled in <transfer> ..." on stderr
 * and in athena_mp_last_error, the handler of athena_mp_set_stall_handler if one is registered, exit code 3; no retry, no
 * cleanup that could block in the same communicator, nothing on the host program's stdout.  (The file bootstrap of
 * _comm_create_from_file waits ATHENA_MP_COLLECTIVE_TIMEOUT_S when set, 300 s otherwise.) */

/* ---- residency of host arrays: the *_host entry points without the PCIe round trip per op ---------------------------- *
 * athena's layers exchange array_type nodes whose %val lives on the host (athena_network_sub.f90:2752,2761: forward_generic2d
 * hands layer%output on; :2856: the optimiser reads the gradients).  With athena_mp_resident_mode(1) the result of a *_host
 * call stays in HBM, registered under its host array's (address, byte length), and a later *_host call that receives the
 * same array as an input uses the device copy: a chain of HIP ops moves its first input up once and nothing else until
 * athena_mp_resident_flush(host_ptr) materialises an array at the edge of the HIP island (NULL: all of them).
 * FORWARD results park; the outputs of the REVERSE entry points (every *_bwd_*_host, *_dw_host, *_dx_host and the *_pair_host
 * calls) are always copied home as well: they are the partials diffstruc's grad_reverse accumulates with host arithmetic the
 * moment a get_partial_*_val callback returns -- inside the island only as far as their INPUTS go.
 * Safety: an array whose only valid copy is on the device carries a 16-byte sentinel at both ends of its host storage; if
 * host code wrote it (or the allocator handed the address to another array) the sentinel is gone and the host content is
 * uploaded instead -- and a flush (explicit, forced by an overlapping argument, or athena_mp_resident_mode(0)) then leaves the
 * host array ALONE: the host content is the newer one (round 3 copied the stale device data over it).  The sentinel guards
 * the two ENDS of the array: host code that rewrites only interior elements of a parked result without touching its first
 * or last 16 bytes is outside the contract (flush the array before writing into it).  An argument that merely overlaps a
 * registered array (a slice) materialises that array first.
 * Arrays under 64 bytes are always staged.  athena_mp_resident_mode(0) flushes everything and releases the device copies.
 * LIFETIME CONTRACT: the table holds host ADDRESSES.  An array must be dropped (athena_mp_resident_drop) before its host
 * storage is deallocated -- in the finaliser of the node / layer that owns it -- because a flush (explicit, at
 * athena_mp_resident_mode(0), or forced by an overlapping argument) WRITES to that address.  athena_mp_resident_drop(NULL)
 * forgets everything without touching host memory (the safe way out when lifetimes are unknown: flush what you read first).
 * Explicit form for a shim that manages an array itself: _acquire (dirty_host = 1: upload now; 0: allocate / trust the
 * device copy) -> device pointer for the *_dev entry points; _release(dirty_dev = 1) marks the device copy as the valid
 * one; _drop forgets an array WITHOUT copying back (call it when the host array is deallocated). */
int athena_mp_resident_mode(int32_t on);
int athena_mp_resident_acquire(const void *host_ptr, uint64_t bytes, int32_t dirty_host, void **dev_ptr);
int athena_mp_resident_release(const void *host_ptr, int32_t dirty_dev);
int athena_mp_resident_flush(const void *host_ptr);
int athena_mp_resident_drop(const void *host_ptr);
int athena_mp_resident_stats(int64_t *arrays, int64_t *h2d_bytes, int64_t *d2h_bytes, int64_t *reused_inputs,
                             int64_t *lazy_outputs);

#ifdef __cplusplus
}
#endif
#endif /* ATHENA_MP_H */
