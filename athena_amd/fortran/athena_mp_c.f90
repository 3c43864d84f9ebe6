!! ISO_C_BINDING view of include/athena_mp.h -- the thin shim the north star asks for.
!! Every interface is `bind(C)`; the ones invoked from diffstruc's `pure` get_partial_*_val
!! callbacks are declared `pure` (legal for bind(C) interfaces; SURVEY.md 8b).
!! Host arrays are passed by reference exactly as Fortran holds them: array_type%val(F,N) is
!! column-major, i.e. the row-major [N][F] the kernels read; adj_ia / adj_ja keep their 1-based
!! values (athena_mp_graph_create converts once).
module athena_mp_c
  use, intrinsic :: iso_c_binding
  implicit none
  private

  integer(c_int), parameter, public :: ATHENA_MP_ACT_NONE = 0, ATHENA_MP_ACT_RELU = 1, &
       ATHENA_MP_ACT_SIGMOID = 2, ATHENA_MP_ACT_TANH = 3

  public :: athena_mp_init, athena_mp_initialized, athena_mp_finalize, athena_mp_last_error, athena_mp_synchronize
  public :: athena_mp_graph_create, athena_mp_graph_destroy, athena_mp_graph_key
  public :: athena_mp_graph_acquire, athena_mp_graph_release, athena_mp_graph_evict, athena_mp_graph_cache_stats
  public :: athena_mp_kipf_propagate_fwd_host, athena_mp_kipf_propagate_bwd_host
  public :: athena_mp_gemm_fwd_host
  public :: athena_mp_duvenaud_propagate_fwd_host, athena_mp_duvenaud_propagate_bwd_x_host
  public :: athena_mp_duvenaud_update_fwd_host
  public :: athena_mp_duvenaud_propagate_bwd_e_host, athena_mp_duvenaud_update_bwd_a_host, athena_mp_duvenaud_update_bwd_w_host
  public :: athena_mp_gno_aggregate_fwd_host, athena_mp_gno_aggregate_bwd_x_host, athena_mp_gno_aggregate_bwd_theta_host
  public :: athena_mp_gno_aggregate_bwd_coords_host
  public :: athena_mp_gemm_dw_host, athena_mp_gemm_dx_host, athena_mp_softmax_fwd_host, athena_mp_softmax_bwd_host
  public :: athena_mp_activation_fwd_host, athena_mp_activation_bwd_host
  public :: athena_mp_duvenaud_update_act_fwd_host, athena_mp_duvenaud_update_readout_fwd_host
  public :: athena_mp_duvenaud_update_bwd_pair_host, athena_mp_gno_aggregate_bwd_pair_host, athena_mp_pair_stats
  public :: athena_mp_malloc, athena_mp_free, athena_mp_memcpy_h2d, athena_mp_memcpy_d2h
  public :: athena_mp_kipf_propagate_fwd, athena_mp_kipf_propagate_bwd
  public :: athena_mp_gemm_fwd, athena_mp_gemm_dw, athena_mp_gemm_dx
  public :: athena_mp_adam_step, athena_mp_sgd_step, athena_mp_clip, athena_mp_mse_loss
  public :: athena_mp_swish_fwd, athena_mp_swish_bwd, athena_mp_softmax_fwd, athena_mp_softmax_bwd
  public :: athena_mp_concat_fwd, athena_mp_concat_bwd
  public :: athena_mp_activation_param_fwd, athena_mp_activation_param_bwd
  public :: athena_mp_memset_zero, athena_mp_activation_fwd, athena_mp_axpy, athena_mp_kipf_propagate_bwd_dual
  public :: athena_mp_kipf_propagate_act_fwd
  public :: athena_mp_reverse_kipf_propagate_fwd, athena_mp_reverse_kipf_propagate_partial
  public :: athena_mp_reverse_kipf_propagate_partial_val
  public :: athena_mp_duvenaud_propagate_fwd, athena_mp_duvenaud_propagate_bwd_x, athena_mp_duvenaud_propagate_bwd_e
  public :: athena_mp_duvenaud_update_bwd_a, athena_mp_duvenaud_update_bwd_w, athena_mp_duvenaud_update_bwd
  public :: athena_mp_duvenaud_update_bwd_split
  public :: athena_mp_duvenaud_readout_update_bwd
  public :: athena_mp_duvenaud_update_readout_fwd_split
  public :: athena_mp_duvenaud_update_readout_fwd
  public :: athena_mp_segment_sum, athena_mp_segment_sum_bwd
  public :: athena_mp_gno_aggregate_fwd, athena_mp_gno_aggregate_bwd_x, athena_mp_gno_aggregate_bwd_theta
  public :: athena_mp_gno_aggregate_bwd_coords
  public :: athena_mp_gno_saved_bytes, athena_mp_gno_aggregate_fwd_save, athena_mp_gno_aggregate_bwd_theta_saved
  public :: athena_mp_duvenaud_readout_fwd, athena_mp_duvenaud_readout_bwd, athena_mp_duvenaud_update_act_fwd
  public :: athena_mp_kipf_layer_fwd, athena_mp_kipf_layer_bwd_x, athena_mp_activation_bwd
  public :: athena_mp_csr_from_edges, athena_mp_graph_export, athena_mp_graph_create_from_edges
  public :: athena_mp_error_message
  public :: athena_mp_pull_gemm, athena_mp_dev_offset, athena_mp_kipf_layer_bwd
  public :: athena_mp_comm_create, athena_mp_comm_create_from_file, athena_mp_comm_destroy, athena_mp_comm_barrier
  public :: athena_mp_comm_unique_id, athena_mp_allreduce, athena_mp_allreduce_start, athena_mp_allreduce_finish
  public :: athena_mp_shard_create, athena_mp_shard_destroy, athena_mp_shard_dims, athena_mp_shard_graph
  public :: athena_mp_shard_export, athena_mp_halo_start, athena_mp_halo_finish, athena_mp_shard_info
  public :: athena_mp_device_copy, athena_mp_gno_aggregate_bwd, athena_mp_halo_reduce_start, athena_mp_halo_reduce_finish
  public :: athena_mp_shard_edge_reduce
  public :: athena_mp_shard_create_edges, athena_mp_shard_edge_cols, athena_mp_gno_aggregate_bwd_x_pull
  public :: athena_mp_resident_mode, athena_mp_resident_acquire, athena_mp_resident_release, athena_mp_resident_flush
  public :: athena_mp_resident_drop, athena_mp_resident_stats

  interface
     integer(c_int) function athena_mp_init(device) bind(C, name="athena_mp_init")
       import :: c_int
       integer(c_int), value :: device
     end function
     integer(c_int) function athena_mp_initialized() bind(C, name="athena_mp_initialized")
       import :: c_int
     end function
     integer(c_int) function athena_mp_finalize() bind(C, name="athena_mp_finalize")
       import :: c_int
     end function
     integer(c_int) function athena_mp_synchronize() bind(C, name="athena_mp_synchronize")
       import :: c_int
     end function
     type(c_ptr) function athena_mp_last_error() bind(C, name="athena_mp_last_error")
       import :: c_ptr
     end function

     !! residency of host arrays (phase 2 for the *_host entry points): results stay in HBM until flushed at the edge of
     !! the HIP island (athena_network_sub.f90:2752,2761,2856); host_ptr = c_loc(array_type%val)
     integer(c_int) function athena_mp_resident_mode(on) bind(C, name="athena_mp_resident_mode")
       import :: c_int, c_int32_t
       integer(c_int32_t), value :: on
     end function
     integer(c_int) function athena_mp_resident_acquire(host_ptr, bytes, dirty_host, dev_ptr) &
          bind(C, name="athena_mp_resident_acquire")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       type(c_ptr), value :: host_ptr
       integer(c_int64_t), value :: bytes
       integer(c_int32_t), value :: dirty_host
       type(c_ptr), intent(out) :: dev_ptr
     end function
     integer(c_int) function athena_mp_resident_release(host_ptr, dirty_dev) bind(C, name="athena_mp_resident_release")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: host_ptr
       integer(c_int32_t), value :: dirty_dev
     end function
     integer(c_int) function athena_mp_resident_flush(host_ptr) bind(C, name="athena_mp_resident_flush")
       import :: c_int, c_ptr
       type(c_ptr), value :: host_ptr              !! c_null_ptr: every array
     end function
     integer(c_int) function athena_mp_resident_drop(host_ptr) bind(C, name="athena_mp_resident_drop")
       import :: c_int, c_ptr
       type(c_ptr), value :: host_ptr
     end function
     integer(c_int) function athena_mp_resident_stats(arrays, h2d_bytes, d2h_bytes, reused_inputs, lazy_outputs) &
          bind(C, name="athena_mp_resident_stats")
       import :: c_int, c_int64_t
       integer(c_int64_t), intent(out) :: arrays, h2d_bytes, d2h_bytes, reused_inputs, lazy_outputs
     end function

     !! athena_msgpass_layer_sub.f90:144-174 (set_graph) -- build once, keep the handle
     integer(c_int) function athena_mp_graph_create(n_rows, n_cols, nnz, adj_ia, adj_ja, &
          n_edge_cols, row_deg, col_deg, graph) bind(C, name="athena_mp_graph_create")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       integer(c_int32_t), value :: n_rows, n_cols, n_edge_cols
       integer(c_int64_t), value :: nnz
       integer(c_int32_t), intent(in) :: adj_ia(*), adj_ja(2,*)
       type(c_ptr), value :: row_deg, col_deg        !! c_null_ptr: degrees = CSR row lengths
       type(c_ptr), intent(out) :: graph
     end function
     integer(c_int) function athena_mp_graph_destroy(graph) bind(C, name="athena_mp_graph_destroy")
       import :: c_int, c_ptr
       type(c_ptr), value :: graph
     end function
     !! content key of (adj_ia, adj_ja): what a cached handle is valid for (SURVEY.md 8b "Ownership");
     !! key is C's uint64_t -- compared for equality only, so its bit pattern in an int64 serves
     pure integer(c_int) function athena_mp_graph_key(n_rows, nnz, adj_ia, adj_ja, key) &
          bind(C, name="athena_mp_graph_key")
       import :: c_int, c_int32_t, c_int64_t
       integer(c_int32_t), value :: n_rows
       integer(c_int64_t), value :: nnz
       integer(c_int32_t), intent(in) :: adj_ia(*), adj_ja(2,*)
       integer(c_int64_t), intent(out) :: key
     end function
     !! set_graph as a cache lookup (athena_network_sub.f90:2727-2730 calls set_graph before every forward):
     !! the shared, reference-counted handle of this CSR; built only when no live or idle handle matches
     integer(c_int) function athena_mp_graph_acquire(n, nnz, adj_ia, adj_ja, n_edge_cols, graph) &
          bind(C, name="athena_mp_graph_acquire")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       integer(c_int32_t), value :: n, n_edge_cols
       integer(c_int64_t), value :: nnz
       integer(c_int32_t), intent(in) :: adj_ia(*), adj_ja(2,*)
       type(c_ptr), intent(out) :: graph
     end function
     integer(c_int) function athena_mp_graph_release(graph) bind(C, name="athena_mp_graph_release")
       import :: c_int, c_ptr
       type(c_ptr), value :: graph
     end function
     !! release + the cache forgets the handle: after an in-place edit the caller announces (invalidate_graph)
     integer(c_int) function athena_mp_graph_evict(graph) bind(C, name="athena_mp_graph_evict")
       import :: c_int, c_ptr
       type(c_ptr), value :: graph
     end function
     integer(c_int) function athena_mp_graph_cache_stats(handles, hits, builds) &
          bind(C, name="athena_mp_graph_cache_stats")
       import :: c_int, c_int64_t
       integer(c_int64_t), intent(out) :: handles, hits, builds
     end function

     !! kipf_propagate, athena_diffstruc_extd_sub_kipf.f90:7-59 (host arrays, staged)
     pure integer(c_int) function athena_mp_kipf_propagate_fwd_host(graph, F, x, y) &
          bind(C, name="athena_mp_kipf_propagate_fwd_host")
       import :: c_int, c_int32_t, c_ptr, c_float
       type(c_ptr), value :: graph
       integer(c_int32_t), value :: F
       real(c_float), intent(in) :: x(*)
       real(c_float), intent(inout) :: y(*)
     end function
     !! get_partial_kipf_propagate_left_val, ..._sub_kipf.f90:85-111
     pure integer(c_int) function athena_mp_kipf_propagate_bwd_host(graph, F, grad, dx, exact) &
          bind(C, name="athena_mp_kipf_propagate_bwd_host")
       import :: c_int, c_int32_t, c_ptr, c_float
       type(c_ptr), value :: graph
       integer(c_int32_t), value :: F, exact
       real(c_float), intent(in) :: grad(*)
       real(c_float), intent(inout) :: dx(*)
     end function
     !! matmul(params(t), ptr2), athena_kipf_msgpass_layer.f90:951
     pure integer(c_int) function athena_mp_gemm_fwd_host(N, Fi, Fo, P, W, bias, act, Z) &
          bind(C, name="athena_mp_gemm_fwd_host")
       import :: c_int, c_int32_t, c_int64_t, c_ptr, c_float
       integer(c_int64_t), value :: N
       integer(c_int32_t), value :: Fi, Fo, act
       real(c_float), intent(in) :: P(*), W(*)
       type(c_ptr), value :: bias
       real(c_float), intent(inout) :: Z(*)
     end function

     !! duvenaud_propagate, athena_diffstruc_extd_sub_duvenaud.f90:7-59
     pure integer(c_int) function athena_mp_duvenaud_propagate_fwd_host(graph, Fv, Fe, x, e, c) &
          bind(C, name="athena_mp_duvenaud_propagate_fwd_host")
       import :: c_int, c_int32_t, c_ptr, c_float
       type(c_ptr), value :: graph
       integer(c_int32_t), value :: Fv, Fe
       real(c_float), intent(in) :: x(*), e(*)
       real(c_float), intent(inout) :: c(*)
     end function
     !! get_partial_duvenaud_propagate_left_val, :115-141
     pure integer(c_int) function athena_mp_duvenaud_propagate_bwd_x_host(graph, Fv, Fe, grad, dx) &
          bind(C, name="athena_mp_duvenaud_propagate_bwd_x_host")
       import :: c_int, c_int32_t, c_ptr, c_float
       type(c_ptr), value :: graph
       integer(c_int32_t), value :: Fv, Fe
       real(c_float), intent(in) :: grad(*)
       real(c_float), intent(inout) :: dx(*)
     end function
     !! duvenaud_update, :176-228
     pure integer(c_int) function athena_mp_duvenaud_update_fwd_host(graph, Fi, Fo, min_deg, max_deg, a, w, c) &
          bind(C, name="athena_mp_duvenaud_update_fwd_host")
       import :: c_int, c_int32_t, c_ptr, c_float
       type(c_ptr), value :: graph
       integer(c_int32_t), value :: Fi, Fo, min_deg, max_deg
       real(c_float), intent(in) :: a(*), w(*)
       real(c_float), intent(inout) :: c(*)
     end function

     !! get_partial_duvenaud_propagate_right_val, :143-171
     pure integer(c_int) function athena_mp_duvenaud_propagate_bwd_e_host(graph, Fv, Fe, grad, de) &
          bind(C, name="athena_mp_duvenaud_propagate_bwd_e_host")
       import :: c_int, c_int32_t, c_ptr, c_float
       type(c_ptr), value :: graph
       integer(c_int32_t), value :: Fv, Fe
       real(c_float), intent(in) :: grad(*)
       real(c_float), intent(inout) :: de(*)
     end function
     !! get_partial_duvenaud_update_val, :284-324
     pure integer(c_int) function athena_mp_duvenaud_update_bwd_a_host(graph, Fi, Fo, min_deg, max_deg, grad, w, da) &
          bind(C, name="athena_mp_duvenaud_update_bwd_a_host")
       import :: c_int, c_int32_t, c_ptr, c_float
       type(c_ptr), value :: graph
       integer(c_int32_t), value :: Fi, Fo, min_deg, max_deg
       real(c_float), intent(in) :: grad(*), w(*)
       real(c_float), intent(inout) :: da(*)
     end function
     !! get_partial_duvenaud_update_weight_val, :326-368
     pure integer(c_int) function athena_mp_duvenaud_update_bwd_w_host(graph, Fi, Fo, min_deg, max_deg, grad, a, dw) &
          bind(C, name="athena_mp_duvenaud_update_bwd_w_host")
       import :: c_int, c_int32_t, c_ptr, c_float
       type(c_ptr), value :: graph
       integer(c_int32_t), value :: Fi, Fo, min_deg, max_deg
       real(c_float), intent(in) :: grad(*), a(*)
       real(c_float), intent(inout) :: dw(*)
     end function
     !! gno_kernel_eval + gno_aggregate in one op (athena_diffstruc_extd_sub_nop.f90:26-115, :330-397): the per-edge kernel
     !! tensor is never formed; theta = U | b_u | V | b_v as gno_kernel_eval packs it (:74-82)
     pure integer(c_int) function athena_mp_gno_aggregate_fwd_host(graph, d, H, Fi, Fo, theta, coords, x, m) &
          bind(C, name="athena_mp_gno_aggregate_fwd_host")
       import :: c_int, c_int32_t, c_ptr, c_float
       type(c_ptr), value :: graph
       integer(c_int32_t), value :: d, H, Fi, Fo
       real(c_float), intent(in) :: theta(*), coords(*), x(*)
       real(c_float), intent(inout) :: m(*)
     end function
     !! get_partial_gno_aggregate_features_val, :419-458 (through the kernels)
     pure integer(c_int) function athena_mp_gno_aggregate_bwd_x_host(graph, d, H, Fi, Fo, theta, coords, grad, dx) &
          bind(C, name="athena_mp_gno_aggregate_bwd_x_host")
       import :: c_int, c_int32_t, c_ptr, c_float
       type(c_ptr), value :: graph
       integer(c_int32_t), value :: d, H, Fi, Fo
       real(c_float), intent(in) :: theta(*), coords(*), grad(*)
       real(c_float), intent(inout) :: dx(*)
     end function
     !! agg -> kernels (:480-526) chained with kernel -> params (:235-325)
     pure integer(c_int) function athena_mp_gno_aggregate_bwd_theta_host(graph, d, H, Fi, Fo, theta, coords, x, grad, dtheta) &
          bind(C, name="athena_mp_gno_aggregate_bwd_theta_host")
       import :: c_int, c_int32_t, c_ptr, c_float
       type(c_ptr), value :: graph
       integer(c_int32_t), value :: d, H, Fi, Fo
       real(c_float), intent(in) :: theta(*), coords(*), x(*), grad(*)
       real(c_float), intent(inout) :: dtheta(*)
     end function
     !! agg -> kernels chained with kernel -> coords (:137-216)
     pure integer(c_int) function athena_mp_gno_aggregate_bwd_coords_host(graph, d, H, Fi, Fo, theta, coords, x, grad, &
          dcoords) bind(C, name="athena_mp_gno_aggregate_bwd_coords_host")
       import :: c_int, c_int32_t, c_ptr, c_float
       type(c_ptr), value :: graph
       integer(c_int32_t), value :: d, H, Fi, Fo
       real(c_float), intent(in) :: theta(*), coords(*), x(*), grad(*)
       real(c_float), intent(inout) :: dcoords(*)
     end function

     !! ---- what a shim inside athena binds beyond the op-granular entry points (scripts/integration_check/) -----------
     !! the reverse of matmul, for composite nodes that own a dense step: dW = P^T dZ, dP = dZ W^T
     pure integer(c_int) function athena_mp_gemm_dw_host(N, Fi, Fo, P, dZ, dW) bind(C, name="athena_mp_gemm_dw_host")
       import :: c_int, c_int32_t, c_int64_t, c_float
       integer(c_int64_t), value :: N
       integer(c_int32_t), value :: Fi, Fo
       real(c_float), intent(in) :: P(*), dZ(*)
       real(c_float), intent(inout) :: dW(*)
     end function
     pure integer(c_int) function athena_mp_gemm_dx_host(N, Fi, Fo, dZ, W, dP) bind(C, name="athena_mp_gemm_dx_host")
       import :: c_int, c_int32_t, c_int64_t, c_float
       integer(c_int64_t), value :: N
       integer(c_int32_t), value :: Fi, Fo
       real(c_float), intent(in) :: dZ(*), W(*)
       real(c_float), intent(inout) :: dP(*)
     end function
     !! softmax over the features of every column (athena_diffstruc_extd_sub.f90:295-379) and its reverse at the OUTPUT
     pure integer(c_int) function athena_mp_softmax_fwd_host(N, F, z, y) bind(C, name="athena_mp_softmax_fwd_host")
       import :: c_int, c_int32_t, c_int64_t, c_float
       integer(c_int64_t), value :: N
       integer(c_int32_t), value :: F
       real(c_float), intent(in) :: z(*)
       real(c_float), intent(inout) :: y(*)
     end function
     pure integer(c_int) function athena_mp_softmax_bwd_host(N, F, y, g, dz) bind(C, name="athena_mp_softmax_bwd_host")
       import :: c_int, c_int32_t, c_int64_t, c_float
       integer(c_int64_t), value :: N
       integer(c_int32_t), value :: F
       real(c_float), intent(in) :: y(*), g(*)
       real(c_float), intent(inout) :: dz(*)
     end function
     pure integer(c_int) function athena_mp_activation_fwd_host(act, n, z, y) bind(C, name="athena_mp_activation_fwd_host")
       import :: c_int, c_int32_t, c_int64_t, c_float
       integer(c_int32_t), value :: act
       integer(c_int64_t), value :: n
       real(c_float), intent(in) :: z(*)
       real(c_float), intent(inout) :: y(*)
     end function
     pure integer(c_int) function athena_mp_activation_bwd_host(act, n, y, g, dz) bind(C, name="athena_mp_activation_bwd_host")
       import :: c_int, c_int32_t, c_int64_t, c_float
       integer(c_int32_t), value :: act
       integer(c_int64_t), value :: n
       real(c_float), intent(in) :: y(*), g(*)
       real(c_float), intent(inout) :: dz(*)
     end function
     !! update + message activation (+ the readout's per-vertex softmax(R z)) of one Duvenaud time step in ONE launch
     !! (athena_duvenaud_msgpass_layer.f90:790-803, 838-855)
     pure integer(c_int) function athena_mp_duvenaud_update_act_fwd_host(graph, Fi, Fo, min_deg, max_deg, a, w, act, z) &
          bind(C, name="athena_mp_duvenaud_update_act_fwd_host")
       import :: c_int, c_int32_t, c_ptr, c_float
       type(c_ptr), value :: graph
       integer(c_int32_t), value :: Fi, Fo, min_deg, max_deg, act
       real(c_float), intent(in) :: a(*), w(*)
       real(c_float), intent(inout) :: z(*)
     end function
     pure integer(c_int) function athena_mp_duvenaud_update_readout_fwd_host(graph, Fi, Fo, min_deg, max_deg, a, w, act, z, &
          O, R, p) bind(C, name="athena_mp_duvenaud_update_readout_fwd_host")
       import :: c_int, c_int32_t, c_ptr, c_float
       type(c_ptr), value :: graph
       integer(c_int32_t), value :: Fi, Fo, min_deg, max_deg, act, O
       real(c_float), intent(in) :: a(*), w(*), R(*)
       real(c_float), intent(inout) :: z(*), p(*)
     end function
     !! BOTH partials of a two-operand node from one device pass, across diffstruc's two stateless `pure` callbacks: the first
     !! request computes both and parks the other on the device, the second -- same handle, shapes and operand CONTENT -- takes
     !! it (include/athena_mp.h).  which: 0 = left operand's partial, 1 = right operand's (GNO: 2 = coordinates).
     !! act /= ATHENA_MP_ACT_NONE: the node is update + message activation in one; z = its value, grad the gradient w.r.t. z
     pure integer(c_int) function athena_mp_duvenaud_update_bwd_pair_host(graph, Fi, Fo, min_deg, max_deg, act, z, grad, a, w, &
          which, out) bind(C, name="athena_mp_duvenaud_update_bwd_pair_host")
       import :: c_int, c_int32_t, c_ptr, c_float
       type(c_ptr), value :: graph
       integer(c_int32_t), value :: Fi, Fo, min_deg, max_deg, act, which
       real(c_float), intent(in) :: z(*), grad(*), a(*), w(*)
       real(c_float), intent(inout) :: out(*)
     end function
     pure integer(c_int) function athena_mp_gno_aggregate_bwd_pair_host(graph, d, H, Fi, Fo, theta, coords, x, grad, which, &
          out) bind(C, name="athena_mp_gno_aggregate_bwd_pair_host")
       import :: c_int, c_int32_t, c_ptr, c_float
       type(c_ptr), value :: graph
       integer(c_int32_t), value :: d, H, Fi, Fo, which
       real(c_float), intent(in) :: theta(*), coords(*), x(*), grad(*)
       real(c_float), intent(inout) :: out(*)
     end function
     integer(c_int) function athena_mp_pair_stats(fused_passes, handed_over) bind(C, name="athena_mp_pair_stats")
       import :: c_int, c_int64_t
       integer(c_int64_t), intent(out) :: fused_passes, handed_over
     end function

     !! device-resident variants (phase 2: tensors stay in HBM between consecutive HIP layers)
     integer(c_int) function athena_mp_malloc(dev_ptr, bytes) bind(C, name="athena_mp_malloc")
       import :: c_int, c_ptr, c_int64_t
       type(c_ptr), intent(out) :: dev_ptr
       integer(c_int64_t), value :: bytes
     end function
     integer(c_int) function athena_mp_free(dev_ptr) bind(C, name="athena_mp_free")
       import :: c_int, c_ptr
       type(c_ptr), value :: dev_ptr
     end function
     integer(c_int) function athena_mp_memcpy_h2d(dst, src, bytes) bind(C, name="athena_mp_memcpy_h2d")
       import :: c_int, c_ptr, c_int64_t
       type(c_ptr), value :: dst
       type(*), dimension(*), intent(in) :: src        ! any host array (real32 tensors, int32 index arrays)
       integer(c_int64_t), value :: bytes
     end function
     integer(c_int) function athena_mp_memcpy_d2h(dst, src, bytes) bind(C, name="athena_mp_memcpy_d2h")
       import :: c_int, c_ptr, c_int64_t
       type(*), dimension(*), intent(inout) :: dst
       type(c_ptr), value :: src
       integer(c_int64_t), value :: bytes
     end function
     integer(c_int) function athena_mp_kipf_propagate_fwd(graph, F, x_dev, y_dev) &
          bind(C, name="athena_mp_kipf_propagate_fwd")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, x_dev, y_dev
       integer(c_int32_t), value :: F
     end function
     integer(c_int) function athena_mp_kipf_propagate_bwd(graph, F, g_dev, dx_dev, exact) &
          bind(C, name="athena_mp_kipf_propagate_bwd")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, g_dev, dx_dev
       integer(c_int32_t), value :: F, exact
     end function
     !! reverse_kipf_propagate and its partials, athena_diffstruc_extd_sub_kipf.f90:116-205
     integer(c_int) function athena_mp_reverse_kipf_propagate_fwd(graph, F, a_dev, c_dev) &
          bind(C, name="athena_mp_reverse_kipf_propagate_fwd")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, a_dev, c_dev
       integer(c_int32_t), value :: F
     end function
     integer(c_int) function athena_mp_reverse_kipf_propagate_partial(graph, F, up_dev, out_dev) &
          bind(C, name="athena_mp_reverse_kipf_propagate_partial")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, up_dev, out_dev
       integer(c_int32_t), value :: F
     end function
     integer(c_int) function athena_mp_reverse_kipf_propagate_partial_val(graph, F, up_dev, out_dev) &
          bind(C, name="athena_mp_reverse_kipf_propagate_partial_val")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, up_dev, out_dev
       integer(c_int32_t), value :: F
     end function
     integer(c_int) function athena_mp_gemm_fwd(N, Fi, Fo, P_dev, W_dev, bias_dev, act, Z_dev) &
          bind(C, name="athena_mp_gemm_fwd")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       integer(c_int64_t), value :: N
       integer(c_int32_t), value :: Fi, Fo, act
       type(c_ptr), value :: P_dev, W_dev, bias_dev, Z_dev
     end function
     integer(c_int) function athena_mp_gemm_dw(N, Fi, Fo, P_dev, dZ_dev, dW_dev) &
          bind(C, name="athena_mp_gemm_dw")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       integer(c_int64_t), value :: N
       integer(c_int32_t), value :: Fi, Fo
       type(c_ptr), value :: P_dev, dZ_dev, dW_dev
     end function
     integer(c_int) function athena_mp_gemm_dx(N, Fi, Fo, dZ_dev, W_dev, dP_dev) &
          bind(C, name="athena_mp_gemm_dx")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       integer(c_int64_t), value :: N
       integer(c_int32_t), value :: Fi, Fo
       type(c_ptr), value :: dZ_dev, W_dev, dP_dev
     end function

     !! edge list -> CSR on the device (graphstruc's generate_adjacency [+ add_self_loops]); adj_ja = c_null_ptr queries nnz
     integer(c_int) function athena_mp_csr_from_edges(n_vertices, n_pairs, index_list, add_self_loops, adj_ia, &
          adj_ja, capacity, nnz) bind(C, name="athena_mp_csr_from_edges")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       integer(c_int32_t), value :: n_vertices, add_self_loops
       integer(c_int64_t), value :: n_pairs, capacity
       integer(c_int32_t), intent(in) :: index_list(2,*)
       integer(c_int32_t), intent(inout) :: adj_ia(*)
       type(c_ptr), value :: adj_ja                     !! c_loc of an integer(c_int32_t) adj_ja(2,capacity) target
       integer(c_int64_t), intent(out) :: nnz
     end function
     !! edge list -> CSR -> handle with the entries staying in HBM in between; adj_ja_out = c_null_ptr: not wanted
     integer(c_int) function athena_mp_graph_create_from_edges(n_vertices, n_pairs, index_list, add_self_loops, &
          with_edge_ids, adj_ia, adj_ja, capacity, nnz, graph) bind(C, name="athena_mp_graph_create_from_edges")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       integer(c_int32_t), value :: n_vertices, add_self_loops, with_edge_ids
       integer(c_int64_t), value :: n_pairs, capacity
       integer(c_int32_t), intent(in) :: index_list(2,*)
       integer(c_int32_t), intent(inout) :: adj_ia(*)
       type(c_ptr), value :: adj_ja
       integer(c_int64_t), intent(out) :: nnz
       type(c_ptr), intent(out) :: graph
     end function
     !! one array of the handle back on the host (which: see include/athena_mp.h); host_dst = c_null_ptr queries count
     integer(c_int) function athena_mp_graph_export(graph, which, host_dst, capacity, count) &
          bind(C, name="athena_mp_graph_export")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       type(c_ptr), value :: graph, host_dst
       integer(c_int32_t), value :: which
       integer(c_int64_t), value :: capacity
       integer(c_int64_t), intent(out) :: count
     end function

     !! ---- one-launch Kipf layer step on device-resident arrays (update_message_kipf :943-952 and its reverse) ----
     integer(c_int) function athena_mp_kipf_layer_fwd(graph, Fi, Fo, x_dev, W_dev, bias_dev, act, P_dev, Z_dev) &
          bind(C, name="athena_mp_kipf_layer_fwd")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, x_dev, W_dev, bias_dev, P_dev, Z_dev
       integer(c_int32_t), value :: Fi, Fo, act
     end function
     integer(c_int) function athena_mp_kipf_layer_bwd_x(graph, Fi, Fo, dZ_dev, W_dev, exact, dX_dev) &
          bind(C, name="athena_mp_kipf_layer_bwd_x")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, dZ_dev, W_dev, dX_dev
       integer(c_int32_t), value :: Fi, Fo, exact
     end function
     integer(c_int) function athena_mp_activation_bwd(act, n, y_dev, g_dev, dz_dev) &
          bind(C, name="athena_mp_activation_bwd")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       integer(c_int32_t), value :: act
       integer(c_int64_t), value :: n
       type(c_ptr), value :: y_dev, g_dev, dz_dev
     end function

     !! ---- device-resident tail of a train step (network%update, athena_network_sub.f90:2816-2929) ----
     !! minimise_adam, athena_optimiser.f90:1027-1091 (iter = optimiser%iter after the increment)
     integer(c_int) function athena_mp_adam_step(n, lr, beta1, beta2, epsilon, iter, reg_kind, l1, l2, &
          decoupled, param_dev, grad_dev, m_dev, v_dev) bind(C, name="athena_mp_adam_step")
       import :: c_int, c_int32_t, c_int64_t, c_float, c_ptr
       integer(c_int64_t), value :: n
       real(c_float), value :: lr, beta1, beta2, epsilon, l1, l2
       integer(c_int32_t), value :: iter, reg_kind, decoupled
       type(c_ptr), value :: param_dev, grad_dev, m_dev, v_dev
     end function
     !! minimise_sgd, athena_optimiser.f90:634-673
     integer(c_int) function athena_mp_sgd_step(n, lr, momentum, nesterov, reg_kind, l1, l2, &
          param_dev, grad_dev, velocity_dev) bind(C, name="athena_mp_sgd_step")
       import :: c_int, c_int32_t, c_int64_t, c_float, c_ptr
       integer(c_int64_t), value :: n
       real(c_float), value :: lr, momentum, l1, l2
       integer(c_int32_t), value :: nesterov, reg_kind
       type(c_ptr), value :: param_dev, grad_dev, velocity_dev
     end function
     !! apply_clip, athena_clipper.f90:165-210
     integer(c_int) function athena_mp_clip(n, grad_dev, l_min_max, clip_min, clip_max, l_norm, clip_norm) &
          bind(C, name="athena_mp_clip")
       import :: c_int, c_int32_t, c_int64_t, c_float, c_ptr
       integer(c_int64_t), value :: n
       type(c_ptr), value :: grad_dev
       integer(c_int32_t), value :: l_min_max, l_norm
       real(c_float), value :: clip_min, clip_max, clip_norm
     end function
     !! compute_mse, athena_loss.f90:393-430
     integer(c_int) function athena_mp_mse_loss(n, pred_dev, expected_dev, loss_dev, dpred_dev) &
          bind(C, name="athena_mp_mse_loss")
       import :: c_int, c_int64_t, c_ptr
       integer(c_int64_t), value :: n
       type(c_ptr), value :: pred_dev, expected_dev, loss_dev, dpred_dev
     end function

     !! ---- activations with attributes (kind: 0 linear, 1 relu(threshold), 2 sigmoid, 3 tanh,
     !!      4 leaky_relu(alpha), 5 selu(alpha, lambda), 6 gaussian(sigma, mu), 7 piecewise(gradient, limit)) ----
     integer(c_int) function athena_mp_activation_param_fwd(kind, n, scale, p0, p1, x_dev, y_dev) &
          bind(C, name="athena_mp_activation_param_fwd")
       import :: c_int, c_int32_t, c_int64_t, c_float, c_ptr
       integer(c_int32_t), value :: kind
       integer(c_int64_t), value :: n
       real(c_float), value :: scale, p0, p1
       type(c_ptr), value :: x_dev, y_dev
     end function
     integer(c_int) function athena_mp_activation_param_bwd(kind, n, scale, p0, p1, x_dev, grad_dev, dx_dev) &
          bind(C, name="athena_mp_activation_param_bwd")
       import :: c_int, c_int32_t, c_int64_t, c_float, c_ptr
       integer(c_int32_t), value :: kind
       integer(c_int64_t), value :: n
       real(c_float), value :: scale, p0, p1
       type(c_ptr), value :: x_dev, grad_dev, dx_dev
     end function

     !! ---- shaped activations and the concatenate merge ----
     integer(c_int) function athena_mp_swish_fwd(n, beta, x_dev, y_dev) bind(C, name="athena_mp_swish_fwd")
       import :: c_int, c_int64_t, c_float, c_ptr
       integer(c_int64_t), value :: n
       real(c_float), value :: beta
       type(c_ptr), value :: x_dev, y_dev
     end function
     integer(c_int) function athena_mp_swish_bwd(n, beta, x_dev, grad_dev, dx_dev) &
          bind(C, name="athena_mp_swish_bwd")
       import :: c_int, c_int64_t, c_float, c_ptr
       integer(c_int64_t), value :: n
       real(c_float), value :: beta
       type(c_ptr), value :: x_dev, grad_dev, dx_dev
     end function
     integer(c_int) function athena_mp_softmax_fwd(N, F, z_dev, y_dev) bind(C, name="athena_mp_softmax_fwd")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       integer(c_int64_t), value :: N
       integer(c_int32_t), value :: F
       type(c_ptr), value :: z_dev, y_dev
     end function
     integer(c_int) function athena_mp_softmax_bwd(N, F, y_dev, grad_dev, dz_dev) &
          bind(C, name="athena_mp_softmax_bwd")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       integer(c_int64_t), value :: N
       integer(c_int32_t), value :: F
       type(c_ptr), value :: y_dev, grad_dev, dz_dev
     end function
     integer(c_int) function athena_mp_concat_fwd(N, Fa, Fb, a_dev, b_dev, out_dev) &
          bind(C, name="athena_mp_concat_fwd")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       integer(c_int64_t), value :: N
       integer(c_int32_t), value :: Fa, Fb
       type(c_ptr), value :: a_dev, b_dev, out_dev
     end function
     integer(c_int) function athena_mp_concat_bwd(N, Fa, Fb, grad_dev, da_dev, db_dev) &
          bind(C, name="athena_mp_concat_bwd")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       integer(c_int64_t), value :: N
       integer(c_int32_t), value :: Fa, Fb
       type(c_ptr), value :: grad_dev, da_dev, db_dev
     end function

     !! ---- Duvenaud composite entry points (one launch each) ----
     integer(c_int) function athena_mp_duvenaud_update_act_fwd(graph, Fi, Fo, min_deg, max_deg, a_dev, &
          weight_dev, act, z_dev) bind(C, name="athena_mp_duvenaud_update_act_fwd")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, a_dev, weight_dev, z_dev
       integer(c_int32_t), value :: Fi, Fo, min_deg, max_deg, act
     end function
     integer(c_int) function athena_mp_duvenaud_readout_fwd(N, Fv, O, S, seg_dev, z_dev, R_dev, p_dev, &
          out_dev, accumulate) bind(C, name="athena_mp_duvenaud_readout_fwd")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       integer(c_int64_t), value :: N
       integer(c_int32_t), value :: Fv, O, S, accumulate
       type(c_ptr), value :: seg_dev, z_dev, R_dev, p_dev, out_dev
     end function
     integer(c_int) function athena_mp_duvenaud_readout_bwd(N, Fv, O, S, seg_dev, z_dev, R_dev, p_dev, &
          gout_dev, dz_next_dev, act, dc_dev, dR_dev, accumulate) bind(C, name="athena_mp_duvenaud_readout_bwd")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       integer(c_int64_t), value :: N
       integer(c_int32_t), value :: Fv, O, S, act, accumulate
       type(c_ptr), value :: seg_dev, z_dev, R_dev, p_dev, gout_dev, dz_next_dev, dc_dev, dR_dev
     end function

     !! ---- the remaining device-pointer entry points the layer types of athena_mp_layers use ----
     integer(c_int) function athena_mp_memset_zero(dev_ptr, bytes) bind(C, name="athena_mp_memset_zero")
       import :: c_int, c_ptr, c_int64_t
       type(c_ptr), value :: dev_ptr
       integer(c_int64_t), value :: bytes
     end function
     integer(c_int) function athena_mp_activation_fwd(act, n, z_dev, y_dev) bind(C, name="athena_mp_activation_fwd")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       integer(c_int32_t), value :: act
       integer(c_int64_t), value :: n
       type(c_ptr), value :: z_dev, y_dev
     end function
     integer(c_int) function athena_mp_device_copy(dst_dev, src_dev, bytes) bind(C, name="athena_mp_device_copy")
       import :: c_int, c_ptr, c_int64_t
       type(c_ptr), value :: dst_dev, src_dev
       integer(c_int64_t), value :: bytes
     end function
     integer(c_int) function athena_mp_axpy(n, alpha, x_dev, y_dev) bind(C, name="athena_mp_axpy")
       import :: c_int, c_int64_t, c_float, c_ptr
       integer(c_int64_t), value :: n
       real(c_float), value :: alpha
       type(c_ptr), value :: x_dev, y_dev
     end function
     integer(c_int) function athena_mp_kipf_propagate_act_fwd(graph, F, x_dev, act, y_dev) &
          bind(C, name="athena_mp_kipf_propagate_act_fwd")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, x_dev, y_dev
       integer(c_int32_t), value :: F, act
     end function
     integer(c_int) function athena_mp_kipf_propagate_bwd_dual(graph, F, g_dev, dx_plain_dev, dx_coef_dev) &
          bind(C, name="athena_mp_kipf_propagate_bwd_dual")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, g_dev, dx_plain_dev, dx_coef_dev
       integer(c_int32_t), value :: F
     end function
     integer(c_int) function athena_mp_duvenaud_propagate_fwd(graph, Fv, Fe, x_dev, e_dev, c_dev) &
          bind(C, name="athena_mp_duvenaud_propagate_fwd")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, x_dev, e_dev, c_dev
       integer(c_int32_t), value :: Fv, Fe
     end function
     integer(c_int) function athena_mp_duvenaud_propagate_bwd_x(graph, Fv, Fe, grad_dev, dx_dev) &
          bind(C, name="athena_mp_duvenaud_propagate_bwd_x")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, grad_dev, dx_dev
       integer(c_int32_t), value :: Fv, Fe
     end function
     integer(c_int) function athena_mp_duvenaud_propagate_bwd_e(graph, Fv, Fe, grad_dev, de_dev) &
          bind(C, name="athena_mp_duvenaud_propagate_bwd_e")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, grad_dev, de_dev
       integer(c_int32_t), value :: Fv, Fe
     end function
     integer(c_int) function athena_mp_duvenaud_update_bwd_a(graph, Fi, Fo, min_deg, max_deg, grad_dev, &
          weight_dev, da_dev) bind(C, name="athena_mp_duvenaud_update_bwd_a")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, grad_dev, weight_dev, da_dev
       integer(c_int32_t), value :: Fi, Fo, min_deg, max_deg
     end function
     integer(c_int) function athena_mp_duvenaud_update_bwd_w(graph, Fi, Fo, min_deg, max_deg, grad_dev, &
          a_dev, dweight_dev) bind(C, name="athena_mp_duvenaud_update_bwd_w")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, grad_dev, a_dev, dweight_dev
       integer(c_int32_t), value :: Fi, Fo, min_deg, max_deg
     end function
     !! update + activation + the readout's p = softmax(R z) in one launch (athena_duvenaud_msgpass_layer.f90:790-803,838-855)
     integer(c_int) function athena_mp_duvenaud_update_readout_fwd(graph, Fi, Fo, min_deg, max_deg, a_dev, weight_dev, act, &
          z_dev, O, R_dev, p_dev) bind(C, name="athena_mp_duvenaud_update_readout_fwd")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, a_dev, weight_dev, z_dev, R_dev, p_dev
       integer(c_int32_t), value :: Fi, Fo, min_deg, max_deg, act, O
     end function
     !! both partials of duvenaud_update from one pass over grad (:284-368)
     integer(c_int) function athena_mp_duvenaud_update_bwd(graph, Fi, Fo, min_deg, max_deg, grad_dev, a_dev, weight_dev, &
          da_dev, dweight_dev) bind(C, name="athena_mp_duvenaud_update_bwd")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, grad_dev, a_dev, weight_dev, da_dev, dweight_dev
       integer(c_int32_t), value :: Fi, Fo, min_deg, max_deg
     end function
     !! the same with da split where it is written: da_x (Fv, n) and da_e (Fe, n) instead of (Fv + Fe, n)
     integer(c_int) function athena_mp_duvenaud_update_bwd_split(graph, Fv, Fe, Fo, min_deg, max_deg, grad_dev, a_dev, weight_dev, &
          da_x_dev, da_e_dev, dweight_dev) bind(C, name="athena_mp_duvenaud_update_bwd_split")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, grad_dev, a_dev, weight_dev, da_x_dev, da_e_dev, dweight_dev
       integer(c_int32_t), value :: Fv, Fe, Fo, min_deg, max_deg
     end function
     !! one time step of the layer's reverse pass in one call: the readout's reverse (softmax -> matmul(R, z) -> activation) and
     !! the update's reverse, dc never in HBM where the shape allows it (Fv = 64, Fv + Fe <= 96, O <= 16)
     integer(c_int) function athena_mp_duvenaud_readout_update_bwd(graph, Fv, Fe, min_deg, max_deg, O, S, seg_dev, z_dev, R_dev, &
          p_dev, gout_dev, dz_next_dev, act, a_dev, weight_dev, da_x_dev, da_e_dev, dweight_dev, dR_dev, accumulate_dR, &
          accumulate_da_e, a_e_dev) bind(C, name="athena_mp_duvenaud_readout_update_bwd")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, seg_dev, z_dev, R_dev, p_dev, gout_dev, dz_next_dev, a_dev, weight_dev, da_x_dev, da_e_dev, &
            dweight_dev, dR_dev, a_e_dev
       integer(c_int32_t), value :: Fv, Fe, min_deg, max_deg, O, S, act, accumulate_dR, accumulate_da_e
     end function
     !! update + activation + the readout's p with a SPLIT: a_x (Fv, n) and a_e (Fe, n), the edge part gathered once per layer
     integer(c_int) function athena_mp_duvenaud_update_readout_fwd_split(graph, Fv, Fe, Fo, min_deg, max_deg, a_x_dev, a_e_dev, &
          weight_dev, act, z_dev, O, R_dev, p_dev) bind(C, name="athena_mp_duvenaud_update_readout_fwd_split")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, a_x_dev, a_e_dev, weight_dev, z_dev, R_dev, p_dev
       integer(c_int32_t), value :: Fv, Fe, Fo, min_deg, max_deg, act, O
     end function
     integer(c_int) function athena_mp_segment_sum(O, N, S, seg_dev, p_dev, out_dev, accumulate) &
          bind(C, name="athena_mp_segment_sum")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       integer(c_int32_t), value :: O, S, accumulate
       integer(c_int64_t), value :: N
       type(c_ptr), value :: seg_dev, p_dev, out_dev
     end function
     integer(c_int) function athena_mp_segment_sum_bwd(O, N, S, seg_dev, gout_dev, dp_dev) &
          bind(C, name="athena_mp_segment_sum_bwd")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       integer(c_int32_t), value :: O, S
       integer(c_int64_t), value :: N
       type(c_ptr), value :: seg_dev, gout_dev, dp_dev
     end function
     integer(c_int) function athena_mp_gno_aggregate_fwd(graph, d, H, Fi, Fo, theta_dev, coords_dev, x_dev, m_dev) &
          bind(C, name="athena_mp_gno_aggregate_fwd")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, theta_dev, coords_dev, x_dev, m_dev
       integer(c_int32_t), value :: d, H, Fi, Fo
     end function
     integer(c_int) function athena_mp_gno_aggregate_bwd_x(graph, d, H, Fi, Fo, theta_dev, coords_dev, grad_dev, &
          dx_dev) bind(C, name="athena_mp_gno_aggregate_bwd_x")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, theta_dev, coords_dev, grad_dev, dx_dev
       integer(c_int32_t), value :: d, H, Fi, Fo
     end function
     !! the same gradient as a PULL over the graph's own rows (row blocks of a partitioned undirected graph):
     !! dx [n_rows, Fi] from grad_ext [n_cols, Fo] -- athena_diffstruc_extd_sub_nop.f90:419-458 read from the receiving side
     integer(c_int) function athena_mp_gno_aggregate_bwd_x_pull(graph, d, H, Fi, Fo, theta_dev, coords_dev, grad_ext_dev, &
          dx_dev) bind(C, name="athena_mp_gno_aggregate_bwd_x_pull")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, theta_dev, coords_dev, grad_ext_dev, dx_dev
       integer(c_int32_t), value :: d, H, Fi, Fo
     end function
     !! the whole reverse pass of gno_aggregate from ONE G = g . Vmat^T: dx, dtheta and (on request) dcoords; any output
     !! may be c_null_ptr; s_save_dev = the S of athena_mp_gno_aggregate_fwd_save or c_null_ptr
     integer(c_int) function athena_mp_gno_aggregate_bwd(graph, d, H, Fi, Fo, theta_dev, coords_dev, x_dev, grad_dev, &
          s_save_dev, dx_dev, dtheta_dev, dcoords_dev, fused) bind(C, name="athena_mp_gno_aggregate_bwd")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, theta_dev, coords_dev, x_dev, grad_dev, s_save_dev, dx_dev, dtheta_dev, dcoords_dev
       integer(c_int32_t), value :: d, H, Fi, Fo
       integer(c_int32_t), intent(out) :: fused
     end function
     integer(c_int) function athena_mp_gno_aggregate_bwd_theta(graph, d, H, Fi, Fo, theta_dev, coords_dev, x_dev, &
          grad_dev, dtheta_dev) bind(C, name="athena_mp_gno_aggregate_bwd_theta")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, theta_dev, coords_dev, x_dev, grad_dev, dtheta_dev
       integer(c_int32_t), value :: d, H, Fi, Fo
     end function
     integer(c_int) function athena_mp_gno_aggregate_bwd_coords(graph, d, H, Fi, Fo, theta_dev, coords_dev, x_dev, &
          grad_dev, dcoords_dev) bind(C, name="athena_mp_gno_aggregate_bwd_coords")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, theta_dev, coords_dev, x_dev, grad_dev, dcoords_dev
       integer(c_int32_t), value :: d, H, Fi, Fo
     end function
     !! training-mode pair: the forward pass keeps S (bytes from _saved_bytes; 0 = shape not served) for dVaug = S^T g
     integer(c_int) function athena_mp_gno_saved_bytes(graph, d, H, Fi, Fo, bytes) bind(C, name="athena_mp_gno_saved_bytes")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       type(c_ptr), value :: graph
       integer(c_int32_t), value :: d, H, Fi, Fo
       integer(c_int64_t), intent(out) :: bytes
     end function
     integer(c_int) function athena_mp_gno_aggregate_fwd_save(graph, d, H, Fi, Fo, theta_dev, coords_dev, x_dev, m_dev, &
          s_save_dev) bind(C, name="athena_mp_gno_aggregate_fwd_save")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, theta_dev, coords_dev, x_dev, m_dev, s_save_dev
       integer(c_int32_t), value :: d, H, Fi, Fo
     end function
     integer(c_int) function athena_mp_gno_aggregate_bwd_theta_saved(graph, d, H, Fi, Fo, theta_dev, coords_dev, x_dev, &
          grad_dev, s_save_dev, dtheta_dev) bind(C, name="athena_mp_gno_aggregate_bwd_theta_saved")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, theta_dev, coords_dev, x_dev, grad_dev, s_save_dev, dtheta_dev
       integer(c_int32_t), value :: d, H, Fi, Fo
     end function
     !! ---- multi-GPU (csrc/comm.hip): one process per GPU, RCCL over xGMI ------------------------------------------
     !! whole reverse pass of one step from its input X (no stored P): dX (may be c_null_ptr) and dW
     integer(c_int) function athena_mp_kipf_layer_bwd(graph, Fi, Fo, dZ_dev, W_dev, X_dev, exact, dX_dev, dW_dev) &
          bind(C, name="athena_mp_kipf_layer_bwd")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, dZ_dev, W_dev, X_dev, dX_dev, dW_dev
       integer(c_int32_t), value :: Fi, Fo, exact
     end function
     integer(c_int) function athena_mp_pull_gemm(graph, Fi, Fo, dZ_dev, W_dev, exact, dX_dev) &
          bind(C, name="athena_mp_pull_gemm")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: graph, dZ_dev, W_dev, dX_dev
       integer(c_int32_t), value :: Fi, Fo, exact
     end function
     integer(c_int) function athena_mp_comm_unique_id(id128) bind(C, name="athena_mp_comm_unique_id")
       import :: c_int, c_char
       character(kind=c_char), intent(inout) :: id128(128)
     end function
     integer(c_int) function athena_mp_comm_create(rank, world, id128, comm) bind(C, name="athena_mp_comm_create")
       import :: c_int, c_int32_t, c_char, c_ptr
       integer(c_int32_t), value :: rank, world
       character(kind=c_char), intent(in) :: id128(128)
       type(c_ptr), intent(out) :: comm
     end function
     integer(c_int) function athena_mp_comm_create_from_file(rank, world, path, comm) &
          bind(C, name="athena_mp_comm_create_from_file")
       import :: c_int, c_int32_t, c_char, c_ptr
       integer(c_int32_t), value :: rank, world
       character(kind=c_char), intent(in) :: path(*)     !! null-terminated
       type(c_ptr), intent(out) :: comm
     end function
     integer(c_int) function athena_mp_comm_destroy(comm) bind(C, name="athena_mp_comm_destroy")
       import :: c_int, c_ptr
       type(c_ptr), value :: comm
     end function
     integer(c_int) function athena_mp_comm_barrier(comm) bind(C, name="athena_mp_comm_barrier")
       import :: c_int, c_ptr
       type(c_ptr), value :: comm
     end function
     integer(c_int) function athena_mp_allreduce(comm, buf_dev, count) bind(C, name="athena_mp_allreduce")
       import :: c_int, c_int64_t, c_ptr
       type(c_ptr), value :: comm, buf_dev
       integer(c_int64_t), value :: count
     end function
     integer(c_int) function athena_mp_allreduce_start(comm, buf_dev, count) bind(C, name="athena_mp_allreduce_start")
       import :: c_int, c_int64_t, c_ptr
       type(c_ptr), value :: comm, buf_dev
       integer(c_int64_t), value :: count
     end function
     integer(c_int) function athena_mp_allreduce_finish(comm) bind(C, name="athena_mp_allreduce_finish")
       import :: c_int, c_ptr
       type(c_ptr), value :: comm
     end function
     !! the rank's rows of graph_type%adj_ia / adj_ja (adj_ja(1,:) = GLOBAL neighbour ids)
     integer(c_int) function athena_mp_shard_create(comm, n_local, nnz, adj_ia, adj_ja, shard) &
          bind(C, name="athena_mp_shard_create")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       type(c_ptr), value :: comm
       integer(c_int32_t), value :: n_local
       integer(c_int64_t), value :: nnz
       integer(c_int32_t), intent(in) :: adj_ia(*), adj_ja(2,*)
       type(c_ptr), intent(out) :: shard
     end function
     !! ... of a graph WITH edge features: adj_ja(2,:) = GLOBAL edge ids (0 = none); the shard's graphs keep the edge
     !! columns renumbered to the rank's own set (athena_mp_shard_export(7) lists their global ids)
     integer(c_int) function athena_mp_shard_create_edges(comm, n_local, nnz, adj_ia, adj_ja, shard) &
          bind(C, name="athena_mp_shard_create_edges")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       type(c_ptr), value :: comm
       integer(c_int32_t), value :: n_local
       integer(c_int64_t), value :: nnz
       integer(c_int32_t), intent(in) :: adj_ia(*), adj_ja(2,*)
       type(c_ptr), intent(out) :: shard
     end function
     !! sum over the partition of a per-edge-column quantity (dcoords): cut columns exchanged with the peer that shares them
     integer(c_int) function athena_mp_shard_edge_reduce(shard, F, e_dev) bind(C, name="athena_mp_shard_edge_reduce")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: shard, e_dev
       integer(c_int32_t), value :: F
     end function
     integer(c_int) function athena_mp_shard_edge_cols(shard, n_edge_cols) bind(C, name="athena_mp_shard_edge_cols")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: shard
       integer(c_int32_t), intent(out) :: n_edge_cols
     end function
     integer(c_int) function athena_mp_shard_destroy(shard) bind(C, name="athena_mp_shard_destroy")
       import :: c_int, c_ptr
       type(c_ptr), value :: shard
     end function
     integer(c_int) function athena_mp_shard_dims(shard, n_local, n_interior, n_halo, nnz, row_offset, n_total) &
          bind(C, name="athena_mp_shard_dims")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       type(c_ptr), value :: shard
       integer(c_int32_t), intent(out) :: n_local, n_interior, n_halo
       integer(c_int64_t), intent(out) :: nnz, row_offset, n_total
     end function
     !! how the halo travels: mode 0 = grouped send / recv of packed rows, 1 = all-gather of whole blocks
     integer(c_int) function athena_mp_shard_info(shard, mode, fraction, tau, recv_rows) bind(C, name="athena_mp_shard_info")
       import :: c_int, c_int32_t, c_int64_t, c_double, c_ptr
       type(c_ptr), value :: shard
       integer(c_int32_t), intent(out) :: mode
       real(c_double), intent(out) :: fraction, tau
       integer(c_int64_t), intent(out) :: recv_rows
     end function
     integer(c_int) function athena_mp_shard_graph(shard, which, graph) bind(C, name="athena_mp_shard_graph")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: shard
       integer(c_int32_t), value :: which
       type(c_ptr), intent(out) :: graph
     end function
     integer(c_int) function athena_mp_shard_export(shard, which, host_dst, capacity, count) &
          bind(C, name="athena_mp_shard_export")
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       type(c_ptr), value :: shard
       integer(c_int32_t), value :: which
       type(*), dimension(*), intent(inout) :: host_dst
       integer(c_int64_t), value :: capacity
       integer(c_int64_t), intent(out) :: count
     end function
     !! the transpose of the exchange: rows computed for remote vertices (y_ext beyond the local rows) go to their owners and
     !! are added into y_local there -- the reverse pass of a partitioned graph in scatter form
     integer(c_int) function athena_mp_halo_reduce_start(shard, slot, F, y_ext_dev) bind(C, name="athena_mp_halo_reduce_start")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: shard, y_ext_dev
       integer(c_int32_t), value :: slot, F
     end function
     integer(c_int) function athena_mp_halo_reduce_finish(shard, slot, y_local_dev) bind(C, name="athena_mp_halo_reduce_finish")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: shard, y_local_dev
       integer(c_int32_t), value :: slot
     end function
     integer(c_int) function athena_mp_halo_start(shard, slot, F, x_ext_dev) bind(C, name="athena_mp_halo_start")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: shard, x_ext_dev
       integer(c_int32_t), value :: slot, F
     end function
     integer(c_int) function athena_mp_halo_finish(shard, slot) bind(C, name="athena_mp_halo_finish")
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: shard
       integer(c_int32_t), value :: slot
     end function
  end interface

contains

  function athena_mp_dev_offset(p, elements) result(q)
    !! device pointer `elements` float32 values past p (row blocks of a resident tensor)
    type(c_ptr), intent(in) :: p
    integer(c_int64_t), intent(in) :: elements
    type(c_ptr) :: q
    q = transfer(transfer(p, 0_c_intptr_t) + 4_c_intptr_t * int(elements, c_intptr_t), q)
  end function athena_mp_dev_offset

  function athena_mp_error_message() result(msg)
    !! last C-side error as a Fortran string (what the layer hands to coreutils' stop_program)
    character(len=:), allocatable :: msg
    type(c_ptr) :: p
    character(kind=c_char), pointer :: s(:)
    integer :: n
    p = athena_mp_last_error()
    if(.not.c_associated(p))then
       msg = ""
       return
    end if
    call c_f_pointer(p, s, [1024])
    n = 0
    do while(n .lt. 1024)
       if(s(n+1) .eq. c_null_char) exit
       n = n + 1
    end do
    allocate(character(len=n) :: msg)
    msg = transfer(s(1:n), msg)
  end function athena_mp_error_message

end module athena_mp_c
