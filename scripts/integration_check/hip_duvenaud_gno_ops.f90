!! The other two layers' autodiff ops of INTEGRATION.md section 2, as ONE compilable source (compile-only, like
!! hip_kipf_msgpass.f90: scripts/integration_check/run.sh puts it through the Fortran compiler against athena's real module
!! sources and compile-only stand-ins for coreutils / diffstruc / graphstruc; nothing is linked or run).
!!   * duvenaud_propagate_hip  contract of duvenaud_propagate        (athena_diffstruc_extd_sub_duvenaud.f90:7-59),
!!                             partials of :115-141 (vertex features) and :143-171 (edge features)
!!   * duvenaud_update_hip     contract of duvenaud_update           (:176-228), partials of :284-324 (a) and :326-368 (weight)
!!   * gno_kernel_hip + gno_aggregate_hip   the pair gno_kernel_eval / gno_aggregate (athena_diffstruc_extd_sub_nop.f90:26-115,
!!                             :330-397; call site athena_graph_nop_layer.f90:743-758) WITHOUT the [F_out F_in, E] tensor
!!                             between them: the first node's value is the parameter vector itself (identity), the second
!!                             does kernel evaluation and aggregation in one device call and differentiates w.r.t. theta
!! The device graph handle and the op's integer arguments travel with the result node in `indices`, where the reference
!! keeps its copies of adj_ia / adj_ja, so that the `pure` partial callbacks find them.
module athena_mp__hip_ops
  use, intrinsic :: iso_c_binding
  use coreutils, only: real32, stop_program
  use diffstruc, only: array_type
  use athena_mp_c
  implicit none
  private
  public :: duvenaud_propagate_hip, duvenaud_update_hip, gno_kernel_hip, gno_aggregate_hip
  public :: handle_of, n_handle

  integer, parameter :: n_handle = storage_size(c_null_ptr) / storage_size(0)   !! default integers that hold a c_ptr

contains

  pure function handle_of(indices) result(h)
    integer, dimension(:), intent(in) :: indices
    type(c_ptr) :: h
    h = transfer(indices(1:n_handle), h)
  end function handle_of

  ! ---------------------------------------------------------------- duvenaud_propagate
  function duvenaud_propagate_hip(vertex_features, edge_features, graph_handle) result(c)
    class(array_type), intent(in), target :: vertex_features, edge_features
    type(c_ptr), intent(in) :: graph_handle
    type(array_type), pointer :: c
    integer(c_int) :: rc
    integer :: Fv, Fe

    Fv = size(vertex_features%val, 1); Fe = size(edge_features%val, 1)
    c => vertex_features%create_result([Fv + Fe, size(vertex_features%val, 2)])
    rc = athena_mp_duvenaud_propagate_fwd_host(graph_handle, int(Fv, c_int32_t), int(Fe, c_int32_t), &
         vertex_features%val, edge_features%val, c%val)
    if(rc .ne. 0) call stop_program("duvenaud_propagate_hip: "//athena_mp_error_message())
    c%indices = [transfer(graph_handle, [0]), Fv, Fe, size(edge_features%val, 2)]
    c%get_partial_left_val => get_partial_duvenaud_propagate_hip_left_val
    c%get_partial_right_val => get_partial_duvenaud_propagate_hip_right_val
    if(vertex_features%requires_grad .or. edge_features%requires_grad)then
       c%requires_grad = .true.
       c%is_forward = vertex_features%is_forward .or. edge_features%is_forward
       c%operation = 'duvenaud_propagate'
       c%left_operand => vertex_features
       c%right_operand => edge_features
       c%owns_left_operand = vertex_features%is_temporary
       c%owns_right_operand = edge_features%is_temporary
    end if
  end function duvenaud_propagate_hip

  pure subroutine get_partial_duvenaud_propagate_hip_left_val(this, upstream_grad, output)
    class(array_type), intent(in) :: this
    real(real32), dimension(:,:), intent(in) :: upstream_grad
    real(real32), dimension(:,:), intent(out) :: output
    integer(c_int) :: rc
    rc = athena_mp_duvenaud_propagate_bwd_x_host(handle_of(this%indices), int(this%indices(n_handle + 1), c_int32_t), &
         int(this%indices(n_handle + 2), c_int32_t), upstream_grad, output)
    if(rc .ne. 0) error stop "duvenaud_propagate_hip: reverse pass (vertex features) failed"
  end subroutine get_partial_duvenaud_propagate_hip_left_val

  pure subroutine get_partial_duvenaud_propagate_hip_right_val(this, upstream_grad, output)
    class(array_type), intent(in) :: this
    real(real32), dimension(:,:), intent(in) :: upstream_grad
    real(real32), dimension(:,:), intent(out) :: output
    integer(c_int) :: rc
    rc = athena_mp_duvenaud_propagate_bwd_e_host(handle_of(this%indices), int(this%indices(n_handle + 1), c_int32_t), &
         int(this%indices(n_handle + 2), c_int32_t), upstream_grad, output)
    if(rc .ne. 0) error stop "duvenaud_propagate_hip: reverse pass (edge features) failed"
  end subroutine get_partial_duvenaud_propagate_hip_right_val

  ! ---------------------------------------------------------------- duvenaud_update
  function duvenaud_update_hip(a, weight, graph_handle, min_degree, max_degree, num_outputs) result(c)
    !! weight holds one [num_outputs, num_inputs] block per degree bucket, flat, as init_duvenaud lays it out
    class(array_type), intent(in), target :: a, weight
    type(c_ptr), intent(in) :: graph_handle
    integer, intent(in) :: min_degree, max_degree, num_outputs
    type(array_type), pointer :: c
    integer(c_int) :: rc
    integer :: Fi

    Fi = size(a%val, 1)
    c => a%create_result([num_outputs, size(a%val, 2)])
    rc = athena_mp_duvenaud_update_fwd_host(graph_handle, int(Fi, c_int32_t), int(num_outputs, c_int32_t), &
         int(min_degree, c_int32_t), int(max_degree, c_int32_t), a%val, weight%val, c%val)
    if(rc .ne. 0) call stop_program("duvenaud_update_hip: "//athena_mp_error_message())
    c%indices = [transfer(graph_handle, [0]), Fi, num_outputs, min_degree, max_degree]
    c%get_partial_left_val => get_partial_duvenaud_update_hip_val
    c%get_partial_right_val => get_partial_duvenaud_update_hip_weight_val
    if(a%requires_grad .or. weight%requires_grad)then
       c%requires_grad = .true.
       c%is_forward = a%is_forward .or. weight%is_forward
       c%operation = 'duvenaud_update'
       c%left_operand => a
       c%right_operand => weight
       c%owns_left_operand = a%is_temporary
       c%owns_right_operand = weight%is_temporary
    end if
  end function duvenaud_update_hip

  pure subroutine get_partial_duvenaud_update_hip_val(this, upstream_grad, output)
    !! da[:, v] = (g[:, v]^T W_d) / d_v (:284-324); the weights are the node's right operand, a its left operand.
    !! grad_reverse asks for the two partials of this node one after the other; the fused reverse kernel produces both from one
    !! pass over upstream_grad (athena_mp_duvenaud_update_bwd).  The pair entry point computes both on the first of the two
    !! requests, returns the one asked for and parks the other on the device; the second request (same handle, shapes and
    !! operand content) takes it.  Nothing is kept on the Fortran side: this callback is `pure` and `this` is intent(in).
    class(array_type), intent(in) :: this
    real(real32), dimension(:,:), intent(in) :: upstream_grad
    real(real32), dimension(:,:), intent(out) :: output
    integer(c_int) :: rc
    rc = athena_mp_duvenaud_update_bwd_pair_host(handle_of(this%indices), int(this%indices(n_handle + 1), c_int32_t), &
         int(this%indices(n_handle + 2), c_int32_t), int(this%indices(n_handle + 3), c_int32_t), &
         int(this%indices(n_handle + 4), c_int32_t), ATHENA_MP_ACT_NONE, this%val, upstream_grad, this%left_operand%val, &
         this%right_operand%val, 0_c_int32_t, output)
    if(rc .ne. 0) error stop "duvenaud_update_hip: reverse pass (a) failed"
  end subroutine get_partial_duvenaud_update_hip_val

  pure subroutine get_partial_duvenaud_update_hip_weight_val(this, upstream_grad, output)
    !! dW_d[i, j] += g[i, v] a[j, v] / d_v (:326-368): the other half of the pair (which = 1)
    class(array_type), intent(in) :: this
    real(real32), dimension(:,:), intent(in) :: upstream_grad
    real(real32), dimension(:,:), intent(out) :: output
    integer(c_int) :: rc
    rc = athena_mp_duvenaud_update_bwd_pair_host(handle_of(this%indices), int(this%indices(n_handle + 1), c_int32_t), &
         int(this%indices(n_handle + 2), c_int32_t), int(this%indices(n_handle + 3), c_int32_t), &
         int(this%indices(n_handle + 4), c_int32_t), ATHENA_MP_ACT_NONE, this%val, upstream_grad, this%left_operand%val, &
         this%right_operand%val, 1_c_int32_t, output)
    if(rc .ne. 0) error stop "duvenaud_update_hip: reverse pass (weight) failed"
  end subroutine get_partial_duvenaud_update_hip_weight_val

  ! ---------------------------------------------------------------- gno_kernel_eval + gno_aggregate
  function gno_kernel_hip(coords, kernel_params) result(k)
    !! Stands where gno_kernel_eval stands in the graph (athena_graph_nop_layer.f90:743-750) but evaluates nothing: its value
    !! is the packed parameter vector itself (identity in kernel_params), its left operand the edge geometry, so that the
    !! aggregate node downstream finds both and dC/dk = dC/dtheta.  The [F_out F_in, E] kernel tensor is never formed.
    class(array_type), intent(in), target :: coords, kernel_params
    type(array_type), pointer :: k

    k => kernel_params%create_result()
    k%val = kernel_params%val
    k%get_partial_right_val => get_partial_gno_kernel_hip_params_val
    if(kernel_params%requires_grad)then
       k%requires_grad = .true.
       k%is_forward = kernel_params%is_forward
       k%operation = 'gno_kernel'
    end if
    k%left_operand => coords          ! geometry: not differentiated by the layer (requires_grad = .false., :781-785)
    k%right_operand => kernel_params
    k%owns_right_operand = kernel_params%is_temporary
  end function gno_kernel_hip

  pure subroutine get_partial_gno_kernel_hip_params_val(this, upstream_grad, output)
    class(array_type), intent(in) :: this
    real(real32), dimension(:,:), intent(in) :: upstream_grad
    real(real32), dimension(:,:), intent(out) :: output
    output = upstream_grad
  end subroutine get_partial_gno_kernel_hip_params_val

  function gno_aggregate_hip(features, kernel, graph_handle, coord_dim, kernel_hidden, F_in, F_out) result(c)
    !! c[:, i] = sum over row i's entries of reshape(kappa_e, [F_out, F_in]) . x_j, kappa_e = V relu(U dx_e + b_u) + b_v
    !! (gno_aggregate, athena_diffstruc_extd_sub_nop.f90:330-397, with gno_kernel_eval :26-115 folded in);
    !! `kernel` is the node gno_kernel_hip made: value = theta, left operand = geometry [coord_dim, E]
    class(array_type), intent(in), target :: features, kernel
    type(c_ptr), intent(in) :: graph_handle
    integer, intent(in) :: coord_dim, kernel_hidden, F_in, F_out
    type(array_type), pointer :: c
    integer(c_int) :: rc

    c => features%create_result([F_out, size(features%val, 2)])
    rc = athena_mp_gno_aggregate_fwd_host(graph_handle, int(coord_dim, c_int32_t), int(kernel_hidden, c_int32_t), &
         int(F_in, c_int32_t), int(F_out, c_int32_t), kernel%val, kernel%left_operand%val, features%val, c%val)
    if(rc .ne. 0) call stop_program("gno_aggregate_hip: "//athena_mp_error_message())
    c%indices = [transfer(graph_handle, [0]), coord_dim, kernel_hidden, F_in, F_out]
    c%get_partial_left_val => get_partial_gno_aggregate_hip_features_val
    c%get_partial_right_val => get_partial_gno_aggregate_hip_kernel_val
    if(features%requires_grad .or. kernel%requires_grad)then
       c%requires_grad = .true.
       c%is_forward = features%is_forward .or. kernel%is_forward
       c%operation = 'gno_aggregate'
       c%left_operand => features
       c%right_operand => kernel
       c%owns_left_operand = features%is_temporary
       c%owns_right_operand = kernel%is_temporary
    end if
  end function gno_aggregate_hip

  pure subroutine get_partial_gno_aggregate_hip_features_val(this, upstream_grad, output)
    !! dx (get_partial_gno_agg_features_val, athena_diffstruc_extd_sub_nop.f90:419-458) -- one half of the pair: the whole
    !! reverse pass of gno_aggregate comes from ONE G = g . Vmat^T (athena_mp_gno_aggregate_bwd); whichever of the node's two
    !! partials grad_reverse asks for first triggers it, the other is handed over from the device
    class(array_type), intent(in) :: this
    real(real32), dimension(:,:), intent(in) :: upstream_grad
    real(real32), dimension(:,:), intent(out) :: output
    integer(c_int) :: rc
    rc = athena_mp_gno_aggregate_bwd_pair_host(handle_of(this%indices), int(this%indices(n_handle + 1), c_int32_t), &
         int(this%indices(n_handle + 2), c_int32_t), int(this%indices(n_handle + 3), c_int32_t), &
         int(this%indices(n_handle + 4), c_int32_t), this%right_operand%val, this%right_operand%left_operand%val, &
         this%left_operand%val, upstream_grad, 0_c_int32_t, output)
    if(rc .ne. 0) error stop "gno_aggregate_hip: reverse pass (features) failed"
  end subroutine get_partial_gno_aggregate_hip_features_val

  pure subroutine get_partial_gno_aggregate_hip_kernel_val(this, upstream_grad, output)
    !! the gradient w.r.t. the kernel node's value, i.e. w.r.t. theta: agg -> kernels (:480-526) chained with
    !! kernel -> params (:235-325) on the device -- the other half of the pair (which = 1)
    class(array_type), intent(in) :: this
    real(real32), dimension(:,:), intent(in) :: upstream_grad
    real(real32), dimension(:,:), intent(out) :: output
    integer(c_int) :: rc
    rc = athena_mp_gno_aggregate_bwd_pair_host(handle_of(this%indices), int(this%indices(n_handle + 1), c_int32_t), &
         int(this%indices(n_handle + 2), c_int32_t), int(this%indices(n_handle + 3), c_int32_t), &
         int(this%indices(n_handle + 4), c_int32_t), this%right_operand%val, this%right_operand%left_operand%val, &
         this%left_operand%val, upstream_grad, 1_c_int32_t, output)
    if(rc .ne. 0) error stop "gno_aggregate_hip: reverse pass (kernel parameters) failed"
  end subroutine get_partial_gno_aggregate_hip_kernel_val

end module athena_mp__hip_ops
