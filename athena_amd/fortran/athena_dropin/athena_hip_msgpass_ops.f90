!! src/athena/athena_hip_msgpass_ops.f90 of an athena checkout (athena_dropin/install.sh puts it there): the autodiff ops of
!! the three message-passing layers on libathena_mp.so.  Needs nothing of athena but diffstruc's array_type and the binding
!! module athena_mp_c.  Every op keeps the contract of the procedure it stands in for -- a result node from create_result, the
!! value from one device call, `pure` get_partial_*_val callbacks (legal: the bind(C) interfaces are declared pure) -- so the
!! tape diffstruc builds over them has the shape of the reference's:
!!   * kipf_propagate_hip      contract of kipf_propagate            (athena_diffstruc_extd_sub_kipf.f90:7-59), partial :85-111
!!   * matmul_hip              the dense step matmul(params(t), ptr2) (athena_kipf_msgpass_layer.f90:951, athena_graph_nop_layer
!!                             .f90:761, athena_duvenaud_msgpass_layer.f90:842) on the MFMA kernels, with the activation as the
!!                             GEMM's epilogue when the device applies it (none / relu / sigmoid / tanh)
!!   * duvenaud_propagate_hip  contract of duvenaud_propagate        (athena_diffstruc_extd_sub_duvenaud.f90:7-59),
!!                             partials of :115-141 (vertex features) and :143-171 (edge features)
!!   * duvenaud_update_hip     contract of duvenaud_update           (:176-228), partials of :284-324 (a) and :326-368 (weight)
!!   * duvenaud_update_act_readout_hip   update + message activation + the readout's softmax(matmul(R, z)) in ONE launch; the
!!                             readout node's two partials (softmax reverse, then the matmul's) as callbacks of their own
!!   * gno_kernel_hip + gno_aggregate_hip   the pair gno_kernel_eval / gno_aggregate (athena_diffstruc_extd_sub_nop.f90:26-115,
!!                             :330-397; call site athena_graph_nop_layer.f90:743-758) WITHOUT the [F_out F_in, E] tensor
!!                             between them: the first node's value is the parameter vector itself (identity), the second
!!                             does kernel evaluation and aggregation in one device call and differentiates w.r.t. theta
!! The device graph handle and the op's integer arguments travel with the result node in `indices`, where the reference
!! keeps its copies of adj_ia / adj_ja, so that the `pure` partial callbacks find them.
!! scripts/integration_check/run.sh compiles this file against athena's real modules and links it into run_ops / run_layers,
!! which run every op and every callback on the GPU (tests/test_gpu_integration_run.py).
module athena__hip_msgpass_ops
  use, intrinsic :: iso_c_binding
  use coreutils, only: real32, stop_program
  use diffstruc, only: array_type
  use athena_mp_c
  implicit none
  private
  public :: kipf_propagate_hip, matmul_hip
  public :: duvenaud_propagate_hip, duvenaud_update_hip, gno_kernel_hip, gno_aggregate_hip
  public :: duvenaud_update_act_readout_hip, get_partial_readout_softmax_hip_z_val, get_partial_readout_softmax_hip_weight_val
  public :: handle_of, n_handle

  integer, parameter :: n_handle = storage_size(c_null_ptr) / storage_size(0)   !! default integers that hold a c_ptr

contains

  pure function handle_of(indices) result(h)
    integer, dimension(:), intent(in) :: indices
    type(c_ptr) :: h
    h = transfer(indices(1:n_handle), h)
  end function handle_of

  ! ---------------------------------------------------------------- kipf_propagate
  function kipf_propagate_hip(vertex_features, graph_handle) result(c)
    !! HIP-backed kipf_propagate: c%val = A^ x on the device graph `graph_handle`
    class(array_type), intent(in), target :: vertex_features
    type(c_ptr), intent(in) :: graph_handle
    type(array_type), pointer :: c
    integer(c_int) :: rc

    c => vertex_features%create_result()
    rc = athena_mp_kipf_propagate_fwd_host(graph_handle, int(size(vertex_features%val, 1), c_int32_t), &
         vertex_features%val, c%val)
    if(rc .ne. 0) call stop_program("kipf_propagate_hip: "//athena_mp_error_message())
    ! the handle travels with the node in place of copies of adj_ia / adj_ja (athena_diffstruc_extd_sub_kipf.f90:48-49)
    c%indices = transfer(graph_handle, [0])
    c%get_partial_left_val => get_partial_kipf_propagate_hip_left_val
    if(vertex_features%requires_grad)then
       c%requires_grad = .true.
       c%is_forward = vertex_features%is_forward
       c%operation = 'kipf_propagate'
       c%left_operand => vertex_features
       c%owns_left_operand = vertex_features%is_temporary
    end if
  end function kipf_propagate_hip

  pure subroutine get_partial_kipf_propagate_hip_left_val(this, upstream_grad, output)
    !! get_partial_kipf_propagate_left_val (:85-111): the reference's coefficient-free scatter (exact = 0).
    !! `pure`, as diffstruc's callback interface demands -- legal because the bind(C) interface is declared pure.
    class(array_type), intent(in) :: this
    real(real32), dimension(:,:), intent(in) :: upstream_grad
    real(real32), dimension(:,:), intent(out) :: output
    type(c_ptr) :: graph_handle
    integer(c_int) :: rc
    graph_handle = transfer(this%indices, graph_handle)
    rc = athena_mp_kipf_propagate_bwd_host(graph_handle, int(size(upstream_grad, 1), c_int32_t), &
         upstream_grad, output, 0_c_int32_t)
    if(rc .ne. 0) error stop "kipf_propagate_hip: reverse pass failed"
  end subroutine get_partial_kipf_propagate_hip_left_val

  ! ---------------------------------------------------------------- the dense step
  function matmul_hip(weight, input, act) result(c)
    !! c = act(W input) with W = params(t) held flat as [F_out F_in, 1] (column-major W(F_out, F_in), which is the row-major
    !! Wt[F_in][F_out] the kernel reads) and input [F_in, N]: matmul(params(t), ptr2), athena_kipf_msgpass_layer.f90:951,
    !! followed by this%activation%apply(ptr3) (:952) when `act` is one of the device's epilogues.  Left operand = the
    !! weights, right operand = the features, as diffstruc's matmul orders them.
    class(array_type), intent(in), target :: weight, input
    integer(c_int32_t), intent(in), optional :: act
    type(array_type), pointer :: c
    integer(c_int) :: rc
    integer(c_int32_t) :: act_
    integer :: Fi, Fo

    act_ = ATHENA_MP_ACT_NONE
    if(present(act)) act_ = act
    Fi = size(input%val, 1)
    Fo = size(weight%val, 1) / Fi
    if(Fo * Fi .ne. size(weight%val, 1)) call stop_program("matmul_hip: the weights do not hold [F_out, F_in]")
    c => input%create_result([Fo, size(input%val, 2)])
    rc = athena_mp_gemm_fwd_host(int(size(input%val, 2), c_int64_t), int(Fi, c_int32_t), int(Fo, c_int32_t), input%val, &
         weight%val, c_null_ptr, act_, c%val)
    if(rc .ne. 0) call stop_program("matmul_hip: "//athena_mp_error_message())
    c%indices = [Fi, Fo, int(act_)]
    c%get_partial_left_val => get_partial_matmul_hip_weight_val
    c%get_partial_right_val => get_partial_matmul_hip_input_val
    if(weight%requires_grad .or. input%requires_grad)then
       c%requires_grad = .true.
       c%is_forward = weight%is_forward .or. input%is_forward
       c%operation = 'matmul'
       c%left_operand => weight
       c%right_operand => input
       c%owns_left_operand = weight%is_temporary
       c%owns_right_operand = input%is_temporary
    end if
  end function matmul_hip

  pure subroutine get_partial_matmul_hip_weight_val(this, upstream_grad, output)
    !! dW = (act'(y) * g) P^T, flat as params(t)%val(:,1)
    class(array_type), intent(in) :: this
    real(real32), dimension(:,:), intent(in) :: upstream_grad
    real(real32), dimension(:,:), intent(out) :: output
    real(real32), allocatable :: dz(:,:)
    integer(c_int) :: rc
    integer(c_int64_t) :: N
    N = int(size(upstream_grad, 2), c_int64_t)
    if(this%indices(3) .eq. ATHENA_MP_ACT_NONE)then
       rc = athena_mp_gemm_dw_host(N, int(this%indices(1), c_int32_t), int(this%indices(2), c_int32_t), &
            this%right_operand%val, upstream_grad, output)
    else
       allocate(dz(size(upstream_grad, 1), size(upstream_grad, 2)))
       rc = athena_mp_activation_bwd_host(int(this%indices(3), c_int32_t), N * size(upstream_grad, 1), this%val, upstream_grad, dz)
       if(rc .eq. 0) rc = athena_mp_gemm_dw_host(N, int(this%indices(1), c_int32_t), int(this%indices(2), c_int32_t), &
            this%right_operand%val, dz, output)
    end if
    if(rc .ne. 0) error stop "matmul_hip: reverse pass (weights) failed"
  end subroutine get_partial_matmul_hip_weight_val

  pure subroutine get_partial_matmul_hip_input_val(this, upstream_grad, output)
    !! dP = W^T (act'(y) * g)
    class(array_type), intent(in) :: this
    real(real32), dimension(:,:), intent(in) :: upstream_grad
    real(real32), dimension(:,:), intent(out) :: output
    real(real32), allocatable :: dz(:,:)
    integer(c_int) :: rc
    integer(c_int64_t) :: N
    N = int(size(upstream_grad, 2), c_int64_t)
    if(this%indices(3) .eq. ATHENA_MP_ACT_NONE)then
       rc = athena_mp_gemm_dx_host(N, int(this%indices(1), c_int32_t), int(this%indices(2), c_int32_t), upstream_grad, &
            this%left_operand%val, output)
    else
       allocate(dz(size(upstream_grad, 1), size(upstream_grad, 2)))
       rc = athena_mp_activation_bwd_host(int(this%indices(3), c_int32_t), N * size(upstream_grad, 1), this%val, upstream_grad, dz)
       if(rc .eq. 0) rc = athena_mp_gemm_dx_host(N, int(this%indices(1), c_int32_t), int(this%indices(2), c_int32_t), dz, &
            this%left_operand%val, output)
    end if
    if(rc .ne. 0) error stop "matmul_hip: reverse pass (features) failed"
  end subroutine get_partial_matmul_hip_input_val

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

  ! ---------------------------------------------------------------- update + activation + readout softmax, fused
  function duvenaud_update_act_readout_hip(a, weight, readout_weight, p, graph_handle, min_degree, max_degree, &
       num_outputs, act) result(c)
    !! ONE launch for three statements of the reference: duvenaud_update (athena_diffstruc_extd_sub_duvenaud.f90:176-228), the
    !! message activation (athena_duvenaud_msgpass_layer.f90:800-802) and the readout's per-vertex softmax(matmul(R, z))
    !! (:842-849).  The result node stands where the ACTIVATION's node stands in the reference's tape: value z, left operand a,
    !! right operand the bucketed weights; its partials are the update's, with the activation's factor act'(z) applied on the
    !! device in the same call.  p%val receives softmax(R z) and is wired up as a node by the caller (it needs z where the layer
    !! keeps it).
    class(array_type), intent(in), target :: a, weight, readout_weight
    type(array_type), intent(inout) :: p
    type(c_ptr), intent(in) :: graph_handle
    integer, intent(in) :: min_degree, max_degree, num_outputs
    integer(c_int32_t), intent(in) :: act
    type(array_type), pointer :: c
    integer(c_int) :: rc
    integer :: Fi, O, N

    Fi = size(a%val, 1); N = size(a%val, 2)
    O = size(readout_weight%val, 1) / num_outputs            ! params(T+t)%val(:,1) = R(O, F_v) flat
    c => a%create_result([num_outputs, N])
    if(allocated(p%val))then
       if(any(shape(p%val) .ne. [O, N])) deallocate(p%val)
    end if
    if(.not.allocated(p%val)) allocate(p%val(O, N))
    rc = athena_mp_duvenaud_update_readout_fwd_host(graph_handle, int(Fi, c_int32_t), int(num_outputs, c_int32_t), &
         int(min_degree, c_int32_t), int(max_degree, c_int32_t), a%val, weight%val, act, c%val, int(O, c_int32_t), &
         readout_weight%val, p%val)
    if(rc .ne. 0) call stop_program("duvenaud_update_act_readout_hip: "//athena_mp_error_message())
    c%indices = [transfer(graph_handle, [0]), Fi, num_outputs, min_degree, max_degree, int(act)]
    c%get_partial_left_val => get_partial_update_act_hip_val
    c%get_partial_right_val => get_partial_update_act_hip_weight_val
    if(a%requires_grad .or. weight%requires_grad)then
       c%requires_grad = .true.
       c%is_forward = a%is_forward .or. weight%is_forward
       c%operation = 'duvenaud_update_act'
       c%left_operand => a
       c%right_operand => weight
       c%owns_left_operand = a%is_temporary
       c%owns_right_operand = weight%is_temporary
    end if
  end function duvenaud_update_act_readout_hip

  pure subroutine get_partial_update_act_hip_val(this, upstream_grad, output)
    !! da = ((act'(z) * g)^T W_d) / d  -- :284-324 behind the activation's reverse.  grad_reverse asks this node for its two
    !! partials one after the other; athena_mp_duvenaud_update_bwd produces both from ONE pass over the gradient rows.  The
    !! pair entry point runs that pass on the first request, returns the partial asked for and parks the other on the device;
    !! the second request -- same handle, same shapes, same CONTENT of (upstream_grad, z, a, weights) -- is served from there.
    class(array_type), intent(in) :: this
    real(real32), dimension(:,:), intent(in) :: upstream_grad
    real(real32), dimension(:,:), intent(out) :: output
    integer(c_int) :: rc
    rc = athena_mp_duvenaud_update_bwd_pair_host(handle_of(this%indices), int(this%indices(n_handle + 1), c_int32_t), &
         int(this%indices(n_handle + 2), c_int32_t), int(this%indices(n_handle + 3), c_int32_t), &
         int(this%indices(n_handle + 4), c_int32_t), int(this%indices(n_handle + 5), c_int32_t), this%val, upstream_grad, &
         this%left_operand%val, this%right_operand%val, 0_c_int32_t, output)
    if(rc .ne. 0) error stop "duvenaud_update_act_readout_hip: reverse pass (a) failed"
  end subroutine get_partial_update_act_hip_val

  pure subroutine get_partial_update_act_hip_weight_val(this, upstream_grad, output)
    !! dW_d += (act'(z) * g) a^T / d  (:326-368): the other half of the same device pass (which = 1)
    class(array_type), intent(in) :: this
    real(real32), dimension(:,:), intent(in) :: upstream_grad
    real(real32), dimension(:,:), intent(out) :: output
    integer(c_int) :: rc
    rc = athena_mp_duvenaud_update_bwd_pair_host(handle_of(this%indices), int(this%indices(n_handle + 1), c_int32_t), &
         int(this%indices(n_handle + 2), c_int32_t), int(this%indices(n_handle + 3), c_int32_t), &
         int(this%indices(n_handle + 4), c_int32_t), int(this%indices(n_handle + 5), c_int32_t), this%val, upstream_grad, &
         this%left_operand%val, this%right_operand%val, 1_c_int32_t, output)
    if(rc .ne. 0) error stop "duvenaud_update_act_readout_hip: reverse pass (weight) failed"
  end subroutine get_partial_update_act_hip_weight_val

  pure subroutine get_partial_readout_softmax_hip_z_val(this, upstream_grad, output)
    !! the node p = softmax(matmul(R, z)) (athena_duvenaud_msgpass_layer.f90:842-849 as one node): partial w.r.t. z.
    !!   dl = p * (g - <g, p>) per vertex  (softmax reverse at the output, athena_diffstruc_extd_sub.f90:331-379)
    !!   dz = R^T dl                       (matmul reverse w.r.t. its right operand)
    class(array_type), intent(in) :: this
    real(real32), dimension(:,:), intent(in) :: upstream_grad
    real(real32), dimension(:,:), intent(out) :: output
    real(real32), allocatable :: dl(:,:)
    integer(c_int) :: rc
    integer :: O, N
    O = size(this%val, 1); N = size(this%val, 2)
    allocate(dl(O, N))
    rc = athena_mp_softmax_bwd_host(int(N, c_int64_t), int(O, c_int32_t), this%val, upstream_grad, dl)
    if(rc .eq. 0) rc = athena_mp_gemm_dx_host(int(N, c_int64_t), int(size(output, 1), c_int32_t), int(O, c_int32_t), dl, &
         this%right_operand%val, output)
    if(rc .ne. 0) error stop "duvenaud readout (hip): reverse pass (z) failed"
  end subroutine get_partial_readout_softmax_hip_z_val

  pure subroutine get_partial_readout_softmax_hip_weight_val(this, upstream_grad, output)
    !! partial w.r.t. the readout matrix: dR = dl z^T, flat as params(T+t)%val(:,1) = R(O, F_v)
    class(array_type), intent(in) :: this
    real(real32), dimension(:,:), intent(in) :: upstream_grad
    real(real32), dimension(:,:), intent(out) :: output
    real(real32), allocatable :: dl(:,:)
    integer(c_int) :: rc
    integer :: O, N
    O = size(this%val, 1); N = size(this%val, 2)
    allocate(dl(O, N))
    rc = athena_mp_softmax_bwd_host(int(N, c_int64_t), int(O, c_int32_t), this%val, upstream_grad, dl)
    if(rc .eq. 0) rc = athena_mp_gemm_dw_host(int(N, c_int64_t), int(size(this%left_operand%val, 1), c_int32_t), &
         int(O, c_int32_t), this%left_operand%val, dl, output)
    if(rc .ne. 0) error stop "duvenaud readout (hip): reverse pass (weight) failed"
  end subroutine get_partial_readout_softmax_hip_weight_val

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

end module athena__hip_msgpass_ops
