!! LINKED AND RUN on the GPU (tests/test_gpu_integration_run.py): the autodiff ops of
!! athena_amd/fortran/athena_dropin/athena_hip_msgpass_ops.f90 -- the source a maintainer adds to athena -- driven from Fortran
!! through the stand-in's working tape (standins.f90) with diffstruc's callback protocol.  For a Duvenaud step (propagate ->
!! update), the dense step and a graph-neural-operator aggregate it builds the nodes, runs the reverse pass (every node asked
!! once for its left partial, then for its right one, each through its `pure` callback) and holds every leaf gradient against
!! the op-granular C entry points called directly; athena_mp_pair_stats must show ONE fused device pass and ONE hand-over per
!! two-partial node.  Prints "RUN_OPS_OK <fused passes> <hand-overs>" or stops with a message.
program run_ops
  use, intrinsic :: iso_c_binding
  use coreutils, only: real32
  use diffstruc, only: array_type, weighted_sum, operator(+)
  use athena_mp_c
  use athena__hip_msgpass_ops
  implicit none
  integer(c_int64_t) :: f0, h0, f1, h1
  integer(c_int64_t) :: fused_total, handed_total

  if(athena_mp_init(0_c_int) .ne. 0) call fail("athena_mp_init")
  if(athena_mp_pair_stats(f0, h0) .ne. 0) call fail("pair_stats")
  call duvenaud_case(n=40, fv=8, fe=2, fo=4, resident=.false.)          ! VALU kernels
  call duvenaud_case(n=3000, fv=64, fe=8, fo=64, resident=.false.)      ! the fused MFMA reverse kernel (configs[2]'s widths)
  call duvenaud_case(n=3000, fv=64, fe=8, fo=64, resident=.true.)       ! ... with every %val resident in HBM between the calls
  call kipf_case(n=500, f=32)
  call matmul_case(n=2000, fi=64, fo=128, act=ATHENA_MP_ACT_RELU)             ! the MFMA dense step with its epilogue
  call matmul_case(n=37, fi=6, fo=10, act=ATHENA_MP_ACT_NONE)
  call fused_duvenaud_case(n=3000, fv=64, fe=8, fo=64, no=10)           ! ONE launch for update + sigmoid + softmax(R z): z and p as nodes
  call gno_case(n=600, d=3, h=64, fi=64, fo=64)                         ! the one-contraction reverse pass
  call gno_case(n=50, d=2, h=7, fi=5, fo=9)                             ! generic shapes (separate entry points behind the pair)
  if(athena_mp_pair_stats(f1, h1) .ne. 0) call fail("pair_stats")
  fused_total = f1 - f0
  handed_total = h1 - h0
  if(fused_total .ne. 6 .or. handed_total .ne. 6) then
     write(0, *) "pair slots: fused passes", fused_total, " hand-overs", handed_total, " (expected 6 and 6)"
     error stop 1
  end if
  if(athena_mp_finalize() .ne. 0) call fail("finalize")
  write(*, '(A,I0,1X,I0)') "RUN_OPS_OK ", fused_total, handed_total

contains

  subroutine fail(what)
    character(*), intent(in) :: what
    write(0, *) what//" failed: "//athena_mp_error_message()
    error stop 1
  end subroutine fail

  subroutine chain_graph(n, ia, ja, n_edges)
    !! a chain with a self loop on every vertex, as add_self_loops leaves it: row i lists (i, edge id 0), (i-1, edge i-1), (i+1, edge i)
    integer, intent(in) :: n
    integer, allocatable, intent(out) :: ia(:), ja(:,:)
    integer, intent(out) :: n_edges
    integer :: i, w
    n_edges = n - 1
    allocate(ia(n + 1), ja(2, 3 * n - 2))
    w = 0
    do i = 1, n
       ia(i) = w + 1
       w = w + 1; ja(:, w) = [i, 0]
       if(i .gt. 1)then
          w = w + 1; ja(:, w) = [i - 1, i - 1]
       end if
       if(i .lt. n)then
          w = w + 1; ja(:, w) = [i + 1, i]
       end if
    end do
    ia(n + 1) = w + 1
  end subroutine chain_graph

  subroutine fill(a, seed)
    real(real32), intent(out) :: a(:,:)
    integer, intent(in) :: seed
    integer :: i, j
    do j = 1, size(a, 2)
       do i = 1, size(a, 1)
          a(i, j) = real(mod(7 * i + 13 * j + 31 * seed + i * j, 97), real32) / 97._real32 - 0.45_real32
       end do
    end do
  end subroutine fill

  subroutine close_to(got, want, what)
    real(real32), intent(in) :: got(:,:), want(:,:)
    character(*), intent(in) :: what
    real(real32) :: scale
    scale = max(maxval(abs(want)), 1.e-30_real32)
    if(any(shape(got) .ne. shape(want)) .or. maxval(abs(got - want)) .gt. 1.e-5_real32 * scale)then
       write(0, *) what, ": worst", maxval(abs(got - want)) / scale
       error stop 1
    end if
  end subroutine close_to

  subroutine duvenaud_case(n, fv, fe, fo, resident)
    integer, intent(in) :: n, fv, fe, fo
    logical, intent(in) :: resident
    integer, allocatable :: ia(:), ja(:,:)
    integer :: ne, mn, mx
    type(c_ptr) :: handle
    type(array_type), target :: x, e, w
    type(array_type), pointer :: a, c
    real(real32), allocatable :: up(:,:), da(:,:), dw(:,:), dx(:,:), de(:,:)
    integer(c_int) :: rc

    mn = 1; mx = 4
    call chain_graph(n, ia, ja, ne)
    if(athena_mp_graph_acquire(int(n, c_int32_t), int(size(ja, 2), c_int64_t), ia, ja, int(ne, c_int32_t), handle) .ne. 0) call fail("graph_acquire")
    allocate(x%val(fv, n), e%val(fe, ne), w%val(fo * (fv + fe) * (mx - mn + 1), 1), up(fo, n))
    call fill(x%val, 1); call fill(e%val, 2); call fill(w%val, 3); call fill(up, 4)
    x%requires_grad = .true.; e%requires_grad = .true.; w%requires_grad = .true.
    x%is_temporary = .false.; e%is_temporary = .false.; w%is_temporary = .false.
    if(resident)then
       if(athena_mp_resident_mode(1_c_int32_t) .ne. 0) call fail("resident_mode")
    end if
    a => duvenaud_propagate_hip(x, e, handle)
    c => duvenaud_update_hip(a, w, handle, mn, mx, fo)
    call c%grad_reverse_from(up)                                ! left partial, then right partial, of every node
    if(resident)then
       if(athena_mp_resident_mode(0_c_int32_t) .ne. 0) call fail("resident_mode off")     ! everything goes home
    end if
    if(c%partial_calls .ne. 2 .or. a%partial_calls .ne. 2) call fail("grad_reverse did not ask every node for both partials")
    ! the same gradients from the op-granular entry points, called directly on the same host arrays
    allocate(da(fv + fe, n), dw(size(w%val, 1), 1), dx(fv, n), de(fe, ne))
    rc = athena_mp_duvenaud_update_bwd_a_host(handle, int(fv + fe, c_int32_t), int(fo, c_int32_t), int(mn, c_int32_t), int(mx, c_int32_t), up, w%val, da)
    if(rc .eq. 0) rc = athena_mp_duvenaud_update_bwd_w_host(handle, int(fv + fe, c_int32_t), int(fo, c_int32_t), int(mn, c_int32_t), int(mx, c_int32_t), &
         up, a%val, dw)
    if(rc .eq. 0) rc = athena_mp_duvenaud_propagate_bwd_x_host(handle, int(fv, c_int32_t), int(fe, c_int32_t), da, dx)
    if(rc .eq. 0) rc = athena_mp_duvenaud_propagate_bwd_e_host(handle, int(fv, c_int32_t), int(fe, c_int32_t), da, de)
    if(rc .ne. 0) call fail("direct reverse entry points")
    call close_to(a%grad%val, da, "duvenaud: da through the tape")
    call close_to(w%grad%val, dw, "duvenaud: dW through the tape")
    call close_to(x%grad%val, dx, "duvenaud: dx through the tape")
    call close_to(e%grad%val, de, "duvenaud: de through the tape")
    if(athena_mp_graph_release(handle) .ne. 0) call fail("graph_release")
  end subroutine duvenaud_case

  subroutine kipf_case(n, f)
    integer, intent(in) :: n, f
    integer, allocatable :: ia(:), ja(:,:)
    integer :: ne
    type(c_ptr) :: handle
    type(array_type), target :: x
    type(array_type), pointer :: c
    real(real32), allocatable :: up(:,:), dx(:,:)

    call chain_graph(n, ia, ja, ne)
    ! as set_graph_hip_kipf acquires it: the handle carries graph%num_edges though Kipf reads no edge features
    if(athena_mp_graph_acquire(int(n, c_int32_t), int(size(ja, 2), c_int64_t), ia, ja, int(ne, c_int32_t), handle) .ne. 0) call fail("graph_acquire")
    allocate(x%val(f, n), up(f, n), dx(f, n))
    call fill(x%val, 11); call fill(up, 12)
    x%requires_grad = .true.; x%is_temporary = .false.
    c => kipf_propagate_hip(x, handle)
    call c%grad_reverse_from(up)
    if(athena_mp_kipf_propagate_bwd_host(handle, int(f, c_int32_t), up, dx, 0_c_int32_t) .ne. 0) call fail("kipf_propagate_bwd_host")
    call close_to(x%grad%val, dx, "kipf: dx through the tape")
    if(athena_mp_graph_release(handle) .ne. 0) call fail("graph_release")
  end subroutine kipf_case

  subroutine matmul_case(n, fi, fo, act)
    !! matmul_hip: c = act(W p), dW and dp through the tape against gemm_dw / gemm_dx behind activation_bwd called directly
    integer, intent(in) :: n, fi, fo
    integer(c_int32_t), intent(in) :: act
    type(array_type), target :: w, p
    type(array_type), pointer :: c
    real(real32), allocatable :: up(:,:), dz(:,:), dw(:,:), dp(:,:), z(:,:)
    integer(c_int) :: rc

    allocate(w%val(fo * fi, 1), p%val(fi, n), up(fo, n), dz(fo, n), dw(fo * fi, 1), dp(fi, n), z(fo, n))
    w%shape = [fo, fi]
    call fill(w%val, 31); call fill(p%val, 32); call fill(up, 33)
    w%requires_grad = .true.; p%requires_grad = .true.
    w%is_temporary = .false.; p%is_temporary = .false.
    c => matmul_hip(w, p, act)
    z = matmul(reshape(w%val(:, 1), [fo, fi]), p%val)
    if(act .eq. ATHENA_MP_ACT_RELU) z = max(z, 0._real32)
    call close_to(c%val, z, "matmul_hip: value against the intrinsic matmul")
    call c%grad_reverse_from(up)
    rc = athena_mp_activation_bwd_host(act, int(fo, c_int64_t) * n, c%val, up, dz)
    if(rc .eq. 0) rc = athena_mp_gemm_dw_host(int(n, c_int64_t), int(fi, c_int32_t), int(fo, c_int32_t), p%val, dz, dw)
    if(rc .eq. 0) rc = athena_mp_gemm_dx_host(int(n, c_int64_t), int(fi, c_int32_t), int(fo, c_int32_t), dz, w%val, dp)
    if(rc .ne. 0) call fail("direct dense entry points")
    call close_to(w%grad%val, dw, "matmul_hip: dW through the tape")
    call close_to(p%grad%val, dp, "matmul_hip: dP through the tape")
  end subroutine matmul_case

  subroutine fused_duvenaud_case(n, fv, fe, fo, no)
    !! what hip_duvenaud_msgpass_layer_type's fused branch builds per time step: propagate -> ONE launch for update + sigmoid +
    !! the readout's softmax(matmul(R, z)); z and p are both nodes; the gradient reaches z twice (directly, and through p)
    integer, intent(in) :: n, fv, fe, fo, no
    integer, allocatable :: ia(:), ja(:,:)
    integer :: ne, mn, mx, fi
    type(c_ptr) :: handle
    type(array_type), target :: x, e, w, r, p
    type(array_type), pointer :: a, z, loss
    real(real32), allocatable :: gz(:,:), gp(:,:), dl(:,:), dzp(:,:), dc1(:,:), dc2(:,:), da1(:,:), da2(:,:), dw1(:,:), dw2(:,:), dr(:,:)
    integer(c_int) :: rc
    integer(c_int32_t) :: sig

    mn = 1; mx = 4; fi = fv + fe; sig = ATHENA_MP_ACT_SIGMOID
    call chain_graph(n, ia, ja, ne)
    if(athena_mp_graph_acquire(int(n, c_int32_t), int(size(ja, 2), c_int64_t), ia, ja, int(ne, c_int32_t), handle) .ne. 0) call fail("graph_acquire")
    allocate(x%val(fv, n), e%val(fe, ne), w%val(fo * fi * (mx - mn + 1), 1), r%val(no * fo, 1), gz(fo, n), gp(no, n))
    call fill(x%val, 21); call fill(e%val, 22); call fill(w%val, 23); call fill(r%val, 24); call fill(gz, 25); call fill(gp, 26)
    x%requires_grad = .true.; e%requires_grad = .true.; w%requires_grad = .true.; r%requires_grad = .true.
    x%is_temporary = .false.; e%is_temporary = .false.; w%is_temporary = .false.; r%is_temporary = .false.
    a => duvenaud_propagate_hip(x, e, handle)
    z => duvenaud_update_act_readout_hip(a, w, r, p, handle, mn, mx, fo, sig)
    ! p = softmax(matmul(R, z)) holds its value already: wire it up as the layer type does
    p%get_partial_left_val => get_partial_readout_softmax_hip_z_val
    p%get_partial_right_val => get_partial_readout_softmax_hip_weight_val
    p%left_operand => z
    p%right_operand => r
    p%requires_grad = .true.
    ! the gradient arriving from the next time step (gz, on z) and the readout's (gp, on p) as ONE scalar loss: z collects both
    ! before it is asked for its partials -- one fused device pass for the node, not one per arriving gradient
    loss => weighted_sum(z, gz) + weighted_sum(p, gp)
    call loss%grad_reverse()
    ! expected, from the op-granular entry points
    allocate(dl(no, n), dzp(fo, n), dc1(fo, n), dc2(fo, n), da1(fi, n), da2(fi, n), dw1(size(w%val, 1), 1), dw2(size(w%val, 1), 1), dr(no * fo, 1))
    rc = athena_mp_softmax_bwd_host(int(n, c_int64_t), int(no, c_int32_t), p%val, gp, dl)
    if(rc .eq. 0) rc = athena_mp_gemm_dx_host(int(n, c_int64_t), int(fo, c_int32_t), int(no, c_int32_t), dl, r%val, dzp)
    if(rc .eq. 0) rc = athena_mp_gemm_dw_host(int(n, c_int64_t), int(fo, c_int32_t), int(no, c_int32_t), z%val, dl, dr)
    if(rc .eq. 0) rc = athena_mp_activation_bwd_host(sig, int(fo, c_int64_t) * n, z%val, gz, dc1)
    if(rc .eq. 0) rc = athena_mp_activation_bwd_host(sig, int(fo, c_int64_t) * n, z%val, dzp, dc2)
    if(rc .eq. 0) rc = athena_mp_duvenaud_update_bwd_a_host(handle, int(fi, c_int32_t), int(fo, c_int32_t), int(mn, c_int32_t), int(mx, c_int32_t), dc1, w%val, da1)
    if(rc .eq. 0) rc = athena_mp_duvenaud_update_bwd_a_host(handle, int(fi, c_int32_t), int(fo, c_int32_t), int(mn, c_int32_t), int(mx, c_int32_t), dc2, w%val, da2)
    if(rc .eq. 0) rc = athena_mp_duvenaud_update_bwd_w_host(handle, int(fi, c_int32_t), int(fo, c_int32_t), int(mn, c_int32_t), int(mx, c_int32_t), dc1, a%val, dw1)
    if(rc .eq. 0) rc = athena_mp_duvenaud_update_bwd_w_host(handle, int(fi, c_int32_t), int(fo, c_int32_t), int(mn, c_int32_t), int(mx, c_int32_t), dc2, a%val, dw2)
    if(rc .ne. 0) call fail("direct entry points (fused duvenaud case)")
    call close_to(r%grad%val, dr, "fused duvenaud: dR through the tape")
    call close_to(a%grad%val, da1 + da2, "fused duvenaud: da through the tape")
    call close_to(w%grad%val, dw1 + dw2, "fused duvenaud: dW through the tape")
    if(z%partial_calls .ne. 2 .or. p%partial_calls .ne. 2) call fail("fused duvenaud: callbacks per node")
    if(athena_mp_graph_release(handle) .ne. 0) call fail("graph_release")
  end subroutine fused_duvenaud_case

  subroutine gno_case(n, d, h, fi, fo)
    integer, intent(in) :: n, d, h, fi, fo
    integer, allocatable :: ia(:), ja(:,:)
    integer :: ne, w0
    type(c_ptr) :: handle
    type(array_type), target :: x, coords, theta
    type(array_type), pointer :: k, m
    real(real32), allocatable :: up(:,:), dx(:,:), dth(:,:)
    integer(c_int) :: rc

    call chain_graph(n, ia, ja, ne)
    ! no self loops for this layer: drop each row's first entry (edge id 0)
    do w0 = 1, n
       ia(w0) = ia(w0) - (w0 - 1)
    end do
    ia(n + 1) = ia(n + 1) - n
    ja = ja(:, pack([(w0, w0 = 1, size(ja, 2))], ja(2, :) .gt. 0))
    if(athena_mp_graph_acquire(int(n, c_int32_t), int(size(ja, 2), c_int64_t), ia, ja, int(ne, c_int32_t), handle) .ne. 0) call fail("graph_acquire")
    allocate(x%val(fi, n), coords%val(d, ne), theta%val(h * d + h + fo * fi * h + fo * fi, 1), up(fo, n))
    call fill(x%val, 5); call fill(coords%val, 6); call fill(theta%val, 7); call fill(up, 8)
    x%requires_grad = .true.; theta%requires_grad = .true.
    x%is_temporary = .false.; coords%is_temporary = .false.; theta%is_temporary = .false.
    k => gno_kernel_hip(coords, theta)
    m => gno_aggregate_hip(x, k, handle, d, h, fi, fo)
    call m%grad_reverse_from(up)
    if(m%partial_calls .ne. 2) call fail("grad_reverse did not ask the aggregate node for both partials")
    allocate(dx(fi, n), dth(size(theta%val, 1), 1))
    rc = athena_mp_gno_aggregate_bwd_x_host(handle, int(d, c_int32_t), int(h, c_int32_t), int(fi, c_int32_t), int(fo, c_int32_t), theta%val, &
         coords%val, up, dx)
    if(rc .eq. 0) rc = athena_mp_gno_aggregate_bwd_theta_host(handle, int(d, c_int32_t), int(h, c_int32_t), int(fi, c_int32_t), int(fo, c_int32_t), &
         theta%val, coords%val, x%val, up, dth)
    if(rc .ne. 0) call fail("direct GNO reverse entry points")
    call close_to(x%grad%val, dx, "gno: dx through the tape")
    call close_to(theta%grad%val, dth, "gno: dtheta through the tape")
    if(athena_mp_graph_release(handle) .ne. 0) call fail("graph_release")
  end subroutine gno_case

end program run_ops
