!! Fortran host side of the MI355X message-passing engine: the three layer types with the reference's
!! layer API -- constructor keywords, set_graph, forward, backward, get/set_params, get_gradients --
!! whose arithmetic runs as HIP kernels behind include/athena_mp.h (module athena_mp_c).
!!
!!   kipf_mp_layer_type       athena_kipf_msgpass_layer.f90      (update_message_kipf :915-959)
!!   duvenaud_mp_layer_type   athena_duvenaud_msgpass_layer.f90  (update_message / update_readout :755-859)
!!   graph_nop_mp_layer_type  athena_graph_nop_layer.f90         (update_message_gno :690-788)
!!
!! The reference's types extend base_layer_type and exchange diffstruc `array_type` nodes; diffstruc,
!! graphstruc and coreutils are not vendored with athena, so these types stand on their own: tensors are
!! plain real32 arrays val(features, elements) exactly as `array_type%val` lays them out, graphs carry the
!! components of graphstruc's graph_type the layers read (mp_graph_type), and `backward` walks the tape
!! that diffstruc's grad_reverse would walk (activation -> dense step -> aggregation).  INTEGRATION.md
!! section 3 shows the same calls inside a type that extends the reference's own layer.
!!
!! Parameters, gradients and the tape stay in HBM between calls; host arrays cross the boundary only in
!! forward's input / result, backward's upstream / result and the get/set accessors.  Errors follow the
!! reference's stop_program convention: message on stderr, `error stop`.
module athena_mp_layers
  use, intrinsic :: iso_c_binding
  use athena_mp_c
  implicit none
  private

  integer, parameter, public :: real32 = c_float

  public :: mp_graph_type, mp_actv_type
  public :: kipf_mp_layer_type, duvenaud_mp_layer_type, graph_nop_mp_layer_type

  !! the part of graphstruc's graph_type the message-passing layers read (CSR built by
  !! generate_adjacency [+ add_self_loops]: adj_ia(num_vertices+1), adj_ja(2, nnz) = (neighbour, edge id))
  type :: mp_graph_type
     integer :: num_vertices = 0
     integer :: num_edges = 0
     integer(c_int32_t), allocatable :: adj_ia(:)
     integer(c_int32_t), allocatable :: adj_ja(:,:)
  end type mp_graph_type

  !! base_actv_type with its attributes (athena_activation_*.f90): name, scale and up to two own
  !! attributes -- relu: threshold | leaky_relu: alpha | selu: alpha, lambda | gaussian: sigma, mu |
  !! piecewise: gradient, limit | swish: beta
  type :: mp_actv_type
     character(16) :: name = "none"
     real(real32) :: scale = 1._real32
     real(real32) :: p0 = 0._real32, p1 = 0._real32
  end type mp_actv_type

  interface mp_actv_type
     module procedure actv_setup
  end interface mp_actv_type

  !! a device allocation that only ever grows
  type :: dbuf
     type(c_ptr) :: p = c_null_ptr
     integer(c_int64_t) :: cap = 0_c_int64_t          ! in 4-byte elements
  end type dbuf

  type, abstract :: mp_layer_type
     type(c_ptr) :: graph = c_null_ptr                 ! athena_mp_graph handle of the current batch
     integer :: nv = 0, ne = 0, batch = 0              ! vertices / edge-feature columns / graphs of the batch
     integer, allocatable :: vertex_offset(:)          ! (batch+1), 0-based starts of each graph's vertices
     type(dbuf) :: seg                                 ! the same offsets on the device (int32)
     logical :: keep_edges = .false.
     logical :: inference = .false.                    ! base_layer_type%inference (athena_base_layer.f90:48): no reverse pass follows
     ! what the handle was built from: per graph of the batch its sizes and the content key of its CSR
     ! (athena_mp_graph_key).  set_graph compares these and keeps the handle when nothing changed.
     integer, allocatable :: key_n(:), key_nnz(:), key_ne(:)
     integer(c_int64_t), allocatable :: key_hash(:)
     integer :: graph_builds = 0                       ! times set_graph had to (re)build -- the cache tests read it
     integer :: num_tensors = 0
     integer, allocatable :: psize(:)                  ! size of params(i)%val(:,1)
     type(dbuf), allocatable :: params(:), grads(:)
     type(dbuf), allocatable :: adam_m(:), adam_v(:)   ! created by minimise_adam
     logical, allocatable :: has_grad(:)
   contains
     procedure, pass(this) :: set_graph => layer_set_graph
     procedure, pass(this) :: set_graph_from_edges => layer_set_graph_from_edges
     procedure, pass(this) :: invalidate_graph => layer_invalidate_graph
     procedure, pass(this) :: get_num_params => layer_get_num_params
     procedure, pass(this) :: get_params => layer_get_params
     procedure, pass(this) :: set_params => layer_set_params
     procedure, pass(this) :: get_gradients => layer_get_gradients
     procedure, pass(this) :: alloc_params => layer_alloc_params
     procedure, pass(this) :: release_base => layer_release_base
     procedure, pass(this) :: minimise_base => layer_minimise_base
     procedure, pass(this) :: minimise_adam => layer_minimise_adam
  end type mp_layer_type

  type, extends(mp_layer_type) :: kipf_mp_layer_type
     integer :: num_time_steps = 0
     integer, allocatable :: num_vertex_features(:)    ! (0:T)
     type(mp_actv_type) :: activation
     character(16) :: order = "auto"                   ! auto | aggregate_first | transform_first
     type(dbuf), allocatable :: tape_p(:), tape_y(:), tape_z(:)
     type(dbuf) :: x_in, up_in, dual, scratch(3)
   contains
     procedure, pass(this) :: forward => kipf_forward
     procedure, pass(this) :: backward => kipf_backward
     procedure, pass(this) :: forward_dev => kipf_forward_dev
     procedure, pass(this) :: backward_dev => kipf_backward_dev
     procedure, pass(this) :: destroy => kipf_destroy
     procedure, pass(this), private :: transform_first => kipf_transform_first
  end type kipf_mp_layer_type

  interface kipf_mp_layer_type
     module procedure kipf_setup
  end interface kipf_mp_layer_type

  type, extends(mp_layer_type) :: duvenaud_mp_layer_type
     integer :: num_time_steps = 0, num_outputs = 0
     integer :: min_vertex_degree = 1, max_vertex_degree = 0
     integer, allocatable :: num_vertex_features(:), num_edge_features(:)   ! (0:T)
     type(mp_actv_type) :: activation, activation_readout
     type(dbuf), allocatable :: tape_a(:), tape_z(:), tape_c(:), tape_p(:), tape_l(:)
     type(dbuf) :: x_in, e_in, out_dev, gout, de_acc, dae_sum, a_e, scratch(5)
     logical :: split_a = .false.    !! the tape holds a split: tape_a(t) = a_x (F_v, n), a_e (F_e, n) gathered once per forward
   contains
     procedure, pass(this) :: forward => duvenaud_forward
     procedure, pass(this) :: backward => duvenaud_backward
     procedure, pass(this) :: forward_dev => duvenaud_forward_dev
     procedure, pass(this) :: backward_dev => duvenaud_backward_dev
     procedure, pass(this) :: destroy => duvenaud_destroy
  end type duvenaud_mp_layer_type

  interface duvenaud_mp_layer_type
     module procedure duvenaud_setup
  end interface duvenaud_mp_layer_type

  type, extends(mp_layer_type) :: graph_nop_mp_layer_type
     integer :: num_inputs = 0, num_outputs = 0, coord_dim = 0, kernel_hidden = 16
     logical :: use_bias = .true.
     type(mp_actv_type) :: activation
     type(dbuf) :: x_in, c_in, up_in, z_pre, y_out, ones, dc_out, scratch(3)
     type(c_ptr) :: x_tape = c_null_ptr, c_tape = c_null_ptr     ! the forward inputs (caller-owned in forward_dev)
     !! keep_s: the forward pass keeps S = sum_e [h_e;1] x_j^T per vertex for the reverse pass's dVaug = S^T g (the
     !! reference keeps every forward intermediate on its tape) whenever the shape takes the kernels that do and S is at
     !! most ATHENA_MP_GNO_KEEP_S_MAX_GB (default 64; BASELINE configs[3]: 33 of the 288 GB).  Same results bit for bit.
     logical :: keep_s = .true.
     logical :: s_valid = .false.
     type(dbuf) :: s_save
     type(c_ptr) :: s_graph = c_null_ptr                         ! the handle s_save was built on
   contains
     procedure, pass(this) :: forward => gno_forward
     procedure, pass(this) :: backward => gno_backward
     procedure, pass(this) :: forward_dev => gno_forward_dev
     procedure, pass(this) :: backward_dev => gno_backward_dev
     procedure, pass(this) :: destroy => gno_destroy
  end type graph_nop_mp_layer_type

  interface graph_nop_mp_layer_type
     module procedure gno_setup
  end interface graph_nop_mp_layer_type

contains

!###############################################################################
! helpers
!###############################################################################
  subroutine stop_program(msg)
    !! coreutils' stop_program: print and error-stop
    character(*), intent(in) :: msg
    write(0,'(A)') "ERROR: "//trim(msg)
    error stop 1
  end subroutine stop_program

  subroutine chk(rc, what)
    integer(c_int), intent(in) :: rc
    character(*), intent(in) :: what
    if(rc .ne. 0_c_int) call stop_program(what//": "//athena_mp_error_message())
  end subroutine chk

  subroutine need(b, n)
    !! make b hold at least n 4-byte elements
    type(dbuf), intent(inout) :: b
    integer(c_int64_t), intent(in) :: n
    if(n .le. b%cap) return
    if(c_associated(b%p)) call chk(athena_mp_free(b%p), "free")
    call chk(athena_mp_malloc(b%p, 4_c_int64_t * max(n, 1_c_int64_t)), "malloc")
    b%cap = max(n, 1_c_int64_t)
  end subroutine need

  subroutine release(b)
    type(dbuf), intent(inout) :: b
    if(c_associated(b%p)) call chk(athena_mp_free(b%p), "free")
    b%p = c_null_ptr
    b%cap = 0_c_int64_t
  end subroutine release

  real function keep_s_max_gb()
    !! ATHENA_MP_GNO_KEEP_S_MAX_GB: the largest S a graph_nop layer keeps between forward and backward (default 64)
    character(32) :: v
    integer :: stat, ios
    keep_s_max_gb = 64.
    call get_environment_variable("ATHENA_MP_GNO_KEEP_S_MAX_GB", v, status=stat)
    if(stat .eq. 0)then
       read(v, *, iostat=ios) keep_s_max_gb
       if(ios .ne. 0) keep_s_max_gb = 64.
    end if
  end function keep_s_max_gb

  pure integer(c_int64_t) function i8(n)
    integer, intent(in) :: n
    i8 = int(n, c_int64_t)
  end function i8

  function actv_setup(name, scale, threshold, alpha, lambda, sigma, mu, gradient, limit, beta) result(a)
    !! the `initialise` functions of athena_activation_*.f90: reset_* defaults, then the optional arguments
    character(*), intent(in) :: name
    real(real32), intent(in), optional :: scale, threshold, alpha, lambda, sigma, mu, gradient, limit, beta
    type(mp_actv_type) :: a

    a%name = name
    a%scale = 1._real32
    a%p0 = 0._real32
    a%p1 = 0._real32
    select case(trim(name))
    case("none", "linear", "sigmoid", "tanh", "softmax", "relu")
    case("leaky_relu")
       a%p0 = 0.01_real32                      ! athena_activation_leaky_relu.f90:85
    case("selu")
       a%p0 = 1.67326_real32                   ! athena_activation_selu.f90:98-99
       a%p1 = 1.0507_real32
    case("gaussian")
       a%p0 = 1.5_real32                       ! athena_activation_gaussian.f90:95-96
    case("piecewise")
       a%p0 = 0.1_real32                       ! athena_activation_piecewise.f90:83-84
       a%p1 = 1._real32
    case("swish", "silu")
       a%name = "swish"
       a%p0 = 1._real32
    case default
       call stop_program("unknown activation '"//trim(name)//"'")
    end select
    if(present(scale)) a%scale = scale
    if(present(threshold)) a%p0 = threshold
    if(present(alpha)) a%p0 = alpha
    if(present(lambda)) a%p1 = lambda
    if(present(sigma)) a%p0 = sigma
    if(present(mu)) a%p1 = mu
    if(present(gradient)) a%p0 = gradient
    if(present(limit)) a%p1 = limit
    if(present(beta)) a%p0 = beta
  end function actv_setup

  pure logical function apply_scaling(a)
    type(mp_actv_type), intent(in) :: a
    apply_scaling = abs(a%scale - 1._real32) .gt. 1.e-6_real32
  end function apply_scaling

  pure logical function is_identity(a)
    type(mp_actv_type), intent(in) :: a
    is_identity = (trim(a%name) .eq. "none" .or. trim(a%name) .eq. "linear") .and. .not. apply_scaling(a)
  end function is_identity

  pure integer(c_int32_t) function fused_code(a)
    !! epilogue code of the plain activations, -1 when the activation needs its own launch
    type(mp_actv_type), intent(in) :: a
    fused_code = -1_c_int32_t
    if(apply_scaling(a)) return
    select case(trim(a%name))
    case("none", "linear")
       fused_code = ATHENA_MP_ACT_NONE
    case("relu")
       if(a%p0 .eq. 0._real32) fused_code = ATHENA_MP_ACT_RELU
    case("sigmoid")
       fused_code = ATHENA_MP_ACT_SIGMOID
    case("tanh")
       fused_code = ATHENA_MP_ACT_TANH
    end select
  end function fused_code

  pure logical function needs_input(a)
    !! does the reverse factor take the pre-activation (else the output)?
    type(mp_actv_type), intent(in) :: a
    needs_input = fused_code(a) .lt. 0 .and. trim(a%name) .ne. "softmax"
  end function needs_input

  pure integer(c_int32_t) function param_kind(a)
    type(mp_actv_type), intent(in) :: a
    select case(trim(a%name))
    case("relu");       param_kind = 1
    case("sigmoid");    param_kind = 2
    case("tanh");       param_kind = 3
    case("leaky_relu"); param_kind = 4
    case("selu");       param_kind = 5
    case("gaussian");   param_kind = 6
    case("piecewise");  param_kind = 7
    case default;       param_kind = 0
    end select
  end function param_kind

  subroutine act_apply(a, n, f, z, y)
    !! y = activation(z) on n elements of f features each
    type(mp_actv_type), intent(in) :: a
    integer, intent(in) :: n, f
    type(c_ptr), intent(in) :: z, y
    real(real32) :: sc
    integer(c_int64_t) :: tot

    tot = i8(n) * i8(f)
    if(fused_code(a) .ge. 0)then
       call chk(athena_mp_activation_fwd(fused_code(a), tot, z, y), "activation")
    else if(trim(a%name) .eq. "softmax")then
       if(apply_scaling(a)) call stop_program("softmax takes no scale on the HIP path")
       call chk(athena_mp_softmax_fwd(i8(n), int(f, c_int32_t), z, y), "softmax")
    else if(trim(a%name) .eq. "swish" .and. .not. apply_scaling(a))then
       call chk(athena_mp_swish_fwd(tot, a%p0, z, y), "swish")
    else if(trim(a%name) .eq. "swish")then
       call stop_program("swish takes no scale on the HIP path")
    else
       sc = merge(a%scale, 1._real32, apply_scaling(a))
       call chk(athena_mp_activation_param_fwd(param_kind(a), tot, sc, a%p0, a%p1, z, y), "activation")
    end if
  end subroutine act_apply

  subroutine act_reverse(a, n, f, y, z, g, dz)
    !! dz = upstream g through the activation; y = its output, z = its input (where needs_input)
    type(mp_actv_type), intent(in) :: a
    integer, intent(in) :: n, f
    type(c_ptr), intent(in) :: y, z, g, dz
    real(real32) :: sc
    integer(c_int64_t) :: tot

    tot = i8(n) * i8(f)
    if(fused_code(a) .ge. 0)then
       call chk(athena_mp_activation_bwd(fused_code(a), tot, y, g, dz), "activation reverse")
    else if(trim(a%name) .eq. "softmax")then
       call chk(athena_mp_softmax_bwd(i8(n), int(f, c_int32_t), y, g, dz), "softmax reverse")
    else if(trim(a%name) .eq. "swish")then
       call chk(athena_mp_swish_bwd(tot, a%p0, z, g, dz), "swish reverse")
    else
       sc = merge(a%scale, 1._real32, apply_scaling(a))
       call chk(athena_mp_activation_param_bwd(param_kind(a), tot, sc, a%p0, a%p1, z, g, dz), &
            "activation reverse")
    end if
  end subroutine act_reverse

  subroutine expand_features(nf, t, what, full)
    !! scalar or (T+1)-vector -> (0:T)   (athena_kipf_msgpass_layer.f90:262-274)
    integer, intent(in) :: nf(:)
    integer, intent(in) :: t
    character(*), intent(in) :: what
    integer, allocatable, intent(out) :: full(:)

    allocate(full(0:t))
    if(size(nf) .eq. 1)then
       full = nf(1)
    else if(size(nf) .eq. t + 1)then
       full(0:t) = nf(:)
    else
       call stop_program(what//" must be a scalar or a vector of length num_time_steps + 1")
    end if
  end subroutine expand_features

  subroutine init_weights(w, fan_in, fan_out, relu_family)
    !! default initialiser: he_normal for the relu family, glorot_uniform otherwise
    !! (athena_initialiser_he.f90:229,245; athena_initialiser_glorot.f90:120)
    real(real32), intent(out) :: w(:)
    integer, intent(in) :: fan_in, fan_out
    logical, intent(in) :: relu_family
    real(real32), parameter :: pi = 3.14159265358979_real32
    real(real32) :: u1, u2, lim
    integer :: i

    if(relu_family)then
       do i = 1, size(w)
          call random_number(u1)
          call random_number(u2)
          u1 = max(u1, tiny(1._real32))
          w(i) = sqrt(2._real32 / real(fan_in, real32)) * sqrt(-2._real32 * log(u1)) * cos(2._real32 * pi * u2)
       end do
    else
       lim = sqrt(6._real32 / real(fan_in + fan_out, real32))
       call random_number(w)
       w = (2._real32 * w - 1._real32) * lim
    end if
  end subroutine init_weights

  pure logical function relu_family(a)
    type(mp_actv_type), intent(in) :: a
    select case(trim(a%name))
    case("relu", "leaky_relu", "swish", "selu")
       relu_family = .true.
    case default
       relu_family = .false.
    end select
  end function relu_family

!###############################################################################
! base: graph, learnable accessors (athena_base_layer_sub.f90:545-691)
!###############################################################################
  subroutine layer_set_graph(this, graph)
    !! athena_msgpass_layer_sub.f90:144-174.  The batch becomes ONE block-diagonal device graph: vertex
    !! ids shifted by the vertices before, edge ids (where > 0) by the edge columns before.
    !! The reference calls this before EVERY forward (athena_network_sub.f90:2727-2730) and copies the CSR each
    !! time; here it costs one content key per graph (athena_mp_graph_key: every word of adj_ia / adj_ja, hashed by a
    !! few host threads) -- the handle is rebuilt only when a size or a key differs from what it was built from, and a
    !! rebuilt batch goes through athena_mp_graph_acquire, so layers that are given the same batch share one handle.
    !! An in-place edit is therefore always seen, as in the reference.  Only under ATHENA_MP_GRAPH_KEY_SAMPLED=1 (the
    !! cheap sampled key for graphs of 2^18 entries and more) does an in-place edit need invalidate_graph().
    class(mp_layer_type), intent(inout) :: this
    type(mp_graph_type), intent(in) :: graph(:)
    integer(c_int32_t), allocatable :: ia(:), ja(:,:)
    integer, allocatable :: kn(:), knnz(:), kne(:)
    integer(c_int64_t), allocatable :: kh(:)
    integer :: s, nnz, v0, e0, w0, n, m, b
    logical :: same

    b = size(graph)
    allocate(kn(b), knnz(b), kne(b), kh(b))
    do s = 1, b
       if(.not. allocated(graph(s)%adj_ia) .or. .not. allocated(graph(s)%adj_ja)) &
            call stop_program("set_graph: graph has no adjacency (call generate_adjacency first)")
       kn(s) = graph(s)%num_vertices
       knnz(s) = size(graph(s)%adj_ja, 2)
       kne(s) = graph(s)%num_edges
       if(size(graph(s)%adj_ia) .ne. kn(s) + 1) call stop_program("set_graph: adj_ia must hold num_vertices + 1 row pointers")
       call chk(athena_mp_graph_key(int(kn(s), c_int32_t), int(knnz(s), c_int64_t), graph(s)%adj_ia, graph(s)%adj_ja, &
            kh(s)), "graph_key")
    end do
    same = c_associated(this%graph) .and. allocated(this%key_hash)
    if(same) same = size(this%key_hash) .eq. b
    if(same) same = all(this%key_n .eq. kn) .and. all(this%key_nnz .eq. knnz) .and. all(this%key_ne .eq. kne) &
         .and. all(this%key_hash .eq. kh)
    if(same) return                                    ! the handle, offsets and device segments are still valid

    call layer_drop_graph(this)
    this%batch = b
    if(allocated(this%vertex_offset)) deallocate(this%vertex_offset)
    allocate(this%vertex_offset(b + 1))
    this%vertex_offset(1) = 0
    do s = 1, b
       this%vertex_offset(s + 1) = this%vertex_offset(s) + kn(s)
    end do
    nnz = sum(knnz)
    this%ne = sum(kne)
    this%nv = this%vertex_offset(b + 1)
    if(b .eq. 1 .and. this%keep_edges)then
       ! one graph with its edge ids: the caller's arrays are the block-diagonal batch
       call chk(athena_mp_graph_acquire(int(this%nv, c_int32_t), int(nnz, c_int64_t), graph(1)%adj_ia, graph(1)%adj_ja, &
            int(this%ne, c_int32_t), this%graph), "graph_acquire")
    else
       allocate(ia(this%nv + 1), ja(2, max(nnz, 1)))
       ia(1) = 1
       v0 = 0
       e0 = 0
       w0 = 0
       do s = 1, b                                     ! whole-array shifts (vectorised), no per-entry branches
          n = kn(s)
          m = knnz(s)
          ia(v0 + 2:v0 + n + 1) = graph(s)%adj_ia(2:n + 1) + w0
          ja(1, w0 + 1:w0 + m) = graph(s)%adj_ja(1, 1:m) + v0
          if(this%keep_edges)then
             ja(2, w0 + 1:w0 + m) = merge(graph(s)%adj_ja(2, 1:m) + e0, 0_c_int32_t, graph(s)%adj_ja(2, 1:m) .gt. 0)
          else
             ja(2, w0 + 1:w0 + m) = 0_c_int32_t        ! a Kipf handle carries no edge ids
          end if
          v0 = v0 + n
          e0 = e0 + kne(s)
          w0 = w0 + m
       end do
       call chk(athena_mp_graph_acquire(int(this%nv, c_int32_t), int(nnz, c_int64_t), ia, ja, &
            int(merge(this%ne, 0, this%keep_edges), c_int32_t), this%graph), "graph_acquire")
    end if
    this%graph_builds = this%graph_builds + 1
    call move_alloc(kn, this%key_n)
    call move_alloc(knnz, this%key_nnz)
    call move_alloc(kne, this%key_ne)
    call move_alloc(kh, this%key_hash)
    call need(this%seg, i8(this%batch + 1))
    call chk(athena_mp_memcpy_h2d(this%seg%p, this%vertex_offset, 4_c_int64_t * i8(this%batch + 1)), "h2d")
  end subroutine layer_set_graph

  subroutine layer_drop_graph(this)
    !! one user less of the handle (it stays in the library's cache for the next epoch's batch of the same content)
    class(mp_layer_type), intent(inout) :: this
    if(c_associated(this%graph)) call chk(athena_mp_graph_destroy(this%graph), "graph_destroy")
    this%graph = c_null_ptr
    if(allocated(this%key_hash)) deallocate(this%key_n, this%key_nnz, this%key_ne, this%key_hash)
  end subroutine layer_drop_graph

  subroutine layer_invalidate_graph(this)
    !! the caller edited the adjacency in place and says so: the handle is dropped AND leaves the library's cache
    !! (athena_mp_graph_evict), so the next set_graph builds from the arrays whatever their key is
    class(mp_layer_type), intent(inout) :: this
    if(c_associated(this%graph)) call chk(athena_mp_graph_evict(this%graph), "graph_evict")
    this%graph = c_null_ptr
    if(allocated(this%key_hash)) deallocate(this%key_n, this%key_nnz, this%key_ne, this%key_hash)
  end subroutine layer_invalidate_graph

  subroutine layer_set_graph_from_edges(this, num_vertices, index_list, add_self_loops)
    !! one graph given as its edge list (what generate_adjacency takes): CSR and device handle are built on the GPU in
    !! one call, the entries never visit the host (athena_mp_graph_create_from_edges); edge id = column of index_list
    class(mp_layer_type), intent(inout) :: this
    integer, intent(in) :: num_vertices
    integer(c_int32_t), intent(in) :: index_list(:,:)                  ! (2, num_edges)
    logical, intent(in), optional :: add_self_loops
    integer(c_int32_t), allocatable :: ia(:)
    integer(c_int64_t) :: nnz
    integer(c_int32_t) :: loops

    if(size(index_list, 1) .ne. 2) call stop_program("set_graph_from_edges: index_list must be (2, num_edges)")
    loops = 0
    if(present(add_self_loops)) loops = merge(1_c_int32_t, 0_c_int32_t, add_self_loops)
    call layer_drop_graph(this)
    this%graph_builds = this%graph_builds + 1
    this%batch = 1
    this%nv = num_vertices
    this%ne = size(index_list, 2)
    if(allocated(this%vertex_offset)) deallocate(this%vertex_offset)
    allocate(this%vertex_offset(2), ia(num_vertices + 1))
    this%vertex_offset = [0, num_vertices]
    call chk(athena_mp_graph_create_from_edges(int(num_vertices, c_int32_t), int(size(index_list, 2), c_int64_t), index_list, &
         loops, merge(1_c_int32_t, 0_c_int32_t, this%keep_edges), ia, c_null_ptr, 0_c_int64_t, nnz, this%graph), &
         "graph_create_from_edges")
    call need(this%seg, 2_c_int64_t)
    call chk(athena_mp_memcpy_h2d(this%seg%p, this%vertex_offset, 8_c_int64_t), "h2d")
  end subroutine layer_set_graph_from_edges

  subroutine layer_alloc_params(this, sizes)
    class(mp_layer_type), intent(inout) :: this
    integer, intent(in) :: sizes(:)
    integer :: i

    this%num_tensors = size(sizes)
    allocate(this%psize(size(sizes)), this%params(size(sizes)), this%grads(size(sizes)), this%has_grad(size(sizes)))
    this%psize = sizes
    this%has_grad = .false.
    do i = 1, size(sizes)
       call need(this%params(i), i8(sizes(i)))
       call need(this%grads(i), i8(sizes(i)))
    end do
  end subroutine layer_alloc_params

  pure integer function layer_get_num_params(this) result(n)
    class(mp_layer_type), intent(in) :: this
    n = 0
    if(allocated(this%psize)) n = sum(this%psize)
  end function layer_get_num_params

  function layer_get_params(this) result(params)
    !! flat concatenation of params(i)%val(:,1) in index order
    class(mp_layer_type), intent(in) :: this
    real(real32), allocatable :: params(:)
    integer :: i, o

    allocate(params(this%get_num_params()))
    o = 0
    do i = 1, this%num_tensors
       call chk(athena_mp_memcpy_d2h(params(o + 1:o + this%psize(i)), this%params(i)%p, 4_c_int64_t * i8(this%psize(i))), "d2h")
       o = o + this%psize(i)
    end do
  end function layer_get_params

  subroutine layer_set_params(this, params)
    class(mp_layer_type), intent(inout) :: this
    real(real32), intent(in) :: params(:)
    integer :: i, o

    if(size(params) .ne. this%get_num_params()) call stop_program("set_params: wrong number of parameters")
    o = 0
    do i = 1, this%num_tensors
       call chk(athena_mp_memcpy_h2d(this%params(i)%p, params(o + 1:o + this%psize(i)), 4_c_int64_t * i8(this%psize(i))), "h2d")
       o = o + this%psize(i)
    end do
  end subroutine layer_set_params

  function layer_get_gradients(this) result(gradients)
    !! zeros where no gradient has been produced yet (athena_base_layer_sub.f90:627-631)
    class(mp_layer_type), intent(in) :: this
    real(real32), allocatable :: gradients(:)
    integer :: i, o

    allocate(gradients(this%get_num_params()))
    gradients = 0._real32
    o = 0
    do i = 1, this%num_tensors
       if(this%has_grad(i)) call chk(athena_mp_memcpy_d2h(gradients(o + 1:o + this%psize(i)), this%grads(i)%p, &
            4_c_int64_t * i8(this%psize(i))), "d2h")
       o = o + this%psize(i)
    end do
  end function layer_get_gradients

  subroutine layer_minimise_base(this, learning_rate)
    !! base_optimiser_type%minimise (athena_optimiser.f90:396-418) on the resident tensors: param = param - lr * gradient,
    !! then the gradients count as reset (network%update, athena_network_sub.f90:2816-2929)
    class(mp_layer_type), intent(inout) :: this
    real(real32), intent(in) :: learning_rate
    integer :: i
    do i = 1, this%num_tensors
       if(.not. this%has_grad(i)) call stop_program("Gradient not allocated for parameters")       ! :2857-2861
       call chk(athena_mp_axpy(i8(this%psize(i)), -learning_rate, this%grads(i)%p, this%params(i)%p), "axpy")
       this%has_grad(i) = .false.
    end do
  end subroutine layer_minimise_base

  subroutine layer_minimise_adam(this, learning_rate, iter, beta1, beta2, epsilon)
    !! adam_optimiser_type%minimise (athena_optimiser.f90:1027-1091, no regulariser) per parameter tensor; the moment
    !! vectors live beside the parameters and are created (zero) on first use; iter = optimiser%iter after its increment
    class(mp_layer_type), intent(inout) :: this
    real(real32), intent(in) :: learning_rate
    integer, intent(in) :: iter
    real(real32), intent(in), optional :: beta1, beta2, epsilon
    real(real32) :: b1, b2, eps
    integer :: i

    b1 = 0.9_real32; b2 = 0.999_real32; eps = 1.e-8_real32
    if(present(beta1)) b1 = beta1
    if(present(beta2)) b2 = beta2
    if(present(epsilon)) eps = epsilon
    if(.not. allocated(this%adam_m))then
       allocate(this%adam_m(this%num_tensors), this%adam_v(this%num_tensors))
       do i = 1, this%num_tensors
          call need(this%adam_m(i), i8(this%psize(i)))
          call need(this%adam_v(i), i8(this%psize(i)))
          call chk(athena_mp_memset_zero(this%adam_m(i)%p, 4_c_int64_t * i8(this%psize(i))), "memset")
          call chk(athena_mp_memset_zero(this%adam_v(i)%p, 4_c_int64_t * i8(this%psize(i))), "memset")
       end do
    end if
    do i = 1, this%num_tensors
       if(.not. this%has_grad(i)) call stop_program("Gradient not allocated for parameters")
       call chk(athena_mp_adam_step(i8(this%psize(i)), learning_rate, b1, b2, eps, int(iter, c_int32_t), 0_c_int32_t, &
            0._real32, 0._real32, 0_c_int32_t, this%params(i)%p, this%grads(i)%p, this%adam_m(i)%p, this%adam_v(i)%p), &
            "adam_step")
       this%has_grad(i) = .false.
    end do
  end subroutine layer_minimise_adam

  subroutine layer_release_base(this)
    class(mp_layer_type), intent(inout) :: this
    integer :: i
    call layer_drop_graph(this)
    call release(this%seg)
    do i = 1, this%num_tensors
       call release(this%params(i))
       call release(this%grads(i))
       if(allocated(this%adam_m))then
          call release(this%adam_m(i))
          call release(this%adam_v(i))
       end if
    end do
  end subroutine layer_release_base

!###############################################################################
! Kipf & Welling graph convolution
!###############################################################################
  function kipf_setup(num_vertex_features, num_time_steps, activation, order) result(layer)
    !! layer_setup of athena_kipf_msgpass_layer.f90:214-300; init_kipf :345-380 -- params(t) = W(F_t, F_{t-1})
    integer, intent(in) :: num_vertex_features(:)
    integer, intent(in) :: num_time_steps
    class(*), intent(in), optional :: activation            ! a name or an mp_actv_type, as in the reference
    character(*), intent(in), optional :: order
    type(kipf_mp_layer_type) :: layer
    real(real32), allocatable :: w(:)
    integer, allocatable :: sizes(:)
    integer :: t

    layer%num_time_steps = num_time_steps
    call expand_features(num_vertex_features, num_time_steps, "num_vertex_features", layer%num_vertex_features)
    layer%activation = resolve(activation, "none")
    if(present(order)) layer%order = order
    layer%keep_edges = .false.
    allocate(sizes(num_time_steps))
    do t = 1, num_time_steps
       sizes(t) = layer%num_vertex_features(t) * layer%num_vertex_features(t - 1)
    end do
    call layer%alloc_params(sizes)
    do t = 1, num_time_steps
       allocate(w(sizes(t)))
       call init_weights(w, layer%num_vertex_features(t - 1), layer%num_vertex_features(t), relu_family(layer%activation))
       call chk(athena_mp_memcpy_h2d(layer%params(t)%p, w, 4_c_int64_t * i8(sizes(t))), "h2d")
       deallocate(w)
    end do
    allocate(layer%tape_p(num_time_steps), layer%tape_y(num_time_steps), layer%tape_z(num_time_steps))
  end function kipf_setup

  function resolve(activation, default) result(a)
    class(*), intent(in), optional :: activation
    character(*), intent(in) :: default
    type(mp_actv_type) :: a

    if(.not. present(activation))then
       a = mp_actv_type(default)
       return
    end if
    select type(activation)
    type is(character(*))
       a = mp_actv_type(trim(activation))
    type is(mp_actv_type)
       a = activation
    class default
       call stop_program("activation must be a name or an mp_actv_type")
    end select
  end function resolve

  logical function kipf_transform_first(this, t)
    !! dense step before the aggregation when the step narrows the features (DESIGN.md 3.1c)
    class(kipf_mp_layer_type), intent(in) :: this
    integer, intent(in) :: t
    select case(trim(this%order))
    case("transform_first")
       kipf_transform_first = .true.
    case("aggregate_first")
       kipf_transform_first = .false.
    case default
       kipf_transform_first = 4 * this%num_vertex_features(t) .le. 3 * this%num_vertex_features(t - 1)
    end select
  end function kipf_transform_first

  function kipf_forward(this, vertex_features) result(output)
    !! update_message_kipf :915-959 on host arrays: upload, forward_dev, download
    class(kipf_mp_layer_type), intent(inout) :: this
    real(real32), intent(in) :: vertex_features(:,:)                     ! (F_0, num_vertices of the batch)
    real(real32), allocatable :: output(:,:)
    type(c_ptr) :: y
    integer :: n

    n = this%nv
    if(size(vertex_features, 1) .ne. this%num_vertex_features(0) .or. size(vertex_features, 2) .ne. n) &
         call stop_program("kipf forward: vertex feature shape mismatch")
    call need(this%x_in, i8(n) * i8(this%num_vertex_features(0)))
    call chk(athena_mp_memcpy_h2d(this%x_in%p, vertex_features, 4_c_int64_t * i8(n) * i8(this%num_vertex_features(0))), "h2d")
    y = this%forward_dev(this%x_in%p)
    allocate(output(this%num_vertex_features(this%num_time_steps), n))
    call chk(athena_mp_memcpy_d2h(output, y, 4_c_int64_t * i8(n) * i8(this%num_vertex_features(this%num_time_steps))), "d2h")
  end function kipf_forward

  function kipf_forward_dev(this, x_dev) result(y_dev)
    !! X_t = act( W_t . kipf_propagate(X_{t-1}) ) for all samples of the batch at once, on tensors that stay in HBM:
    !! x_dev = (F_0, vertices) on the device, the result points into the layer's tape (valid until the next forward).
    !! This is how consecutive HIP layers are chained from Fortran without a host round trip.
    class(kipf_mp_layer_type), intent(inout) :: this
    type(c_ptr), intent(in) :: x_dev
    type(c_ptr) :: y_dev
    type(c_ptr) :: cur
    integer :: t, fi, fo, n
    integer(c_int32_t) :: code

    if(.not. c_associated(this%graph)) call stop_program("set_graph must be called before forward")
    n = this%nv
    cur = x_dev
    code = fused_code(this%activation)
    do t = 1, this%num_time_steps
       fi = this%num_vertex_features(t - 1)
       fo = this%num_vertex_features(t)
       call need(this%tape_y(t), i8(n) * i8(fo))
       call need(this%tape_p(t), i8(n) * i8(fi))
       if(this%transform_first(t))then
          ! Y = X W^T, then the aggregation on F_t-wide rows; the tape keeps the step's input in tape_p
          call need(this%tape_z(t), i8(n) * i8(fo))
          call need(this%scratch(1), i8(n) * i8(fo))
          call copy_dev(this%tape_p(t)%p, cur, i8(n) * i8(fi))
          call chk(athena_mp_gemm_fwd(i8(n), int(fi, c_int32_t), int(fo, c_int32_t), cur, this%params(t)%p, c_null_ptr, &
               ATHENA_MP_ACT_NONE, this%scratch(1)%p), "gemm_fwd")
          if(code .ge. 0)then
             ! plain activation: applied in the aggregation's store
             call chk(athena_mp_kipf_propagate_act_fwd(this%graph, int(fo, c_int32_t), this%scratch(1)%p, code, &
                  this%tape_y(t)%p), "propagate")
          else
             call chk(athena_mp_kipf_propagate_fwd(this%graph, int(fo, c_int32_t), this%scratch(1)%p, this%tape_z(t)%p), "propagate")
             call act_apply(this%activation, n, fo, this%tape_z(t)%p, this%tape_y(t)%p)
          end if
       else if(code .ge. 0)then
          ! aggregation + dense step + activation in one launch where the fused kernel exists
          call chk(athena_mp_kipf_layer_fwd(this%graph, int(fi, c_int32_t), int(fo, c_int32_t), cur, this%params(t)%p, &
               c_null_ptr, code, this%tape_p(t)%p, this%tape_y(t)%p), "kipf_layer_fwd")
       else
          call need(this%tape_z(t), i8(n) * i8(fo))
          call chk(athena_mp_kipf_layer_fwd(this%graph, int(fi, c_int32_t), int(fo, c_int32_t), cur, this%params(t)%p, &
               c_null_ptr, ATHENA_MP_ACT_NONE, this%tape_p(t)%p, this%tape_z(t)%p), "kipf_layer_fwd")
          call act_apply(this%activation, n, fo, this%tape_z(t)%p, this%tape_y(t)%p)
       end if
       cur = this%tape_y(t)%p
    end do
    y_dev = cur
  end function kipf_forward_dev

  subroutine copy_dev(dst, src, n)
    !! device-to-device copy of n floats: dst = 0; dst += 1 * src
    type(c_ptr), intent(in) :: dst, src
    integer(c_int64_t), intent(in) :: n
    call chk(athena_mp_memset_zero(dst, 4_c_int64_t * n), "memset")
    call chk(athena_mp_axpy(n, 1._real32, src, dst), "axpy")
  end subroutine copy_dev

  function kipf_backward(this, upstream, exact, need_input_grad) result(dx)
    !! reverse pass on host arrays: upload, backward_dev, download
    class(kipf_mp_layer_type), intent(inout) :: this
    real(real32), intent(in) :: upstream(:,:)                            ! (F_T, num_vertices)
    logical, intent(in), optional :: exact, need_input_grad
    real(real32), allocatable :: dx(:,:)
    type(c_ptr) :: g
    integer :: n, fo
    logical :: want_dx

    n = this%nv
    fo = this%num_vertex_features(this%num_time_steps)
    if(size(upstream, 1) .ne. fo .or. size(upstream, 2) .ne. n) call stop_program("kipf backward: upstream shape mismatch")
    want_dx = .true.
    if(present(need_input_grad)) want_dx = need_input_grad
    call need(this%up_in, i8(n) * i8(fo))
    call chk(athena_mp_memcpy_h2d(this%up_in%p, upstream, 4_c_int64_t * i8(n) * i8(fo)), "h2d")
    g = this%backward_dev(this%up_in%p, exact, need_input_grad)
    if(want_dx)then
       allocate(dx(this%num_vertex_features(0), n))
       call chk(athena_mp_memcpy_d2h(dx, g, 4_c_int64_t * i8(n) * i8(this%num_vertex_features(0))), "d2h")
    else
       allocate(dx(0, 0))
    end if
  end function kipf_backward

  function kipf_backward_dev(this, upstream_dev, exact, need_input_grad) result(dx_dev)
    !! the tape walk of grad_reverse: activation -> matmul (dW, dP) -> get_partial_kipf_propagate_left_val.
    !! exact = .false. (default) reproduces the reference's coefficient-free scatter (:85-111).  upstream_dev =
    !! (F_T, vertices) on the device (read only); the result points into the layer's scratch (valid until the next
    !! backward), c_null_ptr when need_input_grad = .false.  Parameter gradients land in the layer (get_gradients).
    class(kipf_mp_layer_type), intent(inout) :: this
    type(c_ptr), intent(in) :: upstream_dev
    logical, intent(in), optional :: exact, need_input_grad
    type(c_ptr) :: dx_dev
    type(c_ptr) :: gcur, dz, y_prev
    integer :: t, fi, fo, n, ex
    integer(c_int64_t) :: widest
    logical :: want_dx

    n = this%nv
    ex = 0
    if(present(exact)) ex = merge(1, 0, exact)
    want_dx = .true.
    if(present(need_input_grad)) want_dx = need_input_grad
    widest = i8(n) * i8(maxval(this%num_vertex_features))
    call need(this%scratch(1), widest)
    call need(this%scratch(2), widest)
    call need(this%scratch(3), widest)
    gcur = upstream_dev
    dx_dev = c_null_ptr
    do t = this%num_time_steps, 1, -1
       fi = this%num_vertex_features(t - 1)
       fo = this%num_vertex_features(t)
       if(is_identity(this%activation))then
          dz = gcur
       else
          dz = this%scratch(3)%p
          call act_reverse(this%activation, n, fo, this%tape_y(t)%p, this%tape_z(t)%p, gcur, dz)
       end if
       if(this%transform_first(t))then
          ! dW = (A^T dZ)^T X with the coefficient, dX = (scatter of dZ) W without it: one gather, two sums
          y_prev = this%tape_p(t)%p
          if(ex .eq. 1 .or. (t .eq. 1 .and. .not. want_dx))then
             call chk(athena_mp_kipf_propagate_bwd(this%graph, int(fo, c_int32_t), dz, this%scratch(1)%p, 1_c_int32_t), "propagate_bwd")
             call chk(athena_mp_gemm_dw(i8(n), int(fi, c_int32_t), int(fo, c_int32_t), y_prev, this%scratch(1)%p, this%grads(t)%p), "gemm_dw")
             this%has_grad(t) = .true.
             if(t .eq. 1 .and. .not. want_dx) return
          else
             call need(this%dual, i8(n) * i8(fo))
             call chk(athena_mp_kipf_propagate_bwd_dual(this%graph, int(fo, c_int32_t), dz, this%scratch(1)%p, this%dual%p), &
                  "propagate_bwd_dual")
             call chk(athena_mp_gemm_dw(i8(n), int(fi, c_int32_t), int(fo, c_int32_t), y_prev, this%dual%p, this%grads(t)%p), "gemm_dw")
             this%has_grad(t) = .true.
          end if
          call chk(athena_mp_gemm_dx(i8(n), int(fi, c_int32_t), int(fo, c_int32_t), this%scratch(1)%p, this%params(t)%p, &
               this%scratch(2)%p), "gemm_dx")
          gcur = this%scratch(2)%p
          cycle
       end if
       call chk(athena_mp_gemm_dw(i8(n), int(fi, c_int32_t), int(fo, c_int32_t), this%tape_p(t)%p, dz, this%grads(t)%p), "gemm_dw")
       this%has_grad(t) = .true.
       if(t .eq. 1 .and. .not. want_dx) return            ! the input layer's output has requires_grad = .false.
       call chk(athena_mp_kipf_layer_bwd_x(this%graph, int(fi, c_int32_t), int(fo, c_int32_t), dz, this%params(t)%p, &
            int(ex, c_int32_t), this%scratch(1)%p), "kipf_layer_bwd_x")
       ! the result becomes the upstream of step t-1: swap the roles of scratch 1 and 2 instead of copying
       call swap_buf(this%scratch(1), this%scratch(2))
       gcur = this%scratch(2)%p
    end do
    dx_dev = gcur
  end function kipf_backward_dev

  subroutine swap_buf(a, b)
    type(dbuf), intent(inout) :: a, b
    type(dbuf) :: t
    t = a
    a = b
    b = t
  end subroutine swap_buf

  subroutine kipf_destroy(this)
    class(kipf_mp_layer_type), intent(inout) :: this
    integer :: t
    do t = 1, this%num_time_steps
       call release(this%tape_p(t)); call release(this%tape_y(t)); call release(this%tape_z(t))
    end do
    call release(this%x_in); call release(this%up_in); call release(this%dual)
    do t = 1, 3
       call release(this%scratch(t))
    end do
    call this%release_base()
  end subroutine kipf_destroy

!###############################################################################
! Duvenaud neural fingerprint
!###############################################################################
  function duvenaud_setup(num_vertex_features, num_edge_features, num_time_steps, max_vertex_degree, num_outputs, &
       min_vertex_degree, message_activation, readout_activation) result(layer)
    !! layer_setup of athena_duvenaud_msgpass_layer.f90 (defaults :123-124: sigmoid / softmax);
    !! init_duvenaud :547-589 -- T message tensors W(F_t, F_{t-1}+F_e, D), then T readout tensors R(num_outputs, F_t)
    integer, intent(in) :: num_vertex_features(:), num_edge_features(:)
    integer, intent(in) :: num_time_steps, max_vertex_degree, num_outputs
    integer, intent(in), optional :: min_vertex_degree
    class(*), intent(in), optional :: message_activation, readout_activation
    type(duvenaud_mp_layer_type) :: layer
    integer, allocatable :: sizes(:)
    real(real32), allocatable :: w(:)
    integer :: t, d, fi, fo

    layer%num_time_steps = num_time_steps
    layer%num_outputs = num_outputs
    layer%max_vertex_degree = max_vertex_degree
    if(present(min_vertex_degree)) layer%min_vertex_degree = min_vertex_degree
    call expand_features(num_vertex_features, num_time_steps, "num_vertex_features", layer%num_vertex_features)
    call expand_features(num_edge_features, num_time_steps, "num_edge_features", layer%num_edge_features)
    layer%activation = resolve(message_activation, "sigmoid")
    layer%activation_readout = resolve(readout_activation, "softmax")
    layer%keep_edges = .true.
    d = layer%max_vertex_degree - layer%min_vertex_degree + 1
    allocate(sizes(2 * num_time_steps))
    do t = 1, num_time_steps
       sizes(t) = layer%num_vertex_features(t) * (layer%num_vertex_features(t - 1) + layer%num_edge_features(0)) * d
       sizes(num_time_steps + t) = num_outputs * layer%num_vertex_features(t)
    end do
    call layer%alloc_params(sizes)
    do t = 1, 2 * num_time_steps
       allocate(w(sizes(t)))
       if(t .le. num_time_steps)then
          fi = layer%num_vertex_features(t - 1) + layer%num_edge_features(0)
          fo = layer%num_vertex_features(t)
       else
          fi = sum(layer%num_vertex_features)
          fo = num_outputs
       end if
       call init_weights(w, fi, fo, relu_family(layer%activation))
       call chk(athena_mp_memcpy_h2d(layer%params(t)%p, w, 4_c_int64_t * i8(sizes(t))), "h2d")
       deallocate(w)
    end do
    allocate(layer%tape_a(num_time_steps), layer%tape_z(num_time_steps), layer%tape_c(num_time_steps), &
         layer%tape_p(num_time_steps), layer%tape_l(num_time_steps))
  end function duvenaud_setup

  function duvenaud_forward(this, vertex_features, edge_features) result(output)
    !! update_message_duvenaud :755-817 + update_readout_duvenaud :822-859;
    !! output(:, s) = sum_t sum_{v in graph s} act_readout( R_t z_t(:, v) )
    class(duvenaud_mp_layer_type), intent(inout) :: this
    real(real32), intent(in) :: vertex_features(:,:)       ! (F_v, vertices of the batch)
    real(real32), intent(in) :: edge_features(:,:)         ! (F_e, edge columns of the batch)
    real(real32), allocatable :: output(:,:)               ! (num_outputs, batch)
    type(c_ptr) :: out
    integer :: n, fe

    n = this%nv
    fe = this%num_edge_features(0)
    if(size(vertex_features, 1) .ne. this%num_vertex_features(0) .or. size(vertex_features, 2) .ne. n) &
         call stop_program("duvenaud forward: vertex feature shape mismatch")
    if(size(edge_features, 1) .ne. fe .or. size(edge_features, 2) .ne. this%ne) &
         call stop_program("duvenaud forward: edge feature shape mismatch")
    call need(this%x_in, i8(n) * i8(this%num_vertex_features(0)))
    call need(this%e_in, i8(max(this%ne, 1)) * i8(fe))
    call chk(athena_mp_memcpy_h2d(this%x_in%p, vertex_features, 4_c_int64_t * i8(n) * i8(this%num_vertex_features(0))), "h2d")
    if(this%ne .gt. 0) call chk(athena_mp_memcpy_h2d(this%e_in%p, edge_features, 4_c_int64_t * i8(this%ne) * i8(fe)), "h2d")
    out = this%forward_dev(this%x_in%p, this%e_in%p)
    allocate(output(this%num_outputs, this%batch))
    call chk(athena_mp_memcpy_d2h(output, out, 4_c_int64_t * i8(this%batch) * i8(this%num_outputs)), "d2h")
  end function duvenaud_forward

  function duvenaud_forward_dev(this, x_dev, e_dev) result(out_dev)
    !! the same on tensors resident in HBM: x_dev (F_v, vertices), e_dev (F_e, edge columns) -> (num_outputs, batch),
    !! a pointer into the layer (valid until the next forward)
    class(duvenaud_mp_layer_type), intent(inout) :: this
    type(c_ptr), intent(in) :: x_dev, e_dev
    type(c_ptr) :: out_dev
    type(c_ptr) :: cur
    integer :: t, tt, n, fv, fe, fin, fo, o
    integer(c_int32_t) :: code, acc
    logical :: softmax_ro

    if(.not. c_associated(this%graph)) call stop_program("set_graph must be called before forward")
    n = this%nv
    tt = this%num_time_steps
    fe = this%num_edge_features(0)
    o = this%num_outputs
    code = fused_code(this%activation)
    softmax_ro = trim(this%activation_readout%name) .eq. "softmax" .and. .not. apply_scaling(this%activation_readout)
    cur = x_dev
    ! a = [a_x | a_e] kept split for the whole layer where every time step is in the widths of the split launches: the edge
    ! features do not change from time step to time step (update_message_duvenaud passes the same edge_features to every
    ! duvenaud_propagate), so their neighbour sums are gathered ONCE; each time step gathers the vertex part into 256-byte rows
    this%split_a = code .ge. 0 .and. softmax_ro .and. o .le. 16 .and. all(this%num_vertex_features .eq. 64) .and. fe .gt. 0 &
         .and. fe .le. 16
    if(this%split_a)then
       call need(this%a_e, i8(max(n, 1)) * i8(fe))
       call chk(athena_mp_duvenaud_propagate_fwd(this%graph, 0_c_int32_t, int(fe, c_int32_t), c_null_ptr, e_dev, this%a_e%p), &
            "duvenaud_propagate (edge part, once)")
    end if
    do t = 1, tt
       fv = this%num_vertex_features(t - 1)
       fin = fv + fe
       fo = this%num_vertex_features(t)
       call need(this%tape_z(t), i8(n) * i8(fo))
       if(this%split_a)then
          call need(this%tape_a(t), i8(n) * i8(fv))
          call need(this%tape_p(t), i8(n) * i8(o))
          call chk(athena_mp_duvenaud_propagate_fwd(this%graph, int(fv, c_int32_t), 0_c_int32_t, cur, c_null_ptr, &
               this%tape_a(t)%p), "duvenaud_propagate (vertex part)")
          call chk(athena_mp_duvenaud_update_readout_fwd_split(this%graph, int(fv, c_int32_t), int(fe, c_int32_t), &
               int(fo, c_int32_t), int(this%min_vertex_degree, c_int32_t), int(this%max_vertex_degree, c_int32_t), &
               this%tape_a(t)%p, this%a_e%p, this%params(t)%p, code, this%tape_z(t)%p, int(o, c_int32_t), &
               this%params(tt + t)%p, this%tape_p(t)%p), "duvenaud_update + readout (split a)")
          cur = this%tape_z(t)%p
          cycle
       end if
       call need(this%tape_a(t), i8(n) * i8(fin))
       call chk(athena_mp_duvenaud_propagate_fwd(this%graph, int(fv, c_int32_t), int(fe, c_int32_t), cur, e_dev, &
            this%tape_a(t)%p), "duvenaud_propagate")
       if(code .ge. 0 .and. softmax_ro .and. o .le. 16)then
          ! the bucket contraction with the activation AND the readout's p = softmax(R z) in its epilogue: the readout
          ! loop below then only sums p per graph
          call need(this%tape_p(t), i8(n) * i8(o))
          call chk(athena_mp_duvenaud_update_readout_fwd(this%graph, int(fin, c_int32_t), int(fo, c_int32_t), &
               int(this%min_vertex_degree, c_int32_t), int(this%max_vertex_degree, c_int32_t), this%tape_a(t)%p, &
               this%params(t)%p, code, this%tape_z(t)%p, int(o, c_int32_t), this%params(tt + t)%p, this%tape_p(t)%p), &
               "duvenaud_update + readout")
       else if(code .ge. 0)then
          call chk(athena_mp_duvenaud_update_act_fwd(this%graph, int(fin, c_int32_t), int(fo, c_int32_t), &
               int(this%min_vertex_degree, c_int32_t), int(this%max_vertex_degree, c_int32_t), this%tape_a(t)%p, &
               this%params(t)%p, code, this%tape_z(t)%p), "duvenaud_update")
       else
          call need(this%tape_c(t), i8(n) * i8(fo))
          call chk(athena_mp_duvenaud_update_act_fwd(this%graph, int(fin, c_int32_t), int(fo, c_int32_t), &
               int(this%min_vertex_degree, c_int32_t), int(this%max_vertex_degree, c_int32_t), this%tape_a(t)%p, &
               this%params(t)%p, ATHENA_MP_ACT_NONE, this%tape_c(t)%p), "duvenaud_update")
          call act_apply(this%activation, n, fo, this%tape_c(t)%p, this%tape_z(t)%p)
       end if
       cur = this%tape_z(t)%p
    end do
    call need(this%out_dev, i8(this%batch) * i8(o))
    do t = 1, tt
       fo = this%num_vertex_features(t)
       acc = merge(1_c_int32_t, 0_c_int32_t, t .gt. 1)
       call need(this%tape_p(t), i8(n) * i8(o))
       if(softmax_ro .and. code .ge. 0 .and. o .le. 16)then
          call chk(athena_mp_segment_sum(int(o, c_int32_t), i8(n), int(this%batch, c_int32_t), this%seg%p, this%tape_p(t)%p, &
               this%out_dev%p, acc), "segment_sum")            ! p came out of the update launch
       else if(softmax_ro)then
          call chk(athena_mp_duvenaud_readout_fwd(i8(n), int(fo, c_int32_t), int(o, c_int32_t), int(this%batch, c_int32_t), &
               this%seg%p, this%tape_z(t)%p, this%params(tt + t)%p, this%tape_p(t)%p, this%out_dev%p, acc), "readout")
       else
          call need(this%tape_l(t), i8(n) * i8(o))
          call chk(athena_mp_gemm_fwd(i8(n), int(fo, c_int32_t), int(o, c_int32_t), this%tape_z(t)%p, this%params(tt + t)%p, &
               c_null_ptr, ATHENA_MP_ACT_NONE, this%tape_l(t)%p), "gemm_fwd")
          if(is_identity(this%activation_readout))then
             call copy_dev(this%tape_p(t)%p, this%tape_l(t)%p, i8(n) * i8(o))
          else
             call act_apply(this%activation_readout, n, o, this%tape_l(t)%p, this%tape_p(t)%p)
          end if
          call chk(athena_mp_segment_sum(int(o, c_int32_t), i8(n), int(this%batch, c_int32_t), this%seg%p, this%tape_p(t)%p, &
               this%out_dev%p, acc), "segment_sum")
       end if
    end do
    out_dev = this%out_dev%p
  end function duvenaud_forward_dev

  subroutine duvenaud_backward(this, upstream, dx, de)
    !! reverse pass over the readout branches and the message chain; dx / de are produced when present
    class(duvenaud_mp_layer_type), intent(inout) :: this
    real(real32), intent(in) :: upstream(:,:)                         ! (num_outputs, batch)
    real(real32), allocatable, intent(out), optional :: dx(:,:)       ! (F_v, vertices)
    real(real32), allocatable, intent(out), optional :: de(:,:)       ! (F_e, edge columns)
    type(c_ptr) :: dxp, dep
    integer :: n, o, fe

    n = this%nv
    o = this%num_outputs
    fe = this%num_edge_features(0)
    if(size(upstream, 1) .ne. o .or. size(upstream, 2) .ne. this%batch) &
         call stop_program("duvenaud backward: upstream shape mismatch")
    call need(this%gout, i8(this%batch) * i8(o))
    call chk(athena_mp_memcpy_h2d(this%gout%p, upstream, 4_c_int64_t * i8(this%batch) * i8(o)), "h2d")
    if(present(dx) .and. present(de))then
       call this%backward_dev(this%gout%p, dx_dev=dxp, de_dev=dep)
    else if(present(dx))then
       call this%backward_dev(this%gout%p, dx_dev=dxp)
    else if(present(de))then
       call this%backward_dev(this%gout%p, de_dev=dep)
    else
       call this%backward_dev(this%gout%p)
    end if
    if(present(dx))then
       allocate(dx(this%num_vertex_features(0), n))
       call chk(athena_mp_memcpy_d2h(dx, dxp, 4_c_int64_t * i8(n) * i8(this%num_vertex_features(0))), "d2h")
    end if
    if(present(de))then
       allocate(de(fe, this%ne))
       if(this%ne .gt. 0) call chk(athena_mp_memcpy_d2h(de, dep, 4_c_int64_t * i8(this%ne) * i8(fe)), "d2h")
    end if
  end subroutine duvenaud_backward

  subroutine duvenaud_backward_dev(this, upstream_dev, dx_dev, de_dev)
    !! upstream_dev (num_outputs, batch) on the device; dx_dev / de_dev, when present, return pointers into the
    !! layer's scratch (valid until the next backward).  Parameter gradients land in the layer.
    class(duvenaud_mp_layer_type), intent(inout) :: this
    type(c_ptr), intent(in) :: upstream_dev
    type(c_ptr), intent(out), optional :: dx_dev, de_dev
    type(c_ptr) :: gout, dzn, dzn_arg, dc, da, da_e, dl, tmp
    integer :: t, tt, n, o, fe, fv, fo, fin, fmax, fv_e, fe_x
    logical :: have_next, fused_msg, first_de, softmax_readout, split, one_call, have_e_sum
    integer(c_int32_t) :: acc_e
    type(c_ptr) :: a_e_arg
    integer(c_int32_t) :: code, code_arg

    n = this%nv
    tt = this%num_time_steps
    o = this%num_outputs
    fe = this%num_edge_features(0)
    fmax = max(maxval(this%num_vertex_features) + fe, o)
    call need(this%scratch(1), i8(max(n, 1)) * i8(fmax))      ! dl / temporary
    call need(this%scratch(2), i8(max(n, 1)) * i8(fmax))      ! dc
    call need(this%scratch(3), i8(max(n, 1)) * i8(fmax))      ! da
    call need(this%scratch(4), i8(max(n, 1)) * i8(fmax))      ! dz arriving from step t+1
    call need(this%scratch(5), i8(max(n, 1)) * i8(fmax))      ! readout logits' gradient after its activation
    gout = upstream_dev
    code = fused_code(this%activation)
    fused_msg = code .ge. 0
    code_arg = ATHENA_MP_ACT_NONE
    if(fused_msg) code_arg = code
    softmax_readout = trim(this%activation_readout%name) .eq. "softmax" .and. .not. apply_scaling(this%activation_readout)
    have_next = .false.
    first_de = .true.
    have_e_sum = .false.
    dc = this%scratch(2)%p
    da = this%scratch(3)%p
    dzn = this%scratch(4)%p
    if(present(de_dev) .and. this%ne .gt. 0) call need(this%de_acc, i8(this%ne) * i8(fe))
    do t = tt, 1, -1
       fv = this%num_vertex_features(t - 1)
       fo = this%num_vertex_features(t)
       fin = fv + fe
       ! the readout's reverse and the update's reverse of this time step in ONE call where the library's fused launch covers the
       ! shape: dc (n, 64) never reaches HBM (profiles/r05_c3_readout_update_fused_ab.txt)
       one_call = this%split_a .or. (softmax_readout .and. fused_msg .and. fv .eq. 64 .and. fo .eq. 64 .and. fe .gt. 0 .and. &
            (t .gt. 1 .or. present(dx_dev) .or. present(de_dev)))
       a_e_arg = c_null_ptr
       if(this%split_a) a_e_arg = this%a_e%p
       if(one_call)then
          dzn_arg = c_null_ptr
          if(have_next) dzn_arg = dzn
          ! ... and the edge part of da is SUMMED over these time steps in its own buffer (the scatter to the edge features is
          ! linear in it): one duvenaud_propagate reverse (edges) after the loop instead of one + an axpy per time step
          call need(this%dae_sum, i8(max(n, 1)) * i8(fe))
          da_e = this%dae_sum%p
          acc_e = 0_c_int32_t
          if(have_e_sum) acc_e = 1_c_int32_t
          call chk(athena_mp_duvenaud_readout_update_bwd(this%graph, int(fv, c_int32_t), int(fe, c_int32_t), &
               int(this%min_vertex_degree, c_int32_t), int(this%max_vertex_degree, c_int32_t), int(o, c_int32_t), &
               int(this%batch, c_int32_t), this%seg%p, this%tape_z(t)%p, this%params(tt + t)%p, this%tape_p(t)%p, gout, dzn_arg, &
               code_arg, this%tape_a(t)%p, this%params(t)%p, da, da_e, this%grads(t)%p, this%grads(tt + t)%p, 0_c_int32_t, &
               acc_e, a_e_arg), "readout + update reverse (one call)")
          have_e_sum = .true.
          this%has_grad(tt + t) = .true.
          this%has_grad(t) = .true.
          fv_e = 0
          fe_x = 0
       else
       if(softmax_readout)then
          dzn_arg = c_null_ptr
          if(have_next) dzn_arg = dzn
          call chk(athena_mp_duvenaud_readout_bwd(i8(n), int(fo, c_int32_t), int(o, c_int32_t), int(this%batch, c_int32_t), &
               this%seg%p, this%tape_z(t)%p, this%params(tt + t)%p, this%tape_p(t)%p, gout, dzn_arg, code_arg, dc, &
               this%grads(tt + t)%p, 0_c_int32_t), "readout reverse")
       else
          dl = this%scratch(1)%p
          call chk(athena_mp_segment_sum_bwd(int(o, c_int32_t), i8(n), int(this%batch, c_int32_t), this%seg%p, gout, dl), &
               "segment_sum reverse")
          if(.not. is_identity(this%activation_readout))then
             call act_reverse(this%activation_readout, n, o, this%tape_p(t)%p, this%tape_l(t)%p, dl, this%scratch(5)%p)
             dl = this%scratch(5)%p
          end if
          call chk(athena_mp_gemm_dw(i8(n), int(fo, c_int32_t), int(o, c_int32_t), this%tape_z(t)%p, dl, this%grads(tt + t)%p), &
               "gemm_dw")
          call chk(athena_mp_gemm_dx(i8(n), int(fo, c_int32_t), int(o, c_int32_t), dl, this%params(tt + t)%p, dc), "gemm_dx")
          if(have_next) call chk(athena_mp_axpy(i8(n) * i8(fo), 1._real32, dzn, dc), "axpy")
       end if
       this%has_grad(tt + t) = .true.
       ! the message activation, where it did not ride in the fused reverse kernel
       if((softmax_readout .and. .not. fused_msg) .or. (.not. softmax_readout .and. .not. is_identity(this%activation)))then
          tmp = this%scratch(1)%p
          call act_reverse(this%activation, n, fo, this%tape_z(t)%p, this%tape_c(t)%p, dc, tmp)
          call copy_dev(dc, tmp, i8(n) * i8(fo))
       end if
       ! message branch
       this%has_grad(t) = .true.
       if(t .eq. 1 .and. .not. (present(dx_dev) .or. present(de_dev)))then
          call chk(athena_mp_duvenaud_update_bwd_w(this%graph, int(fin, c_int32_t), int(fo, c_int32_t), &
               int(this%min_vertex_degree, c_int32_t), int(this%max_vertex_degree, c_int32_t), dc, this%tape_a(t)%p, &
               this%grads(t)%p), "duvenaud_update reverse (weights)")
          exit
       end if
       ! both reverse products of the update from one pass over dc.  At F_v = 64 with edge features da is written SPLIT where
       ! it is produced -- da_x (n, 64) in `da`, da_e (n, fe) behind it -- so that the two propagate partials gather whole cache
       ! lines (bit-identical dx / de; profiles/r05_c3_split_da_ab.txt)
       split = fv .eq. 64 .and. fe .gt. 0
       if(split)then
          da_e = athena_mp_dev_offset(da, i8(n) * i8(fv))
          call chk(athena_mp_duvenaud_update_bwd_split(this%graph, int(fv, c_int32_t), int(fe, c_int32_t), int(fo, c_int32_t), &
               int(this%min_vertex_degree, c_int32_t), int(this%max_vertex_degree, c_int32_t), dc, this%tape_a(t)%p, &
               this%params(t)%p, da, da_e, this%grads(t)%p), "duvenaud_update reverse (split)")
          fv_e = 0
          fe_x = 0
       else
          da_e = da
          call chk(athena_mp_duvenaud_update_bwd(this%graph, int(fin, c_int32_t), int(fo, c_int32_t), &
               int(this%min_vertex_degree, c_int32_t), int(this%max_vertex_degree, c_int32_t), dc, this%tape_a(t)%p, &
               this%params(t)%p, da, this%grads(t)%p), "duvenaud_update reverse")
          fv_e = fv
          fe_x = fe
       end if
       end if
       if(present(de_dev) .and. this%ne .gt. 0 .and. .not. one_call)then
          if(first_de)then
             call chk(athena_mp_duvenaud_propagate_bwd_e(this%graph, int(fv_e, c_int32_t), int(fe, c_int32_t), da_e, &
                  this%de_acc%p), "duvenaud_propagate reverse (edges)")
             first_de = .false.
          else
             call chk(athena_mp_duvenaud_propagate_bwd_e(this%graph, int(fv_e, c_int32_t), int(fe, c_int32_t), da_e, &
                  this%scratch(1)%p), "duvenaud_propagate reverse (edges)")
             call chk(athena_mp_axpy(i8(this%ne) * i8(fe), 1._real32, this%scratch(1)%p, this%de_acc%p), "axpy")
          end if
       end if
       if(t .gt. 1 .or. present(dx_dev))then
          call chk(athena_mp_duvenaud_propagate_bwd_x(this%graph, int(fv, c_int32_t), int(fe_x, c_int32_t), da, dzn), &
               "duvenaud_propagate reverse (vertices)")
          have_next = .true.
       end if
    end do
    ! the time steps that went through the one-call reverse left the SUM of their da_e: one scatter to the edge features
    if(present(de_dev) .and. this%ne .gt. 0 .and. have_e_sum)then
       if(first_de)then
          call chk(athena_mp_duvenaud_propagate_bwd_e(this%graph, 0_c_int32_t, int(fe, c_int32_t), this%dae_sum%p, &
               this%de_acc%p), "duvenaud_propagate reverse (edges, summed over the time steps)")
          first_de = .false.
       else
          call chk(athena_mp_duvenaud_propagate_bwd_e(this%graph, 0_c_int32_t, int(fe, c_int32_t), this%dae_sum%p, &
               this%scratch(1)%p), "duvenaud_propagate reverse (edges, summed over the time steps)")
          call chk(athena_mp_axpy(i8(this%ne) * i8(fe), 1._real32, this%scratch(1)%p, this%de_acc%p), "axpy")
       end if
    end if
    if(present(dx_dev)) dx_dev = dzn
    if(present(de_dev)) de_dev = this%de_acc%p
  end subroutine duvenaud_backward_dev

  subroutine duvenaud_destroy(this)
    class(duvenaud_mp_layer_type), intent(inout) :: this
    integer :: t
    do t = 1, this%num_time_steps
       call release(this%tape_a(t)); call release(this%tape_z(t)); call release(this%tape_c(t))
       call release(this%tape_p(t)); call release(this%tape_l(t))
    end do
    call release(this%x_in); call release(this%e_in); call release(this%out_dev)
    call release(this%gout); call release(this%de_acc); call release(this%dae_sum); call release(this%a_e)
    do t = 1, 5
       call release(this%scratch(t))
    end do
    call this%release_base()
  end subroutine duvenaud_destroy

!###############################################################################
! graph neural operator
!###############################################################################
  function gno_setup(num_outputs, coord_dim, num_inputs, kernel_hidden, use_bias, activation) result(layer)
    !! layer_setup / init_gno of athena_graph_nop_layer.f90:357-458 -- params(1) = [U(H,d) | b_u(H) | V(F,H) | b_v(F)],
    !! F = num_outputs * num_inputs; params(2) = W(num_outputs, num_inputs); params(3) = b(num_outputs) if use_bias
    integer, intent(in) :: num_outputs, coord_dim, num_inputs
    integer, intent(in), optional :: kernel_hidden
    logical, intent(in), optional :: use_bias
    class(*), intent(in), optional :: activation
    type(graph_nop_mp_layer_type) :: layer
    real(real32), allocatable :: theta(:), w(:)
    integer :: f, h, d, nt

    layer%num_outputs = num_outputs
    layer%num_inputs = num_inputs
    layer%coord_dim = coord_dim
    if(present(kernel_hidden)) layer%kernel_hidden = kernel_hidden
    if(present(use_bias)) layer%use_bias = use_bias
    layer%activation = resolve(activation, "none")
    layer%keep_edges = .true.
    f = num_outputs * num_inputs
    h = layer%kernel_hidden
    d = coord_dim
    nt = h * d + h + f * h + f
    if(layer%use_bias)then
       call layer%alloc_params([nt, f, num_outputs])
    else
       call layer%alloc_params([nt, f])
    end if
    allocate(theta(nt), w(f))
    theta = 0._real32
    call init_weights(theta(1:h * d), d, h, relu_family(layer%activation))
    call init_weights(theta(h * d + h + 1:h * d + h + f * h), h, f, relu_family(layer%activation))
    call init_weights(w, num_inputs + merge(1, 0, layer%use_bias), num_outputs, relu_family(layer%activation))
    call chk(athena_mp_memcpy_h2d(layer%params(1)%p, theta, 4_c_int64_t * i8(nt)), "h2d")
    call chk(athena_mp_memcpy_h2d(layer%params(2)%p, w, 4_c_int64_t * i8(f)), "h2d")
    if(layer%use_bias) call chk(athena_mp_memset_zero(layer%params(3)%p, 4_c_int64_t * i8(num_outputs)), "memset")
  end function gno_setup

  function gno_forward(this, vertex_features, edge_features) result(output)
    !! update_message_gno :690-788 on host arrays: upload, forward_dev, download
    class(graph_nop_mp_layer_type), intent(inout) :: this
    real(real32), intent(in) :: vertex_features(:,:)       ! (num_inputs, vertices)
    real(real32), intent(in) :: edge_features(:,:)         ! (coord_dim, edge columns): the pair's coordinate difference
    real(real32), allocatable :: output(:,:)
    type(c_ptr) :: y
    integer :: n, fi, fo

    n = this%nv
    fi = this%num_inputs
    fo = this%num_outputs
    if(size(vertex_features, 1) .ne. fi .or. size(vertex_features, 2) .ne. n) &
         call stop_program("graph_nop forward: vertex feature shape mismatch")
    if(size(edge_features, 1) .ne. this%coord_dim .or. size(edge_features, 2) .ne. this%ne) &
         call stop_program("graph_nop layer expects vertex and edge feature inputs")       ! :725-728
    call need(this%x_in, i8(n) * i8(fi))
    call need(this%c_in, i8(max(this%ne, 1)) * i8(this%coord_dim))
    call chk(athena_mp_memcpy_h2d(this%x_in%p, vertex_features, 4_c_int64_t * i8(n) * i8(fi)), "h2d")
    if(this%ne .gt. 0) call chk(athena_mp_memcpy_h2d(this%c_in%p, edge_features, 4_c_int64_t * i8(this%ne) * i8(this%coord_dim)), "h2d")
    y = this%forward_dev(this%x_in%p, this%c_in%p)
    allocate(output(fo, n))
    call chk(athena_mp_memcpy_d2h(output, y, 4_c_int64_t * i8(n) * i8(fo)), "d2h")
  end function gno_forward

  function gno_forward_dev(this, x_dev, coords_dev) result(y_dev)
    !! out = act( gno_aggregate(kappa(coords), x) + W x + b ) on tensors resident in HBM.  x_dev and coords_dev are
    !! part of the tape: they must stay valid until backward_dev has run.  The edge geometry is forwarded unchanged
    !! to a following graph_nop layer (output(2,s), :781-785) -- pass the same coords_dev on.
    class(graph_nop_mp_layer_type), intent(inout) :: this
    type(c_ptr), intent(in) :: x_dev, coords_dev
    type(c_ptr) :: y_dev
    type(c_ptr) :: bias, s_new
    integer :: n, fi, fo
    integer(c_int64_t) :: s_bytes

    if(.not. c_associated(this%graph)) call stop_program("set_graph must be called before forward")
    n = this%nv
    fi = this%num_inputs
    fo = this%num_outputs
    this%x_tape = x_dev
    this%c_tape = coords_dev
    call need(this%z_pre, i8(n) * i8(fo))
    call need(this%y_out, i8(n) * i8(fo))
    call need(this%scratch(1), i8(n) * i8(max(fi, fo)))
    this%s_valid = .false.
    s_bytes = 0_c_int64_t
    if(this%keep_s .and. .not. this%inference) call chk(athena_mp_gno_saved_bytes(this%graph, int(this%coord_dim, c_int32_t), &
         int(this%kernel_hidden, c_int32_t), int(fi, c_int32_t), int(fo, c_int32_t), s_bytes), "gno_saved_bytes")
    if(s_bytes .gt. 0_c_int64_t .and. real(s_bytes) .le. keep_s_max_gb() * 1.e9)then
       if(s_bytes / 4_c_int64_t .gt. this%s_save%cap)then       ! no room for S beside the model: rebuild it from now on
          ! the new buffer first, the old one only once the new one is known to exist; when both do not fit side by
          ! side, once more with the old one gone; a refusal leaves no error behind (athena_mp_malloc clears it) and
          ! the layer rebuilds S in the reverse pass from now on
          s_new = c_null_ptr
          if(athena_mp_malloc(s_new, s_bytes) .ne. 0)then
             call release(this%s_save)
             s_new = c_null_ptr
             if(athena_mp_malloc(s_new, s_bytes) .ne. 0) s_new = c_null_ptr
          end if
          call release(this%s_save)
          if(c_associated(s_new))then
             this%s_save%p = s_new
             this%s_save%cap = s_bytes / 4_c_int64_t
          else
             this%keep_s = .false.
             s_bytes = 0_c_int64_t
          end if
       end if
    end if
    if(s_bytes .gt. 0_c_int64_t .and. real(s_bytes) .le. keep_s_max_gb() * 1.e9)then
       call chk(athena_mp_gno_aggregate_fwd_save(this%graph, int(this%coord_dim, c_int32_t), int(this%kernel_hidden, c_int32_t), &
            int(fi, c_int32_t), int(fo, c_int32_t), this%params(1)%p, coords_dev, x_dev, this%scratch(1)%p, this%s_save%p), &
            "gno_aggregate (S kept)")
       this%s_valid = .true.
       this%s_graph = this%graph
    else
       call chk(athena_mp_gno_aggregate_fwd(this%graph, int(this%coord_dim, c_int32_t), int(this%kernel_hidden, c_int32_t), &
            int(fi, c_int32_t), int(fo, c_int32_t), this%params(1)%p, coords_dev, x_dev, this%scratch(1)%p), "gno_aggregate")
    end if
    bias = c_null_ptr
    if(this%use_bias) bias = this%params(3)%p
    call chk(athena_mp_gemm_fwd(i8(n), int(fi, c_int32_t), int(fo, c_int32_t), x_dev, this%params(2)%p, bias, &
         ATHENA_MP_ACT_NONE, this%z_pre%p), "gemm_fwd")
    call chk(athena_mp_axpy(i8(n) * i8(fo), 1._real32, this%scratch(1)%p, this%z_pre%p), "axpy")
    if(is_identity(this%activation))then
       y_dev = this%z_pre%p
    else
       call act_apply(this%activation, n, fo, this%z_pre%p, this%y_out%p)
       y_dev = this%y_out%p
    end if
  end function gno_forward_dev

  subroutine gno_backward(this, upstream, dx, dcoords)
    !! reverse pass on host arrays
    class(graph_nop_mp_layer_type), intent(inout) :: this
    real(real32), intent(in) :: upstream(:,:)                              ! (num_outputs, vertices)
    real(real32), allocatable, intent(out), optional :: dx(:,:)            ! (num_inputs, vertices)
    real(real32), allocatable, intent(out), optional :: dcoords(:,:)       ! (coord_dim, edge columns)
    type(c_ptr) :: dxp, dcp
    integer :: n, fi, fo, d

    n = this%nv
    fi = this%num_inputs
    fo = this%num_outputs
    d = this%coord_dim
    if(size(upstream, 1) .ne. fo .or. size(upstream, 2) .ne. n) call stop_program("graph_nop backward: upstream shape mismatch")
    call need(this%up_in, i8(n) * i8(fo))
    call chk(athena_mp_memcpy_h2d(this%up_in%p, upstream, 4_c_int64_t * i8(n) * i8(fo)), "h2d")
    if(present(dx) .and. present(dcoords))then
       call this%backward_dev(this%up_in%p, dx_dev=dxp, dcoords_dev=dcp)
    else if(present(dx))then
       call this%backward_dev(this%up_in%p, dx_dev=dxp)
    else if(present(dcoords))then
       call this%backward_dev(this%up_in%p, dcoords_dev=dcp)
    else
       call this%backward_dev(this%up_in%p)
    end if
    if(present(dx))then
       allocate(dx(fi, n))
       call chk(athena_mp_memcpy_d2h(dx, dxp, 4_c_int64_t * i8(n) * i8(fi)), "d2h")
    end if
    if(present(dcoords))then
       allocate(dcoords(d, this%ne))
       if(this%ne .gt. 0) call chk(athena_mp_memcpy_d2h(dcoords, dcp, 4_c_int64_t * i8(this%ne) * i8(d)), "d2h")
    end if
  end subroutine gno_backward

  subroutine gno_backward_dev(this, upstream_dev, dx_dev, dcoords_dev)
    !! upstream_dev (num_outputs, vertices) on the device (read only); dx_dev / dcoords_dev, when present, return
    !! pointers into the layer's scratch (valid until the next backward)
    class(graph_nop_mp_layer_type), intent(inout) :: this
    type(c_ptr), intent(in) :: upstream_dev
    type(c_ptr), intent(out), optional :: dx_dev, dcoords_dev
    type(c_ptr) :: dz, y, s_arg, dx_arg, dc_arg
    real(real32), allocatable :: ones(:)
    integer :: n, fi, fo, d, h
    integer(c_int32_t) :: fused

    n = this%nv
    fi = this%num_inputs
    fo = this%num_outputs
    d = this%coord_dim
    h = this%kernel_hidden
    call need(this%scratch(1), i8(n) * i8(max(fi, fo)))
    call need(this%scratch(2), i8(n) * i8(max(fi, fo)))
    call need(this%scratch(3), i8(n) * i8(max(fi, fo)))
    if(is_identity(this%activation))then
       dz = upstream_dev
    else
       dz = this%scratch(2)%p
       y = this%y_out%p
       call act_reverse(this%activation, n, fo, y, this%z_pre%p, upstream_dev, dz)
    end if
    if(this%use_bias)then
       ! db(o) = sum_v dz(o, v): the contraction of a column of ones with dz
       if(this%ones%cap .lt. i8(n))then
          call need(this%ones, i8(n))
          allocate(ones(n))
          ones = 1._real32
          call chk(athena_mp_memcpy_h2d(this%ones%p, ones, 4_c_int64_t * i8(n)), "h2d")
       end if
       call chk(athena_mp_gemm_dw(i8(n), 1_c_int32_t, int(fo, c_int32_t), this%ones%p, dz, this%grads(3)%p), "gemm_dw")
       this%has_grad(3) = .true.
    end if
    call chk(athena_mp_gemm_dw(i8(n), int(fi, c_int32_t), int(fo, c_int32_t), this%x_tape, dz, this%grads(2)%p), "gemm_dw")
    ! the whole reverse pass of gno_aggregate from ONE G = dz . Vmat^T (athena_mp_gno_aggregate_bwd): dtheta always, the
    ! feature gradient and the coordinate gradient when asked for; S of the forward pass when it was kept
    s_arg = c_null_ptr
    if(this%s_valid .and. c_associated(this%s_graph, this%graph)) s_arg = this%s_save%p
    dx_arg = c_null_ptr
    if(present(dx_dev)) dx_arg = this%scratch(1)%p
    dc_arg = c_null_ptr
    if(present(dcoords_dev))then
       dcoords_dev = c_null_ptr
       if(this%ne .gt. 0)then
          call need(this%dc_out, i8(this%ne) * i8(d))
          dc_arg = this%dc_out%p
          dcoords_dev = this%dc_out%p
       end if
    end if
    call chk(athena_mp_gno_aggregate_bwd(this%graph, int(d, c_int32_t), int(h, c_int32_t), int(fi, c_int32_t), int(fo, c_int32_t), &
         this%params(1)%p, this%c_tape, this%x_tape, dz, s_arg, dx_arg, this%grads(1)%p, dc_arg, fused), "gno reverse pass")
    this%has_grad(1:2) = .true.
    if(present(dx_dev))then
       call chk(athena_mp_gemm_dx(i8(n), int(fi, c_int32_t), int(fo, c_int32_t), dz, this%params(2)%p, this%scratch(3)%p), "gemm_dx")
       call chk(athena_mp_axpy(i8(n) * i8(fi), 1._real32, this%scratch(1)%p, this%scratch(3)%p), "axpy")
       dx_dev = this%scratch(3)%p
    end if
  end subroutine gno_backward_dev

  subroutine gno_destroy(this)
    class(graph_nop_mp_layer_type), intent(inout) :: this
    integer :: t
    call release(this%x_in); call release(this%c_in); call release(this%z_pre); call release(this%y_out)
    call release(this%up_in); call release(this%ones); call release(this%dc_out); call release(this%s_save)
    this%s_valid = .false.
    do t = 1, 3
       call release(this%scratch(t))
    end do
    call this%release_base()
  end subroutine gno_destroy

end module athena_mp_layers
