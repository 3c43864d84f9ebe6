!! src/athena/athena_hip_msgpass_layers.f90 of an athena checkout (athena_dropin/install.sh puts it there): the three
!! message-passing layers on libathena_mp.so as DROP-IN layer types.
!!
!!   hip_kipf_msgpass_layer_type      extends(kipf_msgpass_layer_type)      athena_kipf_msgpass_layer.f90:43-98
!!   hip_duvenaud_msgpass_layer_type  extends(duvenaud_msgpass_layer_type)  athena_duvenaud_msgpass_layer.f90:39-121
!!   hip_graph_nop_layer_type         extends(graph_nop_layer_type)         athena_graph_nop_layer.f90:61-104
!!
!! Each extends athena's CONCRETE type and overrides only what touches the device: set_graph (the parent's copies + one cached
!! device handle per sample), update_message (the parent's statements with the HIP ops of athena__hip_msgpass_ops in place of
!! kipf_propagate / duvenaud_propagate / duvenaud_update / gno_kernel_eval / gno_aggregate / matmul), update_readout for
!! Duvenaud, and a finaliser that hands the handles back.  Everything else IS the reference's: get_num_params, set_hyperparams,
!! init (parameter layout, flags, initialisers), print_to_unit / read (the KIPF / DUVENAUD / GRAPH_NOP cards), get_attributes
!! and the ONNX emitters, get_params / set_params / get_gradients / set_gradients (athena_base_layer_sub.f90:545-691),
!! forward (forward_msgpass, athena_msgpass_layer_sub.f90:184-198).  The constructors take the parent constructors' arguments
!! and build the parent part WITH the parent constructor, so `network%add(hip_kipf_msgpass_layer_type(...))` is the one-word
!! change in a user's program (athena_network_sub.f90:764 accepts any class(base_layer_type)).
!!
!! Checkpoints: a hip_* layer keeps its parent's name ('kipf', 'duvenaud', 'graph_nop'), writes the parent's card and reads it
!! back with the parent's reader.  register_hip_msgpass_layers() puts readers that return the hip_* types IN FRONT of the stock
!! ones in athena's registry (list_of_layer_types, athena_container_layer.f90:57-71; network read takes the first match,
!! athena_network_sub.f90:433-447), so every saved network -- also one written by stock athena -- loads onto the device;
!! without that call the same file loads as stock layers.
module athena__hip_msgpass_layers
  use, intrinsic :: iso_c_binding
  use coreutils, only: real32, stop_program
  use graphstruc, only: graph_type
  use diffstruc, only: array_type, sum, matmul, operator(+)
  use athena__misc_types, only: base_actv_type
  use athena__base_layer, only: base_layer_type
  use athena__kipf_msgpass_layer, only: kipf_msgpass_layer_type
  use athena__duvenaud_msgpass_layer, only: duvenaud_msgpass_layer_type
  use athena__graph_nop_layer, only: graph_nop_layer_type
  use athena__container_layer, only: read_layer_container, list_of_layer_types, allocate_list_of_layer_types
  use athena__diffstruc_extd, only: add_bias
  use athena_mp_c
  use athena__hip_msgpass_ops, only: kipf_propagate_hip, matmul_hip, duvenaud_propagate_hip, duvenaud_update_hip, &
       gno_kernel_hip, gno_aggregate_hip, duvenaud_update_act_readout_hip, get_partial_readout_softmax_hip_z_val, &
       get_partial_readout_softmax_hip_weight_val
  implicit none
  private
  public :: hip_kipf_msgpass_layer_type, hip_duvenaud_msgpass_layer_type, hip_graph_nop_layer_type
  public :: read_hip_kipf_msgpass_layer, read_hip_duvenaud_msgpass_layer, read_hip_graph_nop_layer
  public :: hip_msgpass_layer_readers, register_hip_msgpass_layers

  type, extends(kipf_msgpass_layer_type) :: hip_kipf_msgpass_layer_type
     type(c_ptr), allocatable :: handle(:)
     !! one device graph per sample: a reference-counted handle of the library's content-keyed cache (SURVEY F12)
   contains
     procedure, pass(this) :: set_graph => set_graph_hip_kipf
     procedure, pass(this) :: update_message => update_message_hip_kipf
     final :: finalise_hip_kipf
  end type hip_kipf_msgpass_layer_type

  interface hip_kipf_msgpass_layer_type
     module procedure hip_kipf_layer_setup
  end interface hip_kipf_msgpass_layer_type

  type, extends(duvenaud_msgpass_layer_type) :: hip_duvenaud_msgpass_layer_type
     type(array_type), allocatable, dimension(:,:) :: p
     !! (num_time_steps, batch), fused path only: the readout's per-vertex softmax(R_t z_t) -- nodes of the tape in their own
     !! right (left operand z(t,s), right operand params(T+t)), filled by the SAME launch that produces z(t,s)
     type(c_ptr), allocatable :: handle(:)
   contains
     procedure, pass(this) :: set_graph => set_graph_hip_duvenaud
     procedure, pass(this) :: update_message => update_message_hip_duvenaud
     procedure, pass(this) :: update_readout => update_readout_hip_duvenaud
     final :: finalise_hip_duvenaud
  end type hip_duvenaud_msgpass_layer_type

  interface hip_duvenaud_msgpass_layer_type
     module procedure hip_duvenaud_layer_setup
  end interface hip_duvenaud_msgpass_layer_type

  type, extends(graph_nop_layer_type) :: hip_graph_nop_layer_type
     type(c_ptr), allocatable :: handle(:)
   contains
     procedure, pass(this) :: set_graph => set_graph_hip_gno
     procedure, pass(this) :: update_message => update_message_hip_gno
     final :: finalise_hip_gno
  end type hip_graph_nop_layer_type

  interface hip_graph_nop_layer_type
     module procedure hip_graph_nop_layer_setup
  end interface hip_graph_nop_layer_type

contains

  ! ============================================================================================ constructors
  function hip_kipf_layer_setup(num_vertex_features, num_time_steps, activation, kernel_initialiser, verbose) result(layer)
    !! kipf_msgpass_layer_type's constructor (athena_kipf_msgpass_layer.f90:143-212), same arguments
    integer, dimension(:), intent(in) :: num_vertex_features
    integer, intent(in) :: num_time_steps
    class(*), optional, intent(in) :: activation, kernel_initialiser
    integer, optional, intent(in) :: verbose
    type(hip_kipf_msgpass_layer_type) :: layer
    layer%kipf_msgpass_layer_type = kipf_msgpass_layer_type(num_vertex_features=num_vertex_features, &
         num_time_steps=num_time_steps, activation=activation, kernel_initialiser=kernel_initialiser, verbose=verbose)
  end function hip_kipf_layer_setup

  function hip_duvenaud_layer_setup(num_vertex_features, num_edge_features, num_time_steps, max_vertex_degree, num_outputs, &
       min_vertex_degree, message_activation, readout_activation, kernel_initialiser, verbose) result(layer)
    !! duvenaud_msgpass_layer_type's constructor (athena_duvenaud_msgpass_layer.f90:254-351), same arguments
    integer, dimension(:), intent(in) :: num_vertex_features
    integer, dimension(:), intent(in) :: num_edge_features
    integer, intent(in) :: num_time_steps
    integer, intent(in) :: max_vertex_degree
    integer, intent(in) :: num_outputs
    integer, optional, intent(in) :: min_vertex_degree
    class(*), optional, intent(in) :: message_activation, readout_activation
    character(*), optional, intent(in) :: kernel_initialiser
    integer, optional, intent(in) :: verbose
    type(hip_duvenaud_msgpass_layer_type) :: layer
    layer%duvenaud_msgpass_layer_type = duvenaud_msgpass_layer_type(num_vertex_features=num_vertex_features, &
         num_edge_features=num_edge_features, num_time_steps=num_time_steps, max_vertex_degree=max_vertex_degree, &
         num_outputs=num_outputs, min_vertex_degree=min_vertex_degree, message_activation=message_activation, &
         readout_activation=readout_activation, kernel_initialiser=kernel_initialiser, verbose=verbose)
  end function hip_duvenaud_layer_setup

  function hip_graph_nop_layer_setup(num_outputs, coord_dim, kernel_hidden, num_inputs, use_bias, activation, &
       kernel_initialiser, bias_initialiser, verbose) result(layer)
    !! graph_nop_layer_type's constructor (athena_graph_nop_layer.f90:143-221), same arguments
    integer, intent(in) :: num_outputs
    integer, intent(in) :: coord_dim
    integer, optional, intent(in) :: kernel_hidden
    integer, optional, intent(in) :: num_inputs
    logical, optional, intent(in) :: use_bias
    class(*), optional, intent(in) :: activation
    class(*), optional, intent(in) :: kernel_initialiser, bias_initialiser
    integer, optional, intent(in) :: verbose
    type(hip_graph_nop_layer_type) :: layer
    layer%graph_nop_layer_type = graph_nop_layer_type(num_outputs=num_outputs, coord_dim=coord_dim, &
         kernel_hidden=kernel_hidden, num_inputs=num_inputs, use_bias=use_bias, activation=activation, &
         kernel_initialiser=kernel_initialiser, bias_initialiser=bias_initialiser, verbose=verbose)
  end function hip_graph_nop_layer_setup

  ! ============================================================================================ checkpoint readers + registry
  function read_hip_kipf_msgpass_layer(unit, verbose) result(layer)
    !! read_kipf_msgpass_layer (athena_kipf_msgpass_layer.f90:624-646) returning the device-backed type; the KIPF card is
    !! parsed by the inherited read_kipf (:440-620)
    integer, intent(in) :: unit
    integer, optional, intent(in) :: verbose
    class(base_layer_type), allocatable :: layer
    integer :: verbose_
    verbose_ = 0
    if(present(verbose)) verbose_ = verbose
    allocate(layer, source = hip_kipf_msgpass_layer_type(num_time_steps = 1, num_vertex_features = [ 0, 0 ]))
    call layer%read(unit, verbose=verbose_)
  end function read_hip_kipf_msgpass_layer

  function read_hip_duvenaud_msgpass_layer(unit, verbose) result(layer)
    !! read_duvenaud_msgpass_layer (athena_duvenaud_msgpass_layer.f90:918-944), the card parsed by the inherited read_duvenaud
    integer, intent(in) :: unit
    integer, optional, intent(in) :: verbose
    class(base_layer_type), allocatable :: layer
    integer :: verbose_
    verbose_ = 0
    if(present(verbose)) verbose_ = verbose
    allocate(layer, source = hip_duvenaud_msgpass_layer_type(num_time_steps = 1, num_vertex_features = [ 1 ], &
         num_edge_features = [ 1 ], num_outputs = 1, max_vertex_degree = 1))
    call layer%read(unit, verbose=verbose_)
  end function read_hip_duvenaud_msgpass_layer

  function read_hip_graph_nop_layer(unit, verbose) result(layer)
    !! read_graph_nop_layer (athena_graph_nop_layer.f90:668-686), the card parsed by the inherited read_gno (:503-664)
    integer, intent(in) :: unit
    integer, optional, intent(in) :: verbose
    class(base_layer_type), allocatable :: layer
    integer :: verbose_
    verbose_ = 0
    if(present(verbose)) verbose_ = verbose
    allocate(layer, source = hip_graph_nop_layer_type(num_outputs = 1, coord_dim = 1, kernel_hidden = 1, num_inputs = 1))
    call layer%read(unit, verbose=verbose_)
  end function read_hip_graph_nop_layer

  function hip_msgpass_layer_readers() result(list)
    !! the entries for athena's checkpoint registry (read_layer_container, athena_container_layer.f90:57-63)
    type(read_layer_container), dimension(3) :: list
    list(1)%name = 'kipf';      list(1)%read_ptr => read_hip_kipf_msgpass_layer
    list(2)%name = 'duvenaud';  list(2)%read_ptr => read_hip_duvenaud_msgpass_layer
    list(3)%name = 'graph_nop'; list(3)%read_ptr => read_hip_graph_nop_layer
  end function hip_msgpass_layer_readers

  subroutine register_hip_msgpass_layers()
    !! Call once before network%read: the device-backed readers go IN FRONT of the registry, the stock list behind them
    !! (allocate_list_of_layer_types appends to what is there, athena_container_layer_sub.f90:107-147), so a KIPF / DUVENAUD /
    !! GRAPH_NOP card resolves to the hip_* type and every other card to its stock reader.
    type(read_layer_container), allocatable :: stock(:)
    if(allocated(list_of_layer_types))then
       if(size(list_of_layer_types) .ge. 3)then
          if(associated(list_of_layer_types(1)%read_ptr, read_hip_kipf_msgpass_layer)) return      ! already in front
       end if
       stock = list_of_layer_types
       list_of_layer_types = [ hip_msgpass_layer_readers(), stock ]
    else
       list_of_layer_types = hip_msgpass_layer_readers()
       call allocate_list_of_layer_types()
    end if
  end subroutine register_hip_msgpass_layers

  ! ============================================================================================ shared helpers
  subroutine acquire_handles(graph, handle, who)
    !! set_graph runs before EVERY forward (athena_network_sub.f90:2727-2730).  athena_mp_graph_acquire keys on (n, nnz, edge
    !! columns, a hash of every word of adj_ia / adj_ja): an unchanged sample gets its handle back for the price of the key, a
    !! different graph -- also one with the same vertex and entry counts -- gets its own.  Acquire before release, so a sample
    !! that did not change never drops to zero users in between.
    type(graph_type), dimension(:), intent(in) :: graph
    type(c_ptr), allocatable, intent(inout) :: handle(:)
    character(*), intent(in) :: who
    integer :: s
    integer(c_int) :: rc
    type(c_ptr) :: fresh

    ! first use inside a host program that never heard of the library (an unmodified athena program whose layer constructors were
    ! swapped): select a device -- ATHENA_MP_DEVICE, default 0 -- unless the host already did
    if(athena_mp_initialized() .lt. 0)then
       block
         character(len=16) :: env
         integer :: device, stat
         device = 0
         call get_environment_variable("ATHENA_MP_DEVICE", env, status=stat)
         if(stat .eq. 0) read(env, *, iostat=stat) device
         if(stat .ne. 0) device = 0
         if(athena_mp_init(int(device, c_int)) .ne. 0) call stop_program(who//": "//athena_mp_error_message())
       end block
    end if
    if(allocated(handle))then
       if(size(handle) .ne. size(graph)) call release_handles(handle)
    end if
    if(.not.allocated(handle))then
       allocate(handle(size(graph)))
       handle = c_null_ptr
    end if
    do s = 1, size(graph)
       rc = athena_mp_graph_acquire(int(graph(s)%num_vertices, c_int32_t), int(size(graph(s)%adj_ja, 2), c_int64_t), &
            graph(s)%adj_ia, graph(s)%adj_ja, int(graph(s)%num_edges, c_int32_t), fresh)
       if(rc .ne. 0) call stop_program(who//": "//athena_mp_error_message())
       if(c_associated(handle(s))) rc = athena_mp_graph_release(handle(s))
       handle(s) = fresh
    end do
  end subroutine acquire_handles

  subroutine release_handles(handle)
    type(c_ptr), allocatable, intent(inout) :: handle(:)
    integer :: s
    integer(c_int) :: rc
    if(.not.allocated(handle)) return
    do s = 1, size(handle)
       if(c_associated(handle(s))) rc = athena_mp_graph_release(handle(s))
    end do
    deallocate(handle)
  end subroutine release_handles

  pure function fused_code(actv) result(code)
    !! the epilogue code of an activation the device kernels apply themselves, -1 for anything else (attributes, other shapes)
    class(base_actv_type), intent(in) :: actv
    integer(c_int32_t) :: code
    code = -1_c_int32_t
    if(actv%apply_scaling .and. actv%scale .ne. 1._real32) return
    select case(trim(actv%name))
    case("none")
       code = ATHENA_MP_ACT_NONE
    case("relu")
       if(actv%threshold .eq. 0._real32) code = ATHENA_MP_ACT_RELU
    case("sigmoid")
       code = ATHENA_MP_ACT_SIGMOID
    case("tanh")
       code = ATHENA_MP_ACT_TANH
    end select
  end function fused_code

  subroutine flush_output(node, who)
    !! the edge of the HIP island: with athena_mp_resident_mode(1) the ops left their results in HBM; what the next (host)
    !! layer reads is materialised here.  A no-op when the mode is off.
    class(array_type), intent(in), target :: node
    character(*), intent(in) :: who
    if(athena_mp_resident_flush(c_loc(node%val)) .ne. 0) call stop_program(who//": "//athena_mp_error_message())
  end subroutine flush_output

  subroutine hand_over(dst, src, who)
    !! dst takes src's value (diffstruc's assign_and_deallocate_source) at the edge of a HIP op chain.  With the residency table on
    !! the value may live in HBM only: it is materialised FIRST, and the table forgets the source array, whose host storage is
    !! about to be released or moved -- so the hand-over is right whether diffstruc copies the values or moves the allocation.
    type(array_type), intent(inout) :: dst
    type(array_type), pointer, intent(inout) :: src
    character(*), intent(in) :: who
    call flush_output(src, who)
    if(athena_mp_resident_drop(c_loc(src%val)) .ne. 0) call stop_program(who//": "//athena_mp_error_message())
    call dst%zero_grad()
    call dst%assign_and_deallocate_source(src)
    dst%is_temporary = .false.
  end subroutine hand_over

  ! ============================================================================================ Kipf
  subroutine set_graph_hip_kipf(this, graph)
    !! set_graph_msgpass (athena_msgpass_layer_sub.f90:144-174, what the parent inherits) + the device handles
    class(hip_kipf_msgpass_layer_type), intent(inout) :: this
    type(graph_type), dimension(:), intent(in) :: graph
    call this%kipf_msgpass_layer_type%set_graph(graph)
    call acquire_handles(graph, this%handle, "set_graph (kipf)")
  end subroutine set_graph_hip_kipf

  subroutine update_message_hip_kipf(this, input)
    !! update_message_kipf, athena_kipf_msgpass_layer.f90:915-959: per sample, per time step  P = A^ X (the CSR gather
    !! kernel), Z = W_t P (the MFMA kernel), X = activation(Z) -- the activation as the GEMM's epilogue when the device
    !! applies it, athena's own apply otherwise
    class(hip_kipf_msgpass_layer_type), intent(inout), target :: this
    class(array_type), dimension(:,:), intent(in), target :: input
    integer :: s, t
    integer(c_int32_t) :: act
    type(array_type), pointer :: ptr1, ptr2, ptr3

    if(allocated(this%output))then
       if(size(this%output, 2) .ne. size(input, 2))then
          deallocate(this%output)
          allocate(this%output(1, size(input, 2)))
       end if
    else
       allocate(this%output(1, size(input, 2)))
    end if

    act = fused_code(this%activation)
    do s = 1, size(input, 2)
       ptr1 => input(1, s)
       do t = 1, this%num_time_steps
          ptr2 => kipf_propagate_hip(ptr1, this%handle(s))
          if(act .ge. 0)then
             ptr1 => matmul_hip(this%params(t), ptr2, act)
          else
             ptr3 => matmul_hip(this%params(t), ptr2)
             call flush_output(ptr3, "update_message (kipf)")            ! athena's own apply reads it on the host
             ptr1 => this%activation%apply(ptr3)
          end if
       end do
       call hand_over(this%output(1, s), ptr1, "update_message (kipf)")
    end do
  end subroutine update_message_hip_kipf

  subroutine finalise_hip_kipf(this)
    type(hip_kipf_msgpass_layer_type), intent(inout) :: this
    call release_handles(this%handle)
  end subroutine finalise_hip_kipf

  ! ============================================================================================ Duvenaud
  subroutine set_graph_hip_duvenaud(this, graph)
    !! set_graph_duvenaud (athena_duvenaud_msgpass_layer.f90:604-641: its copies and its range check) + the device handles
    class(hip_duvenaud_msgpass_layer_type), intent(inout) :: this
    type(graph_type), dimension(:), intent(in) :: graph
    call this%duvenaud_msgpass_layer_type%set_graph(graph)
    call acquire_handles(graph, this%handle, "set_graph (duvenaud)")
  end subroutine set_graph_hip_duvenaud

  subroutine update_message_hip_duvenaud(this, input)
    !! update_message_duvenaud, athena_duvenaud_msgpass_layer.f90:755-813.  Two ways through it, chosen from the activations
    !! the layer was built with:
    !!   fused        (message activation none / relu / sigmoid / tanh without attributes, readout softmax -- the defaults,
    !!                :123-124)  ONE launch per time step for update + activation + the readout's per-vertex softmax(R z), and
    !!                in the reverse pass ONE launch for both partials of the update (the hand-over between diffstruc's two
    !!                `pure` callbacks happens below the C ABI);
    !!   op-granular  every statement of the reference with the HIP op in place of the host one (any activation object).
    class(hip_duvenaud_msgpass_layer_type), intent(inout), target :: this
    class(array_type), dimension(:,:), intent(in), target :: input
    integer :: s, t, T_
    logical :: fused
    integer(c_int32_t) :: act
    type(array_type), pointer :: ptr1, ptr2, ptr3, ptr_edge, ptr_params

    T_ = this%num_time_steps
    if(allocated(this%z))then
       if(size(this%z, 2) .ne. size(input, 2))then
          deallocate(this%z)
          allocate(this%z(T_, size(input, 2)))
       end if
    else
       allocate(this%z(T_, size(input, 2)))
    end if

    act = ATHENA_MP_ACT_NONE
    if(allocated(this%activation)) act = fused_code(this%activation)
    fused = act .ge. 0 .and. allocated(this%activation_readout)
    if(fused) fused = trim(this%activation_readout%name) .eq. "softmax" .and. &
         .not.(this%activation_readout%apply_scaling .and. this%activation_readout%scale .ne. 1._real32)
    if(fused)then
       if(allocated(this%p))then
          if(size(this%p, 2) .ne. size(input, 2)) deallocate(this%p)
       end if
       if(.not.allocated(this%p)) allocate(this%p(T_, size(input, 2)))
    else if(allocated(this%p))then
       deallocate(this%p)
    end if

    do s = 1, size(input, 2)
       ptr1 => input(1, s)
       ptr_edge => input(2, s)
       do t = 1, T_
          ptr2 => duvenaud_propagate_hip(ptr1, ptr_edge, this%handle(s))
          ptr_params => this%params(t)
          if(fused)then
             ptr3 => duvenaud_update_act_readout_hip(ptr2, ptr_params, this%params(t + T_), this%p(t, s), this%handle(s), &
                  this%min_vertex_degree, this%max_vertex_degree, this%num_vertex_features(t), act)
          else
             ptr3 => duvenaud_update_hip(ptr2, ptr_params, this%handle(s), this%min_vertex_degree, this%max_vertex_degree, &
                  this%num_vertex_features(t))
             if(allocated(this%activation))then
                call flush_output(ptr3, "update_message (duvenaud)")     ! athena's own apply reads it on the host
                ptr3 => this%activation%apply(ptr3)
             end if
          end if
          call hand_over(this%z(t, s), ptr3, "update_message (duvenaud)")
          ptr1 => this%z(t, s)
          if(fused)then
             ! p(t,s) = softmax(matmul(params(T+t), z(t,s))) already holds its value: make it the node the readout sums
             this%p(t, s)%shape = [ size(this%p(t, s)%val, 1) ]
             this%p(t, s)%allocated = .true.
             this%p(t, s)%get_partial_left_val => get_partial_readout_softmax_hip_z_val
             this%p(t, s)%get_partial_right_val => get_partial_readout_softmax_hip_weight_val
             this%p(t, s)%left_operand => this%z(t, s)
             this%p(t, s)%right_operand => this%params(t + T_)
             this%p(t, s)%owns_left_operand = .false.
             this%p(t, s)%owns_right_operand = .false.
             this%p(t, s)%requires_grad = .true.
             this%p(t, s)%is_forward = this%z(t, s)%is_forward
             this%p(t, s)%is_temporary = .false.
             this%p(t, s)%operation = 'duvenaud_readout_softmax'
             call this%p(t, s)%zero_grad()
             call flush_output(this%p(t, s), "update_message (duvenaud)")      ! update_readout sums it with diffstruc's own sum
          end if
       end do
    end do
  end subroutine update_message_hip_duvenaud

  subroutine update_readout_hip_duvenaud(this)
    !! update_readout_duvenaud, athena_duvenaud_msgpass_layer.f90:817-859: out(:, s) = sum_t sum_v readout_act(R_t z_t(:, v))
    class(hip_duvenaud_msgpass_layer_type), intent(inout), target :: this
    integer :: s, t, batch_size, T_
    type(array_type), pointer :: ptr1, ptr2, ptr3, ptr_params, ptr_z

    T_ = this%num_time_steps
    batch_size = size(this%z, 2)
    call this%output(1, 1)%zero_grad()
    do t = 1, T_
       do s = 1, batch_size
          if(allocated(this%p))then
             ptr2 => this%p(t, s)                        ! fused: the softmax node update_message made in the update's launch
          else
             ptr_params => this%params(t + T_)
             ptr_z => this%z(t, s)
             ptr1 => matmul_hip(ptr_params, ptr_z)
             call flush_output(ptr1, "update_readout (duvenaud)")
             ptr2 => this%activation_readout%apply(ptr1)
          end if
          if(t .eq. 1 .and. s .eq. 1)then
             ptr3 => sum(ptr2, dim=2, new_dim_index=s, new_dim_size=batch_size)
          else
             ptr3 => ptr3 + sum(ptr2, dim=2, new_dim_index=s, new_dim_size=batch_size)
          end if
       end do
    end do
    call this%output(1, 1)%assign_and_deallocate_source(ptr3)          ! (the sum is diffstruc's own: a host value)
    this%output(1, 1)%is_temporary = .false.
  end subroutine update_readout_hip_duvenaud

  subroutine finalise_hip_duvenaud(this)
    !! the handles go back to the cache; finalise_duvenaud (:204-216) runs after this one on the parent part
    type(hip_duvenaud_msgpass_layer_type), intent(inout) :: this
    if(allocated(this%p)) deallocate(this%p)
    call release_handles(this%handle)
  end subroutine finalise_hip_duvenaud

  ! ============================================================================================ graph neural operator
  subroutine set_graph_hip_gno(this, graph)
    !! set_graph_msgpass (athena_msgpass_layer_sub.f90:144-174; graph_nop_layer_type does not override it) + device handles.
    !! The edge columns of a handle are the graph's num_edges: both directions of a pair share one column of the geometry.
    class(hip_graph_nop_layer_type), intent(inout) :: this
    type(graph_type), dimension(:), intent(in) :: graph
    call this%graph_nop_layer_type%set_graph(graph)
    call acquire_handles(graph, this%handle, "set_graph (graph_nop)")
  end subroutine set_graph_hip_gno

  subroutine update_message_hip_gno(this, input)
    !! update_message_gno, athena_graph_nop_layer.f90:690-788
    class(hip_graph_nop_layer_type), intent(inout), target :: this
    class(array_type), dimension(:,:), intent(in), target :: input
    integer :: s, F_in, F_out
    type(array_type), pointer :: ptr1, ptr2, ptr3, ptr4

    F_in  = this%num_vertex_features(0)
    F_out = this%num_vertex_features(1)
    if(size(input, 1) .lt. 2)then
       call stop_program('graph_nop layer expects vertex and edge feature inputs')
       return
    end if
    if(allocated(this%output))then
       if(any(shape(this%output) .ne. [2, size(input, 2)]))then
          deallocate(this%output)
          allocate(this%output(2, size(input, 2)))
       end if
    else
       allocate(this%output(2, size(input, 2)))
    end if

    do s = 1, size(input, 2)
       ! steps 1 + 2 (:743-758): the kernel node carries theta and the geometry, the aggregate node evaluates kappa inside the
       ! device call and never forms [F_out F_in, E]; its two partials -- features (:419-458) and, through the kernel node,
       ! theta (:480-526 with :235-325) -- come from ONE contraction, served to diffstruc's two callbacks by the pair entry point
       ptr1 => gno_kernel_hip(input(2, s), this%params(1))
       ptr2 => gno_aggregate_hip(input(1, s), ptr1, this%handle(s), this%coord_dim, this%kernel_hidden, F_in, F_out)
       ! step 3 (:761): the bypass on the MFMA kernel; step 4 (:764): combine
       ptr3 => matmul_hip(this%params(2), input(1, s))
       call flush_output(ptr2, "update_message (graph_nop)")             ! diffstruc's own + / add_bias / apply read both on the host
       call flush_output(ptr3, "update_message (graph_nop)")
       ptr4 => ptr2 + ptr3
       ! step 5 (:767-771)
       if(this%use_bias) ptr4 => add_bias(ptr4, this%params(3), dim=1, dim_act_on_shape=.true.)
       ! step 6 (:774)
       ptr4 => this%activation%apply(ptr4)

       call this%output(1, s)%zero_grad()
       call this%output(1, s)%assign_and_deallocate_source(ptr4)        ! (a host value: the activation is athena's own)
       this%output(1, s)%is_temporary = .false.

       ! the edge geometry travels on to the next layer, not differentiated (:777-785)
       if(this%output(2, s)%allocated) call this%output(2, s)%deallocate()
       call this%output(2, s)%allocate(source=input(2, s)%val)
       call this%output(2, s)%zero_grad()
       call this%output(2, s)%set_requires_grad(.false.)
       this%output(2, s)%is_temporary = .false.
    end do
  end subroutine update_message_hip_gno

  subroutine finalise_hip_gno(this)
    type(hip_graph_nop_layer_type), intent(inout) :: this
    call release_handles(this%handle)
  end subroutine finalise_hip_gno

end module athena__hip_msgpass_layers
