!! The drop-in LAYER TYPES of INTEGRATION.md section 3 for the other two message-passing layers, as ONE compilable source
!! (compile-only like hip_kipf_msgpass.f90: scripts/integration_check/run.sh puts it through the Fortran compiler against
!! athena's REAL athena__misc_types / athena__diffstruc_extd / athena__base_layer / athena__msgpass_layer module sources read in
!! place, plus compile-only stand-ins for coreutils / diffstruc / graphstruc; nothing is linked or run).
!!
!!   * hip_duvenaud_msgpass_layer_type  extends(msgpass_layer_type): what duvenaud_msgpass_layer_type is
!!       (athena_duvenaud_msgpass_layer.f90:39-80) with the device behind it -- set_graph (:604-641) + one cached device
!!       handle per sample, update_message (:755-813: T steps of propagate -> update -> activation, z(t,s) kept),
!!       update_readout (:817-859: matmul -> readout activation -> sum(dim=2, new_dim_index=s), summed over t and s).
!!       Two ways through it, chosen per layer from the activations it was given:
!!         op-granular  every statement of the reference, with duvenaud_propagate_hip / duvenaud_update_hip in place of
!!                      duvenaud_propagate / duvenaud_update (any activation object, any readout activation);
!!         fused        (message activation none / relu / sigmoid / tanh without attributes, readout softmax -- the defaults,
!!                      :123-124)  ONE launch per time step for update + activation + the readout's per-vertex softmax(R z)
!!                      (athena_mp_duvenaud_update_readout_fwd_host), and in the reverse pass ONE launch for both partials of
!!                      the update (athena_mp_duvenaud_update_bwd_pair_host: the hand-over between diffstruc's two `pure`
!!                      callbacks happens below the C ABI).
!!   * hip_graph_nop_layer_type         extends(msgpass_layer_type): graph_nop_layer_type (athena_graph_nop_layer.f90:61-76),
!!       update_message (:690-788): gno_kernel_hip + gno_aggregate_hip (the [F_out F_in, E] kernel tensor is never formed),
!!       bypass matmul, `+`, add_bias(dim=1), activation, output(2,s) = the forwarded edge geometry with requires_grad = .false.
!!       (:777-785).  Both partials of the aggregate node come from ONE contraction (athena_mp_gno_aggregate_bwd_pair_host).
!!
!! Parameters stay in this%params(:)%val, so get_params / set_params / get_gradients / set_gradients are the inherited ones
!! (athena_base_layer_sub.f90:545-691); `network%add(hip_duvenaud_msgpass_layer_type(...))` works unchanged
!! (athena_network_sub.f90:764 accepts any class(base_layer_type)).
module athena_mp__hip_layers
  use, intrinsic :: iso_c_binding
  use coreutils, only: real32, stop_program
  use graphstruc, only: graph_type
  use athena__misc_types, only: base_actv_type
  use diffstruc, only: array_type, sum, matmul, operator(+)
  use athena__msgpass_layer, only: msgpass_layer_type
  use athena__diffstruc_extd, only: add_bias
  use athena_mp_c
  use athena_mp__hip_ops, only: duvenaud_propagate_hip, duvenaud_update_hip, gno_kernel_hip, gno_aggregate_hip, &
       duvenaud_update_act_readout_hip, get_partial_readout_softmax_hip_z_val, get_partial_readout_softmax_hip_weight_val
  implicit none
  private
  public :: hip_duvenaud_msgpass_layer_type, hip_graph_nop_layer_type

  type, extends(msgpass_layer_type) :: hip_duvenaud_msgpass_layer_type
     integer :: min_vertex_degree = 1
     integer :: max_vertex_degree = 0
     class(base_actv_type), allocatable :: activation_readout
     type(array_type), allocatable, dimension(:,:) :: z
     !! (num_time_steps, batch): the vertex features after every time step, kept for the readout (as the reference, :49)
     type(array_type), allocatable, dimension(:,:) :: p
     !! (num_time_steps, batch), fused path only: the readout's per-vertex softmax(R_t z_t) -- nodes of the tape in their own
     !! right (left operand z(t,s), right operand params(T+t)), filled by the SAME launch that produces z(t,s)
     type(c_ptr), allocatable :: handle(:)
     !! one device graph per sample: reference-counted handles of the library's content-keyed cache (SURVEY F12)
   contains
     procedure, pass(this) :: set_graph => set_graph_hip_duvenaud
     procedure, pass(this) :: update_message => update_message_hip_duvenaud
     procedure, pass(this) :: update_readout => update_readout_hip_duvenaud
     procedure, pass(this) :: read => read_hip_duvenaud
     final :: finalise_hip_duvenaud
  end type hip_duvenaud_msgpass_layer_type

  type, extends(msgpass_layer_type) :: hip_graph_nop_layer_type
     integer :: coord_dim = 0
     !! dimensionality of the edge geometry (d)
     integer :: kernel_hidden = 0
     !! hidden width of the kernel MLP (H)
     type(c_ptr), allocatable :: handle(:)
   contains
     procedure, pass(this) :: set_graph => set_graph_hip_gno
     procedure, pass(this) :: update_message => update_message_hip_gno
     procedure, pass(this) :: update_readout => update_readout_hip_gno
     procedure, pass(this) :: read => read_hip_gno
     final :: finalise_hip_gno
  end type hip_graph_nop_layer_type

contains

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
    !! the epilogue code of an activation the fused kernels apply themselves, -1 for anything else (attributes, other shapes)
    class(base_actv_type), intent(in) :: actv
    integer(c_int32_t) :: code
    code = -1_c_int32_t
    if(actv%apply_scaling .and. actv%scale .ne. 1._real32) return
    select case(trim(actv%name))
    case("none", "linear")
       if(trim(actv%name) .eq. "none") code = ATHENA_MP_ACT_NONE
    case("relu")
       if(actv%threshold .eq. 0._real32) code = ATHENA_MP_ACT_RELU
    case("sigmoid")
       code = ATHENA_MP_ACT_SIGMOID
    case("tanh")
       code = ATHENA_MP_ACT_TANH
    end select
  end function fused_code

  ! ============================================================================================ Duvenaud
  subroutine set_graph_hip_duvenaud(this, graph)
    !! set_graph_duvenaud (athena_duvenaud_msgpass_layer.f90:604-641), its copies and its range check, + the device handles
    class(hip_duvenaud_msgpass_layer_type), intent(inout) :: this
    type(graph_type), dimension(:), intent(in) :: graph
    integer :: s

    if(allocated(this%graph))then
       if(size(this%graph) .ne. size(graph))then
          deallocate(this%graph)
          allocate(this%graph(size(graph)))
       end if
    else
       allocate(this%graph(size(graph)))
    end if
    do s = 1, size(graph)
       this%graph(s)%adj_ia = graph(s)%adj_ia
       this%graph(s)%adj_ja = graph(s)%adj_ja
       this%graph(s)%edge_weights = graph(s)%edge_weights
       this%graph(s)%num_edges = graph(s)%num_edges
       this%graph(s)%num_vertices = graph(s)%num_vertices
       if(any(this%graph(s)%adj_ja(1,:) .gt. this%graph(s)%num_vertices))then
          write(*,*) "Error: graph adjacency matrix has indices greater than the number of vertices", s, &
               this%graph(s)%num_vertices
          stop
       end if
    end do
    call acquire_handles(graph, this%handle, "set_graph (duvenaud)")
  end subroutine set_graph_hip_duvenaud

  subroutine update_message_hip_duvenaud(this, input)
    !! update_message_duvenaud, athena_duvenaud_msgpass_layer.f90:755-813
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

    ! which way through: the fused launch applies none / relu / sigmoid / tanh itself and produces the SOFTMAX readout's
    ! per-vertex part; anything else takes the reference's statements one device call each
    act = ATHENA_MP_ACT_NONE
    if(allocated(this%activation)) act = fused_code(this%activation)
    fused = act .ge. 0
    if(fused .and. allocated(this%activation_readout)) fused = trim(this%activation_readout%name) .eq. "softmax" .and. &
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
             if(allocated(this%activation)) ptr3 => this%activation%apply(ptr3)
          end if
          call this%z(t, s)%zero_grad()
          call this%z(t, s)%assign_and_deallocate_source(ptr3)
          this%z(t, s)%is_temporary = .false.
          ptr1 => this%z(t, s)
          if(fused)then
             ! p(t,s) = softmax(matmul(params(T+t), z(t,s))) already holds its value: make it the node the readout sums
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
             ptr1 => matmul(ptr_params, ptr_z)           ! or a matmul_hip node over athena_mp_gemm_fwd_host
             ptr2 => this%activation_readout%apply(ptr1)
          end if
          if(t .eq. 1 .and. s .eq. 1)then
             ptr3 => sum(ptr2, dim=2, new_dim_index=s, new_dim_size=batch_size)
          else
             ptr3 => ptr3 + sum(ptr2, dim=2, new_dim_index=s, new_dim_size=batch_size)
          end if
       end do
    end do
    call this%output(1, 1)%assign_and_deallocate_source(ptr3)
    this%output(1, 1)%is_temporary = .false.
    ! the edge of the HIP island (a no-op unless athena_mp_resident_mode(1) is on): what the next layer reads on the host
    if(athena_mp_resident_flush(c_loc(this%output(1, 1)%val)) .ne. 0) &
         call stop_program("update_readout: "//athena_mp_error_message())
  end subroutine update_readout_hip_duvenaud

  subroutine read_hip_duvenaud(this, unit, verbose)
    !! a DUVENAUD card reads as the stock layer does (read_duvenaud); the handles are run-time state
    class(hip_duvenaud_msgpass_layer_type), intent(inout) :: this
    integer, intent(in) :: unit
    integer, optional, intent(in) :: verbose
  end subroutine read_hip_duvenaud

  subroutine finalise_hip_duvenaud(this)
    !! finalise_duvenaud (:204-216) + the handles go back to the cache
    type(hip_duvenaud_msgpass_layer_type), intent(inout) :: this
    if(allocated(this%z)) deallocate(this%z)
    if(allocated(this%p)) deallocate(this%p)
    call release_handles(this%handle)
  end subroutine finalise_hip_duvenaud

  ! ============================================================================================ graph neural operator
  subroutine set_graph_hip_gno(this, graph)
    !! set_graph_msgpass (athena_msgpass_layer_sub.f90:144-174; graph_nop_layer_type does not override it) + device handles.
    !! The edge columns of a handle are the graph's num_edges: both directions of a pair share one column of the geometry.
    class(hip_graph_nop_layer_type), intent(inout) :: this
    type(graph_type), dimension(:), intent(in) :: graph
    integer :: s

    if(allocated(this%graph))then
       if(size(this%graph) .ne. size(graph)) deallocate(this%graph)
    end if
    if(.not.allocated(this%graph)) allocate(this%graph(size(graph)))
    do s = 1, size(graph)
       this%graph(s)%adj_ia = graph(s)%adj_ia
       this%graph(s)%adj_ja = graph(s)%adj_ja
       this%graph(s)%num_edges = graph(s)%num_edges
       this%graph(s)%num_vertices = graph(s)%num_vertices
    end do
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
       ! step 3 (:761): the bypass; step 4 (:764): combine
       ptr3 => matmul(this%params(2), input(1, s))
       ptr4 => ptr2 + ptr3
       ! step 5 (:767-771)
       if(this%use_bias) ptr4 => add_bias(ptr4, this%params(3), dim=1, dim_act_on_shape=.true.)
       ! step 6 (:774)
       ptr4 => this%activation%apply(ptr4)

       call this%output(1, s)%zero_grad()
       call this%output(1, s)%assign_and_deallocate_source(ptr4)
       this%output(1, s)%is_temporary = .false.
       if(athena_mp_resident_flush(c_loc(this%output(1, s)%val)) .ne. 0) &
            call stop_program("update_message (graph_nop): "//athena_mp_error_message())

       ! the edge geometry travels on to the next layer, not differentiated (:777-785)
       if(this%output(2, s)%allocated) call this%output(2, s)%deallocate()
       call this%output(2, s)%allocate(source=input(2, s)%val)
       call this%output(2, s)%zero_grad()
       call this%output(2, s)%set_requires_grad(.false.)
       this%output(2, s)%is_temporary = .false.
    end do
  end subroutine update_message_hip_gno

  subroutine update_readout_hip_gno(this)
    !! no graph-level readout: the layer's output is per vertex (update_readout_gno, :792-800)
    class(hip_graph_nop_layer_type), intent(inout), target :: this
  end subroutine update_readout_hip_gno

  subroutine read_hip_gno(this, unit, verbose)
    !! a GRAPH_NOP card reads as the stock layer does (read_gno); the handles are run-time state
    class(hip_graph_nop_layer_type), intent(inout) :: this
    integer, intent(in) :: unit
    integer, optional, intent(in) :: verbose
  end subroutine read_hip_gno

  subroutine finalise_hip_gno(this)
    type(hip_graph_nop_layer_type), intent(inout) :: this
    call release_handles(this%handle)
  end subroutine finalise_hip_gno

end module athena_mp__hip_layers
