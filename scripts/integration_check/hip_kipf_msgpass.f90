!! The drop-in shim of INTEGRATION.md sections 2-3, as ONE compilable source: what a maintainer adds to an athena
!! checkout to run the Kipf message-passing layer on libathena_mp.so.
!!   * kipf_propagate_hip      (in hip_duvenaud_gno_ops.f90, module athena_mp__hip_ops, with the other layers' ops: that file is
!!                             also linked and run on the GPU) an autodiff op with the contract of kipf_propagate
!!                             (athena_diffstruc_extd_sub_kipf.f90:7-59): result node from create_result, value from the
!!                             HIP kernel, `pure` get_partial_left_val callback (the coefficient-free scatter of :85-111)
!!   * hip_kipf_msgpass_layer_type   extends(msgpass_layer_type) (athena_msgpass_layer.f90:19-76): set_graph builds the
!!                             device handles once, update_message is update_message_kipf (athena_kipf_msgpass_layer.f90:
!!                             915-959) with the HIP op in place of kipf_propagate
!! scripts/integration_check/run.sh compiles it (never links, never runs) against athena's REAL module sources read in
!! place from the reference checkout plus compile-only stand-ins for coreutils / diffstruc / graphstruc (stubs.f90):
!! syntax and interface evidence only.
module athena_mp__hip_kipf
  use, intrinsic :: iso_c_binding
  use coreutils, only: real32, stop_program
  use graphstruc, only: graph_type
  use diffstruc, only: array_type, matmul
  use athena__msgpass_layer, only: msgpass_layer_type
  use athena_mp_c
  use athena_mp__hip_ops, only: kipf_propagate_hip
  implicit none
  private
  public :: kipf_propagate_hip, hip_kipf_msgpass_layer_type

  type, extends(msgpass_layer_type) :: hip_kipf_msgpass_layer_type
     type(c_ptr), allocatable :: handle(:)               !! one device graph per sample: a reference-counted handle of the
                                                         !! library's content-keyed cache (athena_mp_graph_acquire, SURVEY F12)
   contains
     procedure, pass(this) :: set_graph => set_graph_hip_kipf
     procedure, pass(this) :: update_message => update_message_hip_kipf
     procedure, pass(this) :: update_readout => update_readout_hip_kipf
     procedure, pass(this) :: read => read_hip_kipf
     final :: finalise_hip_kipf
  end type hip_kipf_msgpass_layer_type

contains

  subroutine set_graph_hip_kipf(this, graph)
    !! set_graph_msgpass (athena_msgpass_layer_sub.f90:144-174) + device handles, rebuilt only when the CSR changed
    class(hip_kipf_msgpass_layer_type), intent(inout) :: this
    type(graph_type), dimension(:), intent(in) :: graph
    integer :: s
    integer(c_int) :: rc
    type(c_ptr) :: fresh

    ! the parent type is abstract, so its set_graph cannot be called through the parent component
    ! (this%msgpass_layer_type%set_graph is illegal): the copies of set_graph_msgpass are restated here
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
    if(allocated(this%handle))then
       if(size(this%handle) .ne. size(graph)) call release_handles(this)
    end if
    if(.not.allocated(this%handle))then
       allocate(this%handle(size(graph)))
       this%handle = c_null_ptr
    end if
    ! set_graph runs before EVERY forward (athena_network_sub.f90:2727-2730).  athena_mp_graph_acquire keys on
    ! (n, nnz, edge columns, a hash of the CONTENT of adj_ia / adj_ja): an unchanged sample gets its handle back for the
    ! price of the key, a different graph -- also one with the same vertex and entry counts -- gets its own.  Acquire
    ! before release, so a sample that did not change never drops to zero users in between.
    do s = 1, size(graph)
       rc = athena_mp_graph_acquire(int(graph(s)%num_vertices, c_int32_t), int(size(graph(s)%adj_ja, 2), c_int64_t), &
            graph(s)%adj_ia, graph(s)%adj_ja, int(graph(s)%num_edges, c_int32_t), fresh)
       if(rc .ne. 0) call stop_program("set_graph: "//athena_mp_error_message())
       if(c_associated(this%handle(s))) rc = athena_mp_graph_release(this%handle(s))
       this%handle(s) = fresh
    end do
  end subroutine set_graph_hip_kipf

  subroutine update_message_hip_kipf(this, input)
    !! update_message_kipf, athena_kipf_msgpass_layer.f90:915-959, with the HIP op
    class(hip_kipf_msgpass_layer_type), intent(inout), target :: this
    class(array_type), dimension(:,:), intent(in), target :: input
    integer :: s, t
    type(array_type), pointer :: ptr1, ptr2, ptr3

    do s = 1, size(input, 2)
       ptr2 => kipf_propagate_hip(input(1, s), this%handle(s))
       ptr3 => matmul(this%params(1), ptr2)
       do t = 2, this%num_time_steps
          ptr1 => ptr3                                  ! the activation's apply(ptr3) sits here in the full layer
          ptr2 => kipf_propagate_hip(ptr1, this%handle(s))
          ptr3 => matmul(this%params(t), ptr2)
       end do
       call this%output(1, s)%zero_grad()
       call this%output(1, s)%assign_and_deallocate_source(ptr3)
       this%output(1, s)%is_temporary = .false.
       ! the edge of the HIP island: with athena_mp_resident_mode(1) the ops above left their results in HBM (P never
       ! crossed PCIe); what the next layer reads on the host is materialised here.  A no-op when the mode is off.
       if(athena_mp_resident_flush(c_loc(this%output(1, s)%val)) .ne. 0) &
            call stop_program("update_message: "//athena_mp_error_message())
    end do
  end subroutine update_message_hip_kipf

  subroutine update_readout_hip_kipf(this)
    !! the Kipf layer has no readout (athena_kipf_msgpass_layer.f90: update_readout is empty)
    class(hip_kipf_msgpass_layer_type), intent(inout), target :: this
  end subroutine update_readout_hip_kipf

  subroutine read_hip_kipf(this, unit, verbose)
    !! a KIPF card reads as the stock layer does (read_kipf); the handles are run-time state
    class(hip_kipf_msgpass_layer_type), intent(inout) :: this
    integer, intent(in) :: unit
    integer, optional, intent(in) :: verbose
  end subroutine read_hip_kipf

  subroutine release_handles(this)
    type(hip_kipf_msgpass_layer_type), intent(inout) :: this
    integer :: s
    integer(c_int) :: rc
    if(.not.allocated(this%handle)) return
    do s = 1, size(this%handle)
       if(c_associated(this%handle(s))) rc = athena_mp_graph_release(this%handle(s))
    end do
    deallocate(this%handle)
  end subroutine release_handles

  subroutine finalise_hip_kipf(this)
    type(hip_kipf_msgpass_layer_type), intent(inout) :: this
    call release_handles(this)
  end subroutine finalise_hip_kipf

end module athena_mp__hip_kipf
