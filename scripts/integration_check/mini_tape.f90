!! A MINIMAL, WORKING tape with diffstruc's surface (SURVEY.md Appendix C) -- test harness only, so that the autodiff ops of
!! hip_duvenaud_gno_ops.f90 can be LINKED AND RUN on the GPU from Fortran (run_ops.f90): result nodes, operand links, the `pure`
!! get_partial_*_val callbacks, and a grad_reverse that asks every node for its left partial and then for its right partial,
!! one at a time, exactly the protocol the real diffstruc v1.2.0 drives (it is not in this image).  It is NOT diffstruc, computes
!! nothing of the reference's, and no number of this repository's parity claims comes from it: it exists to execute OUR shim.
module coreutils
  use, intrinsic :: iso_fortran_env, only: real32
  implicit none
  private
  public :: real32, stop_program
contains
  subroutine stop_program(msg)
    character(*), intent(in) :: msg
    write(0, *) msg
    error stop 1
  end subroutine stop_program
end module coreutils

module diffstruc
  use coreutils, only: real32
  implicit none
  private
  public :: array_type

  type :: array_type
     real(real32), allocatable :: val(:,:)
     integer, allocatable :: shape(:)
     integer, allocatable :: indices(:)
     integer, allocatable :: adj_ja(:,:)
     integer :: rank = 2
     logical :: allocated = .false.
     logical :: requires_grad = .false., is_forward = .false., is_temporary = .true.
     logical :: is_sample_dependent = .true., fix_pointer = .false.
     logical :: owns_left_operand = .false., owns_right_operand = .false.
     character(len=64) :: operation = ""
     class(array_type), pointer :: left_operand => null(), right_operand => null()
     type(array_type), pointer :: grad => null()
     procedure(partial_fn), pass(this), pointer :: get_partial_left => null(), get_partial_right => null()
     procedure(partial_val), pass(this), pointer :: get_partial_left_val => null(), get_partial_right_val => null()
     integer :: partial_calls = 0          !! harness bookkeeping: callbacks grad_reverse made on this node
   contains
     procedure, pass(this) :: create_result
     procedure, pass(this) :: zero_grad
     procedure, pass(this) :: grad_reverse
  end type array_type

  abstract interface
     function partial_fn(this, upstream_grad) result(output)
       import :: array_type
       class(array_type), intent(inout) :: this
       type(array_type), intent(in) :: upstream_grad
       type(array_type) :: output
     end function partial_fn
     pure subroutine partial_val(this, upstream_grad, output)
       import :: array_type, real32
       class(array_type), intent(in) :: this
       real(real32), dimension(:,:), intent(in) :: upstream_grad
       real(real32), dimension(:,:), intent(out) :: output
     end subroutine partial_val
  end interface
contains
  function create_result(this, array_shape) result(c)
    class(array_type), intent(in) :: this
    integer, dimension(:), intent(in), optional :: array_shape
    type(array_type), pointer :: c
    allocate(c)
    if(present(array_shape))then
       allocate(c%val(array_shape(1), array_shape(2)))
    else
       allocate(c%val, mold=this%val)
    end if
    c%allocated = .true.
  end function create_result

  subroutine zero_grad(this)
    class(array_type), intent(inout) :: this
    if(associated(this%grad))then
       if(allocated(this%grad%val)) this%grad%val = 0._real32
    end if
  end subroutine zero_grad

  recursive subroutine grad_reverse(this, upstream)
    !! accumulate `upstream` into this node's gradient, then hand each operand that wants one its partial: the LEFT callback
    !! first, then the RIGHT one -- two separate calls with the same upstream, nothing kept between them on this side
    class(array_type), intent(inout) :: this
    real(real32), dimension(:,:), intent(in) :: upstream
    real(real32), allocatable :: partial(:,:)

    if(.not. associated(this%grad))then
       allocate(this%grad)
       allocate(this%grad%val, mold=this%val)
       this%grad%val = 0._real32
    end if
    this%grad%val = this%grad%val + upstream
    if(associated(this%left_operand) .and. associated(this%get_partial_left_val))then
       if(this%left_operand%requires_grad)then
          allocate(partial, mold=this%left_operand%val)
          call this%get_partial_left_val(upstream, partial)
          this%partial_calls = this%partial_calls + 1
          call this%left_operand%grad_reverse(partial)
          deallocate(partial)
       end if
    end if
    if(associated(this%right_operand) .and. associated(this%get_partial_right_val))then
       if(this%right_operand%requires_grad)then
          allocate(partial, mold=this%right_operand%val)
          call this%get_partial_right_val(upstream, partial)
          this%partial_calls = this%partial_calls + 1
          call this%right_operand%grad_reverse(partial)
          deallocate(partial)
       end if
    end if
  end subroutine grad_reverse
end module diffstruc
