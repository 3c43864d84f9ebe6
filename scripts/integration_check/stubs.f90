!! COMPILE-ONLY stand-ins for the three libraries athena depends on and this image lacks (coreutils v0.1.0, diffstruc
!! v1.2.0, graphstruc v0.2.1 -- fpm.toml:18-21).  They declare the surface that athena's msgpass modules and the
!! shim in hip_kipf_msgpass.f90 touch (inferred from call sites, SURVEY.md Appendix C) so that the compiler can check the
!! shim's syntax and interfaces against athena's real module sources.  Nothing here computes anything, nothing is
!! linked or run, and no number in this repository comes from it (the oracle is pinned without it).
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
  public :: array_type, matmul, sum, operator(+)

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
   contains
     procedure, pass(this) :: create_result
     procedure, pass(this) :: zero_grad
     procedure, pass(this) :: assign_and_deallocate_source
     procedure, pass(this) :: allocate => allocate_array
     procedure, pass(this) :: deallocate => deallocate_array
     procedure, pass(this) :: set_requires_grad
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

  interface matmul
     module procedure matmul_arrays
  end interface matmul
  interface sum
     module procedure sum_array
  end interface sum
  interface operator(+)
     module procedure add_arrays
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
  end function create_result
  subroutine zero_grad(this)
    class(array_type), intent(inout) :: this
  end subroutine zero_grad
  subroutine assign_and_deallocate_source(this, source)
    class(array_type), intent(inout) :: this
    type(array_type), pointer, intent(inout) :: source
  end subroutine assign_and_deallocate_source
  subroutine allocate_array(this, array_shape, source)
    class(array_type), intent(inout) :: this
    integer, dimension(:), intent(in), optional :: array_shape
    real(real32), dimension(:,:), intent(in), optional :: source
  end subroutine allocate_array
  subroutine deallocate_array(this)
    class(array_type), intent(inout) :: this
  end subroutine deallocate_array
  subroutine set_requires_grad(this, requires_grad)
    class(array_type), intent(inout) :: this
    logical, intent(in) :: requires_grad
  end subroutine set_requires_grad
  function sum_array(a, dim, new_dim_index, new_dim_size) result(c)
    class(array_type), intent(in), target :: a
    integer, intent(in) :: dim
    integer, intent(in), optional :: new_dim_index, new_dim_size
    type(array_type), pointer :: c
    c => a%create_result([size(a%val, 1), 1])
  end function sum_array
  function add_arrays(a, b) result(c)
    class(array_type), intent(in), target :: a, b
    type(array_type), pointer :: c
    c => a%create_result()
  end function add_arrays
  function matmul_arrays(a, b) result(c)
    class(array_type), intent(in), target :: a, b
    type(array_type), pointer :: c
    c => a%create_result([size(a%val, 1), size(b%val, 2)])
  end function matmul_arrays
end module diffstruc

module graphstruc
  use coreutils, only: real32
  implicit none
  private
  public :: graph_type
  type :: graph_type
     integer :: num_vertices = 0, num_edges = 0, num_vertex_features = 0, num_edge_features = 0
     logical :: is_sparse = .true.
     integer, allocatable :: adj_ia(:), adj_ja(:,:)
     real(real32), allocatable :: vertex_features(:,:), edge_features(:,:), edge_weights(:)
  end type graph_type
end module graphstruc
