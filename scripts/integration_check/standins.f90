!! ONE stand-in for the three libraries athena depends on and this image lacks (coreutils v0.1.0, diffstruc v1.2.0,
!! graphstruc v0.2.1 -- fpm.toml:18-21), doing BOTH jobs of the integration check:
!!   * it declares the surface athena's own modules touch (inferred from their call sites, SURVEY.md Appendix C), so that
!!     athena's REAL sources -- ALL 99 files of src/athena: misc_types, diffstruc_extd (+ submodules), the activations, initialisers,
!!     losses, optimisers, every layer module, base_layer / msgpass_layer (+ submodules), the container registry, network_type
!!     (+ athena_network_sub.f90, athena_onnx_write_sub.f90) -- compile against it, read in place from the reference checkout;
!!   * it is a WORKING tape: result nodes, operand links, the `pure` get_partial_*_val callback protocol, the element-wise /
!!     matmul / sum / mean nodes athena's losses and host layers build, and a grad_reverse that accumulates every node's gradient
!!     before asking it ONCE for its left and then its right partial -- and a working directed graph for network_type's graph of
!!     layers -- so that the hip_* layer types can be built with their constructors and driven by athena's own forward_msgpass /
!!     get_gradients (run_layers.f90), by athena's own network%train / test / print / read beside networks of athena's stock
!!     layers (run_network.f90), and the ops alone by run_ops.f90.
!! It is NOT diffstruc / graphstruc / coreutils: it is test-harness material written from the call sites, it computes
!! nothing of the reference's message-passing path, and no parity claim of this repository rests on it (the oracle is
!! pinned without it).  What it cannot tell: anything about the real diffstruc's traversal order, memory ownership
!! (`is_temporary` / `owns_*_operand` are carried, never acted on: temporaries leak here) or forward-mode products.
module coreutils
  use, intrinsic :: iso_fortran_env, only: real32
  implicit none
  private
  public :: real32, pi, stop_program, print_warning, to_lower, to_upper, to_camel_case, icount, grep

  real(real32), parameter :: pi = 3.14159265358979323846_real32
contains
  subroutine stop_program(msg, exit_code)
    character(*), intent(in) :: msg
    integer, intent(in), optional :: exit_code
    write(0, '(A)') "ERROR: "//trim(msg)
    error stop 1
  end subroutine stop_program

  subroutine print_warning(msg)
    character(*), intent(in) :: msg
    write(0, '(A)') "WARNING: "//trim(msg)
  end subroutine print_warning

  pure function to_lower(s) result(r)
    character(*), intent(in) :: s
    character(len=len(s)) :: r
    integer :: i, c
    r = s
    do i = 1, len(s)
       c = iachar(s(i:i))
       if(c .ge. iachar('A') .and. c .le. iachar('Z')) r(i:i) = achar(c + 32)
    end do
  end function to_lower

  pure function to_upper(s) result(r)
    character(*), intent(in) :: s
    character(len=len(s)) :: r
    integer :: i, c
    r = s
    do i = 1, len(s)
       c = iachar(s(i:i))
       if(c .ge. iachar('a') .and. c .le. iachar('z')) r(i:i) = achar(c - 32)
    end do
  end function to_upper

  pure function to_camel_case(s, capitalise_first_letter) result(r)
    !! snake_case -> camelCase (underscores dropped, the letter after one capitalised)
    character(*), intent(in) :: s
    logical, intent(in), optional :: capitalise_first_letter
    character(len=:), allocatable :: r
    integer :: i
    logical :: up
    r = ""
    up = .false.
    if(present(capitalise_first_letter)) up = capitalise_first_letter
    do i = 1, len_trim(s)
       if(s(i:i) .eq. '_')then
          up = .true.
       else if(up)then
          r = r//to_upper(s(i:i)); up = .false.
       else
          r = r//s(i:i)
       end if
    end do
  end function to_camel_case

  pure integer function icount(line, fs) result(n)
    !! number of fields of `line` separated by blanks (or by the characters of `fs`)
    character(*), intent(in) :: line
    character(*), intent(in), optional :: fs
    integer :: i
    logical :: in_field, sep
    n = 0
    in_field = .false.
    do i = 1, len_trim(line)
       if(present(fs))then
          sep = index(fs, line(i:i)) .ne. 0
       else
          sep = line(i:i) .eq. ' ' .or. line(i:i) .eq. achar(9)
       end if
       if(sep)then
          in_field = .false.
       else if(.not. in_field)then
          in_field = .true.
          n = n + 1
       end if
    end do
  end function icount

  subroutine grep(unit, input, lstart, lline, success)
    !! position `unit` on the line after the first one that contains `input` (lline: after the first line that IS `input`)
    integer, intent(in) :: unit
    character(*), intent(in) :: input
    logical, intent(in), optional :: lstart, lline
    logical, intent(out), optional :: success
    character(1024) :: buffer
    integer :: stat
    logical :: whole
    whole = .false.
    if(present(lline)) whole = lline
    if(present(lstart))then
       if(lstart) rewind(unit)
    else
       rewind(unit)
    end if
    if(present(success)) success = .false.
    do
       read(unit, '(A)', iostat=stat) buffer
       if(stat .ne. 0) return
       if(whole)then
          if(trim(adjustl(buffer)) .eq. trim(input)) exit
       else
          if(index(buffer, trim(input)) .ne. 0) exit
       end if
    end do
    if(present(success)) success = .true.
  end subroutine grep
end module coreutils


module diffstruc
  use coreutils, only: real32, stop_program
  implicit none
  private
  public :: array_type
  public :: operator(+), operator(-), operator(*), operator(/), operator(.gt.), operator(.le.)
  public :: matmul, sum, exp, tanh, sigmoid, max, merge, gaussian, concat, mean, squared, log, abs, sign, pack, reshape
  public :: weighted_sum, tape_callbacks_made

  !! element-wise operation codes of the nodes this module makes itself
  integer, parameter :: OP_ADD = 1, OP_SUB = 2, OP_MUL = 3, OP_DIV = 4, OP_POW = 5, OP_EXP = 6, OP_TANH = 7, OP_SIGMOID = 8, &
       OP_MAX = 9, OP_MERGE = 10, OP_GAUSS = 11, OP_NEG = 12, OP_SQUARED = 13, OP_LOG = 14, OP_ABS = 15, OP_COPY = 16

  type :: array_type
     real(real32), allocatable :: val(:,:)
     !! [product(shape), samples]; the last entry of the array_shape handed to allocate / create_result is the column count
     integer, allocatable :: shape(:)
     integer, allocatable :: indices(:)
     integer, allocatable :: adj_ja(:,:)
     logical, allocatable :: mask(:,:)
     integer :: rank = 1
     integer :: size = 0
     integer :: id = 0
     logical :: allocated = .false.
     logical :: requires_grad = .false., is_forward = .false., is_temporary = .true.
     logical :: is_sample_dependent = .true., is_scalar = .false., fix_pointer = .false.
     logical :: owns_left_operand = .false., owns_right_operand = .false.
     character(len=64) :: operation = ""
     class(array_type), pointer :: left_operand => null(), right_operand => null()
     type(array_type), pointer :: grad => null()
     procedure(partial_fn), pass(this), pointer :: get_partial_left => null(), get_partial_right => null()
     procedure(partial_val), pass(this), pointer :: get_partial_left_val => null(), get_partial_right_val => null()
     procedure(partial_val_sum), pass(this), pointer :: get_partial_left_val_sum => null(), get_partial_right_val_sum => null()
     ! ---- the stand-in's own bookkeeping (not part of the surface athena touches)
     integer :: op = 0                 !! element-wise op code of nodes made here
     real(real32) :: scalar = 0._real32 !! the real operand of array-op-real nodes
     logical :: scalar_left = .false.  !! ... and whether it stood on the left (real - array, real / array)
     integer :: epoch = 0              !! grad_reverse visit mark
     integer :: pending = 0            !! consumers of this node that have not handed their partial over yet
     integer :: partial_calls = 0      !! callbacks grad_reverse made on this node (harness checks)
   contains
     procedure, pass(this) :: create_result
     procedure, pass(this) :: allocate => allocate_array
     procedure, pass(this) :: deallocate => deallocate_array
     procedure, pass(this) :: set => set_array
     procedure, pass(this) :: extract => extract_array
     procedure, pass(this) :: zero_grad
     procedure, pass(this) :: set_requires_grad
     procedure, pass(this) :: assign_and_deallocate_source
     procedure, pass(this) :: assign_shallow
     procedure, pass(this) :: nullify_graph
     procedure, pass(this) :: grad_reverse
     procedure, pass(this) :: grad_reverse_from
     procedure, pass(this) :: assign_array
     generic :: assignment(=) => assign_array
     ! array-op-array, array-op-real and real-op-array are type-bound: athena uses them where it never imports the operator
     ! (`/` and `**` in athena_diffstruc_extd_sub.f90:470, real * array in athena_diffstruc_extd_loss.f90:56).  The module-level
     ! generics that `use diffstruc, only: operator(..)` names hold a form of their own (a plain rank-2 real on the left)
     procedure, pass(a) :: add_aa, add_ar, sub_aa, sub_ar, mul_aa, mul_ar, div_aa, div_ar, pow_ar, gt_ar, le_ar, neg_a
     procedure, pass(b) :: add_ra, sub_ra, mul_ra, div_ra, gt_ra, le_ra
     generic :: operator(+) => add_aa, add_ar, add_ra
     generic :: operator(-) => sub_aa, sub_ar, sub_ra, neg_a
     generic :: operator(*) => mul_aa, mul_ar, mul_ra
     generic :: operator(/) => div_aa, div_ar, div_ra
     generic :: operator(**) => pow_ar
     generic :: operator(.gt.) => gt_ar, gt_ra
     generic :: operator(.le.) => le_ar, le_ra
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
     pure subroutine partial_val_sum(this, upstream_grad, output)
       import :: array_type, real32
       class(array_type), intent(in) :: this
       real(real32), dimension(:,:), intent(in) :: upstream_grad
       real(real32), dimension(:), intent(out) :: output
     end subroutine partial_val_sum
  end interface

  interface operator(+)
     module procedure add_va
  end interface
  interface operator(-)
     module procedure sub_va
  end interface
  interface operator(*)
     module procedure mul_va
  end interface
  interface operator(/)
     module procedure div_va
  end interface
  interface operator(.gt.)
     module procedure gt_va
  end interface
  interface operator(.le.)
     module procedure le_va
  end interface
  interface matmul
     module procedure matmul_arrays
  end interface
  interface sum
     module procedure sum_array
  end interface
  interface exp
     module procedure exp_array
  end interface
  interface tanh
     module procedure tanh_array
  end interface
  interface sigmoid
     module procedure sigmoid_array
  end interface
  interface max
     module procedure max_ar, max_aa
  end interface
  interface merge
     module procedure merge_arrays, merge_array_real
  end interface
  interface gaussian
     module procedure gaussian_array
  end interface
  interface concat
     module procedure concat_arrays
  end interface
  interface mean
     module procedure mean_array
  end interface
  interface squared
     module procedure squared_array
  end interface
  interface log
     module procedure log_array
  end interface
  interface abs
     module procedure abs_array
  end interface
  interface sign
     module procedure sign_real_array
  end interface
  interface pack
     module procedure pack_array
  end interface
  interface reshape
     module procedure reshape_array
  end interface

  integer, save :: current_epoch = 0
  integer, save :: callbacks_made = 0

contains

  ! ====================================================================================== storage
  function create_result(this, array_shape) result(c)
    !! a fresh temporary node shaped like `this`, or as array_shape says (last entry = columns, the rest = %shape)
    class(array_type), intent(in) :: this
    integer, dimension(:), intent(in), optional :: array_shape
    type(array_type), pointer :: c
    allocate(c)
    if(present(array_shape))then
       call c%allocate(array_shape=array_shape)
    else
       allocate(c%val(size(this%val, 1), size(this%val, 2)))
       if(allocated(this%shape))then
          c%shape = this%shape
       else
          c%shape = [size(this%val, 1)]
       end if
       c%rank = size(c%shape)
       c%size = size(c%val, 1)
       c%allocated = .true.
    end if
    c%is_sample_dependent = this%is_sample_dependent
    c%is_temporary = .true.
  end function create_result

  subroutine allocate_array(this, array_shape, source)
    class(array_type), intent(inout) :: this
    integer, dimension(:), intent(in), optional :: array_shape
    class(*), dimension(..), intent(in), optional :: source
    integer :: n
    if(allocated(this%val)) deallocate(this%val)
    if(allocated(this%shape)) deallocate(this%shape)
    if(present(array_shape))then
       n = size(array_shape)
       if(n .lt. 2) call stop_program("array_type%allocate: array_shape needs at least [rows, columns]")
       this%shape = array_shape(1:n-1)
       allocate(this%val(product(this%shape), array_shape(n)))
       this%val = 0._real32
    end if
    if(present(source))then
       select rank(source)
       rank(0)
          select type(source)
          type is(real(real32))
             if(.not.allocated(this%val)) call stop_program("array_type%allocate: scalar source without array_shape")
             this%val = source
          class is(array_type)
             this%val = source%val
             if(allocated(source%shape)) this%shape = source%shape
          class default
             call stop_program("array_type%allocate: unsupported scalar source")
          end select
       rank(1)
          select type(source)
          type is(real(real32))
             if(.not.allocated(this%val)) allocate(this%val(size(source), 1))
             this%val(:, 1) = source
          class default
             call stop_program("array_type%allocate: unsupported rank-1 source")
          end select
       rank(2)
          select type(source)
          type is(real(real32))
             if(allocated(this%val))then
                if(any(shape(this%val) .ne. shape(source))) deallocate(this%val)
             end if
             if(.not.allocated(this%val)) allocate(this%val(size(source, 1), size(source, 2)))
             this%val = source
          class default
             call stop_program("array_type%allocate: unsupported rank-2 source")
          end select
       rank default
          call stop_program("array_type%allocate: unsupported source rank")
       end select
    end if
    if(.not.allocated(this%val)) call stop_program("array_type%allocate: neither array_shape nor source")
    if(.not.allocated(this%shape)) this%shape = [size(this%val, 1)]
    this%rank = size(this%shape)
    this%size = size(this%val, 1)
    this%allocated = .true.
  end subroutine allocate_array

  subroutine deallocate_array(this, keep_shape)
    class(array_type), intent(inout) :: this
    logical, intent(in), optional :: keep_shape
    if(allocated(this%val)) deallocate(this%val)
    if(allocated(this%shape)) deallocate(this%shape)
    if(allocated(this%indices)) deallocate(this%indices)
    if(allocated(this%adj_ja)) deallocate(this%adj_ja)
    if(allocated(this%mask)) deallocate(this%mask)
    call this%nullify_graph()
    this%allocated = .false.
  end subroutine deallocate_array

  pure subroutine set_array(this, input)
    class(array_type), intent(inout) :: this
    real(real32), dimension(..), intent(in) :: input
    select rank(input)
    rank(1)
       if(.not.allocated(this%val)) allocate(this%val(size(input), 1))
       this%val(:, 1) = input
    rank(2)
       if(allocated(this%val))then
          if(any(shape(this%val) .ne. shape(input))) deallocate(this%val)
       end if
       if(.not.allocated(this%val)) allocate(this%val(size(input, 1), size(input, 2)))
       this%val = input
    rank default
       error stop "array_type%set: rank not provided by the stand-in"
    end select
    if(.not.allocated(this%shape)) this%shape = [size(this%val, 1)]
    this%rank = size(this%shape)
    this%size = size(this%val, 1)
    this%allocated = .true.
  end subroutine set_array

  subroutine extract_array(this, output)
    class(array_type), intent(in) :: this
    real(real32), allocatable, dimension(..), intent(out) :: output
    select rank(output)
    rank(1)
       output = reshape(this%val, [size(this%val)])
    rank(2)
       output = this%val
    rank default
       call stop_program("array_type%extract: rank not provided by the stand-in")
    end select
  end subroutine extract_array

  subroutine zero_grad(this)
    class(array_type), intent(inout) :: this
    if(associated(this%grad))then
       if(allocated(this%grad%val)) this%grad%val = 0._real32
    end if
  end subroutine zero_grad

  pure subroutine set_requires_grad(this, requires_grad)
    class(array_type), intent(inout) :: this
    logical, intent(in) :: requires_grad
    this%requires_grad = requires_grad
  end subroutine set_requires_grad

  subroutine copy_node(this, rhs)
    !! values deep, graph links and callbacks shallow
    class(array_type), intent(inout) :: this
    class(array_type), intent(in) :: rhs
    if(allocated(rhs%val))then
       this%val = rhs%val
    else if(allocated(this%val))then
       deallocate(this%val)
    end if
    if(allocated(rhs%shape))then
       this%shape = rhs%shape
    else if(allocated(this%shape))then
       deallocate(this%shape)
    end if
    if(allocated(rhs%indices))then
       this%indices = rhs%indices
    else if(allocated(this%indices))then
       deallocate(this%indices)
    end if
    if(allocated(rhs%adj_ja))then
       this%adj_ja = rhs%adj_ja
    else if(allocated(this%adj_ja))then
       deallocate(this%adj_ja)
    end if
    if(allocated(rhs%mask))then
       this%mask = rhs%mask
    else if(allocated(this%mask))then
       deallocate(this%mask)
    end if
    this%rank = rhs%rank; this%size = rhs%size; this%id = rhs%id
    this%allocated = rhs%allocated
    this%requires_grad = rhs%requires_grad; this%is_forward = rhs%is_forward
    this%is_sample_dependent = rhs%is_sample_dependent; this%is_scalar = rhs%is_scalar
    this%owns_left_operand = rhs%owns_left_operand; this%owns_right_operand = rhs%owns_right_operand
    this%operation = rhs%operation
    this%left_operand => rhs%left_operand
    this%right_operand => rhs%right_operand
    this%get_partial_left => rhs%get_partial_left
    this%get_partial_right => rhs%get_partial_right
    this%get_partial_left_val => rhs%get_partial_left_val
    this%get_partial_right_val => rhs%get_partial_right_val
    this%get_partial_left_val_sum => rhs%get_partial_left_val_sum
    this%get_partial_right_val_sum => rhs%get_partial_right_val_sum
    this%op = rhs%op; this%scalar = rhs%scalar; this%scalar_left = rhs%scalar_left
    this%epoch = 0; this%pending = 0; this%partial_calls = 0
  end subroutine copy_node

  subroutine assign_array(this, rhs)
    class(array_type), intent(inout) :: this
    class(array_type), intent(in) :: rhs
    call copy_node(this, rhs)
    this%is_temporary = rhs%is_temporary
  end subroutine assign_array

  subroutine assign_and_deallocate_source(this, source)
    !! `this` becomes the node `source` points at (value, links, callbacks); the temporary is freed
    class(array_type), intent(inout) :: this
    type(array_type), pointer, intent(inout) :: source
    call copy_node(this, source)
    if(associated(source%grad))then
       if(associated(this%grad)) deallocate(this%grad)
       this%grad => source%grad
       nullify(source%grad)
    end if
    deallocate(source)
    nullify(source)
  end subroutine assign_and_deallocate_source

  subroutine assign_shallow(this, source)
    !! `this` takes over the node `source` is (value, links, callbacks); the source object stays where it is
    class(array_type), intent(inout) :: this
    class(array_type), intent(in) :: source
    call copy_node(this, source)
    this%is_temporary = source%is_temporary
  end subroutine assign_shallow

  subroutine nullify_graph(this)
    class(array_type), intent(inout) :: this
    nullify(this%left_operand)
    nullify(this%right_operand)
    this%owns_left_operand = .false.
    this%owns_right_operand = .false.
  end subroutine nullify_graph

  ! ====================================================================================== the reverse pass
  recursive subroutine count_consumers(node)
    !! first visit of this pass: fresh gradient for every interior node, consumer counts along the edges that carry one
    class(array_type), intent(inout) :: node
    logical :: interior
    node%epoch = current_epoch
    node%pending = 0
    node%partial_calls = 0
    interior = associated(node%left_operand) .or. associated(node%right_operand)
    if(.not. associated(node%grad))then
       allocate(node%grad)
       allocate(node%grad%val(size(node%val, 1), size(node%val, 2)))
       node%grad%val = 0._real32
       if(allocated(node%shape)) node%grad%shape = node%shape
       node%grad%allocated = .true.
    else if(interior)then
       if(any(shape(node%grad%val) .ne. shape(node%val)))then
          deallocate(node%grad%val)
          allocate(node%grad%val(size(node%val, 1), size(node%val, 2)))
       end if
       node%grad%val = 0._real32
    end if
    if(associated(node%left_operand) .and. associated(node%get_partial_left_val))then
       if(node%left_operand%requires_grad)then
          if(node%left_operand%epoch .ne. current_epoch) call count_consumers(node%left_operand)
          node%left_operand%pending = node%left_operand%pending + 1
       end if
    end if
    if(associated(node%right_operand) .and. associated(node%get_partial_right_val))then
       if(node%right_operand%requires_grad)then
          if(node%right_operand%epoch .ne. current_epoch) call count_consumers(node%right_operand)
          node%right_operand%pending = node%right_operand%pending + 1
       end if
    end if
  end subroutine count_consumers

  recursive subroutine hand_down(node)
    !! node%grad is complete: ask the node for its LEFT partial, then for its RIGHT one -- two separate `pure` callbacks with
    !! the same upstream gradient, nothing kept between them on this side -- and descend into an operand once its last
    !! consumer has delivered
    class(array_type), intent(inout) :: node
    real(real32), allocatable :: partial(:,:)
    if(associated(node%left_operand) .and. associated(node%get_partial_left_val))then
       if(node%left_operand%requires_grad)then
          allocate(partial(size(node%left_operand%val, 1), size(node%left_operand%val, 2)))
          call node%get_partial_left_val(node%grad%val, partial)
          node%partial_calls = node%partial_calls + 1
          callbacks_made = callbacks_made + 1
          node%left_operand%grad%val = node%left_operand%grad%val + partial
          deallocate(partial)
          node%left_operand%pending = node%left_operand%pending - 1
          if(node%left_operand%pending .eq. 0) call hand_down(node%left_operand)
       end if
    end if
    if(associated(node%right_operand) .and. associated(node%get_partial_right_val))then
       if(node%right_operand%requires_grad)then
          allocate(partial(size(node%right_operand%val, 1), size(node%right_operand%val, 2)))
          call node%get_partial_right_val(node%grad%val, partial)
          node%partial_calls = node%partial_calls + 1
          callbacks_made = callbacks_made + 1
          node%right_operand%grad%val = node%right_operand%grad%val + partial
          deallocate(partial)
          node%right_operand%pending = node%right_operand%pending - 1
          if(node%right_operand%pending .eq. 0) call hand_down(node%right_operand)
       end if
    end if
  end subroutine hand_down

  subroutine grad_reverse(this, reset_graph)
    !! d(sum of this) / d(every leaf below): the seed is ones, as for a scalar loss
    class(array_type), intent(inout) :: this
    logical, intent(in), optional :: reset_graph
    real(real32), allocatable :: ones(:,:)
    allocate(ones(size(this%val, 1), size(this%val, 2)))
    ones = 1._real32
    call this%grad_reverse_from(ones)
  end subroutine grad_reverse

  subroutine grad_reverse_from(this, upstream)
    !! harness entry: the reverse pass seeded with a given upstream gradient
    class(array_type), intent(inout) :: this
    real(real32), dimension(:,:), intent(in) :: upstream
    current_epoch = current_epoch + 1
    call count_consumers(this)
    this%grad%val = this%grad%val + upstream
    call hand_down(this)
  end subroutine grad_reverse_from

  integer function tape_callbacks_made() result(n)
    n = callbacks_made
  end function tape_callbacks_made

  function partial_left_from_val(this, upstream_grad) result(output)
    !! the function forms (higher-order / forward-mode products in the real library) answer through the value forms here
    class(array_type), intent(inout) :: this
    type(array_type), intent(in) :: upstream_grad
    type(array_type) :: output
    allocate(output%val(size(this%left_operand%val, 1), size(this%left_operand%val, 2)))
    call this%get_partial_left_val(upstream_grad%val, output%val)
    output%allocated = .true.
  end function partial_left_from_val

  function partial_right_from_val(this, upstream_grad) result(output)
    class(array_type), intent(inout) :: this
    type(array_type), intent(in) :: upstream_grad
    type(array_type) :: output
    allocate(output%val(size(this%right_operand%val, 1), size(this%right_operand%val, 2)))
    call this%get_partial_right_val(upstream_grad%val, output%val)
    output%allocated = .true.
  end function partial_right_from_val

  ! ====================================================================================== element-wise nodes
  subroutine link_binary(c, a, b, op)
    type(array_type), intent(inout) :: c
    class(array_type), intent(in), target :: a, b
    integer, intent(in) :: op
    c%op = op
    c%get_partial_left => partial_left_from_val
    c%get_partial_right => partial_right_from_val
    c%get_partial_left_val => ew_left_val
    c%get_partial_right_val => ew_right_val
    c%left_operand => a
    c%right_operand => b
    c%owns_left_operand = a%is_temporary
    c%owns_right_operand = b%is_temporary
    c%requires_grad = a%requires_grad .or. b%requires_grad
    c%is_forward = a%is_forward .or. b%is_forward
  end subroutine link_binary

  subroutine link_unary(c, a, op, scalar, scalar_left)
    type(array_type), intent(inout) :: c
    class(array_type), intent(in), target :: a
    integer, intent(in) :: op
    real(real32), intent(in), optional :: scalar
    logical, intent(in), optional :: scalar_left
    c%op = op
    if(present(scalar)) c%scalar = scalar
    if(present(scalar_left)) c%scalar_left = scalar_left
    c%get_partial_left => partial_left_from_val
    c%get_partial_left_val => ew_left_val
    c%left_operand => a
    c%owns_left_operand = a%is_temporary
    c%requires_grad = a%requires_grad
    c%is_forward = a%is_forward
  end subroutine link_unary

  subroutine same_shape(a, b, what)
    class(array_type), intent(in) :: a, b
    character(*), intent(in) :: what
    if(any(shape(a%val) .ne. shape(b%val))) call stop_program("stand-in diffstruc: "//what//" of unequal shapes (no broadcast here)")
  end subroutine same_shape

  pure subroutine ew_left_val(this, upstream_grad, output)
    class(array_type), intent(in) :: this
    real(real32), dimension(:,:), intent(in) :: upstream_grad
    real(real32), dimension(:,:), intent(out) :: output
    logical :: binary
    binary = associated(this%right_operand)
    select case(this%op)
    case(OP_ADD)
       output = upstream_grad
    case(OP_SUB)
       if(.not. binary .and. this%scalar_left)then
          output = -upstream_grad
       else
          output = upstream_grad
       end if
    case(OP_NEG)
       output = -upstream_grad
    case(OP_MUL)
       if(binary)then
          output = upstream_grad * this%right_operand%val
       else
          output = upstream_grad * this%scalar
       end if
    case(OP_DIV)
       if(binary)then
          output = upstream_grad / this%right_operand%val
       else if(this%scalar_left)then
          output = -upstream_grad * this%scalar / this%left_operand%val**2
       else
          output = upstream_grad / this%scalar
       end if
    case(OP_POW)
       output = upstream_grad * this%scalar * this%left_operand%val**(this%scalar - 1._real32)
    case(OP_EXP)
       output = upstream_grad * this%val
    case(OP_TANH)
       output = upstream_grad * (1._real32 - this%val**2)
    case(OP_SIGMOID)
       output = upstream_grad * this%val * (1._real32 - this%val)
    case(OP_GAUSS)
       output = 0._real32      ! value-only node, see gaussian_array
    case(OP_SQUARED)
       output = 2._real32 * upstream_grad * this%left_operand%val
    case(OP_LOG)
       output = upstream_grad / this%left_operand%val
    case(OP_ABS)
       output = upstream_grad * sign(1._real32, this%left_operand%val)
    case(OP_COPY)
       output = reshape(upstream_grad, shape(output))
    case(OP_MAX)
       if(binary)then
          where(this%left_operand%val .ge. this%right_operand%val)
             output = upstream_grad
          elsewhere
             output = 0._real32
          end where
       else
          where(this%left_operand%val .gt. this%scalar)
             output = upstream_grad
          elsewhere
             output = 0._real32
          end where
       end if
    case(OP_MERGE)
       where(this%mask)
          output = upstream_grad
       elsewhere
          output = 0._real32
       end where
    case default
       output = 0._real32
    end select
  end subroutine ew_left_val

  pure subroutine ew_right_val(this, upstream_grad, output)
    class(array_type), intent(in) :: this
    real(real32), dimension(:,:), intent(in) :: upstream_grad
    real(real32), dimension(:,:), intent(out) :: output
    select case(this%op)
    case(OP_ADD)
       if(size(output, 2) .eq. 1 .and. size(upstream_grad, 2) .gt. 1)then
          output(:, 1) = sum(upstream_grad, dim = 2)       ! the broadcast column of add_aa
       else
          output = upstream_grad
       end if
    case(OP_SUB)
       output = -upstream_grad
    case(OP_MUL)
       output = upstream_grad * this%left_operand%val
    case(OP_DIV)
       output = -upstream_grad * this%left_operand%val / this%right_operand%val**2
    case(OP_MAX)
       where(this%left_operand%val .ge. this%right_operand%val)
          output = 0._real32
       elsewhere
          output = upstream_grad
       end where
    case(OP_MERGE)
       where(this%mask)
          output = 0._real32
       elsewhere
          output = upstream_grad
       end where
    case default
       output = 0._real32
    end select
  end subroutine ew_right_val

  function add_aa(a, b) result(c)
    class(array_type), intent(in), target :: a
    class(array_type), intent(in), target :: b
    type(array_type), pointer :: c
    integer :: s
    if(size(b%val, 1) .eq. size(a%val, 1) .and. size(b%val, 2) .eq. 1 .and. size(a%val, 2) .gt. 1)then
       ! a [F, batch] + b [F, 1]: the same column for every sample (full_layer_type's bias, athena_full_layer.f90:855); the right
       ! partial sums the upstream gradient over the samples (ew_right_val)
       c => a%create_result()
       do s = 1, size(a%val, 2)
          c%val(:, s) = a%val(:, s) + b%val(:, 1)
       end do
    else
       call same_shape(a, b, "+")
       c => a%create_result()
       c%val = a%val + b%val
    end if
    call link_binary(c, a, b, OP_ADD)
    c%operation = 'add'
  end function add_aa

  function add_ar(a, b) result(c)
    class(array_type), intent(in), target :: a
    real(real32), intent(in) :: b
    type(array_type), pointer :: c
    c => a%create_result()
    c%val = a%val + b
    call link_unary(c, a, OP_ADD, b)
    c%operation = 'add_scalar'
  end function add_ar

  function add_ra(a, b) result(c)
    real(real32), intent(in) :: a
    class(array_type), intent(in), target :: b
    type(array_type), pointer :: c
    c => add_ar(b, a)
  end function add_ra

  function sub_aa(a, b) result(c)
    class(array_type), intent(in), target :: a
    class(array_type), intent(in), target :: b
    type(array_type), pointer :: c
    call same_shape(a, b, "-")
    c => a%create_result()
    c%val = a%val - b%val
    call link_binary(c, a, b, OP_SUB)
    c%operation = 'subtract'
  end function sub_aa

  function sub_ar(a, b) result(c)
    class(array_type), intent(in), target :: a
    real(real32), intent(in) :: b
    type(array_type), pointer :: c
    c => a%create_result()
    c%val = a%val - b
    call link_unary(c, a, OP_SUB, b)
    c%operation = 'subtract_scalar'
  end function sub_ar

  function sub_ra(a, b) result(c)
    real(real32), intent(in) :: a
    class(array_type), intent(in), target :: b
    type(array_type), pointer :: c
    c => b%create_result()
    c%val = a - b%val
    call link_unary(c, b, OP_SUB, a, scalar_left=.true.)
    c%operation = 'scalar_subtract'
  end function sub_ra

  function neg_a(a) result(c)
    class(array_type), intent(in), target :: a
    type(array_type), pointer :: c
    c => a%create_result()
    c%val = -a%val
    call link_unary(c, a, OP_NEG)
    c%operation = 'negate'
  end function neg_a

  function mul_aa(a, b) result(c)
    class(array_type), intent(in), target :: a
    class(array_type), intent(in), target :: b
    type(array_type), pointer :: c
    call same_shape(a, b, "*")
    c => a%create_result()
    c%val = a%val * b%val
    call link_binary(c, a, b, OP_MUL)
    c%operation = 'multiply'
  end function mul_aa

  function mul_ar(a, b) result(c)
    class(array_type), intent(in), target :: a
    real(real32), intent(in) :: b
    type(array_type), pointer :: c
    c => a%create_result()
    c%val = a%val * b
    call link_unary(c, a, OP_MUL, b)
    c%operation = 'multiply_scalar'
  end function mul_ar

  function mul_ra(a, b) result(c)
    real(real32), intent(in) :: a
    class(array_type), intent(in), target :: b
    type(array_type), pointer :: c
    c => mul_ar(b, a)
  end function mul_ra

  function div_aa(a, b) result(c)
    class(array_type), intent(in), target :: a
    class(array_type), intent(in), target :: b
    type(array_type), pointer :: c
    call same_shape(a, b, "/")
    c => a%create_result()
    c%val = a%val / b%val
    call link_binary(c, a, b, OP_DIV)
    c%operation = 'divide'
  end function div_aa

  function div_ar(a, b) result(c)
    class(array_type), intent(in), target :: a
    real(real32), intent(in) :: b
    type(array_type), pointer :: c
    c => a%create_result()
    c%val = a%val / b
    call link_unary(c, a, OP_DIV, b)
    c%operation = 'divide_scalar'
  end function div_ar

  function div_ra(a, b) result(c)
    real(real32), intent(in) :: a
    class(array_type), intent(in), target :: b
    type(array_type), pointer :: c
    c => b%create_result()
    c%val = a / b%val
    call link_unary(c, b, OP_DIV, a, scalar_left=.true.)
    c%operation = 'scalar_divide'
  end function div_ra

  function pow_ar(a, b) result(c)
    class(array_type), intent(in), target :: a
    real(real32), intent(in) :: b
    type(array_type), pointer :: c
    c => a%create_result()
    c%val = a%val**b
    call link_unary(c, a, OP_POW, b)
    c%operation = 'power'
  end function pow_ar

  function gt_ar(a, b) result(mask)
    class(array_type), intent(in) :: a
    real(real32), intent(in) :: b
    logical, allocatable :: mask(:,:)
    mask = a%val .gt. b
  end function gt_ar

  function gt_ra(a, b) result(mask)
    real(real32), intent(in) :: a
    class(array_type), intent(in) :: b
    logical, allocatable :: mask(:,:)
    mask = a .gt. b%val
  end function gt_ra

  function le_ar(a, b) result(mask)
    class(array_type), intent(in) :: a
    real(real32), intent(in) :: b
    logical, allocatable :: mask(:,:)
    mask = a%val .le. b
  end function le_ar

  function le_ra(a, b) result(mask)
    real(real32), intent(in) :: a
    class(array_type), intent(in) :: b
    logical, allocatable :: mask(:,:)
    mask = a .le. b%val
  end function le_ra

  ! ---- the module-level forms: a plain rank-2 real on the left, an array_type on the right (value-only constants on the left)
  function add_va(a, b) result(c)
    real(real32), dimension(:,:), intent(in) :: a
    class(array_type), intent(in), target :: b
    type(array_type), pointer :: c
    c => b%create_result()
    c%val = a + b%val
    call link_unary(c, b, OP_ADD)
    c%operation = 'add_constant'
  end function add_va

  function sub_va(a, b) result(c)
    real(real32), dimension(:,:), intent(in) :: a
    class(array_type), intent(in), target :: b
    type(array_type), pointer :: c
    c => b%create_result()
    c%val = a - b%val
    call link_unary(c, b, OP_NEG)
    c%operation = 'constant_subtract'
  end function sub_va

  function mul_va(a, b) result(c)
    real(real32), dimension(:,:), intent(in) :: a
    class(array_type), intent(in), target :: b
    type(array_type), pointer :: c
    type(array_type), pointer :: w
    allocate(w)
    call w%allocate(source=a)
    c => mul_aa(b, w)
  end function mul_va

  function div_va(a, b) result(c)
    real(real32), dimension(:,:), intent(in) :: a
    class(array_type), intent(in), target :: b
    type(array_type), pointer :: c
    type(array_type), pointer :: w
    allocate(w)
    call w%allocate(source=a)
    c => div_aa(w, b)
  end function div_va

  function gt_va(a, b) result(mask)
    real(real32), dimension(:,:), intent(in) :: a
    class(array_type), intent(in) :: b
    logical, allocatable :: mask(:,:)
    mask = a .gt. b%val
  end function gt_va

  function le_va(a, b) result(mask)
    real(real32), dimension(:,:), intent(in) :: a
    class(array_type), intent(in) :: b
    logical, allocatable :: mask(:,:)
    mask = a .le. b%val
  end function le_va

  function exp_array(a) result(c)
    class(array_type), intent(in), target :: a
    type(array_type), pointer :: c
    c => a%create_result()
    c%val = exp(a%val)
    call link_unary(c, a, OP_EXP)
    c%operation = 'exp'
  end function exp_array

  function tanh_array(a) result(c)
    class(array_type), intent(in), target :: a
    type(array_type), pointer :: c
    c => a%create_result()
    c%val = tanh(a%val)
    call link_unary(c, a, OP_TANH)
    c%operation = 'tanh'
  end function tanh_array

  function sigmoid_array(a) result(c)
    class(array_type), intent(in), target :: a
    type(array_type), pointer :: c
    c => a%create_result()
    c%val = 1._real32 / (1._real32 + exp(-a%val))
    call link_unary(c, a, OP_SIGMOID)
    c%operation = 'sigmoid'
  end function sigmoid_array

  function gaussian_array(a, mu, sigma) result(c)
    !! value only (no layer of this check differentiates through it: its partial answers zero)
    class(array_type), intent(in), target :: a
    real(real32), intent(in) :: mu, sigma
    type(array_type), pointer :: c
    c => a%create_result()
    c%val = exp(-(a%val - mu)**2 / (2._real32 * sigma**2)) / (sigma * sqrt(2._real32 * 3.14159265358979_real32))
    call link_unary(c, a, OP_GAUSS, mu)
    c%operation = 'gaussian'
  end function gaussian_array

  function max_ar(a, b) result(c)
    class(array_type), intent(in), target :: a
    real(real32), intent(in) :: b
    type(array_type), pointer :: c
    c => a%create_result()
    c%val = max(a%val, b)
    call link_unary(c, a, OP_MAX, b)
    c%operation = 'max_scalar'
  end function max_ar

  function max_aa(a, b) result(c)
    class(array_type), intent(in), target :: a, b
    type(array_type), pointer :: c
    call same_shape(a, b, "max")
    c => a%create_result()
    c%val = max(a%val, b%val)
    call link_binary(c, a, b, OP_MAX)
    c%operation = 'max'
  end function max_aa

  function merge_arrays(tsource, fsource, mask) result(c)
    class(array_type), intent(in), target :: tsource, fsource
    logical, dimension(:,:), intent(in) :: mask
    type(array_type), pointer :: c
    call same_shape(tsource, fsource, "merge")
    c => tsource%create_result()
    c%val = merge(tsource%val, fsource%val, mask)
    c%mask = mask
    call link_binary(c, tsource, fsource, OP_MERGE)
    c%operation = 'merge'
  end function merge_arrays

  function merge_array_real(tsource, fsource, mask) result(c)
    class(array_type), intent(in), target :: tsource
    real(real32), intent(in) :: fsource
    logical, dimension(:,:), intent(in) :: mask
    type(array_type), pointer :: c
    c => tsource%create_result()
    c%val = merge(tsource%val, fsource, mask)
    c%mask = mask
    call link_unary(c, tsource, OP_MERGE, fsource)
    c%operation = 'merge_scalar'
  end function merge_array_real

  function squared_array(a) result(c)
    class(array_type), intent(in), target :: a
    type(array_type), pointer :: c
    c => a%create_result()
    c%val = a%val**2
    call link_unary(c, a, OP_SQUARED)
    c%operation = 'squared'
  end function squared_array

  function log_array(a) result(c)
    class(array_type), intent(in), target :: a
    type(array_type), pointer :: c
    c => a%create_result()
    c%val = log(a%val)
    call link_unary(c, a, OP_LOG)
    c%operation = 'log'
  end function log_array

  function abs_array(a) result(c)
    class(array_type), intent(in), target :: a
    type(array_type), pointer :: c
    c => a%create_result()
    c%val = abs(a%val)
    call link_unary(c, a, OP_ABS)
    c%operation = 'abs'
  end function abs_array

  function sign_real_array(a, b) result(c)
    !! value only (constant almost everywhere)
    real(real32), intent(in) :: a
    class(array_type), intent(in), target :: b
    type(array_type), pointer :: c
    c => b%create_result()
    c%val = sign(a, b%val)
    c%operation = 'sign'
  end function sign_real_array

  function pack_array(a, dim) result(c)
    !! the flatten layer's op: the value already is [product(shape), samples] -- a copy with a rank-1 shape
    class(array_type), intent(in), target :: a
    integer, intent(in) :: dim
    type(array_type), pointer :: c
    c => a%create_result([size(a%val, 1), size(a%val, 2)])
    c%val = a%val
    call link_unary(c, a, OP_COPY)
    c%operation = 'pack'
  end function pack_array

  function reshape_array(a, new_shape) result(c)
    !! the reshape layer's op: same flat value, another %shape
    class(array_type), intent(in), target :: a
    integer, dimension(:), intent(in) :: new_shape
    type(array_type), pointer :: c
    c => a%create_result([new_shape, size(a%val, 2)])
    c%val = reshape(a%val, shape(c%val))
    call link_unary(c, a, OP_COPY)
    c%operation = 'reshape'
  end function reshape_array

  function mean_array(a, dim) result(c)
    !! no dim: the mean of every element, [1, 1]; dim = 2: over the columns, [rows, 1]; dim = 1: over the rows, [1, columns]
    class(array_type), intent(in), target :: a
    integer, intent(in), optional :: dim
    type(array_type), pointer :: c
    integer :: d
    d = 0
    if(present(dim)) d = dim
    select case(d)
    case(0)
       c => a%create_result([1, 1])
       c%val(1, 1) = sum(a%val) / real(size(a%val), real32)
    case(1)
       c => a%create_result([1, size(a%val, 2)])
       c%val(1, :) = sum(a%val, dim=1) / real(size(a%val, 1), real32)
    case default
       c => a%create_result([size(a%val, 1), 1])
       c%val(:, 1) = sum(a%val, dim=2) / real(size(a%val, 2), real32)
    end select
    c%indices = [d]
    c%get_partial_left => partial_left_from_val
    c%get_partial_left_val => mean_left_val
    c%left_operand => a
    c%owns_left_operand = a%is_temporary
    c%requires_grad = a%requires_grad
    c%is_forward = a%is_forward
    c%operation = 'mean'
  end function mean_array

  pure subroutine mean_left_val(this, upstream_grad, output)
    class(array_type), intent(in) :: this
    real(real32), dimension(:,:), intent(in) :: upstream_grad
    real(real32), dimension(:,:), intent(out) :: output
    integer :: j
    select case(this%indices(1))
    case(0)
       output = upstream_grad(1, 1) / real(size(output), real32)
    case(1)
       do j = 1, size(output, 1)
          output(j, :) = upstream_grad(1, :) / real(size(output, 1), real32)
       end do
    case default
       do j = 1, size(output, 2)
          output(:, j) = upstream_grad(:, 1) / real(size(output, 2), real32)
       end do
    end select
  end subroutine mean_left_val

  function concat_arrays(a, b, dim) result(c)
    !! the merge layers' op (outside the message-passing path): declared so that athena_diffstruc_extd_sub.f90 compiles
    class(array_type), intent(in), target :: a, b
    integer, intent(in) :: dim
    type(array_type), pointer :: c
    call stop_program("stand-in diffstruc: concat is not provided")
    c => a%create_result()
  end function concat_arrays

  ! ====================================================================================== matmul / sum / weighted_sum
  function matmul_arrays(a, b) result(c)
    !! c = A b with A = reshape(a%val(:,1), a%shape) -- a parameter tensor [F_out, F_in] held flat -- and b [F_in, N]
    class(array_type), intent(in), target :: a, b
    type(array_type), pointer :: c
    integer :: m, k
    if(.not.allocated(a%shape)) call stop_program("stand-in diffstruc: matmul needs the left operand's shape")
    if(size(a%shape) .ne. 2 .or. size(a%val, 2) .ne. 1) call stop_program("stand-in diffstruc: matmul(matrix parameter, array) only")
    m = a%shape(1); k = a%shape(2)
    if(k .ne. size(b%val, 1)) call stop_program("stand-in diffstruc: matmul inner dimensions differ")
    c => b%create_result([m, size(b%val, 2)])
    c%val = matmul(reshape(a%val(:, 1), [m, k]), b%val)
    c%get_partial_left => partial_left_from_val
    c%get_partial_right => partial_right_from_val
    c%get_partial_left_val => matmul_left_val
    c%get_partial_right_val => matmul_right_val
    c%left_operand => a
    c%right_operand => b
    c%owns_left_operand = a%is_temporary
    c%owns_right_operand = b%is_temporary
    c%requires_grad = a%requires_grad .or. b%requires_grad
    c%is_forward = a%is_forward .or. b%is_forward
    c%operation = 'matmul'
  end function matmul_arrays

  pure subroutine matmul_left_val(this, upstream_grad, output)
    !! dA = g b^T, flat as the parameter is
    class(array_type), intent(in) :: this
    real(real32), dimension(:,:), intent(in) :: upstream_grad
    real(real32), dimension(:,:), intent(out) :: output
    output(:, 1) = reshape(matmul(upstream_grad, transpose(this%right_operand%val)), [size(output, 1)])
  end subroutine matmul_left_val

  pure subroutine matmul_right_val(this, upstream_grad, output)
    !! db = A^T g
    class(array_type), intent(in) :: this
    real(real32), dimension(:,:), intent(in) :: upstream_grad
    real(real32), dimension(:,:), intent(out) :: output
    output = matmul(transpose(reshape(this%left_operand%val(:, 1), this%left_operand%shape(1:2))), upstream_grad)
  end subroutine matmul_right_val

  function sum_array(a, dim, new_dim_index, new_dim_size) result(c)
    !! dim = 2: the sum over columns; with new_dim_index / new_dim_size the result is [rows, new_dim_size] with the sum in column
    !! new_dim_index and zeros elsewhere (the Duvenaud readout's per-sample slot, athena_duvenaud_msgpass_layer.f90:849-852).
    !! dim = 1: the sum over rows, [1, columns].
    class(array_type), intent(in), target :: a
    integer, intent(in) :: dim
    integer, intent(in), optional :: new_dim_index, new_dim_size
    type(array_type), pointer :: c
    integer :: idx, n
    idx = 1; n = 1
    if(present(new_dim_index)) idx = new_dim_index
    if(present(new_dim_size)) n = new_dim_size
    if(dim .eq. 2)then
       c => a%create_result([size(a%val, 1), n])
       c%val = 0._real32
       c%val(:, idx) = sum(a%val, dim=2)
    else
       c => a%create_result([1, size(a%val, 2)])
       c%val(1, :) = sum(a%val, dim=1)
    end if
    c%indices = [dim, idx]
    c%get_partial_left => partial_left_from_val
    c%get_partial_left_val => sum_left_val
    c%left_operand => a
    c%owns_left_operand = a%is_temporary
    c%requires_grad = a%requires_grad
    c%is_forward = a%is_forward
    c%operation = 'sum'
  end function sum_array

  pure subroutine sum_left_val(this, upstream_grad, output)
    class(array_type), intent(in) :: this
    real(real32), dimension(:,:), intent(in) :: upstream_grad
    real(real32), dimension(:,:), intent(out) :: output
    integer :: j
    if(this%indices(1) .eq. 2)then
       do j = 1, size(output, 2)
          output(:, j) = upstream_grad(:, this%indices(2))
       end do
    else
       do j = 1, size(output, 1)
          output(j, :) = upstream_grad(1, :)
       end do
    end if
  end subroutine sum_left_val

  function weighted_sum(a, weights) result(c)
    !! harness loss: the scalar <a, weights>, so that grad_reverse() hands `weights` to a as its upstream gradient
    class(array_type), intent(in), target :: a
    real(real32), dimension(:,:), intent(in) :: weights
    type(array_type), pointer :: c
    type(array_type), pointer :: w
    allocate(w)
    call w%allocate(source=weights)
    w%requires_grad = .false.
    c => a%create_result([1, 1])
    c%val(1, 1) = sum(a%val * weights)
    c%get_partial_left => partial_left_from_val
    c%get_partial_left_val => weighted_sum_left_val
    c%left_operand => a
    c%right_operand => w
    c%owns_right_operand = .true.
    c%requires_grad = a%requires_grad
    c%operation = 'weighted_sum'
  end function weighted_sum

  pure subroutine weighted_sum_left_val(this, upstream_grad, output)
    class(array_type), intent(in) :: this
    real(real32), dimension(:,:), intent(in) :: upstream_grad
    real(real32), dimension(:,:), intent(out) :: output
    output = upstream_grad(1, 1) * this%right_operand%val
  end subroutine weighted_sum_left_val
end module diffstruc


module graphstruc
  !! graph_type as athena touches it, in its two uses:
  !!  * the DATA graphs the message-passing layers read -- CSR with adj_ja(1,:) = neighbour, adj_ja(2,:) = undirected edge id (0 on a
  !!    self loop), rows in vertex order, a row's neighbours in edge-list order;
  !!  * the small DIRECTED graph network_type keeps of its own layers (auto_graph, athena_network_sub.f90:832-905): vertex(:)%id,
  !!    edge(:)%index = [source, -target] / %id = the merge operator, a dense adjacency(i, j) = number of the edge i -> j,
  !!    add_vertex / add_edge / remove_edges / generate_adjacency() -- restated from those call sites.
  use coreutils, only: real32, stop_program
  implicit none
  private
  public :: graph_type, vertex_type, edge_type
  type :: vertex_type
     integer :: id = 0
     real(real32), allocatable :: feature(:)
  end type vertex_type
  type :: edge_type
     integer :: index(2) = 0
     integer :: id = 0
     real(real32) :: weight = 1._real32
     real(real32), allocatable :: feature(:)
  end type edge_type
  type :: graph_type
     integer :: num_vertices = 0, num_edges = 0, num_vertex_features = 0, num_edge_features = 0
     logical :: is_sparse = .true., directed = .false.
     character(len=128) :: name = ""
     integer, allocatable :: adj_ia(:), adj_ja(:,:)
     ! What athena's own code and tests do with these three decides their attributes:
     !   edge_weights     assigned whole while unallocated (set_graph_msgpass, athena_msgpass_layer_sub.f90:169; test_msgpass_network
     !                    .f90:285) and `allocate`d after set_num_edges (test_kipf_msgpass_layer.f90:97): allocatable, left alone by the setter
     !   edge_features    sections assigned after set_num_edges(n, nf) (test_msgpass_network.f90:288-289), `allocate`d after
     !                    set_num_edges(n) (test_kipf_msgpass_layer.f90:100): allocatable, allocated by the setter only when it is given nf
     !   vertex_features  sections assigned right after set_num_vertices(n, nf) (test_msgpass_network.f90:258-266) AND `allocate`d right
     !                    after the same call (test_kipf_msgpass_layer.f90:70-76): legal together only for a pointer the setter associates
     real(real32), pointer :: vertex_features(:,:) => null()
     real(real32), allocatable :: edge_features(:,:), edge_weights(:)
     type(vertex_type), allocatable :: vertex(:)
     type(edge_type), allocatable :: edge(:)
     integer, allocatable :: adjacency(:,:)
   contains
     procedure, pass(this) :: set_num_vertices
     procedure, pass(this) :: set_num_edges
     procedure, pass(this) :: generate_adjacency
     procedure, pass(this) :: add_self_loops
     procedure, pass(this) :: add_vertex
     procedure, pass(this) :: add_edge
     procedure, pass(this) :: remove_edges
  end type graph_type
contains
  subroutine set_num_vertices(this, num_vertices, num_vertex_features)
    class(graph_type), intent(inout) :: this
    integer, intent(in) :: num_vertices
    integer, intent(in), optional :: num_vertex_features
    this%num_vertices = num_vertices
    if(present(num_vertex_features)) this%num_vertex_features = num_vertex_features
    ! (athena's tests fill vertex_features(f, v) right after this call: test_msgpass_network.f90:258-266)
    allocate(this%vertex_features(this%num_vertex_features, num_vertices), source = 0._real32)
  end subroutine set_num_vertices

  subroutine set_num_edges(this, num_edges, num_edge_features)
    class(graph_type), intent(inout) :: this
    integer, intent(in) :: num_edges
    integer, intent(in), optional :: num_edge_features
    this%num_edges = num_edges
    if(present(num_edge_features)) this%num_edge_features = num_edge_features
    if(present(num_edge_features))then
       if(allocated(this%edge_features)) deallocate(this%edge_features)
       allocate(this%edge_features(num_edge_features, num_edges), source = 0._real32)
    end if
  end subroutine set_num_edges

  subroutine dense_adjacency(this)
    !! adjacency(i, j) = number of the edge i -> j (both ways for an undirected edge: index(2) > 0), 0 where there is none
    class(graph_type), intent(inout) :: this
    integer :: e, i, j
    if(allocated(this%adjacency)) deallocate(this%adjacency)
    allocate(this%adjacency(this%num_vertices, this%num_vertices), source = 0)
    if(.not.allocated(this%edge)) return
    do e = 1, size(this%edge)
       i = abs(this%edge(e)%index(1))
       j = abs(this%edge(e)%index(2))
       if(i .lt. 1 .or. j .lt. 1 .or. i .gt. this%num_vertices .or. j .gt. this%num_vertices) cycle
       this%adjacency(i, j) = e
       if(this%edge(e)%index(2) .gt. 0 .and. .not. this%directed) this%adjacency(j, i) = e
    end do
  end subroutine dense_adjacency

  subroutine add_vertex(this, feature, id, update_adjacency)
    class(graph_type), intent(inout) :: this
    real(real32), dimension(:), intent(in), optional :: feature
    integer, intent(in), optional :: id
    logical, intent(in), optional :: update_adjacency
    type(vertex_type) :: v
    if(present(feature)) v%feature = feature
    if(present(id)) v%id = id
    if(allocated(this%vertex))then
       this%vertex = [ this%vertex, v ]
    else
       this%vertex = [ v ]
    end if
    this%num_vertices = size(this%vertex)
    call dense_adjacency(this)
  end subroutine add_vertex

  subroutine add_edge(this, index, feature, id, weight, directed, update_adjacency)
    class(graph_type), intent(inout) :: this
    integer, dimension(2), intent(in) :: index
    real(real32), dimension(:), intent(in), optional :: feature
    integer, intent(in), optional :: id
    real(real32), intent(in), optional :: weight
    logical, intent(in), optional :: directed, update_adjacency
    type(edge_type) :: e
    e%index = index
    if(present(feature)) e%feature = feature
    if(present(id)) e%id = id
    if(present(weight)) e%weight = weight
    if(allocated(this%edge))then
       this%edge = [ this%edge, e ]
    else
       this%edge = [ e ]
    end if
    this%num_edges = size(this%edge)
    call dense_adjacency(this)
  end subroutine add_edge

  subroutine remove_edges(this, indices, update_adjacency)
    !! the edges numbered `indices` (zeros ignored) go; the others close ranks, the adjacency follows
    class(graph_type), intent(inout) :: this
    integer, dimension(:), intent(in) :: indices
    logical, intent(in), optional :: update_adjacency
    type(edge_type), allocatable :: kept(:)
    integer :: e, k
    if(.not.allocated(this%edge)) return
    allocate(kept(count([(all(indices .ne. e), e = 1, size(this%edge))])))
    k = 0
    do e = 1, size(this%edge)
       if(any(indices .eq. e)) cycle
       k = k + 1
       kept(k) = this%edge(e)
    end do
    call move_alloc(kept, this%edge)
    this%num_edges = size(this%edge)
    call dense_adjacency(this)
  end subroutine remove_edges

  subroutine generate_adjacency(this, index_list)
    !! with an edge list: the CSR of a data graph; without: the dense adjacency of the directed layer graph from its edge(:)
    class(graph_type), intent(inout) :: this
    integer, dimension(:,:), intent(in), optional :: index_list
    integer :: e, v, n
    integer, allocatable :: fill(:)
    if(.not.present(index_list))then
       call dense_adjacency(this)
       this%is_sparse = .false.
       return
    end if
    n = this%num_vertices
    if(allocated(this%adj_ia)) deallocate(this%adj_ia)
    if(allocated(this%adj_ja)) deallocate(this%adj_ja)
    allocate(this%adj_ia(n + 1), fill(n))
    this%adj_ia = 0
    do e = 1, size(index_list, 2)
       this%adj_ia(index_list(1, e) + 1) = this%adj_ia(index_list(1, e) + 1) + 1
       if(index_list(2, e) .ne. index_list(1, e)) this%adj_ia(index_list(2, e) + 1) = this%adj_ia(index_list(2, e) + 1) + 1
    end do
    this%adj_ia(1) = 1
    do v = 1, n
       this%adj_ia(v + 1) = this%adj_ia(v) + this%adj_ia(v + 1)
    end do
    allocate(this%adj_ja(2, this%adj_ia(n + 1) - 1))
    fill = this%adj_ia(1:n)
    do e = 1, size(index_list, 2)
       this%adj_ja(:, fill(index_list(1, e))) = [index_list(2, e), e]
       fill(index_list(1, e)) = fill(index_list(1, e)) + 1
       if(index_list(2, e) .ne. index_list(1, e))then
          this%adj_ja(:, fill(index_list(2, e))) = [index_list(1, e), e]
          fill(index_list(2, e)) = fill(index_list(2, e)) + 1
       end if
    end do
    this%num_edges = size(index_list, 2)
    this%is_sparse = .true.
  end subroutine generate_adjacency

  subroutine add_self_loops(this)
    !! one (v, edge id 0) entry at the head of every row that has no self loop yet
    class(graph_type), intent(inout) :: this
    integer :: v, w, n, k
    integer, allocatable :: ia(:), ja(:,:)
    logical :: has
    n = this%num_vertices
    allocate(ia(n + 1), ja(2, size(this%adj_ja, 2) + n))
    k = 0
    do v = 1, n
       ia(v) = k + 1
       has = .false.
       do w = this%adj_ia(v), this%adj_ia(v + 1) - 1
          if(this%adj_ja(1, w) .eq. v) has = .true.
       end do
       if(.not. has)then
          k = k + 1; ja(:, k) = [v, 0]
       end if
       do w = this%adj_ia(v), this%adj_ia(v + 1) - 1
          k = k + 1; ja(:, k) = this%adj_ja(:, w)
       end do
    end do
    ia(n + 1) = k + 1
    this%adj_ia = ia
    this%adj_ja = ja(:, 1:k)
  end subroutine add_self_loops
end module graphstruc
