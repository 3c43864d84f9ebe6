!! LINKED AND RUN on the GPU (tests/test_gpu_integration_run.py): the three drop-in LAYER TYPES of
!! athena_amd/fortran/athena_dropin/athena_hip_msgpass_layers.f90 inside athena's own layer machinery.  What is linked here:
!! athena's REAL modules, compiled by run.sh from the reference checkout read in place -- athena__base_layer (+ its
!! submodules: get_params / set_params / get_gradients / set_gradients, athena_base_layer_sub.f90:545-691), athena__msgpass_layer
!! (+ submodule: forward_msgpass :184-198, set_graph_msgpass :144-174), the three concrete layer modules (constructors,
!! set_hyperparams, init, print_to_unit, read), the activations, the initialisers, athena__container_layer (+ submodule: the
!! checkpoint registry) -- over the ONE stand-in for coreutils / diffstruc / graphstruc (standins.f90), plus libathena_mp.so.
!!
!! The program reads like the reference's own layer tests (test/test_kipf_msgpass_layer.f90, test_duvenaud_msgpass_layer.f90,
!! test_gno_layer.f90): build each layer WITH ITS CONSTRUCTOR, check it IS-A parent type, set_graph, get / set_params,
!! `call layer%forward(input)` (athena's forward_msgpass -> our update_message / update_readout), a scalar loss node,
!! grad_reverse, get_gradients, print_to_unit -> read through the registry.  Graphs: the reference's hand topologies
!! (6 vertices / 8 edges, test_kipf_msgpass_layer.f90:83-90; 5 vertices / 6 edges, test_msgpass_network.f90:249-276) and
!! batches of >= 1024 vertices (the MFMA / banded routes).  Every output, input gradient and flat parameter gradient is held
!! against the shipped *_mp_layer_type (athena_amd/fortran/athena_mp_layers.f90, oracle-tested under pytest) on the same
!! parameters; with RUN_LAYERS_DUMP=<dir> every case is also written out for tests/ to hold against oracle/layers.py.
!! athena_mp_pair_stats must show ONE fused device pass and ONE hand-over per two-partial node.
!! Prints "RUN_LAYERS_OK <layers passed> 3" or stops with a message.
program run_layers
  use, intrinsic :: iso_c_binding
  use coreutils, only: real32
  use graphstruc, only: graph_type
  use diffstruc, only: array_type, weighted_sum, operator(+)
  use athena__base_layer, only: base_layer_type, learnable_layer_type
  use athena__msgpass_layer, only: msgpass_layer_type
  use athena__kipf_msgpass_layer, only: kipf_msgpass_layer_type
  use athena__duvenaud_msgpass_layer, only: duvenaud_msgpass_layer_type
  use athena__graph_nop_layer, only: graph_nop_layer_type
  use athena__container_layer, only: list_of_layer_types, allocate_list_of_layer_types
  use athena__hip_msgpass_layers
  use athena_mp_c
  use athena_mp_layers, only: mp_graph_type, kipf_mp_layer_type, duvenaud_mp_layer_type, graph_nop_mp_layer_type
  implicit none
  integer :: passed
  logical :: stock_only
  character(len=32) :: mode_arg
  integer(c_int64_t) :: f0, h0, f1, h1
  character(len=512) :: dump_dir
  integer :: dump_len, dump_stat
  integer, allocatable :: pairs68(:,:), pairs56(:,:)

  call get_environment_variable("RUN_LAYERS_DUMP", dump_dir, dump_len, dump_stat)
  if(dump_stat .ne. 0) dump_len = 0
  ! `run_layers stock`: the SAME program with athena's own layer types in the place of the hip_* ones -- no GPU, nothing of
  ! libathena_mp.so is called; with RUN_LAYERS_DUMP the results of the reference's own compiled loops are written out, and
  ! tests/test_integration_compile.py holds oracle/layers.py against them (corroboration of the oracle, not a pin)
  call get_command_argument(1, mode_arg)
  stock_only = trim(mode_arg) .eq. "stock"
  if(.not.stock_only)then
     if(athena_mp_init(0_c_int) .ne. 0) call fail("athena_mp_init: "//athena_mp_error_message())
     call register_hip_msgpass_layers()
  else
     call allocate_list_of_layer_types()
  end if
  f0 = 0; h0 = 0; f1 = 0; h1 = 0

  ! the reference's hand topologies (tests/golden/reference_test_topologies.json holds the same lists)
  pairs68 = reshape([1,2, 1,3, 2,3, 2,4, 3,5, 4,5, 4,6, 5,6], [2, 8])
  pairs56 = reshape([1,2, 1,3, 2,3, 2,4, 3,5, 4,5], [2, 6])

  passed = 0
  ! ---- Kipf: the hand graph (two samples, two time steps, relu), then a 3000-vertex batch at 64 features (MFMA dense step)
  call kipf_case("kipf_hand", [6, 6], pairs68, nvf=[10, 7, 5], steps=2, activation="relu")
  call kipf_case("kipf_wide", [3000], pairs68, nvf=[64, 64], steps=1, activation="none")
  call kipf_case("kipf_swish", [40], pairs68, nvf=[8, 8], steps=1, activation="swish")      ! athena's own apply, not an epilogue
  passed = passed + 1
  ! ---- Duvenaud: defaults (sigmoid / softmax: the fused route) on the hand graph and on >= 1024 vertices at perf widths
  ! (F_v 64 / F_e 8: MFMA update, one-launch reverse), then an activation the device does not fuse (op-granular route)
  if(.not.stock_only)then
     if(athena_mp_pair_stats(f0, h0) .ne. 0) call fail("pair_stats")
  end if
  call duvenaud_case("duvenaud_hand", [5, 6], pairs56, fv=6, fe=1, steps=4, nout=10, mx=10, activation="sigmoid")
  if(.not.stock_only)then
     if(athena_mp_pair_stats(f1, h1) .ne. 0) call fail("pair_stats")
  end if
  if(.not.stock_only .and. (f1 - f0 .ne. 8 .or. h1 - h0 .ne. 8)) call fail_counts("duvenaud_hand", f1 - f0, h1 - h0, 8)
  call duvenaud_case("duvenaud_wide", [1500, 1200], pairs56, fv=64, fe=8, steps=2, nout=10, mx=4, activation="sigmoid")
  if(.not.stock_only)then
     if(athena_mp_pair_stats(f0, h0) .ne. 0) call fail("pair_stats")
  end if
  if(.not.stock_only .and. (f0 - f1 .ne. 4 .or. h0 - h1 .ne. 4)) call fail_counts("duvenaud_wide", f0 - f1, h0 - h1, 4)
  call duvenaud_case("duvenaud_leaky", [5, 6], pairs56, fv=6, fe=2, steps=2, nout=3, mx=4, activation="leaky_relu")
  passed = passed + 1
  ! ---- graph neural operator: the hand graph at generic widths, then 1200 vertices at 64 / 64 / H = 64 (one-contraction reverse)
  if(.not.stock_only)then
     if(athena_mp_pair_stats(f0, h0) .ne. 0) call fail("pair_stats")
  end if
  call gno_case("gno_hand", [6], pairs68, d=2, h=7, fi=5, fo=9, use_bias=.true., activation="tanh")
  call gno_case("gno_wide", [1200], pairs68, d=3, h=64, fi=64, fo=64, use_bias=.true., activation="none")
  if(.not.stock_only)then
     if(athena_mp_pair_stats(f1, h1) .ne. 0) call fail("pair_stats")
  end if
  if(.not.stock_only .and. (f1 - f0 .ne. 2 .or. h1 - h0 .ne. 2)) call fail_counts("gno", f1 - f0, h1 - h0, 2)
  passed = passed + 1

  if(.not.stock_only)then
     if(athena_mp_finalize() .ne. 0) call fail("finalize")
  end if
  write(*, '(A,I0,A)') "RUN_LAYERS_OK ", passed, " 3"

contains

  subroutine fail(what)
    character(*), intent(in) :: what
    write(0, '(A)') "run_layers: "//what
    error stop 1
  end subroutine fail

  subroutine fail_counts(what, fused, handed, want)
    character(*), intent(in) :: what
    integer(c_int64_t), intent(in) :: fused, handed
    integer, intent(in) :: want
    write(0, '(A,A,A,I0,A,I0,A,I0)') "run_layers: ", what, ": fused passes ", fused, " hand-overs ", handed, " expected ", want
    error stop 1
  end subroutine fail_counts

  ! ------------------------------------------------------------------------------------------ data
  subroutine build_graph(g, n, hand_pairs, self_loops, num_edge_features)
    !! n = the hand topology's vertex count: that graph; larger: a chain with a chord every 7th vertex.
    !! Built through graph_type's own calls, as the reference's tests do (test_kipf_msgpass_layer.f90:64-100).
    type(graph_type), intent(out) :: g
    integer, intent(in) :: n
    integer, intent(in) :: hand_pairs(:,:)
    logical, intent(in) :: self_loops
    integer, intent(in) :: num_edge_features
    integer, allocatable :: index_list(:,:)
    integer :: i, k

    if(n .eq. maxval(hand_pairs))then
       index_list = hand_pairs
    else
       allocate(index_list(2, (n - 1) + (n - 1) / 7))
       do i = 1, n - 1
          index_list(:, i) = [i, i + 1]
       end do
       k = n - 1
       do i = 1, n - 8, 7
          k = k + 1
          index_list(:, k) = [i, i + 5]
       end do
       index_list = index_list(:, 1:k)
    end if
    call g%set_num_vertices(n, 1)
    call g%set_num_edges(size(index_list, 2), num_edge_features)
    g%is_sparse = .true.
    call g%generate_adjacency(index_list)
    if(self_loops) call g%add_self_loops()
    allocate(g%edge_weights(g%num_edges))
    g%edge_weights = 1._real32
  end subroutine build_graph

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

  subroutine fill_params(p, seed)
    real(real32), intent(out) :: p(:)
    integer, intent(in) :: seed
    integer :: i
    do i = 1, size(p)
       p(i) = (real(mod(11 * i + 17 * seed + (i / 5) * 3, 89), real32) / 89._real32 - 0.5_real32) * 0.6_real32
    end do
  end subroutine fill_params

  subroutine close_to(got, want, what)
    real(real32), intent(in) :: got(:,:), want(:,:)
    character(*), intent(in) :: what
    real(real32) :: scale
    if(any(shape(got) .ne. shape(want)))then
       write(0, *) what, ": shapes ", shape(got), " and ", shape(want)
       error stop 1
    end if
    scale = max(maxval(abs(want)), 1.e-30_real32)
    if(maxval(abs(got - want)) .gt. 1.e-5_real32 * scale)then
       write(0, *) what, ": worst", maxval(abs(got - want)) / scale
       error stop 1
    end if
  end subroutine close_to

  subroutine close_to_1d(got, want, what)
    real(real32), intent(in) :: got(:), want(:)
    character(*), intent(in) :: what
    call close_to(reshape(got, [size(got), 1]), reshape(want, [size(want), 1]), what)
  end subroutine close_to_1d

  subroutine to_mp_graph(g, m)
    type(graph_type), intent(in) :: g
    type(mp_graph_type), intent(out) :: m
    m%num_vertices = g%num_vertices
    m%num_edges = g%num_edges
    m%adj_ia = g%adj_ia
    m%adj_ja = g%adj_ja
  end subroutine to_mp_graph

  ! ------------------------------------------------------------------------------------------ dumps for the oracle-side check
  subroutine dump_meta(case_name, keys, vals)
    character(*), intent(in) :: case_name
    character(*), intent(in) :: keys(:)
    integer, intent(in) :: vals(:)
    integer :: u, i
    if(dump_len .eq. 0) return
    open(newunit=u, file=dump_dir(1:dump_len)//"/"//case_name//".meta", status='replace', action='write')
    do i = 1, size(keys)
       write(u, '(A,1X,I0)') trim(keys(i)), vals(i)
    end do
    close(u)
  end subroutine dump_meta

  subroutine dump_r(case_name, what, a)
    character(*), intent(in) :: case_name, what
    real(real32), intent(in) :: a(..)
    integer :: u
    if(dump_len .eq. 0) return
    open(newunit=u, file=dump_dir(1:dump_len)//"/"//case_name//"."//what//".f32", status='replace', action='write', &
         access='stream', form='unformatted')
    select rank(a)
    rank(1)
       write(u) a
    rank(2)
       write(u) a
    end select
    close(u)
  end subroutine dump_r

  subroutine dump_i(case_name, what, a)
    character(*), intent(in) :: case_name, what
    integer, intent(in) :: a(..)
    integer :: u
    if(dump_len .eq. 0) return
    open(newunit=u, file=dump_dir(1:dump_len)//"/"//case_name//"."//what//".i32", status='replace', action='write', &
         access='stream', form='unformatted')
    select rank(a)
    rank(1)
       write(u) a
    rank(2)
       write(u) a
    end select
    close(u)
  end subroutine dump_i

  function sfx(s) result(r)
    integer, intent(in) :: s
    character(len=:), allocatable :: r
    character(len=8) :: b
    write(b, '(I0)') s
    r = trim(b)
  end function sfx

  ! ------------------------------------------------------------------------------------------ card round trip through the registry
  subroutine card_round_trip(layer, card, want_params, reread, restores_parameters)
    !! what network%print / network%read do for one layer (athena_network_sub.f90:433-447): header, print_to_unit, footer;
    !! then the FIRST registry entry of that name reads it back.  restores_parameters = .false. for the DUVENAUD card: the
    !! reference's read_duvenaud is an empty body (athena_duvenaud_msgpass_layer.f90:700-709) and its card carries neither the
    !! degree range nor num_outputs, so stock athena does not get the parameters back either; the hip type inherits exactly that.
    class(base_layer_type), intent(in) :: layer
    character(*), intent(in) :: card
    real(real32), intent(in) :: want_params(:)
    class(base_layer_type), allocatable, intent(out) :: reread
    logical, intent(in) :: restores_parameters
    integer :: unit, idx
    character(len=:), allocatable :: path

    path = "run_layers_"//card//".tmp"
    open(newunit=unit, file=path, status='replace', action='write')
    write(unit, '(A)') card
    call layer%print_to_unit(unit)
    write(unit, '(A)') "END "//card
    close(unit)
    idx = findloc([list_of_layer_types(:)%name], layer%name, dim=1)
    if(idx .eq. 0) call fail("no registry entry for "//layer%name)
    open(newunit=unit, file=path, status='old', action='read')
    read(unit, *)
    allocate(reread, source=list_of_layer_types(idx)%read_ptr(unit))
    close(unit, status='delete')
    if(.not. restores_parameters) return
    select type(reread)
    class is(learnable_layer_type)
       if(reread%num_params .ne. size(want_params)) call fail(card//": parameter count after the card round trip")
       ! cards hold E16.8E2: 8 significant digits
       call close_to_1d(reread%get_params(), want_params, card//": parameters after the card round trip")
    class default
       call fail(card//": the reader returned a layer without parameters")
    end select
  end subroutine card_round_trip

  ! ------------------------------------------------------------------------------------------ Kipf
  subroutine kipf_case(case_name, sizes, hand_pairs, nvf, steps, activation)
    character(*), intent(in) :: case_name, activation
    integer, intent(in) :: sizes(:), hand_pairs(:,:), nvf(:), steps
    class(base_layer_type), allocatable :: layer, reread
    type(graph_type), allocatable :: graph(:)
    type(array_type), allocatable, target :: input(:,:)
    type(array_type), pointer :: loss
    real(real32), allocatable :: params(:), grads(:), ref_grads(:), x(:,:), up(:,:), ref_out(:,:), ref_dx(:,:)
    real(real32), allocatable :: xs(:,:), ups(:,:)
    type(kipf_mp_layer_type) :: ref
    type(mp_graph_type), allocatable :: mg(:)
    integer :: s, batch, n_total, v0, fo

    batch = size(sizes)
    fo = nvf(size(nvf))
    if(stock_only)then
       layer = kipf_msgpass_layer_type(num_vertex_features=nvf, num_time_steps=steps, activation=activation)
    else
       layer = hip_kipf_msgpass_layer_type(num_vertex_features=nvf, num_time_steps=steps, activation=activation)
    end if
    if(layer%name .ne. 'kipf') call fail(case_name//": layer name")
    select type(layer)
    class is(kipf_msgpass_layer_type)            ! IS-A stock Kipf layer: every consumer of the parent type takes it
       if(layer%num_time_steps .ne. steps) call fail(case_name//": num_time_steps")
       if(layer%num_vertex_features(steps) .ne. fo) call fail(case_name//": num_vertex_features")
    class default
       call fail(case_name//": not a kipf_msgpass_layer_type")
    end select

    allocate(graph(batch), mg(batch), input(2, batch))
    n_total = sum(sizes)
    allocate(xs(nvf(1), n_total), ups(fo, n_total))
    v0 = 0
    do s = 1, batch
       call build_graph(graph(s), sizes(s), hand_pairs, self_loops=.true., num_edge_features=1)
       call to_mp_graph(graph(s), mg(s))
       allocate(x(nvf(1), sizes(s)), up(fo, sizes(s)))
       call fill(x, 10 + s); call fill(up, 20 + s)
       call input(1, s)%allocate(source=x)
       call input(1, s)%set_requires_grad(.true.)
       input(1, s)%is_temporary = .false.
       allocate(x(1, graph(s)%num_edges)); x = 0._real32         ! edge features: carried, not read by this layer
       call input(2, s)%allocate(source=x)
       input(2, s)%is_temporary = .false.
       xs(:, v0 + 1:v0 + sizes(s)) = input(1, s)%val
       ups(:, v0 + 1:v0 + sizes(s)) = up
       v0 = v0 + sizes(s)
       deallocate(x, up)
    end do

    call layer%set_graph(graph)
    select type(layer)
    class is(learnable_layer_type)
       params = layer%get_params()
       if(size(params) .ne. layer%get_num_params()) call fail(case_name//": get_params size")
       call fill_params(params, 3)
       call layer%set_params(params)
       call layer%forward(input)                                  ! forward_msgpass -> update_message_hip_kipf
       if(any(shape(layer%output) .ne. [1, batch])) call fail(case_name//": output shape")
       v0 = 0
       do s = 1, batch
          if(s .eq. 1)then
             loss => weighted_sum(layer%output(1, s), ups(:, v0 + 1:v0 + sizes(s)))
          else
             loss => loss + weighted_sum(layer%output(1, s), ups(:, v0 + 1:v0 + sizes(s)))
          end if
          v0 = v0 + sizes(s)
       end do
       call loss%grad_reverse(reset_graph=.true.)
       grads = layer%get_gradients()
    class default
       call fail(case_name//": not learnable")
    end select

    ! the shipped layer type on the same parameters: the batch as one block-diagonal graph
    if(.not.stock_only)then
       ref = kipf_mp_layer_type(nvf, steps, activation=activation)
       call ref%set_graph(mg)
       call ref%set_params(params)
       ref_out = ref%forward(xs)
       ref_dx = ref%backward(ups)
       ref_grads = ref%get_gradients()
    end if
    v0 = 0
    do s = 1, batch
       if(.not.stock_only) call close_to(layer%output(1, s)%val, ref_out(:, v0 + 1:v0 + sizes(s)), case_name//": output, sample "//sfx(s))
       if(.not.stock_only) call close_to(input(1, s)%grad%val, ref_dx(:, v0 + 1:v0 + sizes(s)), case_name//": input gradient, sample "//sfx(s))
       call dump_i(case_name, "ia"//sfx(s), graph(s)%adj_ia)
       call dump_i(case_name, "ja"//sfx(s), graph(s)%adj_ja)
       call dump_r(case_name, "x"//sfx(s), input(1, s)%val)
       call dump_r(case_name, "up"//sfx(s), ups(:, v0 + 1:v0 + sizes(s)))
       call dump_r(case_name, "out"//sfx(s), layer%output(1, s)%val)
       call dump_r(case_name, "dx"//sfx(s), input(1, s)%grad%val)
       v0 = v0 + sizes(s)
    end do
    if(.not.stock_only) call close_to_1d(grads, ref_grads, case_name//": flat parameter gradients (get_gradients)")
    call dump_r(case_name, "params", params)
    call dump_r(case_name, "grads", grads)
    call dump_meta(case_name, [character(16) :: "batch", "steps", "f_in", "f_out"], [batch, steps, nvf(1), fo])
    call dump_i(case_name, "nvf", nvf)

    call card_round_trip(layer, "KIPF", params, reread, restores_parameters=.true.)
    if(.not.stock_only)then
       select type(reread)
       type is(hip_kipf_msgpass_layer_type)
       class default
          call fail(case_name//": the registry did not hand back a hip_kipf_msgpass_layer_type")
       end select
    end if
    if(.not.stock_only) call ref%destroy()
  end subroutine kipf_case

  ! ------------------------------------------------------------------------------------------ Duvenaud
  subroutine duvenaud_case(case_name, sizes, hand_pairs, fv, fe, steps, nout, mx, activation)
    character(*), intent(in) :: case_name, activation
    integer, intent(in) :: sizes(:), hand_pairs(:,:), fv, fe, steps, nout, mx
    class(base_layer_type), allocatable :: layer, reread
    type(graph_type), allocatable :: graph(:)
    type(array_type), allocatable, target :: input(:,:)
    type(array_type), pointer :: loss
    real(real32), allocatable :: params(:), grads(:), ref_grads(:), x(:,:), e(:,:), up(:,:), ref_out(:,:), ref_dx(:,:), &
         ref_de(:,:), xs(:,:), es(:,:)
    type(duvenaud_mp_layer_type) :: ref
    type(mp_graph_type), allocatable :: mg(:)
    integer :: s, batch, n_total, e_total, v0, e0

    batch = size(sizes)
    if(stock_only)then
       layer = duvenaud_msgpass_layer_type(num_vertex_features=[fv], num_edge_features=[fe], num_time_steps=steps, &
            max_vertex_degree=mx, num_outputs=nout, message_activation=activation)
    else
       layer = hip_duvenaud_msgpass_layer_type(num_vertex_features=[fv], num_edge_features=[fe], num_time_steps=steps, &
            max_vertex_degree=mx, num_outputs=nout, message_activation=activation)
    end if
    if(layer%name .ne. 'duvenaud') call fail(case_name//": layer name")
    select type(layer)
    class is(duvenaud_msgpass_layer_type)
       if(layer%max_vertex_degree .ne. mx .or. layer%min_vertex_degree .ne. 1) call fail(case_name//": degree range")
       if(trim(layer%activation_readout%name) .ne. 'softmax') call fail(case_name//": default readout activation")
    class default
       call fail(case_name//": not a duvenaud_msgpass_layer_type")
    end select

    allocate(graph(batch), mg(batch), input(2, batch))
    do s = 1, batch
       ! (stock mode: no self loops.  A self-loop entry carries edge id 0, and the reference reads e(:, 0) / writes output(:, 0) for
       ! it -- athena_diffstruc_extd_sub_duvenaud.f90:34-42, 166-169: one column in front of the array, SURVEY.md F7 -- where the
       ! oracle and the HIP path contribute a zero edge-feature vector; without self loops the two are the same map)
       call build_graph(graph(s), sizes(s), hand_pairs, self_loops=.not.stock_only, num_edge_features=fe)
       call to_mp_graph(graph(s), mg(s))
    end do
    n_total = sum(sizes)
    e_total = sum(graph(:)%num_edges)
    allocate(xs(fv, n_total), es(fe, e_total), up(nout, batch))
    call fill(up, 41)
    v0 = 0; e0 = 0
    do s = 1, batch
       allocate(x(fv, sizes(s)), e(fe, graph(s)%num_edges))
       call fill(x, 30 + s); call fill(e, 35 + s)
       x = x + 0.45_real32; e = e + 0.45_real32                 ! features in [0, 1), as the chemical example's
       call input(1, s)%allocate(source=x)
       call input(1, s)%set_requires_grad(.true.)
       input(1, s)%is_temporary = .false.
       call input(2, s)%allocate(source=e)
       call input(2, s)%set_requires_grad(.true.)
       input(2, s)%is_temporary = .false.
       xs(:, v0 + 1:v0 + sizes(s)) = x
       es(:, e0 + 1:e0 + graph(s)%num_edges) = e
       v0 = v0 + sizes(s); e0 = e0 + graph(s)%num_edges
       deallocate(x, e)
    end do

    call layer%set_graph(graph)
    select type(layer)
    class is(learnable_layer_type)
       params = layer%get_params()
       if(size(params) .ne. layer%get_num_params()) call fail(case_name//": get_params size")
       call fill_params(params, 5)
       call layer%set_params(params)
       call layer%forward(input)              ! forward_msgpass -> update_message_hip_duvenaud, update_readout_hip_duvenaud
       if(any(shape(layer%output) .ne. [1, 1])) call fail(case_name//": output shape")
       if(any(shape(layer%output(1, 1)%val) .ne. [nout, batch])) call fail(case_name//": output value shape")
       loss => weighted_sum(layer%output(1, 1), up)
       call loss%grad_reverse(reset_graph=.true.)
       grads = layer%get_gradients()
    class default
       call fail(case_name//": not learnable")
    end select

    if(.not.stock_only)then
       ref = duvenaud_mp_layer_type([fv], [fe], steps, mx, nout, message_activation=activation)
       call ref%set_graph(mg)
       call ref%set_params(params)
       ref_out = ref%forward(xs, es)
       call ref%backward(up, dx=ref_dx, de=ref_de)
       ref_grads = ref%get_gradients()
    end if
    if(.not.stock_only) call close_to(layer%output(1, 1)%val, ref_out, case_name//": output")
    v0 = 0; e0 = 0
    do s = 1, batch
       if(.not.stock_only) call close_to(input(1, s)%grad%val, ref_dx(:, v0 + 1:v0 + sizes(s)), case_name//": vertex-feature gradient, sample "//sfx(s))
       if(.not.stock_only)then
          call close_to(input(2, s)%grad%val, ref_de(:, e0 + 1:e0 + graph(s)%num_edges), &
               case_name//": edge-feature gradient, sample "//sfx(s))
       end if
       call dump_i(case_name, "ia"//sfx(s), graph(s)%adj_ia)
       call dump_i(case_name, "ja"//sfx(s), graph(s)%adj_ja)
       call dump_r(case_name, "x"//sfx(s), input(1, s)%val)
       call dump_r(case_name, "e"//sfx(s), input(2, s)%val)
       call dump_r(case_name, "dx"//sfx(s), input(1, s)%grad%val)
       call dump_r(case_name, "de"//sfx(s), input(2, s)%grad%val)
       v0 = v0 + sizes(s); e0 = e0 + graph(s)%num_edges
    end do
    if(.not.stock_only) call close_to_1d(grads, ref_grads, case_name//": flat parameter gradients (get_gradients)")
    call dump_r(case_name, "up", up)
    call dump_r(case_name, "out", layer%output(1, 1)%val)
    call dump_r(case_name, "params", params)
    call dump_r(case_name, "grads", grads)
    call dump_meta(case_name, [character(16) :: "batch", "steps", "fv", "fe", "nout", "min_degree", "max_degree"], &
         [batch, steps, fv, fe, nout, 1, mx])

    call card_round_trip(layer, "DUVENAUD", params, reread, restores_parameters=.false.)
    if(.not.stock_only)then
       select type(reread)
       type is(hip_duvenaud_msgpass_layer_type)
       class default
          call fail(case_name//": the registry did not hand back a hip_duvenaud_msgpass_layer_type")
       end select
    end if
    if(.not.stock_only) call ref%destroy()
  end subroutine duvenaud_case

  ! ------------------------------------------------------------------------------------------ graph neural operator
  subroutine gno_case(case_name, sizes, hand_pairs, d, h, fi, fo, use_bias, activation)
    character(*), intent(in) :: case_name, activation
    integer, intent(in) :: sizes(:), hand_pairs(:,:), d, h, fi, fo
    logical, intent(in) :: use_bias
    class(base_layer_type), allocatable :: layer, reread
    type(graph_type), allocatable :: graph(:)
    type(array_type), allocatable, target :: input(:,:)
    type(array_type), pointer :: loss
    real(real32), allocatable :: params(:), grads(:), ref_grads(:), x(:,:), c(:,:), up(:,:), ref_out(:,:), ref_dx(:,:)
    type(graph_nop_mp_layer_type) :: ref
    type(mp_graph_type), allocatable :: mg(:)
    integer :: batch

    batch = size(sizes)
    if(batch .ne. 1) call fail(case_name//": one sample per case")
    if(stock_only)then
       layer = graph_nop_layer_type(num_outputs=fo, coord_dim=d, kernel_hidden=h, num_inputs=fi, use_bias=use_bias, &
            activation=activation)
    else
       layer = hip_graph_nop_layer_type(num_outputs=fo, coord_dim=d, kernel_hidden=h, num_inputs=fi, use_bias=use_bias, &
            activation=activation)
    end if
    if(layer%name .ne. 'graph_nop') call fail(case_name//": layer name")
    select type(layer)
    class is(graph_nop_layer_type)
       if(layer%coord_dim .ne. d .or. layer%kernel_hidden .ne. h) call fail(case_name//": coord_dim / kernel_hidden")
    class default
       call fail(case_name//": not a graph_nop_layer_type")
    end select

    allocate(graph(1), mg(1), input(2, 1))
    call build_graph(graph(1), sizes(1), hand_pairs, self_loops=.false., num_edge_features=d)
    call to_mp_graph(graph(1), mg(1))
    allocate(x(fi, sizes(1)), c(d, graph(1)%num_edges), up(fo, sizes(1)))
    call fill(x, 51); call fill(c, 52); call fill(up, 53)
    call input(1, 1)%allocate(source=x)
    call input(1, 1)%set_requires_grad(.true.)
    input(1, 1)%is_temporary = .false.
    call input(2, 1)%allocate(source=c)
    input(2, 1)%is_temporary = .false.

    call layer%set_graph(graph)
    select type(layer)
    class is(learnable_layer_type)
       params = layer%get_params()
       if(size(params) .ne. layer%get_num_params()) call fail(case_name//": get_params size")
       call fill_params(params, 7)
       call layer%set_params(params)
       call layer%forward(input)                                  ! forward_msgpass -> update_message_hip_gno
       if(any(shape(layer%output) .ne. [2, 1])) call fail(case_name//": output shape")
       if(layer%output(2, 1)%requires_grad) call fail(case_name//": the forwarded geometry must not be differentiated")
       call close_to(layer%output(2, 1)%val, c, case_name//": forwarded edge geometry")
       loss => weighted_sum(layer%output(1, 1), up)
       call loss%grad_reverse(reset_graph=.true.)
       grads = layer%get_gradients()
    class default
       call fail(case_name//": not learnable")
    end select

    if(.not.stock_only)then
       ref = graph_nop_mp_layer_type(fo, d, fi, kernel_hidden=h, use_bias=use_bias, activation=activation)
       call ref%set_graph(mg)
       call ref%set_params(params)
       ref_out = ref%forward(x, c)
       call ref%backward(up, dx=ref_dx)
       ref_grads = ref%get_gradients()
    end if
    if(.not.stock_only) call close_to(layer%output(1, 1)%val, ref_out, case_name//": output")
    if(.not.stock_only) call close_to(input(1, 1)%grad%val, ref_dx, case_name//": input gradient")
    if(.not.stock_only) call close_to_1d(grads, ref_grads, case_name//": flat parameter gradients (get_gradients)")
    call dump_i(case_name, "ia1", graph(1)%adj_ia)
    call dump_i(case_name, "ja1", graph(1)%adj_ja)
    call dump_r(case_name, "x1", x)
    call dump_r(case_name, "c1", c)
    call dump_r(case_name, "up1", up)
    call dump_r(case_name, "out1", layer%output(1, 1)%val)
    call dump_r(case_name, "dx1", input(1, 1)%grad%val)
    call dump_r(case_name, "params", params)
    call dump_r(case_name, "grads", grads)
    call dump_meta(case_name, [character(16) :: "batch", "d", "h", "fi", "fo", "use_bias"], &
         [1, d, h, fi, fo, merge(1, 0, use_bias)])

    call card_round_trip(layer, "GRAPH_NOP", params, reread, restores_parameters=.true.)
    if(.not.stock_only)then
       select type(reread)
       type is(hip_graph_nop_layer_type)
       class default
          call fail(case_name//": the registry did not hand back a hip_graph_nop_layer_type")
       end select
    end if
    if(.not.stock_only) call ref%destroy()
  end subroutine gno_case

end program run_layers
