!! Runs one layer of athena_mp_layers on a case file and writes what it computed -- the bridge the
!! parity tests use to hold the FORTRAN host side against the oracle (tests/test_gpu_fortran_layers.py
!! writes the case, runs this program on the GPU box and compares).
!!
!!   athena_mp_layer_run <case.bin> <result.bin>
!!
!! Both files are unformatted streams of int32 / real32 in the order read below.
program athena_mp_layer_run
  use, intrinsic :: iso_c_binding
  use athena_mp_c
  use athena_mp_layers
  implicit none
  character(1024) :: fin, fout
  integer :: uin, uout, kind, batch, s, nv, ne, nnz, nparams, exact_flag
  type(mp_graph_type), allocatable :: graphs(:)
  real(real32), allocatable :: params(:), x(:,:), e(:,:), up(:,:), out(:,:), dx(:,:), de(:,:)
  integer :: dims(2)

  if(command_argument_count() .ne. 2)then
     write(0,*) "usage: athena_mp_layer_run <case.bin> <result.bin>"
     stop 2
  end if
  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(newunit=uin, file=trim(fin), access="stream", form="unformatted", status="old", action="read")
  open(newunit=uout, file=trim(fout), access="stream", form="unformatted", status="replace", action="write")
  if(athena_mp_init(0_c_int) .ne. 0) stop 3

  read(uin) kind, batch
  allocate(graphs(batch))
  do s = 1, batch
     read(uin) nv, ne, nnz
     graphs(s)%num_vertices = nv
     graphs(s)%num_edges = ne
     allocate(graphs(s)%adj_ia(nv + 1), graphs(s)%adj_ja(2, nnz))
     read(uin) graphs(s)%adj_ia
     if(nnz .gt. 0) read(uin) graphs(s)%adj_ja
  end do

  select case(kind)
  case(1)
     call run_kipf()
  case(2)
     call run_duvenaud()
  case(3)
     call run_gno()
  case(4)
     call run_kipf_chain()
  case(5)
     call run_kipf_graph_swap()
  case(6)
     call run_duvenaud(timed=.true.)
  case default
     write(0,*) "unknown layer kind", kind
     stop 4
  end select
  close(uin)
  close(uout)
  if(athena_mp_finalize() .ne. 0) stop 5

contains

  function read_actv() result(a)
    type(mp_actv_type) :: a
    character(16) :: name
    real(real32) :: v(3)
    read(uin) name
    read(uin) v
    a = mp_actv_type(trim(name))          ! validates the name, sets the reference's defaults
    a%scale = v(1)
    a%p0 = v(2)
    a%p1 = v(3)
  end function read_actv

  subroutine read_matrix(a)
    real(real32), allocatable, intent(out) :: a(:,:)
    read(uin) dims
    allocate(a(dims(1), dims(2)))
    if(size(a) .gt. 0) read(uin) a
  end subroutine read_matrix

  subroutine write_matrix(a)
    real(real32), intent(in) :: a(:,:)
    write(uout) int(shape(a), c_int32_t)
    if(size(a) .gt. 0) write(uout) a
  end subroutine write_matrix

  subroutine write_vector(a)
    real(real32), intent(in) :: a(:)
    write(uout) int(size(a), c_int32_t)
    if(size(a) .gt. 0) write(uout) a
  end subroutine write_vector

  subroutine run_kipf()
    type(kipf_mp_layer_type) :: layer
    type(mp_actv_type) :: act
    integer :: t, nf, order_code, need_dx
    integer, allocatable :: nvf(:)
    character(16) :: order

    read(uin) t, nf
    allocate(nvf(nf))
    read(uin) nvf
    act = read_actv()
    read(uin) order_code, exact_flag, need_dx
    order = "auto"
    if(order_code .eq. 1) order = "aggregate_first"
    if(order_code .eq. 2) order = "transform_first"
    layer = kipf_mp_layer_type(num_vertex_features=nvf, num_time_steps=t, activation=act, order=order)
    read(uin) nparams
    allocate(params(nparams))
    read(uin) params
    if(layer%get_num_params() .ne. nparams) stop 6
    call layer%set_params(params)
    call layer%set_graph(graphs)
    call read_matrix(x)
    call read_matrix(up)
    call write_vector(layer%get_gradients())                 ! zeros before any reverse pass
    out = layer%forward(x)
    call write_matrix(out)
    dx = layer%backward(up, exact=exact_flag .eq. 1, need_input_grad=need_dx .eq. 1)
    call write_matrix(dx)
    call write_vector(layer%get_gradients())
    call write_vector(layer%get_params())
    out = layer%forward(x)                                   ! a second pass reproduces the first bit for bit
    call write_matrix(out)
    call layer%destroy()
  end subroutine run_kipf

  subroutine run_duvenaud(timed)
    !! timed: after the parity outputs, `reps` x (forward_dev + backward_dev) on tensors resident in HBM, timed with the
    !! host clock around a device synchronize -- the T-step layer as BASELINE configs[2] states it, driven from Fortran
    logical, intent(in), optional :: timed
    type(duvenaud_mp_layer_type) :: layer
    type(mp_actv_type) :: act, act_r
    integer :: t, fv, fe, mn, mx, nout, reps, k
    type(c_ptr) :: x_dev, e_dev, up_dev, o_dev, dx_dev, de_dev
    integer(c_int64_t) :: c0, c1, rate
    real(c_double) :: ms

    read(uin) t, fv, fe, mn, mx, nout
    act = read_actv()
    act_r = read_actv()
    layer = duvenaud_mp_layer_type(num_vertex_features=[fv], num_edge_features=[fe], num_time_steps=t, &
         max_vertex_degree=mx, num_outputs=nout, min_vertex_degree=mn, message_activation=act, &
         readout_activation=act_r)
    read(uin) nparams
    allocate(params(nparams))
    read(uin) params
    if(layer%get_num_params() .ne. nparams) stop 6
    call layer%set_params(params)
    call layer%set_graph(graphs)
    call read_matrix(x)
    call read_matrix(e)
    call read_matrix(up)
    out = layer%forward(x, e)
    call write_matrix(out)
    call layer%backward(up, dx=dx, de=de)
    call write_matrix(dx)
    call write_matrix(de)
    call write_vector(layer%get_gradients())
    if(present(timed))then
       read(uin) reps
       if(athena_mp_malloc(x_dev, 4_c_int64_t * size(x, kind=c_int64_t)) .ne. 0) stop 7
       if(athena_mp_malloc(e_dev, 4_c_int64_t * max(size(e, kind=c_int64_t), 1_c_int64_t)) .ne. 0) stop 7
       if(athena_mp_malloc(up_dev, 4_c_int64_t * size(up, kind=c_int64_t)) .ne. 0) stop 7
       if(athena_mp_memcpy_h2d(x_dev, x, 4_c_int64_t * size(x, kind=c_int64_t)) .ne. 0) stop 7
       if(size(e) .gt. 0)then
          if(athena_mp_memcpy_h2d(e_dev, e, 4_c_int64_t * size(e, kind=c_int64_t)) .ne. 0) stop 7
       end if
       if(athena_mp_memcpy_h2d(up_dev, up, 4_c_int64_t * size(up, kind=c_int64_t)) .ne. 0) stop 7
       do k = 1, 3
          o_dev = layer%forward_dev(x_dev, e_dev)
          call layer%backward_dev(up_dev, dx_dev=dx_dev, de_dev=de_dev)
       end do
       if(athena_mp_synchronize() .ne. 0) stop 7
       call system_clock(c0, rate)
       do k = 1, reps
          o_dev = layer%forward_dev(x_dev, e_dev)
          call layer%backward_dev(up_dev, dx_dev=dx_dev, de_dev=de_dev)
       end do
       if(athena_mp_synchronize() .ne. 0) stop 7
       call system_clock(c1)
       ms = real(c1 - c0, c_double) / real(rate, c_double) * 1.d3 / real(reps, c_double)
       call write_vector([real(ms, real32)])
       call write_vector(layer%get_gradients())                ! after the timed loop: the same gradients, bit for bit
       if(athena_mp_free(x_dev) .ne. 0 .or. athena_mp_free(e_dev) .ne. 0 .or. athena_mp_free(up_dev) .ne. 0) stop 7
    end if
    call layer%destroy()
  end subroutine run_duvenaud

  subroutine run_gno()
    type(graph_nop_mp_layer_type) :: layer
    type(mp_actv_type) :: act
    integer :: fi, fo, d, h, bias

    read(uin) fi, fo, d, h, bias
    act = read_actv()
    layer = graph_nop_mp_layer_type(num_outputs=fo, coord_dim=d, num_inputs=fi, kernel_hidden=h, &
         use_bias=bias .eq. 1, activation=act)
    read(uin) nparams
    allocate(params(nparams))
    read(uin) params
    if(layer%get_num_params() .ne. nparams) stop 6
    call layer%set_params(params)
    call layer%set_graph(graphs)
    call read_matrix(x)
    call read_matrix(e)
    call read_matrix(up)
    out = layer%forward(x, e)
    call write_matrix(out)
    call layer%backward(up, dx=dx, dcoords=de)
    call write_matrix(dx)
    call write_matrix(de)
    call write_vector(layer%get_gradients())
    call layer%destroy()
  end subroutine run_gno

  subroutine run_kipf_chain()
    !! two Kipf layers chained on the device: the first layer's tape feeds the second (forward_dev), the second's
    !! input gradient feeds the first (backward_dev); only the ends of the chain cross the host boundary
    type(kipf_mp_layer_type) :: l1, l2
    type(mp_actv_type) :: a1, a2
    integer :: f0, f1, f2, n, n1, n2
    real(real32), allocatable :: p1(:), p2(:)
    type(c_ptr) :: x_dev, up_dev, y1, y2, g1, g0

    read(uin) f0, f1, f2
    a1 = read_actv()
    a2 = read_actv()
    l1 = kipf_mp_layer_type(num_vertex_features=[f0, f1], num_time_steps=1, activation=a1)
    l2 = kipf_mp_layer_type(num_vertex_features=[f1, f2], num_time_steps=1, activation=a2)
    read(uin) n1
    allocate(p1(n1)); read(uin) p1
    read(uin) n2
    allocate(p2(n2)); read(uin) p2
    call l1%set_params(p1)
    call l2%set_params(p2)
    call l1%set_graph(graphs)
    call l2%set_graph(graphs)
    call read_matrix(x)
    call read_matrix(up)
    n = size(x, 2)
    if(athena_mp_malloc(x_dev, 4_c_int64_t * size(x, kind=c_int64_t)) .ne. 0) stop 7
    if(athena_mp_malloc(up_dev, 4_c_int64_t * size(up, kind=c_int64_t)) .ne. 0) stop 7
    if(athena_mp_memcpy_h2d(x_dev, x, 4_c_int64_t * size(x, kind=c_int64_t)) .ne. 0) stop 7
    if(athena_mp_memcpy_h2d(up_dev, up, 4_c_int64_t * size(up, kind=c_int64_t)) .ne. 0) stop 7
    y1 = l1%forward_dev(x_dev)
    y2 = l2%forward_dev(y1)
    allocate(out(f2, n))
    if(athena_mp_memcpy_d2h(out, y2, 4_c_int64_t * size(out, kind=c_int64_t)) .ne. 0) stop 7
    call write_matrix(out)
    g1 = l2%backward_dev(up_dev)
    g0 = l1%backward_dev(g1)
    allocate(dx(f0, n))
    if(athena_mp_memcpy_d2h(dx, g0, 4_c_int64_t * size(dx, kind=c_int64_t)) .ne. 0) stop 7
    call write_matrix(dx)
    call write_vector(l1%get_gradients())
    call write_vector(l2%get_gradients())
    if(athena_mp_free(x_dev) .ne. 0) stop 7
    if(athena_mp_free(up_dev) .ne. 0) stop 7
    call l1%destroy()
    call l2%destroy()
  end subroutine run_kipf_chain

  subroutine run_kipf_graph_swap()
    !! set_graph before every forward, as athena_network_sub.f90:2727-2730 does: batch A, batch B (same vertex and entry
    !! counts, different adjacency), A again, A a fourth time, then A's arrays OVERWRITTEN IN PLACE with B's.  Every
    !! forward must see the graph it was given; only real changes may rebuild the device handle.
    type(kipf_mp_layer_type) :: layer, twin
    type(mp_actv_type) :: act
    type(mp_graph_type), allocatable :: other(:)
    integer :: t, nf
    integer, allocatable :: nvf(:)
    integer(c_int64_t) :: handles, hits, builds0, builds1
    integer :: counts(6)

    read(uin) t, nf
    allocate(nvf(nf))
    read(uin) nvf
    act = read_actv()
    layer = kipf_mp_layer_type(num_vertex_features=nvf, num_time_steps=t, activation=act)
    twin = kipf_mp_layer_type(num_vertex_features=nvf, num_time_steps=t, activation=act)
    read(uin) nparams
    allocate(params(nparams))
    read(uin) params
    call layer%set_params(params)
    call twin%set_params(params)
    allocate(other(batch))
    do s = 1, batch
       read(uin) nv, ne, nnz
       other(s)%num_vertices = nv
       other(s)%num_edges = ne
       allocate(other(s)%adj_ia(nv + 1), other(s)%adj_ja(2, nnz))
       read(uin) other(s)%adj_ia
       if(nnz .gt. 0) read(uin) other(s)%adj_ja
    end do
    call read_matrix(x)
    if(athena_mp_graph_cache_stats(handles, hits, builds0) .ne. 0) stop 8

    call layer%set_graph(graphs)
    call write_matrix(layer%forward(x))                      ! A
    counts(1) = layer%graph_builds
    call layer%set_graph(other)
    call write_matrix(layer%forward(x))                      ! B: same sizes, other adjacency
    counts(2) = layer%graph_builds
    call layer%set_graph(graphs)
    call write_matrix(layer%forward(x))                      ! A again
    counts(3) = layer%graph_builds
    call layer%set_graph(graphs)
    call twin%set_graph(graphs)                              ! a second layer on the same batch shares the handle
    call write_matrix(twin%forward(x))
    counts(4) = layer%graph_builds
    do s = 1, batch                                          ! in-place edit: same arrays, B's content
       graphs(s)%adj_ia = other(s)%adj_ia
       graphs(s)%adj_ja = other(s)%adj_ja
    end do
    call layer%set_graph(graphs)
    call write_matrix(layer%forward(x))
    counts(5) = layer%graph_builds
    if(athena_mp_graph_cache_stats(handles, hits, builds1) .ne. 0) stop 8
    counts(6) = int(builds1 - builds0)                       ! device handles actually built: A and B, once each
    write(uout) int(counts, c_int32_t)
    call layer%destroy()
    call twin%destroy()
  end subroutine run_kipf_graph_swap

end program athena_mp_layer_run
