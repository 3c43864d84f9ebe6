!! The headline step driven from FORTRAN: one interior Kipf layer forward + backward on a random graph,
!! through kipf_mp_layer_type's device-pointer methods (tensors resident in HBM, as between consecutive HIP
!! layers of a network).  Graph: `pairs` undirected pairs drawn uniformly (u /= v) + self loops, CSR built on the
!! GPU by athena_mp_graph_create_from_edges -- the shape of BASELINE configs[1], though not bench.py's exact graph
!! (this program uses Fortran's random_number).  bench.py remains the measurement of record.
!!
!!   bench_kipf_layer [vertices] [pairs] [features] [steps]
program bench_kipf_layer
  use, intrinsic :: iso_c_binding
  use athena_mp_c
  use athena_mp_layers
  implicit none
  integer :: n, f, steps, i, k
  integer(c_int64_t) :: pairs, nnz, c0, c1, rate
  character(32) :: arg
  type(kipf_mp_layer_type) :: layer
  integer(c_int32_t), allocatable :: index_list(:,:)
  type(mp_graph_type), allocatable :: graphs(:)
  integer(c_int32_t), allocatable, target :: ja_host(:,:)
  integer(c_int64_t) :: handles, hits, builds0, builds1
  real(c_double) :: ms_set, ms_first
  real(real32), allocatable :: x(:,:), u(:)
  type(c_ptr) :: x_dev, dz_dev, y_dev, dx_dev
  real(real32) :: r(2)
  real(c_double) :: ms

  n = 1000000; pairs = 4500000_c_int64_t; f = 128; steps = 50
  if(command_argument_count() .ge. 1)then
     call get_command_argument(1, arg); read(arg, *) n
  end if
  if(command_argument_count() .ge. 2)then
     call get_command_argument(2, arg); read(arg, *) pairs
  end if
  if(command_argument_count() .ge. 3)then
     call get_command_argument(3, arg); read(arg, *) f
  end if
  if(command_argument_count() .ge. 4)then
     call get_command_argument(4, arg); read(arg, *) steps
  end if
  if(athena_mp_init(0_c_int) .ne. 0) stop 1

  allocate(index_list(2, pairs))
  do i = 1, int(pairs)
     do
        call random_number(r)
        index_list(1, i) = min(n, 1 + int(r(1) * n))
        index_list(2, i) = min(n, 1 + int(r(2) * n))
        if(index_list(1, i) .ne. index_list(2, i)) exit
     end do
  end do
  layer = kipf_mp_layer_type(num_vertex_features=[f], num_time_steps=1, activation="none")
  call system_clock(c0, rate)
  call layer%set_graph_from_edges(n, index_list, add_self_loops=.true.)     ! CSR + handle on the GPU, one call
  call system_clock(c1)
  nnz = 2_c_int64_t * pairs + n
  write(*,'(A,F8.2,A)') "# graph handle from the edge list: ", real(c1 - c0, c_double) / real(rate, c_double) * 1.d3, " ms"
  ! the same graph as a graph_type's host arrays, for the set_graph-before-every-forward measurement below
  allocate(graphs(1))
  graphs(1)%num_vertices = n
  graphs(1)%num_edges = int(pairs)
  allocate(graphs(1)%adj_ia(n + 1), ja_host(2, nnz))
  call must(athena_mp_csr_from_edges(int(n, c_int32_t), pairs, index_list, 1_c_int32_t, graphs(1)%adj_ia, c_loc(ja_host), &
       nnz, c1), "csr_from_edges")
  call move_alloc(ja_host, graphs(1)%adj_ja)
  deallocate(index_list)

  allocate(x(f, n))
  call random_number(x)
  x = 2._real32 * x - 1._real32
  call must(athena_mp_malloc(x_dev, 4_c_int64_t * int(f, c_int64_t) * n), "malloc")
  call must(athena_mp_malloc(dz_dev, 4_c_int64_t * int(f, c_int64_t) * n), "malloc")
  call must(athena_mp_memcpy_h2d(x_dev, x, 4_c_int64_t * int(f, c_int64_t) * n), "h2d")
  call random_number(x)
  call must(athena_mp_memcpy_h2d(dz_dev, x, 4_c_int64_t * int(f, c_int64_t) * n), "h2d")

  do k = 1, 5
     y_dev = layer%forward_dev(x_dev)
     dx_dev = layer%backward_dev(dz_dev)
  end do
  call must(athena_mp_synchronize(), "sync")
  call system_clock(c0, rate)
  do k = 1, steps
     y_dev = layer%forward_dev(x_dev)
     dx_dev = layer%backward_dev(dz_dev)
  end do
  call must(athena_mp_synchronize(), "sync")
  call system_clock(c1)
  ms = real(c1 - c0, c_double) / real(rate, c_double) * 1.d3 / steps

  ! set_graph before EVERY forward, as athena_network_sub.f90:2727-2730 does: the first call builds the handle from the
  ! host arrays, the others cost one content key each (athena_mp_graph_key) and keep it
  call must(athena_mp_graph_cache_stats(handles, hits, builds0), "cache_stats")
  call system_clock(c0, rate)
  call layer%set_graph(graphs)
  call must(athena_mp_synchronize(), "sync")
  call system_clock(c1)
  ms_first = real(c1 - c0, c_double) / real(rate, c_double) * 1.d3
  do k = 1, 3
     call layer%set_graph(graphs)
     y_dev = layer%forward_dev(x_dev)
     dx_dev = layer%backward_dev(dz_dev)
  end do
  call must(athena_mp_synchronize(), "sync")
  call system_clock(c0, rate)
  do k = 1, steps
     call layer%set_graph(graphs)
     y_dev = layer%forward_dev(x_dev)
     dx_dev = layer%backward_dev(dz_dev)
  end do
  call must(athena_mp_synchronize(), "sync")
  call system_clock(c1)
  ms_set = real(c1 - c0, c_double) / real(rate, c_double) * 1.d3 / steps
  call must(athena_mp_graph_cache_stats(handles, hits, builds1), "cache_stats")

  ! one sanity value back on the host: the gradient of the first weight
  u = layer%get_gradients()
  write(*,'(A,I0,A,I0,A,I0,A,I0,A,F9.4,A,ES12.5,A,ES12.5,A,F9.4,A,F9.3,A,I0,A)') &
       '{"driver": "fortran kipf_mp_layer_type%forward_dev/backward_dev", "vertices": ', n, ', "entries": ', nnz, &
       ', "features": ', f, ', "steps": ', steps, ', "ms_per_step": ', ms, ', "edges_per_s": ', &
       real(nnz, c_double) / (ms * 1.d-3), ', "dW(1)": ', u(1), ', "ms_per_step_set_graph_every_forward": ', ms_set, &
       ', "first_set_graph_ms": ', ms_first, ', "handles_built_by_the_set_graph_calls": ', builds1 - builds0, '}'
  call layer%destroy()
  call must(athena_mp_free(x_dev), "free")
  call must(athena_mp_free(dz_dev), "free")
  if(athena_mp_finalize() .ne. 0) stop 1

contains
  subroutine must(rc, what)
    integer(c_int), intent(in) :: rc
    character(*), intent(in) :: what
    if(rc .ne. 0)then
       write(0,*) what//" failed: "//athena_mp_error_message()
       stop 1
    end if
  end subroutine must
end program bench_kipf_layer
