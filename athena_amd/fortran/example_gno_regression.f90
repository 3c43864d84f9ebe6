!! The reference's example/gno_regression (src/main.f90) on the HIP path, from Fortran: a 20-vertex chain graph whose
!! edge feature is the coordinate difference of the pair, two stacked graph_nop layers (1 -> 8 relu -> 2, kernel_hidden
!! 8) regressing [sin(x), cos(x)], mean-squared-error loss, plain gradient descent (base_optimiser_type, lr = 0.01),
!! 200 epochs of the single sample.  The layers are chained on the device (forward_dev / backward_dev), the loss and its
!! gradient are athena_mp_mse_loss, the update is minimise_base on the resident parameters: nothing but the inputs,
!! the targets and the printed loss crosses the host boundary.
!!
!!   example_gno_regression [epochs]          prints "Final loss" like the reference example; exit status 1 if the loss did
!!                                            not fall below a tenth of its first value
program example_gno_regression
  use, intrinsic :: iso_c_binding
  use athena_mp_c
  use athena_mp_layers
  implicit none
  integer, parameter :: num_vertices = 20, coord_dim = 1, F_in = 1, F_hidden = 8, F_out = 2
  integer, parameter :: num_edges = num_vertices - 1
  real(real32), parameter :: pi = 4.0_real32 * atan(1.0_real32), lr = 0.01_real32
  integer :: num_epochs, i, epoch
  character(32) :: arg
  integer(c_int32_t) :: index_list(2, num_edges)
  real(real32) :: coords(coord_dim, num_vertices), edge_coords(coord_dim, num_edges)
  real(real32) :: features(F_in, num_vertices), targets(F_out, num_vertices), loss_val(1), first_loss
  type(graph_nop_mp_layer_type) :: layer1, layer2
  type(c_ptr) :: x_dev, c_dev, t_dev, loss_dev, dl_dev, h_dev, y_dev, g1_dev

  num_epochs = 200
  if(command_argument_count() .ge. 1)then
     call get_command_argument(1, arg); read(arg, *) num_epochs
  end if
  if(athena_mp_init(0_c_int) .ne. 0) stop 1
  block   ! fixed seed: the reference example seeds nothing (its numbers vary run to run); this one doubles as a test
    integer :: nseed
    integer, allocatable :: seed(:)
    call random_seed(size=nseed)
    allocate(seed(nseed))
    seed = 20260424
    call random_seed(put=seed)
  end block

  do i = 1, num_edges
     index_list(:, i) = [i, i + 1]
  end do
  do i = 1, num_vertices
     coords(1, i) = real(i - 1, real32) / real(num_vertices - 1, real32) * 2.0_real32 * pi
     features(:, i) = 1.0_real32
     targets(1, i) = sin(coords(1, i))
     targets(2, i) = cos(coords(1, i))
  end do
  do i = 1, num_edges
     edge_coords(:, i) = coords(:, i) - coords(:, i + 1)
  end do

  write(*,*) "Building GNO graph regression network..."
  layer1 = graph_nop_mp_layer_type(num_inputs=F_in, num_outputs=F_hidden, coord_dim=coord_dim, kernel_hidden=8, &
       activation="relu")
  layer2 = graph_nop_mp_layer_type(num_inputs=F_hidden, num_outputs=F_out, coord_dim=coord_dim, kernel_hidden=8)
  call layer1%set_graph_from_edges(num_vertices, index_list)
  call layer2%set_graph_from_edges(num_vertices, index_list)
  write(*,*) "Number of parameters:", layer1%get_num_params() + layer2%get_num_params()

  call must(athena_mp_malloc(x_dev, int(4 * F_in * num_vertices, c_int64_t)))
  call must(athena_mp_malloc(c_dev, int(4 * coord_dim * num_edges, c_int64_t)))
  call must(athena_mp_malloc(t_dev, int(4 * F_out * num_vertices, c_int64_t)))
  call must(athena_mp_malloc(dl_dev, int(4 * F_out * num_vertices, c_int64_t)))
  call must(athena_mp_malloc(loss_dev, 4_c_int64_t))
  call must(athena_mp_memcpy_h2d(x_dev, features, int(4 * F_in * num_vertices, c_int64_t)))
  call must(athena_mp_memcpy_h2d(c_dev, edge_coords, int(4 * coord_dim * num_edges, c_int64_t)))
  call must(athena_mp_memcpy_h2d(t_dev, targets, int(4 * F_out * num_vertices, c_int64_t)))

  write(*,*) "Training..."
  first_loss = 0._real32
  do epoch = 1, num_epochs
     h_dev = layer1%forward_dev(x_dev, c_dev)
     y_dev = layer2%forward_dev(h_dev, c_dev)             ! the edge geometry is forwarded unchanged (output(2,s))
     call must(athena_mp_mse_loss(int(F_out * num_vertices, c_int64_t), y_dev, t_dev, loss_dev, dl_dev))
     call layer2%backward_dev(dl_dev, dx_dev=g1_dev)
     call layer1%backward_dev(g1_dev)
     call layer1%minimise_base(lr)
     call layer2%minimise_base(lr)
     if(epoch .eq. 1 .or. mod(epoch, 20) .eq. 0)then
        call must(athena_mp_memcpy_d2h(loss_val, loss_dev, 4_c_int64_t))
        if(epoch .eq. 1) first_loss = loss_val(1)
        write(*,'("epoch=",I0,", batch=1, lr=",ES8.2,", loss=",F0.3)') epoch, lr, loss_val(1)
     end if
  end do
  call must(athena_mp_memcpy_d2h(loss_val, loss_dev, 4_c_int64_t))
  write(*,*)
  write(*,'(A, F12.6)') " Final loss: ", loss_val(1)
  write(*,*)
  write(*,*) "GNO graph regression example completed."
  call layer1%destroy()
  call layer2%destroy()
  call must(athena_mp_free(x_dev)); call must(athena_mp_free(c_dev)); call must(athena_mp_free(t_dev))
  call must(athena_mp_free(dl_dev)); call must(athena_mp_free(loss_dev))
  if(athena_mp_finalize() .ne. 0) stop 1
  if(.not. (loss_val(1) .lt. 0.1_real32 * first_loss)) stop 1

contains
  subroutine must(rc)
    integer(c_int), intent(in) :: rc
    if(rc .ne. 0)then
       write(0,*) "athena_mp call failed: "//athena_mp_error_message()
       stop 1
    end if
  end subroutine must
end program example_gno_regression
