!! LINKED AND RUN on the GPU (tests/test_gpu_integration_run.py): athena's OWN network_type -- add / compile / train / test / print /
!! read, its optimiser, its loss, its input layers and its graph of layers (athena_network_sub.f90, compiled from the reference
!! checkout read in place, over the one stand-in for coreutils / diffstruc / graphstruc) -- with the drop-in layer types in it.
!!
!! Every case is run TWICE through the same code: once with athena's stock layer type (its own host loops, on the stand-in's tape),
!! once with the hip_* type in its place -- `network%add(hip_kipf_msgpass_layer_type(...))` where the program said
!! `kipf_msgpass_layer_type(...)`, nothing else changed -- from the same initial parameters.  Held against each other after
!! training: the loss and accuracy network%test reports, the parameters the optimiser left, the prediction.  The programs
!! are the reference's own in structure: test/test_msgpass_network.f90 (Kipf and Duvenaud networks on its 5-vertex graph, 5 epochs
!! of SGD on an MSE loss), example/gno_regression/src/main.f90 (two stacked graph_nop layers, sin / cos of the coordinate) and
!! example/msgpass_chemical/src/main.f90 (Duvenaud into three full layers: the HIP layer among athena's host layers), plus one batch
!! of larger graphs per family, and a saved network read back through the registry as hip_* layers.
!!
!! `run_network stock` runs the stock halves only (no GPU: what the build container can execute); without an argument it needs
!! libathena_mp.so and a device and prints "RUN_NETWORK_OK <cases> <cases>".
program run_network
  use, intrinsic :: iso_c_binding
  use coreutils, only: real32
  use graphstruc, only: graph_type
  use diffstruc, only: array_type
  use athena__network, only: network_type
  use athena__base_layer, only: base_layer_type
  use athena__kipf_msgpass_layer, only: kipf_msgpass_layer_type
  use athena__duvenaud_msgpass_layer, only: duvenaud_msgpass_layer_type
  use athena__graph_nop_layer, only: graph_nop_layer_type
  use athena__optimiser, only: base_optimiser_type, sgd_optimiser_type, adam_optimiser_type
  use athena__clipper, only: clip_type
  use athena__full_layer, only: full_layer_type
  use athena__hip_msgpass_layers
  use athena_mp_c
  implicit none
  logical :: stock_only
  character(len=32) :: arg
  class(clip_type), allocatable :: clip      ! program scope, as in example/msgpass_chemical/src/main.f90:55 (lives to the end)
  integer :: passed, total

  call get_command_argument(1, arg)
  stock_only = trim(arg) .eq. "stock"
  if(.not.stock_only)then
     if(athena_mp_init(0_c_int) .ne. 0) call fail("athena_mp_init: "//athena_mp_error_message())
     ! `run_network resident`: the residency table on (INTEGRATION.md section 3, phase 2) -- the HIP ops' forward results stay in HBM
     ! between the ops of a layer and are materialised where athena's host code reads them (the layers' flush at the island's edge)
     if(trim(arg) .eq. "resident")then
        if(athena_mp_resident_mode(1_c_int32_t) .ne. 0) call fail("resident_mode: "//athena_mp_error_message())
     end if
  end if
  passed = 0
  total = 0

  ! ---- test/test_msgpass_network.f90: one Kipf layer 8 -> 8 on the 5-vertex graph, the graph itself as target, 5 epochs
  call kipf_case("kipf_reference_test", n_graphs=1, nv=5, nf=[8, 8], steps=1, epochs=5, lr=0.01_real32)
  ! ... two time steps with relu on a batch of four 300-vertex graphs at 64 features (MFMA dense step, fused epilogue)
  call kipf_case("kipf_batch_64", n_graphs=4, nv=300, nf=[64, 64, 64], steps=2, epochs=3, lr=0.002_real32)
  ! ---- the same file's Duvenaud network: 8 vertex / 2 edge features, degrees up to 4, 3 outputs, 'ones', linear readout
  call duvenaud_case("duvenaud_reference_test", n_graphs=1, nv=5, fv=8, fe=2, steps=1, nout=3, mx=4, epochs=5, lr=0.1_real32, &
       reference_form=.true.)
  ! ... the defaults (sigmoid / softmax: the fused route) at F_v 64 / F_e 8 on a batch of three 400-vertex graphs, T = 2
  call duvenaud_case("duvenaud_batch_64", n_graphs=3, nv=400, fv=64, fe=8, steps=2, nout=10, mx=6, epochs=3, lr=1.e-6_real32, &
       reference_form=.false.)
  ! ---- example/msgpass_chemical/src/main.f90:129-196 (BASELINE configs[0]): Duvenaud T = 4 with 10 outputs into three full layers
  ! (128, 64, 1; leaky_relu), Adam with norm clipping, batches of 8 molecule-sized graphs, one target per graph -- the HIP layer
  ! between athena's input layer and athena's host layers
  ! (held at 1e-5 with SGD in the place of the example's Adam; with Adam itself at 1e-3: its g / sqrt(v) turns last-bit differences of
  ! near-zero gradient elements into steps of +- the learning rate, so two correct runs part ways at the 5e-5 level within three epochs)
  call chemical_case("msgpass_chemical_sgd", n_graphs=16, batch=8, epochs=3, adam=.false., tol=1.e-5_real32)
  call chemical_case("msgpass_chemical_example", n_graphs=16, batch=8, epochs=3, adam=.true., tol=1.e-3_real32)
  ! ---- example/gno_regression: two stacked graph_nop layers on a chain of points, targets sin / cos of the coordinate
  call gno_case("gno_regression_example", nv=32, f_hidden=16, kernel_hidden=8, epochs=8, lr=0.01_real32)
  ! ... at 64 hidden features / 64 kernel width (the one-contraction reverse)
  call gno_case("gno_regression_64", nv=600, f_hidden=64, kernel_hidden=64, epochs=3, lr=0.002_real32)

  if(.not.stock_only)then
     if(trim(arg) .eq. "resident")then
        block
          integer(c_int64_t) :: arrays, h2d, d2h, reused, lazy
          if(athena_mp_resident_stats(arrays, h2d, d2h, reused, lazy) .ne. 0) call fail("resident_stats")
          write(*, '(A,I0,A,I0,A,I0,A,I0)') "RESIDENT inputs found in HBM ", reused, ", outputs left in HBM ", lazy, &
               ", bytes up ", h2d, ", bytes down ", d2h
          if(reused .le. 0 .or. lazy .le. 0) call fail("the residency table was on but never used")
        end block
        if(athena_mp_resident_drop(c_null_ptr) .ne. 0) call fail("resident_drop")      ! forget everything, touch no host memory
     end if
     if(athena_mp_finalize() .ne. 0) call fail("finalize")
  end if
  if(passed .ne. total) call fail("some cases failed")
  write(*, '(A,I0,A,I0)') "RUN_NETWORK_OK ", passed, " ", total

contains

  subroutine fail(what)
    character(*), intent(in) :: what
    write(0, '(A)') "run_network: "//what
    error stop 1
  end subroutine fail

  ! ------------------------------------------------------------------------------------------------ graphs
  subroutine ring_graph(graph, nv, nvf, nef, seed)
    !! a connected graph in the reference test's style: a chain 1-2-...-nv plus a chord every third vertex (degrees <= 4);
    !! nv = 5 is exactly test_msgpass_network.f90:268-276
    type(graph_type), intent(inout) :: graph
    integer, intent(in) :: nv, nvf, nef, seed
    integer, allocatable :: index_list(:,:)
    integer :: ne, i, k
    real(real32) :: u
    if(nv .eq. 5)then
       index_list = reshape([1,2, 1,3, 2,3, 2,4, 3,5, 4,5], [2, 6])
    else
       ne = (nv - 1) + (nv - 3) / 3
       allocate(index_list(2, ne))
       do i = 1, nv - 1
          index_list(:, i) = [i, i + 1]
       end do
       k = nv - 1
       do i = 1, nv - 3, 3
          if(k .ge. ne) exit
          k = k + 1
          index_list(:, k) = [i, i + 2]
       end do
    end if
    graph%is_sparse = .true.
    call graph%set_num_vertices(nv, nvf)
    call graph%set_num_edges(size(index_list, 2), nef)
    do i = 1, nv
       do k = 1, nvf
          u = real(modulo(7 * i + 13 * k + 31 * seed, 97), real32) / 97._real32
          graph%vertex_features(k, i) = u
       end do
    end do
    call graph%generate_adjacency(index_list)
    graph%edge_weights = [(1._real32, i = 1, size(index_list, 2))]
    do i = 1, size(index_list, 2)
       do k = 1, nef
          graph%edge_features(k, i) = real(modulo(5 * i + 11 * k + 17 * seed, 89), real32) / 89._real32
       end do
    end do
  end subroutine ring_graph

  ! ------------------------------------------------------------------------------------------------ comparison
  subroutine compare(name, what, a, b, tol)
    character(*), intent(in) :: name, what
    real(real32), dimension(:), intent(in) :: a, b
    real(real32), intent(in) :: tol
    real(real32) :: scale, err
    if(size(a) .ne. size(b)) call fail(name//": "//what//": sizes differ")
    scale = max(maxval(abs(b)), 1.e-30_real32)
    err = maxval(abs(a - b)) / scale
    write(*, '(2X,A,": ",A,T60," rel. deviation ",ES10.3)') name, what, err
    flush(6)
    if(.not.(err .le. tol)) call fail(name//": "//what//" differs between the hip_* and the stock network")
  end subroutine compare

  subroutine train_and_hold(name, stock, hip, x, y, epochs, round_trip, tol)
    !! the reference test's sequence on both networks: train, test, parameters -- then the checkpoint round trip of the hip one
    character(*), intent(in) :: name
    type(network_type), intent(inout) :: stock, hip
    type(graph_type), dimension(:,:), intent(in) :: x
    class(*), dimension(:,:), intent(in) :: y
    integer, intent(in) :: epochs
    logical, intent(in) :: round_trip
    real(real32), intent(in), optional :: tol
    real(real32) :: tol_
    real(real32), allocatable :: p0(:), ps(:), ph(:)
    type(network_type) :: loaded
    character(len=256) :: file
    integer :: l

    total = total + 1
    tol_ = 1.e-5_real32
    if(present(tol)) tol_ = tol
    p0 = stock%get_params()
    call stock%train(x, y, num_epochs=epochs, shuffle_batches=.false., verbose=0)
    call stock%test(x, y)
    ps = stock%get_params()
    write(*, '(A,": stock network  loss ",ES14.6," accuracy ",ES14.6)') name, stock%loss_val, stock%accuracy_val
    flush(6)
    if(stock_only)then
       if(.not.(stock%loss_val .ge. 0._real32)) call fail(name//": stock loss is not a number")
       if(maxval(abs(ps - p0)) .le. 0._real32) call fail(name//": training left the stock parameters untouched")
       passed = passed + 1
       return
    end if
    call hip%set_params(p0)
    call hip%train(x, y, num_epochs=epochs, shuffle_batches=.false., verbose=0)
    call hip%test(x, y)
    ph = hip%get_params()
    write(*, '(A,": hip_* network  loss ",ES14.6," accuracy ",ES14.6)') name, hip%loss_val, hip%accuracy_val
    flush(6)
    if(maxval(abs(ph - p0)) .le. 0._real32) call fail(name//": training left the hip parameters untouched")
    call compare(name, "parameters after training", ph, ps, tol_)
    call compare(name, "loss of network%test", [hip%loss_val], [stock%loss_val], tol_)
    call compare(name, "accuracy of network%test (+ 1)", [hip%accuracy_val + 1._real32], [stock%accuracy_val + 1._real32], tol_)
    if(.not.round_trip)then
       ! (a DUVENAUD card cannot be read back by athena itself: read_duvenaud is an empty body, athena_duvenaud_msgpass_layer.f90:703-714,
       ! so network%read trips over the card's lines -- with the stock type as with the drop-in, which inherits that reader)
       passed = passed + 1
       return
    end if
    ! the trained network saved by athena's writer, read back by athena's reader with the hip readers in front of the registry:
    ! the layers come back as hip_* types and give the same test loss
    write(file, '("run_network_",A,".txt")') name
    call hip%print(file=trim(file))
    call register_hip_msgpass_layers()
    call loaded%read(file=trim(file))
    do l = 1, loaded%num_layers
       select type(layer => loaded%model(l)%layer)
       class is(kipf_msgpass_layer_type)
          select type(layer)
          type is(hip_kipf_msgpass_layer_type)
          class default
             call fail(name//": a kipf card was not read back as hip_kipf_msgpass_layer_type")
          end select
       class is(duvenaud_msgpass_layer_type)
          select type(layer)
          type is(hip_duvenaud_msgpass_layer_type)
          class default
             call fail(name//": a duvenaud card was not read back as hip_duvenaud_msgpass_layer_type")
          end select
       class is(graph_nop_layer_type)
          select type(layer)
          type is(hip_graph_nop_layer_type)
          class default
             call fail(name//": a graph_nop card was not read back as hip_graph_nop_layer_type")
          end select
       end select
    end do
    call loaded%compile(optimiser=base_optimiser_type(learning_rate=0.01_real32), loss_method="mse", accuracy_method="mse", &
         metrics=["loss"], batch_size=hip%batch_size, verbose=0)
    ! the prediction of the network read back against the one that was saved, as example/msgpass_chemical/src/main.f90:216-223 and
    ! :318-335 compare an imported network with its original (cards carry 8 significant digits: E16.8E2)
    call compare(name, "parameters of the network read back", loaded%get_params(), ph, 1.e-6_real32)
    call hip%set_batch_size(1)
    call hip%set_inference_mode()
    call hip%forward(x(:, 1:1))
    call loaded%set_batch_size(1)
    call loaded%set_inference_mode()
    call loaded%forward(x(:, 1:1))
    call compare(name, "prediction of the network read back", &
         reshape(loaded%model(loaded%auto_graph%vertex(loaded%leaf_vertices(1))%id)%layer%output(1, 1)%val, &
         [size(loaded%model(loaded%auto_graph%vertex(loaded%leaf_vertices(1))%id)%layer%output(1, 1)%val)]), &
         reshape(hip%model(hip%auto_graph%vertex(hip%leaf_vertices(1))%id)%layer%output(1, 1)%val, &
         [size(hip%model(hip%auto_graph%vertex(hip%leaf_vertices(1))%id)%layer%output(1, 1)%val)]), 1.e-5_real32)
    open(unit=77, file=trim(file), status="old")
    close(77, status="delete")
    passed = passed + 1
  end subroutine train_and_hold

  ! ------------------------------------------------------------------------------------------------ Kipf
  subroutine kipf_case(name, n_graphs, nv, nf, steps, epochs, lr)
    character(*), intent(in) :: name
    integer, intent(in) :: n_graphs, nv, steps, epochs
    integer, dimension(:), intent(in) :: nf
    real(real32), intent(in) :: lr
    type(network_type) :: stock, hip
    type(graph_type), allocatable :: x(:,:), y(:,:)
    integer :: s
    ! (graph targets go one sample per step, as in the reference's test: with a larger batch athena's loss_eval hands compute_mse the
    ! [vertex | edge] pairs of expected_array through an explicit-shape dummy of predicted's shape -- athena_network_sub.f90:2604-2627,
    ! athena_loss.f90:400-403 -- which pairs sample s with the wrong element for s > 1)
    allocate(x(1, n_graphs), y(1, n_graphs))
    do s = 1, n_graphs
       call ring_graph(x(1, s), nv + 3 * (s - 1), nf(1), 0, s)
       call ring_graph(y(1, s), nv + 3 * (s - 1), nf(size(nf)), 0, s + 50)      ! the target: a graph with the output features
    end do
    if(steps .eq. 1)then
       call stock%add(kipf_msgpass_layer_type(num_vertex_features=nf, num_time_steps=steps))
       if(.not.stock_only) call hip%add(hip_kipf_msgpass_layer_type(num_vertex_features=nf, num_time_steps=steps))
    else
       call stock%add(kipf_msgpass_layer_type(num_vertex_features=nf, num_time_steps=steps, activation="relu"))
       if(.not.stock_only) call hip%add(hip_kipf_msgpass_layer_type(num_vertex_features=nf, num_time_steps=steps, activation="relu"))
    end if
    call stock%compile(optimiser=sgd_optimiser_type(learning_rate=lr), loss_method="mse", accuracy_method="mse", metrics=["loss"], &
         batch_size=1, verbose=0)
    if(stock%num_layers .ne. 2) call fail(name//": wrong number of layers (input + msgpass expected)")
    if(.not.stock_only)then
       call hip%compile(optimiser=sgd_optimiser_type(learning_rate=lr), loss_method="mse", accuracy_method="mse", &
            metrics=["loss"], batch_size=1, verbose=0)
       if(hip%num_layers .ne. 2 .or. hip%get_num_params() .ne. stock%get_num_params()) call fail(name//": the two networks differ")
    end if
    call train_and_hold(name, stock, hip, x, y, epochs, round_trip=.true.)
  end subroutine kipf_case

  ! ------------------------------------------------------------------------------------------------ Duvenaud
  subroutine duvenaud_case(name, n_graphs, nv, fv, fe, steps, nout, mx, epochs, lr, reference_form)
    character(*), intent(in) :: name
    integer, intent(in) :: n_graphs, nv, fv, fe, steps, nout, mx, epochs
    real(real32), intent(in) :: lr
    logical, intent(in) :: reference_form
    type(network_type) :: stock, hip
    type(graph_type), allocatable :: x(:,:)
    type(array_type), allocatable :: y(:,:)
    integer :: s, o
    allocate(x(1, n_graphs))
    do s = 1, n_graphs
       call ring_graph(x(1, s), nv + 3 * (s - 1), fv, fe, s)
    end do
    ! the target: one [num_outputs, batch] array (test_msgpass_network.f90:124-126)
    allocate(y(1, 1))
    call y(1, 1)%allocate(array_shape=[nout, n_graphs])
    do s = 1, n_graphs
       do o = 1, nout
          y(1, 1)%val(o, s) = real(modulo(3 * o + 5 * s, 7), real32) / 7._real32
       end do
    end do
    if(reference_form)then
       call stock%add(duvenaud_msgpass_layer_type(num_vertex_features=[fv], num_edge_features=[fe], num_time_steps=steps, &
            max_vertex_degree=mx, num_outputs=nout, kernel_initialiser='ones', readout_activation='linear'))
       if(.not.stock_only) call hip%add(hip_duvenaud_msgpass_layer_type(num_vertex_features=[fv], num_edge_features=[fe], &
            num_time_steps=steps, max_vertex_degree=mx, num_outputs=nout, kernel_initialiser='ones', readout_activation='linear'))
    else
       call stock%add(duvenaud_msgpass_layer_type(num_vertex_features=[fv], num_edge_features=[fe], num_time_steps=steps, &
            max_vertex_degree=mx, num_outputs=nout))
       if(.not.stock_only) call hip%add(hip_duvenaud_msgpass_layer_type(num_vertex_features=[fv], num_edge_features=[fe], &
            num_time_steps=steps, max_vertex_degree=mx, num_outputs=nout))
    end if
    call stock%compile(optimiser=sgd_optimiser_type(learning_rate=lr), loss_method="mse", accuracy_method="mse", metrics=["loss"], &
         batch_size=n_graphs, verbose=0)
    if(stock%num_layers .ne. 2) call fail(name//": wrong number of layers (input + msgpass expected)")
    if(.not.stock_only)then
       call hip%compile(optimiser=sgd_optimiser_type(learning_rate=lr), loss_method="mse", accuracy_method="mse", &
            metrics=["loss"], batch_size=n_graphs, verbose=0)
       if(hip%get_num_params() .ne. stock%get_num_params()) call fail(name//": the two networks differ")
    end if
    call train_and_hold(name, stock, hip, x, y, epochs, round_trip=.false.)
  end subroutine duvenaud_case

  ! ------------------------------------------------------------------------------------------------ msgpass_chemical
  subroutine chemical_case(name, n_graphs, batch, epochs, adam, tol)
    character(*), intent(in) :: name
    integer, intent(in) :: n_graphs, batch, epochs
    logical, intent(in) :: adam
    real(real32), intent(in) :: tol
    type(network_type) :: stock, hip
    type(graph_type), allocatable :: x(:,:)
    type(array_type), allocatable :: y(:,:)
    integer :: s
    allocate(x(1, n_graphs))
    do s = 1, n_graphs
       call ring_graph(x(1, s), 9 + modulo(5 * s, 17), 6, 1, s)          ! 9 .. 25 atoms, 6 vertex / 1 edge feature as the example's data
       ! (the example also calls add_self_loops, main.f90:107-110; not here: a self-loop entry carries edge id 0 and the reference's
       ! duvenaud_propagate then reads e(:, 0), one column in front of the array -- SURVEY.md F7 -- so what the STOCK network
       ! computes would depend on the heap's contents; the hip_* types and the oracle contribute a zero edge-feature vector there)
    end do
    allocate(y(1, 1))
    call y(1, 1)%allocate(array_shape=[1, n_graphs])
    do s = 1, n_graphs
       y(1, 1)%val(1, s) = real(modulo(3 * s, 11), real32) / 11._real32
    end do
    call build_chemical(name, stock, .false., batch, adam)
    if(.not.stock_only) call build_chemical(name, hip, .true., batch, adam)
    call train_and_hold(name, stock, hip, x, y, epochs, round_trip=.false., tol=tol)
  end subroutine chemical_case

  subroutine build_chemical(name, net, use_hip, batch, adam)
    character(*), intent(in) :: name
    type(network_type), intent(inout) :: net
    logical, intent(in) :: use_hip, adam
    integer, intent(in) :: batch
    if(use_hip)then
       call net%add(hip_duvenaud_msgpass_layer_type(num_time_steps=4, num_vertex_features=[6], num_edge_features=[1], &
            num_outputs=10, kernel_initialiser='glorot_normal', readout_activation='softmax', min_vertex_degree=1, &
            max_vertex_degree=10))
    else
       call net%add(duvenaud_msgpass_layer_type(num_time_steps=4, num_vertex_features=[6], num_edge_features=[1], &
            num_outputs=10, kernel_initialiser='glorot_normal', readout_activation='softmax', min_vertex_degree=1, &
            max_vertex_degree=10))
    end if
    call net%add(full_layer_type(num_inputs=10, num_outputs=128, activation='leaky_relu', kernel_initialiser='he_normal', &
         bias_initialiser='ones'))
    call net%add(full_layer_type(num_outputs=64, activation='leaky_relu', kernel_initialiser='he_normal', bias_initialiser='ones'))
    call net%add(full_layer_type(num_outputs=1, activation='leaky_relu', kernel_initialiser='he_normal', bias_initialiser='ones'))
    if(.not.allocated(clip)) allocate(clip, source=clip_type(clip_norm=1.E-1_real32))
    if(adam)then
       call net%compile(optimiser=adam_optimiser_type(clip_dict=clip, learning_rate=1.E-2_real32), loss_method="mse", &
            accuracy_method="mse", metrics=["loss"], batch_size=batch, verbose=0)
    else
       call net%compile(optimiser=sgd_optimiser_type(clip_dict=clip, learning_rate=1.E-2_real32), loss_method="mse", &
            accuracy_method="mse", metrics=["loss"], batch_size=batch, verbose=0)
    end if
    if(net%num_layers .ne. 5) call fail(name//": wrong number of layers (input + duvenaud + 3 full expected)")
  end subroutine build_chemical

  ! ------------------------------------------------------------------------------------------------ graph neural operator
  subroutine gno_case(name, nv, f_hidden, kernel_hidden, epochs, lr)
    !! example/gno_regression/src/main.f90:60-150
    character(*), intent(in) :: name
    integer, intent(in) :: nv, f_hidden, kernel_hidden, epochs
    real(real32), intent(in) :: lr
    real(real32), parameter :: pi = 3.14159265358979_real32
    type(network_type) :: stock, hip
    type(graph_type), allocatable :: x(:,:), y(:,:)
    integer, allocatable :: index_list(:,:)
    real(real32), allocatable :: coords(:)
    integer :: i
    allocate(index_list(2, nv - 1), coords(nv), x(1, 1), y(1, 1))
    do i = 1, nv - 1
       index_list(:, i) = [i, i + 1]
    end do
    do i = 1, nv
       coords(i) = real(i - 1, real32) / real(nv - 1, real32) * 2._real32 * pi
    end do
    x(1, 1)%is_sparse = .true.
    call x(1, 1)%set_num_vertices(nv, 1)
    call x(1, 1)%set_num_edges(nv - 1, 1)
    x(1, 1)%vertex_features = 1._real32
    call x(1, 1)%generate_adjacency(index_list)
    x(1, 1)%edge_weights = [(1._real32, i = 1, nv - 1)]
    do i = 1, nv - 1
       x(1, 1)%edge_features(1, i) = coords(i) - coords(i + 1)
    end do
    y(1, 1) = x(1, 1)
    call y(1, 1)%set_num_vertices(nv, 2)
    do i = 1, nv
       y(1, 1)%vertex_features(:, i) = [sin(coords(i)), cos(coords(i))]
    end do
    call stock%add(graph_nop_layer_type(num_inputs=1, num_outputs=f_hidden, coord_dim=1, kernel_hidden=kernel_hidden, &
         activation="relu"))
    call stock%add(graph_nop_layer_type(num_outputs=2, coord_dim=1, kernel_hidden=kernel_hidden))
    call stock%compile(optimiser=base_optimiser_type(learning_rate=lr), loss_method="mse", accuracy_method="mse", metrics=["loss"], &
         verbose=0)
    call stock%set_batch_size(1)
    if(.not.stock_only)then
       call hip%add(hip_graph_nop_layer_type(num_inputs=1, num_outputs=f_hidden, coord_dim=1, kernel_hidden=kernel_hidden, &
            activation="relu"))
       call hip%add(hip_graph_nop_layer_type(num_outputs=2, coord_dim=1, kernel_hidden=kernel_hidden))
       call hip%compile(optimiser=base_optimiser_type(learning_rate=lr), loss_method="mse", accuracy_method="mse", &
            metrics=["loss"], verbose=0)
       call hip%set_batch_size(1)
       if(hip%get_num_params() .ne. stock%get_num_params()) call fail(name//": the two networks differ")
    end if
    call train_and_hold(name, stock, hip, x, y, epochs, round_trip=.true.)
  end subroutine gno_case

end program run_network
