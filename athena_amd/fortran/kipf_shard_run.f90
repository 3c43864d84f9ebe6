!! One rank of a row-partitioned Kipf layer step driven from FORTRAN through the C ABI (one process per GPU):
!! communicator, shard (the [local | halo] renumbering, send lists, halo degrees and the four row-block graph handles
!! built by athena_mp_shard_create), halo exchange of X and of dZ overlapped with the interior rows, dW all-reduced.
!! What the step replaces on each rank: update_message_kipf's propagate + matmul and their reverse pass
!! (athena_kipf_msgpass_layer.f90:940-957, athena_diffstruc_extd_sub_kipf.f90:29-46, :100-109).
!!
!!   kipf_shard_run <rank> <world> <device> <id-file> <vertices> <pairs> <features> <out-prefix>
!!
!! Every rank generates the same global edge list (a linear congruential stream), keeps its contiguous block of rows,
!! and writes <out-prefix>_r<rank>.bin = n, F, P, Z, dX (original local row order), dW; rank 0 also writes
!! <out-prefix>_problem.bin = N, nnz, F, adj_ia, adj_ja, X, dZ, W for the checker (tests/test_gpu_dist.py holds the
!! results against the CPU oracle on the whole graph).  Transport: RCCL; ATHENA_MP_COMM_TRANSPORT=shm for several ranks
!! on ONE GPU (tests only).
program kipf_shard_run
  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only: real32, int64
  use athena_mp_c
  implicit none
  integer :: rank, world, device, nv, f, i, k, v, u, r0, r1, n, q, unit
  integer(int64) :: pairs, nnz_g, nnz, state, p
  character(512) :: arg, idfile, prefix
  integer(c_int32_t), allocatable :: eu(:), ev(:), ia_g(:), ja_g(:,:), fill(:), ia(:), ja(:,:), order(:)
  real(real32), allocatable :: x(:,:), dz(:,:), w(:), xl(:,:), dzl(:,:), buf(:,:), pp(:,:), zz(:,:), dxx(:,:), dw(:)
  type(c_ptr) :: comm, shard, g(0:3), x_ext, dz_ext, w_dev, p_dev, z_dev, dx_dev, dw_dev
  integer(c_int32_t) :: n_loc, n_int, n_halo
  integer(c_int64_t) :: nnz_s, row_off, n_tot, cnt

  if(command_argument_count() .lt. 8) stop "usage: kipf_shard_run rank world device idfile vertices pairs features prefix"
  call get_command_argument(1, arg); read(arg, *) rank
  call get_command_argument(2, arg); read(arg, *) world
  call get_command_argument(3, arg); read(arg, *) device
  call get_command_argument(4, idfile)
  call get_command_argument(5, arg); read(arg, *) nv
  call get_command_argument(6, arg); read(arg, *) pairs
  call get_command_argument(7, arg); read(arg, *) f
  call get_command_argument(8, prefix)
  call must(athena_mp_init(int(device, c_int)), "init")

  ! ---- the global graph: `pairs` undirected pairs (u /= v) + one self loop per vertex, rows in vertex order --------
  allocate(eu(pairs), ev(pairs))
  state = 20260424_int64
  do p = 1, pairs
     do
        eu(p) = min(nv, 1 + int(next(state) * nv))
        ev(p) = min(nv, 1 + int(next(state) * nv))
        if(eu(p) .ne. ev(p)) exit
     end do
  end do
  nnz_g = 2 * pairs + nv
  allocate(ia_g(nv + 1), ja_g(2, nnz_g), fill(nv))
  ia_g = 0
  do v = 1, nv
     ia_g(v + 1) = 1                                   ! the self loop
  end do
  do p = 1, pairs
     ia_g(eu(p) + 1) = ia_g(eu(p) + 1) + 1
     ia_g(ev(p) + 1) = ia_g(ev(p) + 1) + 1
  end do
  ia_g(1) = 1
  do v = 1, nv
     ia_g(v + 1) = ia_g(v) + ia_g(v + 1)
  end do
  do v = 1, nv
     ja_g(1, ia_g(v)) = v;  ja_g(2, ia_g(v)) = 0       ! self loop first, edge id 0
     fill(v) = ia_g(v) + 1
  end do
  do p = 1, pairs
     u = eu(p); v = ev(p)
     ja_g(1, fill(u)) = v;  ja_g(2, fill(u)) = int(p);  fill(u) = fill(u) + 1
     ja_g(1, fill(v)) = u;  ja_g(2, fill(v)) = int(p);  fill(v) = fill(v) + 1
  end do
  ! ---- inputs, the same on every rank: X, dZ in [-1, 1), W ~ scaled ---------------------------------------------------
  allocate(x(f, nv), dz(f, nv), w(f * f))
  do v = 1, nv
     do i = 1, f
        x(i, v) = 2._real32 * next(state) - 1._real32
     end do
  end do
  do v = 1, nv
     do i = 1, f
        dz(i, v) = 2._real32 * next(state) - 1._real32
     end do
  end do
  do i = 1, f * f
     w(i) = (2._real32 * next(state) - 1._real32) * sqrt(3._real32 * 2._real32 / real(f, real32))
  end do

  ! ---- this rank's contiguous block of rows ----------------------------------------------------------------------------
  r0 = int((int(rank, int64) * nv) / world) + 1
  r1 = int((int(rank + 1, int64) * nv) / world)
  n = r1 - r0 + 1
  nnz = ia_g(r1 + 1) - ia_g(r0)
  allocate(ia(n + 1), ja(2, max(nnz, 1_int64)))
  ia = ia_g(r0:r1 + 1) - ia_g(r0) + 1
  ja(:, 1:nnz) = ja_g(:, ia_g(r0):ia_g(r1 + 1) - 1)     ! adj_ja(1,:) keeps GLOBAL vertex ids

  call must(athena_mp_comm_create_from_file(int(rank, c_int32_t), int(world, c_int32_t), trim(idfile)//c_null_char, comm), &
       "comm_create_from_file")
  call must(athena_mp_shard_create(comm, int(n, c_int32_t), int(nnz, c_int64_t), ia, ja, shard), "shard_create")
  call must(athena_mp_shard_dims(shard, n_loc, n_int, n_halo, nnz_s, row_off, n_tot), "shard_dims")
  if(n_loc .ne. n .or. row_off .ne. r0 - 1 .or. n_tot .ne. nv) stop "shard_dims disagrees with the partition"
  allocate(order(n))
  call must(athena_mp_shard_export(shard, 0_c_int32_t, order, int(n, c_int64_t), cnt), "shard_export(order)")
  do k = 0, 3
     call must(athena_mp_shard_graph(shard, int(k, c_int32_t), g(k)), "shard_graph")
  end do

  ! ---- resident tensors: x_ext / dz_ext = [n local rows (interior first) | n_halo halo rows] ----------------------------
  allocate(xl(f, n), dzl(f, n))
  do k = 1, n
     xl(:, k) = x(:, r0 + order(k))                     ! order(k) = original local id (0-based) of new row k
     dzl(:, k) = dz(:, r0 + order(k))
  end do
  call must(athena_mp_malloc(x_ext, bytes(n + n_halo, f)), "malloc")
  call must(athena_mp_malloc(dz_ext, bytes(n + n_halo, f)), "malloc")
  call must(athena_mp_malloc(p_dev, bytes(n, f)), "malloc")
  call must(athena_mp_malloc(z_dev, bytes(n, f)), "malloc")
  call must(athena_mp_malloc(dx_dev, bytes(n, f)), "malloc")
  call must(athena_mp_malloc(w_dev, bytes(f, f)), "malloc")
  call must(athena_mp_malloc(dw_dev, bytes(f, f)), "malloc")
  call must(athena_mp_memcpy_h2d(x_ext, xl, bytes(n, f)), "h2d")
  call must(athena_mp_memcpy_h2d(dz_ext, dzl, bytes(n, f)), "h2d")
  call must(athena_mp_memcpy_h2d(w_dev, w, bytes(f, f)), "h2d")

  ! ---- forward: halo of X in flight under the interior rows -----------------------------------------------------------
  call must(athena_mp_halo_start(shard, 0_c_int32_t, int(f, c_int32_t), x_ext), "halo_start(X)")
  call must(athena_mp_kipf_layer_fwd(g(0), int(f, c_int32_t), int(f, c_int32_t), x_ext, w_dev, c_null_ptr, 0_c_int32_t, &
       p_dev, z_dev), "kipf_layer_fwd(interior)")
  call must(athena_mp_halo_finish(shard, 0_c_int32_t), "halo_finish(X)")
  call must(athena_mp_kipf_layer_fwd(g(1), int(f, c_int32_t), int(f, c_int32_t), x_ext, w_dev, c_null_ptr, 0_c_int32_t, &
       athena_mp_dev_offset(p_dev, elems(n_int, f)), athena_mp_dev_offset(z_dev, elems(n_int, f))), "kipf_layer_fwd(boundary)")
  ! ---- backward: halo of dZ in flight under dW and the interior rows; dW summed over the ranks -------------------------
  call must(athena_mp_halo_start(shard, 1_c_int32_t, int(f, c_int32_t), dz_ext), "halo_start(dZ)")
  call must(athena_mp_gemm_dw(int(n, c_int64_t), int(f, c_int32_t), int(f, c_int32_t), p_dev, dz_ext, dw_dev), "gemm_dw")
  call must(athena_mp_allreduce_start(comm, dw_dev, int(f, c_int64_t) * f), "allreduce_start")
  call must(athena_mp_pull_gemm(g(2), int(f, c_int32_t), int(f, c_int32_t), dz_ext, w_dev, 0_c_int32_t, dx_dev), &
       "pull_gemm(interior)")
  call must(athena_mp_halo_finish(shard, 1_c_int32_t), "halo_finish(dZ)")
  call must(athena_mp_pull_gemm(g(3), int(f, c_int32_t), int(f, c_int32_t), dz_ext, w_dev, 0_c_int32_t, &
       athena_mp_dev_offset(dx_dev, elems(n_int, f))), "pull_gemm(boundary)")
  call must(athena_mp_allreduce_finish(comm), "allreduce_finish")
  call must(athena_mp_synchronize(), "synchronize")

  ! ---- results back in the ORIGINAL local row order -------------------------------------------------------------------
  allocate(buf(f, n), pp(f, n), zz(f, n), dxx(f, n), dw(f * f))
  call must(athena_mp_memcpy_d2h(buf, p_dev, bytes(n, f)), "d2h");  call unpermute(buf, pp)
  call must(athena_mp_memcpy_d2h(buf, z_dev, bytes(n, f)), "d2h");  call unpermute(buf, zz)
  call must(athena_mp_memcpy_d2h(buf, dx_dev, bytes(n, f)), "d2h"); call unpermute(buf, dxx)
  call must(athena_mp_memcpy_d2h(dw, dw_dev, bytes(f, f)), "d2h")
  write(arg, '(I0)') rank
  open(newunit=unit, file=trim(prefix)//"_r"//trim(arg)//".bin", access="stream", form="unformatted", status="replace")
  write(unit) int(n, c_int32_t), int(f, c_int32_t), int(n_int, c_int32_t), int(n_halo, c_int32_t)
  write(unit) pp, zz, dxx, dw
  close(unit)
  if(rank .eq. 0)then
     open(newunit=unit, file=trim(prefix)//"_problem.bin", access="stream", form="unformatted", status="replace")
     write(unit) int(nv, c_int32_t), int(nnz_g, c_int32_t), int(f, c_int32_t)
     write(unit) ia_g, ja_g, x, dz, w
     close(unit)
  end if
  write(*,'(A,I0,A,I0,A,I0,A,I0,A,I0)') "rank ", rank, ": rows ", n, " interior ", n_int, " halo ", n_halo, " entries ", nnz

  call must(athena_mp_comm_barrier(comm), "barrier")
  call must(athena_mp_shard_destroy(shard), "shard_destroy")
  call must(athena_mp_comm_destroy(comm), "comm_destroy")
  call must(athena_mp_free(x_ext), "free"); call must(athena_mp_free(dz_ext), "free"); call must(athena_mp_free(p_dev), "free")
  call must(athena_mp_free(z_dev), "free"); call must(athena_mp_free(dx_dev), "free"); call must(athena_mp_free(w_dev), "free")
  call must(athena_mp_free(dw_dev), "free")
  if(athena_mp_finalize() .ne. 0) stop 1

contains
  function next(s) result(r)
    !! 48-bit linear congruential stream in [0, 1) (the same on every rank)
    integer(int64), intent(inout) :: s
    real(real32) :: r
    s = iand(s * 25214903917_int64 + 11_int64, 281474976710655_int64)
    r = real(ishft(s, -24), real32) / 16777216._real32
  end function next

  integer(c_int64_t) function bytes(rows, cols)
    integer, intent(in) :: rows, cols
    bytes = 4_c_int64_t * int(max(rows, 1), c_int64_t) * int(cols, c_int64_t)
  end function bytes

  integer(c_int64_t) function elems(rows, cols)
    integer, intent(in) :: rows, cols
    elems = int(rows, c_int64_t) * int(cols, c_int64_t)
  end function elems

  subroutine unpermute(a, b)
    real(real32), intent(in) :: a(:,:)
    real(real32), intent(out) :: b(:,:)
    integer :: kk
    do kk = 1, size(a, 2)
       b(:, order(kk) + 1) = a(:, kk)
    end do
  end subroutine unpermute

  subroutine must(rc, what)
    integer(c_int), intent(in) :: rc
    character(*), intent(in) :: what
    if(rc .ne. 0)then
       write(0,*) what//" failed: "//athena_mp_error_message()
       stop 1
    end if
  end subroutine must
end program kipf_shard_run
