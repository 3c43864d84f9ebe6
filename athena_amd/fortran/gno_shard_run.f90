!! One rank of graph_nop_layer's forward + reverse pass on ONE mesh cut by rows, driven from FORTRAN through the C ABI (one
!! process per GPU; SURVEY.md 8e "GNO: as Kipf plus replicated theta and all-reduce of dtheta"):
!!   shard     athena_mp_shard_create_edges -- the rank's rows of graph_type%adj_ia / adj_ja with GLOBAL vertex ids in
!!             adj_ja(1,:) and GLOBAL edge ids in adj_ja(2,:); the shard renumbers both ([local | halo] vertices, the rank's
!!             own set of edge columns) and checks across all ranks that the two directions of a pair share one column;
!!   forward   halo exchange of x under the interior rows of gno_aggregate; out = m + W x + b
!!             (update_message_gno, athena_graph_nop_layer.f90:690-788; activation none);
!!   reverse   halo exchange of dz under db, dW, dtheta (local rows of dz only) and the interior rows of the pull
!!             dx_v = sum_{(u,e) in row v} K_e^T dz_u (athena_diffstruc_extd_sub_nop.f90:419-458 read from the receiving
!!             side); [dtheta | dW | db] summed over the ranks in ONE all-reduce.
!!
!!   gno_shard_run <rank> <world> <device> <id-file> <problem-file> <out-prefix> [pull|reduce]
!!
!! reverse = pull (default): as above.  reverse = reduce: NO exchange of dz -- athena_mp_gno_aggregate_bwd on each forward block
!! gives dtheta and the scatter-form dx of every column the block's rows touch from ONE contraction; the rows computed for
!! remote vertices travel to their owners and are added there (athena_mp_halo_reduce_start / _finish), under the interior block.
!!
!! <problem-file> (written by the checker, read by every rank): int32 N, nnz, E, Fi, Fo, d, H; adj_ia(N+1); adj_ja(2,nnz);
!! coords(d,E); x(Fi,N); up(Fo,N); theta; W(Fo*Fi); b(Fo).  Each rank writes <out-prefix>_r<rank>.bin = n, n_int, n_halo,
!! n_edge_cols, out(Fo,n), dx(Fi,n) (original local row order), grads [dtheta | dW | db].  tests/test_gpu_dist.py holds the
!! assembled results against the materialising CPU oracle on the whole mesh.  Transport: RCCL; ATHENA_MP_COMM_TRANSPORT=shm
!! for several ranks on ONE GPU (tests only).
program gno_shard_run
  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only: real32, int64
  use athena_mp_c
  implicit none
  integer :: rank, world, device, unit, k, r0, r1, n, nth, ng
  integer(c_int32_t) :: nv, nnz_g, ne_g, fi, fo, d, h
  integer(int64) :: nnz
  character(512) :: arg, idfile, problem, prefix, revmode
  logical :: reduce_form
  integer(c_int32_t) :: fused
  type(c_ptr) :: dxb_dev, dxi_dev
  integer(c_int32_t), allocatable :: ia_g(:), ja_g(:,:), ia(:), ja(:,:), order(:)
  integer(c_int64_t), allocatable :: edge_ids(:)
  real(real32), allocatable :: coords(:,:), x(:,:), up(:,:), theta(:), w(:), b(:), xl(:,:), upl(:,:), cl(:,:)
  real(real32), allocatable :: buf(:,:), outv(:,:), dxv(:,:), grads(:), ones(:)
  type(c_ptr) :: comm, shard, g(0:3), x_ext, g_ext, th_dev, w_dev, b_dev, c_dev, m_dev, z_dev, dx_dev, t_dev, grad_dev, ones_dev, &
       dth2_dev, s_dev(0:1)
  integer(c_int32_t) :: n_loc, n_int, n_halo, n_ec
  integer(c_int64_t) :: nnz_s, row_off, n_tot, cnt, s_bytes(0:1)

  if(command_argument_count() .lt. 6) stop "usage: gno_shard_run rank world device idfile problem prefix"
  call get_command_argument(1, arg); read(arg, *) rank
  call get_command_argument(2, arg); read(arg, *) world
  call get_command_argument(3, arg); read(arg, *) device
  call get_command_argument(4, idfile)
  call get_command_argument(5, problem)
  call get_command_argument(6, prefix)
  revmode = "pull"
  if(command_argument_count() .ge. 7) call get_command_argument(7, revmode)
  reduce_form = trim(revmode) .eq. "reduce"
  call must(athena_mp_init(int(device, c_int)), "init")

  open(newunit=unit, file=trim(problem), access="stream", form="unformatted", status="old")
  read(unit) nv, nnz_g, ne_g, fi, fo, d, h
  nth = h * d + h + fo * fi * h + fo * fi
  allocate(ia_g(nv + 1), ja_g(2, nnz_g), coords(d, ne_g), x(fi, nv), up(fo, nv), theta(nth), w(fo * fi), b(fo))
  read(unit) ia_g, ja_g, coords, x, up, theta, w, b
  close(unit)

  ! ---- this rank's contiguous block of rows: GLOBAL vertex ids and GLOBAL edge ids stay as graph_type holds them --------
  r0 = int((int(rank, int64) * nv) / world) + 1
  r1 = int((int(rank + 1, int64) * nv) / world)
  n = r1 - r0 + 1
  nnz = ia_g(r1 + 1) - ia_g(r0)
  allocate(ia(n + 1), ja(2, max(nnz, 1_int64)))
  ia = ia_g(r0:r1 + 1) - ia_g(r0) + 1
  ja(:, 1:nnz) = ja_g(:, ia_g(r0):ia_g(r1 + 1) - 1)

  call must(athena_mp_comm_create_from_file(int(rank, c_int32_t), int(world, c_int32_t), trim(idfile)//c_null_char, comm), &
       "comm_create_from_file")
  call must(athena_mp_shard_create_edges(comm, int(n, c_int32_t), int(nnz, c_int64_t), ia, ja, shard), "shard_create_edges")
  call must(athena_mp_shard_dims(shard, n_loc, n_int, n_halo, nnz_s, row_off, n_tot), "shard_dims")
  if(n_loc .ne. n .or. row_off .ne. r0 - 1 .or. n_tot .ne. nv) stop "shard_dims disagrees with the partition"
  call must(athena_mp_shard_edge_cols(shard, n_ec), "shard_edge_cols")
  allocate(order(n), edge_ids(max(n_ec, 1)))
  call must(athena_mp_shard_export(shard, 0_c_int32_t, order, int(n, c_int64_t), cnt), "shard_export(order)")
  call must(athena_mp_shard_export(shard, 7_c_int32_t, edge_ids, int(n_ec, c_int64_t), cnt), "shard_export(edge ids)")
  if(cnt .ne. n_ec) stop "shard_export(7) disagrees with shard_edge_cols"
  do k = 0, 3
     call must(athena_mp_shard_graph(shard, int(k, c_int32_t), g(k)), "shard_graph")
  end do

  ! ---- resident tensors: rows interior first, the edge geometry of the rank's own columns -------------------------------
  allocate(xl(fi, n), upl(fo, n), cl(d, max(n_ec, 1)))
  do k = 1, n
     xl(:, k) = x(:, r0 + order(k))
     upl(:, k) = up(:, r0 + order(k))
  end do
  do k = 1, n_ec
     cl(:, k) = coords(:, edge_ids(k) + 1)              ! edge_ids: 0-based global column of local column k
  end do
  ng = nth + fo * fi + fo
  call must(athena_mp_malloc(x_ext, bytes(n + n_halo, fi)), "malloc")
  call must(athena_mp_malloc(g_ext, bytes(n + n_halo, fo)), "malloc")
  call must(athena_mp_malloc(th_dev, bytes(nth, 1)), "malloc")
  call must(athena_mp_malloc(w_dev, bytes(fo, fi)), "malloc")
  call must(athena_mp_malloc(b_dev, bytes(fo, 1)), "malloc")
  call must(athena_mp_malloc(c_dev, bytes(max(n_ec, 1), d)), "malloc")
  call must(athena_mp_malloc(m_dev, bytes(n, fo)), "malloc")
  call must(athena_mp_malloc(z_dev, bytes(n, fo)), "malloc")
  call must(athena_mp_malloc(dx_dev, bytes(n, fi)), "malloc")
  call must(athena_mp_malloc(t_dev, bytes(n, fi)), "malloc")
  call must(athena_mp_malloc(grad_dev, bytes(ng, 1)), "malloc")
  call must(athena_mp_malloc(dth2_dev, bytes(nth, 1)), "malloc")
  call must(athena_mp_malloc(ones_dev, bytes(n, 1)), "malloc")
  allocate(ones(max(n, 1)))
  ones = 1._real32
  call must(athena_mp_memcpy_h2d(ones_dev, ones, bytes(n, 1)), "h2d")
  call must(athena_mp_memcpy_h2d(x_ext, xl, bytes(n, fi)), "h2d")
  call must(athena_mp_memcpy_h2d(g_ext, upl, bytes(n, fo)), "h2d")            ! activation none: dz = upstream
  call must(athena_mp_memcpy_h2d(th_dev, theta, bytes(nth, 1)), "h2d")
  call must(athena_mp_memcpy_h2d(w_dev, w, bytes(fo, fi)), "h2d")
  call must(athena_mp_memcpy_h2d(b_dev, b, bytes(fo, 1)), "h2d")
  if(n_ec .gt. 0) call must(athena_mp_memcpy_h2d(c_dev, cl, bytes(n_ec, d)), "h2d")
  ! the training-mode forward keeps S per block where the shape takes the kernels that do (DESIGN.md 3.5)
  do k = 0, 1
     s_dev(k) = c_null_ptr
     call must(athena_mp_gno_saved_bytes(g(k), d, h, fi, fo, s_bytes(k)), "gno_saved_bytes")
     if(s_bytes(k) .gt. 0) call must(athena_mp_malloc(s_dev(k), s_bytes(k)), "malloc(S)")
  end do

  ! ---- forward: halo of x in flight under the interior rows and the bypass W x + b --------------------------------------
  call must(athena_mp_halo_start(shard, 0_c_int32_t, fi, x_ext), "halo_start(x)")
  call aggregate(0, m_dev)
  call must(athena_mp_gemm_fwd(int(n, c_int64_t), fi, fo, x_ext, w_dev, b_dev, ATHENA_MP_ACT_NONE, z_dev), "gemm_fwd")
  call must(athena_mp_halo_finish(shard, 0_c_int32_t), "halo_finish(x)")
  call aggregate(1, athena_mp_dev_offset(m_dev, elems(n_int, fo)))
  call must(athena_mp_axpy(elems(n, fo), 1._c_float, m_dev, z_dev), "axpy")                   ! out = m + W x + b

  if(reduce_form)then
     ! ---- reverse, scatter form: no exchange of dz; dx rows computed for remote vertices are reduced at their owners ------
     call must(athena_mp_malloc(dxb_dev, bytes(n + n_halo, fi)), "malloc")
     call must(athena_mp_malloc(dxi_dev, bytes(n + n_halo, fi)), "malloc")
     call must(athena_mp_gemm_dw(int(n, c_int64_t), 1_c_int32_t, fo, ones_dev, g_ext, athena_mp_dev_offset(grad_dev, &
          int(nth + fo * fi, c_int64_t))), "gemm_dw(db)")
     call must(athena_mp_gemm_dw(int(n, c_int64_t), fi, fo, x_ext, g_ext, athena_mp_dev_offset(grad_dev, int(nth, c_int64_t))), &
          "gemm_dw(dW)")
     if(n_int .lt. n)then                               ! the boundary block first: only its rows touch remote columns
        call must(athena_mp_gno_aggregate_bwd(g(1), d, h, fi, fo, th_dev, c_dev, x_ext, athena_mp_dev_offset(g_ext, elems(n_int, fo)), &
             s_dev(1), dxb_dev, grad_dev, c_null_ptr, fused), "gno_aggregate_bwd(boundary)")
     else
        call must(athena_mp_memset_zero(dxb_dev, bytes(n + n_halo, fi)), "memset")
        call must(athena_mp_memset_zero(grad_dev, bytes(nth, 1)), "memset")
     end if
     call must(athena_mp_halo_reduce_start(shard, 1_c_int32_t, fi, dxb_dev), "halo_reduce_start")
     if(n_int .gt. 0)then                               ! the interior block under the transfer
        call must(athena_mp_gno_aggregate_bwd(g(0), d, h, fi, fo, th_dev, c_dev, x_ext, g_ext, s_dev(0), dxi_dev, dth2_dev, &
             c_null_ptr, fused), "gno_aggregate_bwd(interior)")
        call must(athena_mp_axpy(int(nth, c_int64_t), 1._c_float, dth2_dev, grad_dev), "axpy(dtheta)")
     end if
     call must(athena_mp_allreduce_start(comm, grad_dev, int(ng, c_int64_t)), "allreduce_start")
     call must(athena_mp_gemm_dx(int(n, c_int64_t), fi, fo, g_ext, w_dev, dx_dev), "gemm_dx")
     call must(athena_mp_axpy(elems(n, fi), 1._c_float, dxb_dev, dx_dev), "axpy(dx boundary block)")
     if(n_int .gt. 0) call must(athena_mp_axpy(elems(n, fi), 1._c_float, dxi_dev, dx_dev), "axpy(dx interior block)")
     call must(athena_mp_halo_reduce_finish(shard, 1_c_int32_t, dx_dev), "halo_reduce_finish")
     call must(athena_mp_allreduce_finish(comm), "allreduce_finish")
     call must(athena_mp_synchronize(), "synchronize")
  else
  ! ---- reverse: halo of dz in flight under everything that needs its LOCAL rows only -------------------------------------
  call must(athena_mp_halo_start(shard, 1_c_int32_t, fo, g_ext), "halo_start(dz)")
  call must(athena_mp_gemm_dw(int(n, c_int64_t), 1_c_int32_t, fo, ones_dev, g_ext, athena_mp_dev_offset(grad_dev, &
       int(nth + fo * fi, c_int64_t))), "gemm_dw(db)")
  call must(athena_mp_gemm_dw(int(n, c_int64_t), fi, fo, x_ext, g_ext, athena_mp_dev_offset(grad_dev, int(nth, c_int64_t))), &
       "gemm_dw(dW)")
  call dtheta_block(0, g_ext, grad_dev)
  call dtheta_block(1, athena_mp_dev_offset(g_ext, elems(n_int, fo)), dth2_dev)
  call must(athena_mp_axpy(int(nth, c_int64_t), 1._c_float, dth2_dev, grad_dev), "axpy(dtheta)")
  call must(athena_mp_allreduce_start(comm, grad_dev, int(ng, c_int64_t)), "allreduce_start")
  call must(athena_mp_gno_aggregate_bwd_x_pull(g(2), d, h, fi, fo, th_dev, c_dev, g_ext, dx_dev), "bwd_x_pull(interior)")
  call must(athena_mp_halo_finish(shard, 1_c_int32_t), "halo_finish(dz)")
  call must(athena_mp_gno_aggregate_bwd_x_pull(g(3), d, h, fi, fo, th_dev, c_dev, g_ext, &
       athena_mp_dev_offset(dx_dev, elems(n_int, fi))), "bwd_x_pull(boundary)")
  call must(athena_mp_gemm_dx(int(n, c_int64_t), fi, fo, g_ext, w_dev, t_dev), "gemm_dx")
  call must(athena_mp_axpy(elems(n, fi), 1._c_float, t_dev, dx_dev), "axpy(dx)")
  call must(athena_mp_allreduce_finish(comm), "allreduce_finish")
  call must(athena_mp_synchronize(), "synchronize")
  end if

  ! ---- results back in the ORIGINAL local row order -------------------------------------------------------------------
  allocate(outv(fo, n), dxv(fi, n), grads(ng))
  allocate(buf(fo, n))
  call must(athena_mp_memcpy_d2h(buf, z_dev, bytes(n, fo)), "d2h");  call unpermute(buf, outv)
  deallocate(buf); allocate(buf(fi, n))
  call must(athena_mp_memcpy_d2h(buf, dx_dev, bytes(n, fi)), "d2h"); call unpermute(buf, dxv)
  call must(athena_mp_memcpy_d2h(grads, grad_dev, bytes(ng, 1)), "d2h")
  write(arg, '(I0)') rank
  open(newunit=unit, file=trim(prefix)//"_r"//trim(arg)//".bin", access="stream", form="unformatted", status="replace")
  write(unit) int(n, c_int32_t), n_int, n_halo, n_ec
  write(unit) outv, dxv, grads
  close(unit)
  write(*,'(A,I0,A,I0,A,I0,A,I0,A,I0,A,L1)') "rank ", rank, ": rows ", n, " interior ", n_int, " halo ", n_halo, &
       " edge columns ", n_ec, " S kept ", (s_bytes(0) + s_bytes(1) .gt. 0)

  call must(athena_mp_comm_barrier(comm), "barrier")
  call must(athena_mp_shard_destroy(shard), "shard_destroy")
  call must(athena_mp_comm_destroy(comm), "comm_destroy")
  if(athena_mp_finalize() .ne. 0) stop 1

contains
  subroutine aggregate(k, m_blk)
    !! gno_aggregate of row block k (0 interior, 1 boundary) into its rows of m; keeps S when the shape allows
    integer, intent(in) :: k
    type(c_ptr), intent(in) :: m_blk
    if((k .eq. 0 .and. n_int .eq. 0) .or. (k .eq. 1 .and. n_int .eq. n)) return
    if(s_bytes(k) .gt. 0)then
       call must(athena_mp_gno_aggregate_fwd_save(g(k), d, h, fi, fo, th_dev, c_dev, x_ext, m_blk, s_dev(k)), "gno_aggregate (S kept)")
    else
       call must(athena_mp_gno_aggregate_fwd(g(k), d, h, fi, fo, th_dev, c_dev, x_ext, m_blk), "gno_aggregate")
    end if
  end subroutine aggregate

  subroutine dtheta_block(k, dz_blk, dst)
    integer, intent(in) :: k
    type(c_ptr), intent(in) :: dz_blk, dst
    if((k .eq. 0 .and. n_int .eq. 0) .or. (k .eq. 1 .and. n_int .eq. n))then
       call must(athena_mp_memset_zero(dst, bytes(nth, 1)), "memset")
    else if(s_bytes(k) .gt. 0)then
       call must(athena_mp_gno_aggregate_bwd_theta_saved(g(k), d, h, fi, fo, th_dev, c_dev, x_ext, dz_blk, s_dev(k), dst), &
            "gno_aggregate_bwd_theta (S kept)")
    else
       call must(athena_mp_gno_aggregate_bwd_theta(g(k), d, h, fi, fo, th_dev, c_dev, x_ext, dz_blk, dst), "gno_aggregate_bwd_theta")
    end if
  end subroutine dtheta_block

  integer(c_int64_t) function bytes(rows, cols)
    integer, intent(in) :: rows, cols
    bytes = 4_c_int64_t * int(max(rows, 1), c_int64_t) * int(max(cols, 1), c_int64_t)
  end function bytes

  integer(c_int64_t) function elems(rows, cols)
    integer, intent(in) :: rows, cols
    elems = int(rows, c_int64_t) * int(cols, c_int64_t)
  end function elems

  subroutine unpermute(a, bb)
    real(real32), intent(in) :: a(:,:)
    real(real32), intent(out) :: bb(:,:)
    integer :: kk
    do kk = 1, size(a, 2)
       bb(:, order(kk) + 1) = a(:, kk)
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
end program gno_shard_run
