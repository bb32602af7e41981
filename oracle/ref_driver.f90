! TEST INFRASTRUCTURE ONLY (oracle/): driver that runs the REAL reference -- GALAHAD SLS with the
! SPRAL-SSIDS CPU backend (or the LAPACK sytr/potr/pbtr arms) built by oracle/build_ref.sh -- on a
! problem file written by oracle/refio.py, and writes results back as a flat binary file.
! It is the parity checker's ground truth and the "reference" leg of bench.py's cpu_baseline.
! It is never linked into, or called by, the product under galahad_amd/.
!
!   usage: ref_driver <problem.bin> <result.bin>
!
! problem.bin (native endian, stream):
!   int32  magic(=1397509959 'GSLS') version(=1)
!   int32  n ne nrhs solver(0 ssids,1 sytr,2 potr,3 pbtr) pivot_control max_refine nemin
!          have_perm repeat dump_struct scaling ordering   (solver 4 = gsls, drop-in build only)
!   real64 relative_pivot_tolerance absolute_pivot_tolerance
!   int32  row(ne) col(ne) ; real64 val(ne) ; int32 perm(n) [if have_perm] ; real64 rhs(n*nrhs)
! result.bin:
!   int32  status_analyse status_factorize status_solve
!   int64  entries_in_factors flops_elimination
!   int32  rank negative_eigenvalues two_by_two delayed max_front max_depth num_sup spare
!   real64 t_analyse t_factorize_best t_solve_best t_factorize_median t_solve_median
!   real64 x(n*nrhs)
!   [dump_struct: direct SSIDS_analyse on the lower CSC SLS would hand over]
!   int32  flag nnodes ; int32 sptr(nnodes+1) sparent(nnodes) ; int64 rptr(nnodes+1) ;
!   int32  rlist(rptr(nnodes+1)-1) ; int32 order(n) ; int64 num_factor num_flops ;
!   int64  nptr(nnodes+1) ; int64 nlist(2, nz)
program gsls_ref_driver
  use GALAHAD_SLS_double
  use spral_ssids, only : ssids_akeep, ssids_options, ssids_inform, ssids_analyse
  implicit none
  integer, parameter :: wp = kind(1.0d0), long = selected_int_kind(18)
  type(SMT_type) :: matrix
  type(SLS_data_type) :: data
  type(SLS_control_type) :: control
  type(SLS_inform_type) :: inform
  character(len=1024) :: fin, fout
  character(len=8) :: solver
  integer :: magic, version, n, ne, nrhs, isolver, pivot_control, max_refine, nemin
  integer :: have_perm, repeat, dump_struct, scaling, ordering
  real(wp) :: rpt, apt
  integer, allocatable :: perm(:)
  real(wp), allocatable :: rhs(:,:), x(:,:), tf(:), ts(:)
  integer :: s, u, k, st_a, st_f, st_s, i
  real(wp) :: t0, t1, t_an
  integer(long) :: c0, c1, crate

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read(u) magic, version
  if (magic /= 1397509959 .or. version /= 1) stop 'ref_driver: bad problem file'
  read(u) n, ne, nrhs, isolver, pivot_control, max_refine, nemin, have_perm, repeat, &
          dump_struct, scaling, ordering
  read(u) rpt, apt
  call SMT_put(matrix%type, 'COORDINATE', s)
  matrix%n = n ; matrix%ne = ne
  allocate(matrix%row(ne), matrix%col(ne), matrix%val(ne), perm(n), rhs(n,nrhs), x(n,nrhs))
  read(u) matrix%row
  read(u) matrix%col
  read(u) matrix%val
  if (have_perm /= 0) read(u) perm
  read(u) rhs
  close(u)

  select case (isolver)
  case (0) ; solver = 'ssids'
  case (1) ; solver = 'sytr'
  case (2) ; solver = 'potr'
  case (3) ; solver = 'pbtr'
  case (4) ; solver = 'gsls'   ! only in the drop-in build (integration/patch_sls.py)
  case default ; stop 'ref_driver: bad solver id'
  end select

  call SLS_initialize(trim(solver), data, control, inform)
  control%pivot_control = pivot_control
  control%max_iterative_refinements = max_refine
  control%acceptable_residual_relative = 0.0_wp
  control%acceptable_residual_absolute = 0.0_wp
  if (nemin > 0) control%node_amalgamation = nemin
  if (rpt >= 0.0_wp) control%relative_pivot_tolerance = rpt
  if (apt >= 0.0_wp) control%absolute_pivot_tolerance = apt
  if (scaling /= 0) control%scaling = scaling
  if (ordering /= -999) control%ordering = ordering

  call system_clock(c0, crate)
  if (have_perm /= 0) then
    call SLS_analyse(matrix, data, control, inform, PERM=perm)
  else
    call SLS_analyse(matrix, data, control, inform)
  end if
  call system_clock(c1)
  t_an = real(c1 - c0, wp) / real(crate, wp)
  st_a = inform%status ; st_f = -999 ; st_s = -999
  repeat = max(repeat, 1)
  allocate(tf(repeat), ts(repeat))
  tf = 0.0_wp ; ts = 0.0_wp
  x = 0.0_wp
  if (st_a >= 0) then
    do k = 1, repeat
      call system_clock(c0)
      call SLS_factorize(matrix, data, control, inform)
      call system_clock(c1)
      tf(k) = real(c1 - c0, wp) / real(crate, wp)
      st_f = inform%status
      if (st_f < 0) exit
      x = rhs
      call system_clock(c0)
      if (nrhs == 1) then
        call SLS_solve(matrix, x(:,1), data, control, inform)
      else
        call SLS_solve(matrix, x, data, control, inform)
      end if
      call system_clock(c1)
      ts(k) = real(c1 - c0, wp) / real(crate, wp)
      st_s = inform%status
    end do
  end if

  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(u) st_a, st_f, st_s
  write(u) int(inform%entries_in_factors, long), int(inform%flops_elimination, long)
  write(u) inform%rank, inform%negative_eigenvalues, inform%two_by_two_pivots, &
           inform%delayed_pivots, inform%max_front_size, inform%max_depth_assembly_tree, &
           inform%ssids_inform%num_sup, 0
  write(u) t_an, minval(tf), minval(ts), median(tf), median(ts)
  write(u) x
  if (dump_struct /= 0) call dump_structure(u)
  close(u)
  call SLS_terminate(data, control, inform)

contains

  real(wp) function median(v)
    real(wp), intent(in) :: v(:)
    real(wp) :: w(size(v)), t
    integer :: a, b
    w = v
    do a = 2, size(w)
      t = w(a) ; b = a - 1
      do while (b >= 1)
        if (w(b) <= t) exit
        w(b+1) = w(b) ; b = b - 1
      end do
      w(b+1) = t
    end do
    median = w((size(w)+1)/2)
  end function median

  ! Build the lower-triangle CSC (sorted rows, explicit diagonal, duplicates merged) that SLS hands
  ! to SSIDS, call SSIDS_analyse directly with the supplied order, dump akeep's symbolic arrays.
  subroutine dump_structure(u)
    integer, intent(in) :: u
    type(ssids_akeep) :: akeep
    type(ssids_options) :: options
    type(ssids_inform) :: sinform
    integer(long), allocatable :: ptr(:)
    integer, allocatable :: rowi(:), cnt(:), order(:), tmp(:)
    integer :: l, r, c, j, m, a, b, t, nn
    integer(long) :: p, q, w

    allocate(cnt(n), ptr(n+1), order(n))
    cnt = 1                               ! explicit diagonal
    do l = 1, ne
      r = max(matrix%row(l), matrix%col(l)) ; c = min(matrix%row(l), matrix%col(l))
      if (c < 1 .or. r > n) cycle
      if (r /= c) cnt(c) = cnt(c) + 1
    end do
    ptr(1) = 1
    do j = 1, n
      ptr(j+1) = ptr(j) + cnt(j)
    end do
    allocate(rowi(ptr(n+1)-1))
    do j = 1, n
      rowi(ptr(j)) = j
      cnt(j) = 1
    end do
    do l = 1, ne
      r = max(matrix%row(l), matrix%col(l)) ; c = min(matrix%row(l), matrix%col(l))
      if (c < 1 .or. r > n) cycle
      if (r /= c) then
        rowi(ptr(c) + cnt(c)) = r
        cnt(c) = cnt(c) + 1
      end if
    end do
    ! sort each column, squeeze duplicates
    w = 1
    do j = 1, n
      p = ptr(j) ; m = cnt(j)
      allocate(tmp(m))
      tmp = rowi(p:p+m-1)
      do a = 2, m
        t = tmp(a) ; b = a - 1
        do while (b >= 1)
          if (tmp(b) <= t) exit
          tmp(b+1) = tmp(b) ; b = b - 1
        end do
        tmp(b+1) = t
      end do
      ptr(j) = w
      do a = 1, m
        if (a > 1) then
          if (tmp(a) == tmp(a-1)) cycle
        end if
        rowi(w) = tmp(a) ; w = w + 1
      end do
      deallocate(tmp)
    end do
    ptr(n+1) = w

    if (have_perm /= 0) then
      order = perm
    else
      do j = 1, n
        order(j) = j
      end do
    end if
    options%ordering = 0
    options%print_level = -1
    if (nemin > 0) options%nemin = nemin
    call ssids_analyse(.false., n, ptr, rowi(1:ptr(n+1)-1), akeep, options, sinform, order=order)
    nn = akeep%nnodes
    write(u) sinform%flag, nn
    if (sinform%flag >= 0) then
      write(u) akeep%sptr(1:nn+1)
      write(u) akeep%sparent(1:nn)
      write(u) akeep%rptr(1:nn+1)
      write(u) akeep%rlist(1:akeep%rptr(nn+1)-1)
      write(u) order(1:n)
      write(u) sinform%num_factor, sinform%num_flops
      write(u) akeep%nptr(1:nn+1)
      q = akeep%nptr(nn+1) - 1
      write(u) akeep%nlist(1:2, 1:q)
    end if
  end subroutine dump_structure
end program gsls_ref_driver
