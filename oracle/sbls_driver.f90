! TEST INFRASTRUCTURE ONLY (oracle/): drives the reference's SBLS (src/sbls/sbls.f90) -- the KKT /
! saddle-point layer CQP calls every interior-point iteration (src/cqp/cqp.f90:4781-4896) -- on
!     K = [ H  A^T ; A  -C ],  preconditioner = 2 (exact K), explicit factorization,
! with the symmetric solver chosen by name: 'ssids' (reference CPU path) in oracle/_ref/sbls_driver,
! or 'gsls' (MI355X backend behind the patched SLS facade) in oracle/_ref/sbls_gsls_driver.
!
!   usage: sbls_driver <problem.bin> <result.bin>
! problem.bin: int32 magic(1396853330 'SBLS') version(1)
!              int32 n m h_ne a_ne c_ne solver(0 ssids, 1 sytr, 4 gsls) factorization repeat itref_max spare
!              int32 Hrow Hcol ; real64 Hval ; int32 Arow Acol ; real64 Aval ; int32 Crow Ccol ;
!              real64 Cval ; real64 rhs(n+m)
!              spare = SBLS print level + 100 * drift + 1000 * get_norm_residual.  drift = 1: round k > 1 factorizes
!              H * (1 + (k-1)/4), C * (1 + (k-1)/2) and, on odd k only (new_a = 0 on even k), A * (1 - (k-1)/10): the
!              values change in the CALLER's arrays between factorizations, as in an interior-point loop
! result.bin : int32 status_factorize status_solve factorization_used rank negative_eigenvalues spare
!              real64 t_factorize_best t_solve_best t_factorize_median t_solve_median
!              real64 sol(n+m) ; real64 norm_residual (inform%norm_residual of the last solve)
program gsls_sbls_driver
  use GALAHAD_SBLS_double
  implicit none
  integer, parameter :: wp = kind(1.0d0), long = selected_int_kind(18)
  type(SMT_type) :: H, A, C
  type(SBLS_data_type) :: data
  type(SBLS_control_type) :: control
  type(SBLS_inform_type) :: inform
  character(len=1024) :: fin, fout
  integer :: magic, version, n, m, h_ne, a_ne, c_ne, isolver, factorization, repeat, itref, spare
  real(wp), allocatable :: rhs(:), sol(:), tf(:), ts(:), Hv0(:), Av0(:), Cv0(:)
  integer :: s, u, k, st_f, st_s, drift, getnorm
  integer(long) :: c0, c1, crate

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read(u) magic, version
  if (magic /= 1396853330 .or. version /= 1) stop 'sbls_driver: bad problem file'
  read(u) n, m, h_ne, a_ne, c_ne, isolver, factorization, repeat, itref, spare
  call SMT_put(H%type, 'COORDINATE', s)
  call SMT_put(A%type, 'COORDINATE', s)
  call SMT_put(C%type, 'COORDINATE', s)
  allocate(H%row(h_ne), H%col(h_ne), H%val(h_ne), A%row(a_ne), A%col(a_ne), A%val(a_ne))
  allocate(C%row(c_ne), C%col(c_ne), C%val(c_ne), rhs(n+m), sol(n+m))
  H%n = n ; H%m = n ; H%ne = h_ne ; A%m = m ; A%n = n ; A%ne = a_ne ; C%n = m ; C%m = m ; C%ne = c_ne
  read(u) H%row ; read(u) H%col ; read(u) H%val
  read(u) A%row ; read(u) A%col ; read(u) A%val
  read(u) C%row ; read(u) C%col ; read(u) C%val
  read(u) rhs
  close(u)

  call SBLS_initialize(data, control, inform)
  control%preconditioner = 2
  control%factorization = factorization
  control%itref_max = itref
  getnorm = spare / 1000 ; drift = mod(spare, 1000) / 100 ; spare = mod(spare, 100)
  control%get_norm_residual = getnorm /= 0
  control%print_level = spare      ! SBLS print level (debugging aid)
  allocate(Hv0(h_ne), Av0(a_ne), Cv0(c_ne))
  Hv0 = H%val ; Av0 = A%val ; Cv0 = C%val
  if (isolver == 4) then
    control%symmetric_linear_solver = 'gsls'
    control%definite_linear_solver = 'gsls'
  else if (isolver == 1) then          ! dense LAPACK arm: the only reference backend that needs no
    control%symmetric_linear_solver = 'sytr'   ! ordering package (METIS/MC68 are stubs, SBLS passes no PERM)
    control%definite_linear_solver = 'sytr'
  else
    control%symmetric_linear_solver = 'ssids'
    control%definite_linear_solver = 'ssids'
  end if
  repeat = max(repeat, 1)
  allocate(tf(repeat), ts(repeat))
  tf = 0.0_wp ; ts = 0.0_wp ; st_s = -999
  call system_clock(c0, crate)
  do k = 1, repeat
    if (k > 1) then      ! same structure, new values: what an interior-point iteration does
      control%new_a = 1 ; control%new_h = 1 ; control%new_c = 1
      if (drift /= 0) then
        H%val = Hv0 * (1.0_wp + 0.25_wp * real(k - 1, wp))
        C%val = Cv0 * (1.0_wp + 0.5_wp * real(k - 1, wp))
        if (mod(k, 2) == 1) then
          A%val = Av0 * (1.0_wp - 0.1_wp * real(k - 1, wp))
        else
          control%new_a = 0
        end if
      end if
    end if
    call system_clock(c0)
    call SBLS_form_and_factorize(n, m, H, A, C, data, control, inform)
    call system_clock(c1)
    tf(k) = real(c1 - c0, wp) / real(crate, wp)
    st_f = inform%status
    if (st_f < 0) exit
    sol = rhs
    call system_clock(c0)
    call SBLS_solve(n, m, A, C, data, control, inform, sol)
    call system_clock(c1)
    ts(k) = real(c1 - c0, wp) / real(crate, wp)
    st_s = inform%status
  end do

  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(u) st_f, st_s, inform%factorization, inform%rank, inform%SLS_inform%negative_eigenvalues, 0
  write(u) minval(tf), minval(ts), median(tf), median(ts)
  write(u) sol
  write(u) inform%norm_residual
  close(u)
  call SBLS_terminate(data, control, inform)

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
end program gsls_sbls_driver
