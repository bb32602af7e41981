! TEST INFRASTRUCTURE ONLY (oracle/): BASELINE.json configs[0] -- the reference's interior-point QP solver CQP
! (src/cqp/cqp.f90) on the QPBAND problem (examples/QPBAND.SIF: H = tridiag(2,-1), g_i = -i/N, constraints
! x_i + x_{M+i} = 1 for i = 1..M = N/2 [c_l = 1, c_u = infinity in QPBAND.qplib], 0 <= x <= 2), with the symmetric
! solver of its SBLS calls chosen by name.  Built only above the patched SLS facade (oracle/build_ref.sh), which knows
! both the reference's own solvers and 'gsls'; the dense LAPACK arm 'sytr' is the reference run the gsls run is compared
! with (ssids cannot be reached from CQP here: it needs METIS or MC68, both stubs, and CQP passes no PERM).
! CQP's two side users of linear algebra, the dependency check (FDC) and the crossover (CRO), are switched off: the
! path under test is CQP -> SBLS -> SLS (cqp.f90:4781-4896).
!
!   usage: cqp_driver <problem.bin> <result.bin>
! problem.bin: int32 magic(1129336146 'CQP ') version(1) ; int32 n m h_ne a_ne solver(1 sytr, 4 gsls) print_level
!              int32 Hrow Hcol ; real64 Hval ; int32 Arow Acol ; real64 Aval
!              real64 g(n) c_l(m) c_u(m) x_l(n) x_u(n)
! result.bin : int32 status iter factorizations spare ; real64 obj primal_infeasibility dual_infeasibility
!              complementary_slackness time_total time_factorize time_solve ; real64 x(n) y(m) z(n)
program gsls_cqp_driver
  use GALAHAD_CQP_double
  implicit none
  integer, parameter :: wp = kind(1.0d0)
  real(wp), parameter :: infinity = 1.0e20_wp
  type(QPT_problem_type) :: p
  type(CQP_data_type) :: data
  type(CQP_control_type) :: control
  type(CQP_inform_type) :: inform
  character(len=1024) :: fin, fout
  integer :: magic, version, n, m, h_ne, a_ne, isolver, plevel, s, u
  integer, allocatable :: C_stat(:), B_stat(:)

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read(u) magic, version
  if (magic /= 1129336146 .or. version /= 1) stop 'cqp_driver: bad problem file'
  read(u) n, m, h_ne, a_ne, isolver, plevel
  allocate(p%G(n), p%X_l(n), p%X_u(n), p%C(m), p%C_l(m), p%C_u(m), p%X(n), p%Y(m), p%Z(n))
  allocate(B_stat(n), C_stat(m))
  call SMT_put(p%H%type, 'COORDINATE', s)
  call SMT_put(p%A%type, 'COORDINATE', s)
  allocate(p%H%row(h_ne), p%H%col(h_ne), p%H%val(h_ne), p%A%row(a_ne), p%A%col(a_ne), p%A%val(a_ne))
  read(u) p%H%row ; read(u) p%H%col ; read(u) p%H%val
  read(u) p%A%row ; read(u) p%A%col ; read(u) p%A%val
  read(u) p%G ; read(u) p%C_l ; read(u) p%C_u ; read(u) p%X_l ; read(u) p%X_u
  close(u)
  p%new_problem_structure = .true.
  p%n = n ; p%m = m ; p%f = 0.0_wp ; p%H%ne = h_ne ; p%A%ne = a_ne
  p%X = 0.0_wp ; p%Y = 0.0_wp ; p%Z = 0.0_wp

  call CQP_initialize(data, control, inform)
  control%infinity = infinity
  control%print_level = plevel
  if (plevel > 3) then      ! (debugging aid: the layers below)
    control%SBLS_control%print_level = plevel - 3
    control%SBLS_control%SLS_control%print_level = plevel - 3
  end if
  control%remove_dependencies = .false.
  control%crossover = .false.
  if (isolver == 4) then
    control%SBLS_control%symmetric_linear_solver = 'gsls'
    control%SBLS_control%definite_linear_solver = 'gsls'
  else
    control%SBLS_control%symmetric_linear_solver = 'sytr'
    control%SBLS_control%definite_linear_solver = 'sytr'
  end if
  call CQP_solve(p, data, control, inform, C_stat, B_stat)

  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(u) inform%status, inform%iter, inform%nfacts, 0
  write(u) inform%obj, inform%primal_infeasibility, inform%dual_infeasibility, inform%complementary_slackness, &
           real(inform%time%clock_total, wp), real(inform%time%clock_factorize, wp), real(inform%time%clock_solve, wp)
  write(u) p%X ; write(u) p%Y ; write(u) p%Z
  close(u)
  call CQP_terminate(data, control, inform)
end program gsls_cqp_driver
