! TEST INFRASTRUCTURE ONLY (oracle/): drives the reference's TRS_solve (src/trs/trs.f90) --
!     minimize  1/2 x^T H x + c^T x + f   subject to  ||x||_M <= radius,   M = I
! which analyses H + lambda M once and then factorizes it repeatedly with pivot_control = 2 inside a
! secular-equation loop (trs.f90:1942-1964, 2260-2275), refines with IR_solve and calls
! SLS_part_solve for the derivative terms (trs.f90:2618-2742).  definite_linear_solver is chosen by
! name: 'gsls' (MI355X backend, drop-in build) -- the reference's own sparse backends need an ordering
! package that is a stub in the tree, so only the dense 'sytr'/'potr' arms can run beside it.
!
!   usage: trs_driver <problem.bin> <result.bin>
! problem.bin: int32 magic(1414681344 'TRS\0') version(1) ; int32 n h_ne solver(1 sytr, 4 gsls) print
!              real64 radius f mdiag(0 = no M, else M = mdiag*I) ; int32 Hrow Hcol ; real64 Hval ; real64 c(n)
! result.bin : int32 status factorizations spare spare ; real64 obj multiplier x_norm time ; real64 x(n)
program gsls_trs_driver
  use GALAHAD_TRS_double
  implicit none
  integer, parameter :: wp = kind(1.0d0), long = selected_int_kind(18)
  type(SMT_type) :: H, M
  type(TRS_data_type) :: data
  type(TRS_control_type) :: control
  type(TRS_inform_type) :: inform
  character(len=1024) :: fin, fout
  integer :: magic, version, n, h_ne, isolver, iprint, s, u
  real(wp) :: radius, f, t, mdiag
  real(wp), allocatable :: c(:), x(:)
  integer(long) :: c0, c1, crate

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read(u) magic, version
  if (magic /= 1414681344 .or. version /= 1) stop 'trs_driver: bad problem file'
  read(u) n, h_ne, isolver, iprint
  read(u) radius, f, mdiag
  call SMT_put(H%type, 'COORDINATE', s)
  H%n = n ; H%ne = h_ne
  allocate(H%row(h_ne), H%col(h_ne), H%val(h_ne), c(n), x(n))
  read(u) H%row ; read(u) H%col ; read(u) H%val
  read(u) c
  close(u)

  call TRS_initialize(data, control, inform)
  control%print_level = iprint
  if (isolver == 4) then
    control%symmetric_linear_solver = 'gsls'
    control%definite_linear_solver = 'gsls'
  else
    control%symmetric_linear_solver = 'sytr'
    control%definite_linear_solver = 'sytr'
  end if
  call system_clock(c0, crate)
  if (mdiag /= 0.0_wp) then
    call SMT_put(M%type, 'DIAGONAL', s)
    allocate(M%val(n)) ; M%val = mdiag ; M%n = n
    call TRS_solve(n, radius, f, c, H, x, data, control, inform, M = M)
  else
    call TRS_solve(n, radius, f, c, H, x, data, control, inform)
  end if
  call system_clock(c1)
  t = real(c1 - c0, wp) / real(crate, wp)

  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(u) inform%status, inform%factorizations, 0, 0
  write(u) inform%obj, inform%multiplier, inform%x_norm, t
  write(u) x
  close(u)
  call TRS_terminate(data, control, inform)
end program gsls_trs_driver
