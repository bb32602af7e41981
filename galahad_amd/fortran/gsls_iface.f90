! GALAHAD_GSLS_double -- ISO_C_BINDING interface from GALAHAD's Fortran host code to the MI355X
! sparse symmetric backend (C ABI: include/gsls.h, library: galahad_amd/libgsls.so).
!
! The public routines deliberately have the shape of the SPRAL-SSIDS calls that SLS makes today, so
! that a `CASE ( 'gsls' )` arm in src/sls/sls.f90 is a copy of the `CASE ( 'ssids' )` arm with the
! names changed (INTEGRATION.md shows the arms):
!
!   GSLS_analyse( n, ptr, row, keep, options, inform, order )      <-> ssids_analyse  ssids.f90:148
!   GSLS_factor( posdef, val, keep, options, inform [, scale] )    <-> ssids_factor   ssids.f90:770
!   GSLS_solve( x, keep, options, inform [, job] )                 <-> ssids_solve    ssids.f90:1114
!   GSLS_solve_mult( nrhs, x, ldx, keep, options, inform [, job] ) <-> ssids_solve    ssids.f90:1139
!   GSLS_enquire_posdef / GSLS_enquire_indef / GSLS_alter          <-> ssids.f90:1255-1384
!   GSLS_free( keep )                                              <-> ssids_free     ssids.f90:1388
!
! gsls_keep plays the role of the (akeep, fkeep) pair: it owns the C handle (symbolic data, factors,
! device memory).  gsls_options / gsls_inform are bind(C) mirrors of struct gsls_options /
! struct gsls_inform and carry the fields of ssids_options / ssids_inform that SLS reads or sets
! (src/sls/sls.f90:1385-1439, 1737-1786).
module GALAHAD_GSLS_double
  use, intrinsic :: iso_c_binding
  implicit none
  private
  public :: gsls_keep, gsls_options, gsls_inform
  public :: GSLS_initialize, GSLS_analyse, GSLS_factor, GSLS_solve, GSLS_solve_mult
  public :: GSLS_enquire_posdef, GSLS_enquire_indef, GSLS_alter, GSLS_free
  public :: GSLS_set_coo, GSLS_factor_coo, GSLS_residual, GSLS_get_order, GSLS_solve_ir, GSLS_set_value_part
  ! one system over several GPUs, one process per GPU (include/gsls.h, "multi-GPU with the exchange INSIDE the library";
  ! what one ssids_factor / ssids_solve call does for several devices, src/ssids/fkeep.F90:99-174, 229-318)
  public :: GSLS_comm_unique_id, GSLS_comm_init, GSLS_comm_factor, GSLS_comm_solve, GSLS_comm_free

  integer, parameter :: wp = c_double
  integer, parameter :: long = c_int64_t

  type, bind(C) :: gsls_options
    integer(c_int32_t) :: print_level = -1
    integer(c_int32_t) :: ordering = 1      ! 0 user order, 1 built-in nested dissection, 3 natural
    integer(c_int32_t) :: nemin = 32
    integer(c_int32_t) :: scaling = 0
    integer(c_int32_t) :: action = 1
    integer(c_int32_t) :: device = -1
    integer(c_int32_t) :: reserved2 = 0
    integer(c_int32_t) :: reserved0 = 0
    real(c_double) :: u = 0.01_wp
    real(c_double) :: small = 1.0e-20_wp
    real(c_double) :: multiplier = 1.1_wp
    real(c_double) :: reserved1 = 0.0_wp
  end type gsls_options

  type, bind(C) :: gsls_inform
    integer(c_int32_t) :: flag = 0
    integer(c_int32_t) :: matrix_dup = 0
    integer(c_int32_t) :: matrix_missing_diag = 0
    integer(c_int32_t) :: matrix_outrange = 0
    integer(c_int32_t) :: matrix_rank = 0
    integer(c_int32_t) :: maxdepth = 0
    integer(c_int32_t) :: maxfront = 0
    integer(c_int32_t) :: num_delay = 0
    integer(c_int64_t) :: num_factor = 0
    integer(c_int64_t) :: num_flops = 0
    integer(c_int32_t) :: num_neg = 0
    integer(c_int32_t) :: num_sup = 0
    integer(c_int32_t) :: num_two = 0
    integer(c_int32_t) :: stat = 0
    integer(c_int32_t) :: hip_error = 0
    integer(c_int32_t) :: not_first_pass = 0
    integer(c_int32_t) :: nlevels = 0
    integer(c_int32_t) :: reserved0 = 0
    integer(c_int64_t) :: factor_bytes = 0
    integer(c_int64_t) :: solve_bytes = 0
    real(c_double) :: time_analyse = 0.0_wp
    real(c_double) :: time_factor = 0.0_wp
    real(c_double) :: time_solve = 0.0_wp
    real(c_double) :: reserved1 = 0.0_wp
  end type gsls_inform

  type :: gsls_keep
    type(c_ptr) :: handle = c_null_ptr
    integer :: n = 0
  end type gsls_keep

  interface
    subroutine c_gsls_default_options(options) bind(C, name='gsls_default_options')
      import :: gsls_options
      type(gsls_options), intent(out) :: options
    end subroutine
    integer(c_int) function c_gsls_create(handle) bind(C, name='gsls_create')
      import :: c_ptr, c_int
      type(c_ptr), intent(out) :: handle
    end function
    integer(c_int) function c_gsls_destroy(handle) bind(C, name='gsls_destroy')
      import :: c_ptr, c_int
      type(c_ptr), intent(inout) :: handle
    end function
    integer(c_int) function c_gsls_analyse(handle, n, ptr, row, order, options, inform) &
        bind(C, name='gsls_analyse')
      import :: c_ptr, c_int, c_int32_t, c_int64_t, gsls_options, gsls_inform
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: n
      integer(c_int64_t), intent(in) :: ptr(*)
      integer(c_int32_t), intent(in) :: row(*)
      integer(c_int32_t), intent(inout) :: order(*)
      type(gsls_options), intent(in) :: options
      type(gsls_inform), intent(out) :: inform
    end function
    integer(c_int) function c_gsls_analyse_matching(handle, n, ptr, row, val, order, options, inform) &
        bind(C, name='gsls_analyse_matching')
      import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double, gsls_options, gsls_inform
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: n
      integer(c_int64_t), intent(in) :: ptr(*)
      integer(c_int32_t), intent(in) :: row(*)
      real(c_double), intent(in) :: val(*)
      integer(c_int32_t), intent(inout) :: order(*)
      type(gsls_options), intent(in) :: options
      type(gsls_inform), intent(out) :: inform
    end function
    integer(c_int) function c_gsls_factor(handle, posdef, val, scale, options, inform) &
        bind(C, name='gsls_factor')
      import :: c_ptr, c_int, c_int32_t, c_double, gsls_options, gsls_inform
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: posdef
      real(c_double), intent(in) :: val(*)
      type(c_ptr), value :: scale
      type(gsls_options), intent(in) :: options
      type(gsls_inform), intent(out) :: inform
    end function
    integer(c_int) function c_gsls_solve(handle, job, nrhs, x, ldx, options, inform) &
        bind(C, name='gsls_solve')
      import :: c_ptr, c_int, c_int32_t, c_double, gsls_options, gsls_inform
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: job, nrhs, ldx
      real(c_double), intent(inout) :: x(*)
      type(gsls_options), intent(in) :: options
      type(gsls_inform), intent(out) :: inform
    end function
    integer(c_int) function c_gsls_enquire_posdef(handle, d, inform) bind(C, name='gsls_enquire_posdef')
      import :: c_ptr, c_int, c_double, gsls_inform
      type(c_ptr), value :: handle
      real(c_double), intent(out) :: d(*)
      type(gsls_inform), intent(out) :: inform
    end function
    integer(c_int) function c_gsls_enquire_indef(handle, piv_order, d, inform) &
        bind(C, name='gsls_enquire_indef')
      import :: c_ptr, c_int, gsls_inform
      type(c_ptr), value :: handle
      type(c_ptr), value :: piv_order, d
      type(gsls_inform), intent(out) :: inform
    end function
    integer(c_int) function c_gsls_set_coo(handle, ne, row, col, map) bind(C, name='gsls_set_coo')
      import :: c_ptr, c_int, c_int32_t, c_int64_t
      type(c_ptr), value :: handle
      integer(c_int64_t), value :: ne
      type(c_ptr), value :: row, col
      integer(c_int32_t), intent(in) :: map(*)
    end function
    integer(c_int) function c_gsls_factor_coo(handle, posdef, val, scale, options, inform) &
        bind(C, name='gsls_factor_coo')
      import :: c_ptr, c_int, c_int32_t, c_double, gsls_options, gsls_inform
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: posdef
      real(c_double), intent(in) :: val(*)
      type(c_ptr), value :: scale
      type(gsls_options), intent(in) :: options
      type(gsls_inform), intent(out) :: inform
    end function
    integer(c_int) function c_gsls_set_value_part(handle, part, val, len, mult) bind(C, name='gsls_set_value_part')
      import :: c_ptr, c_int, c_int32_t, c_int64_t, c_double
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: part
      type(c_ptr), value :: val
      integer(c_int64_t), value :: len
      real(c_double), value :: mult
    end function
    integer(c_int) function c_gsls_residual(handle, nrhs, x, ldx, b, ldb, r, ldr, inform) &
        bind(C, name='gsls_residual')
      import :: c_ptr, c_int, c_int32_t, c_double, gsls_inform
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: nrhs, ldx, ldb, ldr
      real(c_double), intent(in) :: x(*), b(*)
      real(c_double), intent(out) :: r(*)
      type(gsls_inform), intent(out) :: inform
    end function
    integer(c_int) function c_gsls_solve_ir(handle, x, max_ref, res_abs, res_rel, iters, options, inform) &
        bind(C, name='gsls_solve_ir')
      import :: c_ptr, c_int, c_int32_t, c_double, gsls_options, gsls_inform
      type(c_ptr), value :: handle
      real(c_double), intent(inout) :: x(*)
      integer(c_int32_t), value :: max_ref
      real(c_double), value :: res_abs, res_rel
      integer(c_int32_t), intent(out) :: iters
      type(gsls_options), intent(in) :: options
      type(gsls_inform), intent(out) :: inform
    end function
    integer(c_int) function c_gsls_get_order(handle, order) bind(C, name='gsls_get_order')
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: handle
      integer(c_int32_t), intent(out) :: order(*)
    end function
    integer(c_int) function c_gsls_alter(handle, d, inform) bind(C, name='gsls_alter')
      import :: c_ptr, c_int, c_double, gsls_inform
      type(c_ptr), value :: handle
      real(c_double), intent(in) :: d(*)
      type(gsls_inform), intent(out) :: inform
    end function
    integer(c_int) function c_gsls_comm_unique_id(id) bind(C, name='gsls_comm_unique_id')
      import :: c_int, c_char
      character(kind=c_char), intent(out) :: id(128)
    end function
    integer(c_int) function c_gsls_comm_init(handle, nranks, rank, id, options) bind(C, name='gsls_comm_init')
      import :: c_ptr, c_int, c_int32_t, c_char, gsls_options
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: nranks, rank
      character(kind=c_char), intent(in) :: id(128)
      type(gsls_options), intent(in) :: options
    end function
    integer(c_int) function c_gsls_comm_factor(handle, posdef, val, options, inform) bind(C, name='gsls_comm_factor')
      import :: c_ptr, c_int, c_int32_t, c_double, gsls_options, gsls_inform
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: posdef
      real(c_double), intent(in) :: val(*)
      type(gsls_options), intent(in) :: options
      type(gsls_inform), intent(out) :: inform
    end function
    integer(c_int) function c_gsls_comm_solve(handle, x, inform) bind(C, name='gsls_comm_solve')
      import :: c_ptr, c_int, c_double, gsls_inform
      type(c_ptr), value :: handle
      real(c_double), intent(inout) :: x(*)
      type(gsls_inform), intent(out) :: inform
    end function
    integer(c_int) function c_gsls_comm_destroy(handle) bind(C, name='gsls_comm_destroy')
      import :: c_ptr, c_int
      type(c_ptr), value :: handle
    end function
  end interface

contains

  ! ---- one system over several GPUs: every rank (process) analyses the same matrix, rank 0 obtains the 128-byte id
  ! and hands it to the others by whatever means the host program has (MPI_Bcast, a file), then all ranks call
  ! GSLS_comm_init and, any number of times, GSLS_comm_factor / GSLS_comm_solve.  status: the C ABI's flag.
  subroutine GSLS_comm_unique_id(id, status)
    character(kind=c_char), intent(out) :: id(128)
    integer, intent(out) :: status
    status = int(c_gsls_comm_unique_id(id))
  end subroutine GSLS_comm_unique_id

  subroutine GSLS_comm_init(nranks, rank, id, keep, options, status)
    integer, intent(in) :: nranks, rank          ! rank = 0 .. nranks - 1
    character(kind=c_char), intent(in) :: id(128)
    type(gsls_keep), intent(inout) :: keep
    type(gsls_options), intent(in) :: options
    integer, intent(out) :: status
    status = int(c_gsls_comm_init(keep%handle, int(nranks, c_int32_t), int(rank, c_int32_t), id, options))
  end subroutine GSLS_comm_init

  ! val: the sorted lower-by-columns values, as for GSLS_factor; inform holds the totals on every rank
  subroutine GSLS_comm_factor(posdef, val, keep, options, inform)
    logical, intent(in) :: posdef
    real(wp), intent(in) :: val(:)
    type(gsls_keep), intent(inout) :: keep
    type(gsls_options), intent(in) :: options
    type(gsls_inform), intent(out) :: inform
    integer(c_int) :: rc
    rc = c_gsls_comm_factor(keep%handle, merge(1_c_int32_t, 0_c_int32_t, posdef), val, options, inform)
  end subroutine GSLS_comm_factor

  ! x = b on entry (the same on every rank), the whole solution on every rank on exit
  subroutine GSLS_comm_solve(x, keep, inform)
    real(wp), intent(inout) :: x(:)
    type(gsls_keep), intent(inout) :: keep
    type(gsls_inform), intent(out) :: inform
    integer(c_int) :: rc
    rc = c_gsls_comm_solve(keep%handle, x, inform)
  end subroutine GSLS_comm_solve

  subroutine GSLS_comm_free(keep, status)
    type(gsls_keep), intent(inout) :: keep
    integer, intent(out) :: status
    status = 0
    if (c_associated(keep%handle)) status = int(c_gsls_comm_destroy(keep%handle))
  end subroutine GSLS_comm_free

  subroutine GSLS_initialize(keep, options)
    type(gsls_keep), intent(inout) :: keep
    type(gsls_options), intent(out) :: options
    integer(c_int) :: rc
    call c_gsls_default_options(options)
    if (.not. c_associated(keep%handle)) rc = c_gsls_create(keep%handle)
  end subroutine GSLS_initialize

  ! order(i) = position of variable i in the pivot sequence; used on entry when options%ordering = 0,
  ! always set on exit (ssids.f90:381)
  ! val present: matching-based ordering and scaling, as ssids_analyse does with val and options%ordering = 2
  ! (ssids.f90:305-320); factorize with options%scaling = 3 to use the saved scaling
  subroutine GSLS_analyse(n, ptr, row, keep, options, inform, order, val)
    integer, intent(in) :: n
    integer(long), intent(in) :: ptr(:)
    integer, intent(in) :: row(:)
    type(gsls_keep), intent(inout) :: keep
    type(gsls_options), intent(in) :: options
    type(gsls_inform), intent(out) :: inform
    integer, intent(inout) :: order(:)
    real(wp), optional, intent(in) :: val(:)
    integer(c_int) :: rc
    if (.not. c_associated(keep%handle)) rc = c_gsls_create(keep%handle)
    keep%n = n
    if (present(val)) then
      rc = c_gsls_analyse_matching(keep%handle, int(n, c_int32_t), ptr, row, val, order, options, inform)
    else
      rc = c_gsls_analyse(keep%handle, int(n, c_int32_t), ptr, row, order, options, inform)
    end if
  end subroutine GSLS_analyse

  subroutine GSLS_factor(posdef, val, keep, options, inform, scale)
    logical, intent(in) :: posdef
    real(wp), intent(in) :: val(:)
    type(gsls_keep), intent(inout) :: keep
    type(gsls_options), intent(in) :: options
    type(gsls_inform), intent(out) :: inform
    real(wp), optional, target, intent(in) :: scale(:)
    integer(c_int) :: rc
    type(c_ptr) :: sp
    sp = c_null_ptr
    if (present(scale)) sp = c_loc(scale)
    rc = c_gsls_factor(keep%handle, merge(1_c_int32_t, 0_c_int32_t, posdef), val, sp, options, inform)
  end subroutine GSLS_factor

  subroutine GSLS_solve(x, keep, options, inform, job)
    real(wp), intent(inout) :: x(:)
    type(gsls_keep), intent(inout) :: keep
    type(gsls_options), intent(in) :: options
    type(gsls_inform), intent(out) :: inform
    integer, optional, intent(in) :: job
    integer(c_int) :: rc
    integer(c_int32_t) :: j
    j = 0
    if (present(job)) j = int(job, c_int32_t)
    rc = c_gsls_solve(keep%handle, j, 1_c_int32_t, x, int(max(keep%n, 1), c_int32_t), options, inform)
  end subroutine GSLS_solve

  subroutine GSLS_solve_mult(nrhs, x, ldx, keep, options, inform, job)
    integer, intent(in) :: nrhs, ldx
    real(wp), intent(inout) :: x(ldx, nrhs)
    type(gsls_keep), intent(inout) :: keep
    type(gsls_options), intent(in) :: options
    type(gsls_inform), intent(out) :: inform
    integer, optional, intent(in) :: job
    integer(c_int) :: rc
    integer(c_int32_t) :: j
    j = 0
    if (present(job)) j = int(job, c_int32_t)
    rc = c_gsls_solve(keep%handle, j, int(nrhs, c_int32_t), x, int(ldx, c_int32_t), options, inform)
  end subroutine GSLS_solve_mult

  subroutine GSLS_enquire_posdef(keep, inform, d)
    type(gsls_keep), intent(inout) :: keep
    type(gsls_inform), intent(out) :: inform
    real(wp), intent(out) :: d(:)
    integer(c_int) :: rc
    rc = c_gsls_enquire_posdef(keep%handle, d, inform)
  end subroutine GSLS_enquire_posdef

  subroutine GSLS_enquire_indef(keep, inform, piv_order, d)
    type(gsls_keep), intent(inout) :: keep
    type(gsls_inform), intent(out) :: inform
    integer, optional, target, intent(out) :: piv_order(:)
    real(wp), optional, target, intent(out) :: d(:, :)
    integer(c_int) :: rc
    type(c_ptr) :: pp, dp
    pp = c_null_ptr ; dp = c_null_ptr
    if (present(piv_order)) pp = c_loc(piv_order)
    if (present(d)) dp = c_loc(d)
    rc = c_gsls_enquire_indef(keep%handle, pp, dp, inform)
  end subroutine GSLS_enquire_indef

  subroutine GSLS_alter(d, keep, inform)
    real(wp), intent(in) :: d(:, :)
    type(gsls_keep), intent(inout) :: keep
    type(gsls_inform), intent(out) :: inform
    integer(c_int) :: rc
    rc = c_gsls_alter(keep%handle, d, inform)
  end subroutine GSLS_alter

  ! the caller's COORDINATE storage (row, col optional) and SLS's map from its entries to the sorted
  ! lower-by-columns values (src/sls/sls.f90:8409-8578) are handed to the backend once, after GSLS_analyse
  subroutine GSLS_set_coo(ne, map, keep, inform, row, col)
    integer, intent(in) :: ne
    integer, intent(in) :: map(:)
    type(gsls_keep), intent(inout) :: keep
    type(gsls_inform), intent(inout) :: inform
    integer, optional, target, intent(in) :: row(:), col(:)
    integer(c_int) :: rc
    type(c_ptr) :: rp, cp
    rp = c_null_ptr ; cp = c_null_ptr
    if (present(row) .and. present(col)) then
      rp = c_loc(row) ; cp = c_loc(col)
    end if
    rc = c_gsls_set_coo(keep%handle, int(ne, c_int64_t), rp, cp, map)
    inform%flag = int(rc)
  end subroutine GSLS_set_coo

  ! SLS_factorize without the host scatter: val in the caller's entry order (sls.f90:4113-4150 + 4273-4297)
  subroutine GSLS_factor_coo(posdef, val, keep, options, inform, scale)
    logical, intent(in) :: posdef
    real(wp), intent(in) :: val(:)
    type(gsls_keep), intent(inout) :: keep
    type(gsls_options), intent(in) :: options
    type(gsls_inform), intent(out) :: inform
    real(wp), optional, target, intent(in) :: scale(:)
    integer(c_int) :: rc
    type(c_ptr) :: sp
    sp = c_null_ptr
    if (present(scale)) sp = c_loc(scale)
    rc = c_gsls_factor_coo(keep%handle, merge(1_c_int32_t, 0_c_int32_t, posdef), val, sp, options, inform)
  end subroutine GSLS_factor_coo

  ! The next GSLS_factor_coo reads its values from the registered arrays laid end to end (part = 0, 1, 2 ...), each times
  ! mult, instead of from its val argument: SBLS's K%val = [ A%val | H%val | -C%val ] without the host copy
  ! (sbls.f90:3349, 3404, 3967).  The arrays are read when GSLS_factor_coo runs, so they must be the caller's own storage
  ! (contiguous: no temporary), alive until then.  Without val: forget every registration.
  subroutine GSLS_set_value_part(keep, part, val, mult)
    type(gsls_keep), intent(inout) :: keep
    integer, intent(in) :: part
    real(wp), optional, contiguous, target, intent(in) :: val(:)
    real(wp), optional, intent(in) :: mult
    integer(c_int) :: rc
    real(c_double) :: f
    if (.not. present(val)) then
      rc = c_gsls_set_value_part(keep%handle, -1_c_int32_t, c_null_ptr, 0_c_int64_t, 1.0_c_double)
      return
    end if
    f = 1.0_c_double
    if (present(mult)) f = mult
    if (size(val) > 0) then
      rc = c_gsls_set_value_part(keep%handle, int(part, c_int32_t), c_loc(val), int(size(val), c_int64_t), f)
    else
      rc = c_gsls_set_value_part(keep%handle, int(part, c_int32_t), c_null_ptr, 0_c_int64_t, f)
    end if
  end subroutine GSLS_set_value_part

  ! r = b - A x with the matrix of the last GSLS_factor_coo (the residual step of SLS_solve_ir, sls.f90:4826-4934)
  subroutine GSLS_residual(x, b, r, keep, inform)
    real(wp), intent(in) :: x(:), b(:)
    real(wp), intent(out) :: r(:)
    type(gsls_keep), intent(inout) :: keep
    type(gsls_inform), intent(out) :: inform
    integer(c_int) :: rc
    integer(c_int32_t) :: n
    n = int(max(keep%n, 1), c_int32_t)
    rc = c_gsls_residual(keep%handle, 1_c_int32_t, x, n, b, n, r, n, inform)
  end subroutine GSLS_residual

  ! SLS_solve_ir (sls.f90:4770-4949) as one call: x = b on entry, the refined solution on exit
  subroutine GSLS_solve_ir(x, max_refinements, residual_absolute, residual_relative, iterations, keep, options, inform)
    real(wp), intent(inout) :: x(:)
    integer, intent(in) :: max_refinements
    real(wp), intent(in) :: residual_absolute, residual_relative
    integer, intent(out) :: iterations
    type(gsls_keep), intent(inout) :: keep
    type(gsls_options), intent(in) :: options
    type(gsls_inform), intent(out) :: inform
    integer(c_int) :: rc
    integer(c_int32_t) :: it
    rc = c_gsls_solve_ir(keep%handle, x, int(max_refinements, c_int32_t), residual_absolute, residual_relative, &
                         it, options, inform)
    iterations = int(it)
  end subroutine GSLS_solve_ir

  ! order(i) = position of variable i in the pivot sequence the factors are in
  subroutine GSLS_get_order(order, keep)
    integer, intent(out) :: order(:)
    type(gsls_keep), intent(inout) :: keep
    integer(c_int) :: rc
    rc = c_gsls_get_order(keep%handle, order)
  end subroutine GSLS_get_order

  subroutine GSLS_free(keep, status)
    type(gsls_keep), intent(inout) :: keep
    integer, intent(out) :: status
    status = 0
    if (c_associated(keep%handle)) status = int(c_gsls_destroy(keep%handle))
    keep%handle = c_null_ptr
  end subroutine GSLS_free

end module GALAHAD_GSLS_double
