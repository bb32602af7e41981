! Fortran driver for the binding: the 5x5 indefinite known-answer system of GALAHAD's own SLS tests
! (src/sls/slss.f90:17-24, src/sls/slst.f90:29-40; solution 1..5) pushed through GALAHAD_GSLS_double.
! The lower triangle by columns below is what SLS_analyse hands to a backend for that matrix
! (explicit diagonal, duplicates merged; src/sls/sls.f90:8409-8578).  Exit status 0 = pass.
program gsls_kat
  use GALAHAD_GSLS_double
  use, intrinsic :: iso_c_binding
  implicit none
  integer, parameter :: wp = c_double, n = 5
  type(gsls_keep) :: keep
  type(gsls_options) :: options
  type(gsls_inform) :: inform
  integer(c_int64_t) :: ptr(n + 1) = (/ 1, 3, 6, 8, 9, 10 /)
  integer :: row(9) = (/ 1, 2, 2, 3, 5, 3, 4, 4, 5 /)
  real(wp) :: val(9) = (/ 2.0_wp, 3.0_wp, 0.0_wp, 4.0_wp, 6.0_wp, 1.0_wp, 5.0_wp, 0.0_wp, 1.0_wp /)
  real(wp) :: x(n) = (/ 8.0_wp, 45.0_wp, 31.0_wp, 15.0_wp, 17.0_wp /)
  real(wp) :: d(2, n)
  integer :: order(n), piv(n), i, status

  call GSLS_initialize(keep, options)
  options%ordering = 0
  do i = 1, n
    order(i) = i
  end do
  call GSLS_analyse(n, ptr, row, keep, options, inform, order)
  if (inform%flag < 0) then
    write(*, '(a,i0)') ' gsls_kat: analyse failed, flag = ', inform%flag ; stop 1
  end if
  write(*, '(a,i0,a,i0)') ' gsls_kat: analyse ok, entries_in_factors = ', inform%num_factor, &
       ' flops = ', inform%num_flops
  call GSLS_factor(.false., val, keep, options, inform)
  if (inform%flag == -51) then
    write(*, '(a)') ' gsls_kat: no HIP device (flag -51): numeric phase not run' ; stop 2
  end if
  if (inform%flag < 0) then
    write(*, '(a,i0)') ' gsls_kat: factor failed, flag = ', inform%flag ; stop 1
  end if
  call GSLS_solve(x, keep, options, inform)
  write(*, '(a,5f6.2)') ' Solution is', x
  call GSLS_enquire_indef(keep, inform, piv_order=piv, d=d)
  write(*, '(a,i0,a,i0)') ' negative eigenvalues = ', inform%num_neg, ' 2x2 pivots = ', inform%num_two
  call GSLS_free(keep, status)
  do i = 1, n
    if (abs(x(i) - real(i, wp)) > 1.0e-8_wp) stop 1
  end do
  if (inform%num_neg /= 2) stop 1
  write(*, '(a)') ' gsls_kat: PASS'
end program gsls_kat
