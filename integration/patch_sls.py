#!/usr/bin/env python3
"""Produces a copy of GALAHAD's src/sls/sls.f90 with the `gsls` backend arms of INTEGRATION.md
inserted, so that the REAL SLS facade can be built against the MI355X backend and driven end to end
(tests/test_sls_dropin.py).  The reference source is read where it lies and the patched copy is
written to the path given on the command line (a scratch directory -- never the repository).

The repository holds only what a maintainer would add: the arms below.  Each arm is inserted in
front of the k-th `CASE ( 'ssids' )` of the file, the arm it mirrors (src/sls/sls.f90 line numbers of
the v4.0 tree in the comments), and 'gsls' joins the two shared CASE lists that select the sorted
lower-by-columns storage (:2849, :4106).  It does NOT join the list at :2263: that one routes
control%ordering >= 0 to MC68 (a stub here), whereas SBLS reaches SLS with ordering = 0 and expects
the solver's own default ordering.
"""
import re
import sys

ARMS = [
    # 0: SLS_initialize, solver-specific defaults (ssids arm :887-892)
    """     CASE ( 'gsls' )
       CALL GSLS_initialize( data%gsls_keep, data%gsls_options )
       control%scaling = 0
       IF ( control%ordering == 0 ) control%ordering = - 1
       control%node_amalgamation = 0          ! the backend chooses from the tree it finds (24 or 64)

""",
    # 1: SLS_initialize_solver (:1024)
    """     CASE ( 'gsls' )
       data%must_be_definite = .FALSE.

""",
    # 2: SLS_analyse (:3115-3150)
    """       CASE ( 'gsls' )
         CALL SLS_copy_control_to_gsls( control, data%gsls_options )
         CALL CPU_time( time ) ; CALL CLOCK_time( clock )
         IF ( PRESENT( PERM ) .OR. mc6168_ordering ) THEN
           data%gsls_options%ordering = 0        ! use data%ORDER as given
         ELSE IF ( control%ordering == 1 .OR. control%ordering == 2 ) THEN
           data%gsls_options%ordering = 2        ! built-in approximate minimum degree
         ELSE
           data%gsls_options%ordering = 1        ! built-in nested dissection (control%ordering <= 0, 3, ...)
         END IF
         IF ( ALLOCATED( data%gsls_ptr ) ) THEN
           IF ( SIZE( data%gsls_ptr ) < matrix%n + 1 ) DEALLOCATE( data%gsls_ptr )
         END IF
         IF ( .NOT. ALLOCATED( data%gsls_ptr ) )                                &
           ALLOCATE( data%gsls_ptr( matrix%n + 1 ), STAT = inform%alloc_status )
         IF ( inform%alloc_status /= 0 ) THEN
           inform%status = GALAHAD_error_allocate ; GO TO 900 ; END IF
         data%gsls_ptr( : matrix%n + 1 ) = data%matrix%PTR( : matrix%n + 1 )
         CALL GSLS_analyse( data%matrix%n, data%gsls_ptr, data%matrix%COL,      &
                            data%gsls_keep, data%gsls_options,                  &
                            data%gsls_inform, data%ORDER )
         CALL SLS_copy_inform_from_gsls( inform, data%gsls_inform )
         IF ( inform%status /= GALAHAD_ok ) GO TO 800
!  the user's storage and the map MAPS stay with the backend (HBM): the value scatter of SLS_factorize and the
!  residual of SLS_solve_ir run there
         IF ( SMT_get( matrix%type ) == 'COORDINATE' ) THEN
           CALL GSLS_set_coo( data%matrix_ne, data%MAPS, data%gsls_keep,        &
                              data%gsls_inform, matrix%ROW, matrix%COL )
         ELSE
           CALL GSLS_set_coo( data%matrix_ne, data%MAPS, data%gsls_keep,        &
                              data%gsls_inform )
         END IF
         CALL SLS_copy_inform_from_gsls( inform, data%gsls_inform )
         IF ( inform%status /= GALAHAD_ok ) GO TO 800

""",
    # 3: SLS_factorize (:4273-4297)
    """       CASE ( 'gsls' )
         CALL SLS_copy_control_to_gsls( control, data%gsls_options )
         CALL CPU_time( time ) ; CALL CLOCK_time( clock )
!  explicit scalings (control%scaling = 1..3, MC64 / MC77): the facade has scattered AND scaled the values into
!  data%matrix%VAL (:4107-4159) and divides x by SCALE around every solve (:4757-4761) -- factorize exactly those
         IF ( data%explicit_scaling ) THEN
           CALL GSLS_factor( data%must_be_definite,                             &
                             data%matrix%VAL( : data%matrix%PTR( data%matrix%n + 1 ) - 1 ), &
                             data%gsls_keep, data%gsls_options, data%gsls_inform )
         ELSE
           CALL GSLS_factor_coo( data%must_be_definite,                         &
                                 matrix%VAL( : data%matrix_ne ), data%gsls_keep,&
                                 data%gsls_options, data%gsls_inform )
         END IF
         CALL SLS_copy_inform_from_gsls( inform, data%gsls_inform )
!  the order the factors are in (pre-ordering, repaired and learned pivot sequence): what SLS_enquire reports
         IF ( inform%status == GALAHAD_ok )                                     &
           CALL GSLS_get_order( data%ORDER, data%gsls_keep )

""",
    # 4: SLS_solve_one_rhs (:5392-5397)
    """     CASE ( 'gsls' )
       CALL CPU_time( time ) ; CALL CLOCK_time( clock )
       CALL GSLS_solve( X( : data%n ), data%gsls_keep, data%gsls_options,       &
                        data%gsls_inform )
       CALL SLS_copy_inform_from_gsls( inform, data%gsls_inform )

""",
    # 5: SLS_solve_multiple_rhs (:5693-5700)
    """     CASE ( 'gsls' )
       lx = SIZE( X, 1 ) ; nrhs = SIZE( X, 2 )
       CALL CPU_time( time ) ; CALL CLOCK_time( clock )
       CALL GSLS_solve_mult( nrhs, X, lx, data%gsls_keep, data%gsls_options,    &
                             data%gsls_inform )
       CALL SLS_copy_inform_from_gsls( inform, data%gsls_inform )

""",
    # 6: SLS_terminate (:5972-5974)
    """     CASE ( 'gsls' )
       CALL SPACE_dealloc_array( data%X2, inform%status, inform%alloc_status )
       CALL GSLS_free( data%gsls_keep, inform%status )
       inform%status = 0

""",
    # 7: SLS_enquire (:6312-6345)
    """     CASE ( 'gsls' )
       IF ( PRESENT( PERM ) ) PERM = data%ORDER( : data%n )
       IF ( PRESENT( PERTURBATION ) ) inform%status = GALAHAD_error_access_pert
       IF ( data%must_be_definite ) THEN
         IF ( PRESENT( PIVOTS ) ) inform%status = GALAHAD_error_access_pivots
         IF ( PRESENT( D ) ) THEN
           CALL GSLS_enquire_posdef( data%gsls_keep, data%gsls_inform, D( 1, : ) )
           D( 2, : ) = 0.0_wp
         END IF
       ELSE
!  PIVOTS as SLS's callers read it (SILS_enquire, sils.f90:2477-2479; MA57): entry k is the variable eliminated at the
!  k-th pivot, negative when it belongs to a 2x2 pivot -- FDC walks PIVOTS( k ) beside D( :, k ) by pivot position
!  (fdc.f90:926-975).  The backend ABI reports, like SSIDS (ssids.f90:1299-1341), the position of every VARIABLE; the
!  ssids arm hands that on unconverted, which FDC then misreads whenever the order is not the identity
         IF ( PRESENT( PIVOTS ) ) THEN
           CALL SPACE_resize_array( data%n, data%PIVOTS,                        &
                                    inform%status, inform%alloc_status )
           IF ( inform%status /= GALAHAD_ok ) GO TO 900
           IF ( PRESENT( D ) ) THEN
             CALL GSLS_enquire_indef( data%gsls_keep, data%gsls_inform,         &
                                      piv_order = data%PIVOTS, d = D )
           ELSE
             CALL GSLS_enquire_indef( data%gsls_keep, data%gsls_inform,         &
                                      piv_order = data%PIVOTS )
           END IF
           DO k = 1, data%n
             PIVOTS( ABS( data%PIVOTS( k ) ) ) = SIGN( k, data%PIVOTS( k ) )
           END DO
         ELSE IF ( PRESENT( D ) ) THEN
           CALL GSLS_enquire_indef( data%gsls_keep, data%gsls_inform, d = D )
         END IF
       END IF

""",
    # 8: SLS_alter_d (:6488-6494)
    """     CASE ( 'gsls' )
       IF ( data%must_be_definite ) THEN
         inform%status = GALAHAD_ok
       ELSE
         CALL GSLS_alter( D, data%gsls_keep, data%gsls_inform )
         CALL SLS_copy_inform_from_gsls( inform, data%gsls_inform )
       END IF

""",
    # 9: SLS_part_solve (:6886-6920; the ssids arm returns "unavailable", gsls implements L, D, U and S)
    """     CASE ( 'gsls' )
       CALL CPU_time( time ) ; CALL CLOCK_time( clock )
       IF ( part == 'L' ) THEN
         CALL GSLS_solve( X( : data%n ), data%gsls_keep, data%gsls_options,     &
                          data%gsls_inform, job = 1 )
         CALL SLS_copy_inform_from_gsls( inform, data%gsls_inform )
       ELSE IF ( part == 'D' ) THEN
         IF ( data%must_be_definite ) THEN
           inform%status = 0
         ELSE
           CALL GSLS_solve( X( : data%n ), data%gsls_keep, data%gsls_options,   &
                            data%gsls_inform, job = 2 )
           CALL SLS_copy_inform_from_gsls( inform, data%gsls_inform )
         END IF
       ELSE IF ( part == 'U' ) THEN
         CALL GSLS_solve( X( : data%n ), data%gsls_keep, data%gsls_options,     &
                          data%gsls_inform, job = 3 )
         CALL SLS_copy_inform_from_gsls( inform, data%gsls_inform )
       ELSE IF ( part == 'S' ) THEN
!  L sqrt(D): a Cholesky factor is L sqrt(D) already; otherwise as the MA57 arm does
         CALL GSLS_solve( X( : data%n ), data%gsls_keep, data%gsls_options,     &
                          data%gsls_inform, job = 1 )
         CALL SLS_copy_inform_from_gsls( inform, data%gsls_inform )
         IF ( inform%status == GALAHAD_ok .AND. .NOT. data%must_be_definite ) THEN
           CALL SPACE_resize_array( data%n, data%WORK,                         &
                                    inform%status, inform%alloc_status )
           IF ( inform%status /= GALAHAD_ok ) GO TO 900
           data%WORK( : data%n ) = X( : data%n )
           CALL GSLS_solve( X( : data%n ), data%gsls_keep, data%gsls_options,   &
                            data%gsls_inform, job = 2 )
           CALL SLS_copy_inform_from_gsls( inform, data%gsls_inform )
           IF ( inform%status /= GALAHAD_ok ) GO TO 900
           DO i = 1, data%n
             IF ( X( i ) == 0.0_wp .AND. data%WORK( i ) == 0.0_wp ) CYCLE
             IF ( ( X( i ) == 0.0_wp .AND. data%WORK( i ) /= 0.0_wp ) .OR.     &
                  ( X( i ) /= 0.0_wp .AND. data%WORK( i ) == 0.0_wp ) .OR.     &
                  ( X( i ) > 0.0_wp .AND. data%WORK( i ) < 0.0_wp ) .OR.       &
                  ( X( i ) < 0.0_wp .AND. data%WORK( i ) > 0.0_wp ) ) THEN
               inform%status = GALAHAD_error_inertia ; GO TO 900
             END IF
             IF ( X( i ) > 0.0_wp ) THEN
               X( i ) = SQRT( X( i ) ) * SQRT( data%WORK( i ) )
             ELSE
               X( i ) = - SQRT( - X( i ) ) * SQRT( - data%WORK( i ) )
             END IF
           END DO
         END IF
       ELSE
         inform%status = GALAHAD_unavailable_option
       END IF
       GO TO 900

""",
]

# SLS_solve_ir (:4770-4949): the refinement loop -- solves, updates, residuals b - A x, norms -- as ONE backend call
# with every vector resident in HBM; at n = 1.2e6 the host loop's ten passes over the vectors cost five times what
# the solves do
REFINE_ARM = """       IF ( data%solver( 1 : data%len_solver ) == 'gsls' .AND.                   &
            SMT_get( matrix%type ) == 'COORDINATE' .AND.                        &
            .NOT. data%explicit_scaling ) THEN
         CALL GSLS_solve_ir( X( : n ), control%max_iterative_refinements,       &
                             control%acceptable_residual_absolute,              &
                             control%acceptable_residual_relative,              &
                             inform%iterative_refinements, data%gsls_keep,      &
                             data%gsls_options, data%gsls_inform )
         CALL SLS_copy_inform_from_gsls( inform, data%gsls_inform )
         GO TO 900
       END IF
"""

COPY_ROUTINES = """
!-*-   S L S _ C O P Y _ C O N T R O L _ T O _ G S L S  S U B R O U T I N E  -*-

     SUBROUTINE SLS_copy_control_to_gsls( control, control_gsls )

!  copy control parameters to their GSLS equivalents (cf. SLS_copy_control_to_ssids)

     TYPE ( SLS_control_type ), INTENT( IN ) :: control
     TYPE ( gsls_options ), INTENT( INOUT ) :: control_gsls

     control_gsls%print_level = control%print_level_solver - 1
     control_gsls%nemin = control%node_amalgamation
     control_gsls%small = control%absolute_pivot_tolerance
!  the backend's own scalings, as for ssids (SLS_copy_control_to_ssids, :1405-1413)
     IF ( control%scaling == - 1 ) THEN
       control_gsls%scaling = 1
     ELSE IF ( control%scaling == - 2 ) THEN
       control_gsls%scaling = 2
     ELSE IF ( control%scaling == - 3 ) THEN
       control_gsls%scaling = 3
     ELSE
       control_gsls%scaling = 0
     END IF
     IF ( control%pivot_control == 2 .OR. control%pivot_control == 4 ) THEN
       control_gsls%u = 0.0_wp ; control_gsls%action = 1
     ELSE IF ( control%pivot_control == 3 ) THEN
       control_gsls%u = 0.0_wp ; control_gsls%action = 0
     ELSE
       control_gsls%u = control%relative_pivot_tolerance ; control_gsls%action = 1
     END IF
     RETURN
     END SUBROUTINE SLS_copy_control_to_gsls

!-*-*-*-*-   S L S _ G S L S _ V A L U E _ P A R T  S U B R O U T I N E  -*-*-*-*-

     SUBROUTINE SLS_gsls_value_part( data, part, VAL, mult )

!  (gsls only; used by SBLS) the next SLS_factorize takes the values of the matrix from the registered arrays laid end
!  to end (part = 0, 1, 2), each times mult, instead of from matrix%val: gsls_set_value_part, include/gsls.h.
!  Without VAL: forget every registration.  The arrays are read when SLS_factorize runs

     TYPE ( SLS_data_type ), INTENT( INOUT ) :: data
     INTEGER, INTENT( IN ) :: part
     REAL ( KIND = wp ), INTENT( IN ), OPTIONAL, CONTIGUOUS, TARGET,            &
                                       DIMENSION( : ) :: VAL
     REAL ( KIND = wp ), INTENT( IN ), OPTIONAL :: mult

     IF ( .NOT. SLS_gsls_parts_ok( data ) ) RETURN
     CALL GSLS_set_value_part( data%gsls_keep, part, VAL, mult )
     RETURN
     END SUBROUTINE SLS_gsls_value_part

!-*-*-*-*-*-   S L S _ G S L S _ P A R T S _ O K   F U N C T I O N  -*-*-*-*-*-

     FUNCTION SLS_gsls_parts_ok( data )

!  is this an analysed gsls factorization whose values go to the backend in the caller's storage order?

     LOGICAL :: SLS_gsls_parts_ok
     TYPE ( SLS_data_type ), INTENT( IN ) :: data

     SLS_gsls_parts_ok = .FALSE.
     IF ( data%len_solver /= 4 ) RETURN
     IF ( data%solver( 1 : 4 ) /= 'gsls' ) RETURN
     SLS_gsls_parts_ok = .NOT. data%explicit_scaling
     RETURN
     END FUNCTION SLS_gsls_parts_ok

!-*-*-*-*-*-   S L S _ G S L S _ R E S I D U A L  S U B R O U T I N E  -*-*-*-*-*-

     SUBROUTINE SLS_gsls_residual( data, X, B, R )

!  (gsls only; used by SBLS) R = B - A X with the matrix of the last SLS_factorize, on the device

     TYPE ( SLS_data_type ), INTENT( INOUT ) :: data
     REAL ( KIND = wp ), INTENT( IN ), DIMENSION( : ) :: X, B
     REAL ( KIND = wp ), INTENT( OUT ), DIMENSION( : ) :: R

     CALL GSLS_residual( X, B, R, data%gsls_keep, data%gsls_inform )
     RETURN
     END SUBROUTINE SLS_gsls_residual

!-*-   S L S _ C O P Y _ I N F O R M _ F R O M _ G S L S  S U B R O U T I N E  -*-

     SUBROUTINE SLS_copy_inform_from_gsls( inform, info_gsls )

!  copy inform parameters from their GSLS equivalents (cf. SLS_copy_inform_from_ssids: the flag
!  space is that of SSIDS and the mapping is the same, with ONE deliberate difference: "not positive
!  definite" (-6) becomes GALAHAD_error_inertia, as for MA57/MA97/SYTR, not GALAHAD_error_restrictions
!  as in the ssids arm -- TRS/RQS steer their secular iteration on exactly that code
!  (src/trs/trs.f90:1957, 2287) and cannot work with the ssids arm's mapping)

     TYPE ( SLS_inform_type ), INTENT( INOUT ) :: inform
     TYPE ( gsls_inform ), INTENT( IN ) :: info_gsls

     inform%status = info_gsls%flag
     SELECT CASE( inform%status )
     CASE ( 0 : )
       inform%status = GALAHAD_ok
       inform%two_by_two_pivots = info_gsls%num_two
       inform%rank = info_gsls%matrix_rank
       inform%negative_eigenvalues = info_gsls%num_neg
       inform%delayed_pivots = info_gsls%num_delay
       inform%entries_in_factors = info_gsls%num_factor
       inform%flops_elimination = info_gsls%num_flops
       inform%max_front_size  = info_gsls%maxfront
       inform%max_depth_assembly_tree = info_gsls%maxdepth
     CASE ( - 30  )
       inform%status = GALAHAD_error_allocate
     CASE ( - 31  )
       inform%status = GALAHAD_error_deallocate
     CASE( - 1, - 2, - 3, - 4, - 5, - 9, - 10, - 12, - 13, - 14, - 15 )
       inform%status = GALAHAD_error_restrictions
     CASE ( - 11 )
       inform%status = GALAHAD_error_permutation
     CASE ( - 6, - 7, - 8  )
       inform%status = GALAHAD_error_inertia
     CASE ( - 32, GALAHAD_unavailable_option  )
       inform%status = GALAHAD_unavailable_option
     CASE DEFAULT
       inform%status = GALAHAD_error_technical
     END SELECT
     RETURN
     END SUBROUTINE SLS_copy_inform_from_gsls

"""


def main(src, dst):
    lines = open(src).read().split("\n")
    out = []
    k = 0
    shared = refine = scatter = 0
    refine_done = False
    mc68_done = False
    public_done = False
    for ln in lines:
        if re.fullmatch(r"\s*CASE \( 'ssids' \)", ln) and k < len(ARMS):
            out.extend(ARMS[k].rstrip("\n").split("\n"))
            out.append("")
            k += 1
        if "'ssids'" in ln and "'ma86'" in ln and "'ma77'" not in ln and "CASE (" in ln and "'gsls'" not in ln:
            shared += 1
            ln = ln.replace("'ssids'", "'ssids', 'gsls'")          # the shared CASE lists (:2849 analyse, :4106 factorize)
        # SLS_factorize: the host loop that scatters the values through MAPS (:4107-4150) is skipped for gsls --
        # GSLS_factor_coo maps them on the device -- unless the facade scales the scattered copy itself
        # (data%explicit_scaling, control%scaling = 1..3): then the reference's scatter + scaling block runs and
        # the arm factorizes data%matrix%VAL
        if shared == 2 and scatter == 0 and ln.strip() == "data%matrix%n = matrix%n":
            out.append(ln)
            out.append("       IF ( data%solver( 1 : data%len_solver ) /= 'gsls' .OR.                   &")
            out.append("            data%explicit_scaling ) THEN")
            scatter = 1
            continue
        if scatter == 1 and ln.strip() == "!  apply calculated scaling factors":
            while out[-1].strip() == "":
                out.pop()
            out.append("       END IF")
            out.append("")
            scatter = 2
        # SLS_solve_ir (:4770-4949): for gsls and COORDINATE storage the refinement loop runs on the device
        if ln.strip() == "!  Iterative refinement is required":
            refine += 1
        if refine == 1 and not refine_done and ln.strip() == "n = MATRIX%n":
            out.append(ln)
            out.extend(REFINE_ARM.rstrip("\n").split("\n"))
            refine_done = True
            continue
        # SLS_analyse (:2262-2267): the backend orders by itself -- nested dissection, or AMD when control%ordering asks for
        # the (approximate) minimum degree orderings that the other solvers get from MC68 (control%ordering = 1, 2)
        if ln.strip() == "mc6168_ordering = control%ordering > 0 .AND. .NOT. PRESENT( PERM )" and not mc68_done:
            assert out[-1].strip() == "CASE DEFAULT"
            out.insert(len(out) - 1, "     CASE ( 'gsls' )")
            out.insert(len(out) - 1, "       mc6168_ordering = .FALSE.")
            mc68_done = True
        if ln.strip().startswith("PUBLIC :: SLS_initialize, SLS_analyse, SLS_factorize, SLS_solve,") and not public_done:
            out.append("     PUBLIC :: SLS_gsls_value_part, SLS_gsls_parts_ok, SLS_gsls_residual")
            public_done = True
        out.append(ln)
        if ln.strip() == "USE SPRAL_SSIDS":
            out.append("     USE GALAHAD_GSLS_double")
        if ln.strip() == "TYPE ( SSIDS_inform ) :: ssids_inform" and "gsls_keep" not in "\n".join(out[-8:]) \
                and any("TYPE ( SSIDS_akeep ) :: ssids_akeep" in x for x in out[-6:]):
            out.append("       TYPE ( gsls_keep ) :: gsls_keep")
            out.append("       TYPE ( gsls_options ) :: gsls_options")
            out.append("       TYPE ( gsls_inform ) :: gsls_inform")
            out.append("       INTEGER ( KIND = long ), ALLOCATABLE, DIMENSION( : ) :: gsls_ptr")
        if ln.strip() == "END SUBROUTINE SLS_copy_inform_from_ssids":
            out.extend(COPY_ROUTINES.split("\n"))
    assert k == len(ARMS), "expected %d ssids arms, patched %d" % (len(ARMS), k)
    assert shared == 2 and refine_done and scatter == 2 and mc68_done and public_done, \
        (shared, refine, refine_done, scatter, mc68_done, public_done)
    text = "\n".join(out)
    assert "USE GALAHAD_GSLS_double" in text and "TYPE ( gsls_keep ) :: gsls_keep" in text
    open(dst, "w").write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
