#!/usr/bin/env python3
"""Writes a copy of the reference's src/sbls/sbls.f90 in which SBLS_solve_explicit leaves its refinement loop to the
backend when K is factorized by 'gsls':

    python3 integration/patch_sbls.py /path/to/GALAHAD/src/sbls/sbls.f90 out.f90

SBLS_solve_explicit (sbls.f90:5073-5388) refines x by   DO iter = 0, itref_max: solve K dx = r; x = x + dx;
r = b - K x (three host loops over the entries of K, :5343-5372).  With the patched SLS (patch_sls.py) one SLS_solve
with control%max_iterative_refinements = itref_max runs the same recurrence on the device with every vector resident in
HBM (gsls_solve_ir): the loop here then makes ONE pass.  The acceptable residuals are set to zero so that exactly
itref_max refinements are made, as SBLS makes them.  Everything else -- the Schur-complement branch, the residual asked
for by control%get_norm_residual, every other solver -- is the reference's code, untouched.

Two more changes serve the same arm (K's values assembled on the device, vectors left where they are):

  * SBLS_form_n_factorize_explicit (sbls.f90:3319-3984) copies A%val, H%val and -C%val into K%val on the host before every
    factorization (19 MB per interior-point iteration at n = 1.2e6, a millisecond of one core).  On a refactorization
    (same structure, G = H, no perturbation) the patched routine registers the three arrays with the backend instead
    (SLS_gsls_value_part -> gsls_set_value_part, include/gsls.h): they cross the link from where the caller keeps them
    and K's values are put together in HBM.  K%val on the host is then stale (efactors%gsls_stale); the next
    factorization that takes the reference's path copies all three parts again.
  * SBLS_solve_explicit (sbls.f90:5126-5128, 5249) copies the right-hand side twice, zeroes SOL and adds the solution
    back: four passes over n + m doubles around a solve that needs none of them.  The gsls arm solves in SOL
    (SLS_solve with the refinement count of SBLS); control%get_norm_residual is served by the backend's residual
    (SLS_gsls_residual -> gsls_residual: b - K x with the matrix that was factorized).

The reference file is read, never modified; the output belongs in a scratch directory (oracle/build_ref.sh) or in the
user's own GALAHAD tree.
"""
import sys

DECL = """      INTEGER :: itref_loop
      TYPE ( SLS_control_type ) :: K_control_ir"""

SETUP = """!  gsls: the refinement loop runs inside SLS_solve (SLS_solve_ir on the device, every vector resident in HBM): one
!  call with max_iterative_refinements = itref_max replaces itref_max + 1 solves and the host residuals between them

      itref_loop = control%itref_max
      K_control_ir = efactors%K_control
      IF ( inform%factorization /= 1 .AND. control%itref_max > 0 .AND.         &
           ( TRIM( control%symmetric_linear_solver ) == 'gsls' .OR.            &
             TRIM( control%symmetric_linear_solver ) == 'GSLS' ) ) THEN
        K_control_ir%max_iterative_refinements = control%itref_max
        K_control_ir%acceptable_residual_relative = zero
        K_control_ir%acceptable_residual_absolute = zero
        itref_loop = 0
      END IF
"""


FORM_DECIDE = """
!  gsls, a refactorization with G = H: A%val, H%val and -C%val go to the backend as they are and K's values are put
!  together in HBM (SLS_gsls_value_part); K%val on the host is then stale until a factorization on the reference's path
!  has copied every part again

      gsls_parts = .NOT. resize .AND. new_a <= 1 .AND. new_h <= 1 .AND.        &
        new_c <= 1 .AND. .NOT. efactors%analyse .AND.                          &
        inform%preconditioner == 2 .AND. .NOT. inform%perturbed .AND.          &
        .NOT. PRESENT( H_lm ) .AND. a_ne > 0 .AND. h_ne > 0 .AND.              &
        .NOT. ( control%print_level >= 4 .AND. control%out > 0 ) .AND.         &
        SLS_gsls_parts_ok( efactors%K_data )
      IF ( gsls_parts ) gsls_parts = ALLOCATED( A%val ) .AND. ALLOCATED( H%val )
      IF ( gsls_parts ) gsls_parts = SMT_get( H%type ) /= 'SCALED_IDENTITY'    &
        .AND. SMT_get( H%type ) /= 'IDENTITY'
      IF ( gsls_parts .AND. c_ne > 0 ) gsls_parts = ALLOCATED( C%val ) .AND.   &
        SMT_get( C%type ) /= 'SCALED_IDENTITY' .AND.                           &
        SMT_get( C%type ) /= 'IDENTITY'
      IF ( gsls_parts ) THEN
        CALL SLS_gsls_value_part( efactors%K_data, 0, A%val( : a_ne ) )
        CALL SLS_gsls_value_part( efactors%K_data, 1, H%val( : h_ne ) )
        IF ( c_ne > 0 ) CALL SLS_gsls_value_part( efactors%K_data, 2,          &
                                                  C%val( : c_ne ), - one )
        efactors%gsls_stale = .TRUE.
      ELSE
        CALL SLS_gsls_value_part( efactors%K_data, - 1 )
        IF ( efactors%gsls_stale ) THEN
          new_a = MAX( new_a, 1 ) ; new_h = MAX( new_h, 1 )
          new_c = MAX( new_c, 1 )
          efactors%gsls_stale = .FALSE.
        END IF
      END IF
"""

SOLVE_FAST = """
!  gsls, factors of the augmented system: solve in SOL, the refinement inside SLS_solve (every vector resident in HBM);
!  the residual norm, if asked for, from the backend (b - K x with the matrix that was factorized)

      IF ( inform%factorization /= 1 .AND.                                     &
           SLS_gsls_parts_ok( efactors%K_data ) .AND.                          &
           .NOT. ( control%print_level >= 4 .AND. control%out > 0 ) ) THEN
        K_control_ir = efactors%K_control
        K_control_ir%max_iterative_refinements = MAX( control%itref_max, 0 )
        K_control_ir%acceptable_residual_relative = zero
        K_control_ir%acceptable_residual_absolute = zero
        IF ( control%get_norm_residual )                                       &
          efactors%RHS_orig( : npm ) = SOL( : npm )
        CALL SLS_solve( efactors%K, SOL, efactors%K_data, K_control_ir,        &
                        inform%SLS_inform )
        inform%sls_solve_status = inform%SLS_inform%status
        IF ( inform%sls_solve_status < 0 ) THEN
          IF ( control%out > 0 .AND. control%print_level > 0 )                 &
            WRITE( control%out, "( A, ' solve exit status = ', I0 )" )         &
              prefix, inform%sls_solve_status
          inform%status = GALAHAD_error_solve
          RETURN
        END IF
        IF ( control%get_norm_residual ) THEN
          CALL SLS_gsls_residual( efactors%K_data, SOL( : npm ),               &
                                  efactors%RHS_orig( : npm ),                  &
                                  efactors%RHS( : npm ) )
          inform%norm_residual = MAXVAL( ABS( efactors%RHS( : npm ) ) )
        END IF
        inform%status = GALAHAD_ok
        RETURN
      END IF
"""


def patch(src):
    lines = src.split("\n")
    out = []
    inside = False
    inform = False
    done = {"decl": False, "loop": False, "call": False, "test": False, "stale": False, "fdecl": False,
            "decide": False, "site_a": False, "site_h": False, "site_c": False, "fast": False}
    i = 0
    while i < len(lines):
        ln = lines[i]
        s = ln.strip()
        if s.startswith("SUBROUTINE SBLS_solve_explicit("):
            inside = True
        if s.startswith("SUBROUTINE SBLS_form_n_factorize_explicit("):
            inform = True
        if inform and s.startswith("END SUBROUTINE SBLS_form_n_factorize_explicit"):
            inform = False
        # ---- the type that holds K: a flag "K%val on the host is behind the values that were factorized"
        if not done["stale"] and s == "TYPE ( SLS_data_type ) :: K_data":
            out.append(ln)
            out.append("        LOGICAL :: gsls_stale = .FALSE.")
            done["stale"] = True
            i += 1
            continue
        # ---- SBLS_form_n_factorize_explicit: the three value copies
        if inform and not done["fdecl"] and s == "LOGICAL :: printi, resize, use_schur_complement":
            out.append(ln)
            out.append("      LOGICAL :: gsls_parts")
            done["fdecl"] = True
            i += 1
            continue
        if inform and not done["decide"] and s == "efactors%k_pert = efactors%k_c + c_ne":
            out.append(ln)
            out.extend(FORM_DECIDE.rstrip("\n").split("\n"))
            done["decide"] = True
            i += 1
            continue
        if inform and done["decide"] and not done["site_a"] and \
                s == "IF ( resize .OR. new_a > 0 ) efactors%K%val( : a_ne ) = A%val( : a_ne )":
            out.append("      IF ( ( resize .OR. new_a > 0 ) .AND. .NOT. gsls_parts )                  &")
            out.append("        efactors%K%val( : a_ne ) = A%val( : a_ne )")
            done["site_a"] = True
            i += 1
            continue
        if inform and done["site_a"] and not done["site_h"] and \
                s == "IF ( resize .OR. new_a > 1 .OR. new_h > 0 ) THEN":       # (the first one: CASE( 2 ), G = H)
            out.append("        IF ( ( resize .OR. new_a > 1 .OR. new_h > 0 ) .AND.                    &")
            out.append("             .NOT. gsls_parts ) THEN")
            done["site_h"] = True
            i += 1
            continue
        if inform and done["site_h"] and not done["site_c"] and s == "IF ( new_c > 0 ) THEN":
            out.append(ln.replace("IF ( new_c > 0 ) THEN", "IF ( new_c > 0 .AND. .NOT. gsls_parts ) THEN"))
            done["site_c"] = True
            i += 1
            continue
        # ---- SBLS_solve_explicit: in place
        if inside and done["decl"] and not done["fast"] and s == "efactors%RHS_orig( : npm ) = SOL( : npm )":
            while out[-1].strip() == "" or out[-1].strip() == "!  Compute the original residual":
                out.pop()
            out.extend(SOLVE_FAST.rstrip("\n").split("\n"))
            out.append("")
            out.append("!  Compute the original residual")
            out.append("")
            done["fast"] = True
        if inside and s.startswith("END SUBROUTINE SBLS_solve_explicit"):
            inside = False
        if inside and not done["decl"] and s == "CHARACTER ( LEN = 80 ) :: array_name":
            out.append(ln)
            out.extend(DECL.split("\n"))
            done["decl"] = True
            i += 1
            continue
        if inside and not done["loop"] and s == "DO iter = 0, control%itref_max":
            out.extend(SETUP.rstrip("\n").split("\n"))
            out.append("      DO iter = 0, itref_loop")
            done["loop"] = True
            i += 1
            continue
        # the augmented-system branch: CALL SLS_solve( efactors%K, efactors%RHS, efactors%K_data, & / efactors%K_control, ...
        if inside and done["loop"] and not done["call"] and \
                s.startswith("CALL SLS_solve( efactors%K, efactors%RHS, efactors%K_data,"):
            nxt = lines[i + 1]
            assert "efactors%K_control, inform%SLS_inform )" in nxt, nxt
            out.append(ln)
            out.append(nxt.replace("efactors%K_control,", "K_control_ir,"))
            done["call"] = True
            i += 2
            continue
        if inside and not done["test"] and s == "IF ( iter < control%itref_max .OR. control%get_norm_residual ) THEN":
            out.append(ln.replace("control%itref_max", "itref_loop"))
            done["test"] = True
            i += 1
            continue
        out.append(ln)
        i += 1
    missing = [k for k, v in done.items() if not v]
    if missing:
        raise SystemExit("patch_sbls: anchors not found: %s (has the reference's sbls.f90 changed?)" % missing)
    return "\n".join(out)


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    with open(sys.argv[1]) as f:
        src = f.read()
    with open(sys.argv[2], "w") as f:
        f.write(patch(src))
