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


def patch(src):
    lines = src.split("\n")
    out = []
    inside = False
    done = {"decl": False, "loop": False, "call": False, "test": False}
    i = 0
    while i < len(lines):
        ln = lines[i]
        s = ln.strip()
        if s.startswith("SUBROUTINE SBLS_solve_explicit("):
            inside = True
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
