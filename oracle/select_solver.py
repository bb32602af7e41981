#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Some of the reference's own test programs take whatever linear solver the package's
initialize call sets (SBLS: 'sils' = HSL MA27, a STOP stub in this tree).  This writes a scratch copy of such a
program in which, after every `CALL <PKG>_initialize( data, control, inform )`, the solver named by the environment
variable GSLS_CTEST_SOLVER (if set) is stored in the control components that name the package's linear solvers (table
ASSIGN) -- what a user would do in a few assignments.  Nothing else in the program changes; the copy never enters the repository
(oracle/build_ref.sh writes it to its scratch directory).
usage: select_solver.py <reference test .f90> <scratch copy> <PKG>"""
import re
import sys


# what a user assigns to route every factorization of the package to one solver: the SLS names, and for the
# packages that also hold an unsymmetric solver (ULS: 'gls' = HSL MA33, a stub here) the LAPACK arm 'getr'
ASSIGN = {
    "SBLS": ["control%symmetric_linear_solver = gsls_ctest_solver", "control%definite_linear_solver = gsls_ctest_solver"],
    "RQS": ["control%symmetric_linear_solver = gsls_ctest_solver", "control%definite_linear_solver = gsls_ctest_solver"],
    "TRS": ["control%symmetric_linear_solver = gsls_ctest_solver", "control%definite_linear_solver = gsls_ctest_solver"],
    "CQP": ["control%SBLS_control%symmetric_linear_solver = gsls_ctest_solver",
            "control%SBLS_control%definite_linear_solver = gsls_ctest_solver",
            "control%SBLS_control%unsymmetric_linear_solver = 'getr'",
            "control%FDC_control%use_sls = .TRUE.",       # (ULS 'gls' is a stub and this ULS knows no LAPACK arm)
            "control%FDC_control%symmetric_linear_solver = gsls_ctest_solver",
            "control%FDC_control%unsymmetric_linear_solver = 'getr'",
            "control%CRO_control%symmetric_linear_solver = gsls_ctest_solver",
            "control%CRO_control%unsymmetric_linear_solver = 'getr'"],
}


def main(src, dst, pkg):
    out, decl, calls = [], False, 0
    for ln in open(src).read().split("\n"):
        out.append(ln)
        if not decl and ln.strip().upper() == "IMPLICIT NONE":
            out.append("   CHARACTER ( LEN = 30 ) :: gsls_ctest_solver")
            out.append("   INTEGER :: gsls_ctest_len, gsls_ctest_stat")
            decl = True
        if re.fullmatch(r"\s*CALL %s_initialize\( data, control, inform \)\s*(!.*)?" % pkg, ln):
            ind = ln[: len(ln) - len(ln.lstrip())]
            out.append(ind + "CALL GET_ENVIRONMENT_VARIABLE( 'GSLS_CTEST_SOLVER', gsls_ctest_solver,               &")
            out.append(ind + "                               gsls_ctest_len, gsls_ctest_stat )")
            out.append(ind + "IF ( gsls_ctest_stat == 0 .AND. gsls_ctest_len > 0 ) THEN")
            for a in ASSIGN[pkg]:
                out.append(ind + "  " + a)
            out.append(ind + "END IF")
            calls += 1
    assert decl and calls >= 1, (decl, calls)
    open(dst, "w").write("\n".join(out))


if __name__ == "__main__":
    main(*sys.argv[1:4])
