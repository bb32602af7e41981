"""GPU (-m gpu): the reference's OWN test programs and spec-sheet examples of the layers above SLS -- compiled where they
lie under /root/reference by oracle/build_ref.sh, above the patched SLS / SBLS facade, the solver named 'gsls' -- against
the outputs the reference stores for them (tests/golden/*.output are copies of those data files: src/sbls/sblsdt.output,
src/cqp/cqpds.output, src/rqs/rqsds.output, src/trs/trsds.output; tests/golden/QPBAND.qplib = examples/QPBAND.qplib).

  * sblsti_gsls         src/sbls/sblsti.f90: SBLS's interface layer (SBLS_import / factorize_matrix / solve_system -- the
                        calls include/sbls.h's C functions wrap one to one), seven storage schemes
  * cqps_gsls           src/cqp/cqps.f90 (CQP spec-sheet example)
  * rqss_gsls           src/rqs/rqss.f90 (RQS spec-sheet example; north_star names TRS *and* RQS)
  * trss_gsls           src/trs/trss.f90
  * runcqp_qplib_gsls   src/cqp/incqp.f90: BASELINE.json configs[0] as the reference runs it (QPLIB file read by RPD, solvers
                        named in the spec file RUNCQP.SPC)
"""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REFDIR = os.path.join(HERE, "..", "oracle", "_ref")
GOLD = os.path.join(HERE, "golden")


def _run(exe, cwd=None, stdin=None, solver="gsls"):
    path = os.path.join(REFDIR, exe)
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/%s not built (oracle/build_ref.sh needs /root/reference)" % exe)
    env = dict(os.environ, GSLS_CTEST_SOLVER=solver, OMP_NUM_THREADS="4", OMP_CANCELLATION="true")
    r = subprocess.run([path], cwd=cwd, stdin=stdin, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    return r.stdout


def _floats(s):
    return [float(x) for x in re.findall(r"[-+]?\d\.\d+[Ee][-+]\d+", s)]


def test_reference_sbls_interface_test_with_gsls():
    """src/sbls/sblsti.f90: K = [H A^T; A -C] in seven storage schemes through SBLS_import / SBLS_factorize_matrix /
    SBLS_solve_system with preconditioner 2, factorization 2.  The stored output of the same seven systems through the C
    interface (sblsdt.output) has residuals <= 8.9e-16; the bar here is ten times the largest of them."""
    out = _run("sblsti_gsls")
    stored = _floats(open(os.path.join(GOLD, "sblsdt.output")).read())
    assert len(stored) == 7
    rows = [ln.split() for ln in out.splitlines() if ln.strip() and ln.split()[0] in
            ("coordinate", "sparse-by-rows", "dense", "diagonal", "scaled-identity", "identity", "zero")]
    assert len(rows) == 7, out
    for r in rows:
        assert r[1:4] == ["2", "2", "0"], out             # preconditioner, factorization, status
        assert float(r[4]) <= 10 * max(stored), out


def test_cqp_spec_sheet_example_matches_the_stored_output():
    """src/cqp/cqps.f90 vs src/cqp/cqpds.output: the optimal value and solution to the printed digits (the iteration
    count is a property of the linear solver's rounding, not of the path)."""
    out = _run("cqps_gsls")
    ref = open(os.path.join(GOLD, "cqpds.output")).read()
    assert "CQP_solve exit status" not in out, out
    a, b = _floats(out), _floats(ref)
    assert len(a) == len(b) == 4, (out, ref)
    assert np.allclose(a, b, rtol=2e-4, atol=0), (out, ref)        # five printed digits
    it = int(re.search(r"CQP: (\d+) iterations", out).group(1))
    assert abs(it - 10) <= 3, out


def test_rqs_spec_sheet_example_matches_the_stored_output():
    """src/rqs/rqss.f90 vs src/rqs/rqsds.output (n = 10 000, regularised quadratic subproblem: every factorization of
    H + lambda M and the part solves for the secular-equation derivatives go through SLS('gsls'))."""
    out = _run("rqss_gsls")
    ref = open(os.path.join(GOLD, "rqsds.output")).read()
    assert "exit status" not in out, out
    assert np.allclose(_floats(out), _floats(ref), rtol=2e-4, atol=0), (out, ref)
    assert int(out.split()[0]) == int(ref.split()[0]), (out, ref)     # factorizations


def test_trs_spec_sheet_example_matches_the_stored_output():
    out = _run("trss_gsls")
    ref = open(os.path.join(GOLD, "trsds.output")).read()
    assert "exit status" not in out, out
    assert np.allclose(_floats(out), _floats(ref), rtol=2e-4, atol=0), (out, ref)
    assert int(out.split()[0]) == int(ref.split()[0]), (out, ref)


SPEC = """BEGIN RUNCQP SPECIFICATIONS
   print-full-solution                               yes
   write-solution                                    yes
   solution-file-name                                CQPSOL.d
   write-result-summary                              yes
   result-summary-file-name                          CQPRES.d
END RUNCQP SPECIFICATIONS

BEGIN CQP SPECIFICATIONS
   remove-linear-dependencies                        no
   cross-over-solution                               no
END CQP SPECIFICATIONS

BEGIN SBLS SPECIFICATIONS
   symmetric-linear-equation-solver                  gsls
   definite-linear-equation-solver                   gsls
END SBLS SPECIFICATIONS
"""


def test_qpband_qplib_through_the_reference_main_program_with_a_spec_file(tmp_path):
    """BASELINE.json configs[0] the way the reference runs it (src/cqp/makemaster:57, bin/dgal): examples/QPBAND.qplib read
    by RPD (src/rpd/rpd.f90) inside the reference's main program src/cqp/incqp.f90, the linear solver chosen by KEYWORD
    in RUNCQP.SPC (sbls.f90's read_specfile -> 'symmetric-linear-equation-solver gsls').  The reference stores no output
    for this run; the optimal value is compared with an independent solve of the same five-variable QP."""
    import problems as P
    (tmp_path / "RUNCQP.SPC").write_text(SPEC)
    with open(os.path.join(GOLD, "QPBAND.qplib")) as f:
        out = _run("runcqp_qplib_gsls", cwd=str(tmp_path), stdin=f, solver="")
    assert "GSLS symmetric equation solver used" in out, out[-1500:]
    res = (tmp_path / "CQPRES.d").read_text().split()
    assert res[0] == "QPBAND" and [int(v) for v in res[1:3]] == [5, 2], res
    status, obj = int(res[6]), float(res[5])
    assert status == 0, out[-1500:]
    n, m, H, A, g, c_l, c_u, x_l, x_u = P.qpband(5)
    from scipy.optimize import minimize, LinearConstraint, Bounds
    import scipy.sparse as sp
    Hm = sp.coo_matrix((H[2], (H[0] - 1, H[1] - 1)), shape=(n, n)).toarray()
    Hm = Hm + Hm.T - np.diag(np.diag(Hm))
    Am = sp.coo_matrix((A[2], (A[0] - 1, A[1] - 1)), shape=(m, n)).toarray()
    ref = minimize(lambda v: 0.5 * v @ Hm @ v + g @ v, np.ones(n), jac=lambda v: Hm @ v + g, method="SLSQP",
                   bounds=Bounds(x_l, x_u), constraints=[LinearConstraint(Am, c_l, np.inf)],
                   options={"ftol": 1e-14, "maxiter": 1000})
    assert ref.success and abs(ref.fun - obj) <= 1e-4 * max(1.0, abs(ref.fun)), (ref.fun, obj)
