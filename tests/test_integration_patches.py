"""The integration scripts against the reference's sources (CPU, needs /root/reference; skipped on the GPU box): every
anchor they look for exists exactly once, the edits are the documented ones and nothing else in the files changes."""
import difflib
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "integration"))
REF = "/root/reference/src"


def _changed(a, b):
    sm = difflib.SequenceMatcher(None, a.split("\n"), b.split("\n"), autojunk=False)
    return [op for op in sm.get_opcodes() if op[0] != "equal"]


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "sbls", "sbls.f90")), reason="reference tree not present")
def test_patch_sbls_edits_only_the_refinement_loop_of_solve_explicit():
    import patch_sbls
    src = open(os.path.join(REF, "sbls", "sbls.f90")).read()
    out = patch_sbls.patch(src)
    lines = src.split("\n")
    beg = next(k for k, ln in enumerate(lines) if ln.strip().startswith("SUBROUTINE SBLS_solve_explicit("))
    end = next(k for k, ln in enumerate(lines) if ln.strip().startswith("END SUBROUTINE SBLS_solve_explicit"))
    ops = _changed(src, out)
    assert len(ops) == 4, ops                                    # declaration, loop header, control argument, residual test
    assert all(beg < op[1] and op[2] <= end for op in ops), ops  # all inside SBLS_solve_explicit (sbls.f90:5073-5388)
    assert out.count("K_control_ir") == 6 and out.count("itref_loop") == 5
    assert "DO iter = 0, itref_loop" in out and out.count("DO iter = 0, control%itref_max") == \
        src.count("DO iter = 0, control%itref_max") - 1       # the other solve routines keep their loops
    # applying it twice must fail loudly, not produce a half-patched file
    with pytest.raises(SystemExit):
        patch_sbls.patch(out)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "sls", "sls.f90")), reason="reference tree not present")
def test_patch_sls_adds_arms_and_keeps_the_rest(tmp_path):
    import subprocess
    dst = tmp_path / "sls_gsls.f90"
    script = os.path.join(HERE, "..", "integration", "patch_sls.py")
    subprocess.run([sys.executable, script, os.path.join(REF, "sls", "sls.f90"), str(dst)], check=True)
    src = open(os.path.join(REF, "sls", "sls.f90")).read()
    out = dst.read_text()
    assert out.count("CASE ( 'gsls' )") >= 7          # initialize_solver, analyse, factorize, solve (x2), terminate, enquire ...
    assert "GSLS_solve_ir" in out and "SLS_copy_control_to_gsls" in out
    # no line of the reference is lost: the patch only inserts (and wraps the host scatter in one IF)
    removed = [ln for tag, i1, i2, j1, j2 in _changed(src, out) if tag in ("delete", "replace")
               for ln in src.split("\n")[i1:i2] if ln.strip()]
    assert len(removed) <= 4, removed[:10]
    # explicit scalings (control%scaling = 1..3): the host scatter + the reference's scaling block must still
    # run for gsls, and the arm must then factorize the scaled copy data%matrix%VAL, not the caller's values
    i_if = out.index("IF ( data%solver( 1 : data%len_solver ) /= 'gsls' .OR.")
    i_sc = out.index("!  apply calculated scaling factors", i_if)
    assert "data%explicit_scaling ) THEN" in out[i_if:i_if + 200]
    assert i_if < out.index("data%matrix%VAL( k ) = matrix%VAL( l )", i_if) < i_sc
    arm = out[out.index("CALL GSLS_factor( data%must_be_definite") - 200:out.index("CALL GSLS_factor_coo( data%must_be_definite")]
    assert "IF ( data%explicit_scaling ) THEN" in arm and "data%matrix%VAL(" in arm
