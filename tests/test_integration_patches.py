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
    """...and, since round 3, the three value copies of SBLS_form_n_factorize_explicit plus the in-place solve (f1)."""
    import patch_sbls
    src = open(os.path.join(REF, "sbls", "sbls.f90")).read()
    out = patch_sbls.patch(src)
    lines = src.split("\n")

    def span(name):
        b = next(k for k, ln in enumerate(lines) if ln.strip().startswith("SUBROUTINE %s(" % name))
        e = next(k for k, ln in enumerate(lines) if ln.strip().startswith("END SUBROUTINE %s" % name))
        return b, e
    beg, end = span("SBLS_solve_explicit")
    fbeg, fend = span("SBLS_form_n_factorize_explicit")
    ops = _changed(src, out)
    in_solve = [op for op in ops if beg < op[1] and op[2] <= end]
    in_form = [op for op in ops if fbeg < op[1] and op[2] <= fend]
    other = [op for op in ops if op not in in_solve and op not in in_form]
    assert len(in_solve) == 5, in_solve   # declaration, in-place arm, loop header, control argument, residual test
    assert len(in_form) == 5, in_form     # declaration, decision + registration, the copies of A%val, H%val, -C%val
    assert len(other) == 1 and "gsls_stale" in "\n".join(out.split("\n")[other[0][3]:other[0][4]]), other
    # no statement of the reference is lost: replaced lines are the three IF headers that gained ".NOT. gsls_parts" and the
    # four of the refinement loop
    removed = [ln.strip() for tag, i1, i2, j1, j2 in ops if tag in ("delete", "replace") for ln in lines[i1:i2] if ln.strip()]
    assert len(removed) <= 8, removed
    assert out.count("K_control_ir") == 11 and out.count("itref_loop") == 5
    assert "DO iter = 0, itref_loop" in out and out.count("DO iter = 0, control%itref_max") == \
        src.count("DO iter = 0, control%itref_max") - 1       # the other solve routines keep their loops
    # the host copies are skipped only behind the flag, and a stale K%val is refreshed on the reference's path
    assert out.count(".NOT. gsls_parts") == 3
    assert "new_a = MAX( new_a, 1 ) ; new_h = MAX( new_h, 1 )" in out and "efactors%gsls_stale = .TRUE." in out
    assert "CALL SLS_gsls_value_part( efactors%K_data, - 1 )" in out
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
