"""CPU: the plain-C restatement (oracle/gsls_oracle.c) is pinned against the golden vectors produced
by the real reference (tests/golden/make_golden.py) and against the reference's own known-answer
systems (src/sls/slst.f90:29-51)."""
import glob
import os
import subprocess

import numpy as np
import pytest

import problems as P

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = sorted(p for p in glob.glob(os.path.join(HERE, "golden", "*.npz")) if not os.path.basename(p).startswith("scaled_"))


@pytest.fixture(scope="session", autouse=True)
def built_oracle():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)


def test_golden_present():
    assert len(GOLDEN) >= 12


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_matches_reference(path):
    from oracle.oracle import Oracle, lower_csc
    g = np.load(path)
    n = int(g["n"])
    ptr, row, val = lower_csc(n, g["row"], g["col"], g["val"])
    o = Oracle(n, ptr, row, g["perm"], int(g["nemin"]))
    sym = o.symbolic()
    # integer / index work: bit-exact
    assert sym["nnodes"] == int(g["ref_nnodes"])
    for k in ("sptr", "sparent", "rptr", "rlist", "order", "nptr", "nlist"):
        assert np.array_equal(sym[k], g["ref_" + k]), k
    st = o.stats()
    assert st["num_factor"] == int(g["ref_num_factor"])
    assert st["num_flops"] == int(g["ref_num_flops"])
    flag = o.factor(val, bool(g["posdef"]), u=0.01, small=2.220446049250313e-16)
    assert flag == 0
    st = o.stats()
    assert st["num_neg"] == int(g["ref_neg"])          # inertia exact
    x = o.solve(g["rhs"])
    # floating point: forward error against the reference's solution, scaled residual
    scale = max(1.0, np.abs(g["ref_x"]).max())
    assert np.abs(x - g["ref_x"]).max() <= 1e-9 * scale
    rhs, X = np.atleast_2d(g["rhs"].T).T, np.atleast_2d(x.T).T
    for k in range(X.shape[1]):
        assert P.scaled_residual(n, g["row"], g["col"], g["val"], X[:, k], rhs[:, k]) <= 1e-11
    o.close()


def test_known_answer_systems():
    """x = 1..5 to sqrt(eps), the reference's own bar (src/sls/slst.f90:262)."""
    from oracle.oracle import Oracle, lower_csc
    for prob, posdef in ((P.kat_indefinite(), False), (P.kat_definite(), True), (P.kat_definite(), False)):
        n, row, col, val, rhs, xs = prob
        for perm in (np.arange(1, n + 1), np.arange(n, 0, -1)):
            ptr, r, v = lower_csc(n, row, col, val)
            o = Oracle(n, ptr, r, perm)
            assert o.factor(v, posdef) == 0
            assert np.abs(o.solve(rhs) - xs).max() <= np.sqrt(np.finfo(float).eps)
            o.close()


def test_oracle_partial_solves():
    """L, D, U part solves compose to the full solve (src/sls/slst.f90 part-solve sweep)."""
    from oracle.oracle import Oracle, lower_csc
    n, row, col, val, rhs, xs = P.kkt_qpband(120, 30)
    ptr, r, v = lower_csc(n, row, col, val)
    o = Oracle(n, ptr, r, np.arange(1, n + 1))
    assert o.factor(v, False) == 0
    full = o.solve(rhs)
    step = o.solve(o.solve(o.solve(rhs, job=1), job=2), job=3)
    assert np.abs(full - step).max() <= 1e-12 * max(1, np.abs(full).max())
    assert np.abs(o.solve(o.solve(rhs, job=1), job=4) - full).max() <= 1e-12 * max(1, np.abs(full).max())
    o.close()


def test_oracle_not_posdef_and_singular():
    from oracle.oracle import Oracle, lower_csc
    n, row, col, val, rhs, xs = P.kat_indefinite()
    ptr, r, v = lower_csc(n, row, col, val)
    o = Oracle(n, ptr, r, np.arange(1, n + 1))
    assert o.factor(v, True) == -6                      # SSIDS_ERROR_NOT_POS_DEF
    o.close()
    # structurally fine but numerically singular: zero matrix with action -> warning 7, rank 0
    n = 4
    row = col = np.arange(1, n + 1, dtype=np.int32)
    ptr, r, v = lower_csc(n, row, col, np.zeros(n))
    o = Oracle(n, ptr, r, np.arange(1, n + 1))
    assert o.factor(v, False, action=True) == 7
    assert o.stats()["num_zero"] == n
    assert o.factor(v, False, action=False) == -5       # SSIDS_ERROR_SINGULAR
    o.close()
