"""TRS (BASELINE.json configs[3]): the reference's TRS_solve (src/trs/trs.f90, unchanged) above the
patched SLS facade with definite_linear_solver = 'gsls'.  TRS analyses H + lambda*M once and then
factorizes it again and again with pivot_control = 2 (Cholesky; "not positive definite" is an expected
signal while lambda is too small, trs.f90:1942-1964), solves with iterative refinement (IR_solve) and
uses SLS_part_solve for the secular-equation derivatives (trs.f90:2618-2742) -- which the reference's
own ssids arm cannot do (sls.f90:6886-6888) and gsls can.

Parity: the package's stored output (src/trs/trss.f90 + trsds.output: 4 factorizations,
f = -7.0611E+02, multiplier = 7.0712E+00), the dense 'sytr' arm of the reference on the same small
problems, and the KKT conditions of the subproblem at full size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _need():
    from oracle import refio
    if not refio.trs_available(dropin=True):
        pytest.skip("oracle/_ref/trs_gsls_driver not built (needs /root/reference at build time)")
    return refio


def _tridiag(n, d, o):
    i = np.arange(1, n + 1)
    return (np.concatenate([i, i[1:]]), np.concatenate([i, i[:-1]]),
            np.concatenate([np.full(n, d), np.full(n - 1, o)]))


def _hx(n, H, x):
    r, c, v = H[0] - 1, H[1] - 1, H[2]
    y = np.zeros(n)
    np.add.at(y, r, v * x[c])
    off = r != c
    np.add.at(y, c[off], v[off] * x[r[off]])
    return y


def test_trs_spec_sheet_example():
    """src/trs/trss.f90: n = 10000, H = tridiag(-2, 1), M = 2 I, c = 1, f = 1, radius = 10."""
    refio = _need()
    n = 10000
    r = refio.run_trs(n, _tridiag(n, -2.0, 1.0), np.ones(n), 10.0, 1.0, solver="gsls", mdiag=2.0)
    assert r["status"] == 0
    assert abs(r["obj"] - (-7.0611e2)) <= 0.5e-2 and abs(r["multiplier"] - 7.0712) <= 0.5e-4


@pytest.mark.parametrize("n,d,radius", [(300, -2.0, 10.0), (800, 1.0, 2.0), (1200, -0.5, 50.0)])
def test_trs_matches_reference_dense_arm(n, d, radius):
    refio = _need()
    rng = np.random.default_rng(n)
    H = _tridiag(n, d, 1.0)
    c = rng.uniform(-1, 1, n)
    a = refio.run_trs(n, H, c, radius, 0.5, solver="gsls")
    b = refio.run_trs(n, H, c, radius, 0.5, solver="sytr")
    assert a["status"] == 0 and b["status"] == 0
    assert abs(a["obj"] - b["obj"]) <= 1e-8 * max(1.0, abs(b["obj"]))
    assert abs(a["multiplier"] - b["multiplier"]) <= 1e-7 * max(1.0, abs(b["multiplier"]))
    assert np.abs(a["x"] - b["x"]).max() <= 1e-6 * max(1.0, np.abs(b["x"]).max())


def test_trs_cfg4_full_size():
    """5-point Laplacian on a 707 x 707 grid, H = L - I (indefinite), c = -0.5, radius = 1
    (src/trs/trs_paper_large.f90 setting): check the optimality conditions of the answer."""
    refio = _need()
    nx = 707
    n = nx * nx
    idx = np.arange(n).reshape(nx, nx)
    row = np.concatenate([idx.ravel(), idx[:, 1:].ravel(), idx[1:, :].ravel()]) + 1
    col = np.concatenate([idx.ravel(), idx[:, :-1].ravel(), idx[:-1, :].ravel()]) + 1
    val = np.concatenate([np.full(n, 3.0), np.full(idx[:, 1:].size, -1.0), np.full(idx[1:, :].size, -1.0)])
    H = (row, col, val)
    c = np.full(n, -0.5)
    r = refio.run_trs(n, H, c, 1.0, 0.0, solver="gsls", timeout=1500)
    assert r["status"] == 0, r["status"]
    x, lam = r["x"], r["multiplier"]
    assert lam >= 0.0 and abs(np.linalg.norm(x) - 1.0) <= 1e-8          # boundary solution
    g = _hx(n, H, x) + lam * x + c                                        # (H + lambda I) x = -c
    assert np.abs(g).max() <= 1e-8 * max(1.0, np.abs(c).max())
    print("cfg4 through TRS: %d factorizations, %.3f s, multiplier %.6f, f %.6f" % (
        r["factorizations"], r["time"], lam, r["obj"]))
