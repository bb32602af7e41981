"""SBLS row of the scope table: the reference's SBLS_form_and_factorize / SBLS_solve
(src/sbls/sbls.f90:1695-1903, 4937-5388), built above the patched SLS facade with
control%symmetric_linear_solver = 'gsls'; the only change to SBLS is that SBLS_solve_explicit hands its refinement loop
to the backend (integration/patch_sbls.py: one SLS_solve with max_iterative_refinements = itref_max, on the device).  SBLS assembles K = [H A^T; A -C] in COORDINATE form,
analyses/factorizes through SLS, checks the inertia itself (exactly m negative eigenvalues,
sbls.f90:4166-4224) and refines the solve; all of that is the reference's own Fortran, only the
factorize+solve underneath runs on the MI355X.

The reference's own CPU path cannot be timed beside it here: SBLS passes no PERM, and the orderings
SLS would call for ssids (METIS / MC68) are stubs in the tree -- so parity for SBLS is
pinned by the package's own known answer (src/sbls/sblss.f90 + sblsds.output: solution all ones) and
by constructed solutions."""
import numpy as np
import pytest

import problems as P


def _need():
    from oracle import refio
    if not refio.sbls_available(dropin=True):
        pytest.skip("oracle/_ref/sbls_gsls_driver not built (needs /root/reference at build time)")
    return refio


SBLSS = dict(n=3, m=2, H=([1, 2, 3, 3], [1, 2, 3, 1], [1.0, 2.0, 3.0, 4.0]),
             A=([1, 1, 2, 2], [1, 2, 2, 3], [2.0, 1.0, 1.0, 1.0]), C=([2], [1], [1.0]),
             rhs=[7.0, 4.0, 8.0, 2.0, 1.0])                     # src/sbls/sblss.f90:13-29


def test_sbls_reaches_backend_and_fails_loudly_without_gpu(have_gpu):
    refio = _need()
    k = SBLSS
    r = refio.run_sbls(k["n"], k["m"], k["H"], k["A"], k["C"], k["rhs"], solver="gsls")
    if have_gpu:
        assert r["status_factorize"] == 0
    else:
        assert r["status_factorize"] == -10          # GALAHAD_error_factorization, no fake result


@pytest.mark.gpu
@pytest.mark.parametrize("factorization", [0, 1, 2])
def test_sbls_spec_sheet_example(factorization):
    """src/sbls/sblss.f90 / sblsds.output: ' SBLS: Solution = 1.0 1.0 1.0 1.0 1.0' for the automatic
    choice, the Schur-complement and the augmented-system factorizations."""
    refio = _need()
    k = SBLSS
    r = refio.run_sbls(k["n"], k["m"], k["H"], k["A"], k["C"], k["rhs"], solver="gsls",
                       factorization=factorization)
    assert (r["status_factorize"], r["status_solve"]) == (0, 0)
    assert np.abs(r["sol"] - 1.0).max() <= 1e-12


def _qp_kkt(n, m, seed):
    """cfg3 generator split into the H, A, C that SBLS wants (tests/problems.py:kkt_qpband)."""
    rng = np.random.default_rng(seed)
    sigma = 10.0 ** rng.uniform(-4, 4, n)
    i = np.arange(1, n + 1)
    H = (np.concatenate([i, i[1:]]), np.concatenate([i, i[:-1]]),
         np.concatenate([2.0 + sigma, np.full(n - 1, -1.0)]))
    j = np.arange(1, m + 1)
    A = (np.concatenate([j, j]), np.concatenate([j, m + j]), np.ones(2 * m))
    C = (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0))
    return H, A, C


def _kkt_rhs(n, m, H, A, x, y):
    """[H A^T; A 0] [x; y]"""
    r = np.zeros(n + m)
    hr, hc, hv = H[0] - 1, H[1] - 1, H[2]
    np.add.at(r, hr, hv * x[hc])
    off = hr != hc
    np.add.at(r, hc[off], hv[off] * x[hr[off]])
    ar, ac, av = A[0] - 1, A[1] - 1, A[2]
    np.add.at(r, ac, av * y[ar])
    np.add.at(r, n + ar, av * x[ac])
    return r


@pytest.mark.gpu
@pytest.mark.parametrize("n,m", [(2000, 400), (100000, 20000)])
def test_sbls_kkt_constructed_solution(n, m):
    refio = _need()
    H, A, C = _qp_kkt(n, m, 20240102)
    rng = np.random.default_rng(7)
    x, y = rng.uniform(-1, 1, n), rng.uniform(-1, 1, m)
    rhs = _kkt_rhs(n, m, H, A, x, y)
    r = refio.run_sbls(n, m, H, A, C, rhs, solver="gsls", factorization=2, repeat=2)
    assert (r["status_factorize"], r["status_solve"]) == (0, 0)
    assert r["negative_eigenvalues"] == m                       # SBLS's own inertia requirement
    err = np.abs(r["sol"] - np.concatenate([x, y])).max()
    assert err <= 1e-8, err


@pytest.mark.gpu
def test_sbls_refinement_runs_in_the_backend():
    """SBLS_solve_explicit's loop (sbls.f90:5147-5380: solve, update, residual over K on the host, itref_max times) is one
    SLS_solve with max_iterative_refinements = itref_max for 'gsls' (integration/patch_sbls.py).  Same recurrence:
    on a KKT system whose H spans eight decades the residual of the returned solution falls with itref_max exactly as
    refinement steps make it fall, and itref_max = 0 is the plain solve."""
    refio = _need()
    n, m = 20000, 6000
    H, A, C = _qp_kkt(n, m, 5)
    rng = np.random.default_rng(11)
    x, y = rng.uniform(-1, 1, n), rng.uniform(-1, 1, m)
    rhs = _kkt_rhs(n, m, H, A, x, y)
    scale = np.abs(rhs).max()
    res, err = [], []
    for it in (0, 1, 3):
        r = refio.run_sbls(n, m, H, A, C, rhs, solver="gsls", factorization=2, itref_max=it)
        assert (r["status_factorize"], r["status_solve"]) == (0, 0)
        sol = r["sol"]
        res.append(np.abs(_kkt_rhs(n, m, H, A, sol[:n], sol[n:]) - rhs).max() / scale)
        err.append(np.abs(sol - np.concatenate([x, y])).max())
    assert res[0] <= 1e-9 and res[1] <= 1e-14 and res[2] <= 1e-14, res
    assert res[1] < 0.1 * res[0] or res[0] <= 1e-15, res          # one refinement step gains at least a digit
    assert err[2] <= 1e-8, err


@pytest.mark.gpu
@pytest.mark.parametrize("repeat", [1, 2, 3, 4])
def test_sbls_values_assembled_on_the_device(repeat):
    """on a refactorization (same structure) the patched SBLS_form_n_factorize_explicit does not copy A%val,
    H%val, -C%val into K%val (sbls.f90:3349, 3404, 3967) -- the backend takes the three arrays where they are
    (gsls_set_value_part) and K's values are put together in HBM.  The driver CHANGES the caller's arrays between rounds
    (H, C every round; A on odd rounds only, new_a = 0 on even ones), so a stale or misplaced part shows in the solution
    of the last round; C /= 0 checks the sign.  get_norm_residual: the backend's b - K x (gsls_residual) instead of the
    host loops over K%val (sbls.f90:5343-5372), which is stale here."""
    refio = _need()
    n, m = 3000, 900
    H, A, _ = _qp_kkt(n, m, 31)
    j = np.arange(1, m + 1)
    C = (j.astype(np.int32), j.astype(np.int32), np.linspace(0.5, 2.0, m))      # K = [H A^T; A -C]
    rng = np.random.default_rng(13)
    rhs = rng.uniform(-1, 1, n + m)
    r = refio.run_sbls(n, m, H, A, C, rhs, solver="gsls", factorization=2, repeat=repeat, itref_max=1, drift=True,
                       get_norm_residual=True)
    assert (r["status_factorize"], r["status_solve"]) == (0, 0)
    k = repeat
    ka = k if k % 2 == 1 else k - 1                       # A last changed on the last odd round
    Hv = H[2] * (1.0 + 0.25 * (k - 1))
    Cv = C[2] * (1.0 + 0.5 * (k - 1))
    Av = A[2] * (1.0 - 0.1 * (ka - 1))
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    Hm = sp.coo_matrix((Hv, (H[0] - 1, H[1] - 1)), shape=(n, n)).tocsr()
    Hm = Hm + sp.tril(Hm, -1).T
    Am = sp.coo_matrix((Av, (A[0] - 1, A[1] - 1)), shape=(m, n)).tocsr()
    Cm = sp.diags(Cv)
    K = sp.bmat([[Hm, Am.T], [Am, -Cm]]).tocsc()
    ref = spl.spsolve(K, rhs)
    assert np.abs(r["sol"] - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    true_res = np.abs(rhs - K @ r["sol"]).max()
    assert r["norm_residual"] is not None and r["norm_residual"] <= 1e-10
    assert abs(r["norm_residual"] - true_res) <= 1e-11


@pytest.mark.gpu
def test_sbls_cfg3_full_size():
    """BASELINE.json configs[2]: SBLS KKT saddle point of a synthetic QP, n = 1e6, m = 2e5, through the
    real SBLS; three form_and_factorize + solve rounds (new values, same structure) as CQP would do."""
    refio = _need()
    n, m = 1000000, 200000
    H, A, C = _qp_kkt(n, m, 20240102)
    x, y = np.ones(n), np.ones(m)
    rhs = _kkt_rhs(n, m, H, A, x, y)
    r = refio.run_sbls(n, m, H, A, C, rhs, solver="gsls", factorization=2, repeat=3, timeout=1500)
    assert (r["status_factorize"], r["status_solve"]) == (0, 0)
    assert r["negative_eigenvalues"] == m
    assert np.abs(r["sol"] - 1.0).max() <= 1e-7
    print("cfg3 through SBLS: form_and_factorize median %.2f ms, solve median %.2f ms" % (
        1e3 * r["t_factorize_median"], 1e3 * r["t_solve_median"]))


@pytest.mark.gpu
@pytest.mark.parametrize("N", [5, 10, 100, 2000])
def test_cqp_qpband_through_sbls_and_gsls(N):
    """BASELINE.json configs[0]: the reference's interior-point QP solver CQP (src/cqp/cqp.f90, built where it lies with
    its 14 dependencies above the patched facade) on examples/QPBAND.SIF, every KKT system going
    CQP -> SBLS_form_and_factorize / SBLS_solve -> SLS('gsls') (cqp.f90:4781-4896).  Checked: CQP's own exit status and
    residuals, the KKT conditions recomputed here from the returned (x, y, z), and -- small N -- the optimal value
    against an independent solve (scipy SLSQP).  The comparison run on a reference solver is not available: ssids needs
    METIS or MC68 (stubs in the reference tree; CQP passes no PERM) and the dense 'sytr' arm fails inside the reference's
    own SBLS_form_and_factorize with -9 on this problem (seen with the unpatched control flow too)."""
    from oracle import refio
    if not refio.cqp_available():
        pytest.skip("oracle/_ref/cqp_gsls_driver not built")
    n, m, H, A, g, c_l, c_u, x_l, x_u = P.qpband(N)
    r = refio.run_cqp(n, m, H, A, g, c_l, c_u, x_l, x_u, solver="gsls")
    assert r["status"] == 0 and r["iter"] <= 30, r
    assert r["primal_infeasibility"] <= 1e-8 and r["dual_infeasibility"] <= 1e-5 and r["complementary_slackness"] <= 1e-5
    x, y, z = r["x"], r["y"], r["z"]
    Hx = P.sym_matvec(n, H[0] - 1, H[1] - 1, H[2], x)
    Ax = np.zeros(m)
    np.add.at(Ax, A[0] - 1, A[2] * x[A[1] - 1])
    ATy = np.zeros(n)
    np.add.at(ATy, A[1] - 1, A[2] * y[A[0] - 1])
    assert np.abs(Hx + g - ATy - z).max() <= 2e-5                       # stationarity (CQP stops at ~7e-6)
    assert (Ax >= c_l - 1e-8).all() and (x >= x_l - 1e-8).all() and (x <= x_u + 1e-8).all()
    assert (y >= -1e-8).all() and np.abs(y * (Ax - c_l)).max() <= 1e-4  # multipliers of >= constraints, complementarity
    assert abs(r["obj"] - (0.5 * x @ Hx + g @ x)) <= 1e-8 * max(1.0, abs(r["obj"]))
    if N <= 100:
        from scipy.optimize import minimize, LinearConstraint, Bounds
        import scipy.sparse as sp
        Hm = sp.coo_matrix((H[2], (H[0] - 1, H[1] - 1)), shape=(n, n)).toarray()
        Hm = Hm + Hm.T - np.diag(np.diag(Hm))
        Am = sp.coo_matrix((A[2], (A[0] - 1, A[1] - 1)), shape=(m, n)).toarray()
        res = minimize(lambda v: 0.5 * v @ Hm @ v + g @ v, np.ones(n), jac=lambda v: Hm @ v + g, method="SLSQP",
                       bounds=Bounds(x_l, x_u), constraints=[LinearConstraint(Am, c_l, np.inf)],
                       options={"ftol": 1e-14, "maxiter": 1000})
        assert res.success
        assert abs(res.fun - r["obj"]) <= 1e-5 * max(1.0, abs(res.fun))
