"""GPU (-m gpu): parity of the HIP path, called through the C ABI, against
  (1) the golden vectors of the real reference (tests/golden/*.npz),
  (2) the plain-C oracle on the same seeded inputs at sizes it finishes in seconds,
  (3) the reference's own known-answer systems and test sweeps (src/sls/slst.f90),
  (4) size-independent properties at BASELINE.json's full sizes.
Tolerances: integer/index results bit-exact; scaled residual
||b-Ax||_inf/(||A||_inf ||x||_inf+||b||_inf) <= 1e-13 for SPD, <= 1e-10 indefinite without
refinement and <= 1e-14 with one refinement step; forward error vs reference <= 1e-9 * ||x||."""
import glob
import os
import subprocess

import numpy as np
import pytest

import problems as P

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = sorted(p for p in glob.glob(os.path.join(HERE, "golden", "*.npz")) if not os.path.basename(p).startswith("scaled_"))
EPS = np.finfo(float).eps


@pytest.fixture(scope="session", autouse=True)
def need_gpu():
    from galahad_amd._lib import lib
    assert lib.gsls_device_count() > 0, "no HIP device: the -m gpu tests need an MI355X"
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)


def run_gsls(prob, posdef, perm=None, nemin=32, refine=0, ordering_free=False, storage="COORDINATE"):
    from galahad_amd import SLS, SMT, Control, InformSLS
    n, row, col, val, rhs, xs = prob
    m = SMT(n, storage, row=row, col=col, val=val)
    s, c, i = SLS(), Control(), InformSLS()
    s.initialize("gsls", c, i)
    c.pivot_control = 2 if posdef else 1
    c.node_amalgamation = nemin
    c.max_iterative_refinements = refine
    if perm is None and not ordering_free:
        perm = np.arange(1, n + 1)         # natural order = identity PERM (ordering <= 0 means own ND)
    s.analyse(m, c, i, PERM=perm)
    assert i.status == 0, i.status
    s.factorize(m, c, i)
    return s, m, c, i


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_against_reference_golden(path):
    g = np.load(path)
    n = int(g["n"])
    prob = (n, g["row"], g["col"], g["val"], g["rhs"], g["xstar"])
    posdef = bool(g["posdef"])
    s, m, c, i = run_gsls(prob, posdef, perm=g["perm"], nemin=int(g["nemin"]))
    assert i.status == 0, (i.status, i.gsls_inform)
    if i.delayed_pivots == 0:
        # structure-dependent statistics are bit-exact unless pivots were delayed (then the pivot
        # sequence, hence the fill, legitimately differs: compare inertia + residual only)
        assert i.entries_in_factors == int(g["ref_num_factor"])
        assert i.flops_elimination == int(g["ref_num_flops"])
    assert i.rank == int(g["ref_rank"])
    assert i.negative_eigenvalues == int(g["ref_neg"])        # inertia exact
    x = s.solve(m, g["rhs"], c, i)
    assert i.status == 0
    ref = g["ref_x"]
    assert np.abs(x - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    rhs, X = np.atleast_2d(g["rhs"].T).T, np.atleast_2d(x.T).T
    tol = 1e-13 if posdef else 1e-10
    for k in range(X.shape[1]):
        assert P.scaled_residual(n, g["row"], g["col"], g["val"], X[:, k], rhs[:, k]) <= tol
    s.terminate()


@pytest.mark.parametrize("kat,posdef", [(P.kat_indefinite, False), (P.kat_definite, True),
                                         (P.kat_definite, False)])
@pytest.mark.parametrize("ordering", ["default", "provided"])
@pytest.mark.parametrize("refine", [0, 1])
def test_known_answer_sweep(kat, posdef, ordering, refine):
    """src/sls/slst.f90:101-330: orderings x {no refine, +1 refine}; x = 1..5 to sqrt(eps)."""
    prob = kat()
    n = prob[0]
    perm = np.arange(n, 0, -1) if ordering == "provided" else None      # slst.f90:54-56
    s, m, c, i = run_gsls(prob, posdef, perm=perm, refine=refine, ordering_free=(perm is None))
    assert i.status == 0
    x = s.solve(m, prob[4], c, i)
    assert np.abs(x - prob[5]).max() <= np.sqrt(EPS)
    X2 = s.solve(m, np.column_stack([prob[4], 2 * prob[4]]), c, i)      # 2 rhs, slst.f90:276-300
    assert np.abs(X2[:, 0] - prob[5]).max() <= np.sqrt(EPS)
    assert np.abs(X2[:, 1] - 2 * prob[5]).max() <= np.sqrt(EPS)
    s.terminate()


def test_storage_types():
    n, row, col, val, rhs, xs = P.kat_indefinite()
    ptr = np.array([1, 2, 4, 6, 7, 8])
    dense = np.array([2.0, 3.0, 0.0, 0.0, 4.0, 1.0, 0.0, 0.0, 5.0, 0.0, 0.0, 6.0, 0.0, 0.0, 1.0])
    from galahad_amd import SLS, SMT, Control, InformSLS
    for m in (SMT(n, "SPARSE_BY_ROWS", ptr=ptr, col=col, val=val), SMT(n, "DENSE", val=dense)):
        s, c, i = SLS(), Control(), InformSLS()
        s.initialize("gsls", c, i)
        s.analyse(m, c, i)
        s.factorize(m, c, i)
        assert i.status == 0
        assert np.abs(s.solve(m, rhs, c, i) - xs).max() <= np.sqrt(EPS)
        s.terminate()


def test_part_solves_compose():
    """SLS_part_solve L, D, U (src/sls/slst.f90:302-330, tolerance eps^(1/3)): the reference's ssids
    arm returns 'unavailable' (sls.f90:6886-6888); gsls implements them, so check L*D*U == full."""
    prob = P.kkt_qpband(400, 80)
    s, m, c, i = run_gsls(prob, False)
    assert i.status == 0
    full = s.solve(m, prob[4], c, i)
    y = s.part_solve("L", prob[4], c, i)
    y = s.part_solve("D", y, c, i)
    y = s.part_solve("U", y, c, i)
    assert i.status == 0
    assert np.abs(y - full).max() <= EPS ** (1.0 / 3.0) * max(1.0, np.abs(full).max())
    assert np.abs(y - full).max() <= 1e-12 * max(1.0, np.abs(full).max())
    s.terminate()


def test_enquire_and_alter_d():
    """SLS_enquire / SLS_alter_d (sls.f90:6175-6547): D is returned inverted; doubling D^-1 halves...
    i.e. scaling d by 2 doubles the solution of L D L^T x = b."""
    prob = P.grid2d(12, 12, shift=1.0)
    s, m, c, i = run_gsls(prob, False)
    x = s.solve(m, prob[4], c, i)
    out = s.enquire(i, want_perm=True, want_d=True)
    assert i.status == 0
    assert sorted(np.abs(out["PIVOTS"])) == list(range(1, prob[0] + 1))
    d = out["D"]
    piv = out["PIVOTS"]
    order = np.argsort(np.abs(piv))            # variables in pivot sequence
    neg, k = 0, 0
    while k < prob[0]:
        if piv[order[k]] > 0:                   # 1x1
            neg += d[0, k] < 0
            k += 1
        else:                                   # 2x2: inertia from det / trace of the inverse block
            det = d[0, k] * d[0, k + 1] - d[1, k] ** 2
            neg += 1 if det < 0 else (2 if d[0, k] + d[0, k + 1] < 0 else 0)
            k += 2
    assert neg == i.negative_eigenvalues
    s.alter_d(2.0 * d, i)
    assert i.status == 0
    x2 = s.solve(m, prob[4], c, i)
    assert np.abs(x2 - 2 * x).max() <= 1e-12 * np.abs(x).max()
    s.terminate()


def test_not_positive_definite_and_singular_status():
    from galahad_amd import sls as S
    s, m, c, i = run_gsls(P.kat_indefinite(), True)
    assert i.gsls_inform["flag"] == -6             # SSIDS_ERROR_NOT_POS_DEF
    assert i.status == S.GALAHAD_error_inertia     # what TRS expects (trs.f90:1957); the ssids arm's
    #                                                quirk (-6 -> -3, sls.f90:1768) is NOT copied
    s.terminate()
    n = 6
    idx = np.arange(1, n + 1, dtype=np.int32)
    val = np.array([1.0, 2.0, 0.0, 3.0, 0.0, 4.0])
    s, m, c, i = run_gsls((n, idx, idx, val, val, val), False)
    assert i.status == 0 and i.gsls_inform["flag"] == 7      # SSIDS_WARNING_FACT_SINGULAR
    assert i.rank == 4
    s.terminate()


@pytest.mark.parametrize("name,prob,posdef", [
    ("band_20000_63", lambda: P.banded_spd(20000, 63), True),
    ("grid2d_90", lambda: P.grid2d(90, 90), True),
    ("grid2d_80_indef", lambda: P.grid2d(80, 80, shift=1.0), False),
    ("grid3d_16", lambda: P.grid3d(16, 16, 16), True),
    ("kkt_6000_1200", lambda: P.kkt_qpband(6000, 1200), False),
    ("rand_spd_5000", lambda: P.random_sparse(5000, 4, 21, spd=True), True),
    ("grid3d_27pt_cfg5_16_chol", lambda: P.grid3d_27pt_perturbed(16, 16, 16), True),
    ("grid3d_27pt_cfg5_16_ldlt", lambda: P.grid3d_27pt_perturbed(16, 16, 16), False),
])
def test_against_c_oracle(name, prob, posdef):
    """same seeded input through the HIP path and through oracle/gsls_oracle.c, same PERM."""
    from oracle.oracle import Oracle, lower_csc
    prob = prob()
    n, row, col, val, rhs, xs = prob
    rng = np.random.default_rng(5)
    perm = None if name.startswith("band") or name.startswith("kkt") else rng.permutation(n) + 1
    if name.startswith("grid2d") or name.startswith("grid3d"):
        perm = None       # natural order keeps the fill (and the oracle's run time) small
    s, m, c, i = run_gsls(prob, posdef, perm=perm)
    assert i.status == 0, i.gsls_inform
    x = s.solve(m, rhs, c, i)
    ptr, r, v = lower_csc(n, row, col, val)
    o = Oracle(n, ptr, r, np.arange(1, n + 1) if perm is None else perm)
    assert o.factor(v, posdef, small=EPS) in (0,)
    xo = o.solve(rhs)
    st = o.stats()
    if i.delayed_pivots == 0:
        assert i.entries_in_factors == st["num_factor"] and i.flops_elimination == st["num_flops"]
    assert i.negative_eigenvalues == st["num_neg"]
    assert np.abs(x - xo).max() <= 1e-9 * max(1.0, np.abs(xo).max())
    assert P.scaled_residual(n, row, col, val, x, rhs) <= (1e-13 if posdef else 1e-10)
    o.close()
    s.terminate()


def test_refinement_reaches_1e14_on_indefinite():
    prob = P.kkt_qpband(20000, 4000)
    s, m, c, i = run_gsls(prob, False, refine=1)
    assert i.status == 0
    assert i.negative_eigenvalues == 4000 and i.rank == 24000     # inertia (n, m, 0)
    x = s.solve(m, prob[4], c, i)
    assert P.scaled_residual(prob[0], prob[1], prob[2], prob[3], x, prob[4]) <= 1e-14
    s.terminate()


def test_delayed_pivots_are_repaired_and_remembered():
    """Failed pivots (the reference delays them to the parent front, assemble.hxx:244-264) become a
    repaired elimination order kept in the handle: inertia exact on the first factorization, and the
    second factorization of the same structure needs no further repair.  The order is GIVEN here
    (constraints first: every constraint meets a zero pivot), so the backend may not pre-order."""
    prob = P.kkt_qpband(3000, 600)
    n = prob[0]
    s, m, c, i = run_gsls(prob, False, perm=np.arange(n, 0, -1))
    assert i.status == 0, i.gsls_inform
    assert i.negative_eigenvalues == 600 and i.rank == 3600
    moved_first = i.delayed_pivots
    assert moved_first > 0                        # this order does need delays
    x = s.solve(m, prob[4], c, i)
    assert P.scaled_residual(prob[0], prob[1], prob[2], prob[3], x, prob[4]) <= 1e-10
    s.factorize(m, c, i)
    assert i.status == 0 and i.delayed_pivots == 0 and i.negative_eigenvalues == 600
    x2 = s.solve(m, prob[4], c, i)
    assert P.scaled_residual(prob[0], prob[1], prob[2], prob[3], x2, prob[4]) <= 1e-10
    s.terminate()


def test_saddle_point_order_is_value_independent():
    """With its own ordering the backend puts every zero-diagonal variable (constraint row of a KKT
    matrix) after all of its neighbours when the values first arrive: no delayed pivot on the first
    factorization, none when the Hessian's diagonal drifts over four decades (interior-point barrier
    terms), inertia (n, m, 0) throughout."""
    prob = P.kkt_qpband(30000, 6000)
    n, row, col, val, rhs, xs = prob
    s, m, c, i = run_gsls(prob, False, ordering_free=True)
    assert i.status == 0, i.gsls_inform
    assert i.delayed_pivots == 0 and i.negative_eigenvalues == 6000 and i.rank == 36000
    rng = np.random.default_rng(4)
    hd = np.where((row == col) & (row <= 30000))[0]
    for it in range(4):
        m.val[hd] = val[hd] * 10.0 ** rng.uniform(-2, 2, len(hd)) + 2.0      # stays diagonally dominant: H > 0
        s.factorize(m, c, i)
        assert i.status == 0 and i.delayed_pivots == 0 and i.negative_eigenvalues == 6000, (it, i.gsls_inform)
        b = P.sym_matvec(n, row - 1, col - 1, m.val, xs)
        x = s.solve(m, b, c, i)
        assert P.scaled_residual(n, row, col, m.val, x, b) <= 1e-10
    s.terminate()


def test_cfg3_kkt_full_size_properties():
    """BASELINE.json configs[2] shape: KKT saddle point n=1e6, m=2e5 -- inertia (n, m, 0), residual
    with one refinement step, round trip solve(K z) == z."""
    prob = P.kkt_qpband(1000000, 200000)
    n, row, col, val, rhs, xs = prob
    s, m, c, i = run_gsls(prob, False, ordering_free=True, refine=1)
    assert i.status == 0, i.gsls_inform
    assert i.negative_eigenvalues == 200000 and i.rank == 1200000
    x = s.solve(m, rhs, c, i)
    assert P.scaled_residual(n, row, col, val, x, rhs) <= 1e-14
    z = np.random.default_rng(2).uniform(-1, 1, n)
    back = s.solve(m, P.sym_matvec(n, row - 1, col - 1, val, z), c, i)
    assert np.abs(back - z).max() <= 1e-8
    s.terminate()


def test_cfg4_grid_indefinite_and_shifted_posdef():
    """BASELINE.json configs[3] shape: 5-point Laplacian on 707x707 (n=499849).  H - I is indefinite
    (pivoted LDL^T, refined residual); H + lambda I with pivot_control=2 is what TRS factorizes
    repeatedly (src/trs/trs.f90:1942-1964): Cholesky, and 'not positive definite' is an expected
    signal for lambda too small."""
    from galahad_amd import sls as S
    prob = P.grid2d(707, 707, shift=1.0)
    n, row, col, val, rhs, xs = prob
    s, m, c, i = run_gsls(prob, False, ordering_free=True, refine=1)
    assert i.status == 0, i.gsls_inform
    x = s.solve(m, rhs, c, i)
    assert P.scaled_residual(n, row, col, val, x, rhs) <= 1e-13
    s.terminate()
    s, m, c, i = run_gsls(prob, True, ordering_free=True)            # H - I is not PD
    assert i.gsls_inform["flag"] == -6 and i.status == S.GALAHAD_error_inertia
    s.terminate()
    prob = P.grid2d(707, 707, shift=-0.5)                             # H + 0.5 I
    s, m, c, i = run_gsls(prob, True, ordering_free=True)
    assert i.status == 0
    x = s.solve(m, prob[4], c, i)
    assert P.scaled_residual(n, prob[1], prob[2], prob[3], x, prob[4]) <= 1e-13
    s.terminate()


def test_big_fronts_blocked_solve_path():
    """3-D grids give fronts of thousands of columns: the blocked multi-launch solve path (fronts wider
    than 256 columns) against the small-front path's answer on the same system, SPD and indefinite,
    full and partial solves, two right-hand sides."""
    prob = P.grid3d(30, 30, 30)
    n, row, col, val, rhs, xs = prob
    for posdef in (True, False):
        s, m, c, i = run_gsls(prob, posdef, ordering_free=True)
        assert i.status == 0 and i.max_front_size > 256      # exercises the big-front kernels
        X = s.solve(m, np.column_stack([rhs, -2.0 * rhs]), c, i)
        assert P.scaled_residual(n, row, col, val, X[:, 0], rhs) <= 1e-13
        assert np.abs(X[:, 1] + 2.0 * X[:, 0]).max() <= 1e-12
        y = s.part_solve("L", rhs, c, i)
        if not posdef:
            y = s.part_solve("D", y, c, i)
        y = s.part_solve("U", y, c, i)
        assert np.abs(y - X[:, 0]).max() <= 1e-12
        s.terminate()


def test_cfg5_shape_3d_grid_2M():
    """BASELINE.json configs[4] shape on ONE GPU: 3-D stencil on a 126^3 grid, n = 2 000 376
    (nnz(L) ~ 1.8e9, ~12.5 TFlop): residual bar and solve(A z) == z round trip."""
    prob = P.grid3d(126, 126, 126)
    n, row, col, val, rhs, xs = prob
    s, m, c, i = run_gsls(prob, True, ordering_free=True)
    assert i.status == 0, i.gsls_inform
    x = s.solve(m, rhs, c, i)
    assert P.scaled_residual(n, row, col, val, x, rhs) <= 1e-13
    z = np.random.default_rng(3).uniform(-1, 1, n)
    back = s.solve(m, P.sym_matvec(n, row - 1, col - 1, val, z), c, i)
    assert np.abs(back - z).max() <= 1e-10
    s.terminate()


def test_free_ordering_matches_natural_solution():
    prob = P.grid2d(60, 60)
    s1, m, c1, i1 = run_gsls(prob, True)
    s2, _, c2, i2 = run_gsls(prob, True, ordering_free=True)
    x1, x2 = s1.solve(m, prob[4], c1, i1), s2.solve(m, prob[4], c2, i2)
    assert np.abs(x1 - x2).max() <= 1e-10
    s1.terminate()
    s2.terminate()


def test_full_size_cfg2_properties():
    """BASELINE.json configs[1]: banded SPD n=1e5, semi-bandwidth 127 -- size-independent checks:
    residual bar, x* recovery, solve(A*z) == z round trip, repeat factorization is bit-identical."""
    prob = P.banded_spd(100000, 127)
    n, row, col, val, rhs, xs = prob
    s, m, c, i = run_gsls(prob, True)
    assert i.status == 0
    assert i.entries_in_factors == 14339888 and i.flops_elimination == 2065810544
    x = s.solve(m, rhs, c, i)
    assert P.scaled_residual(n, row, col, val, x, rhs) <= 1e-13
    assert np.abs(x - xs).max() <= 1e-10
    z = np.random.default_rng(1).uniform(-1, 1, n)
    back = s.solve(m, P.sym_matvec(n, row - 1, col - 1, val, z), c, i)
    assert np.abs(back - z).max() <= 1e-11
    s.factorize(m, c, i)
    assert np.array_equal(s.solve(m, rhs, c, i), x)      # deterministic: no atomics in the path
    s.terminate()


@pytest.mark.parametrize("nemin", [64, 24, 8])
@pytest.mark.parametrize("case", ["kkt", "grid_indef", "rand_indef", "kkt_perm_reversed"])
def test_refactorization_after_learning(case, nemin):
    """The handle learns from its first pivoted factorization (order repair, in-block pivot sequence,
    2x2 positions) and later factorizations run the optimistic LDL^T kernels (workgroup per 64-column block;
    with small node_amalgamation the wave-per-front kernel for tiny fronts and its blacklist) with the
    complete-pivoting kernel as fallback.  Refactorizing the same values, and then values that drift as in an
    interior-point loop, must keep the inertia and the residual; against the oracle on the same matrix."""
    from galahad_amd import SLS, SMT, Control, InformSLS
    rng = np.random.default_rng(11)
    perm = None
    if case == "kkt":
        prob = P.kkt_qpband(6000, 1200, seed=5)
    elif case == "kkt_perm_reversed":
        prob = P.kkt_qpband(500, 140, seed=6)
        perm = np.arange(prob[0], 0, -1)          # constraints first: zero pivots everywhere
    elif case == "grid_indef":
        prob = P.grid2d(50, 40, shift=1.0)
    else:
        prob = P.random_sparse(3000, 5, seed=9, spd=False)
    n, row, col, val, rhs, xs = prob
    m = SMT(n, "COORDINATE", row=row, col=col, val=val.copy())
    s, c, i = SLS(), Control(), InformSLS()
    s.initialize("gsls", c, i)
    c.pivot_control = 1
    c.node_amalgamation = nemin
    s.analyse(m, c, i, PERM=perm)
    assert i.status == 0
    diag = np.where(row == col)[0]
    neg_ref = None
    for it in range(6):
        if it >= 3:   # drift the diagonal like barrier terms do (keeps the zero block of a KKT matrix zero)
            nzd = diag[np.abs(m.val[diag]) > 0]
            m.val[nzd] *= 10.0 ** rng.uniform(-0.3, 0.3, len(nzd))
        s.factorize(m, c, i)
        assert i.status == 0, (it, i.gsls_inform)
        b = P.sym_matvec(n, row - 1, col - 1, m.val, xs)
        x = s.solve(m, b, c, i)
        assert P.scaled_residual(n, row, col, m.val, x, b) <= 1e-10, it
        if it < 3:
            if neg_ref is None:
                neg_ref = i.negative_eigenvalues
                dense = np.zeros((n, n)) if n <= 700 else None
                if dense is not None:
                    dense[row - 1, col - 1] = m.val
                    dense = dense + np.tril(dense, -1).T
                    assert neg_ref == int((np.linalg.eigvalsh(dense) < 0).sum())
            assert i.negative_eigenvalues == neg_ref      # same matrix: same inertia, whichever kernel ran
    s.terminate()


@pytest.mark.parametrize("posdef", [True, False])
def test_factor_with_scaling_vector(posdef):
    """ssids_factor's optional `scale` (src/ssids/ssids.f90:770-779, fkeep.F90:229-318): the backend
    factorizes S A S and the solve scales on the way in and out, so x solves the ORIGINAL system."""
    import ctypes as C
    from galahad_amd._lib import Inform, lib
    from galahad_amd import SLS, SMT, Control, InformSLS
    prob = P.random_sparse(1500, 6, seed=21, spd=posdef)
    n, row, col, val, rhs, xs = prob
    m = SMT(n, "COORDINATE", row=row, col=col, val=val)
    s, c, i = SLS(), Control(), InformSLS()
    s.initialize("gsls", c, i)
    c.pivot_control = 2 if posdef else 1
    s.analyse(m, c, i)
    assert i.status == 0
    s._copy_control(c)
    VAL = s.scatter_values(m)
    scale = 10.0 ** np.random.default_rng(3).uniform(-2, 2, n)
    ginf = Inform()
    for rep in range(3):      # the refactorizations take the learned path
        f = lib.gsls_factor(s.handle, 1 if posdef else 0, VAL.ctypes.data_as(C.c_void_p),
                            scale.ctypes.data_as(C.c_void_p), C.byref(s.opts), C.byref(ginf))
        assert f >= 0, (f, ginf.as_dict())
        x = np.asfortranarray(rhs.copy())
        f = lib.gsls_solve(s.handle, 0, 1, x.ctypes.data_as(C.c_void_p), n, C.byref(s.opts), C.byref(ginf))
        assert f >= 0
        assert P.scaled_residual(n, row, col, val, x, rhs) <= 1e-10
    s.terminate()


@pytest.mark.parametrize("posdef", [True, False])
def test_factor_coo_with_a_host_scaling_vector_equals_factor(posdef):
    """gsls_factor_coo's optional `scale` (the Fortran binding exposes it): the caller's HOST vector must survive the
    re-upload of the device arrays that the first factorization of a pattern and every order repair trigger (ADVICE
    r2: it was copied back from freed device memory).  Same bits as gsls_factor with the same scale."""
    import ctypes as C
    from galahad_amd._lib import Inform, lib
    from galahad_amd import SLS, SMT, Control, InformSLS
    prob = P.random_sparse(1200, 5, seed=33, spd=posdef)
    n, row, col, val, rhs, xs = prob
    scale = 10.0 ** np.random.default_rng(5).uniform(-2, 2, n)
    sols = []
    for coo in (False, True):
        m = SMT(n, "COORDINATE", row=row, col=col, val=val)
        s, c, i = SLS(), Control(), InformSLS()
        s.initialize("gsls", c, i)
        c.pivot_control = 2 if posdef else 1
        s.analyse(m, c, i)
        assert i.status == 0
        s._copy_control(c)
        ginf = Inform()
        for rep in range(2):
            if coo:
                f = lib.gsls_factor_coo(s.handle, 1 if posdef else 0, np.ascontiguousarray(val).ctypes.data_as(C.c_void_p),
                                        scale.ctypes.data_as(C.c_void_p), C.byref(s.opts), C.byref(ginf))
            else:
                VAL = s.scatter_values(m)
                f = lib.gsls_factor(s.handle, 1 if posdef else 0, VAL.ctypes.data_as(C.c_void_p),
                                    scale.ctypes.data_as(C.c_void_p), C.byref(s.opts), C.byref(ginf))
            assert f >= 0, (coo, f, ginf.as_dict())
            x = np.asfortranarray(rhs.copy())
            f = lib.gsls_solve(s.handle, 0, 1, x.ctypes.data_as(C.c_void_p), n, C.byref(s.opts), C.byref(ginf))
            assert f >= 0
            assert P.scaled_residual(n, row, col, val, x, rhs) <= 1e-10
        sols.append(x)
        s.terminate()
    assert np.array_equal(sols[0], sols[1])


@pytest.mark.parametrize("nb", [500, 700, 2500])
def test_all_zero_diagonal_saddle_needs_2x2_everywhere(nb):
    """K = [0 B; B^T 0]: no variable has a pivot of its own, every elimination is a 2x2 pivot.  The backend
    pairs each variable with its strongest neighbour when the values arrive (partners share a supernode) and, where
    the 64-column blocked kernels still cannot place a pivot, falls back to threshold partial pivoting over the
    whole front (k_front_tpp) and to delays towards the root (gsls_api.cpp: plan_repair); inertia (k, k, 0).
    nb = 700 is the instance that ended with flag -98 in round 1 (order-repair cycle); nb = 2500 is the
    5000-variable case."""
    from galahad_amd import SLS, SMT, Control, InformSLS
    rng = np.random.default_rng(1)
    n = 2 * nb
    r, c, v = [], [], []
    for i in range(nb):
        for j in {i} | set(rng.integers(0, nb, 3).tolist()):
            r.append(nb + j)
            c.append(i)
            v.append(2.0 + rng.uniform(0, 1) if j == i else rng.uniform(-0.3, 0.3))
    row, col, val = np.array(r, dtype=np.int32) + 1, np.array(c, dtype=np.int32) + 1, np.array(v)
    xs = rng.uniform(-1, 1, n)
    rhs = P.sym_matvec(n, row - 1, col - 1, val, xs)
    m = SMT(n, "COORDINATE", row=row, col=col, val=val)
    s, ct, i = SLS(), Control(), InformSLS()
    s.initialize("gsls", ct, i)
    ct.pivot_control = 1
    s.analyse(m, ct, i)
    for rep in range(3):
        s.factorize(m, ct, i)
        assert i.status == 0, i.gsls_inform
        assert i.negative_eigenvalues == nb and i.rank == n and i.two_by_two_pivots > 0
        x = s.solve(m, rhs, ct, i)
        assert P.scaled_residual(n, row, col, val, x, rhs) <= 1e-12
    assert i.delayed_pivots == 0          # the repaired order (and the fronts that need whole-front pivoting) are remembered
    s.terminate()


# ---- part solves, enquire, alter against the ORACLE (not against themselves) -----------------------------------
def _oracle_for(prob, perm, nemin=32):
    from oracle.oracle import Oracle, lower_csc
    n, row, col, val, rhs, xs = prob
    ptr, r, v = lower_csc(n, row, col, val)
    o = Oracle(n, ptr, r, np.asarray(perm, dtype=np.int32), nemin=nemin)
    return o, v


def test_part_solves_and_enquire_match_dense_ldlt_when_pivots_are_unique():
    """An SPD matrix through the pivoted LDL^T kernels with a given PERM: no pivoting happens, so L and D are THE
    LDL^T factors of P A P^T (unique), computed here densely with numpy.  Every partial solve (jobs 1..4 of
    ssids_solve, fkeep.F90:229-318; SLS_part_solve L/D/U/S, sls.f90:6551-7220) and the enquired (piv_order, D^-1)
    (ssids_enquire_indef, fkeep.F90:321-377) must agree with them; the full solve with the C oracle's.
    (The C oracle cannot serve for the parts: its TPP takes 2x2 pivots even on SPD matrices.)"""
    prob = P.grid2d(17, 13)
    n, row, col, val = prob[0], prob[1], prob[2], prob[3]
    rng = np.random.default_rng(3)
    perm = (rng.permutation(n) + 1).astype(np.int32)
    s, m, c, i = run_gsls(prob, False, perm=perm)
    assert i.status == 0 and i.negative_eigenvalues == 0 and i.two_by_two_pivots == 0
    A = np.zeros((n, n))
    A[row - 1, col - 1] = val
    A = A + np.tril(A, -1).T
    pos = s.ORDER.astype(np.int64) - 1           # variable -> pivot position (a postorder-equivalent of perm)
    Ap = np.zeros_like(A)
    Ap[np.ix_(pos, pos)] = A
    Cf = np.linalg.cholesky(Ap)
    d = np.diag(Cf) ** 2
    Lf = Cf / np.diag(Cf)[None, :]
    b = rng.uniform(-1, 1, n)
    bp = np.zeros(n)
    bp[pos] = b

    def back(v):                                 # pivot order -> variable order
        return v[pos]
    tol = 1e-12
    got = s.part_solve("L", b, c, i)
    assert np.abs(got - back(np.linalg.solve(Lf, bp))).max() <= tol * np.abs(got).max()
    got = s.part_solve("D", b, c, i)
    assert np.abs(got - back(bp / d)).max() <= tol * np.abs(got).max()
    got = s.part_solve("U", b, c, i)
    assert np.abs(got - back(np.linalg.solve(Lf.T, bp))).max() <= tol * np.abs(got).max()
    got = s._backend_solve(np.array(b), 4, i)    # D (PL)^T x = b
    assert np.abs(got - back(np.linalg.solve(Lf.T, bp / d))).max() <= tol * np.abs(got).max()
    got = s.part_solve("S", b, c, i)             # L sqrt(D)
    assert i.status == 0
    assert np.abs(got - back(np.linalg.solve(Lf, bp) / np.sqrt(d))).max() <= tol * np.abs(got).max()
    out = s.enquire(i, want_perm=True, want_pivots=True, want_d=True)
    assert np.array_equal(out["PIVOTS"], s.ORDER)                     # no pivoting, all 1x1
    assert np.abs(out["D"][0] - 1.0 / d).max() <= 1e-13 * (1.0 / d).max() and not out["D"][1].any()
    o, v = _oracle_for(prob, s.ORDER)
    assert o.factor(v, False) == 0
    x = s.solve(m, b, c, i)
    assert np.abs(x - o.solve(b)).max() <= 1e-12 * np.abs(x).max()
    s.alter_d(2.0 * out["D"], i)                 # D^-1 doubled: the solution doubles
    x2 = s.solve(m, b, c, i)
    assert np.abs(x2 - 2.0 * x).max() <= 1e-12 * np.abs(x2).max()
    s.terminate()
    o.close()


@pytest.mark.parametrize("case", ["kkt", "rand_indef", "grid_indef"])
def test_part_solves_compose_to_the_oracle_solution_indefinite(case):
    """Indefinite systems: the backends choose different pivots (ours: given order first, the oracle's TPP tries 2x2
    first), so L, D and U differ part by part -- what must agree is everything that does not depend on that
    choice: the composition U^-1 D^-1 L^-1 b = the oracle's full solve, job 4 after job 1 likewise, the inertia
    (Sylvester) of the enquired D, and 'S' failing with GALAHAD_error_inertia on an indefinite D."""
    from galahad_amd import sls as S
    prob = {"kkt": P.kkt_qpband(600, 120), "rand_indef": P.random_sparse(900, 5, seed=5, spd=False),
            "grid_indef": P.grid2d(21, 19, shift=1.0)}[case]
    n = prob[0]
    s, m, c, i = run_gsls(prob, False, ordering_free=True)
    assert i.status == 0
    order = s.ORDER.copy()                       # the order the factors are in: the oracle gets the same
    o, v = _oracle_for(prob, order)
    assert o.factor(v, False) == 0
    b = prob[4]
    want = o.solve(b)
    y = s.part_solve("L", b, c, i)
    y = s.part_solve("D", y, c, i)
    y = s.part_solve("U", y, c, i)
    assert np.abs(y - want).max() <= 1e-9 * max(1.0, np.abs(want).max())
    y = s._backend_solve(s.part_solve("L", b, c, i), 4, i)
    assert np.abs(y - want).max() <= 1e-9 * max(1.0, np.abs(want).max())
    assert i.negative_eigenvalues == o.stats()["num_neg"]
    piv_o, d_o = o.enquire_indef()

    def inertia(piv, d):
        seq = np.argsort(np.abs(piv))
        neg, k = 0, 0
        while k < n:
            if piv[seq[k]] > 0:
                neg += d[0, k] < 0
                k += 1
            else:
                det = d[0, k] * d[0, k + 1] - d[1, k] ** 2
                neg += 1 if det < 0 else (2 if d[0, k] + d[0, k + 1] < 0 else 0)
                k += 2
        return neg
    out = s.enquire(i, want_perm=True, want_pivots=True, want_d=True)
    assert inertia(out["PIVOTS"], out["D"]) == inertia(piv_o, d_o) == i.negative_eigenvalues
    # PERM is the order the factors are in: |PIVOTS| is that order up to the pivoting inside the fronts
    assert sorted(np.abs(out["PIVOTS"])) == list(range(1, n + 1))
    assert np.array_equal(out["PERM"], s.ORDER) and sorted(out["PERM"]) == list(range(1, n + 1))
    moved = np.abs(np.abs(out["PIVOTS"]).astype(np.int64) - out["PERM"].astype(np.int64)).max()
    assert moved < 4096          # a pivot stays inside its front (blocks that pivot at run time move it by < one front)
    s.part_solve("S", b, c, i)
    assert i.status == S.GALAHAD_error_inertia
    s.terminate()
    o.close()


def test_value_map_and_residual_on_device():
    """gsls_set_coo / gsls_factor_coo / gsls_residual against the host loops they replace
    (SLS_factorize's scatter sls.f90:4113-4150, SLS_solve_ir's residual sls.f90:4826-4934): duplicates summed,
    out-of-range entries ignored, both triangles applied."""
    import ctypes as C
    from galahad_amd import SLS, SMT, Control, InformSLS
    from galahad_amd._lib import Inform, lib
    n, row, col, val, rhs, xs = P.random_sparse(700, 6, seed=9, spd=False)
    rng = np.random.default_rng(4)
    # duplicates (split some values in two) and a few out-of-range entries
    k = rng.choice(len(val), 150, replace=False)
    row = np.concatenate([row, row[k], [0, n + 1]]).astype(np.int32)
    col = np.concatenate([col, col[k], [1, 2]]).astype(np.int32)
    half = 0.37 * val[k]
    val2 = val.copy()
    val2[k] -= half
    val2 = np.concatenate([val2, half, [5.0, 7.0]])
    m = SMT(n, "COORDINATE", row=row, col=col, val=val2)
    s, c, i = SLS(), Control(), InformSLS()
    s.initialize("gsls", c, i)
    s.analyse(m, c, i)
    assert i.status == 0
    s.factorize(m, c, i)            # gsls_factor_coo
    assert i.status == 0
    x = s.solve(m, rhs, c, i)
    assert P.scaled_residual(n, row[:-2], col[:-2], val2[:-2], x, rhs) <= 1e-11
    # the same factorization from host-scattered values: bitwise the same solution
    s2, c2, i2 = SLS(), Control(), InformSLS()
    s2.initialize("gsls", c2, i2)
    s2.analyse(m, c2, i2)
    VAL = s2.scatter_values(m)
    ginf = Inform()
    assert lib.gsls_factor(s2.handle, 0, VAL.ctypes.data_as(C.c_void_p), None, C.byref(s2.opts), C.byref(ginf)) >= 0
    x2 = s2._backend_solve(np.array(rhs), 0, i2)
    assert np.array_equal(x, x2)
    # residual on the device against the facade's host loop
    X = rng.uniform(-1, 1, (n, 2))
    B = rng.uniform(-1, 1, (n, 2))
    r_dev = s._residual_dev(m, B, X)
    r_host = s._residual(m, B, X)
    assert np.abs(r_dev - r_host).max() <= 1e-13 * np.abs(r_host).max()
    c.max_iterative_refinements = 2
    xr = s.solve(m, rhs, c, i)
    assert P.scaled_residual(n, row[:-2], col[:-2], val2[:-2], xr, rhs) <= 1e-14
    s.terminate()
    s2.terminate()


@pytest.mark.parametrize("nrhs", [2, 3, 8, 13])
def test_blocked_multi_rhs_equals_column_by_column(nrhs):
    """SLS_solve with several right-hand sides (ssids_solve_mult, ssids.f90:1139-1249): the Cholesky path takes blocks of
    8 / 4 / 2 columns through one pass over L (k_solve_*_chol_mr).  Every column must carry exactly the bits the
    single-column kernels produce, for the whole solve and for the partial solves (job L and U)."""
    prob = P.banded_spd(6000, 47, seed=3)
    n, row, col, val, rhs, xs = prob
    s, m, c, i = run_gsls(prob, True)
    assert i.status == 0
    c.max_iterative_refinements = 0
    rng = np.random.default_rng(nrhs)
    B = np.asfortranarray(rng.uniform(-1, 1, (n, nrhs)))
    X = s.solve(m, B, c, i)
    assert i.status == 0 and X.shape == (n, nrhs)
    for k in range(nrhs):
        xk = s.solve(m, B[:, k].copy(), c, i)
        assert np.array_equal(X[:, k], xk), k
        assert P.scaled_residual(n, row, col, val, X[:, k], B[:, k]) <= 1e-13
    Y = s.part_solve("L", B, c, i)
    Z = s.part_solve("U", Y, c, i)
    for k in range(nrhs):
        yk = s.part_solve("L", B[:, k].copy(), c, i)
        assert np.array_equal(Y[:, k], yk)
    assert np.array_equal(Z, X)
    s.terminate()


@pytest.mark.parametrize("kind", ["kkt", "kkt_pivoting", "grid3d", "grid3d_chol", "scaled"])
@pytest.mark.parametrize("nrhs", [2, 4, 5, 9, 16, 17])
def test_multi_rhs_columns_in_one_launch_equal_column_by_column(kind, nrhs):
    """Several right-hand sides on the LDL^T path go through every launch of the single-column kernels together (up to
    eight columns: block b works on column b % R, struct Cols in gsls_device.hip; ssids_solve_mult, ssids.f90:1139-1249).
    Every column must carry exactly the bits of a single-column solve -- whole solves and the partial solves -- with the
    wave tier (kkt), with 2x2 pivots everywhere (kkt_pivoting), with the blocked big-front path (grid3d), on the
    Cholesky path for fronts too tall for its own blocked multi-column kernels (grid3d_chol) and with a scaling vector
    (scaled); repeated, so that the column sets are reused."""
    if kind == "kkt":
        prob = P.kkt_qpband(4000, 1500, seed=5)
    elif kind == "kkt_pivoting":          # K = [0 B; B^T 0]: every pivot is 2x2
        rng = np.random.default_rng(1)
        nb = 300
        r, cc, v = [], [], []
        for a in range(nb):
            for b in {a} | set(rng.integers(0, nb, 3).tolist()):
                r.append(nb + b), cc.append(a), v.append(2.0 + rng.uniform(0, 1) if b == a else rng.uniform(-0.3, 0.3))
        row0, col0, val0 = np.array(r), np.array(cc), np.array(v)
        x0 = rng.uniform(-1, 1, 2 * nb)
        prob = (2 * nb, (row0 + 1).astype(np.int32), (col0 + 1).astype(np.int32), val0,
                P.sym_matvec(2 * nb, row0, col0, val0, x0), x0)
    elif kind in ("grid3d", "grid3d_chol"):   # fronts of hundreds of columns: the blocked big-front path
        prob = P.grid3d(30, 30, 30) if kind == "grid3d_chol" else P.grid3d(14, 14, 14)
    else:
        prob = P.kkt_qpband(3000, 1000, seed=11)
    n, row, col, val, rhs, xs = prob
    from galahad_amd import SLS, SMT, Control, InformSLS
    s, c, i = SLS(), Control(), InformSLS()
    s.initialize("gsls", c, i)
    c.pivot_control, c.node_amalgamation, c.max_iterative_refinements = 1, 24, 0
    if kind == "scaled":
        c.scaling = 1
    if kind == "grid3d_chol":
        c.pivot_control = 2
    m = SMT(n, "COORDINATE", row=row, col=col, val=val)
    s.analyse(m, c, i)
    s.factorize(m, c, i)
    assert i.status == 0, i.gsls_inform
    s.factorize(m, c, i)                      # the refactorization path (wave-per-front kernels, packed images)
    assert i.status == 0
    rng = np.random.default_rng(100 + nrhs)
    B = np.asfortranarray(rng.uniform(-1, 1, (n, nrhs)))
    for rep in range(2):
        X = s.solve(m, B, c, i)
        assert i.status == 0 and X.shape == (n, nrhs)
        for k in range(nrhs):
            xk = s.solve(m, B[:, k].copy(), c, i)
            assert np.array_equal(X[:, k], xk), (rep, k)
            assert P.scaled_residual(n, row, col, val, X[:, k], B[:, k]) <= 1e-10
    for part in ("L", "U") if kind == "grid3d_chol" else ("L", "D", "U", "S"):
        Y = s.part_solve(part, B, c, i)
        for k in range(0, nrhs, 3):
            assert np.array_equal(Y[:, k], s.part_solve(part, B[:, k].copy(), c, i)), (part, k)
    s.terminate()


@pytest.mark.parametrize("kind", ["kkt", "kkt_scaled", "band_spd", "grid_indef"])
@pytest.mark.parametrize("nrhs", [1, 3])
def test_solve_with_the_right_hand_side_in_another_device_array(kind, nrhs):
    """gsls_solve_dev_rhs: b stays where it is (bit for bit), x = the bits of the in-place solve of a copy -- on the wave
    tier's direct path (one column, whole solve: the bottom-stage launch reads b itself), with a scaling vector, on the
    Cholesky path and with several columns (those copy first)."""
    import ctypes as C
    import torch
    from galahad_amd import SLS, SMT, Control, InformSLS
    from galahad_amd._lib import lib, Inform
    if kind.startswith("kkt"):
        prob = P.kkt_qpband(5000, 1500, seed=3)
    elif kind == "band_spd":
        prob = P.banded_spd(3000, 20)
    else:
        prob = P.grid2d(40, 30, shift=1.0)
    n, row, col, val, rhs, xs = prob
    s, c, i = SLS(), Control(), InformSLS()
    s.initialize("gsls", c, i)
    c.pivot_control, c.node_amalgamation = (2 if kind == "band_spd" else 1), 24
    if kind == "kkt_scaled":
        c.scaling = -1
    m = SMT(n, "COORDINATE", row=row, col=col, val=val)
    s.analyse(m, c, i)
    s.factorize(m, c, i)
    s.factorize(m, c, i)
    assert i.status == 0
    rng = np.random.default_rng(5)
    B = torch.from_numpy(rng.uniform(-1, 1, (nrhs, n))).cuda()           # row k = column k, ldx = n
    inf = Inform()
    for job in (0, 1):                                                    # whole solve; forward only (never the direct path)
        ref = B.clone()
        assert lib.gsls_solve_dev(s.handle, job, nrhs, C.c_void_p(ref.data_ptr()), n, C.byref(s.opts), C.byref(inf)) >= 0
        keep = B.clone()
        X = torch.full_like(B, float("nan"))
        assert lib.gsls_solve_dev_rhs(s.handle, job, nrhs, C.c_void_p(B.data_ptr()), C.c_void_p(X.data_ptr()), n,
                                      C.byref(s.opts), C.byref(inf)) >= 0
        torch.cuda.synchronize()
        assert torch.equal(B, keep)
        assert torch.equal(X, ref), (kind, nrhs, job)
    # d_b = d_x and d_b = NULL are the in-place call
    Y = B.clone()
    assert lib.gsls_solve_dev_rhs(s.handle, 0, nrhs, C.c_void_p(Y.data_ptr()), C.c_void_p(Y.data_ptr()), n,
                                  C.byref(s.opts), C.byref(inf)) >= 0
    Z = B.clone()
    assert lib.gsls_solve_dev_rhs(s.handle, 0, nrhs, None, C.c_void_p(Z.data_ptr()), n, C.byref(s.opts), C.byref(inf)) >= 0
    ref = B.clone()
    assert lib.gsls_solve_dev(s.handle, 0, nrhs, C.c_void_p(ref.data_ptr()), n, C.byref(s.opts), C.byref(inf)) >= 0
    torch.cuda.synchronize()
    assert torch.equal(Y, ref) and torch.equal(Z, ref)
    s.terminate()


@pytest.mark.parametrize("posdef", [False, True])
def test_multi_rhs_with_padded_leading_dimension(posdef):
    """gsls_solve / gsls_solve_dev take X(ldx, nrhs) with ldx >= n (ssids_solve_mult's ldx, ssids.f90:1139-1160): columns
    ldx apart, the rows from n on untouched -- on the host entry point and on the device one."""
    import ctypes as C
    import torch
    from galahad_amd._lib import lib, Inform
    prob = P.banded_spd(3000, 21, seed=4) if posdef else P.kkt_qpband(2500, 900, seed=9)
    n, row, col, val, rhs, xs = prob
    s, m, c, i = run_gsls(prob, posdef, nemin=24, ordering_free=True)
    assert i.status == 0
    c.max_iterative_refinements = 0
    nrhs, ldx = 11, n + 37
    rng = np.random.default_rng(3)
    B = rng.uniform(-1, 1, (nrhs, ldx))                  # row k = column k of the Fortran array X(ldx, nrhs)
    ref = np.stack([s.solve(m, B[k, :n].copy(), c, i) for k in range(nrhs)])
    inf = Inform()
    X = B.copy()
    f = lib.gsls_solve(s.handle, 0, nrhs, C.c_void_p(X.ctypes.data), ldx, C.byref(s.opts), C.byref(inf))
    assert f >= 0
    assert np.array_equal(X[:, :n], ref) and np.array_equal(X[:, n:], B[:, n:])
    Xd = torch.from_numpy(B.copy()).cuda()
    f = lib.gsls_solve_dev(s.handle, 0, nrhs, C.c_void_p(Xd.data_ptr()), ldx, C.byref(s.opts), C.byref(inf))
    assert f >= 0
    Xh = Xd.cpu().numpy()
    assert np.array_equal(Xh[:, :n], ref) and np.array_equal(Xh[:, n:], B[:, n:])
    s.terminate()


SCALED = sorted(glob.glob(os.path.join(HERE, "golden", "scaled_*.npz")))


@pytest.mark.parametrize("path", SCALED, ids=[os.path.basename(p)[:-4] for p in SCALED])
def test_reference_scalings(path):
    """control%scaling = -1 (Hungarian) / -2 (auction) on systems whose rows span eight decades, against the reference
    run with the same control (fixtures from tests/golden/make_golden.py scaled).  The scale vectors are not unique, so
    the comparison is what a caller sees: inertia, residual, the solution to the accuracy the reference itself reaches
    against the generating x, and far fewer delayed pivots than without scaling (why the option exists)."""
    from galahad_amd import SLS, SMT, Control, InformSLS
    g = np.load(path)
    n = int(g["n"])
    row, col, val, rhs = g["row"], g["col"], g["val"], g["rhs"]
    m = SMT(n, "COORDINATE", row=row, col=col, val=val)
    out = {}
    for scaling in (int(g["scaling"]), 0):
        s, c, i = SLS(), Control(), InformSLS()
        s.initialize("gsls", c, i)
        c.pivot_control, c.node_amalgamation, c.scaling = 1, int(g["nemin"]), scaling
        c.max_iterative_refinements = 0
        s.analyse(m, c, i, PERM=g["perm"])
        s.factorize(m, c, i)
        assert i.status == 0, i.gsls_inform
        x = s.solve(m, rhs, c, i)
        out[scaling] = (x, i.delayed_pivots, i.negative_eigenvalues, i.rank)
        if scaling != 0:
            sc = np.zeros(n)
            from galahad_amd._lib import lib
            import ctypes as C
            assert lib.gsls_get_scaling(s.handle, sc.ctypes.data_as(C.POINTER(C.c_double))) == 0
            sv = np.abs(val) * sc[row - 1] * sc[col - 1]
            assert sv.max() <= (1.0 + 1e-10 if scaling == -1 else 1e3)
        s.terminate()
    x, delayed, neg, rank = out[int(g["scaling"])]
    assert neg == int(g["ref_neg"]) and rank == int(g["ref_rank"])
    assert P.scaled_residual(n, row, col, val, x, rhs) <= 1e-10
    ref_err = np.abs(g["ref_x"] - g["xstar"]).max()
    assert np.abs(x - g["xstar"]).max() <= 100 * max(ref_err, 1e-12)
    assert out[0][2] == neg                                   # inertia does not depend on the scaling
    assert delayed <= out[0][1]


@pytest.mark.parametrize("path", [p for p in SCALED if "hungarian" in p],
                         ids=[os.path.basename(p)[:-4] for p in SCALED if "hungarian" in p])
def test_matching_based_ordering_and_its_saved_scaling(path):
    """gsls_analyse_matching + options.scaling = 3 (ssids_analyse with val / ordering = 2, then ssids_factor with
    scaling = 3: ssids.f90:305-320, 991-994; spral/match_order.f90) on the badly scaled systems of the scaling fixtures.
    The reference cannot reach this combination through SLS (its ssids arm never passes val to the analysis), so there is
    no reference-generated vector for it: parity here is UNPINNED beyond what the fixtures hold -- the scaling saved is
    the Hungarian one (same matching), so inertia and rank must equal the fixture's, the residual must meet the same bar,
    and scaling = 3 WITHOUT the matching analysis must give -15 as ssids does."""
    import ctypes as C
    from galahad_amd._lib import lib, Options, Inform
    from oracle.oracle import lower_csc
    g = np.load(path)
    n = int(g["n"])
    row, col, val, rhs = g["row"], g["col"], g["val"], g["rhs"]
    ptr, r, v = lower_csc(n, row, col, val)
    ptr, r, v = (np.ascontiguousarray(ptr, np.int64), np.ascontiguousarray(r, np.int32), np.ascontiguousarray(v, np.float64))
    p64, p32, pd = C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_void_p
    o, inf = Options(), Inform()
    lib.gsls_default_options(C.byref(o))
    o.nemin = int(g["nemin"])
    # (a) scaling = 3 after a plain analysis: no saved scaling
    h = C.c_void_p()
    assert lib.gsls_create(C.byref(h)) == 0
    order = np.arange(1, n + 1, dtype=np.int32)
    o.ordering, o.scaling = 1, 0
    assert lib.gsls_analyse(h, n, ptr.ctypes.data_as(p64), r.ctypes.data_as(p32), order.ctypes.data_as(p32), C.byref(o),
                            C.byref(inf)) >= 0
    o.scaling = 3
    assert lib.gsls_factor(h, 0, v.ctypes.data_as(pd), None, C.byref(o), C.byref(inf)) == -15
    lib.gsls_destroy(C.byref(h))
    # (b) the matching-based analysis, then scaling = 3
    h = C.c_void_p()
    assert lib.gsls_create(C.byref(h)) == 0
    o.ordering, o.scaling = 1, 3
    order = np.zeros(n, dtype=np.int32)
    assert lib.gsls_analyse_matching(h, n, ptr.ctypes.data_as(p64), r.ctypes.data_as(p32), v.ctypes.data_as(pd),
                                     order.ctypes.data_as(p32), C.byref(o), C.byref(inf)) >= 0
    assert sorted(order.tolist()) == list(range(1, n + 1))
    assert lib.gsls_factor(h, 0, v.ctypes.data_as(pd), None, C.byref(o), C.byref(inf)) >= 0, inf.flag
    assert inf.num_neg == int(g["ref_neg"]) and inf.matrix_rank == int(g["ref_rank"])
    sc = np.zeros(n)
    assert lib.gsls_get_scaling(h, sc.ctypes.data_as(C.POINTER(C.c_double))) == 0
    assert (np.abs(val) * sc[row - 1] * sc[col - 1]).max() <= 1.0 + 1e-10          # a maximum-product matching scaling
    x = np.array(rhs, dtype=np.float64)
    assert lib.gsls_solve(h, 0, 1, x.ctypes.data_as(pd), n, C.byref(o), C.byref(inf)) >= 0
    assert P.scaled_residual(n, row, col, val, x, rhs) <= 1e-10
    ref_err = np.abs(g["ref_x"] - g["xstar"]).max()
    assert np.abs(x - g["xstar"]).max() <= 100 * max(ref_err, 1e-12)
    lib.gsls_destroy(C.byref(h))


@pytest.mark.parametrize("n", [1, 2, 3, 31, 32, 33, 48, 49, 63, 64, 65, 129])
@pytest.mark.parametrize("kind", ["spd", "indef"])
def test_edge_sizes_dense_single_front(n, kind):
    """one dense front of exactly n columns around every boundary of the kernels' tiers (32 / 48 / 64 columns and rows:
    wave-per-front unrolled, blocked LDS, workgroup kernels; wave tier of the solves up to 64 rows), first factorization
    and refactorization, against numpy's dense solve."""
    from galahad_amd import SLS, SMT, Control, InformSLS
    rng = np.random.default_rng(100 * n + (kind == "spd"))
    B = rng.uniform(-1, 1, (n, n))
    A = B @ B.T / n + np.eye(n) if kind == "spd" else (B + B.T) / 2 + np.diag(np.where(np.arange(n) % 2 == 0, 3.0, -3.0))
    r, c = np.tril_indices(n)
    row, col, val = (r + 1).astype(np.int32), (c + 1).astype(np.int32), A[r, c]
    xs = rng.uniform(-1, 1, n)
    rhs = A @ xs
    m = SMT(n, "COORDINATE", row=row, col=col, val=val)
    s, ctl, i = SLS(), Control(), InformSLS()
    s.initialize("gsls", ctl, i)
    ctl.pivot_control = 2 if kind == "spd" else 1
    ctl.max_iterative_refinements = 0
    s.analyse(m, ctl, i, PERM=np.arange(1, n + 1))
    assert i.status == 0
    xd = np.linalg.solve(A, rhs)
    for rep in range(3):                       # first factorization, then the learned / refactorization path twice
        s.factorize(m, ctl, i)
        assert i.status == 0, (rep, i.gsls_inform)
        assert i.rank == n
        assert i.negative_eigenvalues == int((np.linalg.eigvalsh(A) < 0).sum())
        x = s.solve(m, rhs, ctl, i)
        assert np.abs(x - xd).max() <= 1e-9 * max(1.0, np.abs(xd).max()), rep
    s.terminate()


def test_empty_system():
    """n = 0: SLS refuses it as the reference does (GALAHAD_error_restrictions, sls.f90:2217-2222); at the ABI every call
    is a no-op that succeeds (ssids.f90:256-262, 1180-1183)"""
    import ctypes as C
    from galahad_amd import SLS, SMT, Control, InformSLS
    from galahad_amd._lib import Inform, Options, lib
    m = SMT(0, "COORDINATE", row=np.zeros(0, np.int32), col=np.zeros(0, np.int32), val=np.zeros(0))
    s, ctl, i = SLS(), Control(), InformSLS()
    s.initialize("gsls", ctl, i)
    s.analyse(m, ctl, i)
    assert i.status == -3
    s.terminate()
    h = C.c_void_p()
    assert lib.gsls_create(C.byref(h)) == 0
    o, inf = Options(), Inform()
    lib.gsls_default_options(C.byref(o))
    ptr = np.ones(1, dtype=np.int64)
    assert lib.gsls_analyse(h, 0, ptr.ctypes.data_as(C.POINTER(C.c_int64)), None, None, C.byref(o), C.byref(inf)) == 0
    assert lib.gsls_factor(h, 0, None, None, C.byref(o), C.byref(inf)) == 0
    assert lib.gsls_solve(h, 0, 1, None, 0, C.byref(o), C.byref(inf)) == 0
    lib.gsls_destroy(C.byref(h))


def test_learned_2x2_pivots_stay_on_the_wave_per_front_path():
    """K = [0 B; B^T 0] with a sparse B: every pivot is 2x2.  After the first factorization has learned the pairs, the
    refactorizations must run on the wave-per-front path (k_front_blk eliminates the hinted pairs inside its 4-column
    panels) -- no front on the blacklist, the same inertia and solution every time."""
    import ctypes as C
    from galahad_amd import SLS, SMT, Control, InformSLS
    from galahad_amd._lib import lib
    nb = 3000
    rng = np.random.default_rng(5)
    rows, cols, vals = [], [], []
    for i in range(nb):
        for j in (i - 1, i, i + 1):
            if 0 <= j < nb:
                rows.append(nb + j + 1)
                cols.append(i + 1)
                vals.append(4.0 if i == j else rng.uniform(-1, 1))
    n = 2 * nb
    row, col, val = np.array(rows, np.int32), np.array(cols, np.int32), np.array(vals)
    xs = rng.uniform(-1, 1, n)
    rhs = P.sym_matvec(n, row - 1, col - 1, val, xs)
    m = SMT(n, "COORDINATE", row=row, col=col, val=val)
    s, c, i = SLS(), Control(), InformSLS()
    s.initialize("gsls", c, i)
    c.pivot_control, c.node_amalgamation, c.max_iterative_refinements = 1, 16, 0
    s.analyse(m, c, i)
    xprev = None
    for rep in range(3):
        s.factorize(m, c, i)
        assert i.status == 0, i.gsls_inform
        assert i.negative_eigenvalues == nb and i.two_by_two_pivots == nb and i.rank == n
        x = s.solve(m, rhs, c, i)
        assert P.scaled_residual(n, row, col, val, x, rhs) <= 1e-13
        fb, pb, bl = C.c_int32(), C.c_int32(), C.c_int32()
        lib.gsls_get_factor_stats(s.handle, C.byref(fb), C.byref(pb), C.byref(bl))
        if rep >= 1:
            assert pb.value == 0 and bl.value == 0, (rep, fb.value, pb.value, bl.value)
            if xprev is not None and rep >= 2:
                assert np.array_equal(x, xprev)
        xprev = x
    s.terminate()


def test_wave_tier_backward_with_pivoting_inside_a_front():
    """regression (found by tools/soak.py, seed 4, system 55): a 227-variable indefinite system whose accepted
    factorization still holds a front with numerical pivoting inside (gperm not the identity there).  The wave tier's
    backward step handed the front's solution to its children by pivot slot instead of by analyse-time row: status 0,
    correct inertia, solution wrong by 0.4.  Forward, D and backward part solves and the full solve against numpy."""
    from galahad_amd import SLS, SMT, Control, InformSLS
    g = np.load(os.path.join(HERE, "data", "soak_seed4_it55.npz"))
    A, rhs = g["A"], g["rhs"]
    n = A.shape[0]
    r, c = np.nonzero(np.tril(A))
    m = SMT(n, "COORDINATE", row=(r + 1).astype(np.int32), col=(c + 1).astype(np.int32), val=A[r, c])
    s, ctl, i = SLS(), Control(), InformSLS()
    s.initialize("gsls", ctl, i)
    ctl.pivot_control, ctl.node_amalgamation, ctl.max_iterative_refinements = 1, int(g["nemin"]), 0
    s.analyse(m, ctl, i)
    xd = np.linalg.solve(A, rhs)
    for rep in range(3):
        s.factorize(m, ctl, i)
        assert i.status == 0 and i.negative_eigenvalues == int((np.linalg.eigvalsh(A) < 0).sum())
        x = s.solve(m, rhs, ctl, i)
        assert np.abs(x - xd).max() <= 1e-10, rep
        y = s.part_solve("L", rhs.copy(), ctl, i)
        z = s.part_solve("D", y, ctl, i)
        x2 = s.part_solve("U", z, ctl, i)
        assert np.abs(x2 - xd).max() <= 1e-10, rep
    s.terminate()


@pytest.mark.parametrize("mode,count,seed", [("", 240, 11), ("SOAK_WIDE", 60, 12), ("SOAK_BIG", 24, 13)])
def test_randomised_soak(mode, count, seed):
    """tools/soak.py on seeded systems against numpy (solutions, inertia, rank; three factorizations each, some with the
    backend's scalings, several right-hand sides, refinement on the device): 240 small ones (SPD / indefinite / saddle /
    weak diagonals, 5..260 variables, own ordering or random PERM, nemin 1..64), 60 with wide fronts (dense, banded,
    arrow; diagonally dominant or far from it), 24 bigger sparse ones (300..2500 variables)."""
    import subprocess
    import sys
    env = dict(os.environ)
    if mode:
        env[mode] = "1"
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "tools", "soak.py"), str(count), str(seed)],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0 and "0 failures" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_zero_pivot_needs_the_whole_column_to_be_negligible():
    """regression (tools/soak.py, SOAK_BIG=1, seed 22, system 42): a 2158-variable saddle system under a random PERM.  A
    column whose entries were all zero inside the first 128 rows of a tall front -- all the diagonal kernel sees -- was
    declared a zero pivot although it had entries further down (the panel kernel then multiplied them by 0): status 0,
    two negative eigenvalues missing, solution wrong by 0.46.  The reference tests the whole column
    (ldlt_tpp.cxx:250-262); k_panel now reports such a column as failed and it is delayed like any other."""
    from galahad_amd import SLS, SMT, Control, InformSLS
    g = np.load(os.path.join(HERE, "data", "soak_big_seed22_it42.npz"))
    n, row, col, val, rhs, xs = int(g["n"]), g["row"], g["col"], g["val"], g["rhs"], g["xs"]
    m = SMT(n, "COORDINATE", row=row, col=col, val=val)
    s, ctl, i = SLS(), Control(), InformSLS()
    s.initialize("gsls", ctl, i)
    ctl.pivot_control, ctl.node_amalgamation, ctl.max_iterative_refinements = 1, int(g["nemin"]), 0
    s.analyse(m, ctl, i, PERM=g["perm"])
    import scipy.sparse as sp
    A = sp.coo_matrix((val, (row - 1, col - 1)), shape=(n, n)).toarray()
    A = A + A.T - np.diag(np.diag(A))
    nneg = int((np.linalg.eigvalsh(A) < 0).sum())
    for rep in range(2):
        s.factorize(m, ctl, i)
        assert i.status == 0 and i.rank == n and i.negative_eigenvalues == nneg, (rep, i.rank, i.negative_eigenvalues, nneg)
        x = s.solve(m, rhs, ctl, i)
        assert np.abs(x - xs).max() <= 1e-9
    s.terminate()
