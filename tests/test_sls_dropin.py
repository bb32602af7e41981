"""Drop-in proof: the REAL GALAHAD SLS facade (src/sls/sls.f90 of the reference, patched at build time
with the `CASE ( 'gsls' )` arms of INTEGRATION.md by integration/patch_sls.py) linked against the
MI355X backend -- oracle/_ref/sls_gsls_driver, built by oracle/build_ref.sh next to the plain
reference driver.  Callers say SLS_initialize('gsls', ...) and nothing else changes: SLS's own
coordinate->CSR map, value scatter, iterative refinement and status mapping run unmodified.
The binary is test infrastructure (it contains reference objects) and never ships in galahad_amd/."""
import glob
import os

import numpy as np
import pytest

import problems as P

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = sorted(p for p in glob.glob(os.path.join(HERE, "golden", "*.npz")) if not os.path.basename(p).startswith("scaled_"))


def _need_dropin():
    from oracle import refio
    if not refio.dropin_available():
        pytest.skip("oracle/_ref/sls_gsls_driver not built (needs /root/reference at build time)")
    return refio


def test_dropin_analyse_through_real_sls_and_loud_failure(have_gpu):
    """CPU: SLS_analyse through the facade reaches gsls_analyse (same statistics as the reference's
    ssids arm); without a device SLS_factorize reports GALAHAD_error_technical (-50), not a result."""
    refio = _need_dropin()
    n, row, col, val, rhs, xs = P.kat_indefinite()
    r = refio.run(n, row, col, val, rhs, solver="gsls", perm=np.arange(1, n + 1), nemin=32)
    assert r["status_analyse"] == 0
    if not have_gpu:
        assert r["status_factorize"] == -50
    else:
        assert r["status_factorize"] == 0 and np.abs(r["x"] - xs).max() <= 1.5e-8


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_dropin_matches_reference_golden(path):
    refio = _need_dropin()
    g = np.load(path)
    n = int(g["n"])
    posdef = bool(g["posdef"])
    r = refio.run(n, g["row"], g["col"], g["val"], g["rhs"], solver="gsls", perm=g["perm"],
                  nemin=int(g["nemin"]), pivot_control=2 if posdef else 1)
    assert (r["status_analyse"], r["status_factorize"], r["status_solve"]) == (0, 0, 0)
    assert r["negative_eigenvalues"] == int(g["ref_neg"]) and r["rank"] == int(g["ref_rank"])
    if r["delayed"] == 0:
        assert r["entries_in_factors"] == int(g["ref_num_factor"])
        assert r["flops_elimination"] == int(g["ref_num_flops"])
    assert np.abs(r["x"] - g["ref_x"]).max() <= 1e-9 * max(1.0, np.abs(g["ref_x"]).max())


@pytest.mark.gpu
def test_dropin_defaults_refinement_and_not_posdef():
    """solver-specific defaults (own ordering, no PERM), SLS's own iterative refinement loop
    (sls.f90:4751-4963) around the backend, and 'not positive definite' -> GALAHAD_error_inertia."""
    refio = _need_dropin()
    prob = P.kkt_qpband(20000, 4000)
    n, row, col, val, rhs, xs = prob
    r = refio.run(n, row, col, val, rhs, solver="gsls", max_refine=1)
    assert (r["status_analyse"], r["status_factorize"], r["status_solve"]) == (0, 0, 0)
    assert r["negative_eigenvalues"] == 4000 and r["rank"] == 24000
    assert P.scaled_residual(n, row, col, val, r["x"], rhs) <= 1e-14
    n, row, col, val, rhs, xs = P.kat_indefinite()
    r = refio.run(n, row, col, val, rhs, solver="gsls", pivot_control=2)
    assert r["status_factorize"] == -20


@pytest.mark.gpu
def test_cfg3_full_size_against_the_reference():
    """BASELINE.json configs[2] at FULL size (n = 1e6, m = 2e5, order 1.2e6) against the reference itself: the CPU
    run (GALAHAD SLS + SSIDS) is handed the elimination order the GPU run used, so fill and flop counts must be
    identical numbers, the inertia (n, m, 0), no delayed pivots on either side, and the solutions agree to the
    forward-error bar of SURVEY.md section 8d.  (bench.py prints the same comparison as `cpu_baseline.max_err`.)"""
    from oracle import refio
    if not refio.available():
        pytest.skip("oracle/_ref/ref_driver not built")
    from galahad_amd import SLS, SMT, Control, InformSLS
    n0, m0 = 1000000, 200000
    prob = P.kkt_qpband(n0, m0)
    n, row, col, val, rhs, xs = prob
    m = SMT(n, "COORDINATE", row=row, col=col, val=val)
    s, c, i = SLS(), Control(), InformSLS()
    s.initialize("gsls", c, i)
    c.pivot_control, c.node_amalgamation = 1, 24
    s.analyse(m, c, i)
    s.factorize(m, c, i)
    assert i.status == 0, i.gsls_inform
    x = s.solve(m, rhs, c, i)
    order = s.ORDER.copy()
    r = refio.run(n, row, col, val, rhs, perm=order, pivot_control=1, nemin=24, threads=16, timeout=1200)
    assert (r["status_analyse"], r["status_factorize"], r["status_solve"]) == (0, 0, 0)
    assert r["negative_eigenvalues"] == i.negative_eigenvalues == m0 and r["rank"] == i.rank == n
    assert r["delayed"] == 0 and i.delayed_pivots == 0
    assert r["entries_in_factors"] == i.entries_in_factors and r["flops_elimination"] == i.flops_elimination
    assert np.abs(x - r["x"]).max() <= 1e-9 * np.abs(r["x"]).max()
    assert P.scaled_residual(n, row, col, val, x, rhs) <= 1e-10
    s.terminate()


@pytest.mark.gpu
def test_cfg5_full_size_properties():
    """BASELINE.json configs[4] as SURVEY.md section 8(d) specifies it: 126^3 grid, 27-point-style stencil with 10 %
    of the edges replaced by long-range ones (seed 20240105), diagonally dominant, n = 2 000 376, through the PIVOTED
    LDL^T path on one GPU (the factor is ~94 GB and stays in HBM; the contribution arena is reused by lifetime).
    The CPU reference needs hours for 3e14 flops, so at this size the check is the size-independent properties:
    inertia (n, 0, 0), no delays, scaled residual <= 1e-13 without refinement, forward error against the generating
    solution, and a repeated solve giving the same bits.  The same pattern at 36^3 is compared with the oracle in
    test_gpu_parity.py and runs 2-rank sharded in test_shard_gpu.py."""
    from galahad_amd import SLS, SMT, Control, InformSLS
    prob = P.grid3d_27pt_perturbed(126, 126, 126)
    n, row, col, val, rhs, xs = prob
    assert n == 126 ** 3
    m = SMT(n, "COORDINATE", row=row, col=col, val=val)
    s, c, i = SLS(), Control(), InformSLS()
    s.initialize("gsls", c, i)
    c.pivot_control = 1
    c.max_iterative_refinements = 0
    s.analyse(m, c, i)
    assert i.status == 0
    s.factorize(m, c, i)
    assert i.status == 0, i.gsls_inform
    assert i.negative_eigenvalues == 0 and i.rank == n and i.delayed_pivots == 0
    x = s.solve(m, rhs, c, i)
    assert P.scaled_residual(n, row, col, val, x, rhs) <= 1e-13
    assert np.abs(x - xs).max() <= 1e-10
    x2 = s.solve(m, rhs, c, i)
    assert np.array_equal(x, x2)
    s.terminate()


@pytest.mark.gpu
@pytest.mark.parametrize("exe", ["slst_c_gsls", "slstf_c_gsls"])
def test_reference_c_interface_test_with_gsls(exe):
    """the reference's C interface (include/sls.h, src/sls/C/sls_ciface.f90) compiled above the
    patched facade, and the reference's OWN C tests of it (src/sls/C/slst.c, slstf.c, compiled where they lie by
    oracle/build_ref.sh) run with solver "gsls" instead of "sils" (the solver string is substituted at link time,
    oracle/ciface_solver_wrap.c).  Three storage formats x {solve, solve with refinement, L/D/U part solves}: all ok."""
    import subprocess
    path = os.path.join(os.path.dirname(HERE), "oracle", "_ref", exe)
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/%s not built" % exe)
    p = subprocess.run([path], env=dict(os.environ, GSLS_CTEST_SOLVER="gsls"), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    rows = [ln for ln in p.stdout.splitlines() if ln.strip().startswith(("coordinate", "sparse by rows", "dense"))]
    assert len(rows) == 3, p.stdout
    for ln in rows:
        assert ln.count("ok") == 3 and "fail" not in ln and "status" not in ln, p.stdout


SCALED = sorted(glob.glob(os.path.join(HERE, "golden", "scaled_*.npz")))


@pytest.mark.gpu
@pytest.mark.parametrize("path", SCALED, ids=[os.path.basename(p)[:-4] for p in SCALED])
def test_dropin_scaling_controls_through_real_sls(path):
    """control%scaling = -1 / -2 through the REAL facade (SLS_copy_control_to_gsls in integration/patch_sls.py maps it
    as the ssids arm does, sls.f90:1405-1413) against the reference run with the same control."""
    refio = _need_dropin()
    g = np.load(path)
    n = int(g["n"])
    r = refio.run(n, g["row"], g["col"], g["val"], g["rhs"], solver="gsls", perm=g["perm"], nemin=int(g["nemin"]),
                  pivot_control=1, scaling=int(g["scaling"]))
    assert (r["status_analyse"], r["status_factorize"], r["status_solve"]) == (0, 0, 0)
    assert r["negative_eigenvalues"] == int(g["ref_neg"]) and r["rank"] == int(g["ref_rank"])
    ref_err = np.abs(g["ref_x"] - g["xstar"]).max()
    assert np.abs(r["x"] - g["xstar"]).max() <= 100 * max(ref_err, 1e-12)
    plain = refio.run(n, g["row"], g["col"], g["val"], g["rhs"], solver="gsls", perm=g["perm"], nemin=int(g["nemin"]),
                      pivot_control=1)
    assert r["delayed"] <= plain["delayed"]


@pytest.mark.gpu
def test_dropin_minimum_degree_ordering_through_real_sls():
    """control%ordering = 1 (AMD) through the real facade: for every other solver SLS would call MC68 -- a stub in the
    reference tree -- while the gsls arm keeps mc6168_ordering false and the backend orders by itself
    (integration/patch_sls.py).  Irregular pattern: the ordering must beat the natural order's fill by a wide margin."""
    refio = _need_dropin()
    n, row, col, val, rhs, xs = P.random_sparse(3000, 4, 21, spd=True)
    nat = refio.run(n, row, col, val, rhs, solver="gsls", perm=np.arange(1, n + 1), pivot_control=2)
    amd = refio.run(n, row, col, val, rhs, solver="gsls", ordering=1, pivot_control=2)
    assert (amd["status_analyse"], amd["status_factorize"], amd["status_solve"]) == (0, 0, 0)
    assert amd["entries_in_factors"] <= 0.6 * nat["entries_in_factors"]
    assert np.abs(amd["x"] - xs).max() <= 1e-9


@pytest.mark.gpu
def test_randomised_soak_through_the_real_facade():
    """tools/soak_facade.py: 60 seeded systems as untidy COO (duplicates, upper-triangle and out-of-range entries) through
    the reference's own sls.f90 + the gsls arms, with PERM / AMD / own ordering, scalings -1 / -2 and refinement, against
    numpy's dense solve (solution, inertia, rank)."""
    import subprocess
    import sys
    _need_dropin()
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "tools", "soak_facade.py"), "60", "5"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "0 failures" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
