"""Generates tests/golden/*.npz from the REAL reference (GALAHAD SLS + SPRAL SSIDS CPU) built by
oracle/build_ref.sh from /root/reference.  Run in the build container only:

    python tests/golden/make_golden.py

Each fixture is data only: the inputs (lower-triangle COO, rhs, PERM, options) and what the reference
returned for them (symbolic arrays of akeep, statistics, inertia, solution).  The reference's own
known-answer systems (src/sls/slst.f90:29-51) are included so that their stated solution x = 1..5
pins the fixtures themselves.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems as P  # noqa: E402
from oracle import refio  # noqa: E402


def case(name, prob, posdef, perm=None, nemin=32, nrhs=1):
    n, row, col, val, rhs, xs = prob
    if perm is None:
        perm = np.arange(1, n + 1)
    perm = np.asarray(perm, dtype=np.int32)
    if nrhs > 1:
        rng = np.random.default_rng(99)
        X = np.column_stack([xs] + [rng.uniform(-1, 1, n) for _ in range(nrhs - 1)])
        rhs = np.column_stack([P.sym_matvec(n, row - 1, col - 1, val, X[:, k]) for k in range(nrhs)])
        xs = X
    ref = refio.run(n, row, col, val, rhs, perm=perm, nemin=nemin, dump_struct=True,
                    pivot_control=2 if posdef else 1, threads=1)
    assert ref["status_analyse"] == 0 and ref["status_factorize"] == 0 and ref["status_solve"] == 0, ref
    out = dict(n=n, row=row, col=col, val=val, rhs=rhs, xstar=xs, perm=perm, nemin=nemin,
               posdef=int(posdef), ref_x=ref["x"], ref_neg=ref["negative_eigenvalues"],
               ref_two=ref["two_by_two"], ref_delayed=ref["delayed"], ref_rank=ref["rank"],
               ref_num_factor=ref["num_factor"], ref_num_flops=ref["num_flops"],
               ref_nnodes=ref["nnodes"], ref_sptr=ref["sptr"], ref_sparent=ref["sparent"],
               ref_rptr=ref["rptr"], ref_rlist=ref["rlist"], ref_order=ref["order"],
               ref_nptr=ref["nptr"], ref_nlist=ref["nlist"], ref_max_front=ref["max_front"],
               ref_max_depth=ref["max_depth"])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("%-22s n=%-6d nnodes=%-5d nfact=%-9d neg=%-5d two=%-4d delay=%-3d err=%.2e" % (
        name, n, ref["nnodes"], ref["num_factor"], ref["negative_eigenvalues"], ref["two_by_two"],
        ref["delayed"], np.abs(ref["x"] - xs).max()))


if __name__ == "__main__" and "scaled" not in sys.argv[1:]:
    rng = np.random.default_rng(20240110)
    case("kat_indef_natural", P.kat_indefinite(), False)
    case("kat_indef_reverse", P.kat_indefinite(), False, perm=np.arange(5, 0, -1))   # slst.f90:54-56
    case("kat_def_natural", P.kat_definite(), True)
    case("kat_def_reverse", P.kat_definite(), True, perm=np.arange(5, 0, -1))
    case("kat_def_as_indef", P.kat_definite(), False)
    case("band_n400_bw9", P.banded_spd(400, 9), True)
    case("band_n1500_bw40", P.banded_spd(1500, 40), True, nrhs=3)
    case("band_n600_bw20_perm", P.banded_spd(600, 20), True, perm=rng.permutation(600) + 1, nemin=8)
    case("grid2d_14_spd", P.grid2d(14, 14), True, perm=rng.permutation(196) + 1)
    case("grid2d_16_indef", P.grid2d(16, 16, shift=1.0), False, perm=rng.permutation(256) + 1)
    case("grid3d_7_spd", P.grid3d(7, 7, 7), True, perm=rng.permutation(343) + 1, nemin=16)
    case("kkt_300_60_natural", P.kkt_qpband(300, 60), False)
    case("kkt_300_60_perm", P.kkt_qpband(300, 60), False, perm=rng.permutation(360) + 1)
    case("rand_spd_700", P.random_sparse(700, 4, 11, spd=True), True, perm=rng.permutation(700) + 1, nemin=4)
    case("rand_indef_500", P.random_sparse(500, 5, 7, spd=False), False, perm=rng.permutation(500) + 1, nemin=8)
    case("diag_only_50", (50, np.arange(1, 51, dtype=np.int32), np.arange(1, 51, dtype=np.int32),
                          np.linspace(1, 5, 50), np.linspace(1, 5, 50) * 2.0, np.full(50, 2.0)), True)


def scaled_case(name, prob, scaling, posdef=False, nemin=32):
    """badly scaled system through the reference with control%scaling = -1 (Hungarian) / -2 (auction): what SLS's
    ssids arm does with it (sls.f90:1405-1413 -> ssids.f90:921-990)"""
    n, row, col, val, rhs, xs = prob
    perm = np.arange(1, n + 1, dtype=np.int32)
    ref = refio.run(n, row, col, val, rhs, perm=perm, nemin=nemin, scaling=scaling,
                    pivot_control=2 if posdef else 1, threads=1)
    plain = refio.run(n, row, col, val, rhs, perm=perm, nemin=nemin, pivot_control=2 if posdef else 1, threads=1)
    assert ref["status_analyse"] == 0 and ref["status_factorize"] == 0 and ref["status_solve"] == 0, ref
    out = dict(n=n, row=row, col=col, val=val, rhs=rhs, xstar=xs, perm=perm, nemin=nemin, posdef=int(posdef),
               scaling=scaling, ref_x=ref["x"], ref_neg=ref["negative_eigenvalues"], ref_rank=ref["rank"],
               ref_two=ref["two_by_two"], ref_delayed=ref["delayed"], ref_delayed_unscaled=plain["delayed"],
               ref_x_unscaled=plain["x"])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("%-26s n=%-6d scaling=%d neg=%-5d two=%-4d delayed=%-4d (unscaled run: %d) err=%.2e (unscaled %.2e)" % (
        name, n, scaling, ref["negative_eigenvalues"], ref["two_by_two"], ref["delayed"], plain["delayed"],
        np.abs(ref["x"] - xs).max(), np.abs(plain["x"] - xs).max()))


def badly_scaled(prob, seed, decades=4.0):
    n, row, col, val, rhs, xs = prob
    d = 10.0 ** np.random.default_rng(seed).uniform(-decades, decades, n)
    val2 = val * d[row - 1] * d[col - 1]
    return (n, row, col, val2, P.sym_matvec(n, row - 1, col - 1, val2, xs), xs)


if __name__ == "__main__" and "scaled" in sys.argv[1:]:
    for sc, tag in ((-1, "hungarian"), (-2, "auction")):
        scaled_case("scaled_kkt_%s" % tag, badly_scaled(P.kkt_qpband(300, 60), 1), sc)
        scaled_case("scaled_grid_indef_%s" % tag, badly_scaled(P.grid2d(15, 14, shift=1.0), 2), sc)
        scaled_case("scaled_rand_indef_%s" % tag, badly_scaled(P.random_sparse(400, 5, 7, spd=False), 3), sc, nemin=8)
