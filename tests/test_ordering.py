"""The built-in orderings as host logic (no GPU): approximate minimum degree (gsls_options.ordering = 2; what SLS offers
as control%ordering = 1 through MC68, a stub in the reference tree) against nested dissection and the natural order --
valid permutations, and on irregular patterns markedly less fill than the natural order."""
import ctypes as C

import numpy as np
import pytest

import problems as P
from galahad_amd._lib import Inform, Options, lib
from oracle.oracle import lower_csc


def analyse(n, row, col, val, ordering, nemin=None):
    ptr, r, _ = lower_csc(n, row, col, val)
    ptr = np.ascontiguousarray(ptr, dtype=np.int64)
    r = np.ascontiguousarray(r, dtype=np.int32)
    h = C.c_void_p()
    assert lib.gsls_create(C.byref(h)) == 0
    o, inf = Options(), Inform()
    lib.gsls_default_options(C.byref(o))
    o.ordering = ordering
    if nemin is not None:
        o.nemin = nemin
    order = np.arange(1, n + 1, dtype=np.int32)
    flag = lib.gsls_analyse(h, n, ptr.ctypes.data_as(C.POINTER(C.c_int64)), r.ctypes.data_as(C.POINTER(C.c_int32)),
                            order.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(o), C.byref(inf))
    assert flag >= 0, flag
    perm = np.zeros(n, dtype=np.int32)
    assert lib.gsls_get_order(h, perm.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    lib.gsls_destroy(C.byref(h))
    return perm, inf.num_factor, inf.num_flops


CASES = {
    "rand_5000": lambda: P.random_sparse(5000, 4, 21, spd=True),
    "grid2d_60": lambda: P.grid2d(60, 60),
    "grid3d_27pt_14": lambda: P.grid3d_27pt_perturbed(14, 14, 14),
    "kkt": lambda: P.kkt_qpband(3000, 600),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_amd_is_a_permutation_and_reduces_fill(name):
    n, row, col, val, rhs, xs = CASES[name]()
    fill = {}
    for tag, o in (("natural", 3), ("nd", 1), ("amd", 2)):
        perm, nfac, nflop = analyse(n, row, col, val, o)
        assert sorted(perm.tolist()) == list(range(1, n + 1)), tag
        fill[tag] = nfac
    assert fill["amd"] <= fill["natural"]
    if name.startswith("rand"):
        assert fill["amd"] <= 0.6 * fill["natural"], fill       # irregular patterns: what the ordering is for
    assert fill["amd"] <= 3.0 * fill["nd"], fill                 # and never far from the dissection ordering


def test_amd_on_a_star_and_a_path():
    """degenerate graphs: a star (eliminate the leaves first: no fill) and a path (no fill in any minimum-degree order)"""
    n = 200
    hub = np.full(n - 1, n, dtype=np.int32)
    leaves = np.arange(1, n, dtype=np.int32)
    row = np.r_[np.arange(1, n + 1, dtype=np.int32), hub]
    col = np.r_[np.arange(1, n + 1, dtype=np.int32), leaves]
    perm, nfac, _ = analyse(n, row, col, np.ones(len(row)), 2, nemin=1)
    assert nfac == 2 * n - 1 and perm[n - 1] >= n - 1           # the hub goes last (or ties with the last leaf)
    i = np.arange(1, n + 1, dtype=np.int32)
    row = np.r_[i, i[1:]]
    col = np.r_[i, i[:-1]]
    perm, nfac, _ = analyse(n, row, col, np.ones(len(row)), 2, nemin=1)
    assert nfac == 2 * n - 1


@pytest.mark.parametrize("name", ["grid2d_520", "kkt_250k"])
def test_nested_dissection_is_the_same_on_one_and_on_several_threads(name, monkeypatch):
    """The default ordering shares the pieces of the dissection tree out to worker threads once the top cuts have been
    made (gsls_order.cpp: order_nested_dissection, graphs of 200 000 vertices and more).  Every piece is processed with
    scratch arrays of the worker's own and gets what it would get alone: the permutation must not depend on the number
    of threads or on the schedule (SSIDS' METIS call is deterministic too, ssids.f90:305-320)."""
    prob = P.grid2d(520, 520) if name == "grid2d_520" else P.kkt_qpband(210000, 40000)
    n, row, col, val = prob[:4]
    ref = None
    for threads in ("1", "2", "4", "7", "4"):
        monkeypatch.setenv("GSLS_ND_THREADS", threads)
        perm, nfact, nflops = analyse(n, row, col, val, 1)
        assert np.array_equal(np.sort(perm), np.arange(1, n + 1))
        if ref is None:
            ref = (perm, nfact, nflops)
        else:
            assert np.array_equal(perm, ref[0]) and (nfact, nflops) == ref[1:], threads
