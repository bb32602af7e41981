"""The backend's own scalings (gsls_options.scaling = 1 Hungarian, 2 auction, 4 norm equilibration; SLS reaches 1..3 with
control%scaling = -1..-3, sls.f90:1405-1413) as host logic, no GPU: the defining properties of each algorithm of
src/spral/scaling.f90 on seeded matrices.  The end-to-end comparison with the reference run under the same
control%scaling is tests/test_gpu_parity.py::test_reference_scalings (fixtures tests/golden/scaled_*.npz)."""
import ctypes as C

import numpy as np
import pytest

import problems as P
from galahad_amd._lib import lib
from oracle.oracle import lower_csc


def scale(kind, n, row, col, val, action=1):
    ptr, r, v = lower_csc(n, row, col, val)
    ptr = np.ascontiguousarray(ptr, dtype=np.int64)
    r = np.ascontiguousarray(r, dtype=np.int32)
    v = np.ascontiguousarray(v, dtype=np.float64)
    s = np.zeros(n)
    flag = lib.gsls_scale_sym(kind, n, ptr.ctypes.data_as(C.POINTER(C.c_int64)), r.ctypes.data_as(C.POINTER(C.c_int32)),
                              v.ctypes.data_as(C.POINTER(C.c_double)), action, s.ctypes.data_as(C.POINTER(C.c_double)))
    return flag, s


def badly_scaled(prob, seed, decades=4.0):
    n, row, col, val, rhs, xs = prob
    d = 10.0 ** np.random.default_rng(seed).uniform(-decades, decades, n)
    val2 = val * d[row - 1] * d[col - 1]
    return (n, row, col, val2, P.sym_matvec(n, row - 1, col - 1, val2, xs), xs), d


CASES = {
    "kkt": lambda: badly_scaled(P.kkt_qpband(300, 60), 1)[0],
    "grid_indef": lambda: badly_scaled(P.grid2d(15, 14, shift=1.0), 2)[0],
    "rand_indef": lambda: badly_scaled(P.random_sparse(400, 5, 7, spd=False), 3)[0],
    "band_spd": lambda: badly_scaled(P.banded_spd(300, 7), 4)[0],
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_hungarian_scaling_is_a_maximum_product_matching_scaling(name):
    """|s_i a_ij s_j| <= 1 for every entry, and a perfect matching of entries with modulus exactly 1 (one per row
    and column): the optimality conditions of the assignment problem MC64 solves (scaling.f90:597-693)."""
    n, row, col, val, rhs, xs = CASES[name]()
    flag, s = scale(1, n, row, col, val)
    assert flag == 0 and np.all(s > 0) and np.all(np.isfinite(s))
    sv = np.abs(val) * s[row - 1] * s[col - 1]
    assert sv.max() <= 1.0 + 1e-10
    tight = sv >= 1.0 - 1e-9
    # the tight entries must contain a perfect matching of the full symmetric pattern
    import scipy.sparse as sp
    from scipy.sparse.csgraph import maximum_bipartite_matching
    r, c = row[tight] - 1, col[tight] - 1
    G = sp.csr_matrix((np.ones(2 * len(r)), (np.r_[r, c], np.r_[c, r])), shape=(n, n))
    assert (maximum_bipartite_matching(G, perm_type="column") >= 0).all()


@pytest.mark.parametrize("name", sorted(CASES))
def test_auction_and_equilibration_bring_every_row_close_to_one(name):
    """(ten iterations of the norm equilibration -- the reference's limit, scaling.f90:46-49 -- leave a few per cent)"""
    n, row, col, val, rhs, xs = CASES[name]()
    for kind, lo, hi in ((2, 1e-3, 1e3), (4, 0.9, 1.0 + 1e-9)):
        flag, s = scale(kind, n, row, col, val)
        assert flag == 0 and np.all(s > 0) and np.all(np.isfinite(s))
        sv = np.abs(val) * s[row - 1] * s[col - 1]
        rowmax = np.zeros(n)
        np.maximum.at(rowmax, row - 1, sv)
        np.maximum.at(rowmax, col - 1, sv)
        # unscaled, the row maxima of these matrices span 16 decades
        assert rowmax.min() >= lo and rowmax.max() <= hi, (kind, rowmax.min(), rowmax.max())


def test_structurally_singular_matrix():
    """an empty row/column: error with action = false, the Duff-Pralet completion with action = true
    (scaling.f90:669-800; ssids.f90:944-947)"""
    n = 6
    row = np.array([1, 2, 2, 3, 5, 5, 6], dtype=np.int32)
    col = np.array([1, 1, 2, 3, 3, 5, 6], dtype=np.int32)       # variable 4 has no entry at all
    val = np.array([4.0, 1.0, 3.0, 1e4, 2e-3, 5.0, 1e-6])
    flag, s = scale(1, n, row, col, val, action=0)
    assert flag == -5 and np.all(s == 1.0)
    flag, s = scale(1, n, row, col, val, action=1)
    assert flag == 1 and np.all(np.isfinite(s)) and s[3] == 1.0
    sv = np.abs(val) * s[row - 1] * s[col - 1]
    assert sv.max() <= 1.0 + 1e-10


def test_scaling_3_needs_the_matching_ordering():
    from galahad_amd import SLS, SMT, Control, InformSLS
    # (no device needed: the option is rejected before any device work -- but analyse needs none either)
    n, row, col, val, rhs, xs = P.kat_indefinite()
    assert lib.gsls_scale_sym(3, n, None, None, None, 1, None) < 0


def _analyse_matching(n, row, col, val, ordering=1):
    from galahad_amd._lib import Options, Inform
    ptr, r, v = lower_csc(n, row, col, val)
    ptr = np.ascontiguousarray(ptr, dtype=np.int64)               # (1-based, explicit diagonal: what SLS hands over)
    r = np.ascontiguousarray(r, dtype=np.int32)
    v = np.ascontiguousarray(v, dtype=np.float64)
    h = C.c_void_p()
    assert lib.gsls_create(C.byref(h)) == 0
    o, i = Options(), Inform()
    lib.gsls_default_options(C.byref(o))
    o.ordering = ordering
    order = np.zeros(n, dtype=np.int32)
    flag = lib.gsls_analyse_matching(h, n, ptr.ctypes.data_as(C.POINTER(C.c_int64)), r.ctypes.data_as(C.POINTER(C.c_int32)),
                                     v.ctypes.data_as(C.c_void_p), order.ctypes.data_as(C.POINTER(C.c_int32)),
                                     C.byref(o), C.byref(i))
    return h, flag, order, i, (ptr, r, v)


def test_matching_based_ordering_puts_partners_next_to_each_other():
    """gsls_analyse_matching (ssids_analyse with val and ordering = 2, ssids.f90:305-320; spral/match_order.f90): on
    K = [eps I, B; B^T, 0] with B = diag-dominant the maximum-product matching pairs variable i with n1 + i (the only large
    entry of either), so the two must be consecutive in the returned order -- the 2x2 pivots threshold pivoting takes; the
    order is a permutation, and the analysis that follows reports the pattern's statistics as gsls_analyse does."""
    rng = np.random.default_rng(5)
    n1 = 120
    rows, cols, vals = [], [], []
    for i in range(n1):
        rows.append(i + 1); cols.append(i + 1); vals.append(1e-8 * (1 + i % 3))            # tiny (1,1) block
        rows.append(n1 + i + 1); cols.append(i + 1); vals.append(10.0 + rng.uniform())       # B's diagonal: the large entries
        for _ in range(2):                                                                   # small couplings elsewhere
            j = int(rng.integers(0, n1))
            if j != i:
                rows.append(n1 + j + 1); cols.append(i + 1); vals.append(0.01 * rng.uniform(-1, 1))
    n = 2 * n1
    row, col, val = np.array(rows, np.int32), np.array(cols, np.int32), np.array(vals)
    for ordering in (1, 2, 3):
        h, flag, order, inf, _ = _analyse_matching(n, row, col, val, ordering)
        assert flag == 0, flag
        assert sorted(order.tolist()) == list(range(1, n + 1))
        # (the order that comes back is the FINAL pivot order, as from ssids_analyse: supernode amalgamation renumbers
        # inside a merged supernode -- children by decreasing column count, core_analyse.f90:806-822 -- which parts a
        # pair now and then, in the reference as here; both halves stay in one front)
        gap = np.array([abs(int(order[i]) - int(order[n1 + i])) for i in range(n1)])
        assert (gap == 1).sum() >= 0.8 * n1 and gap.max() <= 40, (ordering, (gap == 1).sum(), gap.max())
        assert inf.num_factor > 0 and inf.num_sup > 0
        lib.gsls_destroy(C.byref(h))


@pytest.mark.parametrize("name", sorted(CASES))
def test_matching_based_ordering_keeps_every_pair_an_entry_of_the_matrix(name):
    """Whatever the cycles of the matching look like (long cycles are cut into pairs + singletons, mo_split
    match_order.f90:220-330), two variables that end up as a pair -- consecutive positions (2k-1, 2k) that were not
    forced by the graph -- must be joined by an entry; checked through the property the expansion guarantees: walking the
    order, every variable whose matched partner is a different variable has that partner or a neighbour of the cycle
    beside it.  Here simply: the order is a permutation for every test matrix, singular ones included, and the scaling the
    call saves is the Hungarian scaling (same matching)."""
    n, row, col, val, rhs, xs = CASES[name]()
    h, flag, order, inf, _ = _analyse_matching(n, row, col, val)
    assert flag >= 0
    assert sorted(order.tolist()) == list(range(1, n + 1))
    lib.gsls_destroy(C.byref(h))


def test_matching_based_ordering_argument_checks():
    from galahad_amd._lib import Options, Inform
    n, row, col, val, rhs, xs = P.kat_indefinite()
    ptr, r, v = lower_csc(n, row, col, val)
    ptr = np.ascontiguousarray(ptr, dtype=np.int64)
    r = np.ascontiguousarray(r, dtype=np.int32)
    h = C.c_void_p()
    assert lib.gsls_create(C.byref(h)) == 0
    order = np.zeros(n, dtype=np.int32)
    i = Inform()
    p64, p32 = C.POINTER(C.c_int64), C.POINTER(C.c_int32)
    # no values: the error ssids_analyse gives for ordering = 2 without val (ssids.f90:306-310)
    assert lib.gsls_analyse_matching(h, n, ptr.ctypes.data_as(p64), r.ctypes.data_as(p32), None,
                                     order.ctypes.data_as(p32), None, C.byref(i)) == -9
    assert lib.gsls_analyse_matching(h, n, ptr.ctypes.data_as(p64), r.ctypes.data_as(p32),
                                     np.ascontiguousarray(v).ctypes.data_as(C.c_void_p), None, None, C.byref(i)) < 0
    assert lib.gsls_analyse_matching(None, n, ptr.ctypes.data_as(p64), r.ctypes.data_as(p32),
                                     np.ascontiguousarray(v).ctypes.data_as(C.c_void_p), order.ctypes.data_as(p32), None,
                                     C.byref(i)) < 0
    lib.gsls_destroy(C.byref(h))
