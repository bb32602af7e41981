"""CPU: the C-ABI library loads, exports every symbol include/gsls.h declares, and its host-side
(integer) analyse phase reproduces the reference's symbolic factorization bit-exactly.  No numeric
kernel is launched here; on a machine without a GPU factorize must fail loudly (flag -51)."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

import problems as P

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = sorted(p for p in glob.glob(os.path.join(HERE, "golden", "*.npz")) if not os.path.basename(p).startswith("scaled_"))


def test_header_symbols_exported():
    from galahad_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "gsls.h")).read()
    declared = set(re.findall(r"\b(gsls_[a-z_]+)\s*\(", hdr))
    assert len(declared) >= 17
    raw = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), "libgsls.so does not export %s" % name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_struct_layout_matches_header():
    from galahad_amd._lib import Inform, Options
    assert C.sizeof(Options) == 8 * 4 + 4 * 8
    assert C.sizeof(Inform) == 8 * 4 + 2 * 8 + 8 * 4 + 2 * 8 + 4 * 8


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_analyse_bit_exact(path):
    from galahad_amd import SLS, SMT, Control, InformSLS
    g = np.load(path)
    n = int(g["n"])
    m = SMT(n, "COORDINATE", row=g["row"], col=g["col"], val=g["val"])
    s, c, i = SLS(), Control(), InformSLS()
    s.initialize("gsls", c, i)
    assert i.status == 0
    c.node_amalgamation = int(g["nemin"])
    s.analyse(m, c, i, PERM=g["perm"])
    assert i.status == 0
    sym = s.symbolic()
    assert sym["nnodes"] == int(g["ref_nnodes"])
    for k in ("sptr", "sparent", "rptr", "rlist", "order", "nptr", "nlist"):
        assert np.array_equal(sym[k], g["ref_" + k]), k
    assert i.entries_in_factors == int(g["ref_num_factor"])
    assert i.flops_elimination == int(g["ref_num_flops"])
    assert i.max_depth_assembly_tree == int(g["ref_max_depth"])
    s.terminate()


def test_storage_types_give_same_structure():
    """COORDINATE / SPARSE_BY_ROWS / DENSE of the 5x5 KAT (src/sls/slst.f90:29-36) analyse alike."""
    from galahad_amd import SLS, SMT, Control, InformSLS
    n, row, col, val, rhs, xs = P.kat_indefinite()
    ptr = np.array([1, 2, 4, 6, 7, 8])
    dense = np.array([2.0, 3.0, 0.0, 0.0, 4.0, 1.0, 0.0, 0.0, 5.0, 0.0, 0.0, 6.0, 0.0, 0.0, 1.0])
    mats = [SMT(n, "COORDINATE", row=row, col=col, val=val),
            SMT(n, "SPARSE_BY_ROWS", ptr=ptr, col=col, val=val),
            SMT(n, "DENSE", val=dense)]
    nfact = []
    for m in mats:
        s, c, i = SLS(), Control(), InformSLS()
        s.initialize("gsls", c, i)
        s.analyse(m, c, i, PERM=np.arange(1, n + 1))
        assert i.status == 0
        nfact.append(i.entries_in_factors)
        VAL = s.scatter_values(m)
        A = np.zeros((n, n))
        for j in range(n):
            for k in range(s.PTR[j] - 1, s.PTR[j + 1] - 1):
                A[s.COL[k] - 1, j] = VAL[k]
        ref = np.zeros((n, n))
        ref[np.maximum(row, col) - 1, np.minimum(row, col) - 1] = val   # (2,5) is an upper entry
        assert np.array_equal(A, ref)
        s.terminate()
    assert nfact[0] == nfact[1] == 15      # dense pattern has explicit zeros but the same single front


def test_coord_map_duplicates_out_of_range_missing_diag():
    """SLS_coord_to_sorted_csr semantics (src/sls/sls.f90:8409-8578): k>0 place, k<0 add, 0 out of
    range; upper entries are mirrored; missing diagonals are inserted."""
    from galahad_amd import SLS, SMT, Control, InformSLS
    n = 4
    row = np.array([1, 2, 2, 1, 4, 9, 3, 3], dtype=np.int32)
    col = np.array([1, 1, 1, 2, 3, 1, 4, 0], dtype=np.int32)   # dup (2,1) thrice (one as upper), oor x2
    val = np.array([5.0, 1.0, 2.0, 4.0, 7.0, 100.0, 1.5, 100.0])
    m = SMT(n, "COORDINATE", row=row, col=col, val=val)
    s, c, i = SLS(), Control(), InformSLS()
    s.initialize("gsls", c, i)
    s.analyse(m, c, i, PERM=np.arange(1, n + 1))
    assert i.status == 0
    assert i.out_of_range == 2 and i.duplicates == 3 and i.upper == 2 and i.missing_diagonals == 3
    assert list(s.MAPS == 0) == [False] * 5 + [True, False, True]
    VAL = s.scatter_values(m)
    A = np.zeros((n, n))
    for j in range(n):
        for k in range(s.PTR[j] - 1, s.PTR[j + 1] - 1):
            A[s.COL[k] - 1, j] = VAL[k]
    expect = np.zeros((n, n))
    expect[0, 0] = 5.0
    expect[1, 0] = 7.0
    expect[3, 2] = 8.5
    assert np.array_equal(A, expect)
    assert [s.COL[s.PTR[j] - 1] for j in range(n)] == [1, 2, 3, 4]   # explicit diagonal first
    s.terminate()


def test_error_behaviour():
    from galahad_amd import SLS, SMT, Control, InformSLS
    from galahad_amd import sls as S
    n, row, col, val, rhs, xs = P.kat_indefinite()
    m = SMT(n, "COORDINATE", row=row, col=col, val=val)
    s, c, i = SLS(), Control(), InformSLS()
    s.initialize("nosuchsolver", c, i)
    assert i.status == S.GALAHAD_error_unknown_solver          # sls.f90:1058
    s.initialize("gsls", c, i)
    s.factorize(m, c, i)
    assert i.status == S.GALAHAD_error_call_order
    s.analyse(m, c, i, PERM=np.array([1, 2, 2, 4, 5]))
    assert i.status == S.GALAHAD_error_permutation             # sls.f90:2243-2258
    s.analyse(SMT(0, "COORDINATE", row=[], col=[], val=[]), c, i)
    assert i.status == S.GALAHAD_error_restrictions            # sls.f90:2217
    s.terminate()


def test_factor_fails_loudly_without_gpu(have_gpu):
    if have_gpu:
        pytest.skip("GPU present")
    from galahad_amd import SLS, SMT, Control, InformSLS
    from galahad_amd import sls as S
    n, row, col, val, rhs, xs = P.kat_definite()
    m = SMT(n, "COORDINATE", row=row, col=col, val=val)
    s, c, i = SLS(), Control(), InformSLS()
    s.initialize("gsls", c, i)
    s.analyse(m, c, i, PERM=np.arange(1, n + 1))
    s.factorize(m, c, i)
    assert i.gsls_inform["flag"] == -51                        # GSLS_ERROR_HIP: no silent CPU path
    assert i.status == S.GALAHAD_error_technical
    s.terminate()


def test_raw_abi_argument_checks():
    from galahad_amd._lib import Inform, Options, lib
    h = C.c_void_p()
    assert lib.gsls_create(C.byref(h)) == 0
    inf, opt = Inform(), Options()
    lib.gsls_default_options(C.byref(opt))
    assert opt.nemin == 32 and opt.u == 0.01 and opt.small == 1e-20 and opt.action == 1
    assert lib.gsls_factor(h, 1, None, None, C.byref(opt), C.byref(inf)) == -1   # call sequence
    ptr = np.array([1, 2, 1], dtype=np.int64)
    row = np.array([1, 2], dtype=np.int32)
    order = np.array([1, 2], dtype=np.int32)
    opt.ordering = 0
    f = lib.gsls_analyse(h, 2, ptr.ctypes.data_as(C.POINTER(C.c_int64)),
                         row.ctypes.data_as(C.POINTER(C.c_int32)),
                         order.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(opt), C.byref(inf))
    assert f == -3                                                               # SSIDS_ERROR_A_PTR
    assert lib.gsls_analyse(h, -1, None, None, None, C.byref(opt), C.byref(inf)) == -2
    assert lib.gsls_destroy(C.byref(h)) == 0 and not h.value
