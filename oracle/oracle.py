"""TEST INFRASTRUCTURE ONLY: ctypes wrapper of oracle/libgsls_oracle.so, the plain-C CPU restatement of
the reference algorithm (oracle/gsls_oracle.c).  Importable from tests/, bench.py (cpu_baseline)
and __graft_entry__.smoke() only -- never from galahad_amd/."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libgsls_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB)
        L.oracle_analyse.restype = C.c_void_p
        L.oracle_analyse.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.oracle_free.argtypes = [C.c_void_p]
        L.oracle_factor.restype = C.c_int
        L.oracle_factor.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int]
        L.oracle_solve.restype = C.c_int
        L.oracle_solve.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_stats.argtypes = [C.c_void_p] + [C.c_void_p] * 7
        L.oracle_get_symbolic.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.oracle_rlist_len.restype = C.c_int64
        L.oracle_rlist_len.argtypes = [C.c_void_p]
        L.oracle_nlist_len.restype = C.c_int64
        L.oracle_nlist_len.argtypes = [C.c_void_p]
        L.oracle_enquire_indef.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def lower_csc(n, row, col, val):
    """What SLS hands to its backend (src/sls/sls.f90:8409-8578): lower triangle by columns, rows
    sorted, explicit diagonal, duplicates summed.  1-based ptr (int64) / row (int32)."""
    import scipy.sparse as sp
    r = np.maximum(row, col).astype(np.int64) - 1
    c = np.minimum(row, col).astype(np.int64) - 1
    r = np.concatenate([r, np.arange(n)])
    c = np.concatenate([c, np.arange(n)])
    v = np.concatenate([np.asarray(val, dtype=np.float64), np.zeros(n)])
    A = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsc()   # sums duplicates
    A.sort_indices()
    # keep explicit zeros (structural entries): coo->csc keeps them
    return (A.indptr.astype(np.int64) + 1), (A.indices.astype(np.int32) + 1), A.data.copy()


class Oracle:
    def __init__(self, n, ptr, row, order, nemin=32):
        self.n = n
        self.ptr = np.ascontiguousarray(ptr, dtype=np.int64)
        self.row = np.ascontiguousarray(row, dtype=np.int32)
        self.order = np.ascontiguousarray(order, dtype=np.int32).copy()
        flag = C.c_int(0)
        self.h = lib().oracle_analyse(n, self.ptr.ctypes.data, self.row.ctypes.data,
                                      self.order.ctypes.data, nemin, C.byref(flag))
        self.flag = flag.value
        if not self.h:
            raise RuntimeError("oracle_analyse failed flag=%d" % self.flag)

    def symbolic(self):
        st = self.stats()
        nn = st["nnodes"]
        rl, nl = lib().oracle_rlist_len(self.h), lib().oracle_nlist_len(self.h)
        sptr = np.zeros(nn + 1, np.int32); sparent = np.zeros(nn, np.int32)
        rptr = np.zeros(nn + 1, np.int64); rlist = np.zeros(max(rl, 1), np.int32)
        nptr = np.zeros(nn + 1, np.int64); nlist = np.zeros((max(nl, 1), 2), np.int64)
        lib().oracle_get_symbolic(self.h, sptr.ctypes.data, sparent.ctypes.data, rptr.ctypes.data,
                                  rlist.ctypes.data, nptr.ctypes.data, nlist.ctypes.data)
        return dict(nnodes=nn, sptr=sptr, sparent=sparent, rptr=rptr, rlist=rlist[:rl], nptr=nptr,
                    nlist=nlist[:nl], order=self.order.copy())

    def stats(self):
        a = [C.c_int64(), C.c_int64(), C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()]
        lib().oracle_stats(self.h, *[C.byref(x) for x in a])
        keys = ["num_factor", "num_flops", "nnodes", "num_neg", "num_two", "num_zero", "num_delay"]
        return {k: x.value for k, x in zip(keys, a)}

    def factor(self, val, posdef, u=0.01, small=1e-20, action=True, scaling=None):
        val = np.ascontiguousarray(val, dtype=np.float64)
        sc = None if scaling is None else np.ascontiguousarray(scaling, dtype=np.float64)
        return lib().oracle_factor(self.h, int(posdef), val.ctypes.data,
                                   None if sc is None else sc.ctypes.data, u, small, int(action))

    def solve(self, x, job=0, scaling=None):
        x = np.array(x, dtype=np.float64, order="F")
        nrhs = 1 if x.ndim == 1 else x.shape[1]
        sc = None if scaling is None else np.ascontiguousarray(scaling, dtype=np.float64)
        rc = lib().oracle_solve(self.h, job, nrhs, x.ctypes.data, self.n,
                                None if sc is None else sc.ctypes.data)
        if rc != 0:
            raise RuntimeError("oracle_solve rc=%d" % rc)
        return x

    def enquire_indef(self):
        piv = np.zeros(self.n, np.int32)
        d = np.zeros((2, self.n), order="F")
        lib().oracle_enquire_indef(self.h, piv.ctypes.data, d.ctypes.data)
        return piv, d

    def close(self):
        if self.h:
            lib().oracle_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
