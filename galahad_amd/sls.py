"""Host-side mirror of GALAHAD's SLS interface for the 'gsls' (MI355X) backend.

The production host for this path is Fortran (galahad_amd/fortran/gsls_iface.f90 + the CASE arms of
INTEGRATION.md); this module restates the same façade in Python so the parity tests read like the
reference's own (src/sls/slst.f90, src/sls/C/slst.c): same routine names, argument meaning, status
codes.  It contains host integer/bookkeeping logic only -- every factorize/solve goes through the
C ABI (include/gsls.h) into the HIP library; there is no numeric fallback.

Reference behaviour mirrored (paths relative to the GALAHAD tree):
  SLS_initialize / SLS_initialize_solver   src/sls/sls.f90:817-921, 959-1062
  SLS_analyse + SLS_coord_to_sorted_csr    src/sls/sls.f90:2178-3517, 8409-8578
  SLS_factorize (value scatter via MAPS)   src/sls/sls.f90:3521-4688 (scatter :4106-4150)
  SLS_solve = SLS_solve_ir / _ir_multiple  src/sls/sls.f90:4692-5270
  SLS_part_solve / enquire / alter_d       src/sls/sls.f90:6551-7220, 6175-6547
  status mapping                           src/sls/sls.f90:1737-1786
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Inform, Options, lib

# GALAHAD status codes (src/symbols/symbols.f90:50-116)
GALAHAD_ok = 0
GALAHAD_error_allocate = -1
GALAHAD_error_deallocate = -2
GALAHAD_error_restrictions = -3
GALAHAD_error_inertia = -20
GALAHAD_error_unknown_solver = -26
GALAHAD_unavailable_option = -29
GALAHAD_error_call_order = -31
GALAHAD_error_permutation = -39
GALAHAD_error_technical = -50


def _status_from_flag(flag):
    """SLS_copy_inform_from_gsls (integration/patch_sls.py) = SLS_copy_inform_from_ssids,
    src/sls/sls.f90:1747-1780, except that -6 'not positive definite' maps to GALAHAD_error_inertia
    like the MA57/MA97/SYTR arms do (TRS steers on that code, src/trs/trs.f90:1957, 2287); the ssids
    arm's -6 -> GALAHAD_error_restrictions makes TRS unusable with it."""
    if flag >= 0:
        return GALAHAD_ok
    if flag == -30:
        return GALAHAD_error_allocate
    if flag == -31:
        return GALAHAD_error_deallocate
    if flag in (-1, -2, -3, -4, -5, -9, -10, -12, -13, -14, -15):
        return GALAHAD_error_restrictions
    if flag == -11:
        return GALAHAD_error_permutation
    if flag in (-6, -7, -8):
        return GALAHAD_error_inertia
    if flag in (-32, GALAHAD_unavailable_option):
        return GALAHAD_unavailable_option
    return GALAHAD_error_technical


class SMT:
    """GALAHAD's SMT_type / ZD11_type (src/zd11/zd11.f90:1-55): lower triangle, 1-based indices."""

    def __init__(self, n, type="COORDINATE", row=None, col=None, ptr=None, val=None):
        self.n = int(n)
        self.type = type.upper()
        as_i = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
        self.row, self.col, self.ptr = as_i(row), as_i(col), as_i(ptr)
        self.val = None if val is None else np.ascontiguousarray(val, dtype=np.float64)
        if self.type == "COORDINATE":
            self.ne = 0 if self.row is None else len(self.row)
        elif self.type == "SPARSE_BY_ROWS":
            self.ne = int(self.ptr[self.n]) - 1
        elif self.type == "DENSE":
            self.ne = self.n * (self.n + 1) // 2
        else:
            self.ne = -1

    def coordinates(self):
        """(row, col) of every stored entry, in storage order (sls.f90:2863-2905)"""
        if self.type == "COORDINATE":
            return self.row[: self.ne], self.col[: self.ne]
        if self.type == "SPARSE_BY_ROWS":
            cnt = np.diff(self.ptr[: self.n + 1])
            return np.repeat(np.arange(1, self.n + 1, dtype=np.int32), cnt), self.col[: self.ne]
        i, j = np.tril_indices(self.n)   # packed by rows: (1,1),(2,1),(2,2),...
        return (i + 1).astype(np.int32), (j + 1).astype(np.int32)


class Control:
    """SLS_control_type fields this path reads (src/sls/sls.f90:129-357), same defaults."""

    def __init__(self):
        self.ordering = 0
        self.scaling = 0
        self.pivot_control = 1
        self.relative_pivot_tolerance = 0.01
        self.absolute_pivot_tolerance = 2.220446049250313e-16
        self.node_amalgamation = 32   # ssids nemin default; SLS passes control%node_amalgamation
        self.max_iterative_refinements = 0
        self.acceptable_residual_relative = 2.220446049250313e-15
        self.acceptable_residual_absolute = 2.220446049250313e-15
        self.print_level = 0


class InformSLS:
    """SLS_inform_type fields filled on this path (src/sls/sls.f90:435-707)."""

    def __init__(self):
        self.status = GALAHAD_ok
        self.alloc_status = 0
        self.entries = 0
        self.duplicates = 0
        self.out_of_range = 0
        self.upper = 0
        self.missing_diagonals = 0
        self.entries_in_factors = -1
        self.flops_elimination = -1
        self.rank = -1
        self.negative_eigenvalues = -1
        self.two_by_two_pivots = -1
        self.delayed_pivots = -1
        self.max_front_size = -1
        self.max_depth_assembly_tree = -1
        self.iterative_refinements = 0
        self.gsls_inform = None


class SLS:
    """data + routines; one instance plays the role of SLS_data_type."""

    SOLVERS = ("gsls",)

    def __init__(self):
        self.handle = C.c_void_p()
        self.solver = None
        self.n = 0
        self.must_be_definite = False
        self.opts = Options()
        lib.gsls_default_options(C.byref(self.opts))
        self.MAPS = None
        self.PTR = None
        self.COL = None
        self.ORDER = None
        self.nz = 0

    # -- SLS_initialize -------------------------------------------------------------------------
    def initialize(self, solver, control, inform):
        inform.status = GALAHAD_ok
        name = solver.strip().lower()
        if name not in self.SOLVERS:
            inform.status = GALAHAD_error_unknown_solver
            return
        self.solver = name
        if not self.handle:
            lib.gsls_create(C.byref(self.handle))
        self.must_be_definite = False
        # the solver-specific defaults of SLS_initialize (sls.f90:887-892 for ssids): own ordering, no
        # scaling, and wider supernodes than the CPU default of 32 (fewer, fatter fronts suit the GPU)
        control.ordering = -1
        control.scaling = 0
        control.node_amalgamation = 0       # the backend chooses from the tree it finds (gsls_analyse: 24 or 64)

    # -- SLS_coord_to_sorted_csr (sls.f90:8409-8578) -----------------------------------------------
    @staticmethod
    def _coord_to_sorted_csr(n, row, col, inform):
        row = row.astype(np.int64)
        col = col.astype(np.int64)
        ne = len(row)
        inr = (row >= 1) & (row <= n) & (col >= 1) & (col <= n)
        inform.out_of_range = int(ne - inr.sum())
        inform.upper = int((inr & (row < col)).sum())
        lo = np.minimum(row, col)   # "row" of the upper-by-rows storage
        hi = np.maximum(row, col)
        idx = np.nonzero(inr)[0]
        key = lo[idx] * (n + 1) + hi[idx]
        order = np.argsort(key, kind="stable")          # ties keep input order: first wins
        sidx, skey = idx[order], key[order]
        first = np.ones(len(sidx), dtype=bool)
        first[1:] = skey[1:] != skey[:-1]
        inform.duplicates = int((~first).sum())
        # explicit diagonal for every row
        ulo, uhi = lo[sidx[first]], hi[sidx[first]]
        has_diag = np.zeros(n + 1, dtype=bool)
        has_diag[ulo[ulo == uhi]] = True
        inform.missing_diagonals = int(n - has_diag[1:].sum())
        cnt = np.bincount(ulo, minlength=n + 1)[1:] + (~has_diag[1:])
        PTR = np.ones(n + 1, dtype=np.int64)
        PTR[1:] = 1 + np.cumsum(cnt)
        # position of each unique entry: rank within its row (+1 if the diagonal had to be inserted)
        rank_in_row = np.arange(len(ulo)) - np.searchsorted(ulo, ulo, side="left")
        pos_unique = PTR[ulo - 1] + rank_in_row + (~has_diag[ulo])
        COL = np.zeros(PTR[n] - 1, dtype=np.int32)
        COL[PTR[:-1] - 1] = np.arange(1, n + 1)
        COL[pos_unique - 1] = uhi
        grp = np.cumsum(first) - 1
        MAPS = np.zeros(ne, dtype=np.int64)
        MAPS[sidx] = np.where(first, pos_unique[grp], -pos_unique[grp])
        return PTR, COL, MAPS

    # -- SLS_analyse -----------------------------------------------------------------------------
    def analyse(self, matrix, control, inform, PERM=None):
        inform.status = GALAHAD_ok
        if self.solver is None:
            inform.status = GALAHAD_error_call_order
            return
        if matrix.n < 1 or (matrix.ne < 0 and matrix.type == "COORDINATE") or matrix.type not in (
                "COORDINATE", "SPARSE_BY_ROWS", "DENSE"):
            inform.status = GALAHAD_error_restrictions
            self.n = 0
            return
        n = self.n = matrix.n
        if control.pivot_control in (2, 3):
            self.must_be_definite = True
        inform.entries = matrix.ne
        if PERM is not None:
            p = np.asarray(PERM, dtype=np.int64)
            if len(p) != n or p.min() < 1 or p.max() > n or len(np.unique(p)) != n:
                inform.status = GALAHAD_error_permutation
                return
        r, c = matrix.coordinates()
        self.PTR, self.COL, self.MAPS = self._coord_to_sorted_csr(n, r, c, inform)
        self.nz = int(self.PTR[n] - 1)
        self._copy_control(control)
        if PERM is not None:
            self.opts.ordering = 0
            self.ORDER = np.ascontiguousarray(PERM, dtype=np.int32).copy()
        else:
            # ssids would call METIS here (sls.f90:3134); gsls uses its own nested dissection for
            # control%ordering <= 0 (SBLS reaches SLS with 0).  Natural order = PERM identity.
            # control%ordering = 1, 2 (the minimum-degree orderings other solvers get from MC68): built-in AMD
            self.opts.ordering = 2 if int(control.ordering) in (1, 2) else 1
            self.ORDER = np.arange(1, n + 1, dtype=np.int32)
        ginf = Inform()
        flag = lib.gsls_analyse(self.handle, n, self.PTR.ctypes.data_as(_lib.p_i64),
                                self.COL.ctypes.data_as(_lib.p_i32),
                                self.ORDER.ctypes.data_as(_lib.p_i32), C.byref(self.opts),
                                C.byref(ginf))
        self._copy_inform(inform, ginf, flag)
        if flag >= 0:
            # the user's storage and SLS's map stay with the backend (HBM): SLS_factorize's scatter loop and the
            # refinement residual run on the device (include/gsls.h, gsls_set_coo)
            self._coo = (np.ascontiguousarray(r, dtype=np.int32), np.ascontiguousarray(c, dtype=np.int32))
            f2 = lib.gsls_set_coo(self.handle, len(self.MAPS), self._coo[0].ctypes.data_as(C.c_void_p),
                                  self._coo[1].ctypes.data_as(C.c_void_p),
                                  np.ascontiguousarray(self.MAPS, dtype=np.int32).ctypes.data_as(C.c_void_p))
            if f2 < 0:
                inform.status = _status_from_flag(f2)

    def _copy_control(self, control):
        """SLS_copy_control_to_ssids, src/sls/sls.f90:1385-1439"""
        self.opts.nemin = control.node_amalgamation
        # control%scaling = -1 / -2 / -3 -> the backend's own scalings 1 / 2 / 3 (sls.f90:1405-1413)
        self.opts.scaling = {-1: 1, -2: 2, -3: 3}.get(int(control.scaling), 0)
        self.opts.small = control.absolute_pivot_tolerance
        self.opts.print_level = control.print_level - 1
        if control.pivot_control == 2:
            self.opts.u, self.opts.action = 0.0, 1
        elif control.pivot_control == 3:
            self.opts.u, self.opts.action = 0.0, 0
        elif control.pivot_control == 4:
            self.opts.u, self.opts.action = 0.0, 1
        else:
            self.opts.u, self.opts.action = control.relative_pivot_tolerance, 1

    @staticmethod
    def _copy_inform(inform, ginf, flag):
        inform.gsls_inform = ginf.as_dict()
        inform.status = _status_from_flag(flag)
        if flag >= 0:
            inform.two_by_two_pivots = ginf.num_two
            inform.rank = ginf.matrix_rank
            inform.negative_eigenvalues = ginf.num_neg
            inform.delayed_pivots = ginf.num_delay
            inform.entries_in_factors = ginf.num_factor
            inform.flops_elimination = ginf.num_flops
            inform.max_front_size = ginf.maxfront
            inform.max_depth_assembly_tree = ginf.maxdepth

    # -- SLS_factorize ---------------------------------------------------------------------------
    def scatter_values(self, matrix):
        """VAL(k) = val(l) / VAL(-k) += val(l)  (sls.f90:4106-4150)"""
        VAL = np.zeros(self.nz, dtype=np.float64)
        m = self.MAPS
        v = matrix.val[: len(m)]
        pos = m > 0
        VAL[m[pos] - 1] = v[pos]
        neg = m < 0
        if neg.any():
            np.add.at(VAL, -m[neg] - 1, v[neg])
        return VAL

    def factorize(self, matrix, control, inform):
        inform.status = GALAHAD_ok
        if self.MAPS is None:
            inform.status = GALAHAD_error_call_order
            return
        self._copy_control(control)
        v = np.ascontiguousarray(matrix.val[: len(self.MAPS)], dtype=np.float64)
        ginf = Inform()
        flag = lib.gsls_factor_coo(self.handle, 1 if self.must_be_definite else 0,
                                   v.ctypes.data_as(C.c_void_p), None, C.byref(self.opts), C.byref(ginf))
        self._copy_inform(inform, ginf, flag)
        if flag >= 0:      # the order the factors are in (pre-ordering, order repair, learned pivots): what PERM reports
            lib.gsls_get_order(self.handle, self.ORDER.ctypes.data_as(_lib.p_i32))
        self._resid_on_device = flag >= 0

    # -- SLS_solve (with iterative refinement, sls.f90:4692-4963 / 4967-5270) -----------------------
    def _backend_solve(self, X, job, inform):
        X = np.asfortranarray(X, dtype=np.float64)
        nrhs = 1 if X.ndim == 1 else X.shape[1]
        ginf = Inform()
        flag = lib.gsls_solve(self.handle, job, nrhs, X.ctypes.data_as(C.c_void_p), self.n,
                              C.byref(self.opts), C.byref(ginf))
        inform.status = _status_from_flag(flag)
        inform.gsls_inform = ginf.as_dict()
        return X

    @staticmethod
    def _residual(matrix, B, X):
        """RES = B - A X with the symmetric COO loop of sls.f90:4826-4934"""
        r, c = matrix.coordinates()
        n = matrix.n
        ok = (np.minimum(r, c) >= 1) & (np.maximum(r, c) <= n)
        r, c, v = r[ok] - 1, c[ok] - 1, matrix.val[: len(ok)][ok]
        RES = B.copy()
        if X.ndim == 1:
            np.subtract.at(RES, r, v * X[c])
            off = r != c
            np.subtract.at(RES, c[off], v[off] * X[r[off]])
        else:
            for k in range(X.shape[1]):
                np.subtract.at(RES[:, k], r, v * X[c, k])
                off = r != c
                np.subtract.at(RES[:, k], c[off], v[off] * X[r[off], k])
        return RES

    def _residual_dev(self, matrix, B, X):
        """RES = B - A X on the device (gsls_residual) -- the matrix of the last factorization; falls back to the
        host loop when the backend does not hold it"""
        if not getattr(self, "_resid_on_device", False):
            return self._residual(matrix, B, X)
        X = np.asfortranarray(X, dtype=np.float64)
        B = np.asfortranarray(B, dtype=np.float64)
        R = np.empty_like(B, order="F")
        nrhs = 1 if X.ndim == 1 else X.shape[1]
        ginf = Inform()
        flag = lib.gsls_residual(self.handle, nrhs, X.ctypes.data_as(C.c_void_p), self.n,
                                 B.ctypes.data_as(C.c_void_p), self.n, R.ctypes.data_as(C.c_void_p), self.n,
                                 C.byref(ginf))
        if flag < 0:
            return self._residual(matrix, B, X)
        return R

    def solve(self, matrix, X, control, inform):
        """X holds b on entry, x on exit (returned)."""
        inform.status = GALAHAD_ok
        X = np.array(X, dtype=np.float64, order="F")
        if control.max_iterative_refinements <= 0:
            return self._backend_solve(X, 0, inform)
        if X.ndim == 1 and getattr(self, "_resid_on_device", False):
            # the whole refinement loop on the device (gsls_solve_ir = SLS_solve_ir, sls.f90:4770-4949)
            ginf, it = Inform(), _lib.i32(0)
            flag = lib.gsls_solve_ir(self.handle, X.ctypes.data_as(C.c_void_p), control.max_iterative_refinements,
                                     control.acceptable_residual_absolute, control.acceptable_residual_relative,
                                     C.byref(it), C.byref(self.opts), C.byref(ginf))
            inform.status = _status_from_flag(flag)
            inform.gsls_inform = ginf.as_dict()
            inform.iterative_refinements = it.value
            return X
        B = X.copy()
        RES = X.copy()
        X = np.zeros_like(B)
        residual_zero = np.abs(B).max(axis=0)
        for it in range(control.max_iterative_refinements + 1):
            inform.iterative_refinements = it
            RES = self._backend_solve(RES, 0, inform)
            if inform.status != GALAHAD_ok:
                return X
            X = X + RES
            if it < control.max_iterative_refinements:
                RES = self._residual_dev(matrix, B, X)
            residual = np.abs(RES).max(axis=0)
            if np.all(residual < np.maximum(control.acceptable_residual_absolute,
                                            control.acceptable_residual_relative * residual_zero)):
                break
        return X

    # -- SLS_part_solve (sls.f90:6551-7220): part in 'L','D','U','S' ---------------------------------
    def part_solve(self, part, X, control, inform):
        part = part.upper()
        inform.status = GALAHAD_ok
        X = np.array(X, dtype=np.float64, order="F")
        if part == "L":
            return self._backend_solve(X, 1, inform)
        if part == "U":
            return self._backend_solve(X, 3, inform)
        if part == "D":
            if self.must_be_definite:
                return X
            return self._backend_solve(X, 2, inform)
        if part == "S":
            # L sqrt(D): for a Cholesky factor that IS L (sls.f90:6837); otherwise the MA57 arm's recipe
            # (sls.f90:6635-6672): y = L^-1 x, z = D^-1 y, x_i = sign * sqrt|z_i| sqrt|y_i|, which needs
            # z_i and y_i of one sign (a positive definite D) -- anything else is GALAHAD_error_inertia
            Y = self._backend_solve(X, 1, inform)
            if self.must_be_definite or inform.status != GALAHAD_ok:
                return Y
            Z = self._backend_solve(Y.copy(), 2, inform)
            if inform.status != GALAHAD_ok:
                return Z
            if np.any((Z == 0.0) != (Y == 0.0)) or np.any(Z * Y < 0.0):
                inform.status = GALAHAD_error_inertia
                return Z
            return np.sign(Z) * np.sqrt(np.abs(Z)) * np.sqrt(np.abs(Y))
        inform.status = GALAHAD_unavailable_option
        return X

    # -- SLS_enquire (sls.f90:6175-6414) -----------------------------------------------------------
    def enquire(self, inform, want_perm=False, want_pivots=False, want_d=False):
        out = {}
        n = self.n
        ginf = Inform()
        if self.must_be_definite:
            d = np.zeros(n)
            flag = lib.gsls_enquire_posdef(self.handle, d.ctypes.data_as(_lib.p_f64), C.byref(ginf))
            inform.status = _status_from_flag(flag)
            if want_d:
                D = np.zeros((2, n), order="F")
                D[0, :] = d ** 2   # SLS returns D of L D L^T: l_ii^2 for a Cholesky factor
                out["D"] = D
            if want_perm or want_pivots:
                out["PERM"] = self.ORDER.copy()
            return out
        piv = np.zeros(n, dtype=np.int32)
        d = np.zeros((2, n), order="F")
        flag = lib.gsls_enquire_indef(self.handle, piv.ctypes.data_as(_lib.p_i32),
                                      d.ctypes.data_as(_lib.p_f64), C.byref(ginf))
        inform.status = _status_from_flag(flag)
        if want_perm or want_pivots:
            out["PIVOTS"] = piv
            # PERM = the order the factors are in (data%ORDER, refreshed from the backend after every factorization: the
            # facade arm of integration/patch_sls.py does the same); PIVOTS adds the pivoting inside the fronts
            out["PERM"] = self.ORDER.copy()
        if want_d:
            out["D"] = d
        return out

    # -- SLS_alter_d (sls.f90:6418-6547) ---------------------------------------------------------------
    def alter_d(self, D, inform):
        D = np.asfortranarray(D, dtype=np.float64)
        ginf = Inform()
        flag = lib.gsls_alter(self.handle, D.ctypes.data_as(_lib.p_f64), C.byref(ginf))
        inform.status = _status_from_flag(flag)

    # -- SLS_terminate ---------------------------------------------------------------------------------
    def terminate(self, control=None, inform=None):
        if self.handle:
            lib.gsls_destroy(C.byref(self.handle))
            self.handle = C.c_void_p()
        self.MAPS = None
        if inform is not None:
            inform.status = GALAHAD_ok

    def __del__(self):
        try:
            self.terminate()
        except Exception:
            pass

    # introspection for the parity tests ---------------------------------------------------------------
    def symbolic(self):
        nn, rl, nl = C.c_int32(), C.c_int64(), C.c_int64()
        lib.gsls_get_symbolic_sizes(self.handle, C.byref(nn), C.byref(rl), C.byref(nl))
        nn, rl, nl = nn.value, rl.value, nl.value
        sptr = np.zeros(nn + 1, dtype=np.int32)
        sparent = np.zeros(nn, dtype=np.int32)
        rptr = np.zeros(nn + 1, dtype=np.int64)
        rlist = np.zeros(max(rl, 1), dtype=np.int32)
        nptr = np.zeros(nn + 1, dtype=np.int64)
        nlist = np.zeros((max(nl, 1), 2), dtype=np.int64)
        lib.gsls_get_symbolic(self.handle, sptr.ctypes.data_as(_lib.p_i32),
                              sparent.ctypes.data_as(_lib.p_i32), rptr.ctypes.data_as(_lib.p_i64),
                              rlist.ctypes.data_as(_lib.p_i32), nptr.ctypes.data_as(_lib.p_i64),
                              nlist.ctypes.data_as(_lib.p_i64))
        return dict(nnodes=nn, sptr=sptr, sparent=sparent, rptr=rptr, rlist=rlist[:rl], nptr=nptr,
                    nlist=nlist[:nl], order=self.ORDER.copy())
