"""TEST INFRASTRUCTURE ONLY (oracle/): file I/O around oracle/_ref/ref_driver, the REAL reference
(GALAHAD SLS + SPRAL SSIDS CPU) built by oracle/build_ref.sh.  Only tests/, bench.py's cpu_baseline
leg, __graft_entry__.smoke() and the fixture generator may import this; the product never does.
The binary layout is documented at the top of oracle/ref_driver.f90.
"""
import os
import struct
import sys
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DRIVER = os.path.join(HERE, "_ref", "ref_driver")
MAGIC = 1397509959
DROPIN = os.path.join(HERE, "_ref", "sls_gsls_driver")   # real SLS facade + the gsls backend
SOLVERS = {"ssids": 0, "sytr": 1, "potr": 2, "pbtr": 3, "gsls": 4}


def available():
    return os.path.exists(DRIVER) and os.access(DRIVER, os.X_OK)


def dropin_available():
    return os.path.exists(DROPIN) and os.access(DROPIN, os.X_OK)


def write_problem(path, n, row, col, val, rhs, *, solver="ssids", pivot_control=1, max_refine=0,
                  nemin=0, perm=None, repeat=1, dump_struct=False, scaling=0, ordering=-999,
                  relative_pivot_tolerance=-1.0, absolute_pivot_tolerance=-1.0):
    """row/col/perm are 1-based (Fortran) int32 arrays; rhs is (n,) or (n, nrhs) column-major."""
    row = np.ascontiguousarray(row, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float64)
    rhs = np.asarray(rhs, dtype=np.float64)
    nrhs = 1 if rhs.ndim == 1 else rhs.shape[1]
    rhs_f = np.asfortranarray(rhs.reshape(n, nrhs))
    with open(path, "wb") as f:
        f.write(struct.pack("<2i", MAGIC, 1))
        f.write(struct.pack("<12i", n, len(row), nrhs, SOLVERS[solver], pivot_control, max_refine,
                            nemin, 0 if perm is None else 1, repeat, int(dump_struct), scaling,
                            ordering))
        f.write(struct.pack("<2d", relative_pivot_tolerance, absolute_pivot_tolerance))
        f.write(row.tobytes())
        f.write(col.tobytes())
        f.write(val.tobytes())
        if perm is not None:
            f.write(np.ascontiguousarray(perm, dtype=np.int32).tobytes())
        f.write(rhs_f.tobytes(order="F"))


def read_result(path, n, nrhs, dump_struct):
    with open(path, "rb") as f:
        buf = f.read()
    o = 0

    def take(fmt, cnt=1):
        nonlocal o
        dt = np.dtype(fmt)
        a = np.frombuffer(buf, dtype=dt, count=cnt, offset=o)
        o += dt.itemsize * cnt
        return a

    st = take("<i4", 3)
    big = take("<i8", 2)
    ints = take("<i4", 8)
    tms = take("<f8", 5)
    x = take("<f8", n * nrhs).reshape(nrhs, n).T.copy()
    res = dict(status_analyse=int(st[0]), status_factorize=int(st[1]), status_solve=int(st[2]),
               entries_in_factors=int(big[0]), flops_elimination=int(big[1]),
               rank=int(ints[0]), negative_eigenvalues=int(ints[1]), two_by_two=int(ints[2]),
               delayed=int(ints[3]), max_front=int(ints[4]), max_depth=int(ints[5]),
               num_sup=int(ints[6]),
               t_analyse=float(tms[0]), t_factorize=float(tms[1]), t_solve=float(tms[2]),
               t_factorize_median=float(tms[3]), t_solve_median=float(tms[4]),
               x=x[:, 0] if nrhs == 1 else x)
    if dump_struct:
        flag, nn = (int(v) for v in take("<i4", 2))
        res["struct_flag"] = flag
        res["nnodes"] = nn
        if flag >= 0:
            res["sptr"] = take("<i4", nn + 1).copy()
            res["sparent"] = take("<i4", nn).copy()
            res["rptr"] = take("<i8", nn + 1).copy()
            res["rlist"] = take("<i4", int(res["rptr"][-1]) - 1).copy()
            res["order"] = take("<i4", n).copy()
            nf = take("<i8", 2)
            res["num_factor"], res["num_flops"] = int(nf[0]), int(nf[1])
            res["nptr"] = take("<i8", nn + 1).copy()
            nz = int(res["nptr"][-1]) - 1
            res["nlist"] = take("<i8", 2 * nz).reshape(nz, 2).copy()
    return res


BLAS_SHIM = os.path.join(HERE, "_ref", "libblas_shim.so")


def openblas_available():
    return os.path.exists(BLAS_SHIM)


def run(n, row, col, val, rhs, *, threads=None, timeout=3600, blas="vendored", **kw):
    """Run the reference on one problem; returns the dict of read_result().
    blas = "openblas": SSIDS' BLAS / LAPACK calls go to the OpenBLAS inside scipy (oracle/blas_shim.c, preloaded;
    one OpenBLAS thread per caller: SSIDS parallelises over OpenMP tasks itself)."""
    if not available():
        raise RuntimeError("oracle/_ref/ref_driver not built (run oracle/build_ref.sh)")
    rhs = np.asarray(rhs, dtype=np.float64)
    nrhs = 1 if rhs.ndim == 1 else rhs.shape[1]
    with tempfile.TemporaryDirectory(prefix="gsls_ref_") as d:
        pin, pout = os.path.join(d, "p.bin"), os.path.join(d, "r.bin")
        write_problem(pin, n, row, col, val, rhs, **kw)
        env = dict(os.environ)
        env["OMP_CANCELLATION"] = "true"          # SSIDS requirement (ssids.f90:1425-1460)
        env.setdefault("OMP_PROC_BIND", "true")
        if threads is not None:
            env["OMP_NUM_THREADS"] = str(threads)
        if blas == "openblas":
            if not openblas_available():
                raise RuntimeError("oracle/_ref/libblas_shim.so not built")
            env["LD_PRELOAD"] = BLAS_SHIM + (":" + env["LD_PRELOAD"] if env.get("LD_PRELOAD") else "")
            env["OPENBLAS_NUM_THREADS"] = "1"
        # n >~ 1e6 needs an unlimited stack for SLS's automatic arrays (sls.f90:8436)
        exe = DROPIN if kw.get("solver") == "gsls" else DRIVER
        cmd = "ulimit -s unlimited 2>/dev/null; exec '%s' '%s' '%s'" % (exe, pin, pout)
        p = subprocess.run(["bash", "-c", cmd], env=env, capture_output=True, text=True,
                           timeout=timeout)
        if p.returncode != 0 or not os.path.exists(pout):
            raise RuntimeError("ref_driver failed rc=%d\n%s\n%s" % (p.returncode, p.stdout, p.stderr))
        return read_result(pout, n, nrhs, kw.get("dump_struct", False))


SBLS_DRIVER = os.path.join(HERE, "_ref", "sbls_driver")            # reference SBLS (ssids / sytr)
SBLS_DROPIN = os.path.join(HERE, "_ref", "sbls_gsls_driver")       # reference SBLS + SLS facade + gsls


def sbls_available(dropin=False):
    p = SBLS_DROPIN if dropin else SBLS_DRIVER
    return os.path.exists(p) and os.access(p, os.X_OK)


def run_sbls(n, m, H, A, Cm, rhs, *, solver="gsls", factorization=2, repeat=1, itref_max=1,
             threads=None, timeout=3600, print_level=0, drift=False, get_norm_residual=False):
    """SBLS_form_and_factorize + SBLS_solve on K = [H A^T; A -C].  H, A, Cm = (row, col, val) with
    1-based indices (H, C lower triangles).  Returns dict(status_factorize, status_solve, sol, ...)."""
    exe = SBLS_DROPIN if solver == "gsls" else SBLS_DRIVER
    if not os.path.exists(exe):
        raise RuntimeError("%s not built" % exe)
    with tempfile.TemporaryDirectory(prefix="gsls_sbls_") as d:
        pin, pout = os.path.join(d, "p.bin"), os.path.join(d, "r.bin")
        with open(pin, "wb") as f:
            f.write(struct.pack("<2i", 1396853330, 1))
            f.write(struct.pack("<10i", n, m, len(H[0]), len(A[0]), len(Cm[0]), SOLVERS[solver],
                                factorization, repeat, itref_max,
                                print_level + (100 if drift else 0) + (1000 if get_norm_residual else 0)))
            for (r, c, v) in (H, A, Cm):
                f.write(np.ascontiguousarray(r, dtype=np.int32).tobytes())
                f.write(np.ascontiguousarray(c, dtype=np.int32).tobytes())
                f.write(np.ascontiguousarray(v, dtype=np.float64).tobytes())
            f.write(np.ascontiguousarray(rhs, dtype=np.float64).tobytes())
        env = dict(os.environ)
        env["OMP_CANCELLATION"] = "true"
        if threads is not None:
            env["OMP_NUM_THREADS"] = str(threads)
        cmd = "ulimit -s unlimited 2>/dev/null; exec '%s' '%s' '%s'" % (exe, pin, pout)
        p = subprocess.run(["bash", "-c", cmd], env=env, capture_output=True, text=True, timeout=timeout)
        if p.returncode != 0 or not os.path.exists(pout):
            raise RuntimeError("sbls driver failed rc=%d\n%s\n%s" % (p.returncode, p.stdout, p.stderr))
        if print_level:
            print(p.stdout)
        if os.environ.get("GSLS_DEBUG"):
            print(p.stderr[-20000:])
        buf = open(pout, "rb").read()
        ints = np.frombuffer(buf, dtype="<i4", count=6)
        tms = np.frombuffer(buf, dtype="<f8", count=4, offset=24)
        sol = np.frombuffer(buf, dtype="<f8", count=n + m, offset=24 + 32).copy()
        nres = float(np.frombuffer(buf, dtype="<f8", count=1, offset=24 + 32 + 8 * (n + m))[0]) \
            if len(buf) >= 24 + 32 + 8 * (n + m + 1) else None
        return dict(norm_residual=nres, status_factorize=int(ints[0]), status_solve=int(ints[1]), factorization=int(ints[2]),
                    rank=int(ints[3]), negative_eigenvalues=int(ints[4]), t_factorize=float(tms[0]),
                    t_solve=float(tms[1]), t_factorize_median=float(tms[2]), t_solve_median=float(tms[3]),
                    sol=sol)


TRS_DRIVER = os.path.join(HERE, "_ref", "trs_driver")
TRS_DROPIN = os.path.join(HERE, "_ref", "trs_gsls_driver")


def trs_available(dropin=False):
    p = TRS_DROPIN if dropin else TRS_DRIVER
    return os.path.exists(p) and os.access(p, os.X_OK)


def run_trs(n, H, c, radius, f=0.0, *, solver="gsls", mdiag=0.0, print_level=0, timeout=3600):
    """TRS_solve: min 1/2 x'Hx + c'x + f s.t. ||x||_M <= radius.  H = (row, col, val) lower, 1-based."""
    exe = TRS_DROPIN if solver == "gsls" else TRS_DRIVER
    if not os.path.exists(exe):
        raise RuntimeError("%s not built" % exe)
    with tempfile.TemporaryDirectory(prefix="gsls_trs_") as d:
        pin, pout = os.path.join(d, "p.bin"), os.path.join(d, "r.bin")
        with open(pin, "wb") as fh:
            fh.write(struct.pack("<2i", 1414681344, 1))
            fh.write(struct.pack("<4i", n, len(H[0]), SOLVERS[solver], print_level))
            fh.write(struct.pack("<3d", radius, f, mdiag))
            fh.write(np.ascontiguousarray(H[0], dtype=np.int32).tobytes())
            fh.write(np.ascontiguousarray(H[1], dtype=np.int32).tobytes())
            fh.write(np.ascontiguousarray(H[2], dtype=np.float64).tobytes())
            fh.write(np.ascontiguousarray(c, dtype=np.float64).tobytes())
        env = dict(os.environ)
        env["OMP_CANCELLATION"] = "true"
        cmd = "ulimit -s unlimited 2>/dev/null; exec '%s' '%s' '%s'" % (exe, pin, pout)
        p = subprocess.run(["bash", "-c", cmd], env=env, capture_output=True, text=True, timeout=timeout)
        if p.returncode != 0 or not os.path.exists(pout):
            raise RuntimeError("trs driver failed rc=%d\n%s\n%s" % (p.returncode, p.stdout, p.stderr))
        if print_level:
            print(p.stdout)
        if os.environ.get("GSLS_DEBUG"):
            print(p.stderr[-20000:])
        buf = open(pout, "rb").read()
        ints = np.frombuffer(buf, dtype="<i4", count=4)
        dbl = np.frombuffer(buf, dtype="<f8", count=4, offset=16)
        x = np.frombuffer(buf, dtype="<f8", count=n, offset=48).copy()
        return dict(status=int(ints[0]), factorizations=int(ints[1]), obj=float(dbl[0]),
                    multiplier=float(dbl[1]), x_norm=float(dbl[2]), time=float(dbl[3]), x=x)


CQP_DROPIN = os.path.join(HERE, "_ref", "cqp_gsls_driver")    # reference CQP + SBLS + patched SLS facade (sytr | gsls)


def cqp_available():
    return os.path.exists(CQP_DROPIN) and os.access(CQP_DROPIN, os.X_OK)


def run_cqp(n, m, H, A, g, c_l, c_u, x_l, x_u, *, solver="gsls", print_level=0, timeout=3600):
    """CQP_solve on min 1/2 x'Hx + g'x s.t. c_l <= Ax <= c_u, x_l <= x <= x_u.  H, A = (row, col, val), 1-based,
    H lower triangle.  solver: 'sytr' (the reference's dense LAPACK arm) or 'gsls'."""
    if not cqp_available():
        raise RuntimeError("oracle/_ref/cqp_gsls_driver not built")
    with tempfile.TemporaryDirectory(prefix="gsls_cqp_") as d:
        pin, pout = os.path.join(d, "p.bin"), os.path.join(d, "r.bin")
        with open(pin, "wb") as f:
            f.write(struct.pack("<2i", 1129336146, 1))
            f.write(struct.pack("<6i", n, m, len(H[0]), len(A[0]), 4 if solver == "gsls" else 1, print_level))
            for (r, c, v) in (H, A):
                f.write(np.ascontiguousarray(r, dtype=np.int32).tobytes())
                f.write(np.ascontiguousarray(c, dtype=np.int32).tobytes())
                f.write(np.ascontiguousarray(v, dtype=np.float64).tobytes())
            for v in (g, c_l, c_u, x_l, x_u):
                f.write(np.ascontiguousarray(v, dtype=np.float64).tobytes())
        env = dict(os.environ)
        env["OMP_CANCELLATION"] = "true"
        cmd = "ulimit -s unlimited 2>/dev/null; exec '%s' '%s' '%s'" % (CQP_DROPIN, pin, pout)
        p = subprocess.run(["bash", "-c", cmd], env=env, capture_output=True, text=True, timeout=timeout)
        if p.returncode != 0 or not os.path.exists(pout):
            raise RuntimeError("cqp driver failed rc=%d\n%s\n%s" % (p.returncode, p.stdout, p.stderr))
        if print_level:
            print(p.stdout)
        if os.environ.get("GSLS_DEBUG"):        # the backend's diagnostics (passes per factorization ...)
            sys.stderr.write(p.stderr)
        buf = open(pout, "rb").read()
        ints = np.frombuffer(buf, dtype="<i4", count=4)
        reals = np.frombuffer(buf, dtype="<f8", count=7, offset=16)
        o = 16 + 56
        x = np.frombuffer(buf, dtype="<f8", count=n, offset=o).copy()
        y = np.frombuffer(buf, dtype="<f8", count=m, offset=o + 8 * n).copy()
        z = np.frombuffer(buf, dtype="<f8", count=n, offset=o + 8 * (n + m)).copy()
        return dict(status=int(ints[0]), iter=int(ints[1]), nfacts=int(ints[2]), obj=float(reals[0]),
                    primal_infeasibility=float(reals[1]), dual_infeasibility=float(reals[2]),
                    complementary_slackness=float(reals[3]), time_total=float(reals[4]),
                    time_factorize=float(reals[5]), time_solve=float(reals[6]), x=x, y=y, z=z)
