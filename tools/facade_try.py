"""Timings of the REAL SLS / SBLS facades over the gsls backend on the metric workload (host arrays in, host x out)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems as P
from oracle import refio
n0, m0 = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000000, 200000)
n, row, col, val, rhs, xs = P.kkt_qpband(n0, m0)
t0 = time.time()
r = refio.run(n, row, col, val, rhs, solver="gsls", pivot_control=1, nemin=24, repeat=7, max_refine=0)
print("SLS facade: status", r["status_analyse"], r["status_factorize"], r["status_solve"], "analyse %.3f s factorize median %.2f ms solve median %.2f ms  err %.1e  (wall %.1f s)" % (
    r["t_analyse"], r["t_factorize_median"] * 1e3, r["t_solve_median"] * 1e3, np.abs(r["x"] - xs).max(), time.time() - t0), flush=True)
r = refio.run(n, row, col, val, rhs, solver="gsls", pivot_control=1, nemin=24, repeat=7, max_refine=1)
print("SLS facade + 1 refinement: factorize median %.2f ms solve median %.2f ms" % (r["t_factorize_median"] * 1e3, r["t_solve_median"] * 1e3), flush=True)
# SBLS: K = [H A^T; A 0]
i = np.arange(n0)
H = (np.concatenate([i, i[1:]]) + 1, np.concatenate([i, i[:-1]]) + 1, val[: 2 * n0 - 1])
A = (np.concatenate([np.arange(m0), np.arange(m0)]) + 1, np.concatenate([np.arange(m0), m0 + np.arange(m0)]) + 1, np.ones(2 * m0))
Cm = (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0))
r = refio.run_sbls(n0, m0, H, A, Cm, rhs, solver="gsls", factorization=2, repeat=7, itref_max=1)
print("SBLS facade: status", r["status_factorize"], r["status_solve"], "form+factorize median %.2f ms solve median %.2f ms err %.1e" % (
    r["t_factorize_median"] * 1e3, r["t_solve_median"] * 1e3, np.abs(r["sol"] - xs).max()), flush=True)
