"""Randomised soak THROUGH THE REAL SLS FACADE (oracle/_ref/sls_gsls_driver = the reference's sls.f90 with the gsls arms,
Fortran host code, its own coordinate -> CSR map, scatter, refinement): seeded systems handed over as untidy COO (entries
split into duplicates, upper-triangle entries, out-of-range entries -- sls.f90:8409-8578 semantics), with the controls
a caller can set (scaling -1 / -2, ordering 1, refinement, PERM or own ordering), against numpy's dense solve."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import soak as SK
from oracle import refio

def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for it in range(N):
        kind = ["spd", "indef", "saddle", "weakdiag"][it % 4]
        n = int(rng.integers(5, 400))
        A = SK.make(rng, kind, n)
        ev = np.linalg.eigvalsh(A)
        if np.abs(ev).min() < 1e-8 * np.abs(ev).max():
            continue
        r, c = np.nonzero(np.tril(A)); v = A[r, c]
        row, col, val = list(r + 1), list(c + 1), list(v)
        for k in rng.choice(len(v), size=max(1, len(v) // 10), replace=False):      # duplicates: a + b = original
            a = rng.uniform(-1, 1); val[k] -= a; row.append(row[k]); col.append(col[k]); val.append(a)
        for k in rng.choice(len(v), size=max(1, len(v) // 15), replace=False):      # entries given in the upper triangle
            row[k], col[k] = col[k], row[k]
        for _ in range(3):                                                           # out of range: ignored
            row.append(n + 1 + int(rng.integers(0, 5))); col.append(int(rng.integers(1, n + 1))); val.append(rng.uniform(-1, 1))
        p = rng.permutation(len(val))
        row, col, val = np.array(row, np.int32)[p], np.array(col, np.int32)[p], np.array(val)[p]
        xs = rng.uniform(-1, 1, n); rhs = A @ xs
        posdef = kind == "spd" and it % 8 < 4
        kw = dict(solver="gsls", nemin=int(rng.choice([4, 8, 16, 32])), pivot_control=2 if posdef else 1)
        if it % 3 == 0:
            kw["perm"] = (rng.permutation(n) + 1).astype(np.int32)
        elif it % 3 == 1:
            kw["ordering"] = 1                      # AMD
        if not posdef and it % 5 == 4:
            kw["scaling"] = -1 if it % 10 == 4 else -2
        if it % 6 == 5:
            kw["max_refine"] = 1
        res = refio.run(n, row, col, val, rhs, **kw)
        st = (res["status_analyse"], res["status_factorize"], res["status_solve"])
        cond = np.abs(ev).max() / np.abs(ev).min()
        err = np.abs(res["x"] - xs).max() if st == (0, 0, 0) else -1.0
        ok = st == (0, 0, 0) and err <= 1e-11 * max(cond, 1e2) and res["negative_eigenvalues"] == int((ev < 0).sum()) and res["rank"] == n
        if not ok:
            bad += 1
            print("FAIL it %d kind %s n %d %s status %s neg %d/%d err %.2e cond %.1e" % (it, kind, n, {k: (v if not hasattr(v, "shape") else "perm") for k, v in kw.items()}, st, res["negative_eigenvalues"], int((ev < 0).sum()), err, cond), flush=True)
    print("soak_facade: %d systems, %d failures" % (N, bad))
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
