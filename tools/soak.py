"""Randomised soak: many small sparse symmetric systems of varied structure (SPD / indefinite / saddle, own ordering or
natural, several supernode widths) through analyse -> factorize x3 -> solve, every solution against numpy's dense solve
and the inertia against the eigenvalues.  Prints the failures (seed and shape) and a summary."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from galahad_amd import SLS, SMT, Control, InformSLS

def make_structured(rng, kind, n):
    """shapes that give wide fronts: dense, banded, arrow (a dense border), each definite or not"""
    if kind == "dense":
        B = rng.uniform(-1, 1, (n, n)); A = (B + B.T) / 2
    elif kind == "band":
        bw = min(int(rng.integers(20, 220)), n - 1); A = np.zeros((n, n))
        for d in range(1, bw + 1):
            v = rng.uniform(-1, 1, n - d) * (rng.uniform(size=n - d) < 0.7)
            A[np.arange(d, n), np.arange(0, n - d)] = v
        A = A + A.T
    else:  # arrow
        k = min(int(rng.integers(30, 300)), n - 1); A = np.zeros((n, n))
        A[n - k:, :] = rng.uniform(-1, 1, (k, n)) * (rng.uniform(size=(k, n)) < 0.5)
        A = np.tril(A, -1); A = A + A.T
        i = np.arange(n - 1); A[i + 1, i] += rng.uniform(-1, 1, n - 1); A[i, i + 1] = A[i + 1, i]
    sgn = np.ones(n) if rng.uniform() < 0.4 else rng.choice([-1.0, 1.0], n)
    weak = rng.uniform() < 0.4                      # far from diagonal dominance: real pivoting on wide fronts
    fac = rng.uniform(0.02, 0.3) if weak else rng.uniform(0.6, 1.3)
    A[np.arange(n), np.arange(n)] = sgn * (np.abs(A).sum(1) * fac + 0.1)
    return A, bool((sgn > 0).all()) and not weak


def make_ipm(rng, n):
    """interior-point KKT system with slacks, K = [H+D 0 A^T; 0 Ds -I; A -I 0] (what CQP hands to SBLS): H tridiagonal,
    barrier terms spanning many decades"""
    m = max(1, n // 5); nx = n - 2 * m
    if nx < 2:
        nx, m = n - 2, 1
    K = np.zeros((n, n))
    i = np.arange(nx)
    K[i, i] = 2.0 + 10.0 ** rng.uniform(-3, 3, nx)
    K[i[1:], i[:-1]] = -1.0; K[i[:-1], i[1:]] = -1.0
    for c in range(m):
        cols = rng.choice(nx, size=min(nx, int(rng.integers(1, 4))), replace=False)
        K[nx + m + c, cols] = rng.uniform(0.5, 1.5, len(cols)); K[cols, nx + m + c] = K[nx + m + c, cols]
        K[nx + c, nx + c] = 10.0 ** rng.uniform(-4, 4)               # y / s
        K[nx + m + c, nx + c] = -1.0; K[nx + c, nx + m + c] = -1.0
    return K


def make(rng, kind, n):
    if kind == "ipm":
        return make_ipm(rng, n)
    dens = (rng.uniform(1.5, 12.0) if os.environ.get("SOAK_BIG") else rng.uniform(1.5, 6.0)) / n
    M = np.where(rng.uniform(size=(n, n)) < dens, rng.uniform(-1, 1, (n, n)), 0.0)
    A = np.tril(M, -1); A = A + A.T
    if kind == "spd":
        A += np.diag(np.abs(A).sum(1) + rng.uniform(0.1, 1.0, n))
    elif kind == "indef":
        A += np.diag(rng.choice([-1.0, 1.0], n) * (np.abs(A).sum(1) * rng.uniform(0.3, 1.5) + 0.1))
    elif kind == "weakdiag":
        A += np.diag(rng.uniform(-0.3, 0.3, n))
        A += np.diag(np.where(np.abs(A).sum(1) == 0, 1.0, 0.0))
    else:  # saddle: [H B^T; B 0]
        k = n // 3
        H = A[: n - k, : n - k] + np.diag(np.abs(A[: n - k, : n - k]).sum(1) + 0.5)
        B = np.where(rng.uniform(size=(k, n - k)) < 3.0 / (n - k), rng.uniform(-1, 1, (k, n - k)), 0.0)
        for i in range(k):
            B[i, rng.integers(0, n - k)] += 1.0 + rng.uniform()
            B[i, (7 * i) % (n - k)] += 1.0
        A = np.zeros((n, n)); A[: n - k, : n - k] = H; A[n - k:, : n - k] = B; A[: n - k, n - k:] = B.T
    return A

def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0; skipped = 0
    only = int(os.environ.get("SOAK_ONLY", "-1"))
    for it in range(N):
        if it % 50 == 49:
            print("... %d of %d systems, %d failures so far" % (it + 1, N, bad), flush=True)   # (a silent run looks hung)
        kind = ["spd", "indef", "saddle", "weakdiag", "ipm"][it % 5] if os.environ.get("SOAK_IPM") else ["spd", "indef", "saddle", "weakdiag"][it % 4]
        n = int(rng.integers(300, 2500)) if os.environ.get("SOAK_BIG") else int(rng.integers(5, 260))
        if os.environ.get("SOAK_WIDE"):          # wide fronts: dense / banded / arrow
            kind = ["dense", "band", "arrow"][it % 3]
            n = int(rng.integers(70, 900))
            A, definite = make_structured(rng, kind, n)
            kind = "spd" if definite else kind
        else:
            A = make(rng, kind, n)
        ev = np.linalg.eigvalsh(A)
        if np.abs(ev).min() < 1e-8 * np.abs(ev).max():
            skipped += 1; continue
        r, c = np.nonzero(np.tril(A)); 
        if not (A[np.arange(n), np.arange(n)] != 0).all():   # keep explicit zero diagonals out of COO: pattern without them
            pass
        row, col, val = (r + 1).astype(np.int32), (c + 1).astype(np.int32), A[r, c]
        xs = rng.uniform(-1, 1, n); rhs = A @ xs
        nem = int(rng.choice([1, 4, 8, 16, 24, 32, 64]))
        own = it % 3 != 0
        permr = None if own else rng.permutation(n) + 1        # (same draws in the same order whether or not the case runs)
        if only >= 0 and it != only:
            continue
        if only >= 0:
            np.savez("/tmp/soak_case.npz", A=A, rhs=rhs, xs=xs, nemin=nem, perm=np.zeros(0) if own else permr, kind=kind)
        m = SMT(n, "COORDINATE", row=row, col=col, val=val)
        s, ctl, i = SLS(), Control(), InformSLS(); s.initialize("gsls", ctl, i)
        ctl.pivot_control = 2 if kind == "spd" and it % 8 < 4 else 1
        ctl.node_amalgamation = nem
        ctl.max_iterative_refinements = 0
        if it % 6 == 5:
            ctl.max_iterative_refinements = 1                  # SLS_solve_ir on the device
        if kind != "spd" and it % 5 == 4:
            ctl.scaling = -1 if it % 10 == 4 else -2          # the backend's own scalings
        s.analyse(m, ctl, i) if own else s.analyse(m, ctl, i, PERM=permr)
        xd = np.linalg.solve(A, rhs)
        cond = np.abs(ev).max() / np.abs(ev).min()
        if kind == "indef" and it % 16 == 5 and ev.min() < 0:
            # asked for a Cholesky factorization of an indefinite matrix: must be refused (SSIDS_ERROR_NOT_POS_DEF ->
            # GALAHAD_error_inertia), never "succeed"
            ctl.pivot_control = 2
            s.analyse(m, ctl, i) if own else s.analyse(m, ctl, i, PERM=permr)
            s.factorize(m, ctl, i)
            if i.status == 0:
                bad += 1
                print("FAIL it %d: Cholesky of an indefinite matrix (n %d, %d negative eigenvalues) returned status 0" % (it, n, int((ev < 0).sum())), flush=True)
            s.terminate()
            continue
        for rep in range(3):
            s.factorize(m, ctl, i)
            ok = i.status == 0
            if ok:
                if it % 7 == 3:           # several right-hand sides at once (3, or 2..17 with random columns)
                    if it % 14 == 3:
                        X = s.solve(m, np.column_stack([rhs, 2 * rhs, -rhs]), ctl, i)
                        lin = max(np.abs(X[:, 1] - 2 * X[:, 0]).max(), np.abs(X[:, 2] + X[:, 0]).max())
                        x = X[:, 0] if lin <= 1e-9 * max(1.0, np.abs(X).max()) else X[:, 0] * np.nan
                    else:                 # every column must carry the bits of a single-column solve
                        r2 = np.random.default_rng(1000 + it)
                        k = int(r2.integers(2, 18))
                        Bm = np.asfortranarray(np.column_stack([rhs] + [r2.uniform(-1, 1, n) for _ in range(k - 1)]))
                        X = s.solve(m, Bm, ctl, i)
                        j = int(r2.integers(1, k))
                        same = ctl.max_iterative_refinements > 0 or \
                            (np.array_equal(X[:, j], s.solve(m, Bm[:, j].copy(), ctl, i)) and
                             np.array_equal(X[:, 0], s.solve(m, rhs.copy(), ctl, i)))
                        x = X[:, 0] if same else X[:, 0] * np.nan
                else:
                    x = s.solve(m, rhs, ctl, i)
                err = np.abs(x - xd).max() / max(1.0, np.abs(xd).max())
                res = np.abs(A @ x - rhs).max() / (np.abs(A).max() * max(1.0, np.abs(x).max()) + np.abs(rhs).max())
                ok = (err <= 1e-11 * max(cond, 1e2) or (cond > 1e8 and res <= 1e-11)) and \
                    i.negative_eigenvalues == int((ev < 0).sum()) and i.rank == n
                if ok and it % 4 == 1 and ctl.max_iterative_refinements == 0:
                    # SLS_part_solve: L, D, U compose to the solve; the pivots SLS_enquire returns carry the inertia
                    y = s.part_solve("L", rhs.copy(), ctl, i)
                    z = s.part_solve("D", y, ctl, i)
                    x3 = s.part_solve("U", z, ctl, i)
                    perr = np.abs(x3 - x).max() / max(1.0, np.abs(x).max())
                    if ctl.pivot_control == 2:
                        d = s.enquire(i, want_d=True)["D"][0]
                        ineg = int((d < 0).sum())
                    else:
                        out = s.enquire(i, want_pivots=True, want_d=True)
                        piv, d = out["PIVOTS"], out["D"]
                        order = np.argsort(np.abs(piv)); ineg = 0; k = 0
                        while k < n:                      # D is stored inverted, in pivot order; 2x2: negative PIVOTS pairs
                            if piv[order[k]] > 0:
                                ineg += d[0, k] < 0; k += 1
                            else:
                                det = d[0, k] * d[0, k + 1] - d[1, k] ** 2
                                ineg += 1 if det < 0 else (2 if d[0, k] + d[0, k + 1] < 0 else 0); k += 2
                    ok = perr <= 1e-12 * max(cond, 1e2) and ineg == int((ev < 0).sum())
                    if not ok:
                        err = -perr
            if not ok:
                bad += 1
                print("FAIL it %d kind %s n %d nemin %d own %d rep %d status %d neg %d/%d err %.2e cond %.1e" % (
                    it, kind, n, ctl.node_amalgamation, own, rep, i.status, i.negative_eigenvalues, int((ev < 0).sum()),
                    err if i.status == 0 else -1, cond), flush=True)
                break
        # the same pattern with other values (what an interior-point or trust-region iteration does): the learned order,
        # hints and fast paths must either still work or fall back -- never give a wrong answer
        for drift in range(2 if i.status == 0 else 0):
            f = 10.0 ** rng.uniform(-1.5, 1.5, n) if drift == 0 else 1.0 + 0.5 * rng.uniform(-1, 1, n)
            val2 = val * np.where(row == col, f[row - 1], 1.0 + 0.3 * rng.uniform(-1, 1, len(val)))
            A2 = np.zeros((n, n)); A2[row - 1, col - 1] = val2; A2 = A2 + A2.T - np.diag(np.diag(A2))
            ev2 = np.linalg.eigvalsh(A2)
            if np.abs(ev2).min() < 1e-8 * np.abs(ev2).max():
                continue
            m2 = SMT(n, "COORDINATE", row=row, col=col, val=val2)
            rhs2 = A2 @ xs
            if ctl.pivot_control == 2 and ev2.min() <= 0:
                continue
            s.factorize(m2, ctl, i)
            ok = i.status == 0
            err = -1.0
            if ok:
                x = s.solve(m2, rhs2, ctl, i)
                xd2 = np.linalg.solve(A2, rhs2)
                cond2 = np.abs(ev2).max() / np.abs(ev2).min()
                err = np.abs(x - xd2).max() / max(1.0, np.abs(xd2).max())
                ok = (err <= 1e-11 * max(cond2, 1e2)) and i.negative_eigenvalues == int((ev2 < 0).sum()) and i.rank == n
            if not ok:
                bad += 1
                print("FAIL(drift %d) it %d kind %s n %d nemin %d own %d status %d neg %d/%d err %.2e" % (
                    drift, it, kind, n, ctl.node_amalgamation, own, i.status, i.negative_eigenvalues, int((ev2 < 0).sum()), err), flush=True)
                break
        s.terminate()
    print("soak: %d systems, %d skipped (singular), %d failures" % (N, skipped, bad))
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
