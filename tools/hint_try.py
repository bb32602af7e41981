"""sparse saddle system with an all-zero diagonal: every pivot is 2x2.  After learning, refactorizations should run on the
wave-per-front path (k_front_blk takes the hinted fronts)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(5)
# K = [0 B; B^T 0], B tridiagonal nb x nb (nonsingular: strong diagonal)
rows, cols, vals = [], [], []
for i in range(nb):
    for j in (i - 1, i, i + 1):
        if 0 <= j < nb:
            rows.append(nb + j + 1); cols.append(i + 1); vals.append(4.0 if i == j else rng.uniform(-1, 1))
n = 2 * nb
row, col, val = np.array(rows, np.int32), np.array(cols, np.int32), np.array(vals)
xs = rng.uniform(-1, 1, n); rhs = P.sym_matvec(n, row - 1, col - 1, val, xs)
m = SMT(n, "COORDINATE", row=row, col=col, val=val)
s, c, i = SLS(), Control(), InformSLS(); s.initialize("gsls", c, i); c.pivot_control = 1; c.node_amalgamation = 16
s.analyse(m, c, i)
for rep in range(4):
    t = time.time(); s.factorize(m, c, i); tf = time.time() - t
    x = s.solve(m, rhs, c, i)
    print("rep %d: status %d neg %d two %d delayed %d factor %.4f s residual %.2e" % (rep, i.status, i.negative_eigenvalues, i.two_by_two_pivots, i.delayed_pivots, tf, P.scaled_residual(n, row, col, val, x, rhs)), flush=True)
