"""K = [0 B; B^T 0] instances through the backend (GPU): passes, inertia, residual, time."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems as P
from galahad_amd import SLS, SMT, Control, InformSLS

def run(nb, seed=1, reps=3):
    rng = np.random.default_rng(seed)
    n = 2 * nb
    r, c, v = [], [], []
    for i in range(nb):
        for j in {i} | set(rng.integers(0, nb, 3).tolist()):
            r.append(nb + j); c.append(i)
            v.append(2.0 + rng.uniform(0, 1) if j == i else rng.uniform(-0.3, 0.3))
    row, col, val = np.array(r, dtype=np.int32) + 1, np.array(c, dtype=np.int32) + 1, np.array(v)
    xs = rng.uniform(-1, 1, n)
    rhs = P.sym_matvec(n, row - 1, col - 1, val, xs)
    m = SMT(n, "COORDINATE", row=row, col=col, val=val)
    s, ct, i = SLS(), Control(), InformSLS()
    s.initialize("gsls", ct, i)
    ct.pivot_control = 1
    s.analyse(m, ct, i)
    for rep in range(reps):
        t0 = time.perf_counter()
        s.factorize(m, ct, i)
        t1 = time.perf_counter()
        x = s.solve(m, rhs, ct, i)
        res = P.scaled_residual(n, row, col, val, x, rhs)
        print("nb=%d rep %d: status %d neg %d rank %d two %d delayed %d  factor %.3fs  residual %.2e" % (
            nb, rep, i.status, i.negative_eigenvalues, i.rank, i.two_by_two_pivots, i.delayed_pivots, t1 - t0, res), flush=True)
    s.terminate()

for nb in [int(a) for a in sys.argv[1:]] or [500, 700]:
    run(nb)
