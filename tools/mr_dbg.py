import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
prob = P.banded_spd(6000, 47, seed=3)
n, row, col, val, rhs, xs = prob
m = SMT(n, "COORDINATE", row=row, col=col, val=val)
s, c, i = SLS(), Control(), InformSLS(); s.initialize("gsls", c, i); c.pivot_control = 2
c.node_amalgamation = 32
s.analyse(m, c, i, PERM=np.arange(1, n + 1)); s.factorize(m, c, i)
print("levels", i.gsls_inform["nlevels"], "sup", i.gsls_inform["num_sup"], "maxfront", i.max_front_size)
c.max_iterative_refinements = 0
B = np.asfortranarray(np.random.default_rng(2).uniform(-1, 1, (n, 2)))
x0 = B[:,0]*0; print("single before skipped", i.status, P.scaled_residual(n, row, col, val, x0, B[:, 0]))
X = s.solve(m, B, c, i); print("multi", i.status, [P.scaled_residual(n, row, col, val, X[:, k], B[:, k]) for k in range(2)])
x1 = s.solve(m, B[:, 0].copy(), c, i); print(i.gsls_inform); print("single after", i.status, P.scaled_residual(n, row, col, val, x1, B[:, 0]), np.abs(x1 - x0).max(), np.abs(X[:, 0] - x0).max())
