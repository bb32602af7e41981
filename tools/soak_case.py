import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from galahad_amd import SLS, SMT, Control, InformSLS
g = np.load(sys.argv[1]); A = g["A"]; rhs = g["rhs"]; n = A.shape[0]
r, c = np.nonzero(np.tril(A)); row, col, val = (r + 1).astype(np.int32), (c + 1).astype(np.int32), A[r, c]
m = SMT(n, "COORDINATE", row=row, col=col, val=val)
s, ctl, i = SLS(), Control(), InformSLS(); s.initialize("gsls", ctl, i)
ctl.pivot_control = 1; ctl.node_amalgamation = int(g["nemin"]); ctl.max_iterative_refinements = 0
perm = g["perm"]
s.analyse(m, ctl, i) if len(perm) == 0 else s.analyse(m, ctl, i, PERM=perm)
xd = np.linalg.solve(A, rhs)
for rep in range(3):
    s.factorize(m, ctl, i)
    x = s.solve(m, rhs, ctl, i)
    print("rep", rep, "status", i.status, "neg", i.negative_eigenvalues, "two", i.two_by_two_pivots, "delayed", i.delayed_pivots, "err %.2e" % np.abs(x - xd).max(), "res %.2e" % (np.abs(A @ x - rhs).max() / np.abs(rhs).max()), flush=True)
