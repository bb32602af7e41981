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
s.factorize(m, ctl, i)
y = s.part_solve("L", rhs.copy(), ctl, i); z = s.part_solve("D", y.copy(), ctl, i); x = s.part_solve("U", z.copy(), ctl, i)
xa = s.solve(m, rhs.copy(), ctl, i)
out = s.enquire(i, want_perm=True, want_pivots=True, want_d=True)
np.savez(sys.argv[2], y=y, z=z, x=x, xa=xa, piv=out["PIVOTS"], d=out["D"], perm=out["PERM"])
