import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
g = int(sys.argv[1]) if len(sys.argv) > 1 else 126
t = time.time(); prob = P.grid3d_27pt_perturbed(g, g, g); n, row, col, val, rhs, xs = prob
print("n %d lower entries %d (full nnz %d)  generated in %.1f s" % (n, len(row), 2 * len(row) - n, time.time() - t), flush=True)
m = SMT(n, "COORDINATE", row=row, col=col, val=val)
s, c, i = SLS(), Control(), InformSLS(); s.initialize("gsls", c, i); c.pivot_control = 1
t = time.time(); s.analyse(m, c, i); print("analyse %.1f s status %d nnzL %.3g flops %.3g fronts max %d" % (time.time() - t, i.status, i.entries_in_factors, i.flops_elimination, i.max_front_size), flush=True)
t = time.time(); s.factorize(m, c, i); tf = time.time() - t
print("factorize %.2f s status %d neg %d two %d delayed %d -> %.1f TF/s" % (tf, i.status, i.negative_eigenvalues, i.two_by_two_pivots, i.delayed_pivots, i.flops_elimination / tf / 1e12), flush=True)
if i.status != 0:
    print(i.gsls_inform, flush=True)
    sys.exit(0)
t = time.time(); s.factorize(m, c, i); tf = time.time() - t
print("refactorize %.2f s -> %.1f TF/s" % (tf, i.flops_elimination / tf / 1e12), flush=True)
t = time.time(); x = s.solve(m, rhs, c, i); print("solve %.3f s residual %.2e err %.2e" % (time.time() - t, P.scaled_residual(n, row, col, val, x, rhs), np.abs(x - xs).max()), flush=True)
