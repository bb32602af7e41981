import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
n, row, col, val, rhs, xs = P.kkt_qpband(1000000, 200000)
m = SMT(n, "COORDINATE", row=row, col=col, val=val)
s, c, i = SLS(), Control(), InformSLS()
s.initialize("gsls", c, i); c.node_amalgamation = 24
s.analyse(m, c, i); s.factorize(m, c, i); s.factorize(m, c, i)
x = s.solve(m, rhs, c, i)
for k in range(4):
    t0 = time.perf_counter(); r = s._residual_dev(m, rhs, x); t1 = time.perf_counter()
    print("residual on device %.2f ms  max|r| %.2e" % ((t1 - t0) * 1e3, np.abs(r).max()))
t0 = time.perf_counter(); s.factorize(m, c, i); t1 = time.perf_counter(); x = s.solve(m, rhs, c, i); t2 = time.perf_counter()
print("python facade factorize %.2f ms solve %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
