import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
cases = {"rs4000": P.random_sparse(4000, 6, seed=11, spd=False), "rs2000d3": P.random_sparse(2000, 3, seed=5, spd=False),
         "grid2d_s1": P.grid2d(60, 50, shift=1.0), "grid2d_s2.5": P.grid2d(60, 50, shift=2.5), "kkt": P.kkt_qpband(3000, 600, seed=3),
         "grid2d_s6": P.grid2d(60, 50, shift=6.0)}
for k, (n, row, col, val, rhs, xs) in cases.items():
    m = SMT(n, "COORDINATE", row=row, col=col, val=val)
    s, c, i = SLS(), Control(), InformSLS()
    s.initialize("gsls", c, i)
    s.analyse(m, c, i)
    s.factorize(m, c, i)
    print(k, "status", i.status, "neg", i.negative_eigenvalues, "two", i.two_by_two_pivots, "delays", i.delayed_pivots, flush=True)
    s.terminate()
