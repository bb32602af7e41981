"""GSLS_DEBUG_LAUNCH=1: the wave-per-front launches of one refactorization of the metric workload (class, count, LDS)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["GSLS_DEBUG_LAUNCH"] = "1"
import numpy as np, problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
n, row, col, val, rhs, xs = P.kkt_qpband(1000000, 200000)
m = SMT(n, "COORDINATE", row=row, col=col, val=val)
s, ct, i = SLS(), Control(), InformSLS(); s.initialize('gsls', ct, i); ct.pivot_control = 1; ct.node_amalgamation = 24
s.analyse(m, ct, i)
s.factorize(m, ct, i)
print("---- refactorization", file=sys.stderr, flush=True)
s.factorize(m, ct, i)
print("status", i.status)
