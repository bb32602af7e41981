"""Where the time of k_wsolve_tail goes (diagnostic build -DGSLS_STAMPS): wave 0's timeline of the FORWARD half --
per front: records fetched, loads arrived, pulls + recurrence + stores done, barrier passed.
usage (GPU box): GSLS_EXTRA=-DGSLS_STAMPS bash galahad_amd/csrc/build.sh && python tools/stamp_tail.py"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
import galahad_amd._lib as L
raw = C.CDLL(L.LIB_PATH)
prob = P.kkt_qpband(1000000, 200000, seed=20240102)
n, row, col, val, rhs, xs = prob
m = SMT(n, "COORDINATE", row=row, col=col, val=val)
s, ct, i = SLS(), Control(), InformSLS(); s.initialize('gsls', ct, i); ct.pivot_control = 1; ct.node_amalgamation = 24
s.analyse(m, ct, i); s.factorize(m, ct, i); s.factorize(m, ct, i)
for rep in range(3):
    x = s.solve(m, rhs, ct, i)
    st = (C.c_ulonglong * 64)(); raw.gsls_debug_stamps(st)
    v = [st[k] for k in range(62)]
    k = max(j for j in range(62) if v[j] > 0 and (j == 0 or v[j] >= v[j - 1])) if v[0] else 0
    print("rep %d: ticks of 10 ns since kernel entry:" % rep, [int(v[j] - v[0]) for j in range(0, k + 1)])
print(np.abs(x - xs).max())
