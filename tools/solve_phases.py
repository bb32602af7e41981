"""Where the waves of the bottom-stage forward kernel (k_wsolve_fwd narrow) spend their time on the metric workload, summed
over one launch.  Needs the diagnostic build: `GSLS_EXTRA=-DGSLS_STAMPS bash galahad_amd/csrc/build.sh`, keep the result as
galahad_amd/libgsls_stamps.so, rebuild the product, and on the GPU box copy the diagnostic library over libgsls.so before
running this (the product library does not export gsls_debug_stamps).  Result of round 2: profiles/r02/solve_experiments.txt."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
import galahad_amd._lib as L
raw = C.CDLL(L.LIB_PATH)
n, row, col, val, rhs, xs = P.kkt_qpband(1000000, 200000)
m = SMT(n, "COORDINATE", row=row, col=col, val=val)
s, ct, i = SLS(), Control(), InformSLS(); s.initialize('gsls', ct, i); ct.pivot_control = 1; ct.node_amalgamation = 24
ct.max_iterative_refinements = 0
s.analyse(m, ct, i); s.factorize(m, ct, i); s.factorize(m, ct, i)
for _ in range(3):
    s.solve(m, rhs, ct, i)
a = (C.c_ulonglong * 64)(); raw.gsls_debug_stamps(a); a = [a[k] for k in range(64)]
s.solve(m, rhs, ct, i)
b = (C.c_ulonglong * 64)(); raw.gsls_debug_stamps(b); b = [b[k] for k in range(64)]
d = [b[k] - a[k] for k in range(32, 38)]
nfr, nw = d[4], d[5]
print("bottom-stage forward kernel: %d waves, %d fronts" % (nw, nfr))
tot = d[0] + d[1] + d[2] + d[3]
for name, v in (("loads issue -> arrival (image, rhs, D, maps)", d[0]), ("recurrence + LDS hand-off + stores issued + next task record", d[1]),
                ("last stores drained", d[2]), ("group record + first task record", d[3])):
    print("  %-62s %7.2f us per front   %5.1f %% of the waves' time" % (name, v / 100.0 / max(nfr, 1), 100.0 * v / max(tot, 1)))
print("  wave lifetime %.1f us on average, %.2f us per front" % (tot / 100.0 / max(nw, 1), tot / 100.0 / max(nfr, 1)))
