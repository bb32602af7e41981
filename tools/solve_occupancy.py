"""Is the bottom stage of the solve bound by latency (time ~ 1 / waves per CU) or by the memory system (time constant)?
GSLS_WS_LDSPAD adds unused dynamic LDS to the narrow wave kernels: 0 -> 20 waves per CU, 22000 -> 16, 35000 -> 12, 40000 -> 8."""
import sys, os, subprocess
import os; os.environ.setdefault("GSLS_SOLVE_PHASES", "1")   # forward / backward separately (events between the phases)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, ctypes as C
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
from galahad_amd._lib import lib
n, row, col, val, rhs, xs = P.kkt_qpband(1000000, 200000)
m = SMT(n, "COORDINATE", row=row, col=col, val=val)
s, c, i = SLS(), Control(), InformSLS(); s.initialize("gsls", c, i); c.pivot_control = 1; c.node_amalgamation = 24
c.max_iterative_refinements = 0
s.analyse(m, c, i); s.factorize(m, c, i); s.factorize(m, c, i); assert i.status == 0
kf, kd, kb = C.c_double(), C.c_double(), C.c_double()
f, b = [], []
for _ in range(12):
    s.solve(m, rhs, c, i)
    lib.gsls_last_solve_kernel_seconds(s.handle, C.byref(kf), C.byref(kd), C.byref(kb))
    f.append(kf.value + kd.value); b.append(kb.value)
print("GSLS_WS_LDSPAD=%%6s: forward %%.1f us, backward %%.1f us (medians of 12)" %% (os.environ.get("GSLS_WS_LDSPAD", "0"), 1e6 * sorted(f)[6], 1e6 * sorted(b)[6]))
''' % (ROOT, ROOT)
for v in ("0", "22000", "35000", "40000"):
    env = dict(os.environ); env["GSLS_WS_LDSPAD"] = v
    subprocess.run([sys.executable, "-c", code], env=env, check=True)
