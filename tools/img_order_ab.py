"""NOT RUN YET (round 2 ended first).  GSLS_IMG_ORDER = 0 (images run by run) against 1 (in the order the waves of a launch reach them): solve sweep and
factorization of the metric workload (the variable is read when the handle is analysed: one process per setting)."""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, time, ctypes as C
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
from galahad_amd._lib import lib
n, row, col, val, rhs, xs = P.kkt_qpband(1000000, 200000)
m = SMT(n, "COORDINATE", row=row, col=col, val=val)
s, c, i = SLS(), Control(), InformSLS(); s.initialize("gsls", c, i); c.pivot_control = 1; c.node_amalgamation = 24
c.max_iterative_refinements = 0
s.analyse(m, c, i); s.factorize(m, c, i); s.factorize(m, c, i); assert i.status == 0
x = s.solve(m, rhs, c, i)
err = float(np.abs(x - xs).max())
kf, kd, kb = C.c_double(), C.c_double(), C.c_double()
f, b = [], []
for _ in range(12):
    s.solve(m, rhs, c, i)
    lib.gsls_last_solve_kernel_seconds(s.handle, C.byref(kf), C.byref(kd), C.byref(kb))
    f.append(kf.value + kd.value); b.append(kb.value)
for _ in range(3):
    s.factorize(m, c, i)
t = time.perf_counter()
for _ in range(20):
    s.factorize(m, c, i)                   # (through the facade: host values in, includes the copy to the device)
tf = (time.perf_counter() - t) / 20
assert i.status == 0
print("GSLS_IMG_ORDER=%%s: forward %%.1f us, backward %%.1f us, SLS_factorize %%.1f us, max error %%.1e" %% (os.environ.get("GSLS_IMG_ORDER"), 1e6 * sorted(f)[6], 1e6 * sorted(b)[6], 1e6 * tf, err))
''' % (ROOT, ROOT)
for v in ("0", "1"):
    env = dict(os.environ); env["GSLS_IMG_ORDER"] = v
    subprocess.run([sys.executable, "-c", code], env=env, check=True)
