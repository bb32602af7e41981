"""cfg3, 8 right-hand sides per launch: XCD-aware column mapping against the plain one (GSLS_COLS_XCD is read once per process)."""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, time, ctypes as C
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch, problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
from galahad_amd._lib import lib, Inform
n, row, col, val, rhs, xs = P.kkt_qpband(1000000, 200000)
m = SMT(n, "COORDINATE", row=row, col=col, val=val)
s, c, i = SLS(), Control(), InformSLS(); s.initialize("gsls", c, i); c.pivot_control = 1; c.node_amalgamation = 24
s.analyse(m, c, i); s.factorize(m, c, i); s.factorize(m, c, i); assert i.status == 0
def timed(nrhs, reps=30):
    B = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (nrhs, n))).cuda()
    inf = Inform()
    for _ in range(3):
        X = B.clone(); lib.gsls_solve_dev(s.handle, 0, nrhs, C.c_void_p(X.data_ptr()), n, C.byref(s.opts), C.byref(inf))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        X = B.clone(); lib.gsls_solve_dev(s.handle, 0, nrhs, C.c_void_p(X.data_ptr()), n, C.byref(s.opts), C.byref(inf))
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3
print("GSLS_COLS_XCD=%%s: 1 rhs %%.3f | 2 rhs %%.3f | 4 rhs %%.3f | 8 rhs %%.3f | 16 rhs %%.3f ms" %% (os.environ.get("GSLS_COLS_XCD"), timed(1), timed(2), timed(4), timed(8), timed(16)))
''' % (ROOT, ROOT)
for v in ("0", "1"):
    env = dict(os.environ); env["GSLS_COLS_XCD"] = v
    subprocess.run([sys.executable, "-c", code], env=env, check=True)
