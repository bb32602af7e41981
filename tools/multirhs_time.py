"""cfg2 (banded SPD, n = 1e5, semi-bandwidth 127): time of SLS_solve for 1 and 8 right-hand sides, values resident in HBM."""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
from galahad_amd._lib import lib, Inform
prob = P.banded_spd(100000, 127)
n, row, col, val, rhs, xs = prob
m = SMT(n, "COORDINATE", row=row, col=col, val=val)
s, c, i = SLS(), Control(), InformSLS(); s.initialize("gsls", c, i); c.pivot_control = 2; c.node_amalgamation = 64
s.analyse(m, c, i); s.factorize(m, c, i); assert i.status == 0
def timed(nrhs, reps=20):
    B = torch.from_numpy(np.asfortranarray(np.random.default_rng(1).uniform(-1, 1, (n, nrhs))).T.copy()).cuda()   # (nrhs, n) rows = columns
    inf = Inform()
    for _ in range(3):
        X = B.clone(); lib.gsls_solve_dev(s.handle, 0, nrhs, C.c_void_p(X.data_ptr()), n, C.byref(s.opts), C.byref(inf))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        X = B.clone(); lib.gsls_solve_dev(s.handle, 0, nrhs, C.c_void_p(X.data_ptr()), n, C.byref(s.opts), C.byref(inf))
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3
t1 = timed(1); t8 = timed(8)
os.environ["GSLS_NO_MULTIRHS"] = "1"
t8loop = timed(8)
print("cfg2 solve: 1 rhs %.3f ms | 8 rhs blocked %.3f ms (%.2fx) | 8 rhs column by column %.3f ms" % (t1, t8, t8 / t1, t8loop))

# cfg3 (KKT, LDL^T): 8 columns side by side on four streams against one after the other
os.environ.pop("GSLS_NO_MULTIRHS")
prob = P.kkt_qpband(1000000, 200000)
n, row, col, val, rhs, xs = prob
m = SMT(n, "COORDINATE", row=row, col=col, val=val)
s, c, i = SLS(), Control(), InformSLS(); s.initialize("gsls", c, i); c.pivot_control = 1; c.node_amalgamation = 24
s.analyse(m, c, i); s.factorize(m, c, i); s.factorize(m, c, i); assert i.status == 0
t1 = timed(1); t4 = timed(4); t8 = timed(8); t16 = timed(16)
os.environ["GSLS_NO_MULTIRHS"] = "1"
t8loop = timed(8)
print("cfg3 solve: 1 rhs %.3f ms | 4 rhs %.3f | 8 rhs in one launch %.3f ms (%.2fx) | 16 rhs %.3f | 8 rhs column by column %.3f ms" % (t1, t4, t8, t8 / t1, t16, t8loop))
