"""cfg3 (KKT, LDL^T): a few SLS_solve calls with nrhs right-hand sides resident in HBM, for rocprofv3 --kernel-trace
(usage: python3 tools/multirhs_prof.py <nrhs> [reps])."""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
from galahad_amd._lib import lib, Inform
nrhs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n, row, col, val, rhs, xs = P.kkt_qpband(1000000, 200000)
m = SMT(n, "COORDINATE", row=row, col=col, val=val)
s, c, i = SLS(), Control(), InformSLS(); s.initialize("gsls", c, i); c.pivot_control = 1; c.node_amalgamation = 24
s.analyse(m, c, i); s.factorize(m, c, i); s.factorize(m, c, i); assert i.status == 0
B = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (nrhs, n))).cuda()
inf = Inform()
for _ in range(reps):
    X = B.clone(); torch.cuda.synchronize(); t = time.perf_counter()
    lib.gsls_solve_dev(s.handle, 0, nrhs, C.c_void_p(X.data_ptr()), n, C.byref(s.opts), C.byref(inf))
    torch.cuda.synchronize(); print("nrhs %d: %.3f ms" % (nrhs, (time.perf_counter() - t) * 1e3))
