import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
from galahad_amd._lib import lib, Inform
prob = P.kkt_qpband(1000000, 200000)
n, row, col, val, rhs, xs = prob
m = SMT(n, "COORDINATE", row=row, col=col, val=val)
for ordering in (-1, 1):
    s, c, i = SLS(), Control(), InformSLS(); s.initialize("gsls", c, i); c.pivot_control = 1; c.node_amalgamation = 24
    c.ordering = ordering
    t = time.time(); s.analyse(m, c, i); ta = time.time() - t
    VAL = s.scatter_values(m); d_val = torch.from_numpy(VAL).cuda(); d_x = torch.from_numpy(rhs).cuda(); inf = Inform()
    for _ in range(4):
        f = lib.gsls_factor_dev(s.handle, 0, C.c_void_p(d_val.data_ptr()), None, C.byref(s.opts), C.byref(inf)); assert f >= 0, f
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20):
        lib.gsls_factor_dev(s.handle, 0, C.c_void_p(d_val.data_ptr()), None, C.byref(s.opts), C.byref(inf))
        x = d_x.clone(); lib.gsls_solve_dev(s.handle, 0, 1, C.c_void_p(x.data_ptr()), n, C.byref(s.opts), C.byref(inf))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print("ordering %2d: analyse %.2f s, levels %d, supernodes %d, nnzL %.3g, flops %.3g, factor+solve %.3f ms, residual %.1e" % (ordering, ta, inf.nlevels, inf.num_sup, inf.num_factor, inf.num_flops, dt * 1e3, P.scaled_residual(n, row, col, val, x.cpu().numpy(), rhs)))
    s.terminate()
