"""Assembly-tree statistics of a workload (host only): fronts per level, sizes, children."""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
from galahad_amd._lib import lib, Inform

wl = sys.argv[1] if len(sys.argv) > 1 else "kkt"
nemin = int(sys.argv[2]) if len(sys.argv) > 2 else (24 if wl == "kkt" else 32)
if wl == "kkt":
    prob = P.kkt_qpband(1000000, 200000)
elif wl == "band":
    prob = P.banded_spd(100000, 127)
n, row, col, val, rhs, xs = prob
m = SMT(n, "COORDINATE", row=row, col=col, val=val)
s, c, inf = SLS(), Control(), InformSLS()
s.initialize("gsls", c, inf)
c.pivot_control = 1 if wl == "kkt" else 2
c.node_amalgamation = nemin
s.analyse(m, c, inf)
VAL = s.scatter_values(m)
if wl == "kkt":
    gi = Inform()
    f = lib.gsls_refine_order(s.handle, VAL.ctypes.data_as(C.c_void_p), C.byref(gi))
    assert f >= 0
nn = C.c_int32(); rl = C.c_int64(); nl = C.c_int64()
lib.gsls_get_symbolic_sizes(s.handle, C.byref(nn), C.byref(rl), C.byref(nl))
nn = nn.value
sptr = np.zeros(nn + 1, np.int32); spar = np.zeros(nn, np.int32); rptr = np.zeros(nn + 1, np.int64)
lib.gsls_get_symbolic(s.handle, sptr.ctypes.data_as(C.POINTER(C.c_int32)), spar.ctypes.data_as(C.POINTER(C.c_int32)),
                      rptr.ctypes.data_as(C.POINTER(C.c_int64)), None, None, None)
ncol = np.diff(sptr); nrow = np.diff(rptr).astype(np.int64)
par = spar - 1
level = np.zeros(nn, np.int64)
nchild = np.zeros(nn, np.int64)
for s_ in range(nn):          # postorder: children before parents
    p = par[s_]
    if p < nn:
        level[p] = max(level[p], level[s_] + 1)
        nchild[p] += 1
nl = level.max() + 1
ent = ncol * nrow - ncol * (ncol - 1) // 2
print("nodes %d levels %d nnzL %d" % (nn, nl, ent.sum()))
tiny = (ncol <= 32) & (nrow <= 64)
for l in range(nl):
    k = level == l
    print("level %2d: %6d fronts  tiny %6d  ncol avg %.1f max %d  nrow avg %.1f max %d  entries %9d  children avg %.1f max %d" % (
        l, k.sum(), (k & tiny).sum(), ncol[k].mean(), ncol[k].max(), nrow[k].mean(), nrow[k].max(), ent[k].sum(), nchild[k].mean(), nchild[k].max()))
# depth from root
depth = np.zeros(nn, np.int64)
for s_ in range(nn - 1, -1, -1):
    p = par[s_]
    depth[s_] = depth[p] + 1 if p < nn else 0
print("max depth", depth.max())
for d in range(depth.max() + 1):
    k = depth == d
    print("depth %2d: %6d fronts entries %9d  leaves %d" % (d, k.sum(), ent[k].sum(), (k & (nchild == 0)).sum()))
np.savez("/tmp/tree_%s.npz" % wl, ncol=ncol, nrow=nrow, par=par, level=level)
