import sys, os, ctypes as C
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
import galahad_amd._lib as L
raw=C.CDLL(L.LIB_PATH)
prob=P.banded_spd(20000,127)
n,row,col,val,rhs,xs=prob
m=SMT(n,"COORDINATE",row=row,col=col,val=val)
s,ct,i=SLS(),Control(),InformSLS(); s.initialize('gsls',ct,i); ct.pivot_control=2
s.analyse(m,ct,i); s.factorize(m,ct,i)
for rep in range(2):
    x=s.solve(m,rhs,ct,i)
    st=(C.c_ulonglong*64)(); raw.gsls_debug_stamps(st)
    v=[st[k] for k in range(40)]
    d=lambda a,b:v[b]-v[a]
    print('fwd cycles: init %d children %d | blk0: matvec %d y %d upd %d | blk1: matvec %d y %d upd %d | store %d | total %d'%(d(8,23),d(23,9),d(9,24),d(24,25),d(25,26),d(26,28),d(28,29),d(29,30),d(10,11),d(8,11)))
    print('bwd cycles: total %d'%(d(16,20)))
print(np.abs(x-xs).max())
