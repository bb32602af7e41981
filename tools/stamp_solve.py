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
    v=[st[k] for k in range(32)]
    print('fwd: gather %d  trsv-blocks %d  store %d  gemv %d | bwd: load %d gemvT %d trsv %d store %d'%(v[9]-v[8],v[10]-v[9],0,v[11]-v[10], v[17]-v[16], v[18]-v[17], v[19]-v[18], v[20]-v[19]))
print(np.abs(x-xs).max())
