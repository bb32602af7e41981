import sys, os, ctypes as C
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
import galahad_amd._lib as L
raw=C.CDLL(L.LIB_PATH)
prob=P.kkt_qpband(50,14)   # order 64, eliminated constraints-first: zero pivots, the optimistic pass fails
n,row,col,val,rhs,xs=prob
m=SMT(n,"COORDINATE",row=row,col=col,val=val)
s,ct,i=SLS(),Control(),InformSLS(); s.initialize('gsls',ct,i); ct.pivot_control=1
s.analyse(m,ct,i,PERM=np.arange(n,0,-1)); s.factorize(m,ct,i)
st=(C.c_ulonglong*64)(); raw.gsls_debug_stamps(st)
v=[st[k] for k in range(40,49)]
print('status',i.status,'2x2',i.two_by_two_pivots,'nodes',i.gsls_inform['num_sup'])
print('cycles: search %d rvbar %d decide %d bar %d swap %d w1 %d update %d looptop %d | total %d'%tuple(v))
