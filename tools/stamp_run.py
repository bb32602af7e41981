import sys, ctypes as C; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
from galahad_amd._lib import lib
import galahad_amd._lib as L
raw=C.CDLL(L.LIB_PATH)
# one dense front: n=192 dense SPD -> nodes? use band with bw large so single node of 3 steps
n=192
A=np.random.default_rng(0).uniform(-1,1,(n,n)); A=A@A.T+n*np.eye(n)
r,c=np.tril_indices(n)
prob=(n,(r+1).astype(np.int32),(c+1).astype(np.int32),A[r,c],A@np.ones(n),np.ones(n))
m=SMT(n,"COORDINATE",row=prob[1],col=prob[2],val=prob[3])
s,ct,i=SLS(),Control(),InformSLS(); s.initialize('gsls',ct,i); ct.pivot_control=2; ct.ordering=0
s.analyse(m,ct,i); 
for rep in range(3):
    s.factorize(m,ct,i)
    st=(C.c_ulonglong*64)(); raw.gsls_debug_stamps(st)
    v=[st[k] for k in range(6)]
    print('nodes',i.gsls_inform['num_sup'],'cycles(100MHz ticks?) gemm %d accP %d regload %d factor %d store %d'%(v[1]-v[0],v[2]-v[1],v[3]-v[2],v[4]-v[3],v[5]-v[4]))
x=s.solve(m,prob[4],ct,i); print(np.abs(x-1).max())
