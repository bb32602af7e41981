import sys, os, ctypes as C
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
from galahad_amd._lib import lib
import galahad_amd._lib as L
raw=C.CDLL(L.LIB_PATH)
# one dense front: n=192 dense SPD -> nodes? use band with bw large so single node of 3 steps
n=int(sys.argv[1]) if len(sys.argv)>1 else 256
A=np.random.default_rng(0).uniform(-1,1,(n,n)); A=A@A.T+n*np.eye(n)
r,c=np.tril_indices(n)
prob=(n,(r+1).astype(np.int32),(c+1).astype(np.int32),A[r,c],A@np.ones(n),np.ones(n))
m=SMT(n,"COORDINATE",row=prob[1],col=prob[2],val=prob[3])
s,ct,i=SLS(),Control(),InformSLS(); s.initialize('gsls',ct,i); ct.pivot_control=2; ct.ordering=0
s.analyse(m,ct,i); 
for rep in range(3):
    s.factorize(m,ct,i)
    st=(C.c_ulonglong*64)(); raw.gsls_debug_stamps(st)
    v=[st[k] for k in range(18)]
    d=lambda a,b:(v[b]-v[a])*10
    print('ns: gemm %d buildP %d |'%(d(0,1),d(1,2)),' '.join('a%d b%d c%d'%(d(2 if q==0 else 3+3*q,4+3*q),d(4+3*q,5+3*q),d(5+3*q,6+3*q)) for q in range(4)),'| store %d total %d'%(d(16,17),d(0,17)))
x=s.solve(m,prob[4],ct,i); print(np.abs(x-1).max())
