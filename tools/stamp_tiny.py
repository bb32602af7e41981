import sys, os, ctypes as C
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
import galahad_amd._lib as L
raw=C.CDLL(L.LIB_PATH)
prob=P.kkt_qpband(1000000,200000)
n,row,col,val,rhs,xs=prob
m=SMT(n,"COORDINATE",row=row,col=col,val=val)
s,ct,i=SLS(),Control(),InformSLS(); s.initialize('gsls',ct,i); ct.pivot_control=1; ct.node_amalgamation=24
s.analyse(m,ct,i)
for rep in range(3):
    s.factorize(m,ct,i)
    st=(C.c_ulonglong*64)(); raw.gsls_debug_stamps(st)
    v=[st[k] for k in range(49,53)]; w=[st[56]]
    print('status',i.status,'root front, 100 MHz ticks: A part %d | children %d | pivots %d | outputs %d'%(w[0]-v[0], v[1]-w[0], v[2]-v[1], v[3]-v[2]))
