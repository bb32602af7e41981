import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
rng=np.random.default_rng(1)
for k,half in ((500,False),(1500,True)):
    # K = [D B; B^T 0] with D = 0 (half=False: every diagonal entry zero) or D = diag(+-1) on the first block
    nb=k; n=2*nb
    r=[];c=[];v=[]
    for i in range(nb):
        cols={i}|set(rng.integers(0,nb,3).tolist())
        for j in cols:
            r.append(nb+j); c.append(i); v.append(2.0+rng.uniform(0,1) if j==i else rng.uniform(-0.3,0.3))
    if half:
        for i in range(nb): r.append(i); c.append(i); v.append(rng.choice([-1.0,1.0]))
    row=np.array(r)+1; col=np.array(c)+1; val=np.array(v)
    xs=rng.uniform(-1,1,n); rhs=P.sym_matvec(n,row-1,col-1,val,xs)
    m=SMT(n,"COORDINATE",row=row.astype(np.int32),col=col.astype(np.int32),val=val)
    s,ct,i=SLS(),Control(),InformSLS(); s.initialize('gsls',ct,i); ct.pivot_control=1
    s.analyse(m,ct,i)
    for rep in range(3):
        s.factorize(m,ct,i)
        x=s.solve(m,rhs,ct,i)
        print('half' if half else 'allzero','n',n,'status',i.status,'neg',i.negative_eigenvalues,'two',i.two_by_two_pivots,'delays',i.delayed_pivots,'res %.1e'%P.scaled_residual(n,row,col,val,x,rhs),flush=True)
    s.terminate()
