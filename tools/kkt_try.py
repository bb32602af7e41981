import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
def run(name, prob, free=True, refine=0):
    n,row,col,val,rhs,xs=prob
    m=SMT(n,"COORDINATE",row=row,col=col,val=val)
    s,c,i=SLS(),Control(),InformSLS(); s.initialize('gsls',c,i); c.pivot_control=1
    c.max_iterative_refinements=refine
    if not free: c.ordering=0
    t=time.time(); s.analyse(m,c,i); ta=time.time()-t
    t=time.time(); s.factorize(m,c,i); tf=time.time()-t
    g=i.gsls_inform
    print(name,'status',i.status,'flag',g['flag'],'delays',g['num_delay'],'neg',g['num_neg'],'two',g['num_two'],'nfact %.3e'%g['num_factor'],'levels',g['nlevels'],'ta %.2f tf %.3f'%(ta,tf),flush=True)
    if i.status==0:
        t=time.time(); s.factorize(m,c,i); tf2=time.time()-t
        x=s.solve(m,rhs,c,i)
        print('    refactor %.3f  res %.2e err %.2e'%(tf2,P.scaled_residual(n,row,col,val,x,rhs),np.abs(x-xs).max()),flush=True)
    s.terminate()
rng=np.random.default_rng(5)
run('kkt300perm',P.kkt_qpband(300,60)+(), free=False) if False else None
g=np.load('tests/golden/kkt_300_60_perm.npz')
n=int(g['n']); m=SMT(n,"COORDINATE",row=g['row'],col=g['col'],val=g['val'])
s,c,i=SLS(),Control(),InformSLS(); s.initialize('gsls',c,i); s.analyse(m,c,i,PERM=g['perm']); s.factorize(m,c,i)
print('golden kkt perm: status',i.status,i.gsls_inform['flag'],'delays',i.gsls_inform['num_delay'],'neg',i.negative_eigenvalues)
x=s.solve(m,g['rhs'],c,i); print('  err vs ref',np.abs(x-g['ref_x']).max()); s.terminate()
run('grid80_indef_nat',P.grid2d(80,80,shift=1.0),free=False)
run('grid80_indef_nd',P.grid2d(80,80,shift=1.0))
run('kkt_6000_nd',P.kkt_qpband(6000,1200))
run('kkt_1e5_nd',P.kkt_qpband(100000,20000))
run('kkt_1e6_nd',P.kkt_qpband(1000000,200000),refine=1)
run('grid707_indef_nd',P.grid2d(707,707,shift=1.0))
