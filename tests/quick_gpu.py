import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import problems as P
from galahad_amd import SLS, SMT, Control, InformSLS

def run(name, prob, posdef, perm=None, refine=0):
    n,row,col,val,rhs,xs = prob
    m=SMT(n,"COORDINATE",row=row,col=col,val=val)
    s=SLS(); c=Control(); i=InformSLS()
    s.initialize('gsls',c,i)
    c.pivot_control = 2 if posdef else 1
    c.max_iterative_refinements=refine
    if perm is None: c.ordering=0
    t0=time.time(); s.analyse(m,c,i,PERM=perm); ta=time.time()-t0
    assert i.status==0,('analyse',i.status)
    t0=time.time(); s.factorize(m,c,i); tf=time.time()-t0
    print(name,'factor status',i.status,i.gsls_inform['flag'],'neg',i.negative_eigenvalues,'rank',i.rank,'nlevels',i.gsls_inform['nlevels'],'nsup',i.gsls_inform['num_sup'], 'ta %.3f tf %.3f'%(ta,tf))
    if i.status!=0: return
    t0=time.time(); x=s.solve(m,rhs,c,i); ts=time.time()-t0
    print('   solve status',i.status,'err',np.abs(x-xs).max(),'res',P.scaled_residual(n,row,col,val,x,rhs),'ts %.3f'%ts)
    s.terminate()

run('kat_def',P.kat_definite(),True)
run('kat_def_revperm',P.kat_definite(),True,perm=np.arange(5,0,-1))
run('kat_indef',P.kat_indefinite(),False)
run('band200',P.banded_spd(200,7),True)
run('band3000',P.banded_spd(3000,127),True)
run('band3000i',P.banded_spd(3000,127),False)
run('grid40',P.grid2d(40,40),True)
rng=np.random.default_rng(0)
run('grid40perm',P.grid2d(40,40),True,perm=rng.permutation(1600)+1)
run('grid40perm_i',P.grid2d(40,40),False,perm=rng.permutation(1600)+1)
run('rand2000',P.random_sparse(2000,4,1),True,perm=rng.permutation(2000)+1)
run('band1e5',P.banded_spd(100000,127),True)
