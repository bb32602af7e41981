import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
def run(name, prob, posdef=True):
    n,row,col,val,rhs,xs=prob
    m=SMT(n,"COORDINATE",row=row,col=col,val=val)
    s,c,i=SLS(),Control(),InformSLS(); s.initialize('gsls',c,i); c.pivot_control=2 if posdef else 1
    t=time.time(); s.analyse(m,c,i); ta=time.time()-t
    g=i.gsls_inform
    print(name,'analyse %.2fs nfact %.3e flops %.3e maxfront %d levels %d nsup %d'%(ta,g['num_factor'],g['num_flops'],g['maxfront'],g['nlevels'],g['num_sup']),flush=True)
    t=time.time(); s.factorize(m,c,i); tf=time.time()-t
    print('   factor status',i.status,i.gsls_inform['flag'],'%.3fs -> %.1f GF/s'%(tf, g['num_flops']/tf/1e9),flush=True)
    if i.status!=0: return
    t=time.time(); s.factorize(m,c,i); tf=time.time()-t
    print('   refactor %.3fs -> %.1f GF/s'%(tf, g['num_flops']/tf/1e9),flush=True)
    t=time.time(); x=s.solve(m,rhs,c,i); ts=time.time()-t
    print('   solve status',i.status,i.gsls_inform['flag'],'%.3fs'%ts,'res %.2e'%P.scaled_residual(n,row,col,val,x,rhs),flush=True)
    s.terminate()
run('grid3d_40',P.grid3d(40,40,40))
run('grid3d_40i',P.grid3d(40,40,40),posdef=False)
run('grid3d_64',P.grid3d(64,64,64))
run('grid3d_100',P.grid3d(100,100,100))
run('grid3d_126',P.grid3d(126,126,126))
