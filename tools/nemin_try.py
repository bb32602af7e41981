import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
def run(name, prob, posdef, nemin):
    n,row,col,val,rhs,xs=prob
    m=SMT(n,"COORDINATE",row=row,col=col,val=val)
    s,c,i=SLS(),Control(),InformSLS(); s.initialize('gsls',c,i); c.pivot_control=2 if posdef else 1; c.node_amalgamation=nemin
    s.analyse(m,c,i); s.factorize(m,c,i)
    assert i.status==0, i.status
    ts=[]
    for _ in range(5):
        t=time.perf_counter(); s.factorize(m,c,i); x=s.solve(m,rhs,c,i); ts.append(time.perf_counter()-t)
    print(name,'nemin',nemin,'factor+solve (host arrays) best %.2f ms'%(min(ts)*1e3),'levels',i.gsls_inform['nlevels'],'nnzL %.2e'%i.entries_in_factors,'res %.1e'%P.scaled_residual(n,row,col,val,x,rhs),flush=True)
    s.terminate()
for ne in (24,64):
    run('grid707 SPD (H+0.5I)',P.grid2d(707,707,shift=-0.5),True,ne)
    run('grid707 indef (H-I)',P.grid2d(707,707,shift=1.0),False,ne)
    run('grid3d 60^3 SPD',P.grid3d(60,60,60),True,ne)
