import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, problems as P
from galahad_amd import SLS, SMT, Control, InformSLS
prob=P.grid2d(14,14)
n,row,col,val,rhs,xs=prob
import itertools
for free,nemin in itertools.product((False, True),(32,8)):
    m=SMT(n,"COORDINATE",row=row,col=col,val=val)
    s,c,i=SLS(),Control(),InformSLS(); s.initialize('gsls',c,i); c.pivot_control=2; c.node_amalgamation=nemin
    s.analyse(m,c,i,PERM=None if free else np.arange(1,n+1)); s.factorize(m,c,i)
    x=s.solve(m,rhs,c,i)
    sym=s.symbolic()
    bad=np.where(~np.isfinite(x))[0]
    print('free',free,'nemin',nemin,'status',i.status,'nbad',len(bad),'err',np.nanmax(np.abs(x-xs)))
    print(' ncol',np.diff(sym['sptr']).tolist()); print(' nrow',np.diff(sym['rptr']).tolist()); print(' parent',sym['sparent'].tolist())
    order=sym['order']; 
    print(' bad vars',bad[:40].tolist()); print(' bad positions', sorted((order[bad]-1).tolist())[:40])
