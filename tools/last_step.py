import csv,glob,sys,collections
d=sys.argv[1]
import os
f=max(glob.glob(d+'/*/*_kernel_trace.csv'), key=os.path.getmtime)
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last bench step = everything after the solve kernels of the step before it (a refactorization that runs on the
# wave-per-front kernels only no longer launches k_iota, the old marker)
names=[r['Kernel_Name'] for r in rows]
def is_solve(n): return any(t in n for t in ('k_wsolve_','k_solve_','k_permute_out','k_big_','k_permute_in'))
end=max(i for i,n in enumerate(names) if is_solve(n))          # last solve kernel of the run
pin=end
while pin>0 and (is_solve(names[pin-1]) or 'copyBuffer' in names[pin-1]): pin-=1   # first kernel of that solve (whole solves
                                                                # read the right-hand side themselves: no k_permute_in)
a=pin-1
while a>=0 and not is_solve(names[a]): a-=1
seg=rows[a+1:]
agg=collections.OrderedDict()
for r in seg:
    k=r['Kernel_Name'][:34]
    du=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    c,t,mx=agg.get(k,(0,0.0,0.0)); agg[k]=(c+1,t+du,max(mx,du))
tot=(int(seg[-1]['End_Timestamp'])-int(seg[0]['Start_Timestamp']))/1e3
print('last step: %d kernels, %.1f us wall'%(len(seg),tot))
for k,(c,t,mx) in sorted(agg.items(),key=lambda x:-x[1][1]):
    print(k.ljust(36),str(c).rjust(4),'%9.1f us total %8.1f avg %8.1f max'%(t,t/c,mx))
if len(sys.argv)>2:
    for r in seg: print(r['Kernel_Name'][:30].ljust(30),(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r.get('Grid_Size_X',r.get('Grid_Size','?')))
