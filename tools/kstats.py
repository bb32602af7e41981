import csv,glob,sys
d=sys.argv[1]
import os
f=max(glob.glob(d+'/*/*_kernel_stats.csv'), key=os.path.getmtime)
for r in list(csv.DictReader(open(f)))[:12]:
    print(r['Name'][:50].ljust(50), r['Calls'].rjust(6), ('%.2f'%(float(r['TotalDurationNs'])/1e6)).rjust(9),'ms', ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(9),'us  max', ('%.1f'%(float(r['MaxNs'])/1e3)).rjust(8), r['Percentage'])
