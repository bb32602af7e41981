import csv, glob, sys, os, collections
d = sys.argv[1]
f = max(glob.glob(d + '/*/*counter_collection.csv'), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
# last dispatch of each (kernel, grid) pair per counter
agg = collections.OrderedDict()
for r in rows:
    k = (r['Kernel_Name'][:44], r.get('Grid_Size', r.get('Grid_Size_X', '?')))
    agg.setdefault(k, {})[r['Counter_Name']] = float(r['Counter_Value'])   # later dispatches overwrite earlier ones
names = sorted({c for v in agg.values() for c in v})
print('kernel'.ljust(46), 'grid'.rjust(9), ' '.join(n[-18:].rjust(18) for n in names))
for (k, g), v in agg.items():
    print(k.ljust(46), str(g).rjust(9), ' '.join(('%.4g' % v.get(n, float('nan'))).rjust(18) for n in names))
