"""Helper: summarise a rocprofv3 --pmc counter_collection.csv per kernel-dispatch index within the last forward."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# columns: Dispatch_Id, Kernel_Name, Counter_Name, Counter_Value, ...
by = collections.OrderedDict()
for r in rows:
    k = int(r['Dispatch_Id'])
    by.setdefault(k, {'name': r['Kernel_Name'].split('(')[0].replace('void ', '').replace('ojf::', '')[:34], 'grid': r.get('Grid_Size', ''), 'vgpr': r.get('VGPR_Count', '')})
    by[k][r['Counter_Name']] = by[k].get(r['Counter_Name'], 0) + float(r['Counter_Value'])
keys = sorted(by)
ctrs = sorted({r['Counter_Name'] for r in rows})
n = len(keys)
last = keys[-(n // int(sys.argv[2])):] if len(sys.argv) > 2 else keys
print('idx name grid vgpr', ' '.join(ctrs))
for i, k in enumerate(last):
    d = by[k]
    print(i, d['name'], d['grid'], d['vgpr'], ' '.join('%.3g' % d.get(c, 0) for c in ctrs))
