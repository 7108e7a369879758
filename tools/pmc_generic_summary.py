"""Helper: per-kernel sums of arbitrary rocprofv3 --pmc counters per dispatch (python tools/pmc_generic_summary.py counter_collection.csv)."""
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').replace('ojf::', '')[:44]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    n[k][r['Counter_Name']] += 1
names = sorted({c for k in acc for c in acc[k]})
print('%-44s %6s ' % ('kernel', 'disp') + ' '.join('%16s' % c[-16:] for c in names))
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    d = max(n[k].values())
    if d < 3: continue
    print('%-44s %6d ' % (k, d) + ' '.join('%16.0f' % (acc[k][c] / d) for c in names))
