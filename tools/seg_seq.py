"""Profiling helper: one forward of tools/seg_probe.py in launch order from rocprofv3 csv files -
python tools/seg_seq.py <kernel_trace.csv> [<counter_collection.csv of an eager run> ...]: start, duration, gap, grid, kernel
(+ per-dispatch counter values matched by position inside a forward)."""
import csv, sys, collections, os
def short(n): return n.replace('(anonymous namespace)::', '').replace('void ', '').replace('ojf::', '').replace('at::native::', '').split('(')[0]
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'seg_pack_input' in r['Kernel_Name']]
k = int(os.environ.get('SEG_PACKS', 2))  # pack launches per forward (2 x frames per pass)
f0, f1 = starts[-2 * k], starts[-k]
ctr = []
for path in sys.argv[2:]:
    per = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        per.setdefault(int(r['Dispatch_Id']), [short(r['Kernel_Name']), {}])[1][r['Counter_Name']] = float(r['Counter_Value'])
    d = [v for k, v in sorted(per.items())]
    st = [i for i, v in enumerate(d) if 'seg_pack_input' in v[0]]
    ctr.append(d[st[-2 * k]:st[-k]])
t0 = int(rows[f0]['Start_Timestamp']); prev = None; tot = 0
for j, r in enumerate(rows[f0:f1]):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev else 0.0
    prev = e; tot += e - s
    extra = ''
    for c in ctr:
        if j < len(c) and c[j][0] == short(r['Kernel_Name']):
            extra += ' ' + ' '.join('%s=%.0f' % (k[-10:], v) for k, v in c[j][1].items())
    print('%7.1f us %6.1f us gap %4.1f grid %4d x %3s x %2s  %-34s%s' % ((s - t0) / 1e3, (e - s) / 1e3, gap,
          int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), r['Grid_Size_Y'], r['Grid_Size_Z'], short(r['Kernel_Name'])[:34], extra))
print('span %.1f us, busy %.1f us, launches %d' % ((int(rows[f1]['Start_Timestamp']) - t0) / 1e3, tot / 1e3, f1 - f0))
