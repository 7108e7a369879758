"""Profiling helper (not a test): per-launch durations of one frame's kernels from a rocprofv3 kernel trace CSV."""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'extract_kernel' in r['Kernel_Name']]
per = {}
for f in range(10, len(idx) - 1):
    seg = rows[idx[f]:idx[f + 1]]
    for j, r in enumerate(seg):
        per.setdefault(j, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
seg = rows[idx[10]:idx[11]]
tot = 0
for j, r in enumerate(seg):
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('ojf::', '')[:40]
    med = float(np.median(per[j]))
    tot += med
    print('%2d %-40s grid=(%s,%s) %7.1f us' % (j, name, r['Grid_Size_X'], r['Grid_Size_Y'], med))
print('sum %.1f us; span %.1f us' % (tot, (int(rows[idx[11]]['Start_Timestamp']) - int(rows[idx[10]]['Start_Timestamp'])) / 1e3))
