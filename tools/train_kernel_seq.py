"""Profiling helper (not a test): the kernels of one steady-state training frame in launch order with grid and duration
(python tools/train_kernel_seq.py <kernel_trace.csv> [filter])."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ''
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'train_pack_input' in r['Kernel_Name']]
f0, f1 = starts[len(starts) // 2], starts[len(starts) // 2 + 1]
t0 = int(rows[f0]['Start_Timestamp'])
for r in rows[f0:f1]:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('ojf::', '')
    if flt and flt not in n:
        continue
    print('%9.1f us  %7.1f us  grid %6s x %4s x %3s  q%s  %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
          int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), r['Grid_Size_Y'], r['Grid_Size_Z'], r.get('Queue_Id', '?'), n[:60]))
