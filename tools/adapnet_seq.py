"""Profiling helper (not a test): the launches of one graph replay of the SEGCONV engine (predict path, 320x240) in start
order with queue, grid and duration - the critical path of AdapNet++ (python tools/adapnet_seq.py <kernel_trace.csv>)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'seg_pack_input' in r['Kernel_Name']]
# two pack launches per forward (image, depth): take a forward near the end (graph replays)
f0 = starts[-4]
f1 = starts[-2]
t0 = int(rows[f0]['Start_Timestamp'])
busy = {}
prev_end = {}
for r in rows[f0:f1]:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('ojf::', '').replace('(anonymous namespace)::', '')
    q = r.get('Queue_Id', '?')
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = e
    busy[q] = busy.get(q, 0) + (e - s) / 1e3
    print('%8.1f us  %6.1f us  gap %5.1f  q%s  grid %5d x %3s x %2s  wg %4s  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, q,
          int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), r['Grid_Size_Y'], r['Grid_Size_Z'], r['Workgroup_Size_X'], n[:48]))
print('span %.1f us, busy per queue %s, launches %d' % ((int(rows[f1]['Start_Timestamp']) - t0) / 1e3, {k: round(v, 1) for k, v in busy.items()}, f1 - f0))
