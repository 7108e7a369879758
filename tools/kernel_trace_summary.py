"""Per-kernel time per frame from a rocprofv3 --kernel-trace CSV: groups the dispatches of the steady-state frames by
kernel name AND grid size (the dense_pair / conv launches of different layers share a name), prints mean microseconds.
usage: kernel_trace_summary.py kernel_trace.csv n_frames(0 = count the extract dispatches) [index of the frame to list]"""
import csv
import sys
from collections import defaultdict

path, frames = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
if frames == 0:  # the frames the traced run fused = its extract dispatches (one per frame on the inference path)
    frames = sum(1 for r in rows if 'extract_tile_kernel' in r['Kernel_Name'])
acc = defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('ojf::', '')
    acc[(name, r.get('Grid_Size_X', ''), r.get('Grid_Size_Y', ''), r.get('LDS_Block_Size', ''))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = 0.0
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    per_frame = sum(v) / 1e3 / frames
    tot += per_frame
    print('%-60s grid %6s x%2s lds %6s  calls/frame %5.2f  mean %7.2f us  per frame %7.2f us' % (k[0][:60], k[1], k[2], k[3], len(v) / frames, sum(v) / len(v) / 1e3, per_frame))
print('sum of kernel time per frame: %.1f us' % tot)
if len(sys.argv) > 3:  # dispatch sequence of one steady-state frame, in start order (the frame that begins at the Nth extract)
    seq = sorted(rows, key=lambda r: int(r['Start_Timestamp']))
    starts = [i for i, r in enumerate(seq) if 'extract' in r['Kernel_Name']]
    i0, i1 = starts[int(sys.argv[3])], starts[int(sys.argv[3]) + 1]
    t0 = int(seq[i0]['Start_Timestamp'])
    prev_end = t0
    for r in seq[i0:i1]:
        st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('ojf::', '')[:48]
        print('  t=%8.2f us  dur %7.2f  gap %6.2f  q%-3s %s' % ((st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3, r.get('Queue_Id', '?'), name))
        prev_end = max(prev_end, en)
    print('  frame span %.2f us' % ((int(seq[i1]['Start_Timestamp']) - t0) / 1e3))
