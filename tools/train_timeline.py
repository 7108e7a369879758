"""Profiling helper (not a test): phases of one steady-state training frame from a rocprofv3 kernel trace
(python tools/train_timeline.py <kernel_trace.csv>): spans of forward / loss / backward / rest, busy time per phase and
stream, idle gaps."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'train_pack_input' in r['Kernel_Name']]
f0 = starts[len(starts) // 2]
f1 = starts[len(starts) // 2 + 1]
# a frame = [extract ... next frame's first extract); walk back from pack_input to the two extract launches
i0 = f0
while i0 > 0 and 'nonzero' not in rows[i0]['Kernel_Name'].lower() and i0 > f0 - 40:
    i0 -= 1
i1 = f1
while i1 > 0 and 'nonzero' not in rows[i1]['Kernel_Name'].lower() and i1 > f1 - 40:
    i1 -= 1
fr = rows[i0:i1]
t0 = int(fr[0]['Start_Timestamp'])
print('frame: %d launches, %.3f ms' % (len(fr), (int(rows[i1]['Start_Timestamp']) - t0) / 1e6))
def name(r):
    n = r['Kernel_Name']
    return n.split('(')[0].replace('void ', '').replace('ojf::', '')[:44]
marks = {}
for r in fr:
    n = r['Kernel_Name']
    for key in ('train_pack_input', 'train_planes_to_nchw', 'train_nchw_to_planes', 'integrate_accumulate', 'integrate_finalize'):
        if key in n and key not in marks:
            marks[key] = (int(r['Start_Timestamp']) - t0) / 1e3
last_wg = max(((int(r['End_Timestamp']) - t0) / 1e3 for r in fr if 'wgrad' in r['Kernel_Name']), default=0)
last_convT = max(((int(r['End_Timestamp']) - t0) / 1e3 for r in fr if 'conv_mfma' in r['Kernel_Name']), default=0)
print('marks (us from frame start):', {k: round(v, 1) for k, v in marks.items()}, 'last wgrad end', round(last_wg, 1), 'last conv end', round(last_convT, 1))
by_q = collections.defaultdict(float)
for r in fr:
    by_q[r.get('Queue_Id', '?')] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print('busy us per queue:', {k: round(v, 1) for k, v in by_q.items()})
# idle gaps on the union of all queues
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in fr)
cur_end, idle, gaps = ev[0][1], 0, []
for s, e in ev[1:]:
    if s > cur_end:
        idle += s - cur_end
        gaps.append((s - cur_end, (cur_end - t0) / 1e3))
    cur_end = max(cur_end, e)
print('device idle inside the frame: %.1f us in %d gaps; largest:' % (idle / 1e3, len(gaps)), [(round(g / 1e3, 1), round(at, 1)) for g, at in sorted(gaps, reverse=True)[:8]])
acc = collections.defaultdict(lambda: [0, 0.0])
for r in fr:
    a = acc[name(r)]
    a[0] += 1
    a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:32]:
    print('  %-46s x%3d %8.1f us' % (k, v[0], v[1]))
