"""Helper: from a rocprofv3 kernel trace, how much of the 2-D network's kernel time (seg* kernels) runs while a kernel of another queue is running."""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t_end = int(rows[-1]['End_Timestamp']); t0 = int(rows[0]['Start_Timestamp'])
rows = [r for r in rows if int(r['Start_Timestamp']) > t0 + 0.6 * (t_end - t0)]  # the timed part
qs = collections.Counter((r['Queue_Id'], 'seg' if 'seg' in r['Kernel_Name'] else 'other') for r in rows)
print('kernels per (queue, kind):', dict(qs))
seg = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id']) for r in rows if 'seg' in r['Kernel_Name']]
oth = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id']) for r in rows if 'seg' not in r['Kernel_Name']]
tot = sum(e - s for s, e, _ in seg); ov = 0; j = 0
for s, e, q in seg:
    while j < len(oth) and oth[j][1] <= s: j += 1
    k = j
    while k < len(oth) and oth[k][0] < e:
        if oth[k][2] != q: ov += max(0, min(e, oth[k][1]) - max(s, oth[k][0]))
        k += 1
print('2-D network kernel time %.1f us, of which beside a kernel of another queue: %.1f us (%.1f %%)' % (tot / 1e3, ov / 1e3, 100.0 * ov / max(tot, 1)))
span = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp']); busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
print('span %.1f us, sum of kernel durations %.1f us' % (span / 1e3, busy / 1e3))
