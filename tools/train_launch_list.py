"""Helper: per-launch list of ONE training frame out of a rocprofv3 kernel trace (python tools/train_launch_list.py kernel_trace.csv):
start (us from the frame's first kernel), duration, grid, block, LDS, queue, kernel name."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: r['Kernel_Name'].replace('void ', '').replace('ojf::', '').replace('(anonymous namespace)::', '').split('(')[0]
idx = [i for i, r in enumerate(rows) if 'train_pack_input' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]  # a frame well inside the run
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    print('%8.1f us %7.1f us  grid %7s x %4s x %2s  wg %4s  lds %6s  q%s  %s' % (
        (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
        r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], r['Workgroup_Size_X'], r.get('LDS_Block_Size', '?'), r['Queue_Id'], name(r)[:70]))
