# Kernel statistics of the training frame step (through gpurun): gpurun_out/train/train_kernel_stats.csv + a top list
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/train
rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python tools/train_throughput.py > $O/throughput_under_rocprof.txt 2> $O/kt.err
F=$(find $O/kt -name '*kernel_stats.csv' | head -1)
if [ -n "$F" ]; then cp $F $O/train_kernel_stats.csv; fi
T=$(find $O/kt -name '*kernel_trace.csv' | head -1)
python - "$T" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'wgrad_mfma' in r['Kernel_Name']:
        acc[(r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print('train_wgrad_mfma_kernel by grid (threads x = 64 * slabs, y = 32x32 tiles, z = taps):')
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    slabs, tiles, taps = int(k[0]) // 64, int(k[1]), int(k[2])
    ideal = tiles * taps * 38400 * 64 / 1024 / 2400.0
    print('  slabs %4d tiles %3d taps %d  calls/frame %5.2f  mean %7.1f us  (MFMA-bound %.1f us)' % (slabs, tiles, taps, len(v) / 48, sum(v) / len(v) / 1e3, ideal))
PY
rm -rf $O/kt
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/train/train_kernel_stats.csv')))
frames = 48
print('kernel ms/frame %.2f, launches/frame %.0f' % (sum(int(r['TotalDurationNs']) for r in rows) / frames / 1e6, sum(int(r['Calls']) for r in rows) / frames))
for r in rows[:22]:
    print('%-64s calls/f %6.1f  us/f %8.1f  avg %7.1f' % (r['Name'][:64], int(r['Calls']) / frames, int(r['TotalDurationNs']) / frames / 1e3, float(r['AverageNs']) / 1e3))
PY
