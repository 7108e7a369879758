"""usage: pmc_calibrate_summary.py <FETCH counter_collection.csv> <WRITE counter_collection.csv>: counter value per launch
of the calibration kernels next to the bytes they are known to move (tools/pmc_calibrate.py)."""
import collections, csv, sys
def per_kernel(path):
    acc, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'][:70]
        acc[k] += float(r['Counter_Value']); n[k] += 1
    return {k: (acc[k] / n[k], n[k]) for k in acc}
f, w = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
for k in sorted(set(f) | set(w)):
    if f.get(k, (0, 0))[1] < 5 and w.get(k, (0, 0))[1] < 5:
        continue
    print('%-72s launches %3d  FETCH_SIZE %10.3f  WRITE_SIZE %10.3f  (counter units per launch; MB if the unit is KB: /1024)' % (
        k, max(f.get(k, (0, 0))[1], w.get(k, (0, 0))[1]), f.get(k, (0, 0))[0], w.get(k, (0, 0))[0]))
print('known bytes per launch: copy 256 MiB read + 256 MiB written; fill 256 MiB written; index_copy 64 MiB read + 64 MiB written (32-byte rows)')
