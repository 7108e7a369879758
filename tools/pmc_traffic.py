"""Helper: per-kernel HBM traffic per frame from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
usage: pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv n_frames out.json
n_frames = 0: the number of frames the profiled run fused = its extract_tile_kernel dispatches (one per frame on the
inference path; bench.py runs warm-up, stage-mark and timed passes, so the frame count is not steps + warmup)."""
import csv, json, sys, collections
def load(path, counter):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter: continue
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('ojf::', '').split('<')[0]
        acc[k] += float(r['Counter_Value']); n[k] += 1
    return acc, n
fetch, nf = load(sys.argv[1], 'FETCH_SIZE')
write, nw = load(sys.argv[2], 'WRITE_SIZE')
frames = int(sys.argv[3]) or nf['extract_tile_kernel']
assert frames and nw['extract_tile_kernel'] == nf['extract_tile_kernel'], 'the two passes must run the same command'
out = {'_frames': frames}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith(('conv', 'chain', 'vortex', 'dense_pair', 'dense_chain', 'entry1x1', 'extract', 'integrate', 'pool', 'colsum', 'gave', 'prepare')):
        continue
    # rocprofv3 units: KiB; gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> x2 (MI355X_MICROARCH.md §HBM)
    out[k] = {'launches_per_frame': nf[k] / frames, 'fetch_bytes_per_frame_raw': fetch[k] * 1024 / frames,
              'fetch_bytes_per_frame_x2': 2 * fetch[k] * 1024 / frames, 'write_bytes_per_frame': write[k] * 1024 / frames}
json.dump(out, open(sys.argv[4], 'w'), indent=1, sort_keys=True)
for k, v in out.items():
    if k.startswith('_'): continue
    print('%-36s launches/frame %5.1f  fetch(x2) %8.2f MB  write %8.2f MB' % (k, v['launches_per_frame'], v['fetch_bytes_per_frame_x2'] / 1e6, v['write_bytes_per_frame'] / 1e6))
