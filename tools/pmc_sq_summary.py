"""Helper: per-kernel SQ counter summary from one rocprofv3 --pmc pass of bench.py
(SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES).
usage: pmc_sq_summary.py counter_collection.csv out.json
WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (quad-cycles, MI355X_MICROARCH.md): the shares say whether
waves sit parked on s_waitcnt / barriers, stall at issue, or issue instructions."""
import collections, csv, json, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').replace('ojf::', '')
    if not k.startswith(('conv', 'chain', 'vortex', 'dense_pair', 'dense_chain', 'entry1x1', 'extract', 'integrate', 'pool', 'colsum', 'gave', 'prepare', 'segconv', 'seg_', 'mesh', 'points', 'train_')):
        continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVES':
        n[k] += 1
out = {}
for k, c in sorted(acc.items()):
    wc = c['SQ_WAVE_CYCLES'] or 1.0
    out[k] = {'dispatches': n[k], 'waves_per_dispatch': c['SQ_WAVES'] / max(n[k], 1),
              'wait_any_share': c['SQ_WAIT_ANY'] / wc, 'wait_inst_share': c['SQ_WAIT_INST_ANY'] / wc,
              'active_inst_share': c['SQ_ACTIVE_INST_ANY'] / wc, 'active_valu_share': c['SQ_ACTIVE_INST_VALU'] / wc,
              'mfma_busy_cycles_per_dispatch': c['SQ_VALU_MFMA_BUSY_CYCLES'] / max(n[k], 1),
              'sq_busy_cycles_per_dispatch': c['SQ_BUSY_CYCLES'] / max(n[k], 1),
              # MFMA pipe time per dispatch if spread evenly over the 1024 SIMDs at the nominal 2.4 GHz: compare with the launch's duration
              'mfma_busy_us_per_simd_per_dispatch': c['SQ_VALU_MFMA_BUSY_CYCLES'] / max(n[k], 1) / 1024 / 2400.0}
    o = out[k]
    print('%-44s parked %.2f  issue-stall %.2f  issuing %.2f (VALU %.2f)  MFMA pipe %.2f us/SIMD/dispatch' %
          (k[:44], o['wait_any_share'], o['wait_inst_share'], o['active_inst_share'], o['active_valu_share'], o['mfma_busy_us_per_simd_per_dispatch']))
json.dump(out, open(sys.argv[2], 'w'), indent=1, sort_keys=True)
