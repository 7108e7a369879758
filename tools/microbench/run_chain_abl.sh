# profiling helper: per-kernel times of the chain/tail kernels with parts of chain_layer ablated.
# Round-1 result (vortex_tail_kernel, 47.3 us): no MFMA 38.9, no LDS weight reads 43.8, no weight DMA 39.3, none of
# the three 25.9 us - the skeleton (fp16 splits, per-layer bias/activation/guard epilogues with their global bias
# loads, 16 barriers) is more than half of the kernel; LDS bandwidth and the MFMA pipe are NOT its bounds.
# The libraries are built from ojf_net.hip with three #if blocks added around the MFMA, the LDS weight reads and the
# DMA in chain_layer (not kept in the product source).
# (libojf_abl{N}.bin are builds of ojf_net.hip with -DOJF_CHAIN_ABL=N: 1 no MFMA, 2 no LDS weight reads, 4 no weight DMA)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cp online_joint_depthfusion_and_semantic_amd/libojf.so /tmp/libojf_keep.so
for a in 0 1 2 4 3 7; do
  cp tools/microbench/libojf_abl$a.bin online_joint_depthfusion_and_semantic_amd/libojf.so
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl$a -o kt -- python bench.py --steps 40 --warmup 5 --cpu-frames 0 > /dev/null 2>&1
  python -c "
import csv
for r in csv.DictReader(open('/tmp/abl$a/kt_kernel_stats.csv')):
    if 'vortex_tail' in r['Name'] or 'chain1x1' in r['Name']: print('ABL=$a', r['Name'][:60], round(float(r['AverageNs'])/1e3,1), 'us')
"
done
cp /tmp/libojf_keep.so online_joint_depthfusion_and_semantic_amd/libojf.so
