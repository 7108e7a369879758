#!/bin/bash
# round 5, run 44: per-launch timeline of the 2-D engine at four and eight frames per pass (final kernels)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_44; mkdir -p $O
for B in 4 8; do
OJF_SEG_TRACE=1 python tools/seg_probe.py eager 1 240 320 $B 2>&1 | grep "^segconv" | tail -100 > $O/trace_b$B.txt
rocprofv3 --kernel-trace --output-format csv -d $O/kt$B -o kt -- python tools/seg_probe.py graph 10 240 320 $B > /dev/null 2> $O/kt$B.err
SEG_PACKS=$((2*B)) python tools/seg_seq.py $(find $O/kt$B -name '*kernel_trace.csv' | head -1) > $O/seq_b$B.txt 2>&1
rm -rf $O/kt$B
done
