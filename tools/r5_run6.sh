#!/bin/bash
# round 5, run 6: A/B on one box - previous commit (ab/prev) against the working tree, SEGCONV engine
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_6; mkdir -p $O
for i in 1 2 3; do
python ab/prev/tools/seg_probe.py graph 50 2>&1 | grep -v amdgpu.ids | sed 's/^/prev /' >> $O/probe.txt
python tools/seg_probe.py graph 50 2>&1 | grep -v amdgpu.ids | sed 's/^/new  /' >> $O/probe.txt
done
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python ab/prev/tools/seg_probe.py graph 10 > /dev/null 2> $O/kt.err
python tools/seg_seq.py $(find $O/kt -name '*kernel_trace.csv' | head -1) > $O/seq_prev.txt 2>&1
rm -rf $O/kt
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python tools/seg_probe.py graph 10 > /dev/null 2> $O/kt.err
python tools/seg_seq.py $(find $O/kt -name '*kernel_trace.csv' | head -1) > $O/seq_new.txt 2>&1
rm -rf $O/kt
