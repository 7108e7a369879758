#!/bin/bash
# round 5, run 20: per-launch timeline of the 2-D engine at four frames per pass
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_20; mkdir -p $O
OJF_SEG_TRACE=1 python tools/seg_probe.py eager 1 240 320 4 2> $O/trace.txt > /dev/null
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python tools/seg_probe.py graph 10 240 320 4 > /dev/null 2> $O/kt.err
SEG_PACKS=8 python tools/seg_seq.py $(find $O/kt -name '*kernel_trace.csv' | head -1) > $O/seq.txt 2>&1
rm -rf $O/kt
