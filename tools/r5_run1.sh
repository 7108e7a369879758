#!/bin/bash
# round 5, run 1: where does the SEGCONV engine's frame go? per-launch timeline + L2 fetch bytes per launch
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_1; mkdir -p $O
python tools/seg_probe.py graph 30 > $O/probe.txt 2>&1
python tools/seg_probe.py eager 30 >> $O/probe.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python tools/seg_probe.py graph 10 > /dev/null 2> $O/kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -o pf -- python tools/seg_probe.py eager 4 > /dev/null 2> $O/pf.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/ph -o ph -- python tools/seg_probe.py eager 4 > /dev/null 2> $O/ph.err
python tools/seg_seq.py $(find $O/kt -name '*kernel_trace.csv' | head -1) $(find $O/pf -name '*counter_collection.csv' | head -1) $(find $O/ph -name '*counter_collection.csv' | head -1) > $O/seq.txt 2>&1
rm -rf $O/kt $O/pf $O/ph
