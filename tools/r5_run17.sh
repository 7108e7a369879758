#!/bin/bash
# round 5, run 17: heterogeneous multi-launches in the 2-D engine - bits and A/B (OJF_SEG_NO_MULTI=1)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_17; mkdir -p $O
python -m pytest tests/test_segconv_gpu.py tests/test_adapnet_engine_gpu.py -x -q -m gpu 2>&1 | tail -8 > $O/tests.txt
python -m pytest tests/test_headline_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu -k "predict or segmentation" 2>&1 | tail -4 >> $O/tests.txt
for i in 1 2 3; do
OJF_SEG_NO_MULTI=1 python tools/seg_probe.py graph 50 2>&1 | grep "seg engine" | sed 's/^/separate /' >> $O/probe.txt
python tools/seg_probe.py graph 50 2>&1 | grep "seg engine" | sed 's/^/multi    /' >> $O/probe.txt
done
OJF_SEG_NO_MULTI=1 python tools/seg_probe.py graph 30 240 320 4 2>&1 | grep "seg engine" | sed 's/^/separate /' >> $O/probe.txt
python tools/seg_probe.py graph 30 240 320 4 2>&1 | grep "seg engine" | sed 's/^/multi    /' >> $O/probe.txt
OJF_SEG_TRACE=1 python tools/seg_probe.py eager 1 2> $O/trace.txt > /dev/null
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python tools/seg_probe.py graph 10 > /dev/null 2> $O/kt.err
python tools/seg_seq.py $(find $O/kt -name '*kernel_trace.csv' | head -1) > $O/seq.txt 2>&1
rm -rf $O/kt
