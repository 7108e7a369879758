#!/bin/bash
# round 5, run 5: SEGCONV engine small-op cleanup (pool_fc, in-engine dropout, zero-pad flag, vector softmax)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_5; mkdir -p $O
python -m pytest tests/test_segconv_gpu.py tests/test_adapnet_engine_gpu.py -x -q -m gpu 2>&1 | tail -15 > $O/tests.txt
python -m pytest tests/test_headline_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu -k "predict or segmentation" 2>&1 | tail -15 >> $O/tests.txt
python tools/seg_probe.py graph 50 2>&1 | grep -v amdgpu.ids >> $O/probe.txt
python tools/seg_probe.py graph 50 2>&1 | grep -v amdgpu.ids >> $O/probe.txt
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python tools/seg_probe.py graph 10 > /dev/null 2> $O/kt.err
python tools/seg_seq.py $(find $O/kt -name '*kernel_trace.csv' | head -1) > $O/seq.txt 2>&1
rm -rf $O/kt
