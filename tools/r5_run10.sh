#!/bin/bash
# round 5, run 10: member index from blockIdx.z (no dependent scalar load) - A/B against the committed build, bits
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_10; mkdir -p $O
python -m pytest tests/test_segconv_gpu.py tests/test_adapnet_engine_gpu.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt
for i in 1 2 3; do
python ab/prev/tools/seg_probe.py graph 50 2>&1 | grep -v amdgpu.ids | sed 's/^/prev /' >> $O/probe.txt
python tools/seg_probe.py graph 50 2>&1 | grep -v amdgpu.ids | sed 's/^/new  /' >> $O/probe.txt
done
