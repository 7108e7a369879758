#!/bin/bash
# round 5, run 45: 128x80 tile + the long-K tile model: bits (forced shape 3), engine at 1 / 2 / 4 / 8 frames with and without the model
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_45; mkdir -p $O
OJF_SEG_GEMM_SHAPE=3 OJF_SEG_GEMM22_MIN=1 OJF_SEG_GEMM_MIN_KB=1 python -m pytest tests/test_segconv_gpu.py -x -q -m gpu 2>&1 | tail -1 | sed "s/^/shape 3: /" >> $O/tests.txt
python -m pytest tests/test_segconv_gpu.py tests/test_adapnet_engine_gpu.py -x -q -m gpu 2>&1 | tail -1 | sed "s/^/model: /" >> $O/tests.txt
for B in 1 2 4 8; do for rep in 1 2; do
python tools/seg_probe.py graph 40 240 320 $B 2>&1 | grep "seg engine" | sed "s/^/model /" >> $O/probe.txt
OJF_SEG_GEMM_MENU=7 python tools/seg_probe.py graph 40 240 320 $B 2>&1 | grep "seg engine" | sed "s/^/rules /" >> $O/probe.txt
done; done
