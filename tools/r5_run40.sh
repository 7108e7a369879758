#!/bin/bash
# round 5, run 40: GEMM-shaped form with the 128x160 tile where it saves a round of blocks: bits per forced tile, engine at 1 / 4 / 8 frames against the two-tile rule
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_40; mkdir -p $O
for k in 0 1 2; do
  OJF_SEG_GEMM_SHAPE=$k OJF_SEG_GEMM22_MIN=1 OJF_SEG_GEMM_MIN_KB=1 python -m pytest tests/test_segconv_gpu.py -x -q -m gpu 2>&1 | tail -1 | sed "s/^/shape $k: /" >> $O/tests.txt
done
python -m pytest tests/test_segconv_gpu.py tests/test_adapnet_engine_gpu.py -x -q -m gpu 2>&1 | tail -1 | sed "s/^/rule: /" >> $O/tests.txt
for B in 1 4 8; do for rep in 1 2; do
python tools/seg_probe.py graph 40 240 320 $B 2>&1 | grep "seg engine" | sed "s/^/three /" >> $O/probe.txt
OJF_SEG_GEMM_MENU=3 python tools/seg_probe.py graph 40 240 320 $B 2>&1 | grep "seg engine" | sed "s/^/two   /" >> $O/probe.txt
done; done
OJF_SEG_TRACE=1 python tools/seg_probe.py eager 1 240 320 4 2>&1 | grep "^segconv" | tail -90 > $O/trace_b4.txt
