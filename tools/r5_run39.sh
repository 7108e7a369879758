#!/bin/bash
# round 5, run 39: tile menu after making the short-share guards compile-time
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_39; mkdir -p $O
for shape in "256 256 3 60 80 1" "256 256 3 60 80 4" "128 128 3 60 80 1"; do
  for k in 0 1 2 3 4 5; do
    OJF_SEG_GEMM_SHAPE=$k OJF_SEG_GEMM22_MIN=1 python tools/seg_layer_bench.py $shape 2>&1 | grep "per launch" >> $O/layers.txt
  done
done
for B in 1 4; do
python tools/seg_probe.py graph 40 240 320 $B 2>&1 | grep "seg engine" | sed "s/^/menu  /" >> $O/probe.txt
OJF_SEG_GEMM_MENU=3 python tools/seg_probe.py graph 40 240 320 $B 2>&1 | grep "seg engine" | sed "s/^/two   /" >> $O/probe.txt
done
