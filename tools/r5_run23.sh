#!/bin/bash
# round 5, run 23: ablations of the GEMM-shaped kernel on one decoder layer (256 -> 256 3x3, 60x80, four frames)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_23; mkdir -p $O
export OJF_SEG_GEMM_MIN=1
for abl in 0 1 2 3 4 8 12 16 28 31; do OJF_SEG_ABL=$abl python tools/seg_layer_bench.py 256 256 3 60 80 4 2>&1 | grep "per launch" >> $O/abl.txt; done
for abl in 0 1 3 4 12 31; do OJF_SEG_ABL=$abl python tools/seg_layer_bench.py 512 2048 1 15 20 4 2>&1 | grep "per launch" >> $O/abl.txt; done
