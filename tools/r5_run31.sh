#!/bin/bash
# round 5, run 31: is the 64x64 GEMM-shaped form waiting on ONE L2 channel (every pixel block of a channel block reads the same weight lines at the same time)?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_31; mkdir -p $O
for shape in "256 256 3 60 80 1" "256 256 3 60 80 4" "128 128 3 60 80 1" "512 2048 1 15 20 4"; do
  python tools/seg_layer_bench.py $shape 2>&1 | grep "per launch" >> $O/abl.txt
  OJF_LIB_PATH=$PWD/ab/libojf_abl32.so python tools/seg_layer_bench.py $shape 2>&1 | grep "per launch" | sed "s/$/ ABL 32/" >> $O/abl.txt
done
