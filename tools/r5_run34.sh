#!/bin/bash
# round 5, run 34: GEMM-shaped form with the taps as the INNER K walk (timing-only ablation): do a tap's pixel lines hit in L1 when the previous K block read its neighbours?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_34; mkdir -p $O
for shape in "256 256 3 60 80 1" "256 256 3 60 80 4" "128 128 3 60 80 1" "256 256 3 30 40 4"; do
  OJF_SEG_GEMM22_MIN=1 python tools/seg_layer_bench.py $shape 2>&1 | grep "per launch" >> $O/abl.txt
  OJF_SEG_GEMM22_MIN=1 OJF_LIB_PATH=$PWD/ab/libojf_abl64.so python tools/seg_layer_bench.py $shape 2>&1 | grep "per launch" | sed "s/$/ TAPS INNER/" >> $O/abl.txt
done
