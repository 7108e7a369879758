#!/bin/bash
# round 5, run 35: K blocks in flight per wave (2 / 3 / 4) in the 64x64 GEMM-shaped form
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_35; mkdir -p $O
for shape in "256 256 3 60 80 1" "128 128 3 60 80 1" "512 2048 1 15 20 4" "256 256 3 30 40 4"; do
  for L in online_joint_depthfusion_and_semantic_amd/libojf.so ab/libojf_p3_pipe0.so ab/libojf_p4_pipe0.so ab/libojf_p4_pipe1.so; do
    OJF_SEG_GEMM22_MIN=1 OJF_LIB_PATH=$PWD/$L python tools/seg_layer_bench.py $shape 2>&1 | grep "per launch" | sed "s|$| $L|" >> $O/p.txt
  done
done
