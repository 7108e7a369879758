#!/bin/bash
# round 5, run 36: build-time ablations of the GEMM-shaped form, loads-only side: 27 = global loads alone, 11 = + barrier, 19 = + staging (no barrier), 3 = loads + staging + barrier
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_36; mkdir -p $O
for shape in "256 256 3 60 80 1" "256 256 3 60 80 4"; do
  for L in online_joint_depthfusion_and_semantic_amd/libojf.so ab/libojf_abl27.so ab/libojf_abl11.so ab/libojf_abl19.so ab/libojf_abl3.so; do
    OJF_LIB_PATH=$PWD/$L python tools/seg_layer_bench.py $shape 2>&1 | grep "per launch" | sed "s|$| $L|" >> $O/p.txt
  done
done
