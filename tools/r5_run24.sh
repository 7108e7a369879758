#!/bin/bash
# round 5, run 24: BUILD-TIME ablations of the GEMM-shaped kernel (ab/libojf_ablN.so: -DOJF_GEMM_ABL=N) on two layers, four frames
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_24; mkdir -p $O
export OJF_SEG_GEMM_MIN=1
for abl in 0 1 2 3 4 8 12 16 28 31; do
  L=$PWD/ab/libojf_abl$abl.so; [ $abl = 0 ] && L=$PWD/online_joint_depthfusion_and_semantic_amd/libojf.so
  for shape in "256 256 3 60 80 4" "512 2048 1 15 20 4"; do
    OJF_LIB_PATH=$L python tools/seg_layer_bench.py $shape 2>&1 | grep "per launch" | sed "s/^/abl $abl: /" >> $O/abl.txt
  done
done
