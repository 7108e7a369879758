#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_26; mkdir -p $O
for shape in "256 256 3 60 80 4" "512 2048 1 15 20 4" "256 256 3 60 80 1" "2048 512 1 15 20 4" "512 256 3 15 20 4" "1024 256 1 15 20 4"; do
  for t in 0 1 2; do OJF_SEG_GEMM_TILE=$t OJF_SEG_GEMM_MIN=1 python tools/seg_layer_bench.py $shape 2>&1 | grep "per launch" >> $O/layers.txt; done
  python tools/seg_layer_bench.py $shape 2>&1 | grep "per launch" >> $O/layers.txt
done
