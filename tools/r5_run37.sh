#!/bin/bash
# round 5, run 37: does the block count's remainder over 256 CUs set the time of the 64x64 GEMM-shaped form?  256 -> 256 3x3 on maps of 48..70 rows of 80 pixels
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_37; mkdir -p $O
for h in 24 32 40 48 51 52 56 60 64 70 76 80 96 102 104; do
  OJF_SEG_GEMM22_MIN=1 OJF_SEG_GEMM_MIN=100000 python tools/seg_layer_bench.py 256 256 3 $h 80 1 2>&1 | grep "per launch" | sed "s/$/ blocks $(( (h*80+63)/64*4 ))/" >> $O/q.txt
done
