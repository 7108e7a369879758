#!/bin/bash
# round 5, run 25: GEMM-shaped kernel with hoisted LDS operand reads, two waves per SIMD - layer times, bits, engine
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_25; mkdir -p $O
for shape in "256 256 3 60 80 4" "512 2048 1 15 20 4" "256 256 3 60 80 1" "2048 512 1 15 20 4" "512 256 3 15 20 4"; do
  OJF_SEG_GEMM_MIN=1 python tools/seg_layer_bench.py $shape 2>&1 | grep "per launch" >> $O/layers.txt
  python tools/seg_layer_bench.py $shape 2>&1 | grep "per launch" >> $O/layers.txt
done
OJF_SEG_GEMM_MIN=1 OJF_SEG_GEMM_MIN_KB=1 python -m pytest tests/test_segconv_gpu.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt
run() { env "$@" python tools/seg_probe.py graph 40 240 320 $B 2>&1 | grep "seg engine" | sed "s/^/$* /" >> $O/probe.txt; }
for B in 1 4; do
run OJF_SEG_GEMM_MIN=1000000
run OJF_SEG_GEMM_MIN=512
run OJF_SEG_GEMM_MIN=256
run OJF_SEG_GEMM_MIN=128
run OJF_SEG_GEMM_MIN=64
done
