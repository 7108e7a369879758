#!/bin/bash
# round 5, run 33: 64x64 GEMM-shaped form with two fragment sets (next block's LDS reads + the staging of the one after under this block's MFMAs) against the plain order
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_33; mkdir -p $O
python -m pytest tests/test_segconv_gpu.py tests/test_adapnet_engine_gpu.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt
NP=$PWD/ab/libojf_nopipe.so
for shape in "256 256 3 60 80 1" "128 128 3 60 80 1" "512 2048 1 15 20 4" "2048 512 1 15 20 4" "64 64 3 60 80 1" "64 256 1 60 80 1" "256 64 1 60 80 4"; do
  python tools/seg_layer_bench.py $shape 2>&1 | grep "per launch" >> $O/layers.txt
  OJF_LIB_PATH=$NP python tools/seg_layer_bench.py $shape 2>&1 | grep "per launch" | sed "s/$/ PLAIN ORDER/" >> $O/layers.txt
done
for B in 1 4; do for rep in 1 2; do
python tools/seg_probe.py graph 40 240 320 $B 2>&1 | grep "seg engine" | sed "s/^/pipe  /" >> $O/probe.txt
OJF_LIB_PATH=$NP python tools/seg_probe.py graph 40 240 320 $B 2>&1 | grep "seg engine" | sed "s/^/plain /" >> $O/probe.txt
done; done
