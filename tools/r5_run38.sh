#!/bin/bash
# round 5, run 38: the GEMM-shaped form's tile menu (64x64, 128x128, 64x80, 64x96, 64x160, 128x160): bits per forced entry, layer times per entry, engine with the model's choice
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_38; mkdir -p $O
for k in 0 1 2 3 4 5; do
  OJF_SEG_GEMM_SHAPE=$k OJF_SEG_GEMM22_MIN=1 OJF_SEG_GEMM_MIN_KB=1 python -m pytest tests/test_segconv_gpu.py -x -q -m gpu 2>&1 | tail -1 | sed "s/^/shape $k: /" >> $O/tests.txt
done
python -m pytest tests/test_segconv_gpu.py tests/test_adapnet_engine_gpu.py -x -q -m gpu 2>&1 | tail -1 | sed "s/^/model: /" >> $O/tests.txt
for shape in "256 256 3 60 80 1" "256 256 3 60 80 4" "128 128 3 60 80 1" "512 2048 1 15 20 4" "64 256 1 60 80 4" "256 64 1 60 80 4" "256 256 3 30 40 4"; do
  python tools/seg_layer_bench.py $shape 2>&1 | grep "per launch" | sed "s/$/ MODEL/" >> $O/layers.txt
  for k in 0 1 2 3 4 5; do
    OJF_SEG_GEMM_SHAPE=$k OJF_SEG_GEMM22_MIN=1 python tools/seg_layer_bench.py $shape 2>&1 | grep "per launch" >> $O/layers.txt
  done
done
for B in 1 4 8; do
python tools/seg_probe.py graph 40 240 320 $B 2>&1 | grep "seg engine" | sed "s/^/menu  /" >> $O/probe.txt
OJF_SEG_GEMM_MENU=3 python tools/seg_probe.py graph 40 240 320 $B 2>&1 | grep "seg engine" | sed "s/^/two   /" >> $O/probe.txt
done
OJF_SEG_TRACE=1 python tools/seg_probe.py eager 1 240 320 1 2>&1 | grep "^segconv" | tail -100 > $O/trace_b1.txt
OJF_SEG_TRACE=1 python tools/seg_probe.py eager 1 240 320 4 2>&1 | grep "^segconv" | tail -100 > $O/trace_b4.txt
