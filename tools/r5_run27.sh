#!/bin/bash
# round 5, run 27: GEMM-shaped kernel on by default (128x128 tile from 256 blocks, 64x64 from OJF_SEG_GEMM22_MIN): tests + sweeps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_27; mkdir -p $O
python -m pytest tests/test_segconv_gpu.py tests/test_adapnet_engine_gpu.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt
run() { env "$@" python tools/seg_probe.py graph 40 240 320 $B 2>&1 | grep "seg engine" | sed "s/^/$* /" >> $O/probe.txt; }
for B in 1 4; do
run OJF_SEG_GEMM_MIN=1000000 OJF_SEG_GEMM22_MIN=1000000
run OJF_SEG_GEMM22_MIN=1000000
run OJF_SEG_GEMM22_MIN=512
run OJF_SEG_GEMM22_MIN=256
run OJF_SEG_GEMM22_MIN=192
run OJF_SEG_GEMM22_MIN=128
run OJF_SEG_GEMM22_MIN=96
run OJF_SEG_GEMM22_MIN=128 OJF_SEG_GEMM_MIN=200
done
OJF_SEG_TRACE=1 python tools/seg_probe.py eager 1 240 320 1 2>&1 | grep "^segconv" > $O/trace_b1.txt
OJF_SEG_TRACE=1 python tools/seg_probe.py eager 1 240 320 4 2>&1 | grep "^segconv" > $O/trace_b4.txt
