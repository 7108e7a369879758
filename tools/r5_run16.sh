#!/bin/bash
# round 5, run 16: float4 residual / gate loads in the SEGCONV epilogue, split-K threshold - A/B on one box
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_16; mkdir -p $O
python -m pytest tests/test_segconv_gpu.py tests/test_adapnet_engine_gpu.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt
for i in 1 2; do
OJF_LIB_PATH=$PWD/ab/libojf_prev.so python tools/seg_probe.py graph 50 2>&1 | grep "seg engine" | sed 's/^/prev /' >> $O/probe.txt
python tools/seg_probe.py graph 50 2>&1 | grep "seg engine" | sed 's/^/new  /' >> $O/probe.txt
OJF_SEG_SPLITK_MIN_KB=12 python tools/seg_probe.py graph 50 2>&1 | grep "seg engine" | sed 's/^/kb12 /' >> $O/probe.txt
OJF_SEG_SPLITK_MIN_KB=20 python tools/seg_probe.py graph 50 2>&1 | grep "seg engine" | sed 's/^/kb20 /' >> $O/probe.txt
OJF_SEG_SPLITK_MIN_KB=40 python tools/seg_probe.py graph 50 2>&1 | grep "seg engine" | sed 's/^/kb40 /' >> $O/probe.txt
done
for B in 4; do
python tools/seg_probe.py graph 30 240 320 $B 2>&1 | grep "seg engine" | sed 's/^/B4 default /' >> $O/probe.txt
OJF_SEG_WIDE_MIN=128 python tools/seg_probe.py graph 30 240 320 $B 2>&1 | grep "seg engine" | sed 's/^/B4 wide128 /' >> $O/probe.txt
OJF_SEG_WIDE_MIN=512 python tools/seg_probe.py graph 30 240 320 $B 2>&1 | grep "seg engine" | sed 's/^/B4 wide512 /' >> $O/probe.txt
OJF_SEG_WIDE_MIN=100000 python tools/seg_probe.py graph 30 240 320 $B 2>&1 | grep "seg engine" | sed 's/^/B4 nowide /' >> $O/probe.txt
OJF_SEG_SPLITK_NW2_MIN=100 python tools/seg_probe.py graph 30 240 320 $B 2>&1 | grep "seg engine" | sed 's/^/B4 nw2_100 /' >> $O/probe.txt
OJF_SEG_SPLITK_MIN_KB=20 python tools/seg_probe.py graph 30 240 320 $B 2>&1 | grep "seg engine" | sed 's/^/B4 kb20 /' >> $O/probe.txt
done
