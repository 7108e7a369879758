#!/bin/bash
# round 5, run 18: NW = 2 split-K rule (150 .. 400 blocks), NW = 1 plain form for big maps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_18; mkdir -p $O
OJF_SEG_PLAIN_NW1_MIN=1 python -m pytest tests/test_segconv_gpu.py -x -q -m gpu 2>&1 | tail -3 > $O/tests.txt
python -m pytest tests/test_segconv_gpu.py tests/test_adapnet_engine_gpu.py -x -q -m gpu 2>&1 | tail -3 >> $O/tests.txt
run() { env "$@" python tools/seg_probe.py graph 50 2>&1 | grep "seg engine" | sed "s/^/$* /" >> $O/probe.txt; }
for i in 1 2; do
run OJF_SEG_NW2_MAX_BLOCKS=0
run OJF_SEG_NW2_MAX_BLOCKS=400
run OJF_SEG_NW2_MAX_BLOCKS=400 OJF_SEG_NW2_MIN_KB=32
run OJF_SEG_NW2_MAX_BLOCKS=320
run OJF_SEG_NW2_MAX_BLOCKS=400 OJF_SEG_PLAIN_NW1_MIN=150
run OJF_SEG_NW2_MAX_BLOCKS=400 OJF_SEG_PLAIN_NW1_MIN=50
done
