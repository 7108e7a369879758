#!/bin/bash
# round 5, run 13: batched SEGCONV (B images per pass): bits, engine timing at B = 1, 2, 4, fuse_many with predicted labels
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_13; mkdir -p $O
python -m pytest tests/test_segconv_gpu.py tests/test_adapnet_engine_gpu.py -x -q -m gpu 2>&1 | tail -12 > $O/tests.txt
python tools/seg_probe.py graph 40 > $O/probe.txt 2>&1
for B in 2 4 8; do python tools/seg_probe.py graph 30 240 320 $B >> $O/probe.txt 2>&1; done
python bench.py --steps 60 --warmup 10 --repeats 3 --scenes 4 --semantics --semantic-strategy predict > $O/bench_predict_S4.json 2> $O/err_predict.txt
python bench.py --steps 60 --warmup 10 --repeats 3 --scenes 2 --semantics --semantic-strategy predict > $O/bench_predict_S2.json 2>> $O/err_predict.txt
