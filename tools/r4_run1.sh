#!/bin/bash
# round 4, run 1: the training guard / trajectory tests, the bench --train A/B of the guard's wait
set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "guard or jump or uncollected or backward_arithmetic or interleaved or invalidate or split" 2>&1 | tail -25 > gpurun_out/r4_1_train_tests.txt
python -m pytest tests/test_headline_gpu.py -x -q -m gpu -s -k "training" 2>&1 | tail -40 > gpurun_out/r4_1_headline_train.txt
python -m pytest tests/test_bench_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r4_1_bench_tests.txt
for i in 1 2; do
python bench.py --train --steps 24 --warmup 16 --repeats 5 > gpurun_out/r4_1_train_guard_$i.json 2> gpurun_out/r4_1_train_guard_$i.err
OJF_TRAIN_NO_STATUS=1 python bench.py --train --steps 24 --warmup 16 --repeats 5 > gpurun_out/r4_1_train_noguard_$i.json 2> gpurun_out/r4_1_train_noguard_$i.err
done
python bench.py > gpurun_out/r4_1_bench.json 2> gpurun_out/r4_1_bench.err
