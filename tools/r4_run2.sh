#!/bin/bash
# round 4, run 2: same-pass dy factors (training tests, bench --train), dense-pair phase stamps
set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r4_2_train_tests.txt
python -m pytest tests/test_headline_gpu.py -x -q -m gpu -s -k "training" 2>&1 | grep -v MIOpen | tail -30 > gpurun_out/r4_2_headline_train.txt
python -m pytest tests/test_bench_gpu.py tests/test_drivers_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r4_2_bench_tests.txt
for i in 1 2; do
python bench.py --train --steps 24 --warmup 16 --repeats 5 > gpurun_out/r4_2_train_$i.json 2> gpurun_out/r4_2_train_$i.err
done
OJF_TRAIN_WGRAD16=0 python bench.py --train --steps 24 --warmup 16 --repeats 5 > gpurun_out/r4_2_train_wgrad32.json 2>/dev/null
./tools/microbench/pair_bench.exe > gpurun_out/r4_2_pair_stamps.txt 2>&1
