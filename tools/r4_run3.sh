#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT
for cfg in 0 2 3 4; do
echo "== cfg $cfg" >> gpurun_out/r4_3_pair.txt
OJF_PAIR_CFG=$cfg ./tools/microbench/pair_bench2.exe >> gpurun_out/r4_3_pair.txt 2>&1
done
for cfg in 2 3 4; do
OJF_PAIR_CFG=$cfg python -m pytest tests/test_net_gpu.py -x -q -m gpu 2>&1 | tail -4 >> gpurun_out/r4_3_net_tests.txt
done
for cfg in 0 2 4 0 2 4; do
OJF_PAIR_CFG=$cfg python bench.py --steps 200 --warmup 20 --repeats 3 --lean >> gpurun_out/r4_3_bench_lean.txt 2>/dev/null
done
