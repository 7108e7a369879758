#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT
for cfg in 0 5; do
echo "== cfg $cfg" >> gpurun_out/r4_4_pair.txt
OJF_PAIR_CFG=$cfg ./tools/microbench/pair_bench2.exe >> gpurun_out/r4_4_pair.txt 2>&1
done
python -m pytest tests/test_net_gpu.py tests/test_headline_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -6 >> gpurun_out/r4_4_tests.txt
for cfg in 0 5 0 5; do
OJF_PAIR_CFG=$cfg python bench.py --steps 200 --warmup 20 --repeats 3 --lean >> gpurun_out/r4_4_bench_lean.txt 2>/dev/null
done
