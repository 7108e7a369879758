#!/bin/bash
# round 5, run 15: finalize with two voxels per lane - A/B against the committed build + extract / integrate bits; wide-net test
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_15; mkdir -p $O
python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "wide_nets" 2>&1 | tail -5 > $O/tests.txt
python -m pytest tests/test_extract_integrate_gpu.py tests/test_headline_gpu.py -x -q -m gpu 2>&1 | tail -5 >> $O/tests.txt
for i in 1 2 3; do
OJF_LIB_PATH=$PWD/ab/libojf_prev.so python bench.py --steps 200 --warmup 20 --repeats 3 --lean 2>/dev/null | sed 's/^/prev /' >> $O/bench.txt
python bench.py --steps 200 --warmup 20 --repeats 3 --lean 2>/dev/null | sed 's/^/new  /' >> $O/bench.txt
done
