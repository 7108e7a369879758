#!/bin/bash
# round 5, run 19: fuse_sequence (look-ahead of the 2-D network): bits + frames/s at L = 1 (fuse), 2, 4, 8; drivers test
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_19; mkdir -p $O
python -m pytest tests/test_pipeline_gpu.py tests/test_drivers_gpu.py -x -q -m gpu -k "fuse_sequence or fuse_many or test_fusion or semantics" 2>&1 | tail -5 > $O/tests.txt
python bench.py --semantics --semantic-strategy predict --steps 96 --warmup 16 --repeats 3 --lean 2>/dev/null > $O/bench_L1.json
for L in 2 4 8; do python bench.py --semantics --semantic-strategy predict --steps 96 --warmup 16 --repeats 3 --lookahead $L 2>/dev/null >> $O/bench_L.json; done
