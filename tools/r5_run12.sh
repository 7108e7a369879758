#!/bin/bash
# round 5, run 12: fuse_many (S scenes side by side on one GPU): bits + aggregate frames/s at S = 1, 2, 3, 4
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_12; mkdir -p $O
python bench.py --steps 200 --warmup 20 --repeats 3 --lean > $O/bench_S1.json 2>/dev/null
for S in 2 3 4; do python bench.py --steps 100 --warmup 10 --repeats 3 --scenes $S > $O/bench_S$S.json 2>/dev/null; done
python bench.py --steps 60 --warmup 10 --repeats 3 --scenes 4 --semantics --semantic-strategy predict > $O/bench_predict_S4.json 2> $O/err_predict.txt
python bench.py --steps 60 --warmup 10 --repeats 3 --scenes 2 --semantics > $O/bench_sem_S2.json 2>/dev/null
