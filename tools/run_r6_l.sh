set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6l
rm -rf $O; mkdir -p $O
python bench.py --train --force-group --steps 64 --warmup 16 --repeats 3 > $O/bench_train_rccl_one_rank.json 2> $O/rccl.err
tail -5 $O/rccl.err; cat $O/bench_train_rccl_one_rank.json | cut -c1-600
