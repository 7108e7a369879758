set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6a
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > $O/pytest_gpu_tail.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -5 $O/pytest_gpu_tail.txt
python tools/bench_brief.py $O/bench.json 2>&1 | tail -30
