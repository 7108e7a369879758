set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6m
rm -rf $O; mkdir -p $O
for rep in 1 2; do for W in 1 0; do if [ $W = 1 ]; then export OJF_NO_WIDE_ENTRY=1; else unset OJF_NO_WIDE_ENTRY; fi; python tools/net_sha.py 240 320 sem 2>&1 | grep -v amdgpu | sed "s/^/legacy=$W /" >> $O/wide_entry.txt; python tools/net_sha.py 480 640 sem 2>&1 | grep -v amdgpu | sed "s/^/legacy=$W /" >> $O/wide_entry.txt; done; done
unset OJF_NO_WIDE_ENTRY
cat $O/wide_entry.txt
timeout 1200 python -m pytest tests/test_net_gpu.py tests/test_headline_gpu.py tests/test_pipeline_gpu.py -q -x 2>&1 | tail -6 > $O/pytest.txt; cat $O/pytest.txt
for W in 1 0 1 0; do if [ $W = 1 ]; then export OJF_NO_WIDE_ENTRY=1; else unset OJF_NO_WIDE_ENTRY; fi; python bench.py --semantics --steps 200 --warmup 20 --repeats 3 --cpu-frames 0 --secondary 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('legacy=$W', round(j['value'],1), j['stages_ms'])" >> $O/bench_sem.txt; done
unset OJF_NO_WIDE_ENTRY
cat $O/bench_sem.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
