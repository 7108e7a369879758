set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6n
rm -rf $O; mkdir -p $O
for rep in 1 2; do for W in 1 0; do if [ $W = 1 ]; then export OJF_NO_HALF_ENTRY=1; else unset OJF_NO_HALF_ENTRY; fi; python tools/net_sha.py 240 320 sem 2>&1 | grep -v amdgpu | sed "s/^/nohalf=$W /" >> $O/half_entry.txt; python tools/net_sha.py 480 640 sem 2>&1 | grep -v amdgpu | sed "s/^/nohalf=$W /" >> $O/half_entry.txt; done; done
unset OJF_NO_HALF_ENTRY
cat $O/half_entry.txt
timeout 1200 python -m pytest tests/test_net_gpu.py tests/test_headline_gpu.py tests/test_pipeline_gpu.py -q -x 2>&1 | tail -6 > $O/pytest.txt; cat $O/pytest.txt
for W in 1 0 1 0; do if [ $W = 1 ]; then export OJF_NO_HALF_ENTRY=1; else unset OJF_NO_HALF_ENTRY; fi; python bench.py --semantics --steps 200 --warmup 20 --repeats 3 --cpu-frames 0 --secondary 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nohalf=$W', round(j['value'],1), j['stages_ms'])" >> $O/bench_sem.txt; done
unset OJF_NO_HALF_ENTRY
cat $O/bench_sem.txt
for L in 4 8; do python bench.py --semantics --semantic-strategy predict --lookahead $L --steps 96 --warmup 16 --repeats 3 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lookahead $L', round(j['value'],1))" >> $O/bench_sem.txt; done
python bench.py --height 480 --width 640 --grid 512 --semantics --n-classes 40 --steps 100 --warmup 10 --repeats 3 --cpu-frames 0 --secondary 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C sem', round(j['value'],1), j['stages_ms'])" >> $O/bench_sem.txt
cat $O/bench_sem.txt
