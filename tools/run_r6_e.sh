set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6e
rm -rf $O; mkdir -p $O
for rep in 1 2; do for P in 0 1; do OJF_CONV_ROW_PERM=$P python tools/net_sha.py 240 320 >> $O/row_perm.txt 2>&1; done; done
for P in 0 1; do OJF_CONV_ROW_PERM=$P python tools/net_sha.py 240 320 sem >> $O/row_perm.txt 2>&1; OJF_CONV_ROW_PERM=$P python tools/net_sha.py 480 640 >> $O/row_perm.txt 2>&1; OJF_CONV_ROW_PERM=$P python tools/net_sha.py 120 160 >> $O/row_perm.txt 2>&1; done
grep -v amdgpu.ids $O/row_perm.txt
for P in 0 1 0 1; do OJF_CONV_ROW_PERM=$P python bench.py --steps 200 --warmup 20 --repeats 5 --cpu-frames 0 --secondary 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PERM=$P', round(j['value'],1), j['stages_ms'])" >> $O/bench_perm.txt; done
cat $O/bench_perm.txt
python tools/train_host_split.py > $O/train_host_split.txt 2>&1; grep -v amdgpu.ids $O/train_host_split.txt
timeout 600 python -m pytest tests/test_net_gpu.py -q -x 2>&1 | tail -5 > $O/pytest_net.txt; cat $O/pytest_net.txt
