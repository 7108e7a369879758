set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6f
rm -rf $O; mkdir -p $O
for P in 0 1; do OJF_CONV_ROW_PERM=$P python tools/net_sha.py 240 320 2>&1 | grep sha >> $O/row_perm_sha.txt; OJF_CONV_ROW_PERM=$P python tools/net_sha.py 240 320 sem 2>&1 | grep sha >> $O/row_perm_sha.txt;  OJF_CONV_ROW_PERM=$P python tools/net_sha.py 120 160 2>&1 | grep sha >> $O/row_perm_sha.txt; done
cat $O/row_perm_sha.txt
timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_train_gpu.py tests/test_drivers_gpu.py -q -x -k "announced or training or train or fuse_training" 2>&1 | tail -8 > $O/pytest_train.txt; cat $O/pytest_train.txt
python tools/train_host_split.py 2>&1 | grep -v amdgpu.ids > $O/train_host_split.txt; cat $O/train_host_split.txt
for A in 1 0 1 0; do OJF_BENCH_NO_ANNOUNCE=$A python bench.py --train --steps 64 --warmup 16 --repeats 5 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NO_ANNOUNCE=$A', round(j['value'],1), j['ms_per_step'], j['host_loop_ms_per_frame'], j['host_bound_ratio'])" >> $O/bench_train_ab.txt; done
cat $O/bench_train_ab.txt
