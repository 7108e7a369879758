set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6g
rm -rf $O; mkdir -p $O
LB0=$GRAFT_REPO_ROOT/online_joint_depthfusion_and_semantic_amd/libojf_lb0.so
for rep in 1 2; do
  OJF_CONV_PERSIST=0 python tools/net_sha.py 240 320 2>&1 | grep -v amdgpu >> $O/persist.txt
  for B in 2 3 4 6; do OJF_CONV_PERSIST=$B python tools/net_sha.py 240 320 2>&1 | grep -v amdgpu >> $O/persist.txt; done
  OJF_LIB_PATH=$LB0 OJF_CONV_PERSIST=0 python tools/net_sha.py 240 320 2>&1 | grep -v amdgpu | sed 's/^/LB0 /' >> $O/persist.txt
  for B in 2 3 4; do OJF_LIB_PATH=$LB0 OJF_CONV_PERSIST=$B python tools/net_sha.py 240 320 2>&1 | grep -v amdgpu | sed 's/^/LB0 /' >> $O/persist.txt; done
done
for B in 0 4; do OJF_CONV_PERSIST=$B python tools/net_sha.py 240 320 sem 2>&1 | grep sha >> $O/persist.txt; OJF_CONV_PERSIST=$B python tools/net_sha.py 480 640 2>&1 | grep sha >> $O/persist.txt; OJF_CONV_PERSIST=$B python tools/net_sha.py 120 160 2>&1 | grep sha >> $O/persist.txt; OJF_CONV_PERSIST=$B python tools/net_sha.py 48 64 2>&1 | grep sha >> $O/persist.txt; done
cat $O/persist.txt
timeout 900 python -m pytest tests/test_net_gpu.py tests/test_pipeline_gpu.py -q -x -k "not guard_policy" 2>&1 | tail -6 > $O/pytest.txt; cat $O/pytest.txt
python tools/fabric_rate.py 2>&1 | grep -v amdgpu > $O/fabric_rate.txt; cat $O/fabric_rate.txt
