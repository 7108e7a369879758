set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6h
rm -rf $O; mkdir -p $O
for H in measured priority probe; do OJF_LOOKAHEAD_STREAM=$H python tools/lookahead_order_probe.py 2>&1 | grep -v amdgpu | sed "s/^/$H /" >> $O/lookahead.txt; done
cat $O/lookahead.txt
