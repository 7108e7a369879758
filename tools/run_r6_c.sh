set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6c
rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_pipeline_gpu.py tests/test_segconv_gpu.py tests/test_train_gpu.py tests/test_volume_gpu.py tests/test_extract_integrate_gpu.py -m gpu -q 2>&1 | tail -40 > $O/pytest_gpu_tail.txt
tail -8 $O/pytest_gpu_tail.txt
bash tools/run_r6_b.sh
