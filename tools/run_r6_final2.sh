set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/final2
rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -60 > $O/pytest_gpu_tail.txt; tail -3 $O/pytest_gpu_tail.txt > $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $O/smoke.txt
B="python bench.py --steps 100 --warmup 10 --lean"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $B > $O/bench_under_rocprof.json 2> $O/kt.err
python tools/kernel_trace_summary.py $(find $O/kt -name '*kernel_trace.csv' | head -1) 0 60 > $O/kernel_trace_summary.txt 2>&1
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/bench_kernel_stats.csv
rm -rf $O/kt
cat $O/pytest_gpu.txt $O/smoke.txt
