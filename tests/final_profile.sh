set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/final/pytest_gpu.txt
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
python bench.py --semantics --steps 100 > gpurun_out/final/bench_sem.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/kt -o kt -- python bench.py --steps 100 --warmup 10 > gpurun_out/final/bench_prof.json 2> gpurun_out/final/kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/final/pf -o pf -- python bench.py --steps 20 --warmup 2 > /dev/null 2> gpurun_out/final/pf.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/final/pw -o pw -- python bench.py --steps 20 --warmup 2 > /dev/null 2> gpurun_out/final/pw.err
find gpurun_out/final -name '*.csv' | head -20
