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
python bench.py --arith f32 --cpu-frames 0 > gpurun_out/final/bench_f32.json 2>/dev/null
python bench.py --height 120 --width 160 --grid 64 --cpu-frames 0 > gpurun_out/final/bench_A.json 2>/dev/null
python bench.py --mode parity --steps 100 --cpu-frames 0 > gpurun_out/final/bench_parity.json 2>/dev/null
python bench.py --height 480 --width 640 --grid 512 --semantics --steps 60 --warmup 5 --cpu-frames 0 > gpurun_out/final/bench_C.json 2>/dev/null
python bench.py --height 480 --width 640 --grid 512 --steps 60 --warmup 5 --cpu-frames 0 > gpurun_out/final/bench_Cgeo.json 2>/dev/null
python bench.py --semantics --semantic-strategy predict --steps 100 --cpu-frames 0 > gpurun_out/final/bench_predict.json 2>/dev/null
python bench.py --semantics --semantic-strategy predict --seg-engine torch --steps 100 --cpu-frames 0 > gpurun_out/final/bench_predict_torch.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/kp -o kp -- python bench.py --semantics --semantic-strategy predict --steps 50 --warmup 10 --cpu-frames 0 > /dev/null 2> gpurun_out/final/kp.err
python tools/adapnet_engine_probe.py 2>&1 | grep -v MIOpen > gpurun_out/final/adapnet_engine_probe.txt
python tools/mesh_timing.py 256 > gpurun_out/final/mesh_timing.txt 2>&1
python tools/mesh_timing.py 512 >> gpurun_out/final/mesh_timing.txt 2>&1
python tools/pmc_traffic.py $(find gpurun_out/final/pf -name '*counter_collection.csv' | head -1) $(find gpurun_out/final/pw -name '*counter_collection.csv' | head -1) 22 gpurun_out/final/traffic_pmc.json > gpurun_out/final/traffic_pmc.txt 2>&1
