# Round profile run (on the GPU box, through gpurun): everything the numbers in DESIGN.md / profiles/ come from.
# Results land in gpurun_out/final/; tools/collect_profiles.sh copies the summaries into profiles/rNN_*.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/final
rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -60 > $O/pytest_gpu_tail.txt; tail -3 $O/pytest_gpu_tail.txt > $O/pytest_gpu.txt
python bench.py > $O/bench.json 2> $O/bench.err
B="python bench.py --steps 100 --warmup 10 --lean"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $B > $O/bench_under_rocprof.json 2> $O/kt.err
python tools/kernel_trace_summary.py $(find $O/kt -name '*kernel_trace.csv' | head -1) 0 60 > $O/kernel_trace_summary.txt 2>&1
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/bench_kernel_stats.csv
bash tools/pmc_profile.sh $O
# counter calibration on known byte counts (FETCH_SIZE / WRITE_SIZE units and the scattered-record pattern)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cf -o cf -- python tools/pmc_calibrate.py > /dev/null 2> $O/cf.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/cw -o cw -- python tools/pmc_calibrate.py > /dev/null 2> $O/cw.err
python tools/pmc_calibrate_summary.py $(find $O/cf -name '*counter_collection.csv' | head -1) $(find $O/cw -name '*counter_collection.csv' | head -1) > $O/pmc_calibration.txt 2>&1
# SQ counters of the predict path (AdapNet++ engine + two-head net): MFMA-pipe utilisation per kernel
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/sqp -o sqp -- python bench.py --semantics --semantic-strategy predict --steps 20 --warmup 2 --lean > /dev/null 2> $O/sqp.err
python tools/pmc_sq_summary.py $(find $O/sqp -name '*counter_collection.csv' | head -1) $O/sq_counters_predict.json > $O/sq_counters_predict.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kp -o kp -- python bench.py --semantics --semantic-strategy predict --steps 50 --warmup 10 --cpu-frames 0 --secondary 0 > $O/bench_predict.json 2> $O/kp.err
cp $(find $O/kp -name '*kernel_stats.csv' | head -1) $O/predict_kernel_stats.csv
python tools/train_throughput.py > $O/train_throughput.txt 2>&1
python tools/train_throughput.py >> $O/train_throughput.txt 2>&1
python bench.py --train --steps 64 --warmup 16 --repeats 5 > $O/bench_train.json 2>/dev/null
python tools/train_host_split.py 2>&1 | grep -v amdgpu.ids > $O/train_host_split.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktr -o ktr -- python tools/train_throughput.py > /dev/null 2> $O/ktr.err
cp $(find $O/ktr -name '*kernel_stats.csv' | head -1) $O/train_kernel_stats.csv
python tools/train_timeline.py $(find $O/ktr -name '*kernel_trace.csv' | head -1) > $O/train_timeline.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/sqt -o sqt -- python tools/train_throughput.py > /dev/null 2> $O/sqt.err
python tools/pmc_sq_summary.py $(find $O/sqt -name '*counter_collection.csv' | head -1) $O/train_sq_counters.json > $O/train_sq_counters.txt 2>&1
python tools/adapnet_engine_probe.py 2>&1 | grep -v MIOpen > $O/adapnet_engine_probe.txt
python bench.py --mode parity --steps 100 --cpu-frames 0 --secondary 0 > $O/bench_parity.json 2>/dev/null
python bench.py --height 120 --width 160 --grid 64 --cpu-frames 0 --secondary 0 > $O/bench_A.json 2>/dev/null
# several scenes per GPU (Pipeline.fuse_many) and the 2-D engine on batches
for S in 2 4 8; do python bench.py --steps 100 --warmup 10 --repeats 3 --scenes $S >> $O/bench_fuse_many.json 2>/dev/null; done
python bench.py --steps 60 --warmup 10 --repeats 3 --scenes 4 --semantics --semantic-strategy predict >> $O/bench_fuse_many.json 2>/dev/null
for B in 1 2 4 8; do python tools/seg_probe.py graph 30 240 320 $B 2>&1 | grep "seg engine" >> $O/seg_engine_batches.txt; done
# per-launch timeline of the 2-D engine (one frame per pass) and the kernel form every layer runs in
OJF_SEG_TRACE=1 python tools/seg_probe.py eager 1 240 320 1 2>&1 | grep "^segconv" | tail -100 > $O/seg_forms_b1.txt
rocprofv3 --kernel-trace --output-format csv -d $O/kts -o kts -- python tools/seg_probe.py graph 10 240 320 1 > /dev/null 2> $O/kts.err
SEG_PACKS=2 python tools/seg_seq.py $(find $O/kts -name '*kernel_trace.csv' | head -1) > $O/seg_launch_timeline.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/kts8 -o kts8 -- python tools/seg_probe.py graph 10 240 320 8 > /dev/null 2> $O/kts8.err
SEG_PACKS=16 python tools/seg_seq.py $(find $O/kts8 -name '*kernel_trace.csv' | head -1) > $O/seg_launch_timeline_b8.txt 2>&1
rm -rf $O/kts8
python tools/fabric_rate.py 2>&1 | grep -v amdgpu.ids > $O/fabric_rate.txt
timeout 120 tools/microbench/grid_barrier > $O/grid_barrier.txt 2>&1
rm -rf $O/kts
rm -rf $O/kt $O/pf $O/pw $O/sq $O/kp $O/ktr $O/cf $O/cw $O/sqp $O/sqt
ls -la $O
