# usage: tools/collect_profiles.sh r02   (after tools/final_profile.sh ran through gpurun): copies the judged summaries
# from the scratch directory gpurun_out/final/ into the tracked profiles/ directory, named per round
R=${1:?round tag, e.g. r02}
S=gpurun_out/final
for f in bench.json bench_under_rocprof.json bench_kernel_stats.csv kernel_trace_summary.txt traffic_pmc.json sq_counters.json \
         bench_predict.json predict_kernel_stats.csv train_throughput.txt train_kernel_stats.csv adapnet_engine_probe.txt \
         bench_parity.json bench_A.json pytest_gpu.txt traffic_pmc.txt sq_counters.txt sq_counters_predict.json sq_counters_predict.txt \
         pmc_calibration.txt bench_train.json train_timeline.txt train_sq_counters.txt train_sq_counters.json bench_fuse_many.json seg_engine_batches.txt seg_forms_b1.txt seg_launch_timeline.txt seg_launch_timeline_b8.txt grid_barrier.txt bench_train_rccl_one_rank.json train_host_split.txt fabric_rate.txt; do
  [ -f $S/$f ] && cp $S/$f profiles/${R}_final_$f
done
ls profiles | grep ${R}_
