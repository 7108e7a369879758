# Counter passes of the round profile (called by tools/final_profile.sh; also usable alone through gpurun):
# HBM traffic (FETCH_SIZE, WRITE_SIZE in separate passes) and SQ counters per kernel, on `bench.py --lean` (every kernel
# runs exactly once per launch site and frame).  usage: bash tools/pmc_profile.sh <output dir under gpurun_out/>
O=${1:?output directory}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p $O
S="python bench.py --steps 20 --warmup 2 --lean"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -o pf -- $S > /dev/null 2> $O/pf.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -o pw -- $S > /dev/null 2> $O/pw.err
python tools/pmc_traffic.py $(find $O/pf -name '*counter_collection.csv' | head -1) $(find $O/pw -name '*counter_collection.csv' | head -1) 0 $O/traffic_pmc.json > $O/traffic_pmc.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/sq -o sq -- $S > /dev/null 2> $O/sq.err
python tools/pmc_sq_summary.py $(find $O/sq -name '*counter_collection.csv' | head -1) $O/sq_counters.json > $O/sq_counters.txt 2>&1
rm -rf $O/pf $O/pw $O/sq
