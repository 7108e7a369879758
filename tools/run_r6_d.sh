set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6d
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_extract_integrate_gpu.py tests/test_abi.py -q -k "many or abi or guard or integrate" 2>&1 | tail -15 > $O/pytest_many.txt
timeout 900 python -m pytest tests/test_pipeline_gpu.py -q -k "fuse_many or guard or fuse_sequence" 2>&1 | tail -15 >> $O/pytest_many.txt
timeout 900 python -m pytest tests/test_headline_gpu.py -q -k "batched_2d" 2>&1 | tail -15 >> $O/pytest_many.txt
cat $O/pytest_many.txt
for S in 2 4 8; do python bench.py --steps 100 --warmup 10 --repeats 3 --scenes $S >> $O/bench_many.json 2>>$O/bench_many.err; done
python bench.py --steps 100 --warmup 10 --repeats 3 --scenes 4 --semantics >> $O/bench_many.json 2>>$O/bench_many.err
python - <<'PY'
import json
for ln in open('gpurun_out/r6d/bench_many.json'):
    if ln.startswith('{'):
        j = json.loads(ln)
        print(j['scenes_per_gpu'], round(j['value'], 1), j.get('stages_ms_per_call'), j.get('extract_integrate_us_per_frame'), (j.get('roofline_hbm') or {}).get('frac'))
PY
python tools/train_host_profile.py > $O/train_host_profile.txt 2>&1
head -60 $O/train_host_profile.txt
