set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6k
rm -rf $O; mkdir -p $O
brief() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['value'],1), j['ms_per_step'], j.get('host_loop_ms_per_frame'))"; }
for K in 0 1 0 1; do
  HIP_FORCE_DEV_KERNARG=$K python bench.py --train --steps 64 --warmup 16 --repeats 3 2>/dev/null | brief "train KERNARG=$K" >> $O/kernarg.txt
  HIP_FORCE_DEV_KERNARG=$K python bench.py --steps 200 --warmup 20 --repeats 3 --cpu-frames 0 --secondary 0 2>/dev/null | brief "headline KERNARG=$K" >> $O/kernarg.txt
  HIP_FORCE_DEV_KERNARG=$K python bench.py --semantics --semantic-strategy predict --steps 100 --warmup 10 --repeats 3 --cpu-frames 0 --secondary 0 2>/dev/null | brief "predict KERNARG=$K" >> $O/kernarg.txt
done
cat $O/kernarg.txt
