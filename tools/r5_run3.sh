#!/bin/bash
# round 5, run 3: guard tests + wide-net test; SEGCONV: wide<1> thresholds
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_3; mkdir -p $O
python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "range_guard" 2>&1 | tail -15 > $O/tests_guard.txt
python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "wide_net or jumps" 2>&1 | tail -15 > $O/tests_wide.txt
for m in 1000000000 256 128 64 32; do
echo "== WIDE1_MIN $m" >> $O/probe.txt
OJF_SEG_WIDE1_MIN=$m python tools/seg_probe.py graph 50 2>&1 | grep -v amdgpu.ids >> $O/probe.txt
done
for m in 64 32; do
OJF_SEG_WIDE1_MIN=$m OJF_SEG_TRACE=1 python tools/seg_probe.py eager 1 2> $O/trace_$m.txt > /dev/null
OJF_SEG_WIDE1_MIN=$m rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python tools/seg_probe.py graph 10 > /dev/null 2> $O/kt.err
python tools/seg_seq.py $(find $O/kt -name '*kernel_trace.csv' | head -1) > $O/seq_$m.txt 2>&1
rm -rf $O/kt
done
