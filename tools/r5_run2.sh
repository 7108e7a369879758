#!/bin/bash
# round 5, run 2: XCD-aware block numbering of the SEGCONV kernels - A/B, bits, per-launch timeline with layer shapes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_2; mkdir -p $O
python -m pytest tests/test_segconv_gpu.py tests/test_adapnet_engine_gpu.py -x -q -m gpu 2>&1 | tail -4 > $O/tests.txt
for i in 1 2; do
OJF_SEG_XCD=0 python tools/seg_probe.py graph 50 2>&1 | grep -v amdgpu.ids >> $O/probe.txt
OJF_SEG_XCD=1 python tools/seg_probe.py graph 50 2>&1 | grep -v amdgpu.ids >> $O/probe.txt
done
OJF_SEG_TRACE=1 python tools/seg_probe.py eager 1 2> $O/trace_all.txt > /dev/null
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python tools/seg_probe.py graph 10 > /dev/null 2> $O/kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -o pf -- python tools/seg_probe.py eager 4 > /dev/null 2> $O/pf.err
python tools/seg_seq.py $(find $O/kt -name '*kernel_trace.csv' | head -1) $(find $O/pf -name '*counter_collection.csv' | head -1) > $O/seq.txt 2>&1
rm -rf $O/kt $O/pf
