#!/bin/bash
# round 5, run 4: SEGCONV wide kernel: pipeline depth x thresholds
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_4; mkdir -p $O
run() { echo "== $*" >> $O/probe.txt; env "$@" python tools/seg_probe.py graph 50 2>&1 | grep -v amdgpu.ids >> $O/probe.txt; }
run OJF_SEG_WIDE_DEPTH=3
run OJF_SEG_WIDE_DEPTH=6
run OJF_SEG_WIDE_DEPTH=8
run OJF_SEG_WIDE_DEPTH=3 OJF_SEG_WIDE_MIN=100
run OJF_SEG_WIDE_DEPTH=6 OJF_SEG_WIDE_MIN=100
run OJF_SEG_WIDE_DEPTH=8 OJF_SEG_WIDE_MIN=100
run OJF_SEG_WIDE_DEPTH=6 OJF_SEG_WIDE_MIN=100 OJF_SEG_WIDE1_MIN=256
run OJF_SEG_WIDE_DEPTH=8 OJF_SEG_WIDE_MIN=100 OJF_SEG_WIDE1_MIN=64
run OJF_SEG_WIDE_DEPTH=8 OJF_SEG_WIDE_MIN=50 OJF_SEG_WIDE1_MIN=64
tr() { tag=$1; shift
env "$@" OJF_SEG_TRACE=1 python tools/seg_probe.py eager 1 2> $O/trace_$tag.txt > /dev/null
env "$@" rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python tools/seg_probe.py graph 10 > /dev/null 2> $O/kt.err
python tools/seg_seq.py $(find $O/kt -name '*kernel_trace.csv' | head -1) > $O/seq_$tag.txt 2>&1
rm -rf $O/kt; }
tr E OJF_SEG_WIDE_DEPTH=6 OJF_SEG_WIDE_MIN=100
tr H OJF_SEG_WIDE_DEPTH=8 OJF_SEG_WIDE_MIN=50 OJF_SEG_WIDE1_MIN=64
OJF_SEG_WIDE_DEPTH=8 OJF_SEG_WIDE_MIN=1 python -m pytest tests/test_segconv_gpu.py -x -q -m gpu 2>&1 | tail -3 > $O/tests_d8.txt
