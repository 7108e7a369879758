#!/bin/bash
# round 5, run 7: register-staged tile kernel (OJF_SEG_TILE=1): bits, thresholds, per-layer times
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_7; mkdir -p $O
OJF_SEG_TILE=1 OJF_SEG_WIDE_MIN=1 python -m pytest tests/test_segconv_gpu.py -x -q -m gpu 2>&1 | tail -5 > $O/tests.txt
OJF_SEG_TILE=1 OJF_SEG_WIDE_MIN=100000 OJF_SEG_WIDE1_MIN=1 python -m pytest tests/test_segconv_gpu.py -x -q -m gpu 2>&1 | tail -5 >> $O/tests.txt
run() { echo "== $*" >> $O/probe.txt; env "$@" python tools/seg_probe.py graph 50 2>&1 | grep -v amdgpu.ids >> $O/probe.txt; }
run OJF_SEG_TILE=0
run OJF_SEG_TILE=1
run OJF_SEG_TILE=1 OJF_SEG_WIDE_MIN=100
run OJF_SEG_TILE=1 OJF_SEG_WIDE_MIN=50
run OJF_SEG_TILE=1 OJF_SEG_WIDE_MIN=100 OJF_SEG_WIDE1_MIN=256
run OJF_SEG_TILE=1 OJF_SEG_WIDE_MIN=100 OJF_SEG_WIDE1_MIN=128
run OJF_SEG_TILE=1 OJF_SEG_WIDE_MIN=100 OJF_SEG_WIDE1_MIN=64
run OJF_SEG_TILE=1 OJF_SEG_WIDE_MIN=50 OJF_SEG_WIDE1_MIN=32
run OJF_SEG_TILE=1 OJF_SEG_WIDE_MIN=20 OJF_SEG_WIDE1_MIN=32
tr() { tag=$1; shift
env "$@" OJF_SEG_TRACE=1 python tools/seg_probe.py eager 1 2> $O/trace_$tag.txt > /dev/null
env "$@" rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python tools/seg_probe.py graph 10 > /dev/null 2> $O/kt.err
python tools/seg_seq.py $(find $O/kt -name '*kernel_trace.csv' | head -1) > $O/seq_$tag.txt 2>&1
rm -rf $O/kt; }
tr A OJF_SEG_TILE=1 OJF_SEG_WIDE_MIN=20 OJF_SEG_WIDE1_MIN=32
tr B OJF_SEG_TILE=1 OJF_SEG_WIDE_MIN=100000 OJF_SEG_WIDE1_MIN=32
