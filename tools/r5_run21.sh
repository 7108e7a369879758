#!/bin/bash
# round 5, run 21: GEMM-shaped SEGCONV kernel: bits, thresholds at 1 and 4 frames per pass
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_21; mkdir -p $O
OJF_SEG_GEMM_MIN=1 OJF_SEG_GEMM_MIN_KB=1 python -m pytest tests/test_segconv_gpu.py -x -q -m gpu 2>&1 | tail -6 > $O/tests.txt
OJF_SEG_GEMM_MIN=1 python -m pytest tests/test_adapnet_engine_gpu.py -x -q -m gpu 2>&1 | tail -4 >> $O/tests.txt
run() { env "$@" python tools/seg_probe.py graph 40 240 320 $B 2>&1 | grep "seg engine" | sed "s/^/$* /" >> $O/probe.txt; }
for B in 1 4; do
run OJF_SEG_GEMM_MIN=1000000
run OJF_SEG_GEMM_MIN=512
run OJF_SEG_GEMM_MIN=256
run OJF_SEG_GEMM_MIN=128
run OJF_SEG_GEMM_MIN=128
run OJF_SEG_GEMM_MIN=128 OJF_SEG_GEMM_MIN_KB=16
done
tr() { tag=$1; B=$2; shift; shift
env "$@" OJF_SEG_TRACE=1 python tools/seg_probe.py eager 1 240 320 $B 2> $O/trace_$tag.txt > /dev/null
env "$@" rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python tools/seg_probe.py graph 10 240 320 $B > /dev/null 2> $O/kt.err
SEG_PACKS=$((2*B)) python tools/seg_seq.py $(find $O/kt -name '*kernel_trace.csv' | head -1) > $O/seq_$tag.txt 2>&1
rm -rf $O/kt; }
tr g4 4 OJF_SEG_GEMM_MIN=128
tr g1 1 OJF_SEG_GEMM_MIN=128
