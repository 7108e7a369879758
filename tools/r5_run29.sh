#!/bin/bash
# round 5, run 29: per-launch timeline of the 2-D engine at one frame per pass with the GEMM-shaped form on; warm against cold weights per layer shape
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_29; mkdir -p $O
OJF_SEG_TRACE=1 python tools/seg_probe.py eager 1 240 320 1 2> $O/trace.txt > /dev/null
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python tools/seg_probe.py graph 10 240 320 1 > /dev/null 2> $O/kt.err
SEG_PACKS=2 python tools/seg_seq.py $(find $O/kt -name '*kernel_trace.csv' | head -1) > $O/seq.txt 2>&1
rm -rf $O/kt
for shape in "512 512 3 15 20 1" "2048 512 1 15 20 1" "512 2048 1 15 20 1" "1024 2048 1 15 20 1" "256 256 3 30 40 1" "256 1024 1 30 40 1" "128 128 3 60 80 1" "256 256 3 60 80 1"; do
  python tools/seg_layer_bench.py $shape 64 1 2>&1 | grep "per launch" >> $O/warm_cold.txt
  python tools/seg_layer_bench.py $shape 64 32 2>&1 | grep "per launch" >> $O/warm_cold.txt
done
