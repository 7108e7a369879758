#!/bin/bash
# round 5, run 30: grid-barrier microbenchmark; K groups inside the 64x64 GEMM-shaped block (2 / 4 wave sets): bits, layer times, engine sweeps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_30; mkdir -p $O
timeout 120 tools/microbench/grid_barrier > $O/grid_barrier.txt 2>&1
OJF_SEG_GEMM_KG2_MAX=1000000 OJF_SEG_GEMM_KG_MIN_KB=2 python -m pytest tests/test_segconv_gpu.py -x -q -m gpu 2>&1 | tail -3 > $O/tests_kg2.txt
OJF_SEG_GEMM_KG4_MAX=1000000 OJF_SEG_GEMM_KG_MIN_KB=2 python -m pytest tests/test_segconv_gpu.py tests/test_adapnet_engine_gpu.py -x -q -m gpu 2>&1 | tail -3 > $O/tests_kg4.txt
for shape in "256 256 3 60 80 1" "128 128 3 60 80 1" "256 256 3 30 40 1" "512 2048 1 15 20 4" "2048 512 1 15 20 4" "64 64 3 60 80 1" "64 256 1 60 80 1"; do
  python tools/seg_layer_bench.py $shape 2>&1 | grep "per launch" >> $O/layers.txt
  OJF_SEG_GEMM22_MIN=1 OJF_SEG_GEMM_KG2_MAX=1000000 python tools/seg_layer_bench.py $shape 2>&1 | grep "per launch" >> $O/layers.txt
  OJF_SEG_GEMM22_MIN=1 OJF_SEG_GEMM_KG4_MAX=1000000 python tools/seg_layer_bench.py $shape 2>&1 | grep "per launch" >> $O/layers.txt
done
run() { env "$@" python tools/seg_probe.py graph 40 240 320 $B 2>&1 | grep "seg engine" | sed "s/^/$* /" >> $O/probe.txt; }
for B in 1 4; do
run OJF_SEG_GEMM_KG2_MAX=0
run OJF_SEG_GEMM_KG2_MAX=320
run OJF_SEG_GEMM_KG2_MAX=640
run OJF_SEG_GEMM_KG2_MAX=1300
run OJF_SEG_GEMM_KG4_MAX=320
run OJF_SEG_GEMM_KG4_MAX=320 OJF_SEG_GEMM_KG2_MAX=640
run OJF_SEG_GEMM_KG4_MAX=640
run OJF_SEG_GEMM_KG4_MAX=320 OJF_SEG_GEMM22_MIN=64
run OJF_SEG_GEMM_KG4_MAX=640 OJF_SEG_GEMM_KG_MIN_KB=8
done
