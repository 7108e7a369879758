set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6i
rm -rf $O; mkdir -p $O
L=$O/win_layers.txt
for shape in "256 256 3 60 80" "288 256 3 60 80" "64 64 3 60 80" "256 256 3 30 40" "128 128 3 30 40" "256 1024 3 30 40"; do
  for B in 1 4 8; do
    for WIN in 0 2 3; do
      OJF_SEG_WIN=$WIN OJF_SEG_TRACE=1 timeout 120 python tools/seg_layer_bench.py $shape $B 64 4 2>&1 | grep -E "us per launch|^segconv" | sort | uniq -c | sort -rn | head -2 >> $L
    done
  done
done
cat $L | cut -c1-220
OJF_SEG_WIN=1 timeout 600 python -m pytest tests/test_segconv_gpu.py tests/test_adapnet_engine_gpu.py -q -x 2>&1 | tail -5 > $O/pytest_win.txt; cat $O/pytest_win.txt
for B in 1 8; do for WIN in 0 1; do OJF_SEG_WIN=$WIN python tools/seg_probe.py graph 30 240 320 $B 2>&1 | grep "seg engine" | sed "s/^/WIN=$WIN /" >> $O/engine.txt; done; done
cat $O/engine.txt
rocprofv3 --kernel-trace --output-format csv -d $O/kts -o kts -- python tools/seg_probe.py graph 10 240 320 8 > /dev/null 2> $O/kts.err
SEG_PACKS=2 python tools/seg_seq.py $(find $O/kts -name '*kernel_trace.csv' | head -1) > $O/seg_launch_timeline_b8.txt 2>&1
rm -rf $O/kts
OJF_SEG_TRACE=1 python tools/seg_probe.py eager 1 240 320 8 2>&1 | grep "^segconv" | tail -100 > $O/seg_forms_b8.txt
tail -80 $O/seg_launch_timeline_b8.txt
