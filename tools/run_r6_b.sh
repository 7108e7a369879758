set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6b
rm -rf $O; mkdir -p $O
L=$O/layers.txt
for B in 1 2 4 8; do
  for WS in 0 1; do
    OJF_SEG_WS=$WS OJF_SEG_TRACE=1 python tools/seg_layer_bench.py 256 256 3 60 80 $B 64 4 2>&1 | grep -E "us per launch|^segconv" | sort | uniq -c | sort -rn | head -3 >> $L
  done
done
for shape in "512 2048 1 15 20" "2048 512 1 15 20" "512 512 3 15 20" "256 1024 1 30 40" "1024 256 1 30 40" "256 256 3 30 40" "64 256 1 60 80" "256 64 1 60 80" "64 64 3 60 80"; do
  for B in 1 4 8; do
    for WS in 0 1; do
      OJF_SEG_WS=$WS OJF_SEG_TRACE=1 python tools/seg_layer_bench.py $shape $B 64 4 2>&1 | grep -E "us per launch|^segconv" | sort | uniq -c | sort -rn | head -2 >> $L
    done
  done
done
for B in 1 2 4 8; do for WS in 0 1; do OJF_SEG_WS=$WS python tools/seg_probe.py graph 30 240 320 $B 2>&1 | grep "seg engine" | sed "s/^/WS=$WS /" >> $O/engine.txt; done; done
cat $L $O/engine.txt
