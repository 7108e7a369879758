#!/bin/bash
# round 5, run 32: SQ counters of the 64x64 GEMM-shaped form on the 60x80 256 -> 256 3x3 layer (one frame) and of the 128x128 form at four frames
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r5_32; mkdir -p $O
pmc() { # name, counters..., then the layer arguments come from $SHAPE
  n=$1; shift
  (cd /tmp; rocprofv3 --pmc "$@" --output-format csv -d $O/$n -o $n -- python $GRAFT_REPO_ROOT/tools/seg_layer_bench.py $SHAPE 8 > /dev/null 2> $O/$n.err)
  echo "== $SHAPE: $*" >> $O/sq.txt
  python tools/pmc_generic_summary.py $(find $O/$n -name '*counter_collection.csv' | head -1) 2>&1 | grep -i "kernel \|segconv" >> $O/sq.txt
  rm -rf $O/$n
}
for SHAPE in "256 256 3 60 80 1" "256 256 3 60 80 4"; do
pmc a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
pmc b SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pmc c SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
pmc d TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCR_TCP_STALL_CYCLES_sum
done
