#!/bin/bash
# round 5, run 22: SQ counters of the GEMM-shaped kernel at four frames per pass
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r5_22; mkdir -p $O
export OJF_SEG_GEMM_MIN=64 SEG_PACKS=8
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python tools/seg_probe.py eager 4 240 320 4 > /dev/null 2> $O/kt.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/p1 -o p1 -- python tools/seg_probe.py eager 4 240 320 4 > /dev/null 2> $O/p1.err
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM --output-format csv -d $O/p2 -o p2 -- python tools/seg_probe.py eager 4 240 320 4 > /dev/null 2> $O/p2.err
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/p3 -o p3 -- python tools/seg_probe.py eager 4 240 320 4 > /dev/null 2> $O/p3.err
OJF_SEG_TRACE=1 python tools/seg_probe.py eager 1 240 320 4 2> $O/trace.txt > /dev/null
python tools/seg_seq.py $(find $O/kt -name '*kernel_trace.csv' | head -1) $(find $O/p1 -name '*counter_collection.csv' | head -1) $(find $O/p2 -name '*counter_collection.csv' | head -1) $(find $O/p3 -name '*counter_collection.csv' | head -1) > $O/seq.txt 2>&1
rm -rf $O/kt $O/p1 $O/p2 $O/p3
