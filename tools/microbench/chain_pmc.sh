#!/bin/bash
# SQ / LDS counters of tools/microbench/chain_bench.exe (dense_pair_kernel x 5 against dense_chain_kernel); through gpurun
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/chain_pmc
rm -rf $O; mkdir -p $O
B="./tools/microbench/chain_bench.exe 240 320 20"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/a -o a -- $B > $O/run_a.txt 2> $O/a.err
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $O/b -o b -- $B > $O/run_b.txt 2> $O/b.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/c -o c -- $B > $O/run_c.txt 2> $O/c.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -o k -- $B > $O/run_k.txt 2> $O/k.err
for p in a b c; do python tools/pmc_generic_summary.py $(find $O/$p -name '*counter_collection.csv' | head -1) > $O/sum_$p.txt 2>&1; done
python tools/pmc_sq_summary.py $(find $O/a -name '*counter_collection.csv' | head -1) $O/sq.json > $O/sq.txt 2>&1
cp $(find $O/k -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
rm -rf $O/a $O/b $O/c $O/k
cat $O/sum_a.txt $O/sum_b.txt $O/sum_c.txt $O/sq.txt; head -5 $O/kernel_stats.csv
