#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s40
mkdir -p $O
cd /tmp
KW_BATCHES=10000 KW_SWEEP='[{"kw_two_kernels":1}]' timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $O/pmc_sq1 -- python $GRAFT_REPO_ROOT/tools/sweep_kw.py > $O/pmc_sq1.log 2>&1
KW_BATCHES=10000 KW_SWEEP='[{"kw_two_kernels":1}]' timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU -d $O/pmc_sq2 -- python $GRAFT_REPO_ROOT/tools/sweep_kw.py > $O/pmc_sq2.log 2>&1
cd $GRAFT_REPO_ROOT
for p in pmc_sq1 pmc_sq2; do python tools/pmc_summary.py $O/$p "kw_" > $O/$p.txt 2>&1; cat $O/$p.txt; done
find $O -name "*.db" -delete; find $O -name "*.csv" -size +5M -delete
