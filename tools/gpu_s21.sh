#!/bin/bash
# session 4: vec_hscan ablations (no epilogue / no per-step Q DMA / no MFMA) + PMC passes of the CB=2 scan
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s21
mkdir -p $O
cd $GRAFT_REPO_ROOT
T=$GRAFT_REPO_ROOT/typesense_amd
VEC_BATCHES=64,256,512 TSGPU_LIBS=$T/libtsgpu.so,$T/libtsgpu_abl1.so,$T/libtsgpu_abl3.so,$T/libtsgpu_abl5.so timeout 600 python tools/sweep_vec.py > $O/abl.txt 2> $O/abl.err; cat $O/abl.txt; tail -3 $O/abl.err
RX="vec_hscan"
VEC="python $GRAFT_REPO_ROOT/bench.py --workload vector --vec-batch 256 --no-cpu-baseline --steps 2 --warmup 1"
run_pmc() { name=$1; ctr=$2; shift 2
  ( cd /tmp && timeout 420 rocprofv3 --kernel-trace --pmc $ctr --kernel-include-regex "$RX" -f csv -d $O/$name -- $VEC > $O/$name.log 2>&1 )
  python tools/pmc_summary.py $O/$name > $O/$name.txt 2>&1; cat $O/$name.txt; rm -rf $O/$name; }
#run_pmc pmc_vec_sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
#run_pmc pmc_vec_sq2 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"
#run_pmc pmc_vec_fetch "FETCH_SIZE"
