#!/bin/bash
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s4
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench_kw.json 2> $O/bench_kw.err; tail -1 $O/bench_kw.json
SW='[{"kw_chunk_blocks":64},{"kw_chunk_blocks":128},{"kw_chunk_blocks":256}]'
for L in libtsgpu.so libtsgpu_r4.so libtsgpu_r16.so; do
  echo "== $L" >> $O/sweep_kw.txt
  KW_SWEEP="$SW" TSGPU_LIB=$GRAFT_REPO_ROOT/typesense_amd/$L timeout 420 python tools/sweep_kw.py >> $O/sweep_kw.txt 2>&1
done
cat $O/sweep_kw.txt
RX='kw_search_kernel|kw_merge_kernel|vec_scan_kernel|vec_select_kernel'
KW="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
VEC="python bench.py --workload vector --n-docs 4000000 --steps 2 --warmup 1 --no-cpu-baseline"
run_pmc() { local name=$1; local ctr=$2; shift 2
  timeout 420 rocprofv3 --kernel-trace --pmc $ctr --kernel-include-regex "$RX" -f csv -d $O/$name -- "$@" > $O/$name.log 2>&1
  python tools/pmc_summary.py $O/$name > $O/$name.txt 2>&1
  find $O/$name -name '*.csv' -size +2M -delete; }
run_pmc kw_fetch "FETCH_SIZE" $KW
run_pmc kw_sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" $KW
run_pmc kw_sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" $KW
run_pmc vec_sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" $VEC
run_pmc vec_sq2 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" $VEC
run_pmc vec_fetch "FETCH_SIZE" $VEC
cat $O/kw_*.txt $O/vec_*.txt
