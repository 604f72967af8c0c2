#!/bin/bash
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s5
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_keyword.py -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench_kw.json 2> $O/bench_kw.err; tail -1 $O/bench_kw.json
SW='[{"kw_chunk_blocks":64},{"kw_chunk_blocks":128}]'
for L in libtsgpu.so libtsgpu_r4.so libtsgpu_r16.so; do
  echo "== $L" >> $O/sweep_kw.txt
  KW_SWEEP="$SW" TSGPU_LIB=$GRAFT_REPO_ROOT/typesense_amd/$L timeout 420 python tools/sweep_kw.py >> $O/sweep_kw.txt 2>&1
done
cat $O/sweep_kw.txt
RX='kw_search_kernel|kw_merge_kernel'
KW="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
run_pmc() { local name=$1; local ctr=$2; shift 2
  timeout 420 rocprofv3 --kernel-trace --pmc $ctr --kernel-include-regex "$RX" -f csv -d $O/$name -- "$@" > $O/$name.log 2>&1
  python tools/pmc_summary.py $O/$name > $O/$name.txt 2>&1
  find $O/$name -name '*.csv' -size +2M -delete; }
run_pmc kw_sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" $KW
run_pmc kw_sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" $KW
cat $O/kw_*.txt
