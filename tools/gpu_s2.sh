#!/bin/bash
# GPU session 2 (round 1): sanity tests, PMC counters for the keyword + vector kernels, knob sweep.
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s2
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
RX='kw_search_kernel|kw_merge_kernel|vec_knn_kernel|vec_merge_kernel'
KW="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
VEC="python bench.py --workload vector --n-docs 2000000 --steps 2 --warmup 1 --no-cpu-baseline"
run_pmc() { # name counters cmd...
  local name=$1; local ctr=$2; shift 2
  timeout 420 rocprofv3 --kernel-trace --pmc $ctr --kernel-include-regex "$RX" -f csv -d $O/$name -- "$@" > $O/$name.log 2>&1
  python tools/pmc_summary.py $O/$name > $O/$name.txt 2>&1
  # raw csv can be large: keep only the summary
  find $O/$name -name '*.csv' -size +2M -delete
}
run_pmc kw_fetch "FETCH_SIZE" $KW
run_pmc kw_sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" $KW
run_pmc kw_sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" $KW
run_pmc kw_tcc "TCC_HIT_sum TCC_MISS_sum" $KW
run_pmc vec_sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" $VEC
KW_SWEEP='[{"kw_chunk_blocks":64},{"kw_chunk_blocks":16},{"kw_chunk_blocks":32},{"kw_chunk_blocks":128},{"kw_chunk_blocks":512}]' timeout 420 python tools/sweep_kw.py > $O/sweep_kw.txt 2>&1
cat $O/*.txt | tail -150
