#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s11
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_keyword.py -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
KW_SWEEP='[{"kw_chunk_blocks":64},{"kw_chunk_blocks":128},{"kw_chunk_blocks":256}]' timeout 420 python tools/sweep_kw.py 2>&1 | grep n_q > $O/sweep_kw.txt
cat $O/sweep_kw.txt
KW_PROF=1 TSGPU_LIB=$GRAFT_REPO_ROOT/typesense_amd/libtsgpu_prof.so KW_SWEEP='[{"kw_chunk_blocks":128}]' timeout 420 python tools/sweep_kw.py 2>&1 | grep -E "n_q|PROF" > $O/prof_kw.txt
cat $O/prof_kw.txt
