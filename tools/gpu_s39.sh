#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s39
mkdir -p $O
cd $GRAFT_REPO_ROOT
KW_BATCHES=10000,1000,100 KW_SWEEP='[{"kw_two_kernels":1}]' timeout 600 python tools/sweep_kw.py 2>&1 | grep -E "n_q|rror" > $O/sweep_kw.txt; cat $O/sweep_kw.txt
timeout 900 python -m pytest tests/test_gpu_keyword.py -m gpu -x -q > $O/pytest_gpu_keyword.txt 2>&1; tail -3 $O/pytest_gpu_keyword.txt
