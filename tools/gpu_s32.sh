#!/bin/bash
# candidate-combination fold (tsgpu_keyword_search_candidates_batch) on the GPU: parity tests + keyword regression
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s32
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_keyword.py -m gpu -x -q > $O/pytest_gpu_keyword.txt 2>&1; tail -5 $O/pytest_gpu_keyword.txt
