#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
KW_BATCHES=10000 KW_SWEEP='[{"kw_two_kernels":0},{"kw_two_kernels":1}]' timeout 600 python tools/sweep_kw.py 2>&1 | grep -E "n_q|rror" | cut -c1-190
timeout 900 python -m pytest tests/test_gpu_keyword.py -m gpu -x -q 2>&1 | tail -1
