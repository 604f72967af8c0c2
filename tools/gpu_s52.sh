#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python tools/sweep_mf.py 2>&1 | grep -E "n_q|rror" | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_keyword.py -m gpu -x -q 2>&1 | tail -2
