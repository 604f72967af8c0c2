#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s42
mkdir -p $O
cd $GRAFT_REPO_ROOT
TSGPU_HOST_TIMING=1 KW_BATCHES=10000 KW_SWEEP='[{"kw_sort_work":1}]' timeout 600 python tools/sweep_kw.py 2>&1 | grep -E "n_q|tsgpu\]" | tail -4
