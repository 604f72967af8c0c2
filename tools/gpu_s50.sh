#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
KW_BATCHES=10000 KW_SWEEP='[{"kw_cost_fixed":4},{"kw_cost_fixed":1},{"kw_cost_fixed":2},{"kw_cost_fixed":8},{"kw_cost_fixed":16},{"kw_cost_fixed":64}]' timeout 600 python tools/sweep_kw.py 2>&1 | grep -E "n_q|rror" | cut -c1-190
