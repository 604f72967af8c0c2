#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
KW_BATCHES=10000 KW_SWEEP='[{"kw_cost_probe_x100":0},{"kw_cost_probe_x100":5},{"kw_cost_probe_x100":10},{"kw_cost_probe_x100":20},{"kw_cost_probe_x100":40},{"kw_cost_probe_x100":0,"kw_cost_r_x10":20},{"kw_cost_r_x10":5},{"kw_cost_r_x10":10,"kw_cost_probe_x100":10,"kw_cost_fixed":8},{"kw_cost_fixed":16,"kw_cost_probe_x100":0}]' timeout 600 python tools/sweep_kw.py 2>&1 | grep -E "n_q|rror" | cut -c1-200
