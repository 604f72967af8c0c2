#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s47
mkdir -p $O
cd $GRAFT_REPO_ROOT
KW_BATCHES=10000,1000,256,100,16 KW_SWEEP='[{"kw_max_partials":16}]' timeout 600 python tools/sweep_kw.py 2>&1 | grep -E "n_q|rror" > $O/sweep_kw.txt; cat $O/sweep_kw.txt | cut -c1-200
