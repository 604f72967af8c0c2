#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s43
mkdir -p $O
cd $GRAFT_REPO_ROOT
KW_BATCHES=10000 KW_SWEEP='[{"kw_chunk_blocks":0},{"kw_chunk_blocks":128},{"kw_chunk_blocks":64},{"kw_chunk_blocks":32}]' timeout 600 python tools/sweep_kw.py 2>&1 | grep -E "n_q|rror" > $O/sweep_kw.txt; cat $O/sweep_kw.txt
