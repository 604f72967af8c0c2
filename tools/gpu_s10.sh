#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s10
mkdir -p $O
cd $GRAFT_REPO_ROOT
KW_PROF=1 TSGPU_LIB=$GRAFT_REPO_ROOT/typesense_amd/libtsgpu_prof.so KW_SWEEP='[{"kw_chunk_blocks":128}]' timeout 420 python tools/sweep_kw.py 2>&1 | grep -E "n_q|PROF" > $O/prof_kw.txt
cat $O/prof_kw.txt
