#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s35
mkdir -p $O
cd $GRAFT_REPO_ROOT
T=$GRAFT_REPO_ROOT/typesense_amd
for L in libtsgpu.so libtsgpu_exp5.so libtsgpu_exp6.so; do
  echo "== $L" >> $O/abl_kw.txt
  KW_BATCHES=10000 KW_SWEEP='[{"kw_chunk_blocks":0}]' TSGPU_LIB=$T/$L timeout 420 python tools/sweep_kw.py 2>&1 | grep -E "n_q" >> $O/abl_kw.txt
done
cat $O/abl_kw.txt
