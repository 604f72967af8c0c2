#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s6
mkdir -p $O
cd $GRAFT_REPO_ROOT
SW='[{"kw_chunk_blocks":64}]'
for L in libtsgpu_e1.so libtsgpu_e2.so libtsgpu_e3.so; do
  echo "== $L" >> $O/sweep_kw.txt
  KW_SWEEP="$SW" TSGPU_LIB=$GRAFT_REPO_ROOT/typesense_amd/$L timeout 420 python tools/sweep_kw.py 2>&1 | grep n_q >> $O/sweep_kw.txt
done
cat $O/sweep_kw.txt
