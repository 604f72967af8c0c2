#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s49
mkdir -p $O
cd $GRAFT_REPO_ROOT
T=$GRAFT_REPO_ROOT/typesense_amd
for L in libtsgpu_c512.so libtsgpu_c1024.so; do
  echo "== $L"
  KW_BATCHES=10000,3000,1000 KW_SWEEP='[{"kw_two_kernels":1}]' TSGPU_LIB=$T/$L timeout 600 python tools/sweep_kw.py 2>&1 | grep -E "n_q|rror" | cut -c1-190
done
