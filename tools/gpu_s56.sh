#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=$GRAFT_REPO_ROOT/typesense_amd
for L in libtsgpu.so libtsgpu_p4.so libtsgpu_p6.so; do
  echo "== $L"
  KW_BATCHES=10000 KW_SWEEP='[{"kw_two_kernels":1}]' TSGPU_LIB=$T/$L timeout 600 python tools/sweep_kw.py 2>&1 | grep -E "n_q|rror" | cut -c1-190
done
