#!/bin/bash
# keyword kernel ablations at 10M docs / 10K queries: full, EXP=1 (stage 1 complete, survivors dropped), EXP=2 (no slot search), EXP=4 (no scoring)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s33
mkdir -p $O
cd $GRAFT_REPO_ROOT
T=$GRAFT_REPO_ROOT/typesense_amd
for L in libtsgpu.so libtsgpu_exp1.so libtsgpu_exp2.so libtsgpu_exp4.so; do
  echo "== $L" >> $O/abl_kw.txt
  KW_BATCHES=10000 KW_SWEEP='[{"kw_chunk_blocks":0}]' TSGPU_LIB=$T/$L timeout 420 python tools/sweep_kw.py 2>&1 | grep -E "n_q" >> $O/abl_kw.txt
done
cat $O/abl_kw.txt
