#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s37
mkdir -p $O
cd $GRAFT_REPO_ROOT
T=$GRAFT_REPO_ROOT/typesense_amd
for L in libtsgpu.so; do
  echo "== $L" >> $O/sweep_kw.txt
  KW_BATCHES=10000 KW_SWEEP='[{"kw_two_kernels":1}]' TSGPU_LIB=$T/$L timeout 600 python tools/sweep_kw.py 2>&1 | grep -E "n_q|rror" >> $O/sweep_kw.txt
done
cat $O/sweep_kw.txt
cd /tmp
KW_BATCHES=10000 KW_SWEEP='[{"kw_two_kernels":1}]' timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_kw -- python $GRAFT_REPO_ROOT/tools/sweep_kw.py > $O/prof_kw.log 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocprof.py $O/prof_kw > $O/prof_kw.stats.txt 2>&1; head -12 $O/prof_kw.stats.txt
find $O -name "*.db" -delete; find $O -name "*.csv" -size +5M -delete
