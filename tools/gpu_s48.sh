#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s48
mkdir -p $O
cd /tmp
T=$GRAFT_REPO_ROOT/typesense_amd
for L in libtsgpu.so libtsgpu_sw5.so libtsgpu_sw6.so; do
  TSGPU_LIB=$T/$L KW_BATCHES=10000 KW_SWEEP='[{"kw_two_kernels":1}]' timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$L -- python $GRAFT_REPO_ROOT/tools/sweep_kw.py > $O/prof_$L.log 2>&1
  echo "== $L"; grep n_q $O/prof_$L.log | cut -c1-170; python $GRAFT_REPO_ROOT/profiles/summarize_rocprof.py $O/prof_$L 2>&1 | grep -E "kw_s"
done
find $O -name "*.db" -delete; find $O -name "*.csv" -size +5M -delete
