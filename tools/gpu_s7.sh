#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s7
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_keyword.py -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
SW='[{"kw_chunk_blocks":64},{"kw_chunk_blocks":128}]'
for L in libtsgpu.so libtsgpu_t1024.so libtsgpu_e1.so libtsgpu_e2.so libtsgpu_e3.so; do
  echo "== $L" >> $O/sweep_kw.txt
  KW_SWEEP="$SW" TSGPU_LIB=$GRAFT_REPO_ROOT/typesense_amd/$L timeout 420 python tools/sweep_kw.py 2>&1 | grep n_q >> $O/sweep_kw.txt
done
cat $O/sweep_kw.txt
timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench_kw.json 2> $O/bench_kw.err; tail -1 $O/bench_kw.json
