#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s25
mkdir -p $O
cd $GRAFT_REPO_ROOT
KW_PROF=1 TSGPU_LIB=$GRAFT_REPO_ROOT/typesense_amd/libtsgpu_prof.so KW_SWEEP='[{"kw_chunk_blocks":0}]' timeout 420 python tools/sweep_kw.py 2>&1 | grep -E "n_q|PROF" > $O/prof_kw.txt
cat $O/prof_kw.txt
timeout 600 python -m pytest tests/test_gpu_keyword.py -m gpu -x -q > $O/pytest_gpu_keyword.txt 2>&1; tail -3 $O/pytest_gpu_keyword.txt
timeout 600 python bench.py --workload keyword --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_kw.json 2> $O/bench_kw.err; python -c "
import json
d=json.loads(open('$O/bench_kw.json').read().strip().splitlines()[-1])
print('BENCH', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['merge_kernel_ms'], d['roofline']['frac'])"
