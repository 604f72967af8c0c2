#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s29
mkdir -p $O
cd $GRAFT_REPO_ROOT
KW_SWEEP='[{"kw_chunk_blocks":0}]' timeout 420 python tools/sweep_kw.py 2>&1 | grep -E "n_q" > $O/sweep_kw.txt; cat $O/sweep_kw.txt | cut -c1-220
timeout 600 python bench.py --workload hybrid --no-cpu-baseline > $O/bench_hybrid.json 2> $O/bench_hybrid.err; python -c "
import json
d=json.loads(open('$O/bench_hybrid.json').read().strip().splitlines()[-1])
print('HYB', d['value'], d['ms_per_step'], 'vec', d['vector']['value'], d['vector']['ms_per_step'])"
timeout 600 python -m pytest tests/test_gpu_keyword.py -m gpu -x -q 2>&1 | tail -2
