#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s28
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --workload hybrid --no-cpu-baseline > $O/bench_hybrid.json 2> $O/bench_hybrid.err; python -c "
import json
d=json.loads(open('$O/bench_hybrid.json').read().strip().splitlines()[-1])
print('HYB', d['value'], d['ms_per_step'], 'vec', d['vector']['value'], d['vector']['ms_per_step'])"
timeout 600 python -m pytest tests/test_gpu_vector.py -m gpu -x -q -k "hybrid" 2>&1 | tail -2
