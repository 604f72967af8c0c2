#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s27
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_vector.py -m gpu -x -q > $O/pytest_gpu_vector.txt 2>&1; tail -3 $O/pytest_gpu_vector.txt
for B in 256 512 1024 129; do timeout 300 python bench.py --workload vector --vec-batch $B --no-cpu-baseline > $O/bench_vec_b$B.json 2> $O/bench_vec_b$B.err; python -c "
import json
d=json.loads(open('$O/bench_vec_b$B.json').read().strip().splitlines()[-1]); r=d['roofline']
print('B=$B qps=%.0f ms/step=%.2f scan_ms=%.2f %s frac=%.3f GB/s=%.0f TF=%.0f'%(d['value'], d['ms_per_step'], r['kernel_ms'], r['bound'], r['frac'], r['hbm_GBs'], r['bf16_mfma_TFs']), r['overflow_rounds'], r['prefilter_fallbacks'])"; done
