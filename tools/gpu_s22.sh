#!/bin/bash
# session 4: keyword tier after filter ids + multi-field; keyword bench (regression check)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s22
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_keyword.py -m gpu -x -q > $O/pytest_gpu_keyword.txt 2>&1; tail -5 $O/pytest_gpu_keyword.txt
timeout 600 python bench.py --workload keyword --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_kw.json 2> $O/bench_kw.err; tail -c 1500 $O/bench_kw.json; tail -3 $O/bench_kw.err
