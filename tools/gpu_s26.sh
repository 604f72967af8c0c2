#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s26
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_vector.py -m gpu -x -q -s -k "hnsw" > $O/pytest_hnsw.txt 2>&1; tail -8 $O/pytest_hnsw.txt
