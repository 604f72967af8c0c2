#!/bin/bash
# rehearsal of bench.py's N>1 path (doc-range shards, all-gather of per-shard top-K, merge) with 2 ranks on the one GPU of this box
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s24
mkdir -p $O
cd $GRAFT_REPO_ROOT
TSGPU_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --n-docs 2000000 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err; tail -c 1500 $O/bench_2rank_gloo.json; tail -15 $O/bench_2rank_gloo.err
timeout 300 python bench.py --steps 2 --warmup 1 --n-docs 2000000 --no-cpu-baseline > $O/bench_1rank_2m.json 2> $O/bench_1rank_2m.err; tail -c 600 $O/bench_1rank_2m.json
timeout 600 python -m pytest tests/test_gpu_keyword.py -m gpu -x -q -k "shard_merge" > $O/pytest_shard_merge.txt 2>&1; tail -3 $O/pytest_shard_merge.txt
