#!/bin/bash
# rehearsal of bench.py's N>1 paths with 2 ranks on the one GPU of this box (gloo collectives): default replicas mode, then --dist-mode shards
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s31
mkdir -p $O
cd $GRAFT_REPO_ROOT
TSGPU_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --n-docs 2000000 > $O/bench_2rank_replicas.json 2> $O/bench_2rank_replicas.err; tail -c 1800 $O/bench_2rank_replicas.json; tail -15 $O/bench_2rank_replicas.err
TSGPU_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 --n-docs 2000000 --dist-mode shards > $O/bench_2rank_shards.json 2> $O/bench_2rank_shards.err; tail -c 1200 $O/bench_2rank_shards.json; tail -15 $O/bench_2rank_shards.err
