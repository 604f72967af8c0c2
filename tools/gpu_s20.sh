#!/bin/bash
# session 4, GPU call 1: bf16-prefilter k-NN — parity on the real matrix cores, then config-3 throughput sweeps
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s20
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_vector.py -m gpu -x -q > $O/pytest_gpu_vector.txt 2>&1; tail -5 $O/pytest_gpu_vector.txt
for B in 256 64 1024; do
  timeout 600 python bench.py --workload vector --vec-batch $B --steps 5 --warmup 2 > $O/bench_vec_pf_b$B.json 2> $O/bench_vec_pf_b$B.err; tail -c 1800 $O/bench_vec_pf_b$B.json; tail -3 $O/bench_vec_pf_b$B.err
done
timeout 600 python bench.py --workload vector --vec-batch 256 --steps 5 --warmup 2 --no-cpu-baseline --opt vec_sample_tiles=2048 > $O/bench_vec_pf_b256_s2048.json 2> $O/bench_vec_pf_b256_s2048.err; tail -c 1500 $O/bench_vec_pf_b256_s2048.json
timeout 600 python bench.py --workload vector --vec-batch 16 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_vec_pf_b16.json 2> $O/bench_vec_pf_b16.err; tail -c 1500 $O/bench_vec_pf_b16.json
