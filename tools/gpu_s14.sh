#!/bin/bash
# vector scan v3 (no select on in-flight loads): tests + bench at B=64/256/1024 + kernel trace
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s14
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_vector.py -m gpu -x -q > $O/pytest_gpu_vec.txt 2>&1; tail -3 $O/pytest_gpu_vec.txt
for b in 64 256 1024; do
  timeout 600 python bench.py --workload vector --vec-batch $b --no-cpu-baseline > $O/bench_vec_b$b.json 2>$O/bench_vec_b$b.err; tail -1 $O/bench_vec_b$b.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'], d['roofline']['kernel_ms'])"
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_vec -- python $GRAFT_REPO_ROOT/bench.py --workload vector --no-cpu-baseline > $O/prof_vec.log 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocprof.py $O/prof_vec > $O/prof_vec.stats.txt 2>&1; head -5 $O/prof_vec.stats.txt | cut -c1-150
find $O -name "*.db" -delete
