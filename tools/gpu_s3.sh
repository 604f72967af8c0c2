#!/bin/bash
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s3
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_vector.py -m gpu -x -q > $O/pytest_gpu_vec.txt 2>&1; tail -5 $O/pytest_gpu_vec.txt
for B in 256 64 1024; do
  timeout 600 python bench.py --workload vector --steps 3 --warmup 1 --batch $B > $O/bench_vec_b$B.json 2> $O/bench_vec_b$B.err; tail -1 $O/bench_vec_b$B.json
done
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_vec -- python bench.py --workload vector --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_vec.log 2>&1
python - <<'PY'
import csv, glob, os
for f in glob.glob(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/s3/prof_vec/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print({k: r[k] for k in list(r)[:8]})
PY
