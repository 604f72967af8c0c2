#!/bin/bash
# session 5: full GPU tier on the two-kernel keyword form — PMC FETCH_SIZE (keyword), tests, bench (all workloads), smoke, rocprof kernel-trace
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s44
mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_kw_fetch -- python $GRAFT_REPO_ROOT/bench.py --workload keyword --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_kw_fetch.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/pmc_kw_fetch > $O/pmc_kw_s5_fetch.txt 2>&1
cp $O/pmc_kw_s5_fetch.txt profiles/r01/pmc_kw_s5_fetch.txt
grep -E "kw_" $O/pmc_kw_s5_fetch.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
( time timeout 900 python bench.py ) > $O/bench_all.json 2> $O/bench_all.err; tail -c 400 $O/bench_all.json; tail -4 $O/bench_all.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_kw -- python $GRAFT_REPO_ROOT/bench.py --workload keyword --no-cpu-baseline > $O/prof_kw.log 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocprof.py $O/prof_kw > $O/prof_kw.stats.txt 2>&1; grep -E "kw_" $O/prof_kw.stats.txt
find $O -name "*.db" -delete; find $O -name "*.csv" -size +5M -delete
du -sh $O
