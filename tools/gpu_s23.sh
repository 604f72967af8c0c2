#!/bin/bash
# session 4: full GPU tier (tests, bench, smoke) + rocprof kernel-trace and FETCH_SIZE pmc of the keyword + vector bench + vector batch sweep
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s23
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
( time timeout 900 python bench.py ) > $O/bench_all.json 2> $O/bench_all.err; tail -c 600 $O/bench_all.json; tail -5 $O/bench_all.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
for B in 16 64 512 1024; do timeout 300 python bench.py --workload vector --vec-batch $B --no-cpu-baseline > $O/bench_vec_b$B.json 2> $O/bench_vec_b$B.err; done
timeout 300 python bench.py --workload vector --vec-batch 256 --no-cpu-baseline --opt vec_prefilter=0 > $O/bench_vec_b256_fp32scan.json 2> $O/bench_vec_b256_fp32scan.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_kw -- python $GRAFT_REPO_ROOT/bench.py --workload keyword --no-cpu-baseline > $O/prof_kw.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_vec -- python $GRAFT_REPO_ROOT/bench.py --workload vector --no-cpu-baseline > $O/prof_vec.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_kw_fetch -- python $GRAFT_REPO_ROOT/bench.py --workload keyword --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_kw_fetch.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_vec_fetch -- python $GRAFT_REPO_ROOT/bench.py --workload vector --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_vec_fetch.log 2>&1
cd $GRAFT_REPO_ROOT
for p in prof_kw prof_vec; do python profiles/summarize_rocprof.py $O/$p > $O/$p.stats.txt 2>&1; done
for p in pmc_kw_fetch pmc_vec_fetch; do python tools/pmc_summary.py $O/$p > $O/$p.txt 2>&1; done
find $O -name "*.db" -delete; find $O -name "*.csv" -size +5M -delete
du -sh $O
