# round 5: where the candidate-combination leg's time goes: host phases (TSGPU_HOST_TIMING) and a kernel trace filtered to the library's kernels
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_cand; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_keyword.py -q -x -k "candidate" > $O/pytest_cand.txt 2>&1; tail -3 $O/pytest_cand.txt
TSGPU_HOST_TIMING=1 timeout 600 python bench.py --workload kwgeneral --steps 5 --warmup 2 --no-cpu-baseline > $O/kwg.json 2> $O/kwg.err
grep "tsgpu\]" $O/kwg.err | tail -6
python - $O/kwg.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for k, v in d["general_kernels"].items():
    print(k, {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ("value", "ms_per_step", "parity") or "ms" in kk})
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace_kwg -- python $GRAFT_REPO_ROOT/bench.py --workload kwgeneral --no-cpu-baseline --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$O/trace_kwg.log 2>&1 )
python profiles/summarize_rocprof.py $O/trace_kwg tsgpu > $O/rocprof_kwgeneral_tsgpu_stats.txt 2>&1; cat $O/rocprof_kwgeneral_tsgpu_stats.txt | cut -c1-130
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete; rm -rf $O/trace_kwg
