# round 6, GPU call A: the GPU test tier with the new at-size tests (durations), the two-field leg with / without the staged multi-field score loads, the keyword headline leg
# usage: bash tools/experiments/run_r06_a.sh "<kwgeneral variants, '-' = default lib>" "<keyword variants>" <tag> [skip_tests]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_$3; mkdir -p $O
if [ "${4:-}" != "skip_tests" ]; then
  ( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 ) > $O/pytest_gpu.txt 2>&1; tail -45 $O/pytest_gpu.txt
fi
for v in $1; do
  L=""; if [ "$v" != "-" ]; then L=$PWD/typesense_amd/variants/libtsgpu_$v.so; fi
  ( cd /tmp && TSGPU_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace_kwg_$v -- python $GRAFT_REPO_ROOT/bench.py --workload kwgeneral --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/kwg_$v.json 2> $GRAFT_REPO_ROOT/$O/kwg_$v.err )
  python profiles/summarize_rocprof.py $O/trace_kwg_$v > $O/rocprof_kwgeneral_$v.txt 2>&1; head -14 $O/rocprof_kwgeneral_$v.txt | cut -c1-200
  python - $O/kwg_$v.json $v <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    g = d.get("general_kernels") or d
    print(sys.argv[2], {k: {kk.split(" ")[0]: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items() if isinstance(vv, (int, float)) or kk == "parity"} for k, v in g.items() if isinstance(v, dict)})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  rm -rf $O/trace_kwg_$v
done
for v in $2; do
  L=""; if [ "$v" != "-" ]; then L=$PWD/typesense_amd/variants/libtsgpu_$v.so; fi
  TSGPU_LIB=$L timeout 600 python bench.py --workload keyword --no-extras --no-cpu-baseline --steps 10 --warmup 3 --detail-out $O/detail_$v.json > $O/kw_$v.json 2> $O/kw_$v.err
  python - $O/detail_$v.json $v <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print(sys.argv[2], "value %.0f dev_only %.0f find %.3f find+score %.3f parity %s" % (d["value"], d["value_device_only"], r["find_kernel_ms"], r["kernel_ms"], d.get("parity")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
