# round 5: the pipelined two-field find kernel (kw_find_mf2_kernel) on the GPU — tests, then the general-kernel bench leg per variant
# usage (through gpurun): bash tools/experiments/run_r05_mf.sh "<variants, '-' = default lib, 'old' = default lib with kw_mf_pipelined=0>" <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_$2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_keyword.py -q -x -k "pipelined or multi_field or counts_the_bytes" > $O/pytest_mf.txt 2>&1; tail -3 $O/pytest_mf.txt
for v in $1; do
  L=""; OPT=""
  if [ "$v" = "old" ]; then OPT="--opt kw_mf_pipelined=0"; elif [ "$v" != "-" ]; then L=$PWD/typesense_amd/variants/libtsgpu_$v.so; fi
  TSGPU_LIB=$L timeout 600 python bench.py --workload kwgeneral --steps 10 --warmup 3 --no-cpu-baseline $OPT > $O/kwg_$v.json 2> $O/kwg_$v.err
  python - $O/kwg_$v.json $v <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    g = d.get("general_kernels") or d
    print(sys.argv[2], {k: {kk: round(vv, 3) for kk, vv in v.items() if isinstance(vv, float) and ("ms" in kk)} for k, v in g.items() if isinstance(v, dict)})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace_kwg -- python $GRAFT_REPO_ROOT/bench.py --workload kwgeneral --no-cpu-baseline --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$O/trace_kwg.log 2>&1 )
python profiles/summarize_rocprof.py $O/trace_kwg > $O/rocprof_kwgeneral_$2_stats.txt 2>&1; head -12 $O/rocprof_kwgeneral_$2_stats.txt
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete; rm -rf $O/trace_kwg
