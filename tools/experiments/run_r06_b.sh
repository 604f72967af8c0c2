# round 6, GPU call B: keyword headline leg per option set (find4 vs find2), kernel stats, the keyword GPU tests
# usage: bash tools/experiments/run_r06_b.sh "<variants, '-' = default lib>" "<option sets: name=v,name=v ... ; '-' = defaults>" <tag> ["pytest -k expression" | skip]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_$3; mkdir -p $O
if [ "${4:-skip}" != "skip" ]; then
  ( time timeout 1200 python -m pytest tests -m gpu -x -q -k "$4" ) > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt
fi
for v in $1; do
  L=""; if [ "$v" != "-" ]; then L=$PWD/typesense_amd/variants/libtsgpu_$v.so; fi
  for os in $2; do
    OPTS=""; if [ "$os" != "-" ]; then for o in ${os//,/ }; do OPTS="$OPTS --opt $o"; done; fi
    tag="${v}_${os//[=,]/_}"
    ( cd /tmp && TSGPU_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace_$tag -- python $GRAFT_REPO_ROOT/bench.py --workload keyword --no-extras --no-cpu-baseline --steps 10 --warmup 3 $OPTS --detail-out $GRAFT_REPO_ROOT/$O/detail_$tag.json > $GRAFT_REPO_ROOT/$O/kw_$tag.json 2> $GRAFT_REPO_ROOT/$O/kw_$tag.err )
    python profiles/summarize_rocprof.py $O/trace_$tag "kw_" > $O/rocprof_keyword_$tag.txt 2>&1; head -8 $O/rocprof_keyword_$tag.txt | cut -c1-190
    python - $O/detail_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print(sys.argv[2], "value %.0f dev_only %.0f find %.3f find+score %.3f parity %s" % (d["value"], d["value_device_only"], r["find_kernel_ms"], r["kernel_ms"], json.dumps(d.get("parity"))[:300]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
    rm -rf $O/trace_$tag
  done
done
