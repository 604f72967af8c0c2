# usage: bash tools/experiments/run_r05_conc.sh <tag> "<opt sets separated by ;>"  e.g. "none;kw_two_kernels=0" : the keyword leg with its concurrency extras per option set
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_$1; mkdir -p $O
IFS=';' read -ra SETS <<< "$2"
i=0
for set in "${SETS[@]}"; do
  OPTS=""; if [ "$set" != "none" ]; then for o in $set; do OPTS="$OPTS --opt $o"; done; fi
  timeout 900 python bench.py --workload keyword --no-cpu-baseline --steps 5 --warmup 2 $OPTS --detail-out $O/detail_$i.json > $O/kw_$i.json 2> $O/kw_$i.err
  python - $O/detail_$i.json "$set" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); c = d.get("concurrency") or {}
print(sys.argv[2], "value %.0f dev_only %.0f |" % (d["value"], d["value_device_only"]), " ".join("T%s: %.0f q/s p50 %.0f p99 %.0f |" % (k, v["value"], v["p50_us"], v["p99_us"]) for k, v in c.items()))
PY
  i=$((i+1))
done
