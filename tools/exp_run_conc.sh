# experiment driver (GPU box): the concurrency leg of the keyword bench under option sets; prints one line per thread count
# usage: bash tools/exp_run_conc.sh "optA=1,optB=2" "optC=3" ...
for set in "$@"; do
  opts=""; for o in ${set//,/ }; do [ "$o" = "-" ] || opts="$opts --opt $o"; done
  python bench.py --workload keyword --no-cpu-baseline --steps 3 --warmup 1 --detail-out /tmp/conc_detail.json $opts > /tmp/conc.json 2>/dev/null   # (the compact stdout line carries only value / p50 / p99 per thread count: the full record is the detail file)
  SET="$set" python - <<P
import json, os
d=json.load(open("/tmp/conc_detail.json"))
def find(o):
    if isinstance(o,dict):
        if "concurrency" in o: return o["concurrency"]
        for v in o.values():
            r=find(v)
            if r: return r
print("==", os.environ["SET"], "batch", round(d["value"]), d["ms_per_step"])
for t,v in find(d).items(): print(t, round(v["value"]), round(v["p50_us"]), round(v["p99_us"]), round(v["queries_per_round"],1), {k:round(x) for k,x in v["us_per_batch"].items()}, {k[:6]:round(x) for k,x in v["us_per_coalesced_call"].items()}, v["parity"]["mismatches"])
P
done
