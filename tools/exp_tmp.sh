for cfg in "5000 50" "3000 70" "2500 75" "4000 60" "1250 60" "1250 70" "2000 80"; do
set -- $cfg
python bench.py --workload keyword --no-cpu-baseline --no-extras --steps 3 --warmup 1 --opt kw_host_split_queries=$1 --opt kw_host_split_first_pct=$2 > /tmp/h.json 2>/dev/null
python - <<P
import json
d=json.loads(open("/tmp/h.json").read().strip().splitlines()[-1])
print("min slice $1 first $2%: value", round(d["value"]), "host delivery", round(d.get("value_with_host_delivery",0)))
P
done
