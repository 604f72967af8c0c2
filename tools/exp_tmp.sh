for w in 300 0 600 150; do
python bench.py --workload vector --no-cpu-baseline --steps 3 --warmup 1 --hnsw-rows 0 --opt vec_batch_post_window_us=$w > /tmp/v.json 2>/dev/null
python - <<P
import json
d=json.loads(open("/tmp/v.json").read().strip().splitlines()[-1])
c=d["concurrency"] if "concurrency" in d else d["vector"]["concurrency"]
print("post window $w:", round(d["value"]), {k:(round(v,1) if isinstance(v,float) else v) for k,v in c.items() if k in ("value","p50_us","p99_us","queries_per_round","failures")}, c["parity"]["mismatches"])
P
done
python -m pytest tests/test_gpu_at_size.py tests/test_gpu_concurrency.py -x -q 2>&1 | tail -2
