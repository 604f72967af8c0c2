for o in 1 0 1 0; do
python bench.py --workload hybrid --no-cpu-baseline --no-extras --steps 10 --warmup 3 --opt hybrid_overlap=$o > /tmp/h.json 2>/dev/null
python - <<P
import json
d=json.loads(open("/tmp/h.json").read().strip().splitlines()[-1])
h=d if "hybrid" not in d else d["hybrid"]
print("overlap $o:", round(h["value"]), round(h["ms_per_step"],3))
P
done
python -m pytest tests/test_gpu_vector.py tests/test_gpu_at_size.py -x -q -k "hybrid" 2>&1 | tail -2
