# usage: bash tools/experiments/run_benchkw.sh <tag> [extra bench.py args]: the keyword leg of bench.py (parity + cpu_baseline included), key numbers printed
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/f2
tag=$1; shift
python bench.py --workload keyword --steps 20 --warmup 5 "$@" > gpurun_out/f2/bench_$tag.json 2> gpurun_out/f2/bench_$tag.err
python - gpurun_out/f2/bench_$tag.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.0f (%.3f ms)  device_only %.0f (%.3f ms)  find %.3f  find+score %.3f  merge %.3f" % (d["value"], d["ms_per_step"], d["value_device_only"], d["ms_per_step_device_only"], r["find_kernel_ms"], r["kernel_ms"], r["merge_kernel_ms"]))
print("parity", d.get("parity"), "cpu", d.get("cpu_baseline", {}).get("value"))
for k, v in (d.get("general_kernels") or {}).items(): print(k, v.get("ms_per_step"), v.get("parity"))
c = d.get("concurrency") or {}
for k, v in c.items(): print("threads", k, "q/s %.0f p50 %.0f p99 %.0f" % (v["value"], v["p50_us"], v["p99_us"]))
PY
