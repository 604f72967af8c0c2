# round 6: bench.py --gpus 8 rehearsed end to end on ONE MI355X — eight ranks share the device, the group's exchange goes over the HOST transport (gloo callbacks)
# usage: bash tools/experiments/run_r06_gpus8.sh [n_docs] [world]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
N=${1:-2000000}; W=${2:-8}
O=gpurun_out/r06_gpus$W; mkdir -p $O
( time TSGPU_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port 29631 \
    bench.py --gpus $W --steps 3 --warmup 1 --n-docs $N --batch 2000 --vec-batch 64 --no-cpu-baseline --detail-out $O/detail.json ) > $O/bench_gpus$W.json 2> $O/bench_gpus$W.err
echo "rc=$?"; tail -3 $O/bench_gpus$W.err | cut -c1-300
python - $O/bench_gpus$W.json <<'PY'
import json, sys
lines = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")]
if not lines:
    print("NO JSON LINE"); sys.exit(0)
d = json.loads(lines[-1])
print({k: d.get(k) for k in ("value", "n_gpus", "scaling", "ms_per_step")}, d.get("shard_parity"), d.get("exchange_check"), {k: (d.get(k) or {}).get("shard_parity") for k in ("vector", "hybrid")}, (d.get("replicas") or {}).get("value"))
PY
