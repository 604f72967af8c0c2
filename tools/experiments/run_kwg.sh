# usage: bash tools/experiments/run_kwg.sh "<variants, '-' = default lib>" <tag>: the general-kernel bench legs per library variant
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/f2
for v in $1; do
  if [ "$v" = "-" ]; then L=""; else L=$PWD/typesense_amd/variants/libtsgpu_$v.so; fi
  TSGPU_LIB=$L python bench.py --workload kwgeneral --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/f2/kwg_$2_$v.json 2> gpurun_out/f2/kwg_$2_$v.err
  python - gpurun_out/f2/kwg_$2_$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
g = d.get("general_kernels") or d
print(sys.argv[2], {k: round(v["ms_per_step"], 2) for k, v in g.items() if isinstance(v, dict) and "ms_per_step" in v})
PY
done
