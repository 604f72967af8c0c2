# usage: bash tools/experiments/run_r05_kw.sh "<variants, '-' = default lib>" <tag>: the keyword headline leg (no extras, no oracle) per library variant: find / find+score / touched tile bytes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_$2; mkdir -p $O
for v in $1; do
  L=""; if [ "$v" != "-" ]; then L=$PWD/typesense_amd/variants/libtsgpu_$v.so; fi
  TSGPU_LIB=$L timeout 600 python bench.py --workload keyword --no-extras --no-cpu-baseline --steps 10 --warmup 3 --detail-out $O/detail_$v.json > $O/kw_$v.json 2> $O/kw_$v.err
  python - $O/detail_$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]; t = r.get("touched") or {}
print(sys.argv[2], "value %.0f dev_only %.0f find %.3f find+score %.3f tile_dma %.2f GB requested %.2f GB" % (d["value"], d["value_device_only"], r["find_kernel_ms"], r["kernel_ms"], t.get("find_tile_dma", 0) / 1e9, r.get("touched_bytes_per_launch", 0) / 1e9))
PY
done
