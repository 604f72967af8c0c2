# usage: bash tools/experiments/run_r05_kwg.sh "<variants, '-' = default lib>" <tag> : general-kernel bench legs per variant (+ the TSGPU_PROF phase line for *prof* variants)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_$2; mkdir -p $O
for v in $1; do
  L=""; if [ "$v" != "-" ]; then L=$PWD/typesense_amd/variants/libtsgpu_$v.so; fi
  P=""; case $v in *prof*) P=1;; esac
  KW_PROF=$P TSGPU_LIB=$L timeout 600 python bench.py --workload kwgeneral --steps 10 --warmup 3 --no-cpu-baseline > $O/kwg_$v.json 2> $O/kwg_$v.err
  grep PROF $O/kwg_$v.err
  python - $O/kwg_$v.json $v <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    g = d.get("general_kernels") or d
    print(sys.argv[2], {k: {kk.split(" ")[0]: round(vv, 3) for kk, vv in v.items() if isinstance(vv, float) and ("ms" in kk)} for k, v in g.items() if isinstance(v, dict)})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
