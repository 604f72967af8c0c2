# round 5: phase profile of kw_find_mf2_kernel (TSGPU_PROF build) + the default build's general-kernel leg
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_$1; mkdir -p $O
for v in - mf_prof; do
  L=""; if [ "$v" != "-" ]; then L=$PWD/typesense_amd/variants/libtsgpu_$v.so; fi
  KW_PROF=1 TSGPU_LIB=$L timeout 600 python bench.py --workload kwgeneral --steps 10 --warmup 3 --no-cpu-baseline > $O/kwg_$v.json 2> $O/kwg_$v.err
  grep PROF $O/kwg_$v.err
  python - $O/kwg_$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
g = d.get("general_kernels") or d
print(sys.argv[2], {k: {kk: round(vv, 3) for kk, vv in v.items() if isinstance(vv, float) and ("ms" in kk)} for k, v in g.items() if isinstance(v, dict)})
PY
done
