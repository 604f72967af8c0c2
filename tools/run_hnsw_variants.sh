# usage: bash tools/run_hnsw_variants.sh "<variants, - = default>" [args of exp_hnsw.py]
V=$1; shift
for v in $V; do
  if [ "$v" = "-" ]; then L=""; else L=typesense_amd/variants/libtsgpu_$v.so; fi
  echo "== $v"
  TSGPU_LIB=$L timeout 600 python tools/exp_hnsw.py "$@" 2>&1 | grep "B=\|HNSW_PROF" | tail -4 | cut -c1-170
done
