# usage: bash tools/run_variants.sh "<variant names, '-' = the default library>" [sweep json]
SW=${2:-'[{"kw_pair_blocks":1},{"kw_pair_blocks":0},{"kw_pair_blocks":1}]'}
for v in $1; do
  if [ "$v" = "-" ]; then L=""; else L=typesense_amd/variants/libtsgpu_$v.so; fi
  echo "== variant: $v"
  TSGPU_LIB=$L KW_BATCHES=${KW_BATCHES:-10000,64} KW_SWEEP="$SW" python tools/sweep_kw.py 2>&1 | grep n_q | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['n_q'], d['opts'], 'wall %.3f search %.3f' % (d['wall_ms'], d['search_ms']))"
done
