# usage: bash tools/experiments/run_f2b.sh "<variants, '-' = default lib>" <out tag>   (env: KW_SWEEP json, KW_BATCHES, KW_PROF)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/f2
SW=${KW_SWEEP:-'[{"kw_pair_blocks":1}]'}
( for v in $1; do
  if [ "$v" = "-" ]; then L=""; else L=typesense_amd/variants/libtsgpu_$v.so; fi
  echo "== variant: $v"
  TSGPU_LIB=$L KW_PROF=${KW_PROF:-} KW_BATCHES=${KW_BATCHES:-10000} KW_SWEEP="$SW" timeout 400 python tools/sweep_kw.py 2>&1 | grep -E "n_q|PROF|Error|error" | cut -c1-400
done ) > gpurun_out/f2/${2:-b}.txt 2>&1
cat gpurun_out/f2/${2:-b}.txt
