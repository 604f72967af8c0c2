cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/f2
( for v in - nobar -; do
  if [ "$v" = "-" ]; then L=""; else L=typesense_amd/variants/libtsgpu_$v.so; fi
  echo "== variant: $v"
  TSGPU_LIB=$L KW_BATCHES=10000 KW_SWEEP='[{"kw_device_plan_min_queries":512}]' timeout 400 python tools/sweep_kw.py 2>&1 | grep -E "n_q|PROF"
done
echo "== prof"
TSGPU_LIB=typesense_amd/variants/libtsgpu_prof.so KW_PROF=1 KW_BATCHES=10000 KW_SWEEP='[{"kw_pair_blocks":1}]' timeout 400 python tools/sweep_kw.py 2>&1 | grep -E "n_q|PROF"
) > gpurun_out/f2/a.txt 2>&1
cat gpurun_out/f2/a.txt
