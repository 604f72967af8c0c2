for pb in 0 1; do
TSGPU_LIB=typesense_amd/variants/libtsgpu_prof.so KW_PROF=1 KW_BATCHES=10000 KW_SWEEP="[{\"kw_pair_blocks\":$pb}]" python tools/sweep_kw.py 2>&1 | grep -E "n_q|PROF"
done
