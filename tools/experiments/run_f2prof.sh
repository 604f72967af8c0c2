# usage: bash tools/experiments/run_f2prof.sh <out tag>   (env: KW_SWEEP json, TSGPU_LIB): rocprofv3 kernel stats of one sweep_kw.py run
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/f2
export TMPDIR=/tmp
SW=${KW_SWEEP:-'[{"kw_pair_blocks":1}]'}
rm -rf /tmp/f2prof
KW_BATCHES=10000 KW_SWEEP="$SW" timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/f2prof -- python tools/sweep_kw.py > gpurun_out/f2/$1.log 2>&1
python profiles/summarize_rocprof.py /tmp/f2prof 2>&1 | head -16 | cut -c1-170 > gpurun_out/f2/$1.txt
cat gpurun_out/f2/$1.txt; grep n_q gpurun_out/f2/$1.log | cut -c1-200
