cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_small; mkdir -p $O
TSGPU_HOST_TIMING=0 timeout 900 python tools/experiments/small_round_probe.py > $O/small_round_probe.txt 2> $O/small_round_probe.err; cat $O/small_round_probe.txt; tail -3 $O/small_round_probe.err
