cd /tmp && export TMPDIR=/tmp
for sm in 17 2; do
rm -rf /tmp/ct
rocprofv3 --kernel-trace --output-format csv -d /tmp/ct -- python $GRAFT_REPO_ROOT/bench.py --workload keyword --no-cpu-baseline --steps 3 --warmup 1 --opt batch_window_us=10 --opt kw_merge_select_min=$sm > /tmp/ct.json 2>/tmp/ct.err
echo "== select_min $sm"
python $GRAFT_REPO_ROOT/tools/exp_conc_trace.py /tmp/ct 2>&1 | grep "tsgpu::kw_merge"
done
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_keyword.py -x -q 2>&1 | tail -2
