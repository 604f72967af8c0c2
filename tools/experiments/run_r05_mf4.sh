# round 5: kw_find_mf2_kernel<., 4> (three and four query_by fields) on the GPU: the 2/3/4-field probe, plain and under a kernel trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05_mf4; mkdir -p $O
timeout 900 python tools/experiments/mf_fields_probe.py > $O/mf_fields_probe.txt 2>&1; cat $O/mf_fields_probe.txt
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -- python $GRAFT_REPO_ROOT/tools/experiments/mf_fields_probe.py > $GRAFT_REPO_ROOT/$O/trace.log 2>&1 )
python profiles/summarize_rocprof.py $O/trace > $O/rocprof_mf_fields_stats.txt 2>&1; head -14 $O/rocprof_mf_fields_stats.txt
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete; rm -rf $O/trace
