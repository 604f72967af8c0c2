# kernel timeline of the sliced host delivery (10 000-query keyword batch): rocprofv3 --kernel-trace (rocpd db), last sliced call printed
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/f2
export TMPDIR=/tmp
rm -rf /tmp/hd_trace
KW_HOST=1 KW_BATCHES=10000 KW_SWEEP='[{"kw_pair_blocks":1}]' timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/hd_trace -- python tools/sweep_kw.py > gpurun_out/f2/hd_trace.log 2>&1
python - <<'PY' > gpurun_out/f2/hd_trace.txt
import glob, sqlite3
db = glob.glob("/tmp/hd_trace/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select start, end, name, grid_x from kernels where name like '%tsgpu%' order by start").fetchall()
finds = [i for i, r in enumerate(rows) if "kw_find2" in r[2]]
# the sliced calls launch three find kernels each: take the last 4 finds (two calls of two slices) before the final unsliced call
sel = rows[finds[-4]:] if len(finds) >= 4 else rows
t0 = sel[0][0]
for s, e, n, gx in sel:
    print("%9.3f -> %9.3f  %7.3f ms  grid %8d  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, gx, n.split("(")[0][-44:]))
PY
tail -60 gpurun_out/f2/hd_trace.txt
