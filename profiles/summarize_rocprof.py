"""rocprofv3 (rocpd .db or *_kernel_stats.csv) -> a small text table of per-kernel time. Usage:
   python profiles/summarize_rocprof.py gpurun_out/prof_kw > profiles/r01/rocprof_keyword_stats.txt"""
import glob, os, sqlite3, sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""            # optional: only kernels whose name contains this
dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
print("# rocprofv3 --kernel-trace --stats summary of %s" % d)
for db in dbs:
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e6, min(end-start)/1e6, max(end-start)/1e6, "
                       "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
                       "from kernels where name like ? group by name order by 3 desc limit 25", ("%" + flt + "%",)).fetchall()
    tot = sum(r[2] for r in rows)
    print("%-72s %6s %10s %9s %9s %9s %5s %5s %5s %7s %7s %10s %5s" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "vgpr", "agpr", "sgpr", "lds", "scratch", "grid_x", "wg_x"))
    for r in rows:
        print("%-72s %6d %10.3f %9.3f %9.3f %9.3f %5s %5s %5s %7s %7s %10s %5s" % ((r[0][:72],) + tuple(r[1:])))
