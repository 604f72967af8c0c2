"""rocprofv3 --pmc output (csv *_counter_collection.csv or rocpd .db) -> per-kernel average counter values.
Usage: python tools/pmc_summary.py gpurun_out/pmc_dir [kernel-regex]"""
import csv, glob, os, re, sqlite3, sys
from collections import defaultdict

d = sys.argv[1]
rx = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
acc = defaultdict(lambda: [0.0, 0, 0.0])
dur = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if rx and not rx.search(k):
            continue
        a = acc[(k[:60], r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1; a[2] = max(a[2], float(r["Counter_Value"]))
        if "Start_Timestamp" in r and r.get("End_Timestamp"):
            t = dur[k[:60]]
            t[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; t[1] += 1
for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
    cur = sqlite3.connect(db).cursor()
    try:
        tabs = [t[0] for t in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
        if "counters_collection" in tabs:
            cols = [c[1] for c in cur.execute("pragma table_info(counters_collection)").fetchall()]
            namecol = "kernel_name" if "kernel_name" in cols else "name"
            for k, cn, v in cur.execute("select %s, counter_name, value from counters_collection" % namecol):
                if rx and not rx.search(k):
                    continue
                a = acc[(k[:60], cn)]
                a[0] += float(v); a[1] += 1; a[2] = max(a[2], float(v))
    except Exception as e:
        print("# db %s: %s" % (db, e))
print("# per-dispatch averages from %s" % d)
for (k, c), (s, n, mx) in sorted(acc.items()):
    print("%-60s %-32s n=%-5d avg=%.6g max=%.6g" % (k, c, n, s / n, mx))
for k, (s, n) in sorted(dur.items()):
    print("%-60s %-32s n=%-5d avg_ms=%.4f" % (k, "duration(profiled)", n, s / n))
