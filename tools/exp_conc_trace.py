"""Summarise a rocprofv3 --kernel-trace CSV of the concurrency leg: per kernel and grid-size bucket the launch count and the
duration quantiles, plus per-queue idle gaps between consecutive kernels (experiment tooling; prints text)."""
import csv, glob, sys, collections
import numpy as np
paths = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for p in paths:
    for r in csv.DictReader(open(p)):
        rows.append((r["Kernel_Name"].split("(")[0][:60], int(r.get("Grid_Size") or r["Grid_Size_X"]) // max(1, int(r.get("Workgroup_Size") or r["Workgroup_Size_X"])), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0")))
b = collections.defaultdict(list)
for n, wg, s, e, q in rows:
    bucket = "<=256wg" if wg <= 256 else "<=4096wg" if wg <= 4096 else ">4096wg"
    b[(n, bucket)].append(((e - s) / 1e3, wg))
print(f"{'kernel':60s} {'bucket':9s} {'n':>7s} {'wg_p50':>7s} {'p50us':>8s} {'p90us':>8s} {'mean':>8s}")
for (n, bu), v in sorted(b.items()):
    d = np.array([x[0] for x in v]); w = np.array([x[1] for x in v])
    print(f"{n:60s} {bu:9s} {len(v):7d} {int(np.median(w)):7d} {np.median(d):8.1f} {np.percentile(d, 90):8.1f} {d.mean():8.1f}")
byq = collections.defaultdict(list)
for n, wg, s, e, q in rows: byq[q].append((s, e, n, wg))
for q, v in sorted(byq.items()):
    v.sort()
    gaps = np.array([(v[i + 1][0] - v[i][1]) / 1e3 for i in range(len(v) - 1)]) if len(v) > 1 else np.array([0.0])
    busy = sum(e - s for s, e, _, _ in v) / 1e3
    span = (v[-1][1] - v[0][0]) / 1e3
    print(f"queue {q}: {len(v)} kernels, busy {busy/1e3:.1f} ms of {span/1e3:.1f} ms, gap p50 {np.median(gaps):.1f} us p90 {np.percentile(gaps,90):.1f} us")
# sample chains: runs of kernels on one queue separated by < 60 us, that contain a kw_merge launch of <= 256 workgroups
shown = 0
for q, v in sorted(byq.items()):
    chain = []
    for i, (s, e, n, wg) in enumerate(v):
        if chain and s - chain[-1][1] > 60000:
            if any("kw_merge" in c[2] and c[3] <= 256 for c in chain) and shown < 12 and len(chain) <= 8:
                t0 = chain[0][0]
                print("chain q%s: " % q + " | ".join("%s[%d] +%.0f..%.0f" % (c[2].replace("void tsgpu::", "").replace("__amd_rocclr_", "")[:14], c[3], (c[0] - t0) / 1e3, (c[1] - t0) / 1e3) for c in chain))
                shown += 1
            chain = []
        chain.append((s, e, n, wg))
# the big batch: the last run of >= 20 consecutive launches (gaps < 200 us) on one queue that contains launches of > 4096 workgroups
import os
if os.environ.get("BIG"):
    for q, v in sorted(byq.items()):
        runs, cur = [], []
        for s, e, n, wg in v:
            if cur and s - cur[-1][1] > 200000: runs.append(cur); cur = []
            cur.append((s, e, n, wg))
        runs.append(cur)
        runs = [r for r in runs if sum(1 for c in r if c[3] > 4096 and "kw_find2" in c[2]) >= 4]
        if not runs: continue
        r = [c for c in runs[-1]]
        firsts = [i for i, c in enumerate(r) if "kw_find2" in c[2]]
        r = r[max(0, firsts[0] - 3):]
        t0 = r[0][0]
        print("big batch on queue %s: %d launches, span %.1f us, kernel time %.1f us" % (q, len(r), (r[-1][1] - t0) / 1e3, sum(c[1] - c[0] for c in r) / 1e3))
        prev = t0
        for c in r:
            print("  %-34s wg %6d  start +%8.1f  dur %7.1f  gap %5.1f" % (c[2].replace("void tsgpu::", "").replace("__amd_rocclr_", "")[:34], c[3], (c[0] - t0) / 1e3, (c[1] - c[0]) / 1e3, (c[0] - prev) / 1e3))
            prev = c[1]
        break
