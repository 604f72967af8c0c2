"""GPU experiment (not part of the product): 1-query callers on the 10M-doc keyword collection — threads x lanes x window sweep with the
host-phase counters. Usage: python tools/exp_concurrency.py [n_docs]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import typesense_amd as T  # noqa: E402
from typesense_amd import _lib as B, synth  # noqa: E402

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
import __graft_entry__
__graft_entry__.build()
g = T.GpuIndex(0)
csr = synth.zipf_corpus_csr(n_docs, 100_000, 32, seed=2)
g.field_create(0, False)
g.terms_load_csr(0, csr["term_ids"], csr["ids_ptr"], csr["ids"], csr["offset_index"], csr["off_ptr"], csr["offsets"])
g.column_set(0, synth.points_column(n_docs))
g.set_num_docs(n_docs)
g.commit()
n_q = 10_000
lo, hi = [int(x) for x in os.environ.get("RANKS", "8,2000").split(",")]
qtok = synth.keyword_queries(n_q, 3, lo, hi, seed=4)
sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
arr = (B.KwQueryC * n_q)()
for i in range(n_q):
    T.KwQuery(qtok[i], sort=sort, topster_size=250).fill(arr[i])
LG = bench.loadgen_lib()
fn = C.cast(g.L.tsgpu_keyword_search_batch, C.c_void_p)
names = ["kw_batches", "kw_plan_us", "kw_upload_us", "kw_launch_us", "kw_wait_us", "kw_book_us", "batch_exec_us", "batch_scatter_us", "batch_rounds", "batch_coalesced_calls"]


def run(threads, calls, qpc=1):
    lat = np.zeros(threads * calls)
    got = np.zeros(n_q, np.uint64)
    fails = C.c_uint64(0)
    LG.tsgpu_loadgen_keyword(fn, g.h, C.cast(arr, C.c_void_p), n_q, 250, 100, threads, max(2, calls // 8), qpc, lat.ctypes.data, got.ctypes.data, C.byref(fails))
    def cpu_stat():
        try:
            d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
            return int(d["usage_usec"]), int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
        except Exception:
            return 0, 0, 0
    def thread_cpu():        # clock ticks (utime + stime) of the threads that exist NOW (runtime helpers, host pools: the load generator's threads come and go)
        out = {}
        for tid in os.listdir("/proc/self/task"):
            try:
                f = open("/proc/self/task/%s/stat" % tid).read()
                comm = f[f.index("(") + 1:f.rindex(")")]
                rest = f[f.rindex(")") + 2:].split()
                out[tid] = (comm, int(rest[11]) + int(rest[12]))
            except Exception:
                pass
        return out
    def proc_cpu():
        rest = open("/proc/self/stat").read()
        rest = rest[rest.rindex(")") + 2:].split()
        return int(rest[11]) + int(rest[12])
    prof = None
    if os.environ.get("PROF"):
        import subprocess
        so = "/tmp/libsigprof.so"
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(os.path.dirname(os.path.abspath(__file__)), "sigprof", "sigprof.cpp"), "-ldl"])
        prof = C.CDLL(so)
        prof.sigprof_start(2000)
    th0, pc0 = thread_cpu(), proc_cpu()
    cs0 = cpu_stat()
    c0 = {n: g.counter(n) for n in names}
    [g.counter(n) for n in ("kw_max_plan_us", "kw_max_upload_us", "kw_max_launch_us", "kw_max_wait_us", "kw_max_queue_us", "kw_max_wake_us")]
    wall = LG.tsgpu_loadgen_keyword(fn, g.h, C.cast(arr, C.c_void_p), n_q, 250, 100, threads, calls, qpc, lat.ctypes.data, got.ctypes.data, C.byref(fails))
    if prof:
        out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "sigprof_%d_%d.txt" % (threads, qpc))
        n = prof.sigprof_stop(out.encode(), 2, 8)
        print("   sigprof: %d samples -> %s" % (n, out), flush=True)
    c = {n: g.counter(n) - c0[n] for n in names}
    cs1 = cpu_stat()
    th1, pc1 = thread_cpu(), proc_cpu()
    tick = os.sysconf("SC_CLK_TCK")
    helpers = {}
    for tid, (comm, t1) in th1.items():
        if tid in th0:
            helpers[comm] = helpers.get(comm, 0) + (t1 - th0[tid][1])
    hsum = sum(helpers.values())
    print("   process CPU %.0f ms over %.0f ms wall; threads that outlive the run (runtime helpers etc.): %.0f ms %s; request threads: %.0f ms = %.1f us per call"
          % ((pc1 - pc0) * 1e3 / tick, wall * 1e3, hsum * 1e3 / tick, {k: round(v * 1e3 / tick) for k, v in helpers.items() if v}, (pc1 - pc0 - hsum) * 1e3 / tick,
             (pc1 - pc0 - hsum) * 1e6 / tick / (threads * calls * qpc)), flush=True)
    print("   cgroup: %.1f CPU-us per call (%.1f CPUs busy), throttled %d times for %.1f ms (cpu.max: %s)" % ((cs1[0] - cs0[0]) / (threads * calls * qpc), (cs1[0] - cs0[0]) / (wall * 1e6), cs1[1] - cs0[1], (cs1[2] - cs0[2]) / 1e3, open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?"), flush=True)
    nb = max(c["kw_batches"], 1)
    print("   slowest batch phase: plan %d upload %d launch %d wait %d us; longest parked->round start %d us, results ready->caller resumes %d us" % tuple(g.counter(n) for n in ("kw_max_plan_us", "kw_max_upload_us", "kw_max_launch_us", "kw_max_wait_us", "kw_max_queue_us", "kw_max_wake_us")), flush=True)
    print("   latency us: mean %.0f p10 %.0f p25 %.0f p50 %.0f p75 %.0f p90 %.0f p99 %.0f max %.0f | wall %.1f ms for %d calls per thread -> %.0f us per call per thread" % (lat.mean(), *[np.percentile(lat, x) for x in (10, 25, 50, 75, 90, 99)], lat.max(), wall * 1e3, calls, wall * 1e6 / calls), flush=True)
    return dict(qps=threads * calls * qpc / wall, p50=float(np.percentile(lat, 50)), p99=float(np.percentile(lat, 99)), fails=fails.value,
                q_per_batch=threads * calls * qpc / nb, plan=c["kw_plan_us"] / nb, upload=c["kw_upload_us"] / nb, launch=c["kw_launch_us"] / nb,
                wait=c["kw_wait_us"] / nb, book=c["kw_book_us"] / nb, exec_round=c["batch_exec_us"] / max(c["batch_rounds"], 1),
                scatter=c["batch_scatter_us"] / max(c["batch_rounds"], 1))


# direct batches of several sizes (single caller): the fixed cost of a batch
hits = T.Hits(n_q, 250)
hs = hits.c_struct()
for nb in (64, 128):
    sub = (B.KwQueryC * nb).from_address(C.addressof(arr))
    g.keyword_search_batch_raw(sub, nb, hs)
    c0 = {n: g.counter(n) for n in names}
    t0 = time.perf_counter()
    for _ in range(20):
        g.keyword_search_batch_raw(sub, nb, hs)
    dt = (time.perf_counter() - t0) / 20
    c = {n: (g.counter(n) - c0[n]) / 20 for n in names}
    print("direct batch %5d: %.0f us/call  plan %.0f upload %.0f launch %.0f wait %.0f book %.0f | gpu search %.3f merge %.3f ms" %
          (nb, dt * 1e6, c["kw_plan_us"], c["kw_upload_us"], c["kw_launch_us"], c["kw_wait_us"], c["kw_book_us"], g.timings().kw_search_ms, g.timings().kw_merge_ms), flush=True)

SWEEP = os.environ.get("SWEEP", "lanes")
if SWEEP == "capacity":   # what the lanes + the GPU deliver for rounds of a fixed size, without the combiner: T threads, qpc-query calls, no coalescing
    g.set_option("batch_max_queries", 0)
    for lanes in (4, 8, 16):
        g.set_option("kw_lanes", lanes)
        for qpc in (16, 32, 64, 128):
            for chunk in (0, 32):
                g.set_option("kw_chunk_blocks", chunk)
                r = run(lanes, max(16, 40000 // (lanes * qpc)), qpc)
                print("capacity lanes=threads %2d qpc %3d chunk %2d: %8.0f q/s p50 %6.0f us | plan %.0f upload %.0f launch %.0f wait %.0f" % (lanes, qpc, chunk, r["qps"], r["p50"], r["plan"], r["upload"], r["launch"], r["wait"]), flush=True)
    g.close()
    sys.exit(0)
if SWEEP == "soak":        # many short runs over every thread count: a rare race in the lock-free combiner would show as a hang or a failure
    import random
    random.seed(1)
    bad = 0
    for rep in range(int(os.environ.get("REPS", "12"))):
        for threads in (2, 3, 5, 8, 16, 47, 49, 64, 128, 256):
            g.set_option("batch_window_us", random.choice([0, 10, 10, 50]))
            g.set_option("batch_round_queries", random.choice([2, 7, 64, 1024]))
            g.set_option("kw_lanes", random.choice([1, 2, 4, 4, 8]))
            lat = np.zeros(threads * 40)
            got = np.zeros(n_q, np.uint64)
            fails = C.c_uint64(0)
            LG.tsgpu_loadgen_keyword(fn, g.h, C.cast(arr, C.c_void_p), n_q, 250, 100, threads, 40, random.choice([1, 1, 1, 2, 5]), lat.ctypes.data, got.ctypes.data, C.byref(fails))
            bad += int(fails.value)
        print("soak rep %d done, failures so far %d" % (rep, bad), flush=True)
    print("SOAK", "OK" if bad == 0 else "FAILED", flush=True)
    g.close()
    sys.exit(0)
if SWEEP == "default":     # the shipped options, several runs: where does the latency tail come from?
    for rep in range(1 if os.environ.get("PROF") else 2):
        for threads in (256, 16):
            r = run(threads, max(16, 40000 // threads) * (60 if os.environ.get("PROF") else 1))
            print("default threads %3d: %8.0f q/s p50 %6.0f p99 %6.0f us | %.1f q/batch plan %.0f upload %.0f launch %.0f wait %.0f | round exec %.0f scatter %.0f us fails %d"
                  % (threads, r["qps"], r["p50"], r["p99"], r["q_per_batch"], r["plan"], r["upload"], r["launch"], r["wait"], r["exec_round"], r["scatter"], r["fails"]), flush=True)
    g.close()
    sys.exit(0)
if SWEEP == "chunk":      # driver blocks per work item under 256 callers (auto = 8 for rounds of <= 128 queries: the single-call latency optimum)
    combos = [(4, 128, 80, 48, ch) for ch in (0, 16, 32, 64, 0)]
else:
    combos = [(l, m, w, b, 0) for l, m, w, b in ((4, 128, 80, 48), (4, 128, 80, 100000), (8, 64, 80, 100000), (8, 64, 40, 48), (6, 96, 60, 100000), (8, 128, 80, 100000))]
for lanes, bmax, window, blk, chunk in combos:
    g.set_option("kw_chunk_blocks", chunk)
    g.set_option("blocking_sync_min_callers", blk)
    g.set_option("kw_lanes", lanes)
    g.set_option("batch_max_queries", bmax)
    g.set_option("batch_window_us", window)
    for threads in [int(x) for x in os.environ.get("THREADS", "256").split(",")]:
        r = run(threads, max(16, 30000 // threads))
        print("chunk %d blk %d lanes %d max %3d window %3d threads %3d: %8.0f q/s p50 %6.0f p99 %6.0f us | %.1f q/batch plan %.0f upload %.0f launch %.0f wait %.0f book %.0f | round exec %.0f scatter %.0f us fails %d"
              % (chunk, blk, lanes, bmax, window, threads, r["qps"], r["p50"], r["p99"], r["q_per_batch"], r["plan"], r["upload"], r["launch"], r["wait"], r["book"], r["exec_round"], r["scatter"], r["fails"]), flush=True)
g.close()
