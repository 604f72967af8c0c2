"""facet counting under skew (GPU probe): 10M result ids, a facet field with V distinct values (one hash per document)"""
import sys, time, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import typesense_amd as T
n = 10_000_000
g = T.GpuIndex(0)
g.set_num_docs(n)
ids = np.arange(n, dtype=np.uint32)
for V in (2, 10, 30, 100, 1000, 1_000_000, -1):
    if V > 0:
        hashes = ((np.arange(n, dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(V)).astype(np.uint32) + np.uint32(17)
    else:                                # Zipf(1.2) over 100 000 values: a heavy head and a long tail in every wave
        hashes = (np.random.default_rng(1).zipf(1.2, size=n) % 100_000).astype(np.uint32) * np.uint32(2654435761)
    ptr = np.arange(n + 1, dtype=np.uint64)
    g.facet_set(3, ptr, hashes)
    best = 1e9
    for _ in range(3):
        t0 = time.time()
        out = g.facet_count_batch(3, [ids], cap=2048)
        best = min(best, time.time() - t0)
    print("V", V, "ms %.2f" % (best * 1e3), flush=True)

# the grouped and the range forms of the walk (10M ids; 100 values; 50 000 groups; 10 ranges over the value column)
hashes = ((np.arange(n, dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(100)).astype(np.uint32) + np.uint32(17)
g.facet_set(3, np.arange(n + 1, dtype=np.uint64), hashes)
g.column_set(1, ((np.arange(n, dtype=np.uint64) * np.uint64(40503)) % np.uint64(50_000)).view(np.int64))
g.column_set(2, ((np.arange(n, dtype=np.uint64) * np.uint64(7919)) % np.uint64(10_000)).view(np.int64))
ranges = [(1000 * (r + 1), 1000 * r) for r in range(10)]
for name, fn in (("grouped 50K groups", lambda: g.facet_count_batch(3, [ids], cap=2048, group_column=1)),
                 ("range x10", lambda: g.facet_range_count_batch(3, 2, ranges, [ids])),
                 ("range x10 grouped", lambda: g.facet_range_count_batch(3, 2, ranges, [ids], group_column=1))):
    best = 1e9
    for _ in range(3):
        t0 = time.time()
        out = fn()
        best = min(best, time.time() - t0)
    print(name, "ms %.2f" % (best * 1e3), flush=True)
