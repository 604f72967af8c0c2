import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import typesense_amd as T
from typesense_amd import _lib as B
n = 10_000_000
g = T.GpuIndex(0)
g.set_num_docs(n)
pts = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) % np.uint64(1000)).astype(np.int64)
g.column_set(0, pts)
g.field_create(0, False); g.commit()
for name, col in (("one group", np.full(n, 12345, np.uint64)), ("two groups 99/1", np.where(np.arange(n) % 100 == 0, 7, 12345).astype(np.uint64)), ("1000 groups", (np.arange(n, dtype=np.uint64) % np.uint64(1000)) + np.uint64(5)),
                  ("10 groups", (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) % np.uint64(10)) + np.uint64(5)), ("30 groups", (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) % np.uint64(30)) + np.uint64(5)),
                  ("100 groups", (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) % np.uint64(100)) + np.uint64(5))):
    if len(sys.argv) > 1 and sys.argv[1] not in name:
        continue
    g.column_set(1, col.view(np.int64))
    q = T.KwQuery([], sort=((B.SORT_INT64_COLUMN, 1, 0), (B.SORT_SEQ_ID, 1, 0)), topster_size=250)
    for fp in (1, 0):
        for L in (3, 50):
            best = 1e9
            for _ in range(2):
                t0 = time.time()
                h, gh = g.keyword_search_grouped_batch([q], [(L, 1, fp, 0, 1)], k_stride=250 * L, g_stride=250)
                best = min(best, time.time() - t0)
            print(name, "first" if fp else "second", "limit", L, "ms %.2f" % (best * 1e3), "groups", int(gh.n_groups[0]), "hits", int(h.n_hits[0]), flush=True)
