"""GPU experiment: the 10 000-query keyword batch with results delivered to pageable host memory, under slicing options
(kw_host_split_queries = smallest slice, kw_host_split_first_pct = the first slice's share). Usage: python tools/exp_host_delivery.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import typesense_amd as T  # noqa: E402
from typesense_amd import _lib as B, synth  # noqa: E402
import __graft_entry__  # noqa: E402

__graft_entry__.build()
n_docs = 10_000_000
g = T.GpuIndex(0)
csr = synth.zipf_corpus_csr(n_docs, 100_000, 32, seed=2)
g.field_create(0, False)
g.terms_load_csr(0, csr["term_ids"], csr["ids_ptr"], csr["ids"], csr["offset_index"], csr["off_ptr"], csr["offsets"])
g.column_set(0, synth.points_column(n_docs))
g.set_num_docs(n_docs)
g.commit()
n_q = 10_000
qtok = synth.keyword_queries(n_q, 3, 8, 2000, seed=4)
sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
arr = (B.KwQueryC * n_q)()
for i in range(n_q):
    T.KwQuery(qtok[i], sort=sort, topster_size=250).fill(arr[i])
hh = T.Hits(n_q, 250)
hs = hh.c_struct(seam_arrays_only=True)
ref = None
for cfg in [(0, 50)] + [tuple(int(x) for x in c.split(":")) for c in os.environ.get("CFGS", "1000:75,1000:72,1000:70,1000:78,1000:75").split(",")] + [(0, 50)]:
    g.set_option("kw_host_split_queries", cfg[0])
    g.set_option("kw_host_split_first_pct", cfg[1])
    g.set_option("kw_host_split_device_plan", cfg[2] if len(cfg) > 2 else 1)
    g.keyword_search_batch_raw(arr, n_q, hs)
    ts = []
    for _ in range(12):
        t0 = time.perf_counter()
        g.keyword_search_batch_raw(arr, n_q, hs)
        ts.append(time.perf_counter() - t0)
    chk = (int(hh.n_hits.sum()), int(hh.num_matched.sum()), int(hh.keys[:, 0][hh.n_hits > 0].sum()), int(hh.scores[:, 0, 1][hh.n_hits > 0].sum()))
    ref = ref or chk
    ts = np.array(ts) * 1e3
    print("min slice %5d first %2d%% dp %s: median %.2f ms (min %.2f max %.2f) -> %.0f q/s  same results: %s" % (cfg[0], cfg[1], cfg[2] if len(cfg) > 2 else 1, np.median(ts), ts.min(), ts.max(), n_q / np.median(ts) * 1e3, chk == ref), flush=True)
g.close()
