"""Multi-field keyword batch (query_by = 2 fields) on a 2M-doc collection: two-kernel form vs fused kernel."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import typesense_amd as T
from typesense_amd import _lib as B, synth
import bench

n_docs = 2_000_000
g = T.GpuIndex(0, os.environ.get('TSGPU_LIB') or None)
for f, seed in ((0, 2), (1, 3)):
    csr = synth.zipf_corpus_csr(n_docs, 100_000, 16, seed=seed)
    g.field_create(f, False)
    g.terms_load_csr(f, csr["term_ids"], csr["ids_ptr"], csr["ids"], csr["offset_index"], csr["off_ptr"], csr["offsets"])
g.column_set(0, synth.points_column(n_docs)); g.set_num_docs(n_docs); g.commit()
sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
for n_q in (2000, 200):
    qtok = synth.keyword_queries(n_q, 3, 8, 2000, seed=4)
    arr = (B.KwQueryC * n_q)()
    for i in range(n_q):
        T.KwQuery(qtok[i], fields=[(0, 3), (1, 1)], sort=sort, topster_size=250).fill(arr[i])
    dev, hs = bench.device_hits(torch, n_q, 250)
    for two in (1, 0):
        g.set_option("kw_two_kernels", two)
        for _ in range(2):
            g.keyword_search_batch_raw(arr, n_q, hs)
        t0 = time.perf_counter(); ks = []
        for _ in range(5):
            g.keyword_search_batch_raw(arr, n_q, hs)
            ks.append(g.timings().kw_search_ms)
        wall = (time.perf_counter() - t0) / 5
        print(json.dumps(dict(n_q=n_q, two_kernels=two, wall_ms=wall * 1e3, qps=n_q / wall, search_ms=float(np.mean(ks)), hits=int(dev["n_hits"].sum().item()))), flush=True)
g.close()
