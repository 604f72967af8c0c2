"""per-batch host phases of the 10 000-query keyword step (TSGPU_HOST_TIMING=1 prints them from inside the library): full collection and a 1/8 doc-range shard"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
torch.zeros(1, device="cuda")
import typesense_amd as T
from typesense_amd import _lib as B, synth
from bench import device_hits
n_docs, n_q = 10_000_000, 10_000
pts = synth.points_column(n_docs)
qtok = synth.keyword_queries(n_q, 3, 8, 2000, seed=4)
sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
arr = (B.KwQueryC * n_q)()
for i in range(n_q):
    T.KwQuery(qtok[i], sort=sort, topster_size=250).fill(arr[i])
for rng in (None, (0, n_docs // 8)):
    csr = synth.zipf_corpus_csr(n_docs, 100_000, 32, seed=2, doc_range=rng)
    g = T.GpuIndex(0); g.field_create(0, False)
    g.terms_load_csr(0, csr["term_ids"], csr["ids_ptr"], csr["ids"], csr["offset_index"], csr["off_ptr"], csr["offsets"])
    g.column_set(0, pts); g.set_num_docs(n_docs); g.commit()
    dev, hs = device_hits(torch, n_q, 250)
    for _ in range(3):
        g.keyword_search_batch_raw(arr, n_q, hs)
    sys.stderr.write("---- doc_range %s\n" % (rng,)); sys.stderr.flush()
    for _ in range(4):
        g.keyword_search_batch_raw(arr, n_q, hs)
    g.close()
