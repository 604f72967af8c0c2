import sys, os, json, time
sys.path.insert(0, "/root/repo")
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
for G in (2, 8):
    members = []
    for i in range(G):
        lo, hi = i * (n_docs // G), (i + 1) * (n_docs // G)
        csr = synth.zipf_corpus_csr(n_docs, 100_000, 32, seed=2, doc_range=(lo, hi))
        g = T.GpuIndex(0); g.field_create(0, False)
        g.terms_load_csr(0, csr["term_ids"], csr["ids_ptr"], csr["ids"], csr["offset_index"], csr["off_ptr"], csr["offsets"])
        g.column_set(0, pts); g.set_num_docs(n_docs); g.commit(); members.append(g); del csr
    grp = T.GpuGroup(members, B.XCHG_COPY)
    gdev, ghs = device_hits(torch, n_q, 100)
    for pruned in (1, 0):
        grp.set_option("kw_exchange_pruned", pruned)
        for _ in range(3):
            grp.keyword_search_batch_raw(arr, n_q, 100, ghs)
        sys.stderr.write("---- G=%d pruned=%d\n" % (G, pruned)); sys.stderr.flush()
        os.environ["TSGPU_GROUP_DEBUG"] = "1"
        grp.keyword_search_batch_raw(arr, n_q, 100, ghs)
        del os.environ["TSGPU_GROUP_DEBUG"]
    grp.close()
    for m in members: m.close()
