"""Round 5: the pipelined multi-field find kernel with 2, 3 and 4 `query_by` fields against kw_search_mf_kernel (kw_mf_pipelined = 0): 4M documents,
2 000 three-term queries, device-resident hits; prints the batch wall time (best of 5) and the library's kernel timings. Not a bench line."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import typesense_amd as T
from typesense_amd import _lib as B, synth

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
SORT = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
pts = synth.points_column(n_docs)
qtok = synth.keyword_queries(2000, 3, 8, 2000, seed=91)
for nf in (2, 3, 4):
    g = T.GpuIndex(0)
    for f in range(nf):
        c = synth.zipf_corpus_csr(n_docs, 50_000, (20, 10, 6, 12)[f], seed=71 + f)
        g.field_create(f, False)
        g.terms_load_csr(f, c["term_ids"], c["ids_ptr"], c["ids"], c["offset_index"], c["off_ptr"], c["offsets"])
    g.column_set(0, pts)
    g.set_num_docs(n_docs)
    g.commit()
    fields = tuple((f, 15 - f) for f in range(nf))
    qs = [T.KwQuery(q, sort=SORT, topster_size=250, fields=fields) for q in qtok]
    outs = {}
    for pipelined in (1, 0):
        g.set_option("kw_mf_pipelined", pipelined)
        g.set_option("kw_timing_min_queries", 1 << 30)
        best = 1e9
        for it in range(6):
            t0 = time.perf_counter()
            h = g.keyword_search_batch(qs, k_stride=250)
            dt = (time.perf_counter() - t0) * 1e3
            if it:
                best = min(best, dt)
        outs[pipelined] = h
        g.set_option("kw_timing_min_queries", 1)
        g.keyword_search_batch(qs, k_stride=250)
        t = g.timings()
        g.set_option("kw_timing_min_queries", 1 << 30)
        tm = "find %.2f search(total kernels) %.2f merge %.2f ms" % (t.kw_find_ms, t.kw_search_ms, t.kw_merge_ms)
        print("fields %d pipelined %d: %.2f ms per 2000-query batch (%.0f K q/s) %s" % (nf, pipelined, best, 2000 / best, tm if tm else ""), flush=True)
    same = all(np.array_equal(getattr(outs[1], n), getattr(outs[0], n)) for n in ("keys", "scores", "n_hits", "num_matched"))
    print("fields %d: identical outputs %s, hits %d" % (nf, same, int(outs[1].n_hits.sum())), flush=True)
    g.close()
