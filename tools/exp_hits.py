"""GPU experiment: distribution of matches per query / per work item in the headline keyword batch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import typesense_amd as T
from typesense_amd import _lib as B, synth
n_docs = 10_000_000
csr = synth.zipf_corpus_csr(n_docs, 100_000, 32, seed=2)
g = T.GpuIndex(0)
g.field_create(0, False)
g.terms_load_csr(0, csr["term_ids"], csr["ids_ptr"], csr["ids"], csr["offset_index"], csr["off_ptr"], csr["offsets"])
g.column_set(0, synth.points_column(n_docs)); g.set_num_docs(n_docs); g.commit()
sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
qtok = synth.keyword_queries(10000, 3, 8, 2000, seed=4)
hits = g.keyword_search_batch([T.KwQuery(q, sort=sort, topster_size=250) for q in qtok], k_stride=250)
nm = hits.num_matched.astype(np.int64)
print("total matches %d, mean %.0f, median %.0f, p90 %.0f, p99 %.0f, max %d; queries with >10K matches: %d holding %.0f%% of all matches" %
      (nm.sum(), nm.mean(), np.median(nm), np.percentile(nm, 90), np.percentile(nm, 99), nm.max(), (nm > 10000).sum(), 100.0 * nm[nm > 10000].sum() / nm.sum()))
tm = g.timings()
print("search %.3f ms (find %.3f), merge %.3f" % (tm.kw_search_ms, tm.kw_find_ms, tm.kw_merge_ms))
g.close()
