"""Round 5: where a small keyword round's time goes (the micro-batcher's rounds hold ~50 queries at 256 callers). 10M documents, the bench's query
distribution; per batch size: wall per batch (hits to host memory), HIP-event kernel times from the library, and every query of the 49-query
batch alone (is the round as slow as its heaviest query?). Under rocprofv3 --kernel-trace the same run gives per-kernel durations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import typesense_amd as T
from typesense_amd import _lib as B, synth
n_docs = int(os.environ.get("N_DOCS", 10_000_000))
pts = synth.points_column(n_docs)
sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
csr = synth.zipf_corpus_csr(n_docs, 100_000, 32, seed=2)
g = T.GpuIndex(0); g.field_create(0, False)
g.terms_load_csr(0, csr["term_ids"], csr["ids_ptr"], csr["ids"], csr["offset_index"], csr["off_ptr"], csr["offsets"])
g.column_set(0, pts); g.set_num_docs(n_docs); g.commit()
qtok = synth.keyword_queries(10_000, 3, 8, 2000, seed=4)
def run(qs, reps=30):
    g.set_option("kw_timing_min_queries", 1 << 30)
    for _ in range(3):
        g.keyword_search_batch(qs, k_stride=250)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); g.keyword_search_batch(qs, k_stride=250); ts.append((time.perf_counter() - t0) * 1e6)
    g.set_option("kw_timing_min_queries", 1)
    g.keyword_search_batch(qs, k_stride=250)
    t = g.timings()
    return float(np.median(ts)), float(np.min(ts)), t
for n in (1, 4, 16, 49, 128, 256):
    for off in (0, 1000, 2000):
        qs = [T.KwQuery(q, sort=sort, topster_size=250) for q in qtok[off:off + n]]
        med, mn, t = run(qs)
        print("batch %3d (queries %d..): wall median %.0f us, min %.0f us | events: find %.0f score+find %.0f merge %.0f total %.0f us" %
              (n, off, med, mn, t.kw_find_ms * 1e3, t.kw_search_ms * 1e3, t.kw_merge_ms * 1e3, t.total_ms * 1e3), flush=True)
alone = []
for i in range(49):
    med, mn, t = run([T.KwQuery(qtok[i], sort=sort, topster_size=250)], reps=8)
    alone.append(med)
alone = np.array(alone)
print("the 49 queries alone: median %.0f us, p90 %.0f, max %.0f (query %d: ranks %s)" % (np.median(alone), np.percentile(alone, 90), alone.max(), int(alone.argmax()), qtok[int(alone.argmax())]), flush=True)
print("sorted:", " ".join("%.0f" % a for a in np.sort(alone)))
g.close()
