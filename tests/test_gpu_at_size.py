"""`-m gpu`: parity AT SIZE inside the test tier (not only inside bench.py), on ONE 10M-document collection shared by the module:
* BASELINE config 2 (seed 2, V = 100 000, 32 tokens per document) — 512 three-term queries, Topster content +
  num_keyword_matches bit-exact vs the oracle;
* BASELINE config 3 — 10M x 768 N(0,1) rows (seed 3), B = 256, k = 100, `ip` AND `cosine`: labels, order and distance BITS of 64 queries vs
  the oracle's exact flat scan of ALL rows, streamed in 2^20-row slabs (process_results_bruteforce, /root/reference/src/index.cpp:3345-3374;
  vector branch :3645-3732);
* BASELINE config 4 — configs 2 + 3 on the same 10M ids: the fused Topster (key, 3 score words, text_match, vector_distance bits) of 32
  queries vs oracle.search_hybrid, with and without rerank_hybrid_matches (/root/reference/src/index.cpp:4036-4221, 8793-8923);
and, at 2M documents, hybrid search with filter / excluded ids and cosine over rows of uneven norms."""
import numpy as np
import pytest

import typesense_amd as T
from typesense_amd import _lib as B, synth
from oracle import oracle_py as O
import helpers as H

pytestmark = pytest.mark.gpu

N10M, DIM, K_VEC, VEC_BATCH, N_VEC_CHECK, N_HYB_CHECK = 10_000_000, 768, 100, 256, 64, 32

SORT = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
OSORT = ((O.SORT_TEXT_MATCH, 0, 1), (O.SORT_INT64_COLUMN, 0, 1))


def _load(n_docs, vocab, tpd, seed):
    csr = synth.zipf_corpus_csr(n_docs, vocab, tpd, seed=seed)
    pts = synth.points_column(n_docs)
    g = T.GpuIndex(0)
    g.field_create(0, False)
    g.terms_load_csr(0, csr["term_ids"], csr["ids_ptr"], csr["ids"], csr["offset_index"], csr["off_ptr"], csr["offsets"])
    g.column_set(0, pts)
    g.set_num_docs(n_docs)
    g.commit()
    orc = O.OracleIndex(1, 1)
    orc.set_num_docs(n_docs)
    orc.set_sort_dense(0, pts)
    return csr, g, orc


def _need(orc, csr, loaded, terms):
    for t in np.unique(np.asarray(terms).ravel()):
        t = int(t)
        if t in loaded:
            continue
        ids, oi, off = synth.csr_term(csr, t)
        if ids.size:
            orc.load_posting(0, t, ids, oi, off)
        loaded.add(t)


class _World:
    pass


@pytest.fixture(scope="module")
def corpus():
    """BASELINE config 2's collection, built once for the module (keyword index + `points` column + a keyword-only oracle)"""
    w = _World()
    w.n_docs = N10M
    w.csr, w.g, w.orc = _load(N10M, 100_000, 32, seed=2)
    w.pts = synth.points_column(N10M)
    w.loaded = set()
    yield w
    w.g.close()


@pytest.fixture(scope="module")
def vectors(corpus):
    """config 3 on the same ids: field 1 = `ip`, field 2 = `cosine` (rows normalised on insert), and the ORACLE's exact top-100 of the first
    64 queries of the bench's query stream (seed 4) over all 10M rows for both metrics — one host copy of every slab serves both"""
    g = corpus.g
    H.load_vector_field(g, 1, B.METRIC_IP, N10M, DIM)
    H.load_vector_field(g, 2, B.METRIC_COSINE, N10M, DIM)
    w = _World()
    w.Q = synth.random_vectors(VEC_BATCH, DIM, seed=4, device="cuda").cpu().numpy()
    Qc = np.ascontiguousarray(w.Q[:N_VEC_CHECK])
    w.exact = H.exact_knn_chunked(N10M, DIM, {O.METRIC_IP: Qc, O.METRIC_COSINE: Qc}, K_VEC)
    return w


def test_config2_10m_docs_512_queries_topster_and_counts_equal_the_oracle(corpus):
    n_docs, n_q = N10M, 512
    csr, g, orc = corpus.csr, corpus.g, corpus.orc
    qtok = synth.keyword_queries(n_q, 3, 8, 2000, seed=4)                     # the bench's own query stream (ranks log-uniform in [8, 2000])
    hits = g.keyword_search_batch([T.KwQuery(q, sort=SORT, topster_size=250) for q in qtok], k_stride=250)
    assert (hits.status == 0).all()
    loaded = corpus.loaded
    bad, with_hits = [], 0
    for i in range(n_q):
        _need(orc, csr, loaded, qtok[i])
        ref = orc.search_keyword(orc.make_query(qtok[i], sort=OSORT, fetch_size=100))
        n = int(hits.n_hits[i])
        with_hits += n > 0
        if n != ref.keys.size or not np.array_equal(hits.keys[i, :n], ref.keys) or not np.array_equal(hits.scores[i, :n], ref.scores) \
                or not np.array_equal(hits.text_match[i, :n], ref.text_match) or int(hits.num_matched[i]) != int(ref.num_keyword_matches):
            bad.append(i)
    assert not bad, "queries that differ from the oracle at 10M docs: %s" % bad[:20]
    assert with_hits > n_q // 2
    # idempotence + batch-size independence at size: the first 64 queries alone give the same lists
    h2 = g.keyword_search_batch([T.KwQuery(q, sort=SORT, topster_size=250) for q in qtok[:64]], k_stride=250)
    for i in range(64):
        n = int(hits.n_hits[i])
        assert int(h2.n_hits[i]) == n and np.array_equal(h2.keys[i, :n], hits.keys[i, :n]) and np.array_equal(h2.scores[i, :n], hits.scores[i, :n])
    # the host-output batch served in chained slices, enqueued in order, the first planned on the device (kw_split_host: 435 + 77 queries by default,
    # 435 + 38 + 39 with two tail slices) gives the same lists
    split0, plan0, tail0 = 1000, 512, 1                                        # the defaults (csrc/tsgpu_host.h), restored below: the module shares this index
    g.set_option("kw_host_split_queries", 32)
    g.set_option("kw_device_plan_min_queries", 256)
    b0 = g.counter("kw_batches")
    h3 = g.keyword_search_batch([T.KwQuery(q, sort=SORT, topster_size=250) for q in qtok], k_stride=250)
    assert g.counter("kw_batches") - b0 == 2 and (h3.status == 0).all()
    g.set_option("kw_host_split_tail_slices", 2)
    b0 = g.counter("kw_batches")
    h4 = g.keyword_search_batch([T.KwQuery(q, sort=SORT, topster_size=250) for q in qtok], k_stride=250)
    assert g.counter("kw_batches") - b0 == 3 and (h4.status == 0).all()
    assert np.array_equal(h4.n_hits, h3.n_hits) and np.array_equal(h4.num_matched, h3.num_matched)
    for i in range(n_q):
        n = int(h3.n_hits[i])
        assert np.array_equal(h4.keys[i, :n], h3.keys[i, :n]) and np.array_equal(h4.scores[i, :n], h3.scores[i, :n])
    assert np.array_equal(h3.n_hits, hits.n_hits) and np.array_equal(h3.num_matched, hits.num_matched)
    for i in range(n_q):
        n = int(hits.n_hits[i])
        assert np.array_equal(h3.keys[i, :n], hits.keys[i, :n]) and np.array_equal(h3.scores[i, :n], hits.scores[i, :n]) and np.array_equal(h3.text_match[i, :n], hits.text_match[i, :n])
    for opt, dflt in (("kw_host_split_queries", split0), ("kw_device_plan_min_queries", plan0), ("kw_host_split_tail_slices", tail0)):
        g.set_option(opt, dflt)


def _knn_equals_exact(g, field, Q, exact):
    ed, el, _ = exact
    dist, lab, cnt = g.vec_knn_batch(field, Q, K_VEC)                          # the whole B = 256 batch of config 3
    assert (cnt == K_VEC).all()
    bad = [i for i in range(ed.shape[0]) if not (np.array_equal(lab[i].astype(np.uint32), el[i]) and np.array_equal(dist[i].view(np.uint32), ed[i].view(np.uint32)))]
    assert not bad, "queries whose top-%d (labels in order + distance bits) differ from the oracle's exact scan of all rows: %s" % (K_VEC, bad[:16])
    return dist, lab


def test_config3_10m_x_768_ip_top100_of_64_queries_equals_the_oracle_flat_scan(corpus, vectors):
    d, l = _knn_equals_exact(corpus.g, 1, vectors.Q, vectors.exact[O.METRIC_IP])
    # size-independent properties over the whole batch: ascending distances, ties -> smaller label, no duplicates; a smaller batch returns the same rows
    assert (np.diff(d, axis=1) >= 0).all()
    ties = np.diff(d, axis=1) == 0
    assert (np.diff(l.astype(np.int64), axis=1)[ties] > 0).all()
    assert all(np.unique(l[i]).size == K_VEC for i in range(l.shape[0]))
    d16, l16, _ = corpus.g.vec_knn_batch(1, vectors.Q[:16], K_VEC)
    assert np.array_equal(l16, l[:16]) and np.array_equal(d16.view(np.uint32), d[:16].view(np.uint32))


def test_config3_10m_x_768_cosine_top100_of_64_queries_equals_the_oracle_flat_scan(corpus, vectors):
    d, _ = _knn_equals_exact(corpus.g, 2, vectors.Q, vectors.exact[O.METRIC_COSINE])
    assert (np.diff(d, axis=1) >= 0).all()


@pytest.mark.parametrize("rerank", [False, True])
def test_config4_10m_hybrid_fused_topster_of_32_queries_equals_the_oracle(corpus, vectors, rerank):
    g, n_q = corpus.g, N_HYB_CHECK
    qtok = synth.keyword_queries(VEC_BATCH, 3, 8, 2000, seed=5)                 # the bench's hybrid query stream
    qs = [T.KwQuery(qtok[i], sort=SORT, topster_size=250) for i in range(VEC_BATCH)]
    hits = g.hybrid_search_batch(qs, 1, vectors.Q, k=K_VEC, fetch_size=100, alpha=0.3, k_stride=250, rerank=rerank)
    assert (hits.status == 0).all()
    _, _, rows = vectors.exact[O.METRIC_IP]
    orc, labs = H.oracle_for_hybrid_at_size(N10M, DIM, corpus.csr, corpus.pts, qtok[:n_q], rows)
    if rerank:      # compute_aux_scores asks for the distance of keyword-only hits by label: the oracle needs those rows too
        H.oracle_add_rows(orc, N10M, DIM, labs, np.concatenate([hits.keys[i, :int(hits.n_hits[i])] for i in range(n_q)]))
    bad, fused = [], 0
    for i in range(n_q):
        ref = orc.search_hybrid(orc.make_query(qtok[i], sort=OSORT, fetch_size=100, topster_size=250), vectors.Q[i], k=K_VEC, alpha=0.3, cap=1024, rerank=rerank)
        n = int(hits.n_hits[i])
        fused += n
        if n != ref.keys.size or not np.array_equal(hits.keys[i, :n], ref.keys) or not np.array_equal(hits.scores[i, :n], ref.scores) \
                or not np.array_equal(hits.text_match[i, :n], ref.text_match) \
                or not np.array_equal(hits.vector_distance[i, :n].view(np.uint32), ref.vector_distance.view(np.uint32)):
            bad.append(i)
    orc.close()
    assert not bad, "hybrid queries (rerank=%s) whose fused Topster differs from oracle.search_hybrid at 10M docs: %s" % (rerank, bad)
    assert fused >= n_q * K_VEC                                               # every query has at least its k vector hits


def test_hybrid_at_2m_docs_with_filter_ids_fused_scores_equal_the_oracle():
    n_docs, dim, n_q = 2_000_000, 64, 24
    csr, g, orc = _load(n_docs, 50_000, 24, seed=7)
    rng = np.random.default_rng(5)
    X = rng.standard_normal((n_docs, dim), dtype=np.float32)
    g.vec_create(1, dim, B.METRIC_IP, n_docs)
    for a in range(0, n_docs, 1 << 19):
        b = min(n_docs, a + (1 << 19))
        g.vec_upsert(1, np.arange(a, b, dtype=np.uint64), X[a:b])
    orc.vec_init(dim, O.METRIC_IP)
    orc.vec_add(np.arange(n_docs, dtype=np.uint32), X)
    Q = rng.standard_normal((n_q, dim)).astype(np.float32)
    qtok = synth.keyword_queries(n_q, 2, 5, 400, seed=21)
    filt_big = np.unique(rng.integers(0, n_docs, size=700_000)).astype(np.uint32)        # ~30 % of the collection
    filt_small = np.unique(rng.integers(0, n_docs, size=5_000)).astype(np.uint32)
    excl = filt_big[::3].copy()
    cases = [dict(), dict(filter_ids=filt_big), dict(filter_ids=filt_small), dict(filter_ids=filt_big, excluded_ids=excl)]
    qs = [T.KwQuery(qtok[i], sort=SORT, topster_size=0, **cases[i % 4]) for i in range(n_q)]
    hits = g.hybrid_search_batch(qs, 1, Q, k=0, fetch_size=100, alpha=0.3, k_stride=250)
    assert (hits.status == 0).all()
    loaded = set()
    for i, q in enumerate(qs):
        c = cases[i % 4]
        _need(orc, csr, loaded, q.tokens)
        oq = orc.make_query(q.tokens, sort=OSORT, fetch_size=100, **c)
        ref = orc.search_hybrid(oq, Q[i], k=0, alpha=0.3)
        n = int(hits.n_hits[i])
        assert n == ref.keys.size, (i, n, ref.keys.size)
        assert np.array_equal(hits.keys[i, :n], ref.keys), (i, hits.keys[i, :10], ref.keys[:10])
        assert np.array_equal(hits.scores[i, :n], ref.scores), i                        # fused RRF score BITS
        assert np.array_equal(hits.text_match[i, :n], ref.text_match), i
        assert np.array_equal(hits.vector_distance[i, :n].view(np.uint32), ref.vector_distance.view(np.uint32)), i
        if "filter_ids" in c:
            assert np.isin(hits.keys[i, :n], c["filter_ids"]).all()
        if "excluded_ids" in c:
            assert not np.isin(hits.keys[i, :n], c["excluded_ids"]).any()
    g.close()


def test_cosine_2m_rows_top100_equals_the_oracle_flat_scan():
    """cosine at size vs the ORACLE (not GPU vs GPU): rows normalised on insert exactly as hnsw_index_t::normalize_vector
    (include/index.h:379-388), query normalised per call; labels, order and distance BITS of the top-100"""
    n, dim, k = 2_000_000, 96, 100
    rng = np.random.default_rng(31)
    X = (rng.standard_normal((n, dim), dtype=np.float32) * rng.uniform(0.2, 5.0, size=(n, 1)).astype(np.float32))   # uneven norms
    g = T.GpuIndex(0)
    g.vec_create(1, dim, B.METRIC_COSINE, n)
    for a in range(0, n, 1 << 19):
        b = min(n, a + (1 << 19))
        g.vec_upsert(1, np.arange(a, b, dtype=np.uint64), X[a:b])
    orc = O.OracleIndex(1, 1)
    orc.vec_init(dim, O.METRIC_COSINE)
    orc.vec_add(np.arange(n, dtype=np.uint32), X)
    for lab in (0, 12345, n - 1):
        assert np.array_equal(g.vec_get(1, lab), orc.vec_get(lab))
    Q = (rng.standard_normal((12, dim)) * 3).astype(np.float32)
    dist, lab, cnt = g.vec_knn_batch(1, Q, k)
    for i in range(Q.shape[0]):
        d, l = orc.flat_knn(Q[i], k)
        assert cnt[i] == k and np.array_equal(lab[i].astype(np.uint32), l), i
        assert np.array_equal(dist[i].view(np.uint32), d.view(np.uint32)), i
    g.close()
