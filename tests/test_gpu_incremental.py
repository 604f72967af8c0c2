"""`-m gpu`: the write path at size. 1 000 new documents are indexed into the 10M-doc collection through the reference's own mutation
calls (posting_t::upsert per (token, document)); tsgpu_commit must publish them incrementally — O(changed blocks), < 50 ms — while a
searcher thread keeps querying: every result it sees is bit-exact vs the oracle for either the old or the new collection."""
import threading
import time

import numpy as np
import pytest

import typesense_amd as T
from typesense_amd import _lib as B, synth
from oracle import oracle_py as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_1000_doc_write_batch_publishes_incrementally_at_10m():
    n_docs, vocab, tpd, n_new = 10_000_000, 100_000, 32, 1000
    csr = synth.zipf_corpus_csr(n_docs, vocab, tpd, seed=2)
    g = T.GpuIndex(0, H.gpu_lib_path())
    g.field_create(0, False)
    g.terms_load_csr(0, csr["term_ids"], csr["ids_ptr"], csr["ids"], csr["offset_index"], csr["off_ptr"], csr["offsets"])
    pts = synth.points_column(n_docs + n_new)
    g.column_set(0, pts)
    g.set_num_docs(n_docs + n_new)
    t0 = time.time()
    g.commit()
    t_full = time.time() - t0
    assert g.counter("commit_full_count") == 1
    new_docs = H.zipf_docs(n_new, vocab, tpd, seed=77)

    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    qtok = synth.keyword_queries(24, 3, 8, 400, seed=5)             # frequent terms: every query gains hits from the new documents
    qs = [T.KwQuery(q, sort=sort, topster_size=250) for q in qtok]
    osort = ((O.SORT_TEXT_MATCH, 0, 1), (O.SORT_INT64_COLUMN, 0, 1))

    def oracle_results(with_new):
        orc = O.OracleIndex(1, 1)
        orc.set_num_docs(n_docs + n_new)
        orc.set_sort_dense(0, pts)
        for t in np.unique(qtok):
            ids, oi, off = synth.csr_term(csr, t)
            ids, oi, off = list(ids), list(oi), list(off)
            if with_new:
                for d in range(n_new):
                    pos = np.nonzero(new_docs[d] == t)[0]
                    if pos.size:
                        oi.append(len(off))
                        off.extend((pos + 1).tolist())
                        if pos[-1] == tpd - 1:
                            off.append(0)
                        ids.append(n_docs + d)
            orc.load_posting(0, int(t), np.array(ids, np.uint32), np.array(oi, np.uint32), np.array(off, np.uint32))
        return [orc.search_keyword(orc.make_query(q, sort=osort, fetch_size=100)) for q in qtok]

    ref_old, ref_new = oracle_results(False), oracle_results(True)
    assert sum(int(a.num_keyword_matches != b.num_keyword_matches) for a, b in zip(ref_old, ref_new)) > 0

    def same(hits, i, ref):
        n = int(hits.n_hits[i])
        return n == ref.keys.size and np.array_equal(hits.keys[i, :n], ref.keys) and np.array_equal(hits.scores[i, :n], ref.scores) \
            and int(hits.num_matched[i]) == int(ref.num_keyword_matches)

    hits = g.keyword_search_batch(qs, k_stride=250)
    assert all(same(hits, i, ref_old[i]) for i in range(len(qs)))

    stop = threading.Event()
    seen = {"old": 0, "new": 0}
    errs = []

    def searcher():
        try:
            while not stop.is_set():
                h = g.keyword_search_batch(qs, k_stride=250)
                a = all(same(h, i, ref_old[i]) for i in range(len(qs)))
                b = all(same(h, i, ref_new[i]) for i in range(len(qs)))
                assert a or b, "a search saw a half-published write batch"
                seen["old" if a else "new"] += 1
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
    th = threading.Thread(target=searcher)
    th.start()
    try:
        t0 = time.time()
        for d in range(n_new):
            g.index_plain_doc(n_docs + d, 0, new_docs[d])
        t_mut = time.time() - t0
        t0 = time.time()
        g.commit()
        t_commit = time.time() - t0
    finally:
        stop.set()
        th.join()
    assert not errs, errs
    commit_us = g.counter("commit_last_us")
    print("full commit %.2f s; %d posting upserts %.3f s (python loop); incremental commit %.1f ms (library: %.1f ms), %d bytes uploaded, searches during the write: %s"
          % (t_full, n_new * tpd, t_mut, 1e3 * t_commit, commit_us / 1e3, g.counter("commit_last_uploaded_bytes"), seen))
    assert g.counter("commit_incremental_count") == 1 and g.counter("commit_full_count") == 1
    assert g.counter("commit_last_uploaded_bytes") < 400e6
    assert commit_us < 50_000, "publishing a 1000-document write batch took %.1f ms" % (commit_us / 1e3)
    hits = g.keyword_search_batch(qs, k_stride=250)
    for i in range(len(qs)):
        assert same(hits, i, ref_new[i]), "query %d after the write batch" % i
    g.close()
