"""`-m gpu` twin of tests/test_emu_concurrency.py at size: 64 native threads issue blocking ONE-query calls on one context
(the reference's calling convention, src/index.cpp:3488) against a 2M-doc collection; the micro-batcher coalesces them, every
call's result equals the batch path's, and matched-id lists are per call. Also: searches during an RCU commit."""
import ctypes as C
import threading

import numpy as np
import pytest

import typesense_amd as T
from typesense_amd import _lib as B, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    n_docs = 2_000_000
    csr = synth.zipf_corpus_csr(n_docs, 50_000, 24, seed=6)
    g = T.GpuIndex(0, H.gpu_lib_path())
    g.field_create(0, False)
    g.terms_load_csr(0, csr["term_ids"], csr["ids_ptr"], csr["ids"], csr["offset_index"], csr["off_ptr"], csr["offsets"])
    g.column_set(0, synth.points_column(n_docs))
    g.set_num_docs(n_docs)
    g.commit()
    yield g, csr, n_docs
    g.close()


def test_native_threads_one_query_calls_match_the_batch_path(big):
    import bench
    g, csr, n_docs = big
    LG = bench.loadgen_lib()
    n_q = 2048
    qtok = synth.keyword_queries(n_q, 3, 8, 1500, seed=8)
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    qs = [T.KwQuery(q, sort=sort, topster_size=250) for q in qtok]
    arr = T.index.make_query_array(qs)
    hits = g.keyword_search_batch(arr, k_stride=250)
    assert (hits.status == 0).all() and hits.n_hits.sum() > 0
    want = np.array([LG.tsgpu_loadgen_hits_checksum(hits.keys[i].ctypes.data, hits.scores[i].ctypes.data, int(hits.n_hits[i]), int(hits.num_matched[i]), 100)
                     for i in range(n_q)], np.uint64)
    threads, calls = 64, 32
    lat = np.zeros(threads * calls)
    got = np.zeros(n_q, np.uint64)
    fails = C.c_uint64(0)
    r0, c0 = g.counter("batch_rounds"), g.counter("batch_coalesced_calls")
    wall = LG.tsgpu_loadgen_keyword(C.cast(g.L.tsgpu_keyword_search_batch, C.c_void_p), g.h, C.cast(arr, C.c_void_p), n_q, 250, 100, threads, calls, 1,
                                    lat.ctypes.data, got.ctypes.data, C.byref(fails))
    assert fails.value == 0 and wall > 0
    assert np.array_equal(got, want), "%d of %d one-query calls differ from the batch path" % (int((got != want).sum()), n_q)
    rounds, ccalls = g.counter("batch_rounds") - r0, g.counter("batch_coalesced_calls") - c0
    assert ccalls > 0 and rounds < ccalls, "64 concurrent callers were never coalesced (%d rounds for %d calls)" % (rounds, ccalls)


def test_per_call_id_lists_under_concurrency(big):
    g, csr, n_docs = big
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_SEQ_ID, 1, 0))
    rng = np.random.default_rng(3)
    qs = [T.KwQuery(rng.choice(np.arange(20, 400), size=2, replace=False), sort=sort, topster_size=100) for _ in range(48)]
    g.keep_result_ids(True)                      # single-caller reference values
    ref_hits = g.keyword_search_batch(qs, k_stride=100)
    ref_ids = [g.result_ids(i) for i in range(len(qs))]
    g.keep_result_ids(False)
    errs = []

    def worker(t):
        try:
            for i in range(t, len(qs), 8):
                hits, ids = g.keyword_search_batch_ids([qs[i]], k_stride=100)
                n = int(ref_hits.n_hits[i])
                assert int(hits.n_hits[0]) == n and np.array_equal(hits.keys[0, :n], ref_hits.keys[i, :n]) and np.array_equal(hits.scores[0, :n], ref_hits.scores[i, :n])
                assert np.array_equal(ids[0], ref_ids[i]), "query %d: another caller's ids" % i
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    if errs:
        raise errs[0]


def test_commit_while_searching_is_old_or_new(big):
    g, csr, n_docs = big
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_SEQ_ID, 1, 0))
    term_a, term_b = 900, 901
    qs = [T.KwQuery([term_a, term_b], sort=sort, topster_size=50)]
    before = g.keyword_search_batch(qs, k_stride=50)
    ids_a, oi_a, off_a = synth.csr_term(csr, term_a)
    ids_b, oi_b, off_b = synth.csr_term(csr, term_b)
    # new content for term_b: exactly term_a's documents -> afterwards the intersection is all of term_a
    stop = threading.Event()
    seen = set()
    errs = []

    def searcher():
        try:
            while not stop.is_set():
                h = g.keyword_search_batch(qs, k_stride=50)
                seen.add(int(h.num_matched[0]))
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
    th = threading.Thread(target=searcher)
    th.start()
    try:
        g.term_upsert(0, term_b, ids_a, oi_a, off_a)
        g.commit()
        after = g.keyword_search_batch(qs, k_stride=50)
    finally:
        stop.set()
        th.join()
        g.term_upsert(0, term_b, ids_b, oi_b, off_b)
        g.commit()
    assert not errs, errs
    assert int(after.num_matched[0]) == ids_a.size
    assert seen <= {int(before.num_matched[0]), ids_a.size}, "a search saw a half-published snapshot: %s" % seen
    again = g.keyword_search_batch(qs, k_stride=50)
    assert int(again.num_matched[0]) == int(before.num_matched[0])
