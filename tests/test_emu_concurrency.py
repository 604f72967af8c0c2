"""Concurrent callers on ONE context (the reference's calling convention: one query per request thread under a shared lock,
src/index.cpp:3488): the in-library micro-batcher coalesces them into rounds, every caller gets exactly its own results and
its own matched-id list, and commits publish RCU snapshots while searches run. Emulator tier (same sources as libtsgpu.so);
the `-m gpu` twin is tests/test_gpu_concurrency.py."""
import threading

import numpy as np
import pytest

import typesense_amd as T
from typesense_amd import _lib as B
from oracle import oracle_py as O
from tests import helpers as H


@pytest.fixture(scope="module")
def pair():
    docs = H.zipf_docs(2000, 200, 10, seed=11)
    orc, g = H.build_pair(docs, H.emu_lib_path())
    yield orc, g, docs
    g.close()


def _run_threads(n, fn):
    errs = []

    def wrap(i):
        try:
            fn(i)
        except BaseException as e:      # noqa: BLE001 (re-raised in the main thread)
            errs.append(e)
    th = [threading.Thread(target=wrap, args=(i,)) for i in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise errs[0]


def test_concurrent_one_query_calls_are_coalesced_and_exact(pair):
    orc, g, _ = pair
    rng = np.random.default_rng(5)
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    n_threads, per_thread = 8, 6
    qs = [[T.KwQuery(rng.choice(np.arange(1, 30), size=int(rng.integers(1, 4)), replace=False), sort=sort, topster_size=64)
           for _ in range(per_thread)] for _ in range(n_threads)]
    refs = [[H.oracle_keyword(orc, q, ids_cap=4096) for q in row] for row in qs]
    g.set_option("batch_window_us", 20000)          # the emulator is slow: give the other threads time to park
    r0 = g.counter("batch_rounds")
    c0 = g.counter("batch_coalesced_calls")
    start = threading.Barrier(n_threads)

    def worker(i):
        start.wait()
        for j, q in enumerate(qs[i]):
            if j % 2 == 0:
                hits = g.keyword_search_batch([q], k_stride=64)
                ids = None
            else:
                hits, ids = g.keyword_search_batch_ids([q], k_stride=64)
            assert hits.status[0] == 0
            H.assert_hits_equal(hits, 0, refs[i][j], "thread %d call %d" % (i, j))
            if ids is not None:
                assert np.array_equal(ids[0], refs[i][j].result_ids), "thread %d call %d: matched ids" % (i, j)
    _run_threads(n_threads, worker)
    rounds = g.counter("batch_rounds") - r0
    calls = g.counter("batch_coalesced_calls") - c0
    assert calls > 0 and rounds > 0
    assert rounds < calls, "no two concurrent calls ever shared a round (%d rounds, %d calls)" % (rounds, calls)
    g.set_option("batch_window_us", 10)


@pytest.mark.parametrize("round_cap,window", [(3, 0), (1024, 0), (5, 200)])
def test_lock_free_combiner_under_churn_every_call_gets_its_own_result(pair, round_cap, window):
    """the micro-batcher without a lock (tsgpu_batcher.h): arrivals push on a stack and elect the leader with one exchange, the leader
    takes the whole stack. Many short rounds with no gather window, and a round cap of 3 queries so that most of a taken stack goes
    BACK on it and the leadership is handed to a parked request: every call must come back exactly once with its own query's result
    (a request taken into a round between its push and its election, or left on the stack without a leader, would hang or cross wires)"""
    orc, g, _ = pair
    rng = np.random.default_rng(17)
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    n_threads, per_thread = 24, 10
    flat = [T.KwQuery(rng.choice(np.arange(1, 30), size=int(rng.integers(1, 4)), replace=False), sort=sort, topster_size=32) for _ in range(n_threads * per_thread)]
    ref = g.keyword_search_batch(flat, k_stride=32)                                  # the batch path (itself checked against the oracle elsewhere)
    assert (ref.status == 0).all()
    g.set_option("batch_window_us", window)
    g.set_option("batch_round_queries", round_cap)
    r0, c0 = g.counter("batch_rounds"), g.counter("batch_coalesced_calls")
    start = threading.Barrier(n_threads)
    done = [0] * n_threads
    try:
        def worker(i):
            start.wait()
            for j in range(per_thread):
                qi = i * per_thread + j
                nq = 2 if (qi % 7 == 0 and j + 1 < per_thread) else 1                 # some calls carry two queries
                hits = g.keyword_search_batch(flat[qi:qi + nq], k_stride=32)
                for u in range(nq):
                    n = int(ref.n_hits[qi + u])
                    assert hits.status[u] == 0 and int(hits.n_hits[u]) == n and int(hits.num_matched[u]) == int(ref.num_matched[qi + u]), (i, j, u)
                    assert np.array_equal(hits.keys[u, :n], ref.keys[qi + u, :n]) and np.array_equal(hits.scores[u, :n], ref.scores[qi + u, :n]), (i, j, u)
                done[i] += 1
        _run_threads(n_threads, worker)
        assert done == [per_thread] * n_threads
        rounds, calls = g.counter("batch_rounds") - r0, g.counter("batch_coalesced_calls") - c0
        assert rounds > 0 and calls >= rounds
        if round_cap <= 5:
            assert calls <= rounds * round_cap, "a round exceeded its cap (%d calls in %d rounds)" % (calls, rounds)
    finally:
        g.set_option("batch_window_us", 10)
        g.set_option("batch_round_queries", 1024)


def test_mixed_k_stride_and_failing_query_stay_per_caller(pair):
    orc, g, _ = pair
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_SEQ_ID, 1, 0))
    g.set_option("batch_window_us", 20000)
    good = T.KwQuery([1, 2], sort=sort, topster_size=32)
    bad = T.KwQuery([1, 2], sort=((B.SORT_INT64_COLUMN, 1, 77),), topster_size=32)      # unknown sort column -> 501 for that query only
    ref = H.oracle_keyword(orc, good)
    start = threading.Barrier(4)

    def worker(i):
        start.wait()
        for _ in range(3):
            if i == 0:
                hits = g.keyword_search_batch([bad], k_stride=32)
                assert hits.status[0] == B.ERR_UNSUPPORTED and hits.n_hits[0] == 0
            else:
                ks = 32 + 16 * i
                hits = g.keyword_search_batch([good], k_stride=ks)
                assert hits.status[0] == 0
                H.assert_hits_equal(hits, 0, ref, "k_stride %d" % ks)
    _run_threads(4, worker)
    g.set_option("batch_window_us", 10)


def test_a_caller_whose_k_stride_is_too_small_fails_alone(pair):
    """a coalesced round stages its results with the widest Topster of the round: the caller whose own k_stride is smaller than its
    topster_size gets 400 for its query, the callers that shared the round get their results (as they would have, called alone)"""
    orc, g, _ = pair
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_SEQ_ID, 1, 0))
    g.set_option("batch_window_us", 20000)
    good = T.KwQuery([1, 2], sort=sort, topster_size=24)
    greedy = T.KwQuery([1], sort=sort, topster_size=200)                 # its Topster holds more hits than its caller's buffer
    ref = H.oracle_keyword(orc, good)
    assert H.oracle_keyword(orc, greedy).keys.size > 24
    start = threading.Barrier(4)

    def worker(i):
        start.wait()
        for _ in range(3):
            if i == 0:
                try:
                    hits = g.keyword_search_batch([greedy], k_stride=24)
                    assert hits.status[0] == B.ERR_INVALID and hits.n_hits[0] == 0      # shared a round: its own query reports 400
                except T.TsgpuError as e:
                    assert e.code == B.ERR_INVALID                                       # ran alone (no other caller inside at that instant): the call reports 400
            else:
                hits = g.keyword_search_batch([good], k_stride=24)
                assert hits.status[0] == 0
                H.assert_hits_equal(hits, 0, ref, "shared a round with a failing caller")
    _run_threads(4, worker)
    g.set_option("batch_window_us", 10)


def test_searches_during_commits_see_old_or_new_snapshot():
    """RCU snapshots: a search that overlaps a commit returns either the pre- or the post-commit result, bit-exact vs the oracle."""
    docs = H.zipf_docs(600, 60, 8, seed=3)
    lib = H.emu_lib_path()
    orc_a, g = H.build_pair(docs[:400], lib)
    orc_b = H.build_pair(docs, lib)[0]
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_SEQ_ID, 1, 0))
    qs = [T.KwQuery([1, 2], sort=sort, topster_size=50), T.KwQuery([3], sort=sort, topster_size=50), T.KwQuery([2, 4, 5], sort=sort, topster_size=50)]
    ref_a = [H.oracle_keyword(orc_a, q) for q in qs]
    ref_b = [H.oracle_keyword(orc_b, q) for q in qs]
    stop = threading.Event()
    seen = {"a": 0, "b": 0}
    errors = []                     # an AssertionError inside a Thread target never reaches pytest: collect, assert on the main thread

    def same(hits, i, ref):
        n = int(hits.n_hits[i])
        return n == ref.keys.size and np.array_equal(hits.keys[i, :n], ref.keys) and np.array_equal(hits.scores[i, :n], ref.scores) \
            and int(hits.num_matched[i]) == int(ref.num_keyword_matches)

    def searcher(_):
        while not stop.is_set():
            hits = g.keyword_search_batch(qs, k_stride=50)
            a = all(same(hits, i, ref_a[i]) for i in range(len(qs)))
            b = all(same(hits, i, ref_b[i]) for i in range(len(qs)))
            if not (a or b):
                errors.append("a search saw a mix of two snapshots")
                return
            seen["a" if a else "b"] += 1

    def guarded(i):
        try:
            searcher(i)
        except Exception as e:      # noqa: BLE001 — whatever a searcher dies of must fail the test
            errors.append(repr(e))

    th = [threading.Thread(target=guarded, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    try:
        # the second half of the collection arrives: upsert every term's full list (the decoded blocks), publish once
        for term in orc_b.terms(0):
            ids, oi, off = orc_b.dump_posting(0, int(term))
            g.term_upsert(0, int(term), ids, oi, off)
        g.set_num_docs(600)
        g.commit()
        hits = g.keyword_search_batch(qs, k_stride=50)
        assert all(same(hits, i, ref_b[i]) for i in range(len(qs)))
    finally:
        stop.set()
        for t in th:
            t.join()
    g.close()
    assert not errors, errors
    assert seen["a"] + seen["b"] > 0


def test_searches_during_incremental_commits_see_a_published_snapshot():
    """the same with INCREMENTAL commits (posting_upsert appends + mid-list rewrites: descriptor scatter into spare entries and tail
    appends happen while searchers run): every search equals the oracle of one of the published states, never a mix"""
    n0, step, n_steps = 500, 60, 3
    docs = H.zipf_docs(n0 + step * n_steps, 50, 7, seed=17)
    lib = H.emu_lib_path()
    g = T.GpuIndex(0, lib)
    g.field_create(0, False)
    for d in range(n0):
        g.index_plain_doc(d, 0, docs[d])
    g.column_set(0, H.points_of(docs.shape[0]))
    g.set_num_docs(docs.shape[0])
    g.commit()
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_SEQ_ID, 1, 0))
    qs = [T.KwQuery([1, 2], sort=sort, topster_size=50), T.KwQuery([3], sort=sort, topster_size=50), T.KwQuery([2, 4, 5], sort=sort, topster_size=50)]

    def state_oracle(cur_docs, n_live):
        orc = O.OracleIndex(1, 1)
        for d in range(n_live):
            orc.index_plain(d, 0, cur_docs[d])
        orc.set_num_docs(docs.shape[0])
        orc.set_sort_dense(0, H.points_of(docs.shape[0]))
        return [H.oracle_keyword(orc, q) for q in qs]

    # the published states, computed up front (the updates are deterministic)
    rng = np.random.default_rng(3)
    plans, cur, states = [], docs.copy(), [state_oracle(docs, n0)]
    for s_ in range(n_steps):
        upd = [(int(d), rng.integers(1, 20, size=docs.shape[1]).astype(np.uint32)) for d in rng.choice(n0, size=5, replace=False)]
        plans.append(upd)
        for d, toks in upd:
            cur[d] = toks
        states.append(state_oracle(cur, n0 + step * (s_ + 1)))
    stop = threading.Event()
    errors, seen = [], [0] * len(states)

    def same(hits, i, ref):
        n = int(hits.n_hits[i])
        return n == ref.keys.size and np.array_equal(hits.keys[i, :n], ref.keys) and np.array_equal(hits.scores[i, :n], ref.scores) \
            and int(hits.num_matched[i]) == int(ref.num_keyword_matches)

    def searcher(_):
        try:
            while not stop.is_set():
                hits = g.keyword_search_batch(qs, k_stride=50)
                which = [k for k, st in enumerate(states) if all(same(hits, i, st[i]) for i in range(len(qs)))]
                if not which:
                    errors.append("a search matched none of the published snapshots")
                    return
                seen[which[0]] += 1
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=searcher, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    try:
        live_docs = docs.copy()
        for s_ in range(n_steps):
            for d in range(n0 + step * s_, n0 + step * (s_ + 1)):
                g.index_plain_doc(d, 0, live_docs[d])
            for d, toks in plans[s_]:
                g.remove_plain_doc(d, 0, live_docs[d])
                live_docs[d] = toks
                g.index_plain_doc(d, 0, toks)
            g.commit()
        assert g.counter("commit_incremental_count") >= n_steps
        hits = g.keyword_search_batch(qs, k_stride=50)
        assert all(same(hits, i, states[-1][i]) for i in range(len(qs)))
    finally:
        stop.set()
        for t in th:
            t.join()
    g.close()
    assert not errors, errors
    assert sum(seen) > 0


def test_concurrent_knn_calls_share_one_scan():
    lib = H.emu_lib_path()
    rng = np.random.default_rng(9)
    X = rng.standard_normal((700, 48)).astype(np.float32)
    Q = rng.standard_normal((12, 48)).astype(np.float32)
    g = T.GpuIndex(0, lib)
    g.vec_create(3, 48, B.METRIC_IP)
    g.vec_upsert(3, np.arange(700, dtype=np.uint64), X)
    want_d, want_l, want_c = g.vec_knn_batch(3, Q, 10)
    g.set_option("batch_window_us", 20000)
    c0 = g.counter("batch_coalesced_calls")
    start = threading.Barrier(6)

    def worker(i):
        start.wait()
        for j in (i, i + 6):
            d, l, c = g.vec_knn_batch(3, Q[j:j + 1], 10)
            assert c[0] == want_c[j] and np.array_equal(l[0], want_l[j])
            assert np.array_equal(d[0].view(np.uint32), want_d[j].view(np.uint32))
    _run_threads(6, worker)
    assert g.counter("batch_coalesced_calls") - c0 > 0
    g.close()
