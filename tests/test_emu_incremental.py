"""Incremental, block-granular commits (tsgpu_index.hip): single-document posting_t::upsert / erase calls mutate one block of one list,
tsgpu_commit uploads only the changed blocks + the touched lists' descriptors to the arena tails and publishes a new descriptor table.
After every commit the mirror must equal an index built from scratch: format round trip vs the oracle's postings and bit-exact
search results (incl. lists whose relocated blocks break the coalesced runs: LIST_HAS_BREAKS -> per-candidate probes).
Emulator tier; the `-m gpu` twin (tests/test_gpu_incremental.py) adds the 10M-doc timing bound."""
import numpy as np
import pytest

import typesense_amd as T
from typesense_amd import _lib as B
from oracle import oracle_py as O
from tests import helpers as H


def fresh_oracle(docs, live):
    orc = O.OracleIndex(1, 1)
    for d in range(docs.shape[0]):
        if live[d]:
            orc.index_plain(d, 0, docs[d])
    orc.set_num_docs(docs.shape[0])
    orc.set_sort_dense(0, H.points_of(docs.shape[0]))
    return orc


def check_equal(g, orc, rng, what, n_queries=10):
    for term in orc.terms(0):
        ids, oi, off = orc.dump_posting(0, int(term))
        gi, go, gf = g.term_download(0, int(term))
        assert np.array_equal(ids, gi) and np.array_equal(oi, go) and np.array_equal(off, gf), "%s: posting list of term %d" % (what, term)
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    qs = [T.KwQuery(rng.choice(np.arange(1, 25), size=int(rng.integers(1, 4)), replace=False), sort=sort, topster_size=250) for _ in range(n_queries)]
    qs.append(T.KwQuery([1, 2], sort=sort, topster_size=250, filter_ids=np.arange(0, 5000, 3, dtype=np.uint32)))
    hits = g.keyword_search_batch(qs, k_stride=250)
    assert (hits.status == 0).all()
    for i, q in enumerate(qs):
        H.assert_hits_equal(hits, i, H.oracle_keyword(orc, q), what)


def test_appends_updates_and_removals_publish_incrementally():
    rng = np.random.default_rng(2)
    n0, n1 = 2600, 3000
    docs = H.zipf_docs(n1, 60, 7, seed=9)
    live = np.zeros(n1, bool)
    live[:n0] = True
    g = T.GpuIndex(0, H.emu_lib_path())
    g.field_create(0, False)
    for d in range(n0):
        g.index_plain_doc(d, 0, docs[d])
    g.column_set(0, H.points_of(n1))
    g.set_num_docs(n1)
    g.commit()
    assert g.counter("commit_full_count") == 1
    check_equal(g, fresh_oracle(docs, live), rng, "initial build through posting_upsert")

    # 1) a write batch of new documents (largest ids: every touched list grows at its end)
    for d in range(n0, n0 + 150):
        g.index_plain_doc(d, 0, docs[d])
        live[d] = True
    g.commit()
    assert g.counter("commit_incremental_count") == 1 and g.counter("commit_full_count") == 1
    full_bytes = g.device_bytes()
    assert g.counter("commit_last_uploaded_bytes") < full_bytes / 4, "an append batch re-uploaded most of the index"
    check_equal(g, fresh_oracle(docs, live), rng, "after appending 150 documents")

    # 2) updates of existing documents (erase the old postings, upsert the new ones: blocks in the middle of lists change, some split)
    for d in rng.choice(n0, size=120, replace=False):
        g.remove_plain_doc(int(d), 0, docs[d])
        docs[d] = rng.integers(1, 30, size=docs.shape[1])
        g.index_plain_doc(int(d), 0, docs[d])
    # 3) removals, incl. every document of a rare term (the term disappears) ...
    for d in rng.choice(n0, size=80, replace=False):
        g.remove_plain_doc(int(d), 0, docs[d])
        live[d] = False
    rare = 59
    for d in range(n1):
        if live[d] and rare in docs[d]:
            g.remove_plain_doc(d, 0, docs[d])
            live[d] = False
    # ... and the rest of the new documents, in one commit
    for d in range(n0 + 150, n1):
        if rare in docs[d]:
            continue
        g.index_plain_doc(d, 0, docs[d])
        live[d] = True
    g.commit()
    assert g.counter("commit_incremental_count") == 2
    orc = fresh_oracle(docs, live)
    assert rare not in set(int(t) for t in orc.terms(0)) and g.term_num_ids(0, rare) == 0
    check_equal(g, orc, rng, "after updates, removals and appends")
    # a vanished term is skipped like any token absent from the index (src/index.cpp:5651-5655)
    hits = g.keyword_search_batch([T.KwQuery([1, rare], topster_size=50)], k_stride=50)
    H.assert_hits_equal(hits, 0, H.oracle_keyword(orc, T.KwQuery([1, rare], topster_size=50)), "vanished term")

    # 4) a forced compaction restores contiguous lists and changes nothing
    g.set_option("commit_full", 1)
    g.commit()
    assert g.counter("commit_full_count") == 2
    check_equal(g, orc, rng, "after compaction")
    g.close()


def test_many_small_commits_until_the_tails_run_out():
    rng = np.random.default_rng(5)
    n = 1500
    docs = H.zipf_docs(n, 40, 6, seed=4)
    live = np.zeros(n, bool)
    g = T.GpuIndex(0, H.emu_lib_path())
    g.set_option("index_min_slack_words", 3000)       # (tiny arenas: the tails fill up within a few batches)
    g.field_create(0, False)
    g.column_set(0, H.points_of(n))
    g.set_num_docs(n)
    for d in range(200):
        g.index_plain_doc(d, 0, docs[d])
        live[d] = True
    g.commit()
    for a in range(200, n, 100):                  # 13 write batches; the arenas were sized for the first 200 documents
        for d in range(a, a + 100):
            g.index_plain_doc(d, 0, docs[d])
            live[d] = True
        g.commit()
    assert g.counter("commit_incremental_count") >= 1 and g.counter("commit_full_count") >= 2, "the tails never filled up / never compacted"
    check_equal(g, fresh_oracle(docs, live), rng, "after 13 write batches", n_queries=6)
    g.close()


def test_rewrites_accumulate_garbage_until_a_commit_compacts():
    """every re-written block leaves its old words behind in the arenas; the commit after which the garbage would outweigh the live words
    takes the full path instead (compaction: re-pack, tail room restored, no list with breaks) — a lightly written index with large
    tails must not keep its garbage (and its per-candidate probes across breaks) for ever"""
    rng = np.random.default_rng(11)
    n = 5000
    docs = H.zipf_docs(n, 30, 6, seed=6)
    live = np.ones(n, bool)
    g = T.GpuIndex(0, H.emu_lib_path())
    g.set_option("index_min_slack_words", 1 << 20)     # (room for hundreds of re-writes: the tails never run out here)
    g.set_option("index_compact_min_words", 0)
    g.field_create(0, False)
    g.column_set(0, H.points_of(n))
    g.set_num_docs(n)
    for d in range(n):
        g.index_plain_doc(d, 0, docs[d])
    g.commit()
    live0 = g.counter("index_live_words")
    assert g.counter("index_used_words") == live0 and g.counter("commit_compactions") == 0
    rounds, peak = 0, 0
    while g.counter("commit_compactions") == 0 and rounds < 40:      # remove + re-add the same documents: the lists' content does not grow, the arenas do
        pick = rng.choice(n, size=4, replace=False)                   # (a few blocks of a few lists per round)
        for d in pick:
            g.remove_plain_doc(int(d), 0, docs[d])
        g.commit()
        for d in pick:
            g.index_plain_doc(int(d), 0, docs[d])
        g.commit()
        rounds += 1
        used, lv = g.counter("index_used_words"), g.counter("index_live_words")
        if g.counter("commit_compactions") == 0:
            peak = max(peak, used)
        assert lv <= used and abs(int(lv) - int(live0)) <= live0 // 8, "live words drifted: %d vs %d at the start" % (lv, live0)
    assert g.counter("commit_compactions") == 1 and rounds >= 2, "garbage never triggered a compaction (%d rounds)" % rounds
    used, lv = g.counter("index_used_words"), g.counter("index_live_words")
    assert used < peak and used - lv < lv // 2 and peak > 3 * live0 // 2, "the compaction did not shrink the arenas: %d words used (peak %d, live %d)" % (used, peak, lv)
    check_equal(g, fresh_oracle(docs, live), rng, "after the compaction", n_queries=6)
    g.close()


def test_a_failing_commit_leaves_the_previous_snapshot_and_the_retry_is_complete():
    """fault injection (emulator: the N-th hipMalloc / hipMemcpy fails): wherever a commit dies — incremental path, its full
    fallback, first upload or last — searches keep answering from the snapshot published before, and the next commit (forced onto
    the full path: a failed attempt has left arena positions in the host lists that never reached the device) publishes everything"""
    import ctypes as C
    lib = H.emu_lib_path()
    emu = C.CDLL(lib)
    emu.hipemu_fail_nth.argtypes = [C.c_long]
    rng = np.random.default_rng(12)
    n0, n1 = 900, 1300
    docs = H.zipf_docs(n1, 40, 6, seed=21)
    live = np.zeros(n1, bool)
    live[:n0] = True
    g = T.GpuIndex(0, lib)
    g.field_create(0, False)
    for d in range(n0):
        g.index_plain_doc(d, 0, docs[d])
    g.column_set(0, H.points_of(n1))
    g.set_num_docs(n1)
    g.commit()
    old = fresh_oracle(docs, live)
    check_equal(g, old, rng, "before the faults", n_queries=4)
    done = n0
    failures = 0
    for nth in (1, 2, 3, 5, 8, 13):                      # die at a different device call of the commit each time
        for d in range(done, done + 40):
            g.index_plain_doc(d, 0, docs[d])
            live[d] = True
        # an update in the middle of published lists as well (re-written blocks, not only appended ones)
        upd = int(rng.integers(0, n0))
        if live[upd]:
            g.remove_plain_doc(upd, 0, docs[upd])
            docs[upd] = rng.integers(1, 30, size=docs.shape[1])
            g.index_plain_doc(upd, 0, docs[upd])
        done += 40
        emu.hipemu_fail_nth(nth)
        try:
            g.commit()
            failed = False
        except T.TsgpuError as e:
            failed = True
            assert e.code in (B.ERR_DEVICE, B.ERR_NO_MEMORY), e
        emu.hipemu_fail_nth(0)
        if failed:
            failures += 1
            check_equal(g, old, rng, "after a commit that failed at device call %d: the previous snapshot must still answer" % nth, n_queries=3)
            full_before = g.counter("commit_full_count")
            g.commit()                                   # the retry
            assert g.counter("commit_full_count") == full_before + 1, "the retry after a failed commit must re-pack from the host lists"
        old = fresh_oracle(docs, live)
        check_equal(g, old, rng, "after the retry of the commit that failed at device call %d" % nth, n_queries=4)
    assert failures >= 4 and g.counter("commit_failed_count") == failures
    g.close()


def test_term_created_after_the_first_commit_is_published_by_the_incremental_commit():
    """a term CREATED after a commit (tsgpu_term_upsert / tsgpu_posting_upsert / tsgpu_terms_load_csr) must be queued
    for the next — incremental — commit. TermHost::dirty starts true, so it cannot double as the 'already queued' marker."""
    docs = H.zipf_docs(1500, 20, 6, seed=4)
    orc, g = H.build_pair(docs, H.emu_lib_path())
    assert g.counter("commit_full_count") == 1
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    # a) a whole new list through term_upsert  b) a new term through single-document posting_upsert  c) through the CSR loader
    ids_a = np.arange(3, 1500, 7, dtype=np.uint32)
    g.term_upsert(0, 900, ids_a, np.arange(ids_a.size, dtype=np.uint32), np.full(ids_a.size, 2, np.uint32))
    for d in (5, 17, 400):
        g.posting_upsert(0, 901, d, np.array([3], np.uint32))
    ids_c = np.arange(1, 1500, 11, dtype=np.uint32)
    g.terms_load_csr(0, np.array([902], np.uint32), np.array([0, ids_c.size], np.uint64), ids_c, np.arange(ids_c.size, dtype=np.uint32),
                     np.array([0, ids_c.size], np.uint64), np.full(ids_c.size, 4, np.uint32))
    g.commit()
    assert g.counter("commit_incremental_count") == 1 and g.counter("commit_full_count") == 1, "the commit did not take the incremental path"
    hits = g.keyword_search_batch([T.KwQuery([900], sort=sort, topster_size=250), T.KwQuery([901], sort=sort, topster_size=250),
                                   T.KwQuery([902], sort=sort, topster_size=250), T.KwQuery([1], sort=sort, topster_size=250)], k_stride=250)
    assert (hits.status == 0).all()
    assert int(hits.num_matched[0]) == ids_a.size and int(hits.num_matched[1]) == 3 and int(hits.num_matched[2]) == ids_c.size
    assert int(hits.num_matched[3]) > 0
    gi, _, _ = g.term_download(0, 901)
    assert np.array_equal(gi, np.array([5, 17, 400], np.uint32))
    # a second write to the SAME new terms is queued again after the commit cleared the marker
    g.posting_upsert(0, 901, 900, np.array([1], np.uint32))
    g.commit()
    assert g.counter("commit_incremental_count") == 2
    gi, _, _ = g.term_download(0, 901)
    assert np.array_equal(gi, np.array([5, 17, 400, 900], np.uint32))
    g.close()


def test_id_directories_follow_commits_and_never_change_a_result():
    """ID DIRECTORIES of the long lists (tsgpu_format.h: {position, bits} per 32 doc ids; probe_list answers from ONE load): which lists carry
    one, that an incremental commit rebuilds exactly the directories of the long lists it touched and shares the others with the previous
    snapshot, that the pool is replaced when the collection outgrows its id range, entries that straddle two blocks (dense lists: most block
    boundaries), ids beyond the pool's range — and that switching the directories off changes nothing (3..5-token queries = the oracle)."""
    rng = np.random.default_rng(31)
    n0, n1 = 3000, 8000
    docs = H.zipf_docs(n1, 40, 9, seed=17)
    docs[:, 0] = 1                                                 # term 1 is in every document: consecutive ids, every entry full, every block boundary inside an entry
    live = np.zeros(n1, bool)
    live[:n0] = True
    g = T.GpuIndex(0, H.emu_lib_path())
    g.field_create(0, False)
    for d in range(n0):
        g.index_plain_doc(d, 0, docs[d])
    g.column_set(0, H.points_of(n1))
    g.set_num_docs(n0)                                             # (the server's num_seq_ids(): the directories' id range is sized from it)
    g.commit()
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))

    def check(what):
        orc = fresh_oracle(docs, live)
        orc.set_num_docs(int(np.nonzero(live)[0].max()) + 1)
        qs = [T.KwQuery(rng.choice(np.arange(1, 30), size=n_tok, replace=False), sort=sort, topster_size=250) for n_tok in (3, 3, 3, 4, 5, 3, 2, 3) for _ in range(3)]
        qs.append(T.KwQuery([35, 1, 2], sort=sort, topster_size=250))        # a short driver against the two longest lists
        qs.append(T.KwQuery([1, 2, 3], sort=sort, topster_size=100, filter_ids=np.arange(0, n1, 3, dtype=np.uint32)))
        hits = g.keyword_search_batch(qs, k_stride=250)
        assert (hits.status == 0).all() and hits.n_hits.sum() > 500
        for i, q in enumerate(qs):
            H.assert_hits_equal(hits, i, H.oracle_keyword(orc, q), what)
        return qs, hits

    n_long = sum(1 for t in range(1, 41) if g.term_num_ids(0, t) >= 256)
    assert n_long >= 10 and g.counter("kw_iddir_lists") == n_long and g.counter("kw_iddir_built") == n_long
    qs, with_dir = check("initial build")
    # a write batch that touches three long lists and one short one: three directories rebuilt, the rest shared
    built = g.counter("kw_iddir_built")
    for d in range(n0, n0 + 40):
        docs[d] = 0
        docs[d, :4] = (1, 2, 3, 39)
        g.index_plain_doc(d, 0, docs[d])
        live[d] = True
    g.commit()
    assert g.counter("commit_incremental_count") == 1
    assert g.counter("kw_iddir_built") == built + 3 and g.counter("kw_iddir_lists") == n_long
    check("after an append batch")
    # updates / removals in the middle of long lists (blocks re-packed at the tail, partial blocks in mid-list)
    built = g.counter("kw_iddir_built")
    for d in rng.choice(n0, size=150, replace=False):
        g.remove_plain_doc(int(d), 0, docs[d])
        if d % 3:
            docs[d] = rng.integers(1, 30, size=docs.shape[1])
            g.index_plain_doc(int(d), 0, docs[d])
        else:
            live[d] = False
    g.commit()
    assert g.counter("kw_iddir_built") > built
    check("after updates and removals")
    # the collection outgrows the pool's id range (3000 + 1/8 + 1024 rounded up to 6144) while num_docs still says 3000: the pool stays, ids
    # beyond its range are found by the regular probe ...
    for d in range(n0 + 40, 6500):
        g.index_plain_doc(d, 0, docs[d])
        live[d] = True
    full = g.counter("commit_full_count")
    g.commit()
    assert g.counter("commit_full_count") == full
    check("ids beyond the directories' range")
    # ... and once num_docs is raised, the next (incremental) commit replaces the pool: every long list gets a directory over the new range
    built = g.counter("kw_iddir_built")
    g.set_num_docs(6500)
    g.index_plain_doc(6500, 0, docs[6500]); live[6500] = True
    g.commit()
    assert g.counter("commit_full_count") == full
    n_long2 = sum(1 for t in range(1, 41) if g.term_num_ids(0, t) >= 256)
    assert g.counter("kw_iddir_lists") == n_long2 and g.counter("kw_iddir_built") == built + n_long2
    check("after the pool was replaced")
    # ... and with the directories switched off every result is the same
    qs2, on = check("directories on")
    g.set_option("kw_iddir_min_ids", 0)
    g.commit()
    assert g.counter("kw_iddir_lists") == 0
    off = g.keyword_search_batch(qs2, k_stride=250)
    for i in range(len(qs2)):
        n = int(on.n_hits[i])
        assert off.n_hits[i] == n and np.array_equal(off.keys[i, :n], on.keys[i, :n]) and np.array_equal(off.scores[i, :n], on.scores[i, :n]) and off.num_matched[i] == on.num_matched[i]
    g.close()
