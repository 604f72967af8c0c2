"""Keyword hot path (kw_kernels.hip.h + the planning code of tsgpu.hip) executed on the CPU under the SIMT
emulator of tests/hipemu, checked bit-exactly against the oracle. Same sources as libtsgpu.so; logic-level
coverage for the container that has no GPU. The `-m gpu` twin is tests/test_gpu_keyword.py."""
import numpy as np
import pytest

import typesense_amd as T
from typesense_amd import _lib as B
from tests import helpers as H


@pytest.fixture(scope="module")
def pair():
    docs = H.zipf_docs(3000, 300, 12, seed=1)
    orc, g = H.build_pair(docs, H.emu_lib_path())
    yield orc, g, docs
    g.close()


def _queries(rng, n, vocab_hi, n_tok, **kw):
    out = []
    for _ in range(n):
        toks = rng.choice(np.arange(1, vocab_hi), size=n_tok, replace=False)
        out.append(T.KwQuery(toks, **kw))
    return out


def test_posting_format_roundtrip(pair):
    orc, g, _ = pair
    for term in [1, 2, 17, 150, 299]:
        ids, oi, off = orc.dump_posting(0, term)
        gi, go, gf = g.term_download(0, term)
        assert np.array_equal(ids, gi) and np.array_equal(oi, go) and np.array_equal(off, gf)


@pytest.mark.parametrize("n_tok", [1, 2, 3])
def test_keyword_topk_bit_exact_small_T(pair, n_tok):
    orc, g, _ = pair
    rng = np.random.default_rng(100 + n_tok)
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    qs = _queries(rng, 6, 40, n_tok, sort=sort, topster_size=250)
    hits = g.keyword_search_batch(qs, k_stride=250)
    assert (hits.status == 0).all()
    for i, q in enumerate(qs):
        H.assert_hits_equal(hits, i, H.oracle_keyword(orc, q), "T=%d" % n_tok)


def test_keyword_generic_T_up_to_10(pair):
    orc, g, _ = pair
    rng = np.random.default_rng(7)
    qs = []
    for n_tok in (4, 5, 7, 10):
        qs += _queries(rng, 2, 12, n_tok, sort=((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0)), topster_size=250)
    hits = g.keyword_search_batch(qs, k_stride=250)
    assert (hits.status == 0).all()
    assert hits.n_hits.sum() > 0
    for i, q in enumerate(qs):
        H.assert_hits_equal(hits, i, H.oracle_keyword(orc, q), "generic")


def test_keyword_duplicate_missing_tokens_and_flags(pair):
    orc, g, _ = pair
    base = dict(topster_size=250)
    qs = [
        T.KwQuery([3, 3], **base),                               # same token twice ("mong mong")
        T.KwQuery([2, 5, 2], **base),
        T.KwQuery([1, 9999, 4], **base),                         # token absent from the index is skipped
        T.KwQuery([9999], **base),                               # nothing left -> 0 hits
        T.KwQuery([1, 2], prioritize_token_position=True, **base),
        T.KwQuery([6], prioritize_token_position=True, **base),
        T.KwQuery([1, 2, 3], prioritize_exact_match=False, **base),
        T.KwQuery([1, 4], prioritize_num_matching_fields=False, **base),
        T.KwQuery([2, 3], match_type=B.MAX_WEIGHT, weight=7, **base),
        T.KwQuery([2, 3], match_type=B.MAX_WEIGHT, weight=0, **base),
        T.KwQuery([2, 3], match_type=B.SUM_SCORE, weight=3, **base),
        T.KwQuery([5, 1], total_cost=3, **base),
        T.KwQuery([8], total_cost=0, **base),                    # single token, verbatim check
        T.KwQuery([1, 2], sort=((B.SORT_INT64_COLUMN, -1, 0), (B.SORT_TEXT_MATCH, 1, 0), (B.SORT_SEQ_ID, -1, 0)), **base),
        T.KwQuery([1, 2], sort=((B.SORT_TEXT_MATCH, -1, 0), (B.SORT_SEQ_ID, 1, 0)), **base),   # ASC text match: :5541 override quirk
        T.KwQuery([1, 3], sort=((B.SORT_SEQ_ID, 1, 0),), **base),                               # no text_match slot
    ]
    hits = g.keyword_search_batch(qs, k_stride=250)
    assert (hits.status == 0).all()
    for i, q in enumerate(qs):
        H.assert_hits_equal(hits, i, H.oracle_keyword(orc, q), "flags")
    assert hits.n_hits[3] == 0


def test_keyword_small_topster_and_result_ids(pair):
    """topster_size 5 forces the threshold/compaction path; matched ids come back ascending like id_buff"""
    orc, g, _ = pair
    g.keep_result_ids(True)
    qs = [T.KwQuery([1, 2], topster_size=5, sort=((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))),
          T.KwQuery([1], topster_size=3), T.KwQuery([2, 1, 3], topster_size=1)]
    hits = g.keyword_search_batch(qs, k_stride=8)
    for i, q in enumerate(qs):
        ref = H.oracle_keyword(orc, q, ids_cap=4000)
        n = int(hits.n_hits[i])
        assert n == min(q.topster_size, ref.keys.size) == ref.keys.size
        H.assert_hits_equal(hits, i, ref, "small-k")
        assert np.array_equal(g.result_ids(i), ref.result_ids)
    g.keep_result_ids(False)


def test_keyword_excluded_ids(pair):
    orc, g, _ = pair
    q0 = T.KwQuery([1, 2], topster_size=250)
    all_ids = H.oracle_keyword(orc, q0, ids_cap=4000).result_ids
    excl = np.sort(all_ids[::3])
    q = T.KwQuery([1, 2], topster_size=250, excluded_ids=excl)
    g.keep_result_ids(True)
    hits = g.keyword_search_batch([q], k_stride=250)
    ref = H.oracle_keyword(orc, q, ids_cap=4000)
    H.assert_hits_equal(hits, 0, ref, "excluded")
    assert np.array_equal(g.result_ids(0), ref.result_ids)
    g.keep_result_ids(False)


def test_unsupported_queries_are_reported_per_query(pair):
    _, g, _ = pair
    qs = [T.KwQuery([1, 2]), T.KwQuery([1], n_fields=5),
          T.KwQuery(list(range(1, 12))), T.KwQuery([1], topster_size=5000)]
    hits = g.keyword_search_batch(qs, k_stride=250)
    assert list(hits.status) == [0, B.ERR_UNSUPPORTED, B.ERR_UNSUPPORTED, B.ERR_UNSUPPORTED]
    assert hits.n_hits[1] == 0 and hits.n_hits[0] > 0


@pytest.mark.parametrize("chunk", [0, 1])
def test_keyword_filter_ids_hits_ids_and_the_reference_match_count(pair, chunk):
    """filter ids inside the AND loop (take_id, src/or_iterator.cpp:218-272): hits = intersection & filter; num_keyword_matches
    = the intersection ids the reference's skip-to-next-filter-id loop VISITS (include/or_iterator.h:61-182), also across
    work-item boundaries (chunk=1: one work item per driver block)"""
    orc, g, _ = pair
    rng = np.random.default_rng(31)
    g.set_option("kw_chunk_blocks", chunk)
    g.keep_result_ids(True)
    try:
        sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
        all12 = H.oracle_keyword(orc, T.KwQuery([1, 2], topster_size=250), ids_cap=4000).result_ids
        filters = [
            np.sort(rng.choice(3000, size=40, replace=False)),                 # sparse filter: most visited ids are NOT hits
            np.sort(rng.choice(3000, size=1500, replace=False)),               # dense filter
            np.arange(3000),                                                   # everything
            np.array([0]), np.array([2999]), np.array([1500, 1501, 1502]),
            all12[::2],                                                        # only true hits
            np.sort(np.concatenate([all12[:5], [all12[5] + 1 if all12[5] + 1 not in all12 else all12[5]]])),
            np.arange(0, 700),                                                 # filter ends early: the loop breaks
            np.arange(2500, 3000),                                             # filter starts late: leading ids are skipped
        ]
        qs = []
        for f in filters:
            qs.append(T.KwQuery([1, 2], sort=sort, topster_size=250, filter_ids=np.unique(f)))
            qs.append(T.KwQuery([3], sort=sort, topster_size=0, filter_ids=np.unique(f)))          # topster capacity min(250, |filter|)
            qs.append(T.KwQuery([2, 1, 4], sort=sort, topster_size=20, filter_ids=np.unique(f)))
        hits = g.keyword_search_batch(qs, k_stride=250)
        assert (hits.status == 0).all()
        for i, q in enumerate(qs):
            ref = H.oracle_keyword(orc, q, ids_cap=4000)
            H.assert_hits_equal(hits, i, ref, "filter chunk=%d" % chunk)
            assert np.array_equal(g.result_ids(i), ref.result_ids)
        assert hits.n_hits.sum() > 100
    finally:
        g.set_option("kw_chunk_blocks", 0)
        g.keep_result_ids(False)


@pytest.mark.parametrize("chunk", [0, 1])
def test_keyword_filter_ids_with_excluded_ids(pair, chunk):
    """curated (excluded) ids + filter: after an excluded id the reference advances instead of skipping to the filter"""
    orc, g, _ = pair
    rng = np.random.default_rng(32)
    g.set_option("kw_chunk_blocks", chunk)
    g.keep_result_ids(True)
    try:
        all1 = H.oracle_keyword(orc, T.KwQuery([1], topster_size=250), ids_cap=4000).result_ids
        qs = []
        for trial in range(6):
            filt = np.sort(rng.choice(3000, size=int(rng.integers(5, 900)), replace=False))
            excl = np.sort(rng.choice(all1, size=int(rng.integers(1, all1.size // 2)), replace=False))
            if trial == 0:
                excl = all1[3:40]                                              # a long run of consecutive excluded hits
            qs.append(T.KwQuery([1], topster_size=250, filter_ids=filt, excluded_ids=excl))
            qs.append(T.KwQuery([1, 2], topster_size=50, filter_ids=filt, excluded_ids=excl))
        hits = g.keyword_search_batch(qs, k_stride=250)
        assert (hits.status == 0).all()
        for i, q in enumerate(qs):
            ref = H.oracle_keyword(orc, q, ids_cap=4000)
            H.assert_hits_equal(hits, i, ref, "filter+excluded chunk=%d" % chunk)
            assert np.array_equal(g.result_ids(i), ref.result_ids)
    finally:
        g.set_option("kw_chunk_blocks", 0)
        g.keep_result_ids(False)


def test_keyword_many_work_items_and_merge(pair):
    """kw_chunk_blocks=1: every 256-id driver block becomes its own work item, so the partial top-K merge kernel,
    cross-chunk id concatenation and per-chunk counters are all exercised"""
    orc, g, _ = pair
    g.set_option("kw_chunk_blocks", 1)
    g.keep_result_ids(True)
    try:
        sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
        qs = [T.KwQuery([1], sort=sort, topster_size=250), T.KwQuery([1, 2], sort=sort, topster_size=100),
              T.KwQuery([2, 1, 3], sort=sort, topster_size=7), T.KwQuery([1, 2, 3, 4], sort=sort, topster_size=250),
              T.KwQuery([1], sort=((B.SORT_SEQ_ID, 1, 0),), topster_size=250)]   # ascending ids all beat the threshold
        assert g.term_num_ids(0, 1) > 4 * 256
        hits = g.keyword_search_batch(qs, k_stride=250)
        for i, q in enumerate(qs):
            ref = H.oracle_keyword(orc, q, ids_cap=4000)
            H.assert_hits_equal(hits, i, ref, "chunks")
            assert np.array_equal(g.result_ids(i), ref.result_ids)
    finally:
        g.set_option("kw_chunk_blocks", 0)
        g.keep_result_ids(False)


def _synthetic_lists(seed):
    """posting lists with extreme length ratios so that one driver block meets runs of 1, ~10, ~40 and >64 blocks of
    the second list, plus driver ids beyond the second list's last id (window recentring, multi-round LDS merge,
    wide-run fallback and early exhaustion in kw_search_kernel stage 1)"""
    rng = np.random.default_rng(seed)
    n_docs = 4_000_000
    b_ids = np.unique(np.concatenate([rng.choice(2_000_000, size=50_000, replace=False),
                                      rng.integers(1_000_000, 1_030_000, size=9_000)])).astype(np.uint32)
    a_ids = np.unique(np.concatenate([rng.choice(b_ids, size=260, replace=False),                 # guaranteed matches
                                      rng.integers(1_000_000, 1_030_000, size=500),                # dense stretch: narrow runs
                                      rng.choice(b_ids[b_ids > 1_000_000], size=100, replace=False),
                                      rng.integers(0, 4_000_000, size=400)])).astype(np.uint32)    # sparse: wide runs + beyond B's end
    c_ids = np.unique(np.concatenate([rng.choice(a_ids, size=a_ids.size // 2, replace=False),
                                      rng.integers(0, 4_000_000, size=3_000)])).astype(np.uint32)
    lists = {}
    for term, ids in ((1, a_ids), (2, b_ids), (3, c_ids)):
        pos = (ids * np.uint32(2654435761) >> np.uint32(27)).astype(np.uint32) % 7               # one position per doc
        lists[term] = (ids, np.arange(ids.size, dtype=np.uint32), pos + 1)
    return n_docs, lists


@pytest.mark.parametrize("chunk", [256, 70])
def test_long_work_items_reload_the_driver_metadata_window(chunk):
    """kw_find2_kernel keeps the driver list's BlockIds in a lane-resident 64-block window (four v_readlane per block); a work item longer than
    64 driver blocks reloads it — 176 driver blocks as ONE work item (two reloads, the last window partial) and as items of 70 blocks
    (a reload one pair before the item ends; odd tail pairs): hits, counts and ids = the oracle, two and three tokens."""
    from oracle import oracle_py as O
    rng = np.random.default_rng(99)
    n_docs = 1_000_000
    a_ids = np.sort(rng.choice(n_docs, size=176 * 256 - 131, replace=False)).astype(np.uint32)
    b_ids = np.sort(rng.choice(n_docs, size=130_000, replace=False)).astype(np.uint32)
    c_ids = np.sort(rng.choice(n_docs, size=300_000, replace=False)).astype(np.uint32)
    pts = H.points_of(n_docs)
    orc = O.OracleIndex(1, 1)
    orc.set_num_docs(n_docs)
    orc.set_sort_dense(0, pts)
    g = T.GpuIndex(0, H.emu_lib_path())
    g.field_create(0, False)
    for term, ids in ((1, a_ids), (2, b_ids), (3, c_ids)):
        pos = (ids * np.uint32(2654435761) >> np.uint32(27)).astype(np.uint32) % 7 + 1
        oi = np.arange(ids.size, dtype=np.uint32)
        orc.load_posting(0, term, ids, oi, pos)
        g.term_upsert(0, term, ids, oi, pos)
    g.column_set(0, pts)
    g.set_num_docs(n_docs)
    g.commit()
    g.set_option("kw_chunk_blocks", chunk)
    g.keep_result_ids(True)
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    qs = [T.KwQuery([1, 2], sort=sort, topster_size=250), T.KwQuery([3, 1, 2], sort=sort, topster_size=250), T.KwQuery([1], sort=sort, topster_size=100)]
    hits = g.keyword_search_batch(qs, k_stride=250)
    assert (hits.status == 0).all() and hits.num_matched[0] > 4000 and hits.num_matched[1] > 1000
    for i, q in enumerate(qs):
        ref = H.oracle_keyword(orc, q, ids_cap=100000)
        H.assert_hits_equal(hits, i, ref, "query %d" % i)
        assert np.array_equal(g.result_ids(i), ref.result_ids)
    g.close()


@pytest.mark.parametrize("chunk,tile", [(64, 0), (1, 0), (64, 512)])
def test_stage1_block_merge_windows_fallback_and_exhaustion(chunk, tile):
    from oracle import oracle_py as O
    lib = H.emu_lib_path("-DTSGPU_KW_TILE_WORDS=%d" % tile, "_tile%d" % tile) if tile else H.emu_lib_path()   # 512-word tile: every wide run takes several rounds
    n_docs, lists = _synthetic_lists(5)
    pts = H.points_of(n_docs)
    orc = O.OracleIndex(1, 1)
    orc.set_num_docs(n_docs)
    orc.set_sort_dense(0, pts)
    g = T.GpuIndex(0, lib)
    g.field_create(0, False)
    for term, (ids, oi, off) in lists.items():
        orc.load_posting(0, term, ids, oi, off)
        g.term_upsert(0, term, ids, oi, off)
    g.column_set(0, pts)
    g.set_num_docs(n_docs)
    g.commit()
    g.set_option("kw_chunk_blocks", chunk)
    g.keep_result_ids(True)
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    qs = [T.KwQuery([1, 2], sort=sort, topster_size=250), T.KwQuery([2, 1, 3], sort=sort, topster_size=250),
          T.KwQuery([3, 2], sort=sort, topster_size=100), T.KwQuery([3, 1], sort=sort, topster_size=250)]
    hits = g.keyword_search_batch(qs, k_stride=250)
    assert (hits.status == 0).all() and hits.n_hits[0] >= 250
    for i, q in enumerate(qs):
        ref = H.oracle_keyword(orc, q, ids_cap=100000)
        H.assert_hits_equal(hits, i, ref, "stage1 chunk=%d" % chunk)
        assert np.array_equal(g.result_ids(i), ref.result_ids)
    g.close()


@pytest.fixture(scope="module")
def pair3():
    """three plain string fields over the same 2500 documents; shorter fields hold zeros (= no token) in most positions"""
    rng = np.random.default_rng(41)
    title = H.zipf_docs(2500, 120, 6, seed=11)
    body = H.zipf_docs(2500, 120, 14, seed=12)
    tags = H.zipf_docs(2500, 120, 4, seed=13)
    title[rng.random(title.shape) < 0.3] = 0
    tags[rng.random(tags.shape) < 0.6] = 0
    body[rng.random(2500) < 0.1] = 0                                         # some documents have no body at all
    orc, g = H.build_pair_fields([title, body, tags], H.emu_lib_path())
    yield orc, g
    g.close()


@pytest.mark.parametrize("chunk,pipelined", [(0, 1), (1, 1), (0, 0), (1, 0)])
def test_multi_field_union_per_token_and_field_aggregation(pair3, chunk, pipelined):
    """query_by over 2-3 fields: token = OR over the fields, query = AND over tokens (or_iterator_t); score_results2 per field
    over the tokens that field holds, folded by match_type with field weights / num_matching_fields (compute_aggregated_score).
    pipelined = 1: kw_find_mf2_kernel (three fields in the batch: its four-list instantiation); 0: kw_search_mf_kernel"""
    orc, g = pair3
    g.set_option("kw_mf_pipelined", pipelined)
    g.set_option("kw_chunk_blocks", chunk)
    g.keep_result_ids(True)
    try:
        sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
        f3, f2 = [(0, 15), (1, 7), (2, 3)], [(2, 2), (0, 9)]
        qs = []
        for toks in ([1], [2, 1], [3, 1, 2], [5, 9], [1, 2, 3, 4], [7, 1, 2, 3, 6], [119], [9999, 2], [4, 4]):
            qs.append(T.KwQuery(toks, fields=f3, sort=sort, topster_size=250))
            qs.append(T.KwQuery(toks, fields=f2, sort=sort, topster_size=30, match_type=B.MAX_WEIGHT))
            qs.append(T.KwQuery(toks, fields=f3, sort=sort, topster_size=250, match_type=B.SUM_SCORE, prioritize_token_position=True))
            qs.append(T.KwQuery(toks, fields=[(1, 1), (0, 1)], topster_size=250, prioritize_num_matching_fields=False, prioritize_exact_match=False))
        qs.append(T.KwQuery([1, 2], fields=f3, sort=sort, topster_size=250, excluded_ids=np.arange(0, 2500, 7)))
        hits = g.keyword_search_batch(qs, k_stride=250)
        assert (hits.status == 0).all()
        for i, q in enumerate(qs):
            ref = H.oracle_keyword(orc, q, ids_cap=4000)
            H.assert_hits_equal(hits, i, ref, "multi-field chunk=%d q=%s" % (chunk, q.tokens))
            assert np.array_equal(g.result_ids(i), ref.result_ids)
        assert hits.n_hits.sum() > 500
        # filter ids together with several fields (take_id() on the union iterators, src/or_iterator.cpp:218-272): hits, ids AND the
        # reference's num_keyword_matches (ids its skip-to-next-filter-id loop lands on = distinct positive filter ranks of the intersection)
        rng = np.random.default_rng(77)
        fq = []
        for toks in ([1], [2, 1], [3, 1, 2], [5, 9], [7, 1, 2, 3, 6], [119]):
            for flt in (np.sort(rng.choice(2500, size=900, replace=False)), np.arange(100, 130), np.array([2499]), np.arange(0, 2500, 2)):
                fq.append(T.KwQuery(toks, fields=f3, sort=sort, topster_size=250, filter_ids=flt))
                fq.append(T.KwQuery(toks, fields=f2, sort=sort, topster_size=25, match_type=B.SUM_SCORE, filter_ids=flt))
        h2 = g.keyword_search_batch(fq, k_stride=250)
        assert (h2.status == 0).all()
        for i, q in enumerate(fq):
            ref = H.oracle_keyword(orc, q, ids_cap=4000)
            H.assert_hits_equal(h2, i, ref, "multi-field + filter chunk=%d q=%s" % (chunk, q.tokens))
            assert np.array_equal(g.result_ids(i), ref.result_ids)
        assert h2.n_hits.sum() > 200
        # filter ids AND excluded ids AND several fields: after an excluded id the reference advances instead of skipping to the filter, so
        # num_keyword_matches is a recurrence over the intersection IN ID ORDER — walked by kw_mf_ordered_count_kernel over the per-field
        # streams of hit records (two-kernel form; the fused form reports 501 for this combination)
        xq = []
        for toks in ([1], [3, 1, 2], [7, 1, 2, 3, 6]):
            for flt in (np.sort(rng.choice(2500, size=900, replace=False)), np.arange(0, 2500, 2)):
                for exc in (np.sort(rng.choice(2500, size=1200, replace=False)), np.arange(100, 300)):
                    xq.append(T.KwQuery(toks, fields=f3, sort=sort, topster_size=250, filter_ids=flt, excluded_ids=exc))
        xq.append(T.KwQuery([1], fields=f2, filter_ids=[1, 2, 3], excluded_ids=[2]))
        h3 = g.keyword_search_batch(xq, k_stride=250)
        assert (h3.status == 0).all()
        for i, q in enumerate(xq):
            ref = H.oracle_keyword(orc, q, ids_cap=4000)
            H.assert_hits_equal(h3, i, ref, "multi-field + filter + excluded chunk=%d q=%s" % (chunk, q.tokens))
            assert np.array_equal(g.result_ids(i), ref.result_ids)
        g.set_option("kw_two_kernels", 0)
        try:
            h4 = g.keyword_search_batch(xq[:3], k_stride=250)
            assert (h4.status == B.ERR_UNSUPPORTED).all()
        finally:
            g.set_option("kw_two_kernels", 1)
    finally:
        g.set_option("kw_mf_pipelined", 1)
        g.set_option("kw_chunk_blocks", 0)
        g.keep_result_ids(False)


def test_wildcard_search_ranks_filter_ids_by_sort_keys(pair):
    """q = "*" (Index::search_wildcard, src/index.cpp:6616-6818): Topster over the filter ids (or every document) ordered by the
    sort keys; the _text_match slot is the constant 100 (sign-flipped for ASC); excluded ids are skipped; ids = the ids ranked"""
    orc, g, _ = pair
    rng = np.random.default_rng(51)
    g.keep_result_ids(True)
    try:
        col_desc = ((B.SORT_INT64_COLUMN, 1, 0), (B.SORT_SEQ_ID, -1, 0))
        sorts = [col_desc, ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, -1, 0)), ((B.SORT_SEQ_ID, 1, 0),),
                 ((B.SORT_TEXT_MATCH, -1, 0), (B.SORT_INT64_COLUMN, 1, 0), (B.SORT_SEQ_ID, 1, 0))]
        filters = [None, np.sort(rng.choice(3000, size=700, replace=False)), np.array([5]), np.arange(100, 400)]
        qs = []
        for f in filters:
            for so in sorts:
                qs.append(T.KwQuery([], sort=so, topster_size=0, filter_ids=f))
                qs.append(T.KwQuery([], sort=so, topster_size=7, filter_ids=f, excluded_ids=np.arange(0, 3000, 5)))
        hits = g.wildcard_search_batch(qs, k_stride=250)
        assert (hits.status == 0).all()
        for i, q in enumerate(qs):
            ref = H.oracle_wildcard(orc, q, ids_cap=4000)
            H.assert_hits_equal(hits, i, ref, "wildcard")
            assert np.array_equal(g.result_ids(i), ref.result_ids)
    finally:
        g.keep_result_ids(False)


def test_two_kernel_form_and_fused_kernel_agree_with_the_oracle(pair):
    """queries of <= 3 tokens run as find kernel + score kernel (hit records through memory) by default, fused when
    kw_two_kernels=0 or when the hit buffer budget would need more than two groups: same hits, counts and ids either way"""
    orc, g, _ = pair
    rng = np.random.default_rng(91)
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    qs = []
    for n_tok in (1, 2, 3, 5):
        qs += _queries(rng, 10 if n_tok <= 3 else 4, 25 if n_tok <= 3 else 10, n_tok, sort=sort, topster_size=250)
        qs += _queries(rng, 3, 25, n_tok, sort=sort, topster_size=9, excluded_ids=np.arange(0, 3000, 4))
        qs += _queries(rng, 3, 25, n_tok, sort=((B.SORT_INT64_COLUMN, -1, 0), (B.SORT_TEXT_MATCH, 1, 0), (B.SORT_SEQ_ID, 1, 0)), topster_size=40,
                       filter_ids=np.sort(rng.choice(3000, size=900, replace=False)))
    g.keep_result_ids(True)
    try:
        outs, groups = [], []
        for opts in ({"kw_two_kernels": 1}, {"kw_two_kernels": 1, "kw_hit_buffer_records": -1}, {"kw_two_kernels": 0}):
            for k, v in opts.items():
                g.set_option(k, v if v >= 0 else g.counter("kw_last_hit_records") * 3 // 5)       # a budget that needs two groups
            hits = g.keyword_search_batch(qs, k_stride=250)
            assert (hits.status == 0).all()
            groups.append(g.counter("kw_last_hit_groups"))
            ids = [g.result_ids(i) for i in range(len(qs))]
            outs.append((hits, ids))
        assert groups[0] == 2 and groups[1] >= 3 and groups[2] == 0, groups      # one group per table (<= 3 tokens / longer), split by the small budget, fused
        for i, q in enumerate(qs):
            ref = H.oracle_keyword(orc, q, ids_cap=4000)
            for hits, ids in outs:
                H.assert_hits_equal(hits, i, ref, "two-kernel/fused")
                assert np.array_equal(ids[i], ref.result_ids)
    finally:
        g.keep_result_ids(False)
        g.set_option("kw_two_kernels", 1)
        g.set_option("kw_hit_buffer_records", 0)


def _candidate_groups(rng, sort, topster_size, **kw):
    """user queries whose positions have 1-3 candidate tokens; the combinations in next_suggestion2 order, total_cost per combination"""
    groups = []
    for shape in [(2, 2), (3, 1), (1, 3, 2), (2, 2, 2), (3, 3), (1,), (2, 5)]:
        cands = [rng.choice(np.arange(1, 30), size=c, replace=False) for c in shape]
        costs = [rng.integers(0, 3, size=c) for c in shape]
        combos = []
        n = int(np.prod(shape))
        for x in range(min(n, 10)):                           # combination_limit = max(10, max_candidates)
            toks, cost, r = [], 0, x
            for pos in range(len(shape) - 1, -1, -1):         # the last position varies fastest
                toks.insert(0, int(cands[pos][r % shape[pos]])); cost += int(costs[pos][r % shape[pos]]); r //= shape[pos]
            combos.append(T.KwQuery(toks, sort=sort, topster_size=topster_size, total_cost=cost, **kw))
        groups.append(combos)
    return groups


@pytest.mark.parametrize("topster_size", [250, 12])
def test_candidate_combinations_fold_like_the_shared_topster_and_id_buff(pair, topster_size):
    """Index::search_all_candidates (src/index.cpp:1794-1894): per key the greatest KV over the passes (the later pass when the
    scores tie), top-K of those, query_index = passes before it that matched, found / ids = sorted-unique union, num_matched =
    the last pass's"""
    orc, g, _ = pair
    rng = np.random.default_rng(77)
    tm_sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    col_sort = ((B.SORT_INT64_COLUMN, 1, 0), (B.SORT_SEQ_ID, -1, 0))       # no _text_match key: passes tie on every shared key
    groups = _candidate_groups(rng, tm_sort, topster_size) + _candidate_groups(rng, col_sort, topster_size)
    groups += _candidate_groups(rng, tm_sort, topster_size, excluded_ids=np.arange(0, 3000, 3), filter_ids=np.arange(0, 3000, 2))[:3]
    same = T.KwQuery([3, 5], sort=tm_sort, topster_size=topster_size)
    groups.append([same, T.KwQuery([3, 5], sort=tm_sort, topster_size=topster_size)])            # identical passes: the later one owns the hits
    groups.append([T.KwQuery([299, 298, 297], sort=tm_sort, topster_size=topster_size), same])     # a pass without matches does not count
    groups.append([])                                                                             # no combination at all
    # a third sort key (the S2 instantiation of the sort-free fold; <= 8 passes: its LDS) in a call of its own
    s3 = ((B.SORT_INT64_COLUMN, -1, 0), (B.SORT_TEXT_MATCH, 1, 0), (B.SORT_SEQ_ID, 1, 0))
    g3 = [grp[:8] for grp in _candidate_groups(rng, s3, topster_size)[:3]] + [[T.KwQuery([1 + j, 2], sort=s3, topster_size=topster_size) for j in range(2, 10)]]
    n0 = g.counter("kw_candidates_rank_launches")
    h3, q3, f3 = g.keyword_search_candidates_batch(g3, k_stride=250)
    assert g.counter("kw_candidates_rank_launches") > n0
    for gi, combos in enumerate(g3):
        ref, ref_qi = H.oracle_candidates(orc, combos, ids_cap=4000)
        H.assert_hits_equal(h3, gi, ref, "candidates, three sort keys g%d" % gi)
        assert np.array_equal(q3[gi, :int(h3.n_hits[gi])], ref_qi) and int(f3[gi]) == int(ref.n_result_ids)
    # the sort-free fold (kw_candidates_rank_kernel, default) and the two-sort kernel give the same arrays
    g.set_option("kw_candidates_rank_fold", 0)
    h0, q0, f0 = g.keyword_search_candidates_batch(groups, k_stride=250)
    g.set_option("kw_candidates_rank_fold", 1)
    n0 = g.counter("kw_candidates_rank_launches")
    hits, qidx, found = g.keyword_search_candidates_batch(groups, k_stride=250)
    assert g.counter("kw_candidates_rank_launches") > n0
    assert (hits.status == 0).all()
    assert np.array_equal(hits.n_hits, h0.n_hits) and np.array_equal(found, f0) and np.array_equal(hits.num_matched, h0.num_matched)
    for gi in range(len(groups)):
        n = int(hits.n_hits[gi])
        for name in ("keys", "scores", "text_match", "match_score_index"):
            assert np.array_equal(getattr(hits, name)[gi, :n], getattr(h0, name)[gi, :n]), (name, gi)
        assert np.array_equal(qidx[gi, :n], q0[gi, :n]), gi
    multi = 0
    for gi, combos in enumerate(groups):
        if not combos:
            assert hits.n_hits[gi] == 0 and found[gi] == 0 and hits.num_matched[gi] == 0
            continue
        ref, ref_qi = H.oracle_candidates(orc, combos, ids_cap=4000)
        H.assert_hits_equal(hits, gi, ref, "candidates g%d" % gi)
        n = int(hits.n_hits[gi])
        assert np.array_equal(qidx[gi, :n], ref_qi), "g%d query_index" % gi
        assert int(found[gi]) == int(ref.n_result_ids)
        assert np.array_equal(g.candidates_result_ids(gi), ref.result_ids)
        multi += int(len(set(ref_qi.tolist())) > 1)
    assert multi >= 5                  # hits really come from different passes
    # plain passes planned ON THE DEVICE (big candidate batches: kw_plan.hip.h lays the id arena out itself, the id-set marks read the device tables)
    plain = [grp for grp in groups[:14] if grp and all(c.filter_ids is None and c.excluded_ids is None for c in grp)]
    g.set_option("kw_device_plan_min_queries", 1)
    n0 = g.counter("kw_device_plans")
    hd, qd, fd = g.keyword_search_candidates_batch(plain, k_stride=250)
    planned = g.counter("kw_device_plans") - n0
    ids_d = [g.candidates_result_ids(gi) for gi in range(len(plain))]
    g.set_option("kw_device_plan_min_queries", 512)
    assert planned == 1
    for gi, combos in enumerate(plain):
        ref, ref_qi = H.oracle_candidates(orc, combos, ids_cap=4000)
        H.assert_hits_equal(hd, gi, ref, "candidates, device plan g%d" % gi)
        assert np.array_equal(qd[gi, :int(hd.n_hits[gi])], ref_qi) and int(fd[gi]) == int(ref.n_result_ids)
        assert np.array_equal(ids_d[gi], ref.result_ids)
    assert (qidx[len(groups) - 3, :int(hits.n_hits[len(groups) - 3])] == 1).all()
    assert (qidx[len(groups) - 2, :int(hits.n_hits[len(groups) - 2])] == 0).all()


def _array_docs(n_docs, vocab, seed):
    """string[] documents: 0-4 elements of 1-5 tokens each (repeats inside and across elements, single-token elements)"""
    rng = np.random.default_rng(seed)
    p = 1.0 / np.arange(1, vocab + 1)
    p /= p.sum()
    docs = []
    for _ in range(n_docs):
        elems = []
        for _e in range(int(rng.integers(0, 5))):
            elems.append(list((rng.choice(vocab, size=int(rng.integers(1, 6)), p=p) + 1).astype(int)))
        docs.append(elems)
    return docs


def make_pair_arr(lib_path):
    """field 0 = string[] ("tags"), field 1 = plain string ("title") over the same documents; lib_path is EXPLICIT: the GPU tier
    passes H.gpu_lib_path() (a module-scoped fixture is set up before any function-scoped monkeypatch of H.emu_lib_path)"""
    from oracle import oracle_py as O
    n_docs = 2000
    arr = _array_docs(n_docs, 40, seed=61)
    title = H.zipf_docs(n_docs, 40, 5, seed=62)
    orc = O.OracleIndex(2, 1)
    for d in range(n_docs):
        if arr[d]:
            orc.index_array(d, 0, arr[d])
        orc.index_plain(d, 1, title[d])
    pts = H.points_of(n_docs)
    orc.set_num_docs(n_docs)
    orc.set_sort_dense(0, pts)
    g = T.GpuIndex(0, lib_path)
    g.field_create(0, True)
    g.field_create(1, False)
    for f in (0, 1):
        for term in orc.terms(f):
            ids, oi, off = orc.dump_posting(f, int(term))
            g.term_upsert(f, int(term), ids, oi, off)
    g.column_set(0, pts)
    g.set_num_docs(n_docs)
    g.commit()
    return orc, g


@pytest.fixture(scope="module")
def pair_arr():
    orc, g = make_pair_arr(H.emu_lib_path())
    yield orc, g
    g.close()


def test_string_array_fields_match_per_element_and_mix_with_plain_fields(pair_arr):
    """string[] offset format (src/index.cpp:1351-1395): Match runs per array element over the tokens present in it, the best
    element wins, unique_words = words_present; single-token verbatim / last-offset readers; alone and next to a plain field"""
    orc, g = pair_arr
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    qs = []
    for toks in ([1], [2], [7], [1, 2], [2, 1], [1, 2, 3], [3, 1], [4, 2, 1, 3], [5, 5], [2, 9, 1, 4, 3, 6], [39], [1, 9999]):
        qs.append(T.KwQuery(toks, fields=[(0, 15)], sort=sort, topster_size=250))
        qs.append(T.KwQuery(toks, fields=[(0, 15)], sort=sort, topster_size=250, prioritize_token_position=True, prioritize_exact_match=False))
        qs.append(T.KwQuery(toks, fields=[(0, 4), (1, 9)], sort=sort, topster_size=250))
        qs.append(T.KwQuery(toks, fields=[(1, 2), (0, 6)], sort=sort, topster_size=40, match_type=B.SUM_SCORE, prioritize_token_position=True))
    hits = g.keyword_search_batch(qs, k_stride=250)
    assert (hits.status == 0).all()
    for i, q in enumerate(qs):
        H.assert_hits_equal(hits, i, H.oracle_keyword(orc, q), "array q=%s fields=%s" % (q.tokens, q.fields))
    assert hits.n_hits.sum() > 2000


def test_device_shard_merge_equals_the_torch_merge():
    """kw_shard_merge_kernel through tsgpu_merge_shard_hits_device (array form of the shard merge) vs a numpy lexsort reference"""
    import ctypes as C
    rng = np.random.default_rng(9)
    G, Bq, K, k = 4, 7, 60, 50
    keys = np.zeros((G, Bq, K), np.int64); scores = np.zeros((G, Bq, K, 3), np.int64); n_hits = np.zeros((G, Bq), np.int32)
    num = rng.integers(0, 1000, size=(G, Bq)).astype(np.int64)
    for g_ in range(G):
        for q in range(Bq):
            n = int(rng.integers(0, K + 1))
            if q == 0:
                n = 0 if g_ else K                                     # one shard holds everything / others nothing
            ks = rng.choice(np.arange(g_ * 100000, (g_ + 1) * 100000), size=n, replace=False)        # shard-unique keys
            sc = np.stack([rng.integers(0, 4, n), rng.integers(-3, 3, n), rng.integers(0, 2, n)], 1).astype(np.int64)   # heavy ties
            order = sorted(range(n), key=lambda i: (sc[i, 0], sc[i, 1], sc[i, 2], ks[i]), reverse=True)
            keys[g_, q, :n] = ks[order]; scores[g_, q, :n] = sc[order]; n_hits[g_, q] = n
    ref = H.reference_shard_merge(keys, scores, n_hits, k)
    g = T.GpuIndex(0, H.emu_lib_path())
    ok_ = np.zeros((Bq, k), np.int64); os_ = np.zeros((Bq, k, 3), np.int64); on = np.zeros(Bq, np.int32); onm = np.zeros(Bq, np.int64)
    hin, hout = B.HitsC(), B.HitsC()
    hin.mem = hout.mem = B.MEM_DEVICE                                   # the emulator's "device" memory is host memory
    hin.k_stride, hout.k_stride = K, k
    for name, a, o in (("keys", keys, ok_), ("scores", scores, os_), ("n_hits", n_hits, on), ("num_matched", num, onm)):
        setattr(hin, name, a.ctypes.data); setattr(hout, name, o.ctypes.data)
    B.check(g.L, g.L.tsgpu_merge_shard_hits_device(g.h, C.byref(hin), G, Bq, k, C.byref(hout)))
    for q in range(Bq):
        n = ref[q][0].size
        assert on[q] == n and np.array_equal(ok_[q, :n], ref[q][0]) and np.array_equal(os_[q, :n], ref[q][1])
        assert onm[q] == num[:, q].sum()
    g.close()


def test_edge_cases_empty_index_missing_tokens_and_degenerate_topsters():
    """empty / ragged inputs: index without any posting, queries whose tokens are all absent, zero queries, Topster of 1,
    every match excluded, a 1-document collection"""
    lib = H.emu_lib_path()
    g = T.GpuIndex(0, lib)
    g.field_create(0, False)
    g.set_num_docs(0)
    g.commit()
    hits = g.keyword_search_batch([T.KwQuery([1, 2]), T.KwQuery([7])], k_stride=250)
    assert (hits.status == 0).all() and (hits.n_hits == 0).all() and (hits.num_matched == 0).all()
    assert g.keyword_search_batch([], k_stride=250).n_hits.size == 0
    w = g.wildcard_search_batch([T.KwQuery([], sort=((B.SORT_SEQ_ID, 1, 0),))], k_stride=250)
    assert w.status[0] == 0 and w.n_hits[0] == 0
    g.close()
    docs = np.array([[5, 6, 5]], np.uint32)                                   # one document, a repeated token
    orc, g = H.build_pair(docs, lib)
    qs = [T.KwQuery([5]), T.KwQuery([5, 6]), T.KwQuery([6, 5, 6]), T.KwQuery([5, 9]), T.KwQuery([9, 9]), T.KwQuery([5], topster_size=1),
          T.KwQuery([5], excluded_ids=[0]), T.KwQuery([5], filter_ids=[0]), T.KwQuery([5], filter_ids=[3])]
    hits = g.keyword_search_batch(qs, k_stride=250)
    assert (hits.status == 0).all()
    for i, q in enumerate(qs):
        H.assert_hits_equal(hits, i, H.oracle_keyword(orc, q), "edge q%d" % i)
    assert hits.n_hits[4] == 0 and hits.n_hits[6] == 0 and hits.n_hits[8] == 0 and hits.n_hits[7] == 1
    g.close()


@pytest.mark.parametrize("select_min", [2, 17, 0, 3])
def test_two_level_merge_many_work_items(select_min):
    """a query cut into more than 16 work items: merged by selection (kw_select_partials: prefix union -> threshold -> candidates -> sort;
    select_min = 3: also the queries with few lists) or, kw_merge_select_min = 0, folded in groups of 8 (kw_merge_groups_kernel), then per query"""
    docs = H.zipf_docs(6200, 3, 5, seed=12, s=0.2)
    orc, g = H.build_pair(docs, H.emu_lib_path())
    g.set_option("kw_merge_select_min", select_min)
    g.set_option("kw_chunk_blocks", 1)
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    # Topster sizes around the selecting merge's regimes: prefix union / candidates within half the LDS buffer (tree of rank merges),
    # beyond it (2k > 512: bitonic sort), and k close to the whole result (few entries: everything gathered)
    qs = [T.KwQuery([1], sort=sort, topster_size=40), T.KwQuery([2, 1], sort=sort, topster_size=250), T.KwQuery([3, 1, 2], sort=sort, topster_size=7),
          T.KwQuery([1, 2], sort=sort, topster_size=250, filter_ids=np.arange(0, 6200, 3, dtype=np.uint32)),
          T.KwQuery([1], sort=sort, topster_size=400), T.KwQuery([2, 1], sort=sort, topster_size=510), T.KwQuery([1], sort=sort, topster_size=300),
          T.KwQuery([1, 2], sort=sort, topster_size=500, filter_ids=np.arange(0, 6200, 9, dtype=np.uint32))]
    assert g.term_num_ids(0, 1) > 17 * 256
    hits = g.keyword_search_batch(qs, k_stride=512)
    assert (hits.status == 0).all()
    for i, q in enumerate(qs):
        H.assert_hits_equal(hits, i, H.oracle_keyword(orc, q), "two-level merge")
    g.close()


def test_deadline_in_flight_returns_partial_hits_with_search_cutoff(pair):
    """search_cutoff (include/or_iterator.h:148-153): a query that runs out of time ON THE DEVICE keeps what it found — status 0, partial
    hits that are a subset of the full result with the same scores, search_cutoff = 1; a query already late at planning time gets 408"""
    import time
    orc, g, _ = pair
    g.set_option("kw_chunk_blocks", 1)               # many work items, each checks the clock
    try:
        sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_SEQ_ID, 1, 0))
        toks = [[1], [2, 1], [3, 1, 2], [1, 4]]
        full = g.keyword_search_batch([T.KwQuery(t, sort=sort, topster_size=250) for t in toks], k_stride=250)
        assert (full.search_cutoff == 0).all()
        now = int(time.time() * 1e6)
        # the emulator needs seconds for this batch: a budget of 2 ms is over before most work items start
        qs = [T.KwQuery(t, sort=sort, topster_size=250, deadline_us=now + 2000) for t in toks] + [T.KwQuery([1], sort=sort, deadline_us=now - 5)]
        hits = g.keyword_search_batch(qs, k_stride=250)
        assert hits.status[4] == B.ERR_DEADLINE and hits.search_cutoff[4] == 1 and hits.n_hits[4] == 0
        assert (hits.status[:4] == 0).all() and hits.search_cutoff[:4].sum() >= 1
        for i in range(4):
            n, nf = int(hits.n_hits[i]), int(full.n_hits[i])
            got = {int(k): tuple(int(x) for x in s) for k, s in zip(hits.keys[i, :n], hits.scores[i, :n])}
            ref = {int(k): tuple(int(x) for x in s) for k, s in zip(full.keys[i, :nf], full.scores[i, :nf])}
            if not hits.search_cutoff[i]:
                assert got == ref
            else:
                assert n <= nf or nf == 250
                if nf < 250:                        # the full Topster held every match: a partial result can only be a subset of it
                    assert all(k in ref and ref[k] == v for k, v in got.items())
                assert int(hits.num_matched[i]) <= int(full.num_matched[i])
        # a generous deadline changes nothing
        late = g.keyword_search_batch([T.KwQuery(t, sort=sort, topster_size=250, deadline_us=now + 3_600_000_000) for t in toks], k_stride=250)
        assert (late.search_cutoff == 0).all() and np.array_equal(late.keys, full.keys) and np.array_equal(late.n_hits, full.n_hits)
    finally:
        g.set_option("kw_chunk_blocks", 0)


@pytest.mark.parametrize("chunk", [64, 3])
def test_pair_find_kernel_matches_the_oracle(pair, chunk):
    """kw_pair_blocks=1: the find kernel that serves two driver blocks per iteration (kw_find2.hip.h) — same hit records as the
    one-block kernel: odd block counts (a lone last block), wide / multi-round / exhausted runs, third-list probes (1..7 tokens: both
    tables), filters"""
    from oracle import oracle_py as O
    orc, g, _ = pair
    rng = np.random.default_rng(123)
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    qs = []
    for n_tok in (1, 2, 3, 4, 5, 7):
        qs += _queries(rng, 8, 25, n_tok, sort=sort, topster_size=250)
        qs += _queries(rng, 3, 25, n_tok, sort=sort, topster_size=9, excluded_ids=np.arange(0, 3000, 4))
        qs += _queries(rng, 3, 25, n_tok, sort=sort, topster_size=40, filter_ids=np.sort(rng.choice(3000, size=900, replace=False)))
    g.set_option("kw_pair_blocks", 1)
    g.set_option("kw_chunk_blocks", chunk)
    g.keep_result_ids(True)
    try:
        hits = g.keyword_search_batch(qs, k_stride=250)
        assert (hits.status == 0).all()
        for i, q in enumerate(qs):
            ref = H.oracle_keyword(orc, q, ids_cap=4000)
            H.assert_hits_equal(hits, i, ref, "pair kernel chunk=%d" % chunk)
            assert np.array_equal(g.result_ids(i), ref.result_ids)
    finally:
        g.set_option("kw_pair_blocks", 1)
        g.set_option("kw_chunk_blocks", 0)
        g.keep_result_ids(False)
    # extreme length ratios
    n_docs, lists = _synthetic_lists(6)
    pts = H.points_of(n_docs)
    orc2 = O.OracleIndex(1, 1)
    orc2.set_num_docs(n_docs)
    orc2.set_sort_dense(0, pts)
    g2 = T.GpuIndex(0, H.emu_lib_path())
    g2.field_create(0, False)
    for term, (ids, oi, off) in lists.items():
        orc2.load_posting(0, term, ids, oi, off)
        g2.term_upsert(0, term, ids, oi, off)
    g2.column_set(0, pts)
    g2.set_num_docs(n_docs)
    g2.commit()
    g2.set_option("kw_pair_blocks", 1)
    g2.set_option("kw_chunk_blocks", chunk)
    g2.keep_result_ids(True)
    qs = [T.KwQuery([1, 2], sort=sort, topster_size=250), T.KwQuery([2, 1, 3], sort=sort, topster_size=250),
          T.KwQuery([3, 2], sort=sort, topster_size=100), T.KwQuery([3, 1], sort=sort, topster_size=250), T.KwQuery([2], sort=sort, topster_size=250)]
    hits = g2.keyword_search_batch(qs, k_stride=250)
    assert (hits.status == 0).all() and hits.n_hits[0] >= 250
    for i, q in enumerate(qs):
        ref = H.oracle_keyword(orc2, q, ids_cap=100000)
        H.assert_hits_equal(hits, i, ref, "pair kernel stage1 chunk=%d" % chunk)
        assert np.array_equal(g2.result_ids(i), ref.result_ids)
    g2.close()


def test_find_kernel_counts_the_bytes_it_requests_without_changing_the_results(pair):
    """option kw_count_touched: the COUNT instantiation of kw_find2_kernel (bench.py `roofline.touched_bytes_per_launch`) — same hits,
    counters that add up and respond to the work: a batch repeated twice requests twice the bytes; every driver id is requested once
    (2 or 4 bytes each); the lists' footprint hook agrees with the ids the oracle holds"""
    orc, g, docs = pair
    rng = np.random.default_rng(77)
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    qs = _queries(rng, 12, 25, 3, sort=sort, topster_size=250) + _queries(rng, 6, 25, 2, sort=sort, topster_size=250) + _queries(rng, 4, 40, 4, sort=sort, topster_size=250)
    g.set_option("kw_pair_blocks", 1)
    base = g.keyword_search_batch(qs, k_stride=250)
    g.set_option("kw_count_touched", 1)
    try:
        hits = g.keyword_search_batch(qs, k_stride=250)
        t1 = g.kw_touched()
        hits2 = g.keyword_search_batch(qs + qs, k_stride=250)
        t2 = g.kw_touched()
    finally:
        g.set_option("kw_count_touched", 0)
        g.set_option("kw_pair_blocks", 1)
    for name in ("keys", "scores", "n_hits", "num_matched", "status"):
        assert np.array_equal(getattr(hits, name), getattr(base, name)), name
        assert np.array_equal(getattr(hits2, name)[:len(qs)], getattr(base, name)), name
    parts = ("find_driver_ids", "find_metadata", "find_tile_dma", "find_probes", "find_records")
    assert t1["find_requested_bytes"] == sum(t1[p] for p in parts) > 0
    assert all(t1[p] > 0 for p in parts)
    for p in parts + ("find_work_items", "find_hit_records", "score_requested_bytes"):
        assert t2[p] == 2 * t1[p], p
    # every id of every query's shortest list is requested exactly once, 2 or 4 bytes each
    df = {}

    def n_docs_of(t):
        if t not in df:
            df[t] = int((docs == t).any(axis=1).sum())
        return df[t]
    drv = sum(min(n_docs_of(int(t)) for t in q.tokens) for q in qs)
    assert 2 * drv <= t1["find_driver_ids"] <= 4 * drv
    assert t1["find_records"] == t1["find_hit_records"] * 4 * (3 + 1) or t1["find_records"] >= t1["find_hit_records"] * 16       # (TMAX 3 and TMAX 10 tables)
    assert t1["find_hit_records"] == int(base.num_matched.sum())
    assert t1["score_requested_bytes"] > 0
    terms = np.unique(np.concatenate([np.asarray(q.tokens) for q in qs])).astype(np.uint32)
    fp = g.kw_lists_footprint(np.zeros(terms.size, np.uint32), terms)
    assert fp["n_lists"] == terms.size
    assert fp["n_ids"] == sum(n_docs_of(int(t)) for t in terms)
    assert 2 * fp["n_ids"] <= fp["ids_bytes"] <= 4 * fp["n_ids"] + 8 * (fp["n_ids"] // 256 + terms.size) and fp["payload_bytes"] > 0 and fp["block_metadata_bytes"] > 0


@pytest.mark.parametrize("chunk", [0, 1])
def test_dropped_tokens_are_scored_when_present_and_never_required(pair, pair3, chunk):
    """the drop_tokens passes of the reference call search_across_fields with the tokens it left out of the AND (`dropped_tokens`,
    src/index.cpp:5427-5464): compute_aggregated_score positions their lists on every hit and scores the ones the document holds after
    the query's own tokens (:5271-5290; query_len counts them). Same hit SETS as without them, different scores / order."""
    orc, g, _ = pair
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    g.set_option("kw_chunk_blocks", chunk)
    g.keep_result_ids(True)
    try:
        cases = [([1], [2]), ([2, 1], [3]), ([3], [1, 2]), ([1, 2, 3], [4, 5]), ([5], [9999]), ([4, 2], [4]), ([7], [1, 2, 3, 4]), ([6, 1, 2], [3])]
        qs = [T.KwQuery(req, sort=sort, topster_size=250, dropped_tokens=dr) for req, dr in cases]
        qs += [T.KwQuery(req, sort=sort, topster_size=40, dropped_tokens=dr, prioritize_token_position=True, match_type=B.SUM_SCORE) for req, dr in cases[:4]]
        qs += [T.KwQuery([1, 2], sort=sort, topster_size=250, dropped_tokens=[3], filter_ids=np.arange(0, 3000, 3)),
               T.KwQuery([1], sort=sort, topster_size=250, dropped_tokens=[2, 3], excluded_ids=np.arange(0, 3000, 5))]
        plain = g.keyword_search_batch([T.KwQuery(q.tokens, sort=sort, topster_size=250) for q in qs[:8]], k_stride=250)
        hits = g.keyword_search_batch(qs, k_stride=250)
        assert (hits.status == 0).all()
        changed = 0
        for i, q in enumerate(qs):
            ref = H.oracle_keyword(orc, q, ids_cap=4000)
            H.assert_hits_equal(hits, i, ref, "dropped tokens chunk=%d q=%s+%s" % (chunk, q.tokens, q.dropped_tokens))
            assert np.array_equal(g.result_ids(i), ref.result_ids)
            if i < 8:
                assert hits.num_matched[i] == plain.num_matched[i]                         # the AND is the query's own tokens only
                n = int(hits.n_hits[i])
                changed += not np.array_equal(hits.scores[i, :n], plain.scores[i, :n])
        assert changed >= 5
        # too many: 501
        bad = g.keyword_search_batch([T.KwQuery(list(range(1, 9)), dropped_tokens=[9, 10, 11])], k_stride=250)
        assert bad.status[0] == B.ERR_UNSUPPORTED
    finally:
        g.set_option("kw_chunk_blocks", 0)
        g.keep_result_ids(False)
    # several query_by fields: the dropped token may sit in another field than the query's tokens
    orc3, g3 = pair3
    f3 = [(0, 15), (1, 7), (2, 3)]
    qs = [T.KwQuery(req, fields=f3, sort=sort, topster_size=250, dropped_tokens=dr) for req, dr in (([1], [2]), ([2, 1], [3, 4]), ([5], [1]), ([3, 1, 2], [9]))]
    qs += [T.KwQuery([1, 2], fields=f3, sort=sort, topster_size=30, dropped_tokens=[3], match_type=B.MAX_WEIGHT, filter_ids=np.arange(0, 2500, 2))]
    h3 = g3.keyword_search_batch(qs, k_stride=250)
    assert (h3.status == 0).all()
    for i, q in enumerate(qs):
        H.assert_hits_equal(h3, i, H.oracle_keyword(orc3, q, ids_cap=4000), "dropped tokens, 3 fields q=%s+%s" % (q.tokens, q.dropped_tokens))


def test_synonym_passes_score_like_score_results2(pair, pair3):
    """a synonym's expansion searched in place of the user's phrase (is_synonym_query, syn_orig_num_tokens, orig_num_tokens,
    demote_synonym_match): query_len = syn_orig_num_tokens (src/index.cpp:5292-5294), the single-token fast path's words_present /
    distance (:6989-6994), the synonym bit, words replaced by the phrase's length when every token matched and every component
    rescaled by orig / syn tokens (:7024-7060)"""
    orc, g, _ = pair
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    flags = [dict(is_synonym_query=True, syn_orig_num_tokens=2, orig_num_tokens=1), dict(is_synonym_query=True, syn_orig_num_tokens=1, orig_num_tokens=3),
             dict(is_synonym_query=True, syn_orig_num_tokens=3, orig_num_tokens=2, demote_synonym_match=True), dict(is_synonym_query=True, syn_orig_num_tokens=4, orig_num_tokens=4),
             dict(is_synonym_query=False, syn_orig_num_tokens=2, orig_num_tokens=2), dict(is_synonym_query=True, syn_orig_num_tokens=-1, orig_num_tokens=2)]
    qs = []
    for toks in ([1], [2, 1], [3, 1, 2], [1, 2, 3, 4]):
        for fl in flags:
            qs.append(T.KwQuery(toks, sort=sort, topster_size=250, **fl))
            qs.append(T.KwQuery(toks, sort=sort, topster_size=60, prioritize_token_position=True, match_type=B.SUM_SCORE, **fl))
    plain = g.keyword_search_batch([T.KwQuery([2, 1], sort=sort, topster_size=250)], k_stride=250)
    hits = g.keyword_search_batch(qs, k_stride=250)
    assert (hits.status == 0).all()
    for i, q in enumerate(qs):
        H.assert_hits_equal(hits, i, H.oracle_keyword(orc, q), "synonym pass q=%s %s" % (q.tokens, (q.is_synonym_query, q.syn_orig_num_tokens, q.orig_num_tokens, q.demote_synonym_match)))
    n = int(plain.n_hits[0])
    assert not np.array_equal(hits.scores[12, :n], plain.scores[0, :n])            # [2, 1] as a synonym pass scores differently
    orc3, g3 = pair3
    f3 = [(0, 15), (1, 7), (2, 3)]
    q3 = [T.KwQuery(toks, fields=f3, sort=sort, topster_size=250, **fl) for toks in ([1], [2, 1], [3, 1, 2]) for fl in flags[:4]]
    q3 += [T.KwQuery([2, 1], fields=f3, sort=sort, topster_size=250, dropped_tokens=[3], **flags[0])]
    h3 = g3.keyword_search_batch(q3, k_stride=250)
    assert (h3.status == 0).all()
    for i, q in enumerate(q3):
        H.assert_hits_equal(h3, i, H.oracle_keyword(orc3, q), "synonym pass, 3 fields q=%s" % q.tokens)


def test_parallel_planning_of_a_batch_gives_the_serial_plan(pair, pair3):
    """big batches are planned in slices on the context's parked host threads (plan_threads / plan_parallel_min_queries); the slices'
    arenas (excluded / filter ids, multi-field descriptors, id segments, rank bitmaps, work items) are concatenated and every query's
    offsets shifted: results, counts and matched ids must equal the serial plan's for a batch that mixes every kind of query"""
    orc, g, _ = pair
    rng = np.random.default_rng(321)
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    qs = []
    for rep in range(40):
        toks = rng.choice(np.arange(1, 30), size=int(rng.integers(1, 5)), replace=False)
        kind = rep % 5
        if kind == 0: qs.append(T.KwQuery(toks, sort=sort, topster_size=250))
        elif kind == 1: qs.append(T.KwQuery(toks, sort=sort, topster_size=40, filter_ids=np.sort(rng.choice(3000, size=700, replace=False))))
        elif kind == 2: qs.append(T.KwQuery(toks, sort=sort, topster_size=250, excluded_ids=np.arange(int(rng.integers(0, 5)), 3000, 5)))
        elif kind == 3: qs.append(T.KwQuery(toks[:2], sort=sort, topster_size=250, dropped_tokens=[int(rng.integers(30, 60))]))
        else: qs.append(T.KwQuery(list(range(1, 13)), sort=sort) if rep == 4 else T.KwQuery(toks, sort=((B.SORT_SEQ_ID, 1, 0),), topster_size=17))
    g.keep_result_ids(True)
    try:
        serial = g.keyword_search_batch(qs, k_stride=250)
        ids_s = [g.result_ids(i) if serial.status[i] == 0 else None for i in range(len(qs))]
        g.set_option("plan_parallel_min_queries", 1)
        g.set_option("plan_threads", 5)
        par = g.keyword_search_batch(qs, k_stride=250)
        assert np.array_equal(par.status, serial.status) and (serial.status == 0).sum() >= 38
        for i in range(len(qs)):
            if serial.status[i] != 0:
                continue
            n = int(serial.n_hits[i])
            assert par.n_hits[i] == n and par.num_matched[i] == serial.num_matched[i]
            assert np.array_equal(par.keys[i, :n], serial.keys[i, :n]) and np.array_equal(par.scores[i, :n], serial.scores[i, :n])
            assert np.array_equal(g.result_ids(i), ids_s[i])
    finally:
        g.set_option("plan_parallel_min_queries", 2048)
        g.set_option("plan_threads", 8)
        g.keep_result_ids(False)
    orc3, g3 = pair3
    f3 = [(0, 15), (1, 7), (2, 3)]
    q3 = []
    for rep in range(24):
        toks = rng.choice(np.arange(1, 20), size=int(rng.integers(1, 4)), replace=False)
        q3.append(T.KwQuery(toks, fields=f3, sort=sort, topster_size=250, filter_ids=np.sort(rng.choice(2500, size=600, replace=False)) if rep % 3 == 0 else None))
    s3 = g3.keyword_search_batch(q3, k_stride=250)
    g3.set_option("plan_parallel_min_queries", 1)
    try:
        p3 = g3.keyword_search_batch(q3, k_stride=250)
    finally:
        g3.set_option("plan_parallel_min_queries", 2048)
    for i in range(len(q3)):
        n = int(s3.n_hits[i])
        assert p3.n_hits[i] == n and p3.num_matched[i] == s3.num_matched[i] and np.array_equal(p3.keys[i, :n], s3.keys[i, :n]) and np.array_equal(p3.scores[i, :n], s3.scores[i, :n])
        H.assert_hits_equal(p3, i, H.oracle_keyword(orc3, q3[i]), "parallel plan, 3 fields")


def test_host_output_batch_served_in_slices_equals_the_single_batch(pair):
    """a large batch with host output is served in slices, each on a lane and host thread of its own, enqueued in slice order (kw_split_host:
    slice i's copies run while slice i + 1 computes); status codes, hits, counts and cut-off flags must equal the unsliced batch's, a failing query fails alone"""
    orc, g, _ = pair
    rng = np.random.default_rng(99)
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    qs = []
    for rep in range(53):
        toks = rng.choice(np.arange(1, 30), size=int(rng.integers(1, 4)), replace=False)
        if rep % 4 == 1: qs.append(T.KwQuery(toks, sort=sort, topster_size=40, filter_ids=np.sort(rng.choice(3000, size=500, replace=False))))
        elif rep == 30: qs.append(T.KwQuery(list(range(1, 13)), sort=sort))            # too many tokens: fails alone
        else: qs.append(T.KwQuery(toks, sort=sort, topster_size=250 if rep % 3 else 17))
    g.set_option("kw_host_split_queries", 0)
    whole = g.keyword_search_batch(qs, k_stride=250)
    try:
        g.set_option("kw_host_split_first_pct", 50)
        g.set_option("kw_host_split_queries", 7)                                       # 26 queries + the other 27 (default: one tail slice)
        r0 = g.counter("kw_batches")
        two = g.keyword_search_batch(qs, k_stride=250)
        assert g.counter("kw_batches") - r0 == 2
        g.set_option("kw_host_split_tail_slices", 2)                                   # 26 queries, then two slices of the other 27
        r0 = g.counter("kw_batches")
        cut = g.keyword_search_batch(qs, k_stride=250)
        assert g.counter("kw_batches") - r0 == 3
        assert np.array_equal(two.status, cut.status) and np.array_equal(two.n_hits, cut.n_hits) and np.array_equal(two.num_matched, cut.num_matched)
        for i in range(len(qs)):
            n = int(cut.n_hits[i])
            assert np.array_equal(two.keys[i, :n], cut.keys[i, :n]) and np.array_equal(two.scores[i, :n], cut.scores[i, :n])
        assert np.array_equal(cut.status, whole.status) and (whole.status != 0).sum() == 1
        assert np.array_equal(cut.search_cutoff, whole.search_cutoff)
        for i in range(len(qs)):
            n = int(whole.n_hits[i])
            assert cut.n_hits[i] == n and cut.num_matched[i] == whole.num_matched[i]
            assert np.array_equal(cut.keys[i, :n], whole.keys[i, :n]) and np.array_equal(cut.scores[i, :n], whole.scores[i, :n])
            assert np.array_equal(cut.text_match[i, :n], whole.text_match[i, :n])
            if whole.status[i] == 0:
                H.assert_hits_equal(cut, i, H.oracle_keyword(orc, qs[i]), "sliced host batch")
    finally:
        g.set_option("kw_host_split_queries", 1000)
        g.set_option("kw_host_split_first_pct", 85)
        g.set_option("kw_host_split_tail_slices", 1)


@pytest.mark.parametrize("chunk_opt,host_threads", [(0, 1), (4, 3)])
def test_device_side_planner_equals_the_host_planner_and_the_oracle(pair, chunk_opt, host_threads):
    """kw_plan.hip.h: a batch of plain single-field queries is planned by three kernels (term table -> handles, chunk rule + cost key,
    rank layout + work items + hit offsets) instead of plan_batch(): same Topsters, counts and statuses as the host plan and the oracle —
    1..7 tokens (both kernel tables), absent tokens, duplicates, every sort-key form, small Topsters; device AND host outputs, the sliced
    host delivery; a batch with one query of another shape (filter ids) is handed back to the host planner whole."""
    orc, g, _ = pair
    rng = np.random.default_rng(77)
    sorts = [((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0)), ((B.SORT_INT64_COLUMN, -1, 0), (B.SORT_TEXT_MATCH, 1, 0), (B.SORT_SEQ_ID, -1, 0)), ((B.SORT_SEQ_ID, 1, 0),)]
    qs = []
    for rep in range(90):
        n_tok = int(rng.choice([1, 2, 3, 3, 3, 4, 5, 7]))
        toks = list(rng.choice(np.arange(1, 60), size=n_tok, replace=False))
        if rep % 11 == 0: toks[0] = 100000 + rep            # a token the index does not hold (skipped; alone: no hits)
        if rep % 13 == 0 and n_tok >= 2: toks[1] = toks[0]  # a duplicated token
        if rep % 17 == 0: toks = [3000000 + rep]            # no token of the query exists (ids from 4M up live in the host map only: those batches go back to the host planner)
        qs.append(T.KwQuery(toks, sort=sorts[rep % 3], topster_size=[250, 40, 7][rep % 3], match_type=rep % 3, prioritize_token_position=bool(rep & 1), total_cost=rep % 4))
    try:
        g.set_option("kw_device_plan_min_queries", 0)
        host = g.keyword_search_batch(qs, k_stride=250)
        g.set_option("kw_device_plan_min_queries", 8)
        g.set_option("kw_chunk_blocks", chunk_opt)
        g.set_option("plan_threads", host_threads)
        g.set_option("plan_parallel_min_queries", 16 if host_threads > 1 else 2048)
        n0, f0 = g.counter("kw_device_plans"), g.counter("kw_device_plan_fallbacks")
        dev = g.keyword_search_batch(qs, k_stride=250)
        assert g.counter("kw_device_plans") == n0 + 1 and g.counter("kw_device_plan_fallbacks") == f0, "the batch was not planned on the device"
        assert np.array_equal(dev.status, host.status) and (dev.status == 0).all()
        for i, q in enumerate(qs):
            n = int(host.n_hits[i])
            assert dev.n_hits[i] == n and dev.num_matched[i] == host.num_matched[i], i
            assert np.array_equal(dev.keys[i, :n], host.keys[i, :n]) and np.array_equal(dev.scores[i, :n], host.scores[i, :n]), i
            assert np.array_equal(dev.text_match[i, :n], host.text_match[i, :n]) and np.array_equal(dev.match_score_index[i, :n], host.match_score_index[i, :n]), i
            H.assert_hits_equal(dev, i, H.oracle_keyword(orc, q), "device plan")
        assert dev.n_hits.sum() > 1000
        # the sliced host delivery (chained slices, enqueued in slice order): the first, large slice is planned on the device, the others on the host
        g.set_option("kw_host_split_queries", 8)
        sl = g.keyword_search_batch(qs, k_stride=250)
        assert g.counter("kw_device_plans") == n0 + 2
        g.set_option("kw_host_split_device_plan", 0)
        sl0 = g.keyword_search_batch(qs, k_stride=250)
        g.set_option("kw_host_split_device_plan", 1)
        assert g.counter("kw_device_plans") == n0 + 2 and np.array_equal(sl0.keys, sl.keys) and np.array_equal(sl0.n_hits, sl.n_hits)
        for i in range(len(qs)):
            n = int(host.n_hits[i])
            assert sl.n_hits[i] == n and np.array_equal(sl.keys[i, :n], host.keys[i, :n]) and np.array_equal(sl.scores[i, :n], host.scores[i, :n]) and sl.num_matched[i] == host.num_matched[i]
        # one query of another shape: the whole batch goes back to the host planner — same results
        g.set_option("kw_host_split_queries", 0)
        mixed = qs[:20] + [T.KwQuery([1, 2], sort=sorts[0], topster_size=250, filter_ids=np.arange(0, 3000, 3, dtype=np.uint32))]
        f1 = g.counter("kw_device_plan_fallbacks")
        mx = g.keyword_search_batch(mixed, k_stride=250)
        assert g.counter("kw_device_plan_fallbacks") == f1 + 1
        for i in range(20):
            n = int(host.n_hits[i])
            assert mx.n_hits[i] == n and np.array_equal(mx.keys[i, :n], host.keys[i, :n])
        H.assert_hits_equal(mx, 20, H.oracle_keyword(orc, mixed[20]), "fallback batch")
    finally:
        for name, v in (("kw_device_plan_min_queries", 512), ("kw_chunk_blocks", 0), ("plan_threads", 8), ("plan_parallel_min_queries", 2048), ("kw_host_split_queries", 2500)):
            g.set_option(name, v)


@pytest.mark.parametrize("chunk,two_kernels,pipelined", [(64, 1, 1), (1, 1, 1), (256, 1, 1), (64, 1, 0), (1, 1, 0), (64, 0, 0), (256, 1, 0)])
def test_multi_field_block_merge_windows_wide_runs_and_exhaustion(chunk, two_kernels, pipelined):
    """kw_mf_merge_field (the multi-field find kernel's block-level merge of a driver block with the SECOND token's lists): extreme length
    ratios in two fields — runs of 1, ~10, ~40 and > 64 blocks under one driver block (window re-centring, runs wider than the window / the
    tile -> per-candidate probes), driver ids beyond a list's end (cursor exhaustion), a second token that only one field holds — in the
    two-kernel form and the fused form (smaller tile): hits, scores, counts and ids = the oracle's or_iterator_t union"""
    from oracle import oracle_py as O
    n_docs, l0 = _synthetic_lists(5)
    _, l1 = _synthetic_lists(9)
    pts = H.points_of(n_docs)
    orc = O.OracleIndex(2, 1)
    orc.set_num_docs(n_docs)
    orc.set_sort_dense(0, pts)
    g = T.GpuIndex(0, H.emu_lib_path())
    for f, lists in ((0, l0), (1, l1)):
        g.field_create(f, False)
        for term, (ids, oi, off) in lists.items():
            if f == 1 and term == 3:
                continue                                         # token 3 lives in field 0 only
            orc.load_posting(f, term, ids, oi, off)
            g.term_upsert(f, term, ids, oi, off)
    g.column_set(0, pts)
    g.set_num_docs(n_docs)
    g.commit()
    g.set_option("kw_chunk_blocks", chunk)
    g.set_option("kw_two_kernels", two_kernels)
    g.set_option("kw_mf_pipelined", pipelined)         # 1: kw_find_mf2_kernel (two query_by fields: both second lists pipelined through LDS tiles), 0: kw_search_mf_kernel
    g.keep_result_ids(True)
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    f2 = [(0, 15), (1, 9)]
    qs = [T.KwQuery([1, 2], fields=f2, sort=sort, topster_size=250), T.KwQuery([2, 1, 3], fields=f2, sort=sort, topster_size=250),
          T.KwQuery([3, 2], fields=f2, sort=sort, topster_size=100), T.KwQuery([3, 1], fields=[(1, 4), (0, 4)], sort=sort, topster_size=250, match_type=B.SUM_SCORE),
          T.KwQuery([1], fields=f2, sort=sort, topster_size=250), T.KwQuery([2, 3], fields=f2, sort=sort, topster_size=250, filter_ids=np.arange(0, n_docs, 3, dtype=np.uint32)),
          T.KwQuery([2], fields=f2, sort=sort, topster_size=250)]       # a driver list of ~230 blocks: with chunk = 64 / 256 a work item reloads its lane-resident metadata window
    hits = g.keyword_search_batch(qs, k_stride=250)
    assert (hits.status == 0).all() and hits.n_hits[0] >= 250
    for i, q in enumerate(qs):
        ref = H.oracle_keyword(orc, q, ids_cap=200000)
        H.assert_hits_equal(hits, i, ref, "mf merge chunk=%d two=%d pipelined=%d q=%s" % (chunk, two_kernels, pipelined, q.tokens))
        assert np.array_equal(g.result_ids(i), ref.result_ids)
    g.close()


@pytest.mark.parametrize("chunk", [0, 1, 3])
def test_pipelined_two_field_find_kernel_equals_the_block_at_a_time_kernel_and_the_oracle(pair3, chunk):
    """kw_find_mf2_kernel (kw_mf_pipelined = 1, launches whose queries have <= 2 query_by fields) against kw_search_mf_kernel (0) and the oracle's
    or_iterator_t union (/root/reference/src/or_iterator.cpp:95-171): every pair of the three fields in either order, 1..7 tokens, a token only
    one field holds, tokens no field holds, duplicate tokens, dropped tokens, filters, excluded ids, small Topsters; a batch that ALSO holds a
    three-field query takes the kernel's four-list instantiation as a whole"""
    orc, g = pair3
    rng = np.random.default_rng(505)
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    pairs = [[(0, 15), (1, 7)], [(1, 7), (0, 15)], [(2, 2), (0, 9)], [(1, 3), (2, 3)], [(2, 1), (1, 1)]]
    qs = []
    for fp in pairs:
        for toks in ([1], [2, 1], [3, 1, 2], [5, 9], [1, 2, 3, 4], [7, 1, 2, 3, 6], [119], [9999, 2], [4, 4], [30, 2, 11], [1, 2, 3, 4, 5, 6, 8]):
            qs.append(T.KwQuery(toks, fields=fp, sort=sort, topster_size=250))
        qs.append(T.KwQuery([2, 1], fields=fp, sort=sort, topster_size=12, match_type=B.SUM_SCORE, excluded_ids=np.arange(0, 2500, 7)))
        qs.append(T.KwQuery([3, 1, 2], fields=fp, sort=sort, topster_size=250, filter_ids=np.sort(rng.choice(2500, size=900, replace=False))))
        qs.append(T.KwQuery([1, 2], fields=fp, sort=sort, topster_size=250, dropped_tokens=[3, 40]))
        qs.append(T.KwQuery([5], fields=fp, sort=sort, topster_size=250, dropped_tokens=[1], match_type=B.MAX_WEIGHT))
        qs.append(T.KwQuery([3, 1], fields=fp, sort=sort, topster_size=250, filter_ids=np.arange(0, 2500, 2), excluded_ids=np.arange(100, 300)))
    g.set_option("kw_chunk_blocks", chunk)
    g.keep_result_ids(True)
    try:
        outs = []
        for pipelined in (1, 0):
            g.set_option("kw_mf_pipelined", pipelined)
            n0 = g.counter("kw_mf_pipelined_launches")
            h = g.keyword_search_batch(qs, k_stride=250)
            assert (h.status == 0).all()
            assert (g.counter("kw_mf_pipelined_launches") > n0) == bool(pipelined)
            outs.append((h, [g.result_ids(i).copy() for i in range(len(qs))]))
        (h1, ids1), (h0, ids0) = outs
        for name in ("keys", "scores", "n_hits", "num_matched"):
            assert np.array_equal(getattr(h1, name), getattr(h0, name)), name
        for i, q in enumerate(qs):
            ref = H.oracle_keyword(orc, q, ids_cap=4000)
            H.assert_hits_equal(h1, i, ref, "pipelined two-field kernel chunk=%d fields=%s q=%s" % (chunk, q.fields, q.tokens))
            assert np.array_equal(ids1[i], ref.result_ids) and np.array_equal(ids0[i], ref.result_ids)
        assert h1.n_hits.sum() > 1000
        # three-field queries in the batch: the whole launch takes the four-list instantiation (the host decides per launch)
        g.set_option("kw_mf_pipelined", 1)
        f3 = [(0, 15), (1, 7), (2, 3)]
        mixed = qs[:8] + [T.KwQuery(t, fields=f3, sort=sort, topster_size=250) for t in ([3, 1, 2], [1], [2, 1], [7, 1, 2, 3, 6], [5, 9], [9999, 2])]
        mixed += [T.KwQuery([3, 1], fields=[(2, 3), (1, 7), (0, 15)], sort=sort, topster_size=250, filter_ids=np.arange(0, 2500, 2)),
                  T.KwQuery([1, 2], fields=f3, sort=sort, topster_size=250, dropped_tokens=[3])]
        n0 = g.counter("kw_mf_pipelined_launches")
        hm = g.keyword_search_batch(mixed, k_stride=250)
        assert g.counter("kw_mf_pipelined_launches") > n0
        for i, q in enumerate(mixed):
            H.assert_hits_equal(hm, i, H.oracle_keyword(orc, q, ids_cap=4000), "mixed 2/3-field batch q=%s" % (q.tokens,))
    finally:
        g.set_option("kw_mf_pipelined", 1)
        g.set_option("kw_chunk_blocks", 0)
        g.keep_result_ids(False)
