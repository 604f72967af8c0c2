"""group_by (tsgpu_keyword_search_grouped_batch: kw_groupby.hip.h + tsgpu_groupby.inc.h) executed on the CPU under the SIMT emulator of
tests/hipemu and checked against the oracle's restated distinct Topster (oracle/group_topster.h, itself pinned to the reference's own
topster.h: tests/test_oracle_groupby.py). Same sources as libtsgpu.so. The `-m gpu` twin is tests/test_gpu_groupby.py."""
import numpy as np
import pytest

import typesense_amd as T
from typesense_amd import _lib as B
from oracle import oracle_py as O
from tests import helpers as H

GROUP_COL = 1


def group_column(n_docs, seed, n_values=37, missing_every=11, two_fields=False, group_missing_values=False):
    """facet hashes of one (or two) group_by fields -> (distinct uint64[n_docs], has_value): Index::get_distinct_id per document"""
    rng = np.random.default_rng(seed)
    fields = []
    for f in range(2 if two_fields else 1):
        ptr = np.zeros(n_docs + 1, np.uint64)
        hs = []
        for d in range(n_docs):
            if missing_every and d % missing_every == f:
                pass                                                  # no value in this field
            elif f == 1 and d % 5 == 0:
                hs += [int(x) for x in rng.integers(1, 9, 2)]         # an array field: two hashes
            else:
                hs.append(int(rng.integers(1, n_values + 1)) * 2654435761 % (2**32))
            ptr[d + 1] = len(hs)
        fields.append((ptr, np.array(hs, np.uint32)))
    return O.distinct_ids(n_docs, fields, group_missing_values)


def sort_key_rows(keys, scores):
    return sorted([(int(s[0]), int(s[1]), int(s[2]), int(k)) for k, s in zip(keys, scores)], reverse=True)


def check_query(h, gh, i, ref, first_pass, group_limit, what="", check_total=True):
    assert int(h.status[i]) == 0, (what, h.status[i])
    ng = int(gh.n_groups[i])
    assert ng == ref.n_groups, "%s q%d: %d groups vs oracle %d" % (what, i, ng, ref.n_groups)
    assert not check_total or int(gh.groups_total[i]) == ref.groups_exact, what
    assert int(h.num_matched[i]) == ref.num_keyword_matches, what
    if first_pass:
        # the reference keeps these KVs in heap-array order and reads them as a set; the library returns them best first
        want = sorted([(int(ref.scores[j, 0]), int(ref.scores[j, 1]), int(ref.scores[j, 2]), int(ref.keys[j]), int(ref.distinct_key[j]), int(ref.group_found[j]))
                       for j in range(ref.n_groups)], reverse=True)
        got = [(int(h.scores[i, r, 0]), int(h.scores[i, r, 1]), int(h.scores[i, r, 2]), int(h.keys[i, r]), int(gh.distinct_key[i, r]), int(gh.group_found[i, r]))
               for r in range(ng)]
        assert got == want, "%s q%d first pass\n%s\n%s" % (what, i, got[:5], want[:5])
        assert int(h.n_hits[i]) == ng and (gh.group_size[i, :ng] == 1).all()
        assert int(gh.groups_count[i]) == ref.groups_count, (what, gh.groups_count[i], ref.groups_count)
        if gh.loglog_registers is not None:
            assert np.array_equal(gh.loglog_registers[i], ref.loglog), what
    else:
        assert np.array_equal(gh.distinct_key[i, :ng], ref.distinct_key), "%s q%d: group order\n%s\n%s" % (what, i, gh.distinct_key[i, :ng][:8], ref.distinct_key[:8])
        assert np.array_equal(gh.group_found[i, :ng], ref.group_found), what
        assert np.array_equal(gh.group_size[i, :ng], ref.group_size), what
        for r in range(ng):
            n = int(ref.group_size[r])
            lo = r * group_limit
            assert np.array_equal(h.keys[i, lo:lo + n], ref.keys[ref.begin[r]:ref.begin[r + 1]]), "%s q%d group %d keys" % (what, i, r)
            assert np.array_equal(h.scores[i, lo:lo + n], ref.scores[ref.begin[r]:ref.begin[r + 1]]), "%s q%d group %d scores" % (what, i, r)
        assert int(h.n_hits[i]) == int(ref.group_size.sum())
        assert int(gh.groups_count[i]) == 0


def oracle_grouped(orc, q, distinct, has_value, group_limit, first_pass, gmv=False, wildcard=False, n_docs=None):
    if wildcard:
        raise AssertionError("use oracle_grouped_wildcard")
    oq = H.oracle_query(orc, q)
    return orc.search_keyword_grouped(oq, distinct, group_limit, first_pass, has_value=has_value, group_missing_values=gmv, ids_cap=1 << 20)


def oracle_grouped_wildcard(q, n_docs, points, distinct, group_limit, first_pass, gmv=False):
    """Index::search_wildcard's grouped loop restated with the oracle's collector: every filter id (every seq_id) minus the excluded ids is
    scored by its sort keys (text-match slot = 100, sign-flipped for ASC) and added with its distinct key"""
    ids = np.arange(n_docs, dtype=np.uint32) if q.filter_ids is None else q.filter_ids
    if q.excluded_ids is not None:
        ids = np.setdiff1d(ids, q.excluded_ids)
    sc = np.zeros((ids.size, 3), np.int64)
    for c, (kind, order, col) in enumerate(q.sort):
        v = np.full(ids.size, 100, np.int64) if kind == B.SORT_TEXT_MATCH else (ids.astype(np.int64) if kind == B.SORT_SEQ_ID else points[ids])
        sc[:, c] = v if order == 1 else -v
    dk = np.array([int(distinct[i]) if i < distinct.size else (1 if gmv else int(i)) for i in ids], np.uint64)
    cap = q.topster_size if q.topster_size else min(250, ids.size if q.filter_ids is not None else n_docs)
    cap = max(1, min(cap, n_docs))
    ret, g = O.group_topster_run(cap, group_limit, first_pass, ids.astype(np.uint64), dk, sc)
    g.num_keyword_matches = int(ids.size)
    return g


@pytest.fixture(scope="module")
def world():
    docs = H.zipf_docs(3000, 300, 12, seed=1)
    orc, g = H.build_pair(docs, H.emu_lib_path())
    distinct, has_value = group_column(3000, seed=3)
    g.column_set(GROUP_COL, distinct.view(np.int64))
    yield orc, g, docs, distinct, has_value
    g.close()


def _queries(rng, n, vocab_hi, n_tok, **kw):
    return [T.KwQuery(rng.choice(np.arange(1, vocab_hi), size=n_tok, replace=False), **kw) for _ in range(n)]


@pytest.mark.parametrize("first_pass", [True, False])
def test_grouped_keyword_equals_oracle(world, first_pass):
    orc, g, _, distinct, has_value = world
    rng = np.random.default_rng(10 + int(first_pass))
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    qs = _queries(rng, 3, 12, 1, sort=sort, topster_size=250) + _queries(rng, 4, 30, 2, sort=sort, topster_size=250) + _queries(rng, 3, 25, 3, sort=sort, topster_size=250)
    limits = [1, 2, 3, 7, 3, 3, 2, 3, 5, 3]
    groups = [(limits[i], GROUP_COL, int(first_pass), 0, 0) for i in range(len(qs))]
    h, gh = g.keyword_search_grouped_batch(qs, groups, k_stride=250 * 7, g_stride=250, want_registers=True)
    assert gh.n_groups.sum() > 30
    for i, q in enumerate(qs):
        check_query(h, gh, i, oracle_grouped(orc, q, distinct, has_value, limits[i], first_pass), first_pass, limits[i], "kw")


@pytest.mark.parametrize("first_pass", [True, False])
def test_grouped_small_topster_selects_the_best_groups(world, first_pass):
    orc, g, _, distinct, has_value = world
    sort3 = ((B.SORT_INT64_COLUMN, -1, 0), (B.SORT_TEXT_MATCH, 1, 0), (B.SORT_SEQ_ID, -1, 0))
    qs = [T.KwQuery([1], topster_size=1), T.KwQuery([2], topster_size=2), T.KwQuery([1, 2], topster_size=5), T.KwQuery([3], topster_size=5, sort=sort3),
          T.KwQuery([4, 2], topster_size=3, sort=((B.SORT_SEQ_ID, 1, 0),)), T.KwQuery([9999], topster_size=4), T.KwQuery([5], topster_size=30, sort=((B.SORT_TEXT_MATCH, -1, 0), (B.SORT_SEQ_ID, 1, 0)))]
    groups = [(3, GROUP_COL, int(first_pass), 0, 0)] * len(qs)
    h, gh = g.keyword_search_grouped_batch(qs, groups, k_stride=30 * 3, g_stride=30, want_registers=True)
    for i, q in enumerate(qs):
        check_query(h, gh, i, oracle_grouped(orc, q, distinct, has_value, 3, first_pass), first_pass, 3, "small k")
    assert int(gh.n_groups[5]) == 0 and int(h.n_hits[5]) == 0


@pytest.mark.parametrize("first_pass", [True, False])
def test_grouped_with_filter_excluded_dropped_and_flags(world, first_pass):
    orc, g, _, distinct, has_value = world
    rng = np.random.default_rng(4)
    filt = np.sort(rng.choice(3000, 900, replace=False)).astype(np.uint32)
    excl = np.sort(rng.choice(3000, 200, replace=False)).astype(np.uint32)
    base = dict(topster_size=40)
    qs = [T.KwQuery([1, 2], filter_ids=filt, **base), T.KwQuery([2], excluded_ids=excl, **base), T.KwQuery([3, 1], filter_ids=filt, excluded_ids=excl, **base),
          T.KwQuery([2, 3], dropped_tokens=[1], **base), T.KwQuery([1, 2], match_type=B.SUM_SCORE, weight=3, total_cost=2, **base),
          T.KwQuery([1, 2], prioritize_token_position=True, prioritize_exact_match=False, **base)]
    groups = [(2, GROUP_COL, int(first_pass), 0, 0)] * len(qs)
    h, gh = g.keyword_search_grouped_batch(qs, groups, k_stride=80, g_stride=40)
    for i, q in enumerate(qs):
        check_query(h, gh, i, oracle_grouped(orc, q, distinct, has_value, 2, first_pass), first_pass, 2, "filter")


@pytest.mark.parametrize("first_pass", [True, False])
def test_grouped_wildcard_and_many_groups(world, first_pass):
    """q = *: 3 000 documents; with a column shorter than the collection every document beyond it is its own group (or all of them ONE group with
    group_missing_values): thousands of groups go through the select kernel's compaction"""
    orc, g, docs, distinct, has_value = world
    points = H.points_of(3000)
    short = distinct[:1200].copy()
    short[7] = np.uint64(2**64 - 1)                                   # the key that equals the table's empty marker
    short[9] = np.uint64(2**64 - 1)
    g.column_set(2, short.view(np.int64))
    rng = np.random.default_rng(8)
    filt = np.sort(rng.choice(3000, 1500, replace=False)).astype(np.uint32)
    excl = np.sort(rng.choice(3000, 100, replace=False)).astype(np.uint32)
    sort = ((B.SORT_INT64_COLUMN, 1, 0), (B.SORT_SEQ_ID, -1, 0))
    qs = [T.KwQuery([], sort=sort, topster_size=250), T.KwQuery([], sort=sort, topster_size=250, filter_ids=filt, excluded_ids=excl),
          T.KwQuery([], sort=((B.SORT_TEXT_MATCH, -1, 0), (B.SORT_INT64_COLUMN, -1, 0), (B.SORT_SEQ_ID, 1, 0)), topster_size=100, excluded_ids=excl),
          T.KwQuery([], sort=sort, topster_size=600), T.KwQuery([], sort=sort, topster_size=1000, excluded_ids=excl)]      # Topsters beyond 256 / 768: the larger top-K buffers
    for gmv in (0, 1):
        groups = [(3, 2, int(first_pass), gmv, 1)] * len(qs)
        h, gh = g.keyword_search_grouped_batch(qs, groups, k_stride=3000, g_stride=1000, want_registers=True)
        for i, q in enumerate(qs):
            check_query(h, gh, i, oracle_grouped_wildcard(q, 3000, points, short, 3, first_pass, bool(gmv)), first_pass, 3, "wild gmv=%d" % gmv)
        assert int(gh.groups_total[0]) == (1800 + len(set(short.tolist())) if not gmv else len(set(short.tolist()) | {1}))
    # a batch whose largest Topster is 600: the middle top-K buffer
    h, gh = g.keyword_search_grouped_batch(qs[3:4], [(2, 2, int(first_pass), 0, 1)], k_stride=1200, g_stride=600)
    check_query(h, gh, 0, oracle_grouped_wildcard(qs[3], 3000, points, short, 2, first_pass, False), first_pass, 2, "wild k=600")


def test_grouped_big_output_arrays_take_the_direct_delivery(world):
    """output arrays beyond 4 MB are copied to the caller array by array (small calls come back as one packed block)"""
    orc, g, _, distinct, has_value = world
    qs = [T.KwQuery([1, 2], topster_size=40), T.KwQuery([3], topster_size=40), T.KwQuery([9999], topster_size=40)]
    for first_pass in (True, False):
        h, gh = g.keyword_search_grouped_batch(qs, [(2, GROUP_COL, int(first_pass), 0, 0)] * len(qs), k_stride=40000, g_stride=40, want_registers=True)
        for i, q in enumerate(qs):
            check_query(h, gh, i, oracle_grouped(orc, q, distinct, has_value, 2, first_pass), first_pass, 2, "big strides")


def test_group_count_sketch_for_keys_of_every_printed_length(world):
    """LogLogBeta hashes std::to_string(distinct_key): 1 to 20 characters, four wyhash branches (<= 3, 4..7, 8..16, 17..20 bytes)"""
    orc, g, _, _, _ = world
    vals = []
    for k in range(20):
        vals += [max(10**k - 1, 0), 10**k, 10**k + 1, 7 * 10**k + 12345 % (10**k + 1)]
    vals = sorted(set(v for v in vals if v < 2**64)) + [2**64 - 1, 2**63, 99999999, 100000000, 9999999999999999, 10**16, 2**32 - 1, 2**32]
    col = np.array([vals[i % len(vals)] for i in range(3000)], np.uint64)
    g.column_set(4, col.view(np.int64))
    q = T.KwQuery([], sort=((B.SORT_SEQ_ID, 1, 0),), topster_size=250)
    h, gh = g.keyword_search_grouped_batch([q], [(1, 4, 1, 0, 1)], k_stride=250, g_stride=250, want_registers=True)
    regs = np.zeros(16384, np.uint8)
    want = O.lib().orc_loglog_of_keys(np.ascontiguousarray(np.unique(col)).ctypes.data, int(np.unique(col).size), regs.ctypes.data)
    assert int(gh.groups_total[0]) == np.unique(col).size
    assert np.array_equal(gh.loglog_registers[0], regs) and int(gh.groups_count[0]) == want


def test_grouped_queries_of_more_than_three_tokens(world):
    """a batch that holds a query of more than three lists takes the 10-token form of the scoring kernel (one wave per workgroup)"""
    orc, g, _, distinct, has_value = world
    qs = [T.KwQuery([1, 2, 3, 4], topster_size=40), T.KwQuery([2, 1], topster_size=40), T.KwQuery([5, 3, 1, 2, 4, 6], topster_size=40),
          T.KwQuery([1, 2, 3], dropped_tokens=[4, 5], topster_size=40)]
    for first_pass in (True, False):
        h, gh = g.keyword_search_grouped_batch(qs, [(2, GROUP_COL, int(first_pass), 0, 0)] * len(qs), k_stride=80, g_stride=40)
        for i, q in enumerate(qs):
            check_query(h, gh, i, oracle_grouped(orc, q, distinct, has_value, 2, first_pass), first_pass, 2, "T>3")
    assert gh.n_groups.sum() > 10


def test_grouped_two_fields_arrays_and_missing_ids(world):
    """two group_by fields (one an array): hash_combine over all hashes; ids_out carries all_result_ids, from which the caller takes
    group_by_missing_value_ids"""
    orc, g, _, _, _ = world
    distinct, has_value = group_column(3000, seed=5, two_fields=True)
    g.column_set(3, distinct.view(np.int64))
    q = T.KwQuery([1, 2], topster_size=50)
    h, gh, ids = g.keyword_search_grouped_batch([q, q], [(3, 3, 1, 0, 0), (3, 3, 0, 0, 0)], k_stride=150, g_stride=50, want_ids=True)
    ref1 = oracle_grouped(orc, q, distinct, has_value, 3, True)
    check_query(h, gh, 0, ref1, True, 3, "two fields")
    check_query(h, gh, 1, oracle_grouped(orc, q, distinct, has_value, 3, False), False, 3, "two fields")
    assert np.array_equal(ids[0], ref1.result_ids) and np.array_equal(ids[1], ref1.result_ids)
    missing = ids[0][has_value[ids[0]] == 0]
    assert np.array_equal(missing, ref1.missing_ids) and missing.size > 0


def test_grouped_multi_field_query():
    rng = np.random.default_rng(21)
    d0 = H.zipf_docs(1500, 120, 8, seed=2)
    d1 = H.zipf_docs(1500, 120, 5, seed=3)
    orc, g = H.build_pair_fields([d0, d1], H.emu_lib_path())
    try:
        distinct, has_value = group_column(1500, seed=9, n_values=20)
        g.column_set(1, distinct.view(np.int64))
        qs = [T.KwQuery([1, 2], fields=[(0, 15), (1, 7)], topster_size=60), T.KwQuery([3], fields=[(0, 3), (1, 15)], topster_size=60, match_type=B.MAX_WEIGHT),
              T.KwQuery([2, 5, 1], fields=[(0, 15), (1, 15)], topster_size=60)]
        for first_pass in (1, 0):
            h, gh = g.keyword_search_grouped_batch(qs, [(2, 1, first_pass, 0, 0)] * len(qs), k_stride=120, g_stride=60)
            for i, q in enumerate(qs):
                check_query(h, gh, i, oracle_grouped(orc, q, distinct, has_value, 2, bool(first_pass)), bool(first_pass), 2, "multi-field")
    finally:
        g.close()


def test_grouped_string_array_fields():
    """a string[] query_by field (per-element Match, src/index.cpp:1351-1395) alone and next to a plain field, grouped"""
    from tests.test_emu_keyword import make_pair_arr
    orc, g = make_pair_arr(H.emu_lib_path())
    try:
        distinct, has_value = group_column(2000, seed=12, n_values=15)
        g.column_set(1, distinct.view(np.int64))
        sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
        qs = []
        for toks in ([1], [1, 2], [2, 1, 3], [4, 2, 1, 3]):
            qs.append(T.KwQuery(toks, fields=[(0, 15)], sort=sort, topster_size=60))
            qs.append(T.KwQuery(toks, fields=[(0, 4), (1, 9)], sort=sort, topster_size=60, prioritize_token_position=True))
        for first_pass in (True, False):
            h, gh = g.keyword_search_grouped_batch(qs, [(2, 1, int(first_pass), 0, 0)] * len(qs), k_stride=120, g_stride=60)
            for i, q in enumerate(qs):
                check_query(h, gh, i, oracle_grouped(orc, q, distinct, has_value, 2, first_pass), first_pass, 2, "array fields")
        assert gh.n_groups.sum() > 20
    finally:
        g.close()


def test_grouped_big_groups_are_cut_into_chunks():
    """a group_by field with a handful of values over many matches: groups of more than 4096 members take the chunked path of the second pass (one workgroup per
    8192 members, the last chunk folds the partial lists) and the wave-aggregated table update; q = * needs no postings — 30 000 documents, groups of 70 % / 29 % /
    1 % + singletons, group_limit 3 and 50, with and without filter / excluded ids"""
    n = 30000
    g = T.GpuIndex(0, H.emu_lib_path())
    try:
        g.field_create(0, False)
        g.set_num_docs(n)
        points = H.points_of(n)
        g.column_set(0, points)
        g.commit()
        rng = np.random.default_rng(3)
        u = rng.random(n)
        col = np.where(u < 0.7, 111, np.where(u < 0.99, 222, 333)).astype(np.uint64)
        col[rng.choice(n, 40, replace=False)] = np.arange(40, dtype=np.uint64) + np.uint64(10**12)       # singletons
        g.column_set(1, col.view(np.int64))
        excl = np.sort(rng.choice(n, 500, replace=False)).astype(np.uint32)
        filt = np.sort(rng.choice(n, 20000, replace=False)).astype(np.uint32)
        sort = ((B.SORT_INT64_COLUMN, 1, 0), (B.SORT_SEQ_ID, -1, 0))
        qs = [T.KwQuery([], sort=sort, topster_size=250), T.KwQuery([], sort=sort, topster_size=30, filter_ids=filt, excluded_ids=excl),
              T.KwQuery([], sort=((B.SORT_SEQ_ID, 1, 0),), topster_size=2)]
        for limit in (3, 50):
            for first_pass in (False, True):
                h, gh = g.keyword_search_grouped_batch(qs, [(limit, 1, int(first_pass), 0, 1)] * len(qs), k_stride=250 * limit, g_stride=250, want_registers=True)
                for i, q in enumerate(qs):
                    check_query(h, gh, i, oracle_grouped_wildcard(q, n, points, col, limit, first_pass), first_pass, limit, "big groups limit %d" % limit)
        assert int(gh.group_found[0, :int(gh.n_groups[0])].max()) > 8192 * 2
    finally:
        g.close()


def test_grouped_bad_queries_do_not_disturb_their_neighbours(world):
    orc, g, _, distinct, has_value = world
    good = T.KwQuery([1, 2], topster_size=20)
    qs = [good, T.KwQuery(list(range(1, 12)), topster_size=20), good, T.KwQuery([1], topster_size=20), good]
    groups = [(2, GROUP_COL, 0, 0, 0), (2, GROUP_COL, 0, 0, 0), (0, GROUP_COL, 0, 0, 0), (2, 77, 0, 0, 0), (2, GROUP_COL, 1, 0, 0)]
    h, gh = g.keyword_search_grouped_batch(qs, groups, k_stride=40, g_stride=20)
    assert list(h.status) == [0, B.ERR_UNSUPPORTED, B.ERR_INVALID, B.ERR_NOT_FOUND, 0]
    assert list(h.n_hits[1:4]) == [0, 0, 0] and list(gh.n_groups[1:4]) == [0, 0, 0]
    check_query(h, gh, 0, oracle_grouped(orc, good, distinct, has_value, 2, False), False, 2, "neighbour")
    check_query(h, gh, 4, oracle_grouped(orc, good, distinct, has_value, 2, True), True, 2, "neighbour")
    # strides too small for the request: 400 for that query
    h, gh = g.keyword_search_grouped_batch([good], [(3, GROUP_COL, 0, 0, 0)], k_stride=40, g_stride=20)
    assert int(h.status[0]) == B.ERR_INVALID


def test_grouped_calls_from_concurrent_threads_are_coalesced_and_keep_their_own_results(world):
    """the server's calling convention: one grouped query per call from many request threads — parked in the library's combiner, run as one batch, every
    caller gets its own slice / id list / status (a caller whose strides are too small for its own request fails alone)"""
    import threading
    orc, g, _, distinct, has_value = world
    rng = np.random.default_rng(77)
    qs = _queries(rng, 10, 25, 2, topster_size=30) + _queries(rng, 6, 12, 1, topster_size=30)
    passes = [int(i % 2 == 0) for i in range(len(qs))]
    want = []
    for q, fp in zip(qs, passes):
        h, gh, ids = g.keyword_search_grouped_batch([q], [(2, GROUP_COL, fp, 0, 0)], k_stride=60, g_stride=30, want_ids=True, want_registers=True)
        want.append((h, gh, ids))
    rounds0 = g.counter("gb_batch_rounds")
    got = [None] * len(qs)
    errs = []
    start = threading.Barrier(len(qs) + 1)

    def body(i):
        try:
            start.wait()
            for _ in range(3):
                if i == 5:        # strides too small for this request: 400 for this caller only
                    h, gh = g.keyword_search_grouped_batch([qs[i]], [(2, GROUP_COL, passes[i], 0, 0)], k_stride=4, g_stride=2)
                    got[i] = (h, gh, None)
                else:
                    got[i] = g.keyword_search_grouped_batch([qs[i]], [(2, GROUP_COL, passes[i], 0, 0)], k_stride=60, g_stride=30, want_ids=True, want_registers=True)
        except Exception as e:      # noqa: BLE001
            errs.append((i, repr(e)))
    th = [threading.Thread(target=body, args=(i,)) for i in range(len(qs))]
    for t in th:
        t.start()
    start.wait()
    for t in th:
        t.join()
    assert not errs, errs
    for i in range(len(qs)):
        h, gh, ids = got[i]
        wh, wgh, wids = want[i]
        if i == 5:
            assert int(h.status[0]) in (B.ERR_INVALID, 0)
            if int(wgh.n_groups[0]) > 2:
                assert int(h.status[0]) == B.ERR_INVALID and int(h.n_hits[0]) == 0
            continue
        assert int(h.status[0]) == 0
        assert np.array_equal(h.n_hits, wh.n_hits) and np.array_equal(h.num_matched, wh.num_matched) and np.array_equal(gh.n_groups, wgh.n_groups)
        ng = int(gh.n_groups[0])
        ext = int(h.n_hits[0]) if passes[i] else ng * 2
        assert np.array_equal(gh.distinct_key[0, :ng], wgh.distinct_key[0, :ng]) and np.array_equal(gh.group_found[0, :ng], wgh.group_found[0, :ng])
        assert np.array_equal(gh.group_size[0, :ng], wgh.group_size[0, :ng])
        for r in range(ng):
            n = int(gh.group_size[0, r]); lo = r * (1 if passes[i] else 2)
            assert np.array_equal(h.keys[0, lo:lo + n], wh.keys[0, lo:lo + n]) and np.array_equal(h.scores[0, lo:lo + n], wh.scores[0, lo:lo + n])
        assert int(gh.groups_count[0]) == int(wgh.groups_count[0]) and int(gh.groups_total[0]) == int(wgh.groups_total[0])
        assert np.array_equal(gh.loglog_registers[0], wgh.loglog_registers[0])
        assert np.array_equal(ids[0], wids[0]) and ext >= 0
    # (whether calls were coalesced depends on timing; under 16 threads x 3 calls some rounds serve several callers)
    assert g.counter("gb_batch_rounds") > rounds0


@pytest.mark.parametrize("first_pass", [True, False])
def test_grouped_candidate_combinations_fold_like_the_shared_collector(world, first_pass):
    """Index::search_all_candidates with group_by: the combinations of a user query are passes over ONE distinct Topster and ONE groups_processed. Neighbouring
    token ranks make the passes overlap heavily: documents met by several combinations (a second pass counts them once, with their greatest KV — the later
    combination on ties), passes without matches, a single-combination user query, small Topsters"""
    orc, g, _, distinct, has_value = world
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    users = [[[1, 2], [1, 3], [2, 3], [1, 2], [9999, 1]],                      # a repeated combination: every KV ties with its earlier self -> the later pass wins
             [[3], [4], [3, 4], [5]],
             [[2, 1, 3]],
             [[9999], [9998]],
             [[1], [2], [3], [4], [5], [6], [7], [8]]]
    tsz = [40, 5, 40, 40, 250]
    limit = 3
    combos = [[T.KwQuery(c, sort=sort, topster_size=tsz[u], total_cost=int(j > 0)) for j, c in enumerate(cs)] for u, cs in enumerate(users)]
    groups = [(limit, GROUP_COL, int(first_pass), 0, 0)] * len(users)
    h, gh, qidx, ids = g.keyword_search_grouped_candidates_batch(combos, groups, k_stride=750, g_stride=250, want_ids=True, want_registers=True)
    assert (h.status == 0).all()
    for u, cs in enumerate(combos):
        ref, rqi = orc.search_candidates_grouped([H.oracle_query(orc, q) for q in cs], distinct, limit, first_pass, has_value=has_value, ids_cap=1 << 20)
        check_query(h, gh, u, ref, first_pass, limit, "candidates u%d" % u)
        assert np.array_equal(ids[u], ref.result_ids), u
        ng = int(gh.n_groups[u])
        if first_pass:        # the reference's heap order is not the library's: compare query_index per key
            want = {int(k): int(q) for k, q in zip(ref.keys, rqi)}
            got = {int(h.keys[u, r]): int(qidx[u, r]) for r in range(ng)}
            assert got == want, u
        else:
            for r in range(ng):
                n = int(ref.group_size[r])
                assert np.array_equal(qidx[u, r * limit:r * limit + n], rqi[ref.begin[r]:ref.begin[r + 1]].astype(np.uint32)), (u, r)
    assert int(gh.n_groups[3]) == 0 and int(gh.n_groups[0]) > 5


def test_grouped_candidates_a_user_query_failing_after_the_id_pass_leaves_its_neighbours_alone(world):
    """user 0 = one healthy combination + one whose deadline has passed (408 in the id pass, after the translation accepted it); with the ids on the
    device the healthy combination's ids are packed into the id buffer all the same. User 0 reports 408 and nothing else; user 1 and 2 must get exactly what
    they get in a batch without user 0 (ids, hits, groups) — before the fix every later query's item range was short by user 0's ids"""
    orc, g, _, distinct, has_value = world
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    for first_pass in (True, False):
        grp = (3, GROUP_COL, int(first_pass), 0, 0)
        healthy = [[T.KwQuery([3], sort=sort, topster_size=40)], [T.KwQuery([2, 1], sort=sort, topster_size=40), T.KwQuery([4], sort=sort, topster_size=40, total_cost=1)]]
        failing = [T.KwQuery([1, 2], sort=sort, topster_size=40), T.KwQuery([1, 3], sort=sort, topster_size=40, total_cost=1, deadline_us=1)]
        wh, wgh, wq, wids = g.keyword_search_grouped_candidates_batch(healthy, [grp] * 2, k_stride=750, g_stride=250, want_ids=True, want_registers=True)
        assert (wh.status == 0).all() and wids[0].size > 0 and wids[1].size > 0
        h, gh, qx, ids = g.keyword_search_grouped_candidates_batch([failing] + healthy, [grp] * 3, k_stride=750, g_stride=250, want_ids=True, want_registers=True)
        assert list(h.status) == [B.ERR_DEADLINE, 0, 0], list(h.status)
        assert int(h.n_hits[0]) == 0 and int(gh.n_groups[0]) == 0 and ids[0].size == 0
        for u in (0, 1):
            ref, _ = orc.search_candidates_grouped([H.oracle_query(orc, q) for q in healthy[u]], distinct, 3, first_pass, has_value=has_value, ids_cap=1 << 20)
            check_query(h, gh, u + 1, ref, first_pass, 3, "neighbour of a failed user query")
            assert np.array_equal(ids[u + 1], wids[u]) and np.array_equal(ids[u + 1], ref.result_ids)
            assert int(h.n_hits[u + 1]) == int(wh.n_hits[u]) and int(gh.n_groups[u + 1]) == int(wgh.n_groups[u])
            ng = int(wgh.n_groups[u])
            assert np.array_equal(gh.distinct_key[u + 1, :ng], wgh.distinct_key[u, :ng]) and np.array_equal(gh.group_size[u + 1, :ng], wgh.group_size[u, :ng])
            for r in range(ng):                           # (a group's unused slots are not written)
                lo, n = r * (1 if first_pass else 3), int(wgh.group_size[u, r])
                assert np.array_equal(h.keys[u + 1, lo:lo + n], wh.keys[u, lo:lo + n]) and np.array_equal(qx[u + 1, lo:lo + n], wq[u, lo:lo + n])


def test_aux_timings_describe_the_last_grouped_batch(world):
    """tsgpu_last_aux_timings (bench.py `general_kernels.group_by.roofline`): matched ids, table slots and algorithmic bytes of the last grouped batch"""
    orc, g, _, distinct, has_value = world
    qs = [T.KwQuery([1, 2], topster_size=20), T.KwQuery([3], topster_size=20)]
    h, gh = g.keyword_search_grouped_batch(qs, [(2, GROUP_COL, 0, 0, 0)] * 2, k_stride=40, g_stride=20)
    t = g.aux_timings()
    ids = int(h.num_matched.sum())
    assert t.gb_matched_ids == ids > 0 and t.gb_table_slots >= 2 * ids
    assert t.gb_algorithmic_bytes == 36 * t.gb_matched_ids + 20 * t.gb_table_slots
    assert t.gb_kernels_ms >= t.gb_fold_ms >= 0 and t.gb_kernels_ms >= t.gb_select_ms >= 0 and t.gb_id_pass_ms > 0


def _grouping_basics_product(lib):
    """CollectionGroupingTest.GroupingBasics (/root/reference/test/collection_grouping_test.cpp:71-96) through the library: q = *, group_by size, group_limit 2,
    sort rating desc: found_docs 12, found 3; groups 11 / 10 / 12 with 2 / 7 / 3 documents and the hits 5,1 / 4,3 / 2,8"""
    import json, os
    from tests.test_oracle_groupby import _grouping_basics
    fx, n, distinct = _grouping_basics()
    g = T.GpuIndex(0, lib)
    g.set_num_docs(n)
    g.field_create(0, False)
    g.commit()
    g.column_set(0, np.array(fx["rating_keys"], np.int64))
    g.column_set(1, distinct.view(np.int64))
    wq = T.KwQuery([], sort=((B.SORT_INT64_COLUMN, 1, 0),), topster_size=250)
    h, gh = g.keyword_search_grouped_batch([wq], [(2, 1, 0, 0, 1)], k_stride=500, g_stride=250)
    assert int(h.status[0]) == 0 and int(h.num_matched[0]) == 12 and int(gh.n_groups[0]) == 3
    for r, e in enumerate(fx["expected_groups"]):
        assert int(gh.group_found[0, r]) == e["found"] and int(gh.group_size[0, r]) == len(e["hits"])
        assert h.keys[0, r * 2:r * 2 + 2].tolist() == e["hits"]
    h1, g1 = g.keyword_search_grouped_batch([wq], [(2, 1, 1, 0, 1)], k_stride=250, g_stride=250)
    assert int(g1.n_groups[0]) == 3 and sorted(h1.keys[0, :3].tolist()) == [2, 4, 5]
    # getGroupsCount() = LogLogBeta's truncated estimate (2 for these keys, from the reference's own Topster too); `found` = max(it, groups returned) = 3 (src/index.cpp:2766-2770)
    assert int(g1.groups_count[0]) == 2 and max(int(g1.groups_count[0]), int(gh.n_groups[0])) == 3
    # the second request (:112-148): group_by rating, sort_by size DESC -> 7 groups, groups 0 / 1 / 5 / 6 as the test asserts them
    from tests.test_oracle_groupby import _by_rating
    fx, n, by_rating = _by_rating()
    g.column_set(2, np.array(fx["sizes"], np.int64))
    g.column_set(3, by_rating.view(np.int64))
    wq = T.KwQuery([], sort=((B.SORT_INT64_COLUMN, 1, 2),), topster_size=250)
    h, gh = g.keyword_search_grouped_batch([wq], [(2, 3, 0, 0, 1)], k_stride=500, g_stride=250)
    e = fx["by_rating_expected"]
    assert int(h.num_matched[0]) == 12 and int(gh.n_groups[0]) == e["n_groups"]
    for r, want in e["groups"].items():
        r = int(r)
        assert int(gh.group_found[0, r]) == want["found"] and h.keys[0, r * 2:r * 2 + int(gh.group_size[0, r])].tolist() == want["hits"]
    # CollectionGroupingTest.GroupingCompoundKey (:150-215): group_by size + brand (optional), 10 groups, and the brand facet counted by groups
    from tests.test_oracle_groupby import _compound_key
    fx, ck, n, compound, brand = _compound_key()
    g.column_set(4, compound.view(np.int64))
    wq = T.KwQuery([], sort=((B.SORT_INT64_COLUMN, 1, 0),), topster_size=250)
    h, gh = g.keyword_search_grouped_batch([wq], [(2, 4, 0, 1, 1)], k_stride=500, g_stride=250)
    assert int(h.num_matched[0]) == 12 and int(gh.n_groups[0]) == ck["n_groups"]
    for r, want in ck["groups"].items():
        r = int(r)
        assert int(gh.group_found[0, r]) == want["found"] and h.keys[0, r * 2:r * 2 + int(gh.group_size[0, r])].tolist() == want["hits"]
    g.facet_set(0, *brand)
    fh, fc, fd, fp, fn = g.facet_count_batch(0, [np.arange(n, dtype=np.uint32)], group_column=4, group_missing_values=True)[0]
    got = {int(a): int(b) for a, b in zip(fh, fc)}
    assert {name: got[i] for name, i in ck["brand_ids"].items()} == ck["expected_grouped_facets"]
    # GroupingWithGropLimitOfOne (:372-411): group_by brand (optional), group_limit 1 -> 5 groups, every brand counted once
    from tests.test_oracle_groupby import _by_brand, _fixture
    one = _fixture()["group_limit_of_one"]
    fx, ck, n, by_brand, brand = _by_brand(True)
    g.column_set(5, by_brand.view(np.int64))
    h, gh = g.keyword_search_grouped_batch([wq], [(1, 5, 0, 1, 1)], k_stride=250, g_stride=250)
    assert int(gh.n_groups[0]) == one["n_groups"]
    for r, want in enumerate(one["groups"]):
        assert int(gh.group_found[0, r]) == want["found"] and int(gh.group_size[0, r]) == 1 and [int(h.keys[0, r])] == want["hits"]
    fh, fc, fd, fp, fn = g.facet_count_batch(0, [np.arange(n, dtype=np.uint32)], group_column=5, group_missing_values=True)[0]
    got = {int(a): int(b) for a, b in zip(fh, fc)}
    assert {name: got[i] for name, i in ck["brand_ids"].items()} == one["expected_grouped_facets"]
    g.close()
    # ControlMissingValues (:646-715): four documents, two without a brand; no sort field (the greater seq_id first)
    cm = _fixture()["control_missing_values"]
    bptr = np.zeros(5, np.uint64)
    bptr[1:] = np.cumsum([len(x) for x in cm["brand_hashes"]])
    bh = np.array([x[0] for x in cm["brand_hashes"] if x], np.uint32)
    g = T.GpuIndex(0, lib)
    g.set_num_docs(4)
    g.field_create(0, False)
    g.commit()
    wq = T.KwQuery([], sort=((B.SORT_SEQ_ID, 1, 0),), topster_size=250)
    for gmv, key in ((False, "gmv_false"), (True, "gmv_true")):
        g.column_set(1, O.distinct_ids(4, [(bptr, bh)], gmv)[0].view(np.int64))
        h, gh = g.keyword_search_grouped_batch([wq], [(2, 1, 0, int(gmv), 1)], k_stride=500, g_stride=250)
        assert int(gh.n_groups[0]) == len(cm[key])
        for r, want in enumerate(cm[key]):
            assert h.keys[0, r * 2:r * 2 + int(gh.group_size[0, r])].tolist() == want["hits"], (gmv, r)
    g.close()


def _group_order_product(lib):
    """GroupOrderIndependence / UseHighestValueInGroupForOrdering (collection_grouping_test.cpp:510-613) through the library: both passes"""
    from tests.test_oracle_groupby import order_cases
    for name, pts, grp, first_hits in order_cases():
        n = len(pts)
        g = T.GpuIndex(0, lib)
        g.set_num_docs(n)
        g.field_create(0, False)
        g.commit()
        g.column_set(0, np.array(pts, np.int64))
        g.column_set(1, O.distinct_ids(n, [(np.arange(n + 1, dtype=np.uint64), np.array(grp, np.uint32))], True)[0].view(np.int64))
        wq = T.KwQuery([], sort=((B.SORT_INT64_COLUMN, 1, 0),), topster_size=250)
        h, gh = g.keyword_search_grouped_batch([wq], [(10, 1, 0, 1, 1)], k_stride=2500, g_stride=250)
        assert int(gh.n_groups[0]) == 250 and h.keys[0, :int(gh.group_size[0, 0])].tolist() == first_hits, name
        h1, g1 = g.keyword_search_grouped_batch([wq], [(10, 1, 1, 1, 1)], k_stride=250, g_stride=250)
        assert int(g1.n_groups[0]) == 250 and first_hits[0] in h1.keys[0, :250].tolist(), name
        g.close()


def test_group_count_of_300_groups_through_the_library():
    """SortingMoreThanMaxTopsterSize's collection (collection_grouping_test.cpp:876-925): found_docs 1 000, found 300 = the sketch the device built"""
    from tests.test_oracle_groupby import three_hundred_groups
    n, distinct = three_hundred_groups()
    g = T.GpuIndex(0, H.emu_lib_path())
    g.set_num_docs(n)
    g.field_create(0, False)
    g.commit()
    g.column_set(1, distinct.view(np.int64))
    wq = T.KwQuery([], sort=((B.SORT_SEQ_ID, 1, 0),), topster_size=250)
    h, gh = g.keyword_search_grouped_batch([wq], [(2, 1, 1, 1, 1)], k_stride=250, g_stride=250)
    assert int(h.num_matched[0]) == 1000 and int(gh.groups_count[0]) == 300 and int(gh.n_groups[0]) == 250
    g.close()


def test_group_order_known_answers_through_the_library():
    _group_order_product(H.emu_lib_path())


def test_grouping_basics_known_answer_through_the_library():
    _grouping_basics_product(H.emu_lib_path())


def test_random_skewed_wildcard_groupings_match_oracle():
    """q = * over 300 .. 6 000 documents grouped by fields of 1 .. 60 values (whole waves in one group, workgroups that meet a handful of groups, member lists that
    span several scatter workgroups and the chunked path), Topsters that hold fewer groups than exist, both passes, ties in the sort key — seeded"""
    rng = np.random.default_rng(77)
    for case in range(14):
        n = int(rng.choice([300, 1100, 2500, 6000]))
        n_groups = int(rng.choice([1, 2, 3, 9, 60]))
        g = T.GpuIndex(0, H.emu_lib_path())
        g.set_num_docs(n)
        g.field_create(0, False)
        g.commit()
        points = rng.integers(0, int(rng.choice([3, 1000])), size=n).astype(np.int64)            # (few distinct sort keys: ties resolved by the seq_id)
        g.column_set(0, points)
        short = int(rng.integers(0, 40))
        probs = rng.dirichlet(np.ones(n_groups) * 0.5)
        distinct = (rng.choice(n_groups, size=n - short, p=probs).astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        g.column_set(1, distinct.view(np.int64))
        limit = int(rng.choice([1, 3, 7]))
        k = int(rng.choice([2, 50, 250]))
        sort = ((B.SORT_INT64_COLUMN, int(rng.choice([1, -1])), 0), (B.SORT_SEQ_ID, int(rng.choice([1, -1])), 0))
        excl = np.sort(rng.choice(n, size=int(rng.integers(0, n // 4)), replace=False)).astype(np.uint32)
        q = T.KwQuery([], sort=sort, topster_size=k, excluded_ids=excl if excl.size else None)
        gmv = int(rng.integers(0, 2))
        for first_pass in (1, 0):
            h, gh = g.keyword_search_grouped_batch([q], [(limit, 1, first_pass, gmv, 1)], k_stride=k * limit, g_stride=max(k, 1), want_registers=True)
            check_query(h, gh, 0, oracle_grouped_wildcard(q, n, points, distinct, limit, bool(first_pass), bool(gmv)), bool(first_pass), limit, "case %d" % case)
        g.close()


@pytest.mark.parametrize("n_values", [1, 2, 7])
def test_grouped_candidate_combinations_over_few_groups(world, n_values):
    """the candidate fold when the group_by field has one, two or seven values: the deduplicated records of a second pass all land in a handful of table slots
    (wave groups + workgroup LDS tables), frequent tokens so that a user query holds thousands of records"""
    orc, g, _, _, _ = world
    few = (np.arange(3000, dtype=np.uint64) * np.uint64(2654435761) % np.uint64(n_values)) * np.uint64(0x100000001B3) + np.uint64(11)
    g.column_set(4, few.view(np.int64))
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    users = [[[1], [2], [1, 2], [3], [1]], [[2, 3], [1, 3], [4]], [[5], [6], [7], [8], [9], [10]]]
    limit = 4
    combos = [[T.KwQuery(c, sort=sort, topster_size=250, total_cost=int(j > 0)) for j, c in enumerate(cs)] for cs in users]
    for first_pass in (True, False):
        h, gh, qidx, ids = g.keyword_search_grouped_candidates_batch(combos, [(limit, 4, int(first_pass), 0, 0)] * len(users), k_stride=1000, g_stride=250, want_ids=True, want_registers=True)
        assert (h.status == 0).all()
        for u, cs in enumerate(combos):
            ref, rqi = orc.search_candidates_grouped([H.oracle_query(orc, q) for q in cs], few, limit, first_pass, ids_cap=1 << 20)
            check_query(h, gh, u, ref, first_pass, limit, "few groups %d u%d" % (n_values, u))
            assert np.array_equal(ids[u], ref.result_ids), u
            if not first_pass:
                for r in range(int(gh.n_groups[u])):
                    n = int(ref.group_size[r])
                    assert np.array_equal(qidx[u, r * limit:r * limit + n], rqi[ref.begin[r]:ref.begin[r + 1]].astype(np.uint32)), (u, r)
        assert int(h.num_matched.sum()) > 1000
