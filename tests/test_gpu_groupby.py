"""`-m gpu`: group_by (tsgpu_keyword_search_grouped_batch) on a real MI355X through the C-ABI, against the oracle's restated distinct Topster
(oracle/group_topster.h, pinned to the reference's own topster.h by tests/test_oracle_groupby.py): the emulator-tier bodies re-run on libtsgpu.so,
plus a 2M-document collection where the group tables are hammered from every XCD at once (hundreds of thousands of matched ids per query, a few
dozen to a million groups) — groups, their order, every KV, groups_processed, the LogLogBeta registers and getGroupsCount()."""
import numpy as np
import pytest

import typesense_amd as T
from typesense_amd import _lib as B
from oracle import oracle_py as O
from tests import helpers as H
from tests import test_emu_groupby as E
from tests.test_gpu_keyword import Corpus

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _real_library(monkeypatch):
    monkeypatch.setattr(H, "emu_lib_path", lambda *a, **k: H.gpu_lib_path())


@pytest.fixture(scope="module")
def world():
    docs = H.zipf_docs(3000, 300, 12, seed=1)
    orc, g = H.build_pair(docs, H.gpu_lib_path())
    distinct, has_value = E.group_column(3000, seed=3)
    g.column_set(E.GROUP_COL, distinct.view(np.int64))
    yield orc, g, docs, distinct, has_value
    g.close()


@pytest.mark.parametrize("first_pass", [True, False])
def test_grouped_keyword_equals_oracle(world, first_pass):
    E.test_grouped_keyword_equals_oracle(world, first_pass)


@pytest.mark.parametrize("first_pass", [True, False])
def test_grouped_small_topster_selects_the_best_groups(world, first_pass):
    E.test_grouped_small_topster_selects_the_best_groups(world, first_pass)


@pytest.mark.parametrize("first_pass", [True, False])
def test_grouped_with_filter_excluded_dropped_and_flags(world, first_pass):
    E.test_grouped_with_filter_excluded_dropped_and_flags(world, first_pass)


@pytest.mark.parametrize("first_pass", [True, False])
def test_grouped_wildcard_and_many_groups(world, first_pass):
    E.test_grouped_wildcard_and_many_groups(world, first_pass)


@pytest.mark.parametrize("first_pass", [True, False])
def test_grouped_candidate_combinations_fold_like_the_shared_collector(world, first_pass):
    E.test_grouped_candidate_combinations_fold_like_the_shared_collector(world, first_pass)


def test_grouped_two_fields_multi_field_and_bad_queries(world):
    E.test_grouped_two_fields_arrays_and_missing_ids(world)
    E.test_grouped_queries_of_more_than_three_tokens(world)
    E.test_group_count_sketch_for_keys_of_every_printed_length(world)
    E.test_grouped_big_output_arrays_take_the_direct_delivery(world)
    E.test_grouped_calls_from_concurrent_threads_are_coalesced_and_keep_their_own_results(world)
    E.test_grouped_multi_field_query()
    E.test_grouped_big_groups_are_cut_into_chunks()
    E.test_grouped_string_array_fields()
    E.test_grouped_bad_queries_do_not_disturb_their_neighbours(world)


@pytest.fixture(scope="module")
def c2m():
    c = Corpus(2_000_000, 50_000, 24, seed=7)
    yield c
    c.g.close()


def _distinct_columns(n_docs):
    ids = np.arange(n_docs, dtype=np.uint64)
    few = (ids * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(58)                     # 64 groups
    few = np.array([O.lib().orc_hash_combine(1, int(v)) for v in range(64)], np.uint64)[few.astype(np.int64)]
    many = ((ids * np.uint64(2654435761)) % np.uint64(300_000)) * np.uint64(0x100000001B3) + np.uint64(7)     # ~300 K groups
    own = ids.copy()                                                                  # every document its own group (no value anywhere, group_missing_values = false)
    return few, many, own


def test_grouped_at_2m_documents_keyword_and_wildcard(c2m):
    c = c2m
    few, many, own = _distinct_columns(c.n_docs)
    for col, arr in ((1, few), (2, many), (3, own)):
        c.g.column_set(col, arr.view(np.int64))
    # frequent terms: 10^5 .. 10^6 matched ids per query
    qs = [T.KwQuery([2, 3], sort=((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0)), topster_size=250),
          T.KwQuery([5, 1, 9], sort=((B.SORT_INT64_COLUMN, -1, 0), (B.SORT_TEXT_MATCH, 1, 0), (B.SORT_SEQ_ID, 1, 0)), topster_size=100),
          T.KwQuery([4], sort=((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_SEQ_ID, -1, 0)), topster_size=1000)]
    matched = 0
    for col, arr in ((1, few), (2, many), (3, own)):
        for first_pass in (1, 0):
            groups = [(3, col, first_pass, 0, 0)] * len(qs)
            h, gh = c.g.keyword_search_grouped_batch(qs, groups, k_stride=3000, g_stride=1000, want_registers=True)
            for i, q in enumerate(qs):
                c.need(q.tokens)
                ref = c.orc.search_keyword_grouped(H.oracle_query(c.orc, q), arr, 3, bool(first_pass), group_cap=4096, kv_cap=16384, ids_cap=0)
                E.check_query(h, gh, i, ref, bool(first_pass), 3, "2M col %d" % col)
                matched += int(h.num_matched[i])
    assert matched > 2_000_000
    # q = *: two million ids, one launch
    points = c.pts
    wq = T.KwQuery([], sort=((B.SORT_INT64_COLUMN, 1, 0), (B.SORT_SEQ_ID, 1, 0)), topster_size=250)
    for col, arr in ((1, few), (2, many)):
        for first_pass in (1, 0):
            h, gh = c.g.keyword_search_grouped_batch([wq], [(3, col, first_pass, 0, 1)], k_stride=750, g_stride=250, want_registers=True)
            ids = np.arange(c.n_docs, dtype=np.uint64)
            sc = np.zeros((c.n_docs, 3), np.int64)
            sc[:, 0] = points
            sc[:, 1] = ids.astype(np.int64)
            ret, ref = O.group_topster_run(250, 3, bool(first_pass), ids, arr, sc)
            ref.num_keyword_matches = c.n_docs
            E.check_query(h, gh, 0, ref, bool(first_pass), 3, "2M wildcard col %d" % col)


def test_grouped_candidate_combinations_at_2m_documents(c2m):
    """ten candidate combinations per user query (neighbouring term ranks, as the ART walk returns them) over the 2M-document collection, both passes"""
    c = c2m
    few, many, own = _distinct_columns(c.n_docs)
    c.g.column_set(2, many.view(np.int64))
    rng = np.random.default_rng(5)
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    users = []
    for u in range(6):
        base = rng.choice(np.arange(20, 400), size=2, replace=False)
        cs = [base.copy()]
        while len(cs) < 10:
            x = cs[int(rng.integers(0, len(cs)))].copy()
            x[int(rng.integers(0, 2))] += int(rng.integers(1, 15))
            if x[0] != x[1] and not any(np.array_equal(x, y) for y in cs):
                cs.append(x)
        users.append(cs)
    combos = [[T.KwQuery(cc, sort=sort, topster_size=250, total_cost=int(j > 0)) for j, cc in enumerate(cs)] for cs in users]
    for first_pass in (True, False):
        h, gh, qidx, ids = c.g.keyword_search_grouped_candidates_batch(combos, [(3, 2, int(first_pass), 0, 0)] * len(users), k_stride=750, g_stride=250, want_ids=True, want_registers=True)
        assert (h.status == 0).all()
        for u, cs in enumerate(combos):
            for q in cs:
                c.need(q.tokens)
            ref, rqi = c.orc.search_candidates_grouped([H.oracle_query(c.orc, q) for q in cs], many, 3, first_pass, group_cap=4096, kv_cap=16384, ids_cap=1 << 22)
            E.check_query(h, gh, u, ref, first_pass, 3, "2M candidates u%d" % u)
            assert np.array_equal(ids[u], ref.result_ids)
            if not first_pass:
                for r in range(int(gh.n_groups[u])):
                    n = int(ref.group_size[r])
                    assert np.array_equal(qidx[u, r * 3:r * 3 + n], rqi[ref.begin[r]:ref.begin[r + 1]].astype(np.uint32))
    assert int(gh.n_groups.sum()) > 100
    # a three-value group_by field: every group far beyond 4096 members — the chunked second pass over deduplicated records (query_index through the document table)
    tiny = (np.arange(c.n_docs, dtype=np.uint64) % np.uint64(3)) + np.uint64(77)
    c.g.column_set(3, tiny.view(np.int64))
    for first_pass in (True, False):
        h, gh, qidx, ids = c.g.keyword_search_grouped_candidates_batch(combos[:3], [(5, 3, int(first_pass), 0, 0)] * 3, k_stride=1250, g_stride=250, want_ids=True, want_registers=True)
        for u, cs in enumerate(combos[:3]):
            ref, rqi = c.orc.search_candidates_grouped([H.oracle_query(c.orc, q) for q in cs], tiny, 5, first_pass, group_cap=4096, kv_cap=16384, ids_cap=1 << 22)
            E.check_query(h, gh, u, ref, first_pass, 5, "2M candidates, three groups u%d" % u)
            if not first_pass:
                for r in range(int(gh.n_groups[u])):
                    n = int(ref.group_size[r])
                    assert np.array_equal(qidx[u, r * 5:r * 5 + n], rqi[ref.begin[r]:ref.begin[r + 1]].astype(np.uint32))
