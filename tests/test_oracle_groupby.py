"""oracle/group_topster.h (the restated distinct Topster, LogLogBeta, wyhash, hash_combine) against the reference.

Pins: TopsterTest.DistinctIntValues (/root/reference/test/topster_test.cpp:181-262) as literal vectors, and — where oracle/_ref/libref_topster.so
exists (compiled by oracle/Makefile from the reference's OWN include/topster.h, loglogbeta.h, wyhash_v5.h where they lie; it travels with the
snapshot) — random KV streams through both collectors: add()'s return values, the first pass' heap array, the second pass' populate_result_kvs
order, getGroupsCount(). Where _ref is missing the golden file tests/golden/group_topster_vectors.json (made from _ref by
tests/golden/make_group_topster_vectors.py) stands in.
"""
import json
import os
import numpy as np
import pytest
from oracle import oracle_py as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "group_topster_vectors.json")

# TopsterTest.DistinctIntValues: {query_index, distinct_key, match_score, primary_attr, secondary_attr}, key = i + 100
DATA = [(0, 1, 11, 20, 30), (0, 1, 12, 20, 32), (0, 2, 4, 20, 30), (2, 3, 7, 20, 30), (0, 4, 14, 20, 30), (1, 5, 9, 20, 30), (1, 5, 10, 20, 32),
        (1, 5, 9, 20, 30), (0, 6, 6, 20, 30), (2, 7, 6, 22, 30), (2, 7, 6, 22, 30), (1, 8, 9, 20, 30), (0, 9, 8, 20, 30), (3, 10, 5, 20, 30)]


def _distinct_int_values():
    return [i + 100 for i in range(14)], [d[1] for d in DATA], [[d[2], d[3], d[4]] for d in DATA]


def test_distinct_int_values_first_pass_golden():
    keys, dk, sc = _distinct_int_values()
    ret, g = O.group_topster_run(7, 2, True, keys, dk, sc)
    # topster_test.cpp:249-258: the heap array as it lies (sort() is a no-op), group_kv_map empty, loglog cardinality 10
    assert list(g.distinct_key) == [7, 5, 3, 4, 1, 9, 8]
    assert list(g.keys) == [110, 106, 103, 104, 101, 112, 111]
    assert g.groups_count == 10 and g.groups_exact == 10


def test_distinct_int_values_second_pass_golden():
    keys, dk, sc = _distinct_int_values()
    ret, g = O.group_topster_run(5, 2, False, keys, dk, sc)
    # topster_test.cpp:217-235: group 1 holds {12, 11}, group 5 holds {10, 9}; every KV goes to its group's Topster (the outer heap stays empty)
    groups = {int(g.distinct_key[i]): (list(g.keys[g.begin[i]:g.begin[i + 1]]), list(g.scores[g.begin[i]:g.begin[i + 1], 0])) for i in range(g.n_groups)}
    assert groups[1] == ([101, 100], [12, 11])
    assert groups[5][1] == [10, 9] and groups[5][0][0] == 106
    # populate_result_kvs: the five best groups by their head, best first
    assert list(g.distinct_key) == [4, 1, 5, 8, 9]
    assert list(ret) == [1] * 14


def _random_stream(rng, n, n_groups, score_range, dup_keys):
    dk = rng.integers(0, n_groups, n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(1)
    if dup_keys:
        keys = rng.integers(0, max(2, n // 2), n).astype(np.uint64)
        # a document belongs to ONE group: derive the group from the key
        dk = (keys % np.uint64(n_groups)) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(1)
    else:
        keys = rng.permutation(n * 3)[:n].astype(np.uint64)
    sc = rng.integers(-score_range, score_range + 1, (n, 3)).astype(np.int64)
    return keys, dk, sc


def _compare(R, cap, distinct, first_pass, keys, dk, sc):
    ret, g = O.group_topster_run(cap, distinct, first_pass, keys, dk, sc)
    rret, gsize, rdk, rkeys, rsc, rcount = O.ref_group_topster_run(R, cap, distinct, first_pass, keys, dk, sc)
    assert np.array_equal(ret, rret)
    assert np.array_equal(g.group_size, gsize) and np.array_equal(g.distinct_key, rdk)
    assert np.array_equal(g.keys, rkeys) and np.array_equal(g.scores, rsc)
    if first_pass:
        assert g.groups_count == rcount


def test_group_topster_equals_reference_topster_on_random_streams():
    R = O.ref_topster_lib()
    if R is None:
        pytest.skip("oracle/_ref/libref_topster.so not built (no /root/reference here); the golden file covers it")
    rng = np.random.default_rng(5)
    n_cases = 0
    for cap in (1, 2, 5, 16, 250):
        for distinct in (1, 2, 3, 7):
            for first_pass in (True, False):
                for n, n_groups, rng_s, dup in ((0, 1, 1, False), (1, 1, 1, False), (30, 4, 2, False), (200, 40, 3, False), (200, 500, 1000, False),
                                                (1500, 300, 5, False), (300, 30, 2, True)):
                    keys, dk, sc = _random_stream(rng, n, n_groups, rng_s, dup)
                    _compare(R, cap, distinct, first_pass, keys, dk, sc)
                    n_cases += 1
    assert n_cases == 5 * 4 * 2 * 7


def test_wyhash_loglog_hash_combine_equal_reference():
    R = O.ref_topster_lib()
    L = O.lib()
    rng = np.random.default_rng(2)
    if R is not None:
        for n in list(range(0, 70)) + [127, 128, 129, 200]:
            for _ in range(10):
                s = bytes(int(x) for x in rng.integers(1, 256, n))
                assert L.orc_hash_wy(s, len(s)) == R.ref_hash_wy(s, len(s))
        for _ in range(1000):
            a, b = int(rng.integers(0, 2**63)) * 2 + 1, int(rng.integers(0, 2**63))
            assert L.orc_hash_combine(a, b) == R.ref_hash_combine(a, b)
        for n in (0, 1, 10, 1000, 20000, 200000):
            dk = rng.integers(0, 2**63, n).astype(np.uint64)
            assert L.orc_loglog_of_keys(dk.ctypes.data, n, None) == R.ref_loglog_of_keys(dk.ctypes.data, n)
    # known answers (made from _ref by tests/golden/make_group_topster_vectors.py): the decimal strings the first pass hashes
    gold = json.load(open(GOLD))
    for s, h in gold["hash_wy"]:
        assert L.orc_hash_wy(s.encode(), len(s)) == int(h)
    for a, b, h in gold["hash_combine"]:
        assert L.orc_hash_combine(int(a), int(b)) == int(h)


def test_group_topster_golden_file():
    gold = json.load(open(GOLD))
    for case in gold["streams"]:
        keys = np.array(case["keys"], np.uint64); dk = np.array([int(x) for x in case["dkeys"]], np.uint64); sc = np.array(case["scores"], np.int64).reshape(-1, 3)
        ret, g = O.group_topster_run(case["capacity"], case["distinct"], case["first_pass"], keys, dk, sc)
        assert list(ret) == case["ret"]
        assert [int(x) for x in g.group_size] == case["group_size"] and [str(int(x)) for x in g.distinct_key] == case["distinct_key"]
        assert [int(x) for x in g.keys] == case["out_keys"]
        if case["first_pass"]:
            assert g.groups_count == case["groups_count"]


def _grouping_basics():
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "facet_group_range.json")))["grouping_basics"]
    n = len(fx["sizes"])
    ptr = np.arange(n + 1, dtype=np.uint64)
    distinct = O.distinct_ids(n, [(ptr, np.array([h[0] for h in fx["size_hashes"]], np.uint32))], False)[0]
    return fx, n, distinct


def test_grouping_basics_known_answer_of_the_reference():
    """CollectionGroupingTest.GroupingBasics (/root/reference/test/collection_grouping_test.cpp:71-96): q = *, group_by size, group_limit 2, default sort (rating desc):
    three groups in the order 11, 10, 12 with the hits 5,1 / 4,3 / 2,8 — the restated distinct Topster's second pass + populate_result_kvs order"""
    fx, n, distinct = _grouping_basics()
    sc = np.zeros((n, 3), np.int64)
    sc[:, 0] = fx["rating_keys"]
    ret, gh = O.group_topster_run(250, 2, False, np.arange(n, dtype=np.uint64), distinct, sc)
    assert gh.n_groups == len(fx["expected_groups"])
    for r, e in enumerate(fx["expected_groups"]):
        a, b = int(gh.begin[r]), int(gh.begin[r + 1])
        assert gh.keys[a:b].tolist() == e["hits"], (r, gh.keys[a:b])
        assert {fx["sizes"][int(k)] for k in gh.keys[a:b]} == {e["size"]}
    # first pass: one KV per group (its greatest). getGroupsCount() is LogLogBeta's TRUNCATED estimate — 2 for these three keys, from the reference's own Topster
    # too —; the response's `found` = max(that, groups returned) (Index::run_search, src/index.cpp:2766-2770) = 3, as the test asserts (:74)
    ret, g1 = O.group_topster_run(250, 2, True, np.arange(n, dtype=np.uint64), distinct, sc)
    assert sorted(g1.keys.tolist()) == sorted(e["hits"][0] for e in fx["expected_groups"])
    assert g1.groups_count == 2 and max(g1.groups_count, gh.n_groups) == 3
    R = O.ref_topster_lib()
    if R is not None:
        assert O.ref_group_topster_run(R, 250, 2, True, np.arange(n, dtype=np.uint64), distinct, sc)[-1] == 2


def _by_rating():
    fx, n, _ = _grouping_basics()
    ptr = np.arange(n + 1, dtype=np.uint64)
    distinct = O.distinct_ids(n, [(ptr, np.array([h[0] for h in fx["rating_hashes"]], np.uint32))], False)[0]
    return fx, n, distinct


def test_grouping_basics_by_rating_known_answer_of_the_reference():
    """the second request of GroupingBasics (collection_grouping_test.cpp:112-148): group_by rating, sort_by size DESC, group_limit 2 -> 7 groups; groups 0, 1, 5, 6 as asserted there"""
    fx, n, distinct = _by_rating()
    sc = np.zeros((n, 3), np.int64)
    sc[:, 0] = fx["sizes"]
    ret, gh = O.group_topster_run(250, 2, False, np.arange(n, dtype=np.uint64), distinct, sc)
    e = fx["by_rating_expected"]
    assert gh.n_groups == e["n_groups"]
    for r, want in e["groups"].items():
        a, b = int(gh.begin[int(r)]), int(gh.begin[int(r) + 1])
        assert gh.keys[a:b].tolist() == want["hits"], (r, gh.keys[a:b])


def _compound_key():
    fx, n, _ = _grouping_basics()
    ck = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "facet_group_range.json")))["compound_key"]
    ptr = np.arange(n + 1, dtype=np.uint64)
    bptr = np.zeros(n + 1, np.uint64)
    bptr[1:] = np.cumsum([len(h) for h in ck["brand_hashes"]])
    brand = (bptr, np.array([h[0] for h in ck["brand_hashes"] if h], np.uint32))
    distinct, has_value = O.distinct_ids(n, [(ptr, np.array([h[0] for h in fx["size_hashes"]], np.uint32)), brand], True)      # group_missing_values = true (the request's default)
    return fx, ck, n, distinct, brand


def test_grouping_compound_key_known_answer_of_the_reference():
    """CollectionGroupingTest.GroupingCompoundKey (collection_grouping_test.cpp:150-215): group_by size + brand with an optional brand -> 10 groups; groups 0, 1, 2, 5 as asserted;
    the facet counts of `brand` under that grouping"""
    fx, ck, n, distinct, brand = _compound_key()
    sc = np.zeros((n, 3), np.int64)
    sc[:, 0] = fx["rating_keys"]
    ret, gh = O.group_topster_run(250, 2, False, np.arange(n, dtype=np.uint64), distinct, sc)
    assert gh.n_groups == ck["n_groups"]
    for r, want in ck["groups"].items():
        a, b = int(gh.begin[int(r)]), int(gh.begin[int(r) + 1])
        assert gh.keys[a:b].tolist() == want["hits"], (r, gh.keys[a:b])
    orc = O.OracleIndex(1, 1)
    orc.facet_set(0, *brand)
    k, c, d, p, nn = orc.facet_count_ex(0, np.arange(n, dtype=np.uint32), distinct_ids=distinct, group_missing_values=True)
    got = {int(a): int(b) for a, b in zip(k, c)}
    assert {name: got[i] for name, i in ck["brand_ids"].items()} == ck["expected_grouped_facets"]


def _fixture():
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "facet_group_range.json")))


def _by_brand(gmv=True):
    fx, ck, n, _, brand = _compound_key()
    distinct, has_value = O.distinct_ids(n, [brand], gmv)
    return fx, ck, n, distinct, brand


def test_group_limit_of_one_and_control_missing_values_known_answers_of_the_reference():
    """CollectionGroupingTest.GroupingWithGropLimitOfOne (collection_grouping_test.cpp:372-411) and ControlMissingValues (:646-715)"""
    one = _fixture()["group_limit_of_one"]
    fx, ck, n, distinct, brand = _by_brand(True)
    sc = np.zeros((n, 3), np.int64)
    sc[:, 0] = fx["rating_keys"]
    ret, gh = O.group_topster_run(250, 1, False, np.arange(n, dtype=np.uint64), distinct, sc)
    assert gh.n_groups == one["n_groups"]
    for r, want in enumerate(one["groups"]):
        assert gh.keys[int(gh.begin[r]):int(gh.begin[r + 1])].tolist() == want["hits"]
    orc = O.OracleIndex(1, 1)
    orc.facet_set(0, *brand)
    k, c, d, p, nn = orc.facet_count_ex(0, np.arange(n, dtype=np.uint32), distinct_ids=distinct, group_missing_values=True)
    got = {int(a): int(b) for a, b in zip(k, c)}
    assert {name: got[i] for name, i in ck["brand_ids"].items()} == one["expected_grouped_facets"]
    cm = _fixture()["control_missing_values"]
    bptr = np.zeros(5, np.uint64)
    bptr[1:] = np.cumsum([len(h) for h in cm["brand_hashes"]])
    bh = np.array([h[0] for h in cm["brand_hashes"] if h], np.uint32)
    for gmv, key in ((False, "gmv_false"), (True, "gmv_true")):
        distinct = O.distinct_ids(4, [(bptr, bh)], gmv)[0]
        ret, gh = O.group_topster_run(250, 2, False, np.arange(4, dtype=np.uint64), distinct, np.zeros((4, 3), np.int64))       # equal scores: the greater seq_id first
        assert gh.n_groups == len(cm[key])
        for r, want in enumerate(cm[key]):
            assert gh.keys[int(gh.begin[r]):int(gh.begin[r + 1])].tolist() == want["hits"], (gmv, r)


def order_cases():
    """GroupOrderIndependence (collection_grouping_test.cpp:510-561) and UseHighestValueInGroupForOrdering (:563-613): more groups than the Topster holds (250),
    sort_by points DESC, group_limit 10 — (points, facet id of `group` per document (handed out in order of first appearance), expected first group's hits)"""
    a_pts = [100 + i for i in range(256)] + [50, 500]
    a_grp = [i + 1 for i in range(256)] + [257, 257]
    b_pts = [100 + i for i in range(250)] + [50, 60]
    b_grp = [i + 1 for i in range(250)] + [250, 251]
    return [("GroupOrderIndependence", a_pts, a_grp, [257, 256]), ("UseHighestValueInGroupForOrdering", b_pts, b_grp, [249, 250])]


def test_group_order_known_answers_of_the_reference():
    for name, pts, grp, first_hits in order_cases():
        n = len(pts)
        distinct = O.distinct_ids(n, [(np.arange(n + 1, dtype=np.uint64), np.array(grp, np.uint32))], True)[0]
        sc = np.zeros((n, 3), np.int64)
        sc[:, 0] = pts
        ret, gh = O.group_topster_run(250, 10, False, np.arange(n, dtype=np.uint64), distinct, sc)
        assert gh.n_groups == 250 and gh.keys[int(gh.begin[0]):int(gh.begin[1])].tolist() == first_hits, name
        # the first pass keeps that group too (its greatest KV), whatever the order the documents arrive in
        ret, g1 = O.group_topster_run(250, 10, True, np.arange(n, dtype=np.uint64), distinct, sc)
        assert first_hits[0] in g1.keys.tolist(), name


def three_hundred_groups():
    """the collection of SortingMoreThanMaxTopsterSize (collection_grouping_test.cpp:876-925): 150 sizes x 4 documents, 100 x 3, 50 x 2 = 1 000 documents in 300 groups"""
    sizes = [i for i in range(150) for _ in range(4)] + [i for i in range(150, 250) for _ in range(3)] + [i for i in range(250, 300) for _ in range(2)]
    n = len(sizes)
    return n, O.distinct_ids(n, [(np.arange(n + 1, dtype=np.uint64), np.array(sizes, np.uint32))], True)[0]


def test_group_count_of_300_groups_is_the_reference_found():
    """`found` = 300 there (:929, :941, :957, :969) with a Topster of 250: it can only come from getGroupsCount() — get_distinct_id -> std::to_string -> wyhash -> LogLogBeta::cardinality()
    must give exactly 300 for these keys"""
    n, distinct = three_hundred_groups()
    ret, g1 = O.group_topster_run(250, 2, True, np.arange(n, dtype=np.uint64), distinct, np.zeros((n, 3), np.int64))
    assert g1.groups_count == 300 and g1.n_groups == 250
    R = O.ref_topster_lib()
    if R is not None:
        assert O.ref_group_topster_run(R, 250, 2, True, np.arange(n, dtype=np.uint64), distinct, np.zeros((n, 3), np.int64))[-1] == 300

