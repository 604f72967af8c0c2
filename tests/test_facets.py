"""Facet counting over matched ids (Index::do_facets hash-index branch, src/index.cpp:1659-1771).
CPU tier: the oracle against the reference's own known answer (CollectionFacetingTest.FacetCounts) and the emulator build of the
product against the oracle; `-m gpu`: libtsgpu.so against the oracle at size, fed by the per-call id lists of a keyword batch."""
import json
import os

import numpy as np
import pytest

import typesense_amd as T
from typesense_amd import _lib as B
from oracle import oracle_py as O
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_reproduces_reference_facet_counts():
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "facet_counts_tags.json")))
    ptr = np.zeros(len(fx["docs"]) + 1, np.uint64)
    ptr[1:] = np.cumsum([len(d) for d in fx["docs"]])
    hashes = np.array([h for d in fx["docs"] for h in d], np.uint32)
    orc = O.OracleIndex(1, 1)
    orc.facet_set(0, ptr, hashes)
    h, c, d, p, n = orc.facet_count(0, np.array(fx["result_ids"], np.uint32))
    got = {int(a): int(b) for a, b in zip(h, c)}
    assert n == len(fx["expected_counts"])
    for value, cnt in fx["expected_counts"].items():
        assert got[fx["values"][value]] == cnt, value


def _csr(rows):
    ptr = np.zeros(len(rows) + 1, np.uint64)
    ptr[1:] = np.cumsum([len(r) for r in rows])
    return ptr, np.array([x for r in rows for x in r], np.uint32)


def test_oracle_reproduces_reference_facet_stats_and_value_index_counts():
    """the reference's own numbers: numeric stats (collection_faceting_test.cpp:240-296) and the value-index counts
    (collection_optimized_faceting_test.cpp:60-110) on test/numeric_array_documents.jsonl"""
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "facet_stats_values.json")))
    ids = np.array(fx["result_ids"], np.uint32)
    orc = O.OracleIndex(1, 1)
    orc.facet_set(0, *_csr(fx["rating_bits"]))
    mn, mx, sm, cnt = orc.facet_stats(0, ids, B.FACET_FLOAT)
    e = fx["rating_expected"]
    assert (mn, mx, sm, cnt) == (e["min"], e["max"], e["sum"], e["count"]) and sm / cnt == e["avg"]          # doubles, bit for bit
    orc.facet_set(1, *_csr(fx["timestamps_hashes"]))
    m = (np.array([h for h, _ in fx["timestamps_map"]], np.uint32), np.array([v for _, v in fx["timestamps_map"]], np.int64))
    mn, mx, sm, cnt = orc.facet_stats(1, ids, B.FACET_INT64, int64_map=m)
    e = fx["timestamps_expected"]
    assert (mn, mx, sm, cnt) == (e["min"], e["max"], e["sum"], e["count"]) and sm / cnt == pytest.approx(e["avg"], rel=1e-7)   # (ASSERT_FLOAT_EQ in the reference)
    vp, vi = _csr(fx["tags_value_ids"])
    orc.facet_value_set(2, vp, vi, [len(r) for r in fx["tags_value_ids"]])
    v, c, d = orc.facet_value_count(2, ids, max_facets=20)
    assert [[fx["tags_values"][int(a)], int(b)] for a, b in zip(v, c)] == fx["tags_expected"]


def test_oracle_reproduces_reference_grouped_and_range_facet_counts():
    """the reference's own numbers: facet counts of a grouped search (collection_grouping_test.cpp:71-110) and range facets, plain and
    grouped (collection_faceting_test.cpp:1500-1590, 3419-3524)"""
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "facet_group_range.json")))
    g = fx["grouping_basics"]
    orc = O.OracleIndex(1, 1)
    orc.facet_set(0, *_csr(g["brand_hashes"]))
    distinct = O.distinct_ids(len(g["size_hashes"]), [_csr(g["size_hashes"])], False)[0]
    k, c, d, p, n = orc.facet_count_ex(0, np.array(g["result_ids"], np.uint32), distinct_ids=distinct)
    got = {int(a): int(b) for a, b in zip(k, c)}
    assert {v: got[hh] for v, hh in g["brand_values"].items()} == g["expected_grouped"]
    for name in ("range_facet_test", "range_facet_with_group_by"):
        r = fx[name]
        hashes = [[7 + v % 5] for v in r["visitors"]]                                   # (the facet hash index of `visitors`: one hash per document)
        orc.facet_set(1, *_csr(hashes))
        ranges = [tuple(x) for x in r["ranges"]]
        def counts(ids, **kw):
            k, c, d, p, n = orc.facet_count_ex(1, np.array(ids, np.uint32), ranges=ranges, doc_vals=np.array(r["visitors"], np.int64), **kw)
            m = {int(a): int(b) for a, b in zip(k.view(np.int64), c)}
            return [m.get(up, 0) for up, lo in ranges]
        assert counts(r["karnataka_ids"]) == r["karnataka_expected"]
        if "gujarat_ids" in r:
            assert counts(r["gujarat_ids"]) == r["gujarat_expected"]
        if "all_ids" in r:
            distinct = O.distinct_ids(5, [_csr(r["rating_hashes"])], False)[0]
            assert counts(r["all_ids"], distinct_ids=distinct) == r["all_grouped_expected"]


def _float_range_cases():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "facet_group_range.json")))["float_ranges"]


def test_oracle_reproduces_reference_float_and_open_ended_range_facets():
    """RangeFacetsFloatRange / RangeFacetsMinMaxRange / RangeFacetRangeNegativeRanges (collection_faceting_test.cpp:1839-2043): float_to_int64_t keys, open bounds"""
    orc = O.OracleIndex(1, 1)
    for case in _float_range_cases():
        n = len(case["vals"])
        orc.facet_set(0, *_csr([[i + 1] for i in range(n)]))
        ranges = [tuple(x) for x in case["ranges"]]
        k, c, d, p, nn = orc.facet_count_ex(0, np.arange(n, dtype=np.uint32), ranges=ranges, doc_vals=np.array(case["vals"], np.int64))
        m = {int(a): int(b) for a, b in zip(k.view(np.int64), c)}
        assert [m.get(up, 0) for up, lo in ranges] == case["expected"], case["test"]


def _float_ranges_product(lib):
    g = T.GpuIndex(0, lib)
    for case in _float_range_cases():
        n = len(case["vals"])
        g.set_num_docs(n)
        g.facet_set(0, *_csr([[i + 1] for i in range(n)]))
        g.column_set(3, np.array(case["vals"], np.int64))
        got = g.facet_range_count_batch(0, 3, [tuple(x) for x in case["ranges"]], [np.arange(n, dtype=np.uint32)])
        assert got[0].tolist() == case["expected"], case["test"]
    g.close()


def test_product_reproduces_reference_float_and_open_ended_range_facets_emulator():
    _float_ranges_product(H.emu_lib_path())


@pytest.mark.gpu
def test_product_reproduces_reference_float_and_open_ended_range_facets_gpu():
    _float_ranges_product(H.gpu_lib_path())


def _grouped_and_ranges(lib, n_docs, n_values, seed=5):
    """the grouped and the range forms of the walk, product vs oracle: array and scalar fields, sampling, facet query, missing sort-index entries,
    few and many groups, distinct ids that differ only above bit 32, range ids that agree in their low 32 bits"""
    rng = np.random.default_rng(seed)
    g = T.GpuIndex(0, lib)
    orc = O.OracleIndex(1, 1)
    g.set_num_docs(n_docs)
    lists = [np.sort(rng.choice(n_docs + 50, size=s, replace=False)).astype(np.uint32) for s in (1, 9, 400, min(n_docs, 6000))] + [np.zeros(0, np.uint32), np.arange(n_docs, dtype=np.uint32)]
    for array in (True, False):
        ptr, hashes = _random_facet_index(rng, n_docs, n_values, array=array)
        g.facet_set(0, ptr, hashes)
        orc.facet_set(0, ptr, hashes)
        for n_groups, short in ((7, 0), (n_docs // 3, 40)):
            distinct = rng.integers(0, n_groups, size=n_docs - short).astype(np.uint64) * np.uint64(0x100000001) + np.uint64(3)   # (the high half repeats the low one: truncation merges nothing it should not)
            if n_groups == 7:
                distinct[::2] += np.uint64(1 << 40)                        # equal low halves, different ids: ONE group for hash_groups
            g.column_set(2, distinct.view(np.int64))
            for gmv in (False, True):
                for kw in ({}, {"sample_mod": 3}, {"allowed_hashes": np.unique(hashes)[::2]}):
                    got = g.facet_count_batch(0, lists, cap=8192, group_column=2, group_missing_values=gmv, **kw)
                    for q, ids in enumerate(lists):
                        k, c, d, p, n = orc.facet_count_ex(0, ids, distinct_ids=distinct, group_missing_values=gmv, **kw)
                        gh, gc, gd, gp, gn = got[q]
                        assert gn == n and np.array_equal(gh, k.astype(np.uint32)) and np.array_equal(gc, c) and np.array_equal(gd, d) and np.array_equal(gp, p), (array, n_groups, gmv, q)
            # ranges over a sort-index column that is shorter than the collection (INT64_MAX beyond it), negative values, an open top, a gap, a value on a bound
            vals = rng.integers(-1000, 5000, size=n_docs - 25).astype(np.int64)
            vals[::17] = 1000
            g.column_set(3, vals)
            big = 1 << 32
            for ranges in ([(0, -500), (1000, 0), (1001, 1000), (3000, 2000), (np.iinfo(np.int64).max, 4000)],
                           [(100, -1000), (100 + big, 100), (100 + 2 * big, 4000)],       # range ids equal in their low 32 bits: one set of groups (hash_groups' uint32 key)
                           [(2500, 2499)]):
                for kw in ({}, {"sample_mod": 4}):
                    for grouped in (False, True):
                        got = g.facet_range_count_batch(0, 3, ranges, lists, group_column=2 if grouped else None, **kw)
                        for q, ids in enumerate(lists):
                            k, c, d, p, n = orc.facet_count_ex(0, ids, ranges=ranges, doc_vals=vals, distinct_ids=distinct if grouped else None, **kw)
                            m = {int(a): int(b) for a, b in zip(k.view(np.int64), c)}
                            assert [m.get(int(up), 0) for up, lo in ranges] == got[q].tolist(), (array, ranges, kw, grouped, q)
    with pytest.raises(T.TsgpuError):
        g.facet_range_count_batch(0, 3, [(5, 0), (5, 1)], lists)                         # upper bounds must ascend strictly
    with pytest.raises(T.TsgpuError):
        g.facet_count_batch(0, lists, group_column=77)
    g.close()


def test_grouped_and_range_facets_match_oracle_emulator():
    _grouped_and_ranges(H.emu_lib_path(), 3000, 60)


@pytest.mark.gpu
def test_grouped_and_range_facets_match_oracle_gpu():
    _grouped_and_ranges(H.gpu_lib_path(), 300_000, 3000)


def _stats_and_values(lib, n_docs, n_values, seed=77):
    """numeric stats of the hash-index walk and the value-index branch, product vs oracle (and vs the reference's fixture)"""
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "facet_stats_values.json")))
    rng = np.random.default_rng(seed)
    g = T.GpuIndex(0, lib)
    orc = O.OracleIndex(1, 1)
    try:
        # -- the reference's documents --
        ids5 = np.array(fx["result_ids"], np.uint32)
        g.facet_set(0, *_csr(fx["rating_bits"]))
        e = fx["rating_expected"]
        mn, mx, sm, cnt, exact = g.facet_stats_batch(0, B.FACET_FLOAT, [ids5])[0]
        assert (mn, mx, cnt) == (e["min"], e["max"], e["count"]) and abs(sm - e["sum"]) <= 1e-12 * abs(e["sum"]) and exact == 0
        g.facet_set(1, *_csr(fx["timestamps_hashes"]))
        m = (np.array([h for h, _ in fx["timestamps_map"]], np.uint32), np.array([v for _, v in fx["timestamps_map"]], np.int64))
        e = fx["timestamps_expected"]
        assert g.facet_stats_batch(1, B.FACET_INT64, [ids5], int64_map=m)[0] == (e["min"], e["max"], e["sum"], e["count"], 1)
        vp, vi = _csr(fx["tags_value_ids"])
        g.facet_value_set(2, vp, vi, [len(r) for r in fx["tags_value_ids"]])
        v, c, d = g.facet_value_count_batch(2, [ids5], max_facets=20)[0]
        assert [[fx["tags_values"][int(a)], int(b)] for a, b in zip(v, c)] == fx["tags_expected"] and d.tolist() == [0, 0, 2, 1]
        # -- random collections vs the oracle --
        lists = [np.sort(rng.choice(n_docs + 40, size=s, replace=False)).astype(np.uint32) for s in (1, 9, 400, min(n_docs, 4000))] + [np.zeros(0, np.uint32)]
        for vt in (B.FACET_INT32, B.FACET_FLOAT, B.FACET_INT64):
            per = rng.integers(0, 4, size=n_docs)
            ptr = np.zeros(n_docs + 1, np.uint64); ptr[1:] = np.cumsum(per)
            if vt == B.FACET_INT32:
                hashes = rng.integers(-50000, 50000, size=int(ptr[-1])).astype(np.int32).view(np.uint32)
                mp = None
            elif vt == B.FACET_FLOAT:
                hashes = (rng.standard_normal(int(ptr[-1])) * 100).astype(np.float32).view(np.uint32)
                mp = None
            else:
                vals = rng.integers(-2**40, 2**40, size=n_values)
                hs = np.unique(rng.integers(0, 2**32, size=n_values * 2, dtype=np.uint64).astype(np.uint32))[:n_values]
                mp = (hs, vals[:hs.size].astype(np.int64))
                hashes = hs[rng.integers(0, hs.size, size=int(ptr[-1]))]
                hashes[::97] = 12345                                   # a hash absent from the map -> INT64_MAX
            g.facet_set(3, ptr, hashes)
            orc.facet_set(3, ptr, hashes)
            for sample_mod in (1, 3):
                got = g.facet_stats_batch(3, vt, lists, sample_mod=sample_mod, int64_map=mp)
                for q, ids in enumerate(lists):
                    mn, mx, sm, cnt = orc.facet_stats(3, ids, vt, sample_mod=sample_mod, int64_map=mp)
                    assert got[q][0] == mn and got[q][1] == mx and got[q][3] == cnt, (vt, q)
                    if vt == B.FACET_FLOAT:
                        assert abs(got[q][2] - sm) <= 1e-12 * max(1.0, abs(sm)), (q, got[q][2], sm)       # order of the double additions differs
                    elif got[q][4]:
                        assert got[q][2] == sm, (vt, q)
                    else:
                        assert abs(got[q][2] - sm) <= 1e-9 * abs(sm)                                          # beyond 2^53: the reference's own sum is rounded
        # value index: zipf-sized value lists incl. ones beyond 300 ids (the estimated walk) and shorter than 64 (compact: never estimated)
        sizes = np.minimum((n_docs / np.arange(1, n_values + 1) ** 0.9).astype(np.int64) + 1, n_docs)
        rows = [np.sort(rng.choice(n_docs, size=int(sz), replace=False)).astype(np.uint32) for sz in sizes]
        vp = np.zeros(n_values + 1, np.uint64); vp[1:] = np.cumsum([r.size for r in rows])
        vi = np.concatenate(rows)
        tot = np.array([r.size for r in rows], np.uint32)
        g.facet_value_set(4, vp, vi, tot)
        orc.facet_value_set(4, vp, vi, tot)
        alpha = rng.permutation(n_values).astype(np.uint32)
        for kw in (dict(max_facets=10), dict(max_facets=2 * n_values), dict(max_facets=7, wildcard_no_filter=True), dict(max_facets=12, estimate=True, sample_interval=3),
                   dict(max_facets=9, order=alpha), dict(max_facets=5, order=alpha[::-1].copy(), estimate=True, sample_interval=7)):
            got = g.facet_value_count_batch(4, lists, cap=2 * n_values, **kw)
            for q, ids in enumerate(lists):
                v, c, d = orc.facet_value_count(4, ids, **kw)
                assert np.array_equal(got[q][0], v) and np.array_equal(got[q][1], c) and np.array_equal(got[q][2], d), (kw, q)
    finally:
        g.close()


def test_facet_stats_and_value_index_match_oracle_emulator():
    _stats_and_values(H.emu_lib_path(), 2500, 60)


@pytest.mark.gpu
def test_facet_stats_and_value_index_match_oracle_gpu():
    _stats_and_values(H.gpu_lib_path(), 300_000, 900)


def _random_facet_index(rng, n_docs, n_values, array=True):
    per = rng.integers(0, 5, size=n_docs) if array else (rng.random(n_docs) < 0.9).astype(np.int64)
    ptr = np.zeros(n_docs + 1, np.uint64)
    ptr[1:] = np.cumsum(per)
    hashes = (rng.zipf(1.3, size=int(ptr[-1])) % n_values).astype(np.uint32) * np.uint32(2654435761)      # repeats inside a document happen
    return ptr, hashes


def _check(g, orc, lists, cap=4096, **kw):
    got = g.facet_count_batch(0, lists, cap=cap, **kw)
    for q, ids in enumerate(lists):
        h, c, d, p, n = orc.facet_count(0, ids, sample_mod=kw.get("sample_mod", 1), allowed_hashes=kw.get("allowed_hashes"))
        gh, gc, gd, gp, gn = got[q]
        assert gn == n, "query %d: %d distinct values, oracle %d" % (q, gn, n)
        h, c, d, p = h[:cap], c[:cap], d[:cap], p[:cap]           # (the first `cap` values in hash order are returned)
        assert np.array_equal(gh, h) and np.array_equal(gc, c) and np.array_equal(gd, d) and np.array_equal(gp, p), "query %d" % q


def _run(lib, n_docs, n_values):
    rng = np.random.default_rng(31)
    g = T.GpuIndex(0, lib)
    orc = O.OracleIndex(1, 1)
    ptr, hashes = _random_facet_index(rng, n_docs, n_values)
    g.facet_set(0, ptr, hashes)
    orc.facet_set(0, ptr, hashes)
    lists = [np.sort(rng.choice(n_docs + 50, size=s, replace=False)).astype(np.uint32) for s in (1, 7, 300, min(n_docs, 5000))] + [np.zeros(0, np.uint32)]
    _check(g, orc, lists)
    _check(g, orc, lists, sample_mod=3)
    allowed = np.unique(hashes)[::3]
    _check(g, orc, lists, allowed_hashes=allowed)
    # a scalar facet field (at most one hash per document) replaces the mirror
    ptr2, hashes2 = _random_facet_index(rng, n_docs, 11, array=False)
    g.facet_set(0, ptr2, hashes2)
    orc.facet_set(0, ptr2, hashes2)
    _check(g, orc, lists)
    _check(g, orc, lists, cap=4)                                  # more values than the caller takes: only the first `cap` hashes are put in order
    # two values over every document (whole waves on one counter: counted once per wave), in an array field whose documents repeat them
    per = rng.integers(1, 4, size=n_docs)
    ptr3 = np.zeros(n_docs + 1, np.uint64)
    ptr3[1:] = np.cumsum(per)
    hashes3 = np.where(rng.random(int(ptr3[-1])) < 0.7, np.uint32(0xDEADBEEF), np.uint32(12345)).astype(np.uint32)
    g.facet_set(0, ptr3, hashes3)
    orc.facet_set(0, ptr3, hashes3)
    everything = [np.arange(n_docs, dtype=np.uint32)] + lists
    _check(g, orc, everything)
    _check(g, orc, everything, sample_mod=7)
    _check(g, orc, everything, allowed_hashes=np.array([12345], np.uint32))
    # workgroups that walk several rounds of ids before they add their LDS counts to the table; and a field of more values than the LDS table holds
    g.set_option("facet_ids_per_block", 1024)
    _check(g, orc, everything)
    ptr4, _ = _random_facet_index(rng, n_docs, 1 << 20)
    hashes4 = rng.integers(0, 1 << 32, size=int(ptr4[-1]), dtype=np.uint64).astype(np.uint32)      # (every hash its own value: the LDS tables overflow)
    g.facet_set(0, ptr4, hashes4)
    orc.facet_set(0, ptr4, hashes4)
    _check(g, orc, everything, cap=1 << 16)                       # (more than 4 096 values in one query: the host orders that list)
    _check(g, orc, everything, cap=100)
    g.set_option("facet_ids_per_block", 0)
    _check(g, orc, everything, cap=1 << 16, sample_mod=2)
    g.close()


def _sort_boundaries(lib):
    """lists of exactly 0 / 1 / 2 / 4 095 / 4 096 (the device orders them) and 4 097 (the host does) distinct values, hashes with the top bit set, caps below and above"""
    rng = np.random.default_rng(8)
    n_docs = 4200
    g = T.GpuIndex(0, lib)
    orc = O.OracleIndex(1, 1)
    hashes = rng.permutation(np.arange(n_docs, dtype=np.uint32) * np.uint32(1022117) + np.uint32(0x80000000))       # one value per document, all distinct
    ptr = np.arange(n_docs + 1, dtype=np.uint64)
    g.facet_set(0, ptr, hashes)
    orc.facet_set(0, ptr, hashes)
    lists = [np.arange(m, dtype=np.uint32) for m in (0, 1, 2, 4095, 4096, 4097, 4200)]
    for cap in (1, 4096, 8192):
        _check(g, orc, lists, cap=cap)
    g.close()


def test_facet_value_order_at_the_device_sort_boundaries_emulator():
    _sort_boundaries(H.emu_lib_path())


@pytest.mark.gpu
def test_facet_value_order_at_the_device_sort_boundaries_gpu():
    _sort_boundaries(H.gpu_lib_path())


def test_facet_counts_match_oracle_emulator():
    _run(H.emu_lib_path(), 3000, 90)


@pytest.mark.gpu
def test_facet_counts_match_oracle_gpu():
    _run(H.gpu_lib_path(), 400_000, 5000)


@pytest.mark.gpu
def test_facets_over_the_id_lists_of_a_keyword_batch():
    docs = H.zipf_docs(20000, 400, 10, seed=3)
    orc, g = H.build_pair(docs, H.gpu_lib_path())
    rng = np.random.default_rng(4)
    ptr, hashes = _random_facet_index(rng, 20000, 40)
    g.facet_set(0, ptr, hashes)
    orc.facet_set(0, ptr, hashes)
    qs = [T.KwQuery(t, topster_size=50) for t in ([1], [2, 3], [5, 1, 9], [400])]
    hits, ids = g.keyword_search_batch_ids(qs, k_stride=50)
    for i, q in enumerate(qs):
        assert np.array_equal(ids[i], H.oracle_keyword(orc, q, ids_cap=30000).result_ids)
    _check(g, orc, ids)
    g.close()


def test_random_facet_walks_match_oracle_emulator():
    """randomized cases (seeded): tiny to mid-size indexes with empty documents, repeated hashes, ids beyond the index, every option combination of the walk —
    plain, grouped and range forms against the oracle"""
    rng = np.random.default_rng(2024)
    g = T.GpuIndex(0, H.emu_lib_path())
    orc = O.OracleIndex(1, 1)
    for case in range(40):
        n_docs = int(rng.integers(1, 700))
        g.set_num_docs(n_docs)
        per = rng.integers(0, 6, size=n_docs) * (rng.random(n_docs) < 0.8)
        ptr = np.zeros(n_docs + 1, np.uint64)
        ptr[1:] = np.cumsum(per)
        n_val = int(rng.choice([1, 2, 5, 50, 5000]))
        hashes = rng.integers(0, n_val, size=int(ptr[-1])).astype(np.uint32) * np.uint32(0x9E3779B1) + np.uint32(rng.integers(0, 2)) * np.uint32(0xFFFFFFFF)
        g.facet_set(0, ptr, hashes)
        orc.facet_set(0, ptr, hashes)
        lists = [np.sort(rng.choice(n_docs + 30, size=int(rng.integers(0, n_docs + 30)), replace=False)).astype(np.uint32) for _ in range(int(rng.integers(1, 5)))]
        mod = int(rng.choice([1, 1, 2, 5]))
        allowed = np.unique(rng.choice(hashes, size=max(1, hashes.size // 3))) if (hashes.size and rng.random() < 0.4) else None
        short = int(rng.integers(0, min(n_docs, 20)))
        distinct = rng.integers(0, int(rng.choice([1, 3, 1000])), size=n_docs - short).astype(np.uint64) + (np.uint64(1) << np.uint64(int(rng.integers(0, 40))))
        vals = rng.integers(-50, 50, size=n_docs - int(rng.integers(0, min(n_docs, 10)))).astype(np.int64)
        g.column_set(2, distinct.view(np.int64))
        g.column_set(3, vals)
        gmv = bool(rng.integers(0, 2))
        cap = int(rng.choice([1, 7, 8192]))
        _check(g, orc, lists, cap=cap, sample_mod=mod, allowed_hashes=allowed)
        got = g.facet_count_batch(0, lists, cap=8192, sample_mod=mod, allowed_hashes=allowed, group_column=2, group_missing_values=gmv)
        edges = np.unique(rng.integers(-60, 60, size=int(rng.integers(1, 6))))
        ranges = [(int(edges[i]), int(edges[i - 1]) if i and rng.random() < 0.8 else int(edges[i]) - int(rng.integers(1, 30))) for i in range(edges.size)]
        rc = g.facet_range_count_batch(0, 3, ranges, lists, sample_mod=mod)
        rg = g.facet_range_count_batch(0, 3, ranges, lists, sample_mod=mod, group_column=2, group_missing_values=gmv)
        for q, ids in enumerate(lists):
            k, c, d, p, n = orc.facet_count_ex(0, ids, sample_mod=mod, allowed_hashes=allowed, distinct_ids=distinct, group_missing_values=gmv)
            gh, gc, gd, gp, gn = got[q]
            assert gn == n and np.array_equal(gh, k.astype(np.uint32)) and np.array_equal(gc, c) and np.array_equal(gd, d) and np.array_equal(gp, p), (case, q)
            for grouped, res in ((False, rc), (True, rg)):
                k, c, d, p, n = orc.facet_count_ex(0, ids, sample_mod=mod, ranges=ranges, doc_vals=vals, distinct_ids=distinct if grouped else None, group_missing_values=gmv)
                m = {int(a): int(b) for a, b in zip(k.view(np.int64), c)}
                assert [m.get(up, 0) for up, lo in ranges] == res[q].tolist(), (case, q, grouped, ranges)
    g.close()


def test_int32_stats_with_a_negative_value_known_answer_of_the_reference():
    """CollectionFacetingTest.FacetingWithNegativeInt (collection_faceting_test.cpp:3892-3929): points 20, 10, -5 (an int32 field's facet hash is the value's bits) ->
    min -5, max 20, sum 25, avg 8.333333333333334 — oracle and library"""
    hashes = np.array([20, 10, -5], np.int32).view(np.uint32)
    ptr = np.arange(4, dtype=np.uint64)
    ids = np.arange(3, dtype=np.uint32)
    orc = O.OracleIndex(1, 1)
    orc.facet_set(0, ptr, hashes)
    mn, mx, sm, cnt = orc.facet_stats(0, ids, B.FACET_INT32)
    assert (mn, mx, sm, cnt) == (-5.0, 20.0, 25.0, 3) and sm / cnt == pytest.approx(8.333333333333334, rel=1e-7)
    g = T.GpuIndex(0, H.emu_lib_path())
    g.set_num_docs(3)
    g.facet_set(0, ptr, hashes)
    fmin, fmax, fsum, fcnt, exact = g.facet_stats_batch(0, B.FACET_INT32, [ids])[0]
    assert (fmin, fmax, fsum, fcnt) == (-5.0, 20.0, 25.0, 3) and exact
    g.close()


def test_float_stats_known_answer_of_the_reference():
    """CollectionFacetingTest.FacetStatOnFloatFields (collection_faceting_test.cpp:645-712) on test/float_documents.jsonl — oracle (double accumulation in document order:
    bit for bit what the reference prints) and library (min / max / count exact, the sum to double rounding)"""
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "facet_group_range.json")))["float_stats"]
    e = fx["expected"]
    ptr, hashes = _csr(fx["average_bits"])
    ids = np.arange(len(fx["average_bits"]), dtype=np.uint32)
    orc = O.OracleIndex(1, 1)
    orc.facet_set(0, ptr, hashes)
    mn, mx, sm, cnt = orc.facet_stats(0, ids, B.FACET_FLOAT)
    assert cnt == e["count"] and np.float32(mn) == np.float32(e["min"]) and mx == e["max"]
    assert sm == pytest.approx(e["sum"], rel=1e-7) and sm / cnt == pytest.approx(e["avg"], rel=1e-7)              # (ASSERT_FLOAT_EQ in the reference)
    g = T.GpuIndex(0, H.emu_lib_path())
    g.set_num_docs(ids.size)
    g.facet_set(0, ptr, hashes)
    fmin, fmax, fsum, fcnt, exact = g.facet_stats_batch(0, B.FACET_FLOAT, [ids])[0]
    assert (fmin, fmax, fcnt) == (mn, mx, cnt) and fsum == pytest.approx(sm, rel=1e-12)
    g.close()


def test_facet_edge_cases_emulator():
    """empty field, empty batch, a field without any document, 1 024 ranges (the maximum) and one more, ranges that no value reaches"""
    g = T.GpuIndex(0, H.emu_lib_path())
    orc = O.OracleIndex(1, 1)
    g.set_num_docs(50)
    ptr = np.zeros(51, np.uint64)                                     # fifty documents, none with a value
    g.facet_set(0, ptr, np.zeros(0, np.uint32))
    orc.facet_set(0, ptr, np.zeros(0, np.uint32))
    lists = [np.arange(50, dtype=np.uint32), np.zeros(0, np.uint32)]
    _check(g, orc, lists)
    assert g.facet_count_batch(0, [], cap=8) == []
    g.column_set(1, np.arange(50, dtype=np.int64))
    assert g.facet_range_count_batch(0, 1, [(10, 0)], lists).tolist() == [[0], [0]]
    got = g.facet_count_batch(0, lists, cap=8, group_column=1)
    assert [r[4] for r in got] == [0, 0]
    # every document one value; 1 024 one-wide ranges over the values 0 .. 49 (most of them empty), then 1 025
    ptr = np.arange(51, dtype=np.uint64)
    hashes = np.arange(50, dtype=np.uint32) + np.uint32(5)
    g.facet_set(0, ptr, hashes)
    orc.facet_set(0, ptr, hashes)
    ranges = [(i - 499, i - 500) for i in range(1024)]
    rc = g.facet_range_count_batch(0, 1, ranges, lists)
    k, c, d, p, n = orc.facet_count_ex(0, lists[0], ranges=ranges, doc_vals=np.arange(50, dtype=np.int64))
    m = {int(a): int(b) for a, b in zip(k.view(np.int64), c)}
    assert rc[0].tolist() == [m.get(up, 0) for up, lo in ranges] and int(rc[0].sum()) == 50 and int(rc[1].sum()) == 0
    with pytest.raises(T.TsgpuError):
        g.facet_range_count_batch(0, 1, [(i, i - 1) for i in range(1025)], lists)
    assert g.facet_range_count_batch(0, 1, [(-10, -20), (1000, 900)], lists).tolist() == [[0, 0], [0, 0]]
    g.close()


def test_repeated_array_values_count_once_known_answer_of_the_reference():
    """CollectionFacetingTest.FacetByArrayField (collection_faceting_test.cpp:1176-1224): data = ["Foo", "Foo"] and ["Foo", "Foo", "Bazinga"] -> Foo 2, Bazinga 1;
    with the facet query that only matches Bazinga -> Bazinga 1 (string facet ids in order of first appearance: Foo 1, Bazinga 2)"""
    ptr, hashes = _csr([[1, 1], [1, 1, 2]])
    ids = np.arange(2, dtype=np.uint32)
    orc = O.OracleIndex(1, 1)
    orc.facet_set(0, ptr, hashes)
    g = T.GpuIndex(0, H.emu_lib_path())
    g.set_num_docs(2)
    g.facet_set(0, ptr, hashes)
    for allowed, want in ((None, {1: 2, 2: 1}), (np.array([2], np.uint32), {2: 1})):
        h, c, d, p, n = orc.facet_count(0, ids, allowed_hashes=allowed)
        assert {int(a): int(b) for a, b in zip(h, c)} == want
        gh, gc, gd, gp, gn = g.facet_count_batch(0, [ids], cap=8, allowed_hashes=allowed)[0]
        assert {int(a): int(b) for a, b in zip(gh, gc)} == want and np.array_equal(gd, d) and np.array_equal(gp, p)
    g.close()


def test_bool_facet_with_the_hash_zero():
    """CollectionFacetingTest.FacetCountsBool (collection_faceting_test.cpp:422-476): in_stock = true, false, true; filter in_stock:true leaves documents 0 and 2 -> true 2.
    A bool field's facet hash is (uint32) the value: `false` is the hash 0, which the tables must tell from an empty slot (all three documents: true 2, false 1)"""
    ptr, hashes = _csr([[1], [0], [1]])
    orc = O.OracleIndex(1, 1)
    orc.facet_set(0, ptr, hashes)
    g = T.GpuIndex(0, H.emu_lib_path())
    g.set_num_docs(3)
    g.facet_set(0, ptr, hashes)
    g.column_set(1, np.array([7, 7, 9], np.int64))
    for ids, want in ((np.array([0, 2], np.uint32), {1: 2}), (np.arange(3, dtype=np.uint32), {0: 1, 1: 2})):
        h, c, d, p, n = orc.facet_count(0, ids)
        assert {int(a): int(b) for a, b in zip(h, c)} == want
        gh, gc, gd, gp, gn = g.facet_count_batch(0, [ids], cap=8)[0]
        assert {int(a): int(b) for a, b in zip(gh, gc)} == want and np.array_equal(gd, d)
    gh, gc, gd, gp, gn = g.facet_count_batch(0, [np.arange(3, dtype=np.uint32)], cap=8, group_column=1)[0]
    assert {int(a): int(b) for a, b in zip(gh, gc)} == {0: 1, 1: 2}                     # (groups 7 and 9 hold `true`, group 7 holds `false`)
    g.close()


def test_grouped_pair_of_all_ones_emulator():
    """the (value, group) pair 0xFFFFFFFF / 0xFFFFFFFF equals the pair table's empty marker: it is counted through a flag of its own, once"""
    ptr, hashes = _csr([[0xFFFFFFFF], [0xFFFFFFFF], [0xFFFFFFFF, 5], [5]])
    distinct = np.array([0xFFFFFFFF, 0x1FFFFFFFF, 0xFFFFFFFF, 3], np.uint64)            # (documents 0, 1, 2: one group for hash_groups' uint32 set)
    orc = O.OracleIndex(1, 1)
    orc.facet_set(0, ptr, hashes)
    g = T.GpuIndex(0, H.emu_lib_path())
    g.set_num_docs(4)
    g.facet_set(0, ptr, hashes)
    g.column_set(1, distinct.view(np.int64))
    ids = np.arange(4, dtype=np.uint32)
    k, c, d, p, n = orc.facet_count_ex(0, ids, distinct_ids=distinct)
    assert {int(a): int(b) for a, b in zip(k, c)} == {5: 2, 0xFFFFFFFF: 1}
    gh, gc, gd, gp, gn = g.facet_count_batch(0, [ids, ids], cap=8, group_column=1)[1]
    assert np.array_equal(gh, k.astype(np.uint32)) and np.array_equal(gc, c) and np.array_equal(gd, d) and np.array_equal(gp, p)
    g.close()
