"""The reference's own vector-distance known answers (test/collection_vector_search_test.cpp:75-137 BasicVectorQuerying,
:806-901 VecSearchWithFiltering, :5093-5196 TestDistanceThresholdWithIP) through the C-ABI: tsgpu_vector_search_batch,
tsgpu_vec_knn_batch (allow ids = the flat path under a filter), tsgpu_vec_distances (by-id = the `_vector_query` sort key).

Fixture: tests/golden/vector_pins.json, written by `oracle/_build/golden_tests tests/golden --dump-vector-pins <path>` with the
reference's generators (std::mt19937 seed 47); the CPU tier re-generates it and compares. Tolerance: 1e-5 relative
(north_star) everywhere, and additionally the stronger facts that hold: the cosine pins and the printed sort-key form
-int64_t_to_float(-float_to_int64_t(d)) (src/collection.cpp:3183 over src/index.cpp:5850, :5901-5903) are BIT-exact."""
import json
import os
import subprocess

import numpy as np
import pytest

import typesense_amd as T
from typesense_amd import _lib as B
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PINS = os.path.join(ROOT, "tests", "golden", "vector_pins.json")
RTOL = 1e-5
Q1 = np.array([0.96826, 0.94, 0.39557, 0.306488], np.float32)                      # std::stof of the test's literals
QIP = np.array([0.11731103425347378, -0.6694758317235057, -0.6211945774857595, -0.27966758971688255, -0.4683744007950299], np.float32)
F32_MAX = np.float32(3.4028234663852886e38)


def f2i(d):
    """Index::float_to_int64_t (src/index.cpp:266-274)"""
    i = np.asarray(d, np.float32).view(np.int32).astype(np.int64)
    return np.where(i < 0, i ^ 0x7FFFFFFF, i)


def i2f(n):
    """Index::int64_t_to_float (src/index.cpp:276-285)"""
    i = np.asarray(n, np.int64).astype(np.int32)
    i = np.where(i < 0, i ^ np.int32(0x7FFFFFFF), i).astype(np.int32)
    return i.view(np.float32)


def printed(d):
    """what Collection::search prints for an ASC `_vector_query` sort key"""
    return -i2f(-f2i(d))


def bits(x):
    return np.asarray(x, np.float32).view(np.uint32)


def pins():
    with open(PINS) as f:
        return json.load(f)


def run_pins(lib_path):
    P = pins()
    g = T.GpuIndex(0, lib_path)
    try:
        # ---- BasicVectorQuerying (:75-137): cosine 4-d, order 1,0,2, three distances ----
        docs = np.array(P["basic_docs"], np.float32)
        g.vec_create(1, 4, B.METRIC_COSINE)
        g.vec_upsert(1, np.arange(3, dtype=np.uint64), docs)
        hits = g.vector_search_batch(1, Q1[None, :], fetch_size=10)
        assert int(hits.n_hits[0]) == 3 and hits.keys[0, :3].tolist() == [1, 0, 2]
        want = np.array([3.409385681152344e-05, 0.04329806566238403, 0.15141665935516357], np.float32)
        got = hits.vector_distance[0, :3]
        assert np.allclose(got, want, rtol=RTOL, atol=0)
        assert np.array_equal(bits(got), bits(want)), (got, want)                  # stronger: bit-exact
        assert np.array_equal(hits.scores[0, :3, 0], -f2i(want))                    # sort key = -float_to_int64_t(d) (ASC)
        dist, lab, cnt = g.vec_knn_batch(1, Q1[None, :], 10, allow_ids=np.array([0, 1], np.uint32))   # points:[0,1]
        assert cnt[0] == 2 and lab[0, :2].tolist() == [1, 0] and np.array_equal(bits(dist[0, :2]), bits(want[:2]))

        # ---- VecSearchWithFiltering (:806-901): seed-47 unit-cube docs, flat path under points:<10 ----
        docs = np.array(P["seed47_unit_docs"], np.float32)
        g.vec_create(2, 4, B.METRIC_COSINE)
        g.vec_upsert(2, np.arange(20, dtype=np.uint64), docs)
        hits = g.vector_search_batch(2, Q1[None, :], fetch_size=20)
        assert int(hits.n_hits[0]) == 20
        filt = np.arange(10, dtype=np.uint32)
        dist, lab, cnt = g.vec_knn_batch(2, Q1[None, :], 3, allow_ids=filt)
        d = np.abs(dist[0])
        assert cnt[0] == 3 and lab[0, 0] == 1 and lab[0, 1] == 5
        assert abs(d[0] - 3.409385e-05) <= RTOL * 3.409385e-05 and abs(d[1] - 0.016780376) <= RTOL * 0.016780376
        dist, lab, cnt = g.vec_knn_batch(2, docs[3][None, :], 4, allow_ids=filt)   # vec:([], id: 3): k+1, the document itself dropped
        keep = [(float(abs(dist[0, i])), int(lab[0, i])) for i in range(int(cnt[0])) if int(lab[0, i]) != 3]
        assert len(keep) == 3 and keep[0][1] == 9 and keep[1][1] == 5
        assert abs(keep[0][0] - 0.050603985) <= RTOL * 0.050603985 and abs(keep[1][0] - 0.100155532) <= RTOL * 0.100155532

        # the same requests through the vector branch itself (tsgpu_vector_search_batch): `flat_search_cutoff: 0` = the k-cut branch,
        # found 10 / 10 hits (:851-863); `flat_search_cutoff: 1000`, per_page 3 = the FLAT branch: found == 10 although k = 3, the
        # Topster (capacity min(250, 10)) holds all ten, hits 1 and 5 with the pinned distances (:865-881)
        hits = g.vector_search_batch(2, Q1[None, :], fetch_size=20, filter_ids=filt, flat_search_cutoff=0)
        assert int(hits.n_hits[0]) == 10 and int(hits.num_matched[0]) == 10
        hits, ids = g.vector_search_batch(2, Q1[None, :], fetch_size=3, filter_ids=filt, flat_search_cutoff=1000, want_ids=True)
        assert int(hits.num_matched[0]) == 10 and int(hits.n_hits[0]) == 10 and ids[0].tolist() == list(range(10))     # :874 found == 10
        assert hits.keys[0, :2].tolist() == [1, 5]
        d = hits.vector_distance[0, :2]
        assert abs(d[0] - 3.409385e-05) <= RTOL * 3.409385e-05 and abs(d[1] - 0.016780376) <= RTOL * 0.016780376
        assert np.array_equal(bits(d[0]), bits(np.float32(3.409385681152344e-05)))   # (the 17-digit pin of the same pair, :120: bit-exact)
        assert np.array_equal(hits.scores[0, :10, 0], -f2i(hits.vector_distance[0, :10])) and np.array_equal(hits.scores[0, :10, 1], hits.keys[0, :10].astype(np.int64))
        cut = g.vector_search_batch(2, Q1[None, :], fetch_size=3, filter_ids=filt, flat_search_cutoff=0)     # without the flat branch: k = 3 cuts, found 3
        assert int(cut.num_matched[0]) == 3 and cut.keys[0, :2].tolist() == [1, 5]
        # `vec:([], id: 3, flat_search_cutoff: 1000)` (:883-901): document 3 is the query and is left out; hits 9 and 5
        for cutoff, found in ((1000, 9), (0, 3)):
            hits = g.vector_search_batch(2, docs[3][None, :], fetch_size=3, filter_ids=filt, flat_search_cutoff=cutoff, query_doc=3)
            assert int(hits.num_matched[0]) == found and hits.keys[0, :2].tolist() == [9, 5], (cutoff, hits.keys[0, :4], hits.num_matched[0])
            d = hits.vector_distance[0, :2]
            assert abs(d[0] - 0.050603985) <= RTOL * 0.050603985 and abs(d[1] - 0.100155532) <= RTOL * 0.100155532

        # ---- TestDistanceThresholdWithIP (:5093-5196): IP 5-d, distance as a sort key over all five documents ----
        docs = np.array(P["seed47_ip_docs"], np.float32)
        rank = np.array(P["seed47_ip_rank_scores"])
        g.vec_create(3, 5, B.METRIC_IP)
        g.vec_upsert(3, np.arange(5, dtype=np.uint64), docs)
        raw = g.vec_distances(3, QIP, np.arange(5, dtype=np.uint64))
        thr = np.where(raw > np.float32(1.0), F32_MAX, raw).astype(np.float32)      # distance_threshold:1 -> FLT_MAX (src/index.cpp:5844-5848)
        pr = printed(thr)
        order = sorted(range(5), key=lambda i: (pr[i], -rank[i]))                   # distance asc, rank_score desc
        assert [int(rank[i]) for i in order] == [93, 51, 94, 80, 18]
        want = np.array([0.2189185470342636, 0.7371898889541626] + [3.4028232635611926e+38] * 3, np.float32)
        assert np.array_equal(bits(pr[order]), bits(want)), (pr[order], want)       # ASSERT_EQ in the reference: bit-exact
        assert np.allclose(raw[order[:2]], want[:2], rtol=RTOL, atol=0)             # raw distances: 1 ulp from the printed pins
        qneg = np.full(5, -100.0, np.float32)
        raw = g.vec_distances(3, qneg, np.arange(5, dtype=np.uint64))
        pr = printed(raw)
        order = np.argsort(pr, kind="stable")
        want = np.array([-45.23314666748047, -38.66290283203125, -36.0988655090332, 9.637892723083496, 288.0364685058594], np.float32)
        assert order.tolist() == [1, 2, 4, 3, 0]
        assert np.array_equal(bits(pr[order]), bits(want)), (pr[order], want)
        assert np.allclose(raw[order], want, rtol=RTOL, atol=0)
        # the same five through the k-NN entry point (searchKnnCloserFirst order)
        dist, lab, cnt = g.vec_knn_batch(3, qneg[None, :], 5)
        assert lab[0].tolist() == [1, 2, 4, 3, 0] and np.array_equal(bits(dist[0]), bits(raw[order]))
    finally:
        g.close()


def test_fixture_regenerates_from_the_reference_generators(tmp_path):
    from oracle import oracle_py
    oracle_py.build()
    out = str(tmp_path / "pins.json")
    p = subprocess.run([os.path.join(ROOT, "oracle", "_build", "golden_tests"), os.path.join(ROOT, "tests", "golden"), "--dump-vector-pins", out],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:]
    assert "rank_score labels reproduce" in p.stdout
    with open(out) as f:
        assert json.load(f) == pins()


def test_oracle_distances_equal_the_reference_pins():
    from oracle import oracle_py as O
    P = pins()
    docs = np.array(P["seed47_ip_docs"], np.float32)
    L = O.lib()
    d = np.array([L.orc_ip_distance(QIP.ctypes.data, np.ascontiguousarray(docs[i]).ctypes.data, 5) for i in range(5)], np.float32)
    pr = printed(np.where(d > 1, F32_MAX, d).astype(np.float32))
    assert sorted(bits(pr).tolist()) == sorted(bits(np.array([0.2189185470342636, 0.7371898889541626] + [3.4028232635611926e+38] * 3, np.float32)).tolist())


def test_reference_vector_pins_on_the_emulator():
    run_pins(H.emu_lib_path())


@pytest.mark.gpu
def test_reference_vector_pins_on_the_gpu():
    run_pins(H.gpu_lib_path())
