"""`-m gpu`: vector / hybrid / shard-merge parity on a real MI355X through libtsgpu.so.
The small cases are the same bodies as tests/test_emu_vector.py (run here against the real library);
the large ones exercise many slabs, many query tiles and the batch sizes of BASELINE config 3."""
import numpy as np
import pytest

import typesense_amd as T
from typesense_amd import _lib as B
from oracle import oracle_py as O
from tests import helpers as H
from tests import test_emu_vector as E

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _real_library(monkeypatch):
    monkeypatch.setattr(H, "emu_lib_path", lambda *a, **k: H.gpu_lib_path())


test_knn_matches_oracle_flat_scan = E.test_knn_matches_oracle_flat_scan
test_cosine_normalisation_is_bit_exact_and_distances_match = E.test_cosine_normalisation_is_bit_exact_and_distances_match
test_ties_prefer_smaller_label_and_many_slabs = E.test_ties_prefer_smaller_label_and_many_slabs
test_upsert_delete_labels_filters_and_by_id_distances = E.test_upsert_delete_labels_filters_and_by_id_distances
test_pure_vector_search_topster_order_matches_oracle = E.test_pure_vector_search_topster_order_matches_oracle
test_hybrid_rank_fusion_matches_oracle_bit_exactly = E.test_hybrid_rank_fusion_matches_oracle_bit_exactly
test_hybrid_with_filter_and_excluded_ids_matches_oracle = E.test_hybrid_with_filter_and_excluded_ids_matches_oracle
test_hybrid_rerank_hybrid_matches_is_compute_aux_scores_bit_exactly = E.test_hybrid_rerank_hybrid_matches_is_compute_aux_scores_bit_exactly
test_hnsw_on_a_knn_heuristic_graph_matches_the_oracle_traversal_of_the_same_graph = E.test_hnsw_on_a_knn_heuristic_graph_matches_the_oracle_traversal_of_the_same_graph
test_shard_merge_equals_unsharded = E.test_shard_merge_equals_unsharded
test_knn_two_pass_threshold_path_is_exact = E.test_knn_two_pass_threshold_path_is_exact
test_knn_two_pass_all_equal_distances_converges = E.test_knn_two_pass_all_equal_distances_converges
test_prefilter_distances_are_bit_identical_to_the_reference_order = E.test_prefilter_distances_are_bit_identical_to_the_reference_order
test_prefilter_brackets_prune_but_never_drop_a_neighbour = E.test_prefilter_brackets_prune_but_never_drop_a_neighbour
test_hnsw_graph_search_replays_the_reference_traversal = E.test_hnsw_graph_search_replays_the_reference_traversal
test_edge_cases_empty_tiny_and_fully_deleted_indexes = E.test_edge_cases_empty_tiny_and_fully_deleted_indexes
test_every_summation_order_of_hnswlibs_distance_is_bit_exact = E.test_every_summation_order_of_hnswlibs_distance_is_bit_exact
test_vector_branch_flat_and_k_cut_match_the_oracle = E.test_vector_branch_flat_and_k_cut_match_the_oracle
test_hnsw_graph_built_inside_the_library_equals_the_oracles_link_for_link = E.test_hnsw_graph_built_inside_the_library_equals_the_oracles_link_for_link
test_hnsw_updates_and_slot_reuse_follow_addpoint_with_replace_deleted = E.test_hnsw_updates_and_slot_reuse_follow_addpoint_with_replace_deleted
test_hnsw_build_after_every_row_was_deleted_relinks_through_the_deleted_entry_point = E.test_hnsw_build_after_every_row_was_deleted_relinks_through_the_deleted_entry_point


def test_flat_branch_at_size_many_work_items_768_dims():
    """row a19 at size: 120 000 documents x 768 dims, 50 000 filter ids (four 16K-id work items per query, partial Topsters merged),
    a third of the rows one duplicated embedding; hits, sort keys, distance bits, found and all_result_ids = the oracle's"""
    E._flat_case(H.gpu_lib_path(), 120_000, 768, B.METRIC_COSINE, 7, 50_000, n_q=2)


def test_hnsw_20k_x_96_batch_of_512_matches_the_cpu_traversal():
    """a graph of 20 000 nodes (M=16, ef_construction=100) mirrored into HBM; 512 concurrent queries (one wavefront each):
    identical to the CPU traversal of the same graph, recall@10 vs the exact scan reported"""
    import time
    rng = np.random.default_rng(12)
    n, dim = 20_000, 96
    X = rng.standard_normal((n, dim)).astype(np.float32)
    g = T.GpuIndex(0)
    g.vec_create(1, dim, B.METRIC_IP, n)
    g.vec_upsert(1, np.arange(n, dtype=np.uint64), X)
    orc = O.OracleIndex(1, 1)
    orc.vec_init(dim, O.METRIC_IP)
    orc.vec_add(np.arange(n, dtype=np.uint32), X)
    orc.hnsw_build(M=16, ef_construction=100, seed=100)
    g.vec_hnsw_load(1, orc.hnsw_export())
    Q = rng.standard_normal((512, dim)).astype(np.float32)
    dist, lab, cnt = g.vec_hnsw_search_batch(1, Q, 10, 100)
    t0 = time.perf_counter()
    dist, lab, cnt = g.vec_hnsw_search_batch(1, Q, 10, 100)
    ms = (time.perf_counter() - t0) * 1e3
    de, le, _ = g.vec_knn_batch(1, Q, 10)
    hit = 0
    for i in range(512):
        if i < 48:
            d, l, _ = orc.hnsw_search(Q[i], 10, 100)
            assert cnt[i] == d.size and np.array_equal(lab[i, :d.size], l) and np.array_equal(dist[i, :d.size].view(np.uint32), d.view(np.uint32))
        hit += len(set(lab[i, :cnt[i]].tolist()) & set(le[i].tolist()))
    print("hnsw 20Kx96 B=512 k=10 ef=100: %.2f ms/batch (host-to-host), recall@10 %.3f" % (ms, hit / 5120))
    assert hit / 5120 > 0.6            # inner product on unnormalised gaussian rows is a hard case for graph search; the CPU traversal gives the same
    g.close()


def test_hnsw_20k_x_96_built_inside_the_library_link_for_link_and_searched_on_the_gpu():
    """tsgpu_vec_hnsw_enable at the reference's construction parameters (M 16, ef_construction 200, seed 100; include/index.h:365-367), one thread:
    20 000 x 96 rows inserted in label order = the oracle's restatement of hnswlib's addPoint link for link (every level, the entry point, every
    list in order); then 64 queries on the GPU = the oracle's traversal of that graph (labels, order, distance bits). PARITY UNPINNED."""
    n, dim = 20000, 96
    rng = np.random.default_rng(2024)
    X = rng.standard_normal((n, dim)).astype(np.float32)
    g = T.GpuIndex(0, H.gpu_lib_path())
    g.vec_create(1, dim, B.METRIC_IP)
    g.vec_hnsw_enable(1, M=16, ef_construction=200, seed=100, threads=1)
    for a in range(0, n, 4096):
        g.vec_upsert(1, np.arange(a, min(n, a + 4096), dtype=np.uint64), X[a:a + 4096])
    orc = O.OracleIndex(1, 1)
    orc.vec_init(dim, B.METRIC_IP)
    orc.vec_add(np.arange(n, dtype=np.uint32), X)
    orc.hnsw_build(M=16, ef_construction=200, seed=100)
    mine, ref = g.vec_hnsw_export(1), orc.hnsw_export()
    assert mine["n"] == n and E._graphs_equal(mine, ref), "the library's graph differs from the oracle's"
    Q = rng.standard_normal((64, dim)).astype(np.float32)
    dist, lab, cnt = g.vec_hnsw_search_batch(1, Q, 100, 100)
    for i in range(Q.shape[0]):
        d, l, _ = orc.hnsw_search(Q[i], 100, 100, functor_present=True)
        assert cnt[i] == d.size and np.array_equal(lab[i, :d.size], l) and np.array_equal(dist[i, :d.size].view(np.uint32), d.view(np.uint32)), i
    g.close()


def test_hnsw_300k_x_128_graph_built_on_the_gpu_matches_the_oracle_on_the_same_graph():
    """the bench's HNSW leg at test size: a graph derived on the GPU from exact k-NN lists + the selection heuristic (hnsw_synth), adopted by
    the oracle; 2 048 concurrent queries (LDS tier 0 at ef=100, tier 1 at ef=300): labels, order and distance bits of the first 96
    equal the CPU traversal of that graph; recall vs the exact scan is reported (parity unpinned: hnswlib is absent)"""
    import torch
    from typesense_amd import synth, hnsw_synth
    n, dim, k = 300_000, 128, 50
    X = synth.latent_vectors(n, dim, seed=21, latent=16, device="cuda")
    g = T.GpuIndex(0)
    g.vec_create(1, dim, B.METRIC_IP, n)
    lab = torch.arange(n, dtype=torch.int64, device="cuda")
    g.vec_upsert_device(1, lab.data_ptr(), X.data_ptr(), n)
    graph = hnsw_synth.build_graph(torch, g, 1, X, M=16, K0=48, seed=3, batch=1024)
    assert graph["link0"][:, 0].max() <= 32 and graph["link0"][:, 0].min() >= 1
    g.vec_hnsw_load(1, graph)
    orc = O.OracleIndex(1, 1)
    orc.vec_init(dim, O.METRIC_IP)
    orc.vec_add(np.arange(n, dtype=np.uint32), X.cpu().numpy())
    orc.hnsw_import(graph)
    Q = synth.latent_vectors(2048, dim, seed=22, latent=16, device="cuda").cpu().numpy()
    de, le, _ = g.vec_knn_batch(1, Q[:128], k)
    for ef in (100, 300):
        dist, labs, cnt = g.vec_hnsw_search_batch(1, Q, k, ef)
        od, ol, oc = orc.hnsw_search_batch(Q[:96], k, ef, threads=16)
        assert np.array_equal(cnt[:96], oc) and np.array_equal(labs[:96], ol) and np.array_equal(dist[:96].view(np.uint32), od.view(np.uint32)), ef
        rec = np.mean([len(set(labs[i].tolist()) & set(le[i].tolist())) / k for i in range(128)])
        print("hnsw 300Kx128 knn-heuristic graph B=2048 k=%d ef=%d: recall %.3f, expansions/query %.0f" % (k, ef, rec, g.counter("hnsw_last_expansions") / 2048))
        assert rec > (0.5 if ef == 100 else 0.8), rec
    g.close()


@pytest.fixture(scope="module")
def big():
    rng = np.random.default_rng(77)
    n, dim = 200_000, 768
    X = rng.standard_normal((n, dim)).astype(np.float32)
    g = T.GpuIndex(0)
    g.vec_create(1, dim, B.METRIC_IP, n)
    g.vec_upsert(1, np.arange(n, dtype=np.uint64), X)
    orc = O.OracleIndex(1, 1)
    orc.vec_init(dim, O.METRIC_IP)
    orc.vec_add(np.arange(n, dtype=np.uint32), X)
    yield g, orc, X, rng
    g.close()


@pytest.mark.parametrize("n_q", [1, 16, 64, 300])
def test_200k_x_768_batches_top100(big, n_q):
    """config-3 shape at 200K rows: distances <= 1e-5 rel, top-100 sets equal modulo a 1e-5 tie band"""
    g, orc, X, rng = big
    Q = rng.standard_normal((n_q, 768)).astype(np.float32)
    dist, lab, cnt = g.vec_knn_batch(1, Q, 100)
    assert (cnt == 100).all()
    for i in range(min(n_q, 24)):
        d, l = orc.flat_knn(Q[i], 100)
        assert np.allclose(dist[i], d, rtol=1e-5, atol=1e-5)
        a, b = set(lab[i].tolist()), set(l.tolist())
        if a != b:      # only members inside the 1e-5 band around the 100th distance may differ
            band = 1e-5 * max(1.0, abs(float(d[-1])))
            for x in a ^ b:
                dx = 1.0 - float(np.dot(Q[i].astype(np.float64), X[x].astype(np.float64)))
                assert abs(dx - float(d[-1])) <= 2 * band
        assert (np.diff(dist[i]) >= 0).all()


@pytest.mark.parametrize("n_q,sample_tiles", [(3, 512), (70, 64), (130, 1562)])
def test_200k_x_768_prefilter_is_bit_identical_to_the_reference_order(big, n_q, sample_tiles):
    """bf16 bracket scan on the real matrix cores + exact fp32 re-score in hnswlib's summation order: same labels in the
    same order and the same distance BITS as the oracle; the fp32 MFMA scan (vec_prefilter=0) returns the same label sets"""
    g, orc, X, rng = big
    g.set_option("vec_sample_tiles", sample_tiles)
    g.set_option("vec_count_rescored", 1)
    Q = rng.standard_normal((n_q, 768)).astype(np.float32)
    f0 = g.counter("vec_prefilter_fallbacks")
    dist, lab, cnt = g.vec_knn_batch(1, Q, 100)
    assert (cnt == 100).all() and g.counter("vec_prefilter_fallbacks") == f0
    assert 100 * n_q <= g.counter("vec_rescored_rows") <= 40 * 100 * n_q        # the brackets really prune (200K rows/query in)
    for i in range(min(n_q, 12)):
        d, l = orc.flat_knn(Q[i], 100)
        assert np.array_equal(lab[i].astype(np.uint32), l)
        assert np.array_equal(dist[i].view(np.uint32), d.view(np.uint32))
    g.set_option("vec_prefilter", 0)
    d0, l0, _ = g.vec_knn_batch(1, Q[:8], 100)
    g.set_option("vec_prefilter", 1)
    g.set_option("vec_sample_tiles", 512)
    g.set_option("vec_count_rescored", 0)
    for i in range(min(n_q, 8)):
        assert np.allclose(d0[i], dist[i], rtol=1e-5, atol=1e-5)
        if set(l0[i].tolist()) != set(lab[i].tolist()):      # fp32-MFMA summation order may flip members inside the 1e-5 tie band
            band = 2e-5 * max(1.0, abs(float(dist[i][-1])))
            for x in set(l0[i].tolist()) ^ set(lab[i].tolist()):
                dx = 1.0 - float(np.dot(Q[i].astype(np.float64), X[int(x)].astype(np.float64)))
                assert abs(dx - float(dist[i][-1])) <= band


def test_knn_is_deterministic_and_slab_invariant(big):
    g, _, _, rng = big
    Q = rng.standard_normal((32, 768)).astype(np.float32)
    d0, l0, _ = g.vec_knn_batch(1, Q, 100)
    d1, l1, _ = g.vec_knn_batch(1, Q, 100)
    assert np.array_equal(d0, d1) and np.array_equal(l0, l1)
    g.set_option("vec_rows_per_slab", 4096)
    d2, l2, _ = g.vec_knn_batch(1, Q, 100)
    g.set_option("vec_rows_per_slab", 128 * 100)
    d3, l3, _ = g.vec_knn_batch(1, Q, 100)
    assert np.array_equal(d0, d2) and np.array_equal(l0, l2) and np.array_equal(d0, d3) and np.array_equal(l0, l3)


@pytest.mark.parametrize("n_clusters,lo,hi", [(32, 8192, 24576), (9, 24576, 65536)])
def test_tight_clusters_keep_a_whole_cluster_inside_the_bracket_and_stay_on_the_bracket_path(n_clusters, lo, hi):
    """unit-length rows in tight clusters of ~12 000: the bf16 bracket of a query's k-th neighbour holds its WHOLE cluster (more than the
    8 192 survivors rounds 1-2 allowed -> every group fell back to the fp32 scan). With the survivor arena at the refine kernel's own limit
    (24 576) the bracket path handles it: no fallback, ~a cluster re-scored per query, labels / order / distance bits = the oracle's exact scan.
    Clusters of ~43 000 (second case) also outgrow that list: the keys spill to the global list, the (slab, query) segments grow, the
    survivor arena (65 536 rows per query) takes the cluster"""
    rng = np.random.default_rng(5)
    n, dim, n_q = 384_000, 128, 48
    cen = rng.standard_normal((n_clusters, dim)).astype(np.float32)
    idx = rng.integers(0, n_clusters, size=n)
    X = cen[idx] + 0.15 * rng.standard_normal((n, dim)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    Q = cen[rng.integers(0, n_clusters, size=n_q)] + 0.15 * rng.standard_normal((n_q, dim)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    g = T.GpuIndex(0)
    g.vec_create(1, dim, B.METRIC_IP, n)
    g.vec_upsert(1, np.arange(n, dtype=np.uint64), X)
    g.set_option("vec_count_rescored", 1)
    f0 = g.counter("vec_prefilter_fallbacks")
    dist, lab, cnt = g.vec_knn_batch(1, Q, 100)
    assert (cnt == 100).all()
    assert g.counter("vec_prefilter_fallbacks") == f0, "the clustered batch fell back to the fp32 scan"
    per_query = g.counter("vec_rescored_rows") / n_q
    assert lo < per_query <= hi, "expected about one cluster (~%d rows) inside every bracket, got %.0f rows per query" % (n // n_clusters, per_query)
    orc = O.OracleIndex(1, 1)
    orc.vec_init(dim, O.METRIC_IP)
    orc.vec_add(np.arange(n, dtype=np.uint32), X)
    for i in range(12):
        d, l = orc.flat_knn(Q[i], 100)
        assert np.array_equal(lab[i].astype(np.uint32), l), i
        assert np.array_equal(dist[i].view(np.uint32), d.view(np.uint32)), i
    g.close()
