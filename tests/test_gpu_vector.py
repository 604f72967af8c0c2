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
    monkeypatch.setattr(H, "emu_lib_path", H.gpu_lib_path)


test_knn_matches_oracle_flat_scan = E.test_knn_matches_oracle_flat_scan
test_cosine_normalisation_is_bit_exact_and_distances_match = E.test_cosine_normalisation_is_bit_exact_and_distances_match
test_ties_prefer_smaller_label_and_many_slabs = E.test_ties_prefer_smaller_label_and_many_slabs
test_upsert_delete_labels_filters_and_by_id_distances = E.test_upsert_delete_labels_filters_and_by_id_distances
test_pure_vector_search_topster_order_matches_oracle = E.test_pure_vector_search_topster_order_matches_oracle
test_hybrid_rank_fusion_matches_oracle_bit_exactly = E.test_hybrid_rank_fusion_matches_oracle_bit_exactly
test_shard_merge_equals_unsharded = E.test_shard_merge_equals_unsharded


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
