"""tsgpu_vec_hnsw_build on the SIMT emulator: the bulk construction of the HNSW graph in batches on the device (csrc/vec_hnsw_build.hip.h) equals, link for
link, the oracle's restatement of the same batched algorithm (oracle/hnsw_graph.h bulk_build: hnswlib's level draw, searchBaseLayer, neighbour heuristic and
reverse-link rule, applied per batch); the graph is valid, searchable, and as good as the one hnswlib's row-by-row insertion gives. hnswlib itself is not under
/root/reference: PARITY UNPINNED, like the search."""
import numpy as np
import pytest

import typesense_amd as T
from typesense_amd import _lib as B
from oracle import oracle_py as O
from tests import helpers as H
from tests.test_emu_vector import _graphs_equal


def _latent(rng, n, dim, latent=8, noise=0.3):
    Z = rng.standard_normal((n, latent)).astype(np.float32)
    P = rng.standard_normal((latent, dim)).astype(np.float32)
    X = Z @ P + noise * rng.standard_normal((n, dim)).astype(np.float32)
    return (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)


def _check_structure(gr, n, M):
    cnts = gr["link0"][:, 0]
    assert cnts.max() <= 2 * M and (cnts > 0).all()
    for i in range(n):
        l = gr["link0"][i, 1:1 + cnts[i]]
        assert (l < n).all() and i not in l and np.unique(l).size == l.size, i
        assert (gr["link0"][i, 1 + cnts[i]:] == 0).all()                     # canonical image: slots behind the count are zero


@pytest.mark.parametrize("n,dim,M,efc,metric,seed_min,max_batch", [(450, 32, 8, 40, B.METRIC_IP, 100, 0), (500, 24, 4, 24, B.METRIC_COSINE, 64, 100), (300, 16, 16, 48, B.METRIC_IP, 80, 0)])
def test_bulk_build_on_the_device_equals_the_oracles_batched_build_link_for_link(n, dim, M, efc, metric, seed_min, max_batch):
    rng = np.random.default_rng(7 + n)
    X = _latent(rng, n, dim) if metric == B.METRIC_IP else rng.standard_normal((n, dim)).astype(np.float32)
    X[n // 2] = X[n // 2 - 1]                                                  # a duplicate row: equal distances everywhere it appears
    g = T.GpuIndex(0, H.emu_lib_path())
    g.vec_create(1, dim, metric)
    g.vec_upsert(1, np.arange(n, dtype=np.uint64), X)
    info = g.vec_hnsw_build(1, M=M, ef_construction=efc, seed=100, threads=1, seed_min=seed_min, max_batch=max_batch)
    assert info["n"] == n and info["unlinked"] == 0 and info["n_batches"] >= 3 and seed_min <= info["n_seed"] < n // 2, info
    orc = O.OracleIndex(1, 1)
    orc.vec_init(dim, metric)
    orc.vec_add(np.arange(n, dtype=np.uint32), X)
    orc.hnsw_bulk_build(M=M, ef_construction=efc, seed=100, seed_min=seed_min, max_batch=max_batch)
    mine, ref = g.vec_hnsw_export(1), orc.hnsw_export()
    _check_structure(mine, n, M)
    bad = [i for i in range(n) if not np.array_equal(mine["link0"][i, :1 + mine["link0"][i, 0]], ref["link0"][i, :1 + ref["link0"][i, 0]])]
    assert not bad, "level-0 lists differ at %d nodes, first %s: %s vs %s" % (len(bad), bad[:5], mine["link0"][bad[0]], ref["link0"][bad[0]])
    assert _graphs_equal(mine, ref)
    # the search serves the device-resident graph and replays the oracle's traversal of the same graph
    Q = rng.standard_normal((6, dim)).astype(np.float32)
    for k, ef in ((10, 10), (10, 60), (10, 200)):                               # (ef 200: the 256-entry LDS tier)
        dist, lab, cnt = g.vec_hnsw_search_batch(1, Q, k, ef)
        for i in range(Q.shape[0]):
            d, l, _ = orc.hnsw_search(Q[i], k, ef, functor_present=True)
            assert cnt[i] == d.size and np.array_equal(lab[i, :d.size], l) and np.array_equal(dist[i, :d.size].view(np.uint32), d.view(np.uint32)), (k, ef, i)
    # a candidate heap that outgrows its LDS tier: those queries — and only those — run again on the largest tier (forced here with a 24-entry heap), same answer
    g.set_option("hnsw_test_tiny_cand", 1)
    r0 = g.counter("hnsw_tier_reruns")
    dist, lab, cnt = g.vec_hnsw_search_batch(1, Q, 10, 60)
    assert g.counter("hnsw_tier_reruns") > r0
    g.set_option("hnsw_test_tiny_cand", 0)
    for i in range(Q.shape[0]):
        d, l, _ = orc.hnsw_search(Q[i], 10, 60, functor_present=True)
        assert cnt[i] == d.size and np.array_equal(lab[i, :d.size], l) and np.array_equal(dist[i, :d.size].view(np.uint32), d.view(np.uint32)), ("re-run", i)
    g.close()


def test_bulk_built_graph_is_as_good_as_the_row_by_row_one():
    """recall@10 at ef = 40 of the batched build vs hnswlib's incremental insertion (tsgpu_vec_hnsw_enable) on the same rows, both against the exact scan"""
    n, dim, M, efc = 600, 24, 8, 40
    rng = np.random.default_rng(99)
    X = _latent(rng, n, dim)
    Q = _latent(rng, 40, dim)
    lib = H.emu_lib_path()
    rec = {}
    for how in ("bulk", "incremental"):
        g = T.GpuIndex(0, lib)
        g.vec_create(1, dim, B.METRIC_IP)
        if how == "incremental":
            g.vec_hnsw_enable(1, M=M, ef_construction=efc, seed=100, threads=1)
        g.vec_upsert(1, np.arange(n, dtype=np.uint64), X)
        if how == "bulk":
            info = g.vec_hnsw_build(1, M=M, ef_construction=efc, seed=100, threads=2, seed_min=100)         # (two host threads on the seed set: not deterministic, valid)
            assert info["unlinked"] == 0
            _check_structure(g.vec_hnsw_export(1), n, M)
        lab = g.vec_hnsw_search_batch(1, Q, 10, 40)[1]
        exact = g.vec_knn_batch(1, Q, 10)[1]
        rec[how] = float(np.mean([len(set(lab[i].tolist()) & set(exact[i].tolist())) for i in range(Q.shape[0])])) / 10
        g.close()
    assert rec["bulk"] >= 0.9 and rec["bulk"] >= rec["incremental"] - 0.03, rec


def test_bulk_build_edge_cases():
    lib = H.emu_lib_path()
    rng = np.random.default_rng(5)
    for n in (0, 1, 2, 40):                                                     # fewer rows than seed_min: everything is inserted on the host
        g = T.GpuIndex(0, lib)
        g.vec_create(1, 8, B.METRIC_IP)
        X = rng.standard_normal((max(n, 1), 8)).astype(np.float32)
        if n:
            g.vec_upsert(1, np.arange(n, dtype=np.uint64), X[:n])
        info = g.vec_hnsw_build(1, M=4, ef_construction=16, seed=100)
        assert info["n"] == n and info["n_seed"] == n and info["n_batches"] == 0
        if n:
            dist, lab, cnt = g.vec_hnsw_search_batch(1, X[:1], min(n, 3), 16)
            assert cnt[0] == min(n, 3) and lab[0, 0] == 0
        g.close()
    g = T.GpuIndex(0, lib)
    g.vec_create(1, 8, B.METRIC_IP)
    g.vec_upsert(1, np.arange(10, dtype=np.uint64), rng.standard_normal((10, 8)).astype(np.float32))
    for bad in (dict(M=1), dict(M=32), dict(ef_construction=2000)):
        with pytest.raises(B.TsgpuError):
            g.vec_hnsw_build(1, **bad)
    with pytest.raises(B.TsgpuError):
        g.vec_hnsw_build(9)                                                     # unknown field
    g.vec_hnsw_enable(1, M=4, ef_construction=16)
    with pytest.raises(B.TsgpuError):
        g.vec_hnsw_build(1, M=4)                                                # the field inserts incrementally
    g.close()
