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


def _random_facet_index(rng, n_docs, n_values, array=True):
    per = rng.integers(0, 5, size=n_docs) if array else (rng.random(n_docs) < 0.9).astype(np.int64)
    ptr = np.zeros(n_docs + 1, np.uint64)
    ptr[1:] = np.cumsum(per)
    hashes = (rng.zipf(1.3, size=int(ptr[-1])) % n_values).astype(np.uint32) * np.uint32(2654435761)      # repeats inside a document happen
    return ptr, hashes


def _check(g, orc, lists, **kw):
    got = g.facet_count_batch(0, lists, cap=4096, **kw)
    for q, ids in enumerate(lists):
        h, c, d, p, n = orc.facet_count(0, ids, sample_mod=kw.get("sample_mod", 1), allowed_hashes=kw.get("allowed_hashes"))
        gh, gc, gd, gp, gn = got[q]
        assert gn == n, "query %d: %d distinct values, oracle %d" % (q, gn, n)
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
    g.close()


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
