"""The drop-in boundary as a server maintainer meets it: the header-only C++ shims (typesense_amd/csrc/host/) are compiled with
g++ against mock types of the reference's shape (tests/host_shims/shim_driver.cpp), linked to the C-ABI library and checked
against the oracle — the posting decode shim (block chains + compact lists -> tsgpu_term_upsert), the keyword seam
(search_across_fields_gpu<KV, Topster>, per-call id lists), its grouped form (build_distinct_column + search_across_fields_grouped_gpu over the
distinct Topster), the hnswlib-shaped adaptor INCLUDING a VectorFilterFunctor-style
predicate passed without a candidate list (src/index.cpp:3384-3386), mirror_hnsw_graph, and the input validators.
CPU tier: links the emulator build of the unmodified product sources. The `-m gpu` twin links libtsgpu.so."""
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import oracle_py as O
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _u32(x):
    return struct.pack("<I", int(x))


def _write_fixture(d):
    rng = np.random.default_rng(21)
    n_docs = 900
    docs = H.zipf_docs(n_docs, 120, 9, seed=4)
    orc = O.OracleIndex(1, 1)
    for i in range(n_docs):
        orc.index_plain(i, 0, docs[i])
    pts = H.points_of(n_docs)
    orc.set_sort_dense(0, pts)
    with open(os.path.join(d, "postings.bin"), "wb") as f:
        terms = orc.terms(0)
        f.write(_u32(terms.size))
        for t in terms:
            ids, oi, off = orc.dump_posting(0, int(t))
            f.write(_u32(t) + _u32(ids.size) + _u32(off.size) + ids.tobytes() + oi.tobytes() + off.tobytes())
    with open(os.path.join(d, "points.bin"), "wb") as f:
        f.write(_u32(n_docs) + pts.astype(np.int64).tobytes())
    sort = ((O.SORT_TEXT_MATCH, 0, 1), (O.SORT_INT64_COLUMN, 0, 1))
    with open(os.path.join(d, "queries.bin"), "wb") as f:
        qs = []
        for i in range(14):
            toks = rng.choice(np.arange(1, 25), size=int(rng.integers(1, 4)), replace=False).astype(np.uint32)
            filt = np.sort(rng.choice(n_docs, size=300, replace=False)).astype(np.uint32) if i % 3 == 1 else np.zeros(0, np.uint32)
            excl = np.sort(rng.choice(n_docs, size=40, replace=False)).astype(np.uint32) if i % 4 == 2 else np.zeros(0, np.uint32)
            qs.append((toks, filt, excl, 40 if i % 2 else 250))
        f.write(_u32(len(qs)))
        for toks, filt, excl, tsz in qs:
            q = orc.make_query(toks, sort=sort, fetch_size=10, topster_size=tsz, filter_ids=filt if filt.size else None, excluded_ids=excl if excl.size else None)
            ref = orc.search_keyword(q, cap=1024, ids_cap=4096)
            f.write(_u32(toks.size) + toks.tobytes() + _u32(filt.size) + filt.tobytes() + _u32(excl.size) + excl.tobytes() + _u32(tsz))
            f.write(_u32(ref.keys.size) + ref.keys.astype(np.uint64).tobytes() + ref.scores.astype(np.int64).tobytes())
            f.write(struct.pack("<Q", int(ref.num_keyword_matches)) + _u32(ref.result_ids.size) + ref.result_ids.astype(np.uint32).tobytes())
    # group_by: the facet hash index of one group_by field (CSR), Index::get_distinct_id per document, grouped passes over all matched documents
    gptr = np.zeros(n_docs + 1, np.uint64)
    ghs = []
    for dd in range(n_docs):
        if dd % 13 != 5:
            ghs.append(int(rng.integers(1, 30)) * 40503 % (2**32))
        gptr[dd + 1] = len(ghs)
    ghs = np.array(ghs, np.uint32)
    distinct, has_value = O.distinct_ids(n_docs, [(gptr, ghs)], False)
    with open(os.path.join(d, "grouped.bin"), "wb") as f:
        f.write(_u32(n_docs) + gptr.tobytes() + _u32(ghs.size) + ghs.tobytes() + distinct.astype(np.uint64).tobytes())
        cases = [([1], 250, 3, 1), ([1], 250, 3, 0), ([2, 3], 5, 2, 1), ([2, 3], 5, 2, 0), ([4], 10, 1, 0), ([1, 2, 3], 250, 4, 1), ([7], 3, 5, 0), ([9999], 10, 2, 1)]
        f.write(_u32(len(cases)))
        for toks, tsz, limit, first in cases:
            toks = np.array(toks, np.uint32)
            q = orc.make_query(toks, sort=sort, fetch_size=10, topster_size=tsz)
            ref = orc.search_keyword_grouped(q, distinct, limit, bool(first), has_value=has_value, ids_cap=4096)
            f.write(_u32(toks.size) + toks.tobytes() + _u32(tsz) + _u32(limit) + _u32(first) + _u32(ref.n_groups))
            for gi in range(ref.n_groups):
                a, b = int(ref.begin[gi]), int(ref.begin[gi + 1])
                f.write(struct.pack("<Q", int(ref.distinct_key[gi])) + _u32(ref.group_found[gi]) + _u32(b - a) + ref.keys[a:b].astype(np.uint64).tobytes() + ref.scores[a:b].astype(np.int64).tobytes())
            f.write(struct.pack("<Q", ref.groups_count) + _u32(ref.missing_ids.size) + ref.missing_ids.astype(np.uint32).tobytes() + struct.pack("<Q", ref.num_keyword_matches))
    # do_facets (hash-index branch): an array facet field, the sort-index column of a range facet, the distinct ids of a grouped search; expectations = the oracle's walk
    fptr = np.zeros(n_docs + 1, np.uint64)
    fptr[1:] = np.cumsum(rng.integers(0, 4, size=n_docs))
    fh = (rng.zipf(1.4, size=int(fptr[-1])) % 40).astype(np.uint32) * np.uint32(2654435761)
    fvals = rng.integers(-500, 3000, size=n_docs).astype(np.int64)
    forc = O.OracleIndex(1, 1)
    forc.facet_set(0, fptr, fh)
    with open(os.path.join(d, "facets.bin"), "wb") as f:
        f.write(_u32(n_docs) + fptr.tobytes() + fh.tobytes() + fvals.tobytes() + distinct.astype(np.uint64).tobytes())
        all_ids = np.arange(n_docs, dtype=np.uint32)
        some = np.sort(rng.choice(n_docs, size=200, replace=False)).astype(np.uint32)
        fq = np.unique(fh)[::3]
        ranges = [(0, -400), (1000, 0), (1001, 1000), (2500, 1500)]
        cases = [(all_ids, 1, None, 0, None), (some, 1, None, 0, None), (all_ids, 3, None, 0, None), (all_ids, 1, fq, 0, None), (all_ids, 1, np.zeros(0, np.uint32), 0, None),
                 (all_ids, 1, None, 1, None), (some, 2, fq, 1, None), (all_ids, 1, None, 0, ranges), (some, 3, None, 0, ranges), (all_ids, 1, None, 1, ranges), (np.zeros(0, np.uint32), 1, None, 0, None)]
        f.write(_u32(len(cases)))
        for ids, mod, allowed, grouped, rngs in cases:
            k, c, dd, p, nn = forc.facet_count_ex(0, ids, sample_mod=mod, allowed_hashes=allowed if (allowed is not None and allowed.size) else None, ranges=rngs, doc_vals=fvals if rngs else None,
                                                  distinct_ids=distinct if grouped else None)
            if allowed is not None and allowed.size == 0:
                k, c, dd, p = k[:0], c[:0], dd[:0], p[:0]                         # a facet query that matched no value: nothing is counted (:1751)
            f.write(_u32(ids.size) + ids.tobytes() + _u32(mod) + _u32(allowed is not None) + _u32(allowed.size if allowed is not None else 0) + (allowed.tobytes() if allowed is not None else b""))
            f.write(_u32(grouped) + _u32(len(rngs) if rngs else 0) + b"".join(struct.pack("<qq", u, l) for u, l in (rngs or [])))
            f.write(_u32(k.size) + b"".join(struct.pack("<QIII", int(a), int(b), int(x), int(y)) for a, b, x, y in zip(k, c, dd, p)))
    with open(os.path.join(d, "grouped_candidates.bin"), "wb") as f:
        f.write(_u32(n_docs) + has_value.astype(np.uint8).tobytes())
        ccases = [([[1, 2], [1, 3], [2, 3], [1, 2], [9999]], 30, 2, 1), ([[1, 2], [1, 3], [2, 3], [1, 2], [9999]], 30, 2, 0), ([[3], [4], [5]], 4, 3, 0), ([[3], [4], [5]], 4, 3, 1)]
        f.write(_u32(len(ccases)))
        for cs, tsz, limit, first in ccases:
            oqs = [orc.make_query(np.array(c, np.uint32), sort=sort, fetch_size=10, topster_size=tsz, total_cost=int(j > 0)) for j, c in enumerate(cs)]
            ref, rqi = orc.search_candidates_grouped(oqs, distinct, limit, bool(first), has_value=has_value, ids_cap=1 << 16)
            f.write(_u32(len(cs)) + _u32(tsz) + _u32(limit) + _u32(first))
            for c in cs:
                f.write(_u32(len(c)) + np.array(c, np.uint32).tobytes())
            f.write(_u32(ref.n_groups))
            for gi in range(ref.n_groups):
                a, b = int(ref.begin[gi]), int(ref.begin[gi + 1])
                f.write(struct.pack("<Q", int(ref.distinct_key[gi])) + _u32(ref.group_found[gi]) + _u32(b - a) + ref.keys[a:b].astype(np.uint64).tobytes()
                        + ref.scores[a:b].astype(np.int64).tobytes() + rqi[a:b].astype(np.uint16).tobytes())
            f.write(struct.pack("<Q", ref.groups_count) + _u32(ref.result_ids.size) + ref.result_ids.astype(np.uint32).tobytes() + struct.pack("<Q", ref.num_keyword_matches))
    # vectors
    n, dim, k = 400, 40, 12
    X = rng.standard_normal((n, dim)).astype(np.float32)
    Q = rng.standard_normal((5, dim)).astype(np.float32)
    ov = O.OracleIndex(1, 1)
    ov.vec_init(dim, O.METRIC_IP)
    ov.vec_add(np.arange(n, dtype=np.uint32), X)
    all_but_deleted = np.array([i for i in range(n) if i != 4], np.uint32)
    even = np.array([i for i in range(0, n, 2) if i not in (4, 10, 20)], np.uint32)
    with open(os.path.join(d, "vec.bin"), "wb") as f:
        f.write(_u32(n) + _u32(dim) + X.tobytes() + _u32(Q.shape[0]) + _u32(k) + Q.tobytes())
        for qi in range(Q.shape[0]):
            for allow in (all_but_deleted, even):
                dd, ll = ov.flat_knn(Q[qi], k, allow_ids=allow)
                f.write(_u32(ll.size) + ll.astype(np.uint64).tobytes() + dd.astype(np.float32).tobytes())
    # hnsw graph: built by the oracle's hnswlib restatement, exported in hnswlib's own layout
    n2, dim2 = 300, 24
    X2 = rng.standard_normal((n2, dim2)).astype(np.float32)
    Q2 = rng.standard_normal((4, dim2)).astype(np.float32)
    oh = O.OracleIndex(1, 1)
    oh.vec_init(dim2, O.METRIC_IP)
    oh.vec_add(np.arange(n2, dtype=np.uint32), X2)
    oh.hnsw_build(M=8, ef_construction=60, seed=100)
    g = oh.hnsw_export()
    with open(os.path.join(d, "hnsw.bin"), "wb") as f:
        f.write(_u32(n2) + _u32(dim2) + _u32(g["M"]) + struct.pack("<i", g["maxlevel"]) + _u32(g["enterpoint"]) + X2.tobytes())
        f.write(g["levels"].astype(np.uint32).tobytes() + g["link0"].astype(np.uint32).tobytes() + g["upper_ptr"].astype(np.uint64).tobytes())
        f.write(_u32(g["upper_links"].shape[0]) + g["upper_links"].astype(np.uint32).tobytes())
        kk, ef = 10, 40
        f.write(_u32(Q2.shape[0]) + _u32(kk) + _u32(ef) + Q2.tobytes())
        for qi in range(Q2.shape[0]):
            dd, ll, _ = oh.hnsw_search(Q2[qi], kk, ef, functor_present=True)
            f.write(_u32(ll.size) + ll.astype(np.uint64).tobytes() + dd.astype(np.float32).tobytes())


def _build_and_run(lib_path, tmp_path):
    d = str(tmp_path)
    _write_fixture(d)
    exe = os.path.join(d, "shim_driver")
    libdir, libname = os.path.dirname(lib_path), os.path.basename(lib_path)
    assert libname.startswith("lib") and libname.endswith(".so")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wno-unused-function", "-o", exe, os.path.join(ROOT, "tests", "host_shims", "shim_driver.cpp"),
           "-L" + libdir, "-l" + libname[3:-3], "-Wl,-rpath," + libdir, "-lpthread"]
    subprocess.check_call(cmd)
    out = subprocess.run([exe, d], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().startswith("OK "), out.stdout
    assert int(out.stdout.split()[1]) > 100


def test_host_shims_compile_and_match_oracle_emulator(tmp_path):
    _build_and_run(H.emu_lib_path(), tmp_path)


@pytest.mark.gpu
def test_host_shims_compile_and_match_oracle_gpu(tmp_path):
    _build_and_run(H.gpu_lib_path(), tmp_path)
