"""Shared test plumbing: synthetic corpora built once by the oracle and mirrored into a tsgpu context, and
HIP-vs-oracle comparison. The oracle (oracle/) is the checker only."""
import os
import subprocess
import numpy as np

from oracle import oracle_py as O
import typesense_amd as T
from typesense_amd import _lib as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def emu_lib_path(defs="", suffix=""):
    """tests/hipemu/_build/libtsgpu_emu.so: the unmodified product sources compiled against the SIMT emulator.
    defs/suffix: a variant build with extra -D flags (e.g. a tiny LDS tile to force the multi-round paths)."""
    if not defs and os.environ.get("TSGPU_EMU_DEFS"):      # run the whole emulator tier on a variant build (experiments: a kernel behind a -D knob)
        defs, suffix = os.environ["TSGPU_EMU_DEFS"], os.environ.get("TSGPU_EMU_SUFFIX", "_env")
    out = subprocess.check_output([os.path.join(ROOT, "tests", "hipemu", "build_emu.sh"), defs, suffix], text=True).strip().splitlines()[-1]
    return out


def gpu_lib_path():
    from typesense_amd import build
    return build.build()


def zipf_docs(n_docs, vocab, tokens_per_doc, seed, s=1.0):
    rng = np.random.default_rng(seed)
    p = 1.0 / np.arange(1, vocab + 1) ** s
    p /= p.sum()
    cdf = np.cumsum(p)
    u = rng.random((n_docs, tokens_per_doc))
    return np.searchsorted(cdf, u).astype(np.uint32) + 1          # term ids 1..vocab (rank order)


def points_of(n_docs):
    ids = np.arange(n_docs, dtype=np.uint64)
    return ((ids * np.uint64(2654435761)) % np.uint64(1000)).astype(np.int64)   # hash(seq_id) mod 1000


def build_pair(docs, lib_path, points=None, n_columns=1, gpu=None):
    """docs: [n_docs][tokens] term ids (plain string field 0). Returns (oracle index, GpuIndex)."""
    n_docs = docs.shape[0]
    orc = O.OracleIndex(1, n_columns)
    for d in range(n_docs):
        orc.index_plain(d, 0, docs[d])
    if points is None:
        points = points_of(n_docs)
    orc.set_sort_dense(0, points)
    g = gpu or T.GpuIndex(0, lib_path)
    g.field_create(0, False)
    for term in orc.terms(0):
        ids, oi, off = orc.dump_posting(0, int(term))
        g.term_upsert(0, int(term), ids, oi, off)
    g.column_set(0, points)
    g.set_num_docs(n_docs)
    g.commit()
    return orc, g


def build_pair_fields(docs_per_field, lib_path, points=None):
    """docs_per_field: list of [n_docs][tokens] term-id arrays, one plain string field each (field ids 0..F-1); a token id 0 =
    no token at that position (fields may be shorter / empty for a document). Returns (oracle index, GpuIndex)."""
    n_fields = len(docs_per_field)
    n_docs = docs_per_field[0].shape[0]
    orc = O.OracleIndex(n_fields, 1)
    for f, docs in enumerate(docs_per_field):
        for d in range(n_docs):
            toks = docs[d][docs[d] != 0]
            if toks.size:
                orc.index_plain(d, f, toks)
    if points is None:
        points = points_of(n_docs)
    orc.set_num_docs(n_docs)
    orc.set_sort_dense(0, points)
    g = T.GpuIndex(0, lib_path)
    for f in range(n_fields):
        g.field_create(f, False)
        for term in orc.terms(f):
            ids, oi, off = orc.dump_posting(f, int(term))
            g.term_upsert(f, int(term), ids, oi, off)
    g.column_set(0, points)
    g.set_num_docs(n_docs)
    g.commit()
    return orc, g


def oracle_keyword(orc, q, cap=2048, ids_cap=0):
    oq = orc.make_query(q.tokens, fields=tuple(q.fields),
                        sort=tuple((s[0], s[2], s[1]) for s in q.sort),
                        fetch_size=10, topster_size=q.topster_size,
                        match_type=q.match_type, prioritize_exact_match=q.prioritize_exact_match,
                        prioritize_token_position=q.prioritize_token_position,
                        prioritize_num_matching_fields=q.prioritize_num_matching_fields, total_cost=q.total_cost,
                        excluded_ids=q.excluded_ids, filter_ids=q.filter_ids, dropped_tokens=getattr(q, "dropped_tokens", None), syn_orig_num_tokens=getattr(q, "syn_orig_num_tokens", -1),
                        orig_num_tokens=getattr(q, "orig_num_tokens", 0), is_synonym_query=getattr(q, "is_synonym_query", False),
                        demote_synonym_match=getattr(q, "demote_synonym_match", False))
    return orc.search_keyword(oq, cap=cap, ids_cap=ids_cap)


def oracle_query(orc, q):
    return orc.make_query(q.tokens, fields=tuple(q.fields), sort=tuple((s[0], s[2], s[1]) for s in q.sort), fetch_size=10,
                          topster_size=q.topster_size, match_type=q.match_type, prioritize_exact_match=q.prioritize_exact_match,
                          prioritize_token_position=q.prioritize_token_position,
                          prioritize_num_matching_fields=q.prioritize_num_matching_fields, total_cost=q.total_cost,
                          excluded_ids=q.excluded_ids, filter_ids=q.filter_ids, dropped_tokens=getattr(q, "dropped_tokens", None), syn_orig_num_tokens=getattr(q, "syn_orig_num_tokens", -1),
                        orig_num_tokens=getattr(q, "orig_num_tokens", 0), is_synonym_query=getattr(q, "is_synonym_query", False),
                        demote_synonym_match=getattr(q, "demote_synonym_match", False))


def oracle_candidates(orc, combos, cap=2048, ids_cap=0):
    """Index::search_all_candidates over the combinations (KwQuery list, pass order) -> (HitList, query_index)"""
    return orc.search_candidates([oracle_query(orc, q) for q in combos], cap=cap, ids_cap=ids_cap)


def oracle_wildcard(orc, q, cap=2048, ids_cap=0):
    oq = orc.make_query([], fields=((0, 0),), sort=tuple((s[0], s[2], s[1]) for s in q.sort), fetch_size=10, topster_size=q.topster_size,
                        excluded_ids=q.excluded_ids, filter_ids=q.filter_ids)
    return orc.search_wildcard(oq, cap=cap, ids_cap=ids_cap)


def assert_hits_equal(hits, qi, ref, what=""):
    n = int(hits.n_hits[qi])
    assert n == ref.keys.size, "%s q%d: n_hits %d vs oracle %d" % (what, qi, n, ref.keys.size)
    assert np.array_equal(hits.keys[qi, :n], ref.keys), "%s q%d: keys differ\n%s\n%s" % (what, qi, hits.keys[qi, :n][:20], ref.keys[:20])
    assert np.array_equal(hits.scores[qi, :n], ref.scores), "%s q%d: scores differ" % (what, qi)
    assert np.array_equal(hits.text_match[qi, :n], ref.text_match), "%s q%d: text_match differ" % (what, qi)
    assert int(hits.num_matched[qi]) == int(ref.num_keyword_matches), "%s q%d: num_matched %d vs %d" % (what, qi, hits.num_matched[qi], ref.num_keyword_matches)


def load_shard(g, orc, lo, hi, n_docs, points, field=0):
    """the postings of documents [lo, hi) of the oracle's field, GLOBAL seq_ids kept, into GpuIndex g (a doc-range shard)"""
    g.field_create(field, False)
    for term in orc.terms(field):
        ids, oi, off = orc.dump_posting(field, int(term))
        sel = np.nonzero((ids >= lo) & (ids < hi))[0]
        if sel.size == 0:
            continue
        ends = np.append(oi[1:], off.size)
        new_off, new_oi = [], []
        for j in sel:
            new_oi.append(len(new_off))
            new_off.extend(off[oi[j]:ends[j]])
        g.term_upsert(field, int(term), ids[sel], new_oi, new_off)
    g.column_set(0, points)
    g.set_num_docs(n_docs)
    g.commit()


def reference_shard_merge(keys, scores, n_hits, k):
    """exact G-way merge of per-shard Topster lists in numpy (test reference for tsgpu_merge_shard_hits_device / kw_shard_merge_kernel):
    keys [G,B,K] int64, scores [G,B,K,3] int64, n_hits [G,B] -> per query (keys[:n], scores[:n]) in KV::is_greater order
    (include/topster.h:146-149: s0, s1, s2, key descending), n = min(k, total hits)"""
    G, Bq, K = keys.shape
    out = []
    for q in range(Bq):
        kk = np.concatenate([keys[g, q, :n_hits[g, q]] for g in range(G)]) if G else np.zeros(0, np.int64)
        sc = np.concatenate([scores[g, q, :n_hits[g, q]] for g in range(G)]) if G else np.zeros((0, 3), np.int64)
        order = np.lexsort((kk, sc[:, 2], sc[:, 1], sc[:, 0]))[::-1][:k] if kk.size else np.zeros(0, np.int64)
        out.append((kk[order], sc[order]))
    return out


# ---------------------------------------------------------------- BASELINE configs 3 / 4 at size (tests/test_gpu_at_size.py, bench.py parity legs)
VEC_SLAB = 1 << 20


def vector_slab(n_docs, dim, a, b, normalize=False, clustered=False):
    """base vectors of global rows [a, b) of the config-3 collection (i.i.d. N(0,1), seed 3) as a CUDA tensor: the collection is
    defined on a fixed grid of 2^20-row slabs (seed 3 + slab start), so every shard / chunk / test regenerates exactly the same rows"""
    import torch
    from typesense_amd import synth
    parts = []
    for s0 in range(a // VEC_SLAB * VEC_SLAB, b, VEC_SLAB):
        n = min(VEC_SLAB, n_docs - s0)
        x = synth.random_vectors(n, dim, seed=3 + s0, device="cuda")
        if clustered:
            # 1024 tight clusters: centre + 0.15 x noise, unit length — many near-ties around every query's k-th neighbour
            gcen = torch.Generator(device="cuda")
            gcen.manual_seed(77)
            cen = torch.randn((1024, dim), generator=gcen, device="cuda", dtype=torch.float32)
            gi = torch.Generator(device="cuda")
            gi.manual_seed(78 + s0)
            idx = torch.randint(0, 1024, (n,), generator=gi, device="cuda")
            x = cen[idx] + 0.15 * x
        if normalize or clustered:
            x /= (x.norm(dim=1, keepdim=True) + 1e-30)
        parts.append(x[max(a, s0) - s0:min(b, s0 + n) - s0])
    return parts[0] if len(parts) == 1 else torch.cat(parts)


def load_vector_field(g, field, metric, n_docs, dim, lo=0, hi=None, **slab_kw):
    """rows [lo, hi) of the config-3 collection into vector field `field` of GpuIndex g, generated on the device slab by slab (label = row)"""
    import torch
    hi = n_docs if hi is None else hi
    g.vec_create(field, dim, metric, hi - lo)
    for a in range(lo, hi, VEC_SLAB):
        b = min(hi, a + VEC_SLAB)
        x = vector_slab(n_docs, dim, a, b, **slab_kw).contiguous()
        labels = torch.arange(a, b, dtype=torch.int64, device="cuda")
        g.vec_upsert_device(field, labels.data_ptr(), x.data_ptr(), b - a)
        del x
    torch.cuda.synchronize()


def exact_knn_chunked(n_docs, dim, queries, k, want_rows=True, slab_kw=None, threads=None):
    """The oracle's flat_knn (exact fp32, hnswlib summation order, ties -> smaller label; process_results_bruteforce,
    /root/reference/src/index.cpp:3345-3374) over ALL rows of the config-3 collection, which is regenerated slab by slab on the GPU, copied
    to the host ONCE per slab and scanned by the oracle, one query per host thread.
    queries: {oracle metric: Qh [nq, dim]} (several metrics share the slab copy; a cosine oracle normalises the rows on add like
    hnsw_index_t::normalize_vector, include/index.h:379-388). Returns {metric: (dist [nq,k], labels [nq,k], {label: raw row} of the final top-k)}."""
    from concurrent.futures import ThreadPoolExecutor
    state = {m: dict(d=np.zeros((Q.shape[0], 0), np.float32), l=np.zeros((Q.shape[0], 0), np.uint32), keep={}) for m, Q in queries.items()}
    nthreads = threads or min(max(Q.shape[0] for Q in queries.values()), os.cpu_count() or 1)
    pool = ThreadPoolExecutor(max_workers=nthreads)
    for a in range(0, n_docs, VEC_SLAB):
        b = min(n_docs, a + VEC_SLAB)
        xs = vector_slab(n_docs, dim, a, b, **(slab_kw or {})).cpu().numpy()
        for metric, Qh in queries.items():
            st = state[metric]
            nq = Qh.shape[0]
            orc = O.OracleIndex(1, 1)
            orc.vec_init(dim, metric)
            orc.vec_add(np.arange(a, b, dtype=np.uint32), xs)
            res = list(pool.map(lambda i: orc.flat_knn(Qh[i], k), range(nq)))          # ctypes releases the GIL: one query per thread
            cd = np.stack([np.pad(r[0], (0, k - r[0].size), constant_values=np.inf) for r in res])
            cl = np.stack([np.pad(r[1], (0, k - r[1].size), constant_values=0xFFFFFFFF) for r in res]).astype(np.uint32)
            if want_rows:
                for i in range(nq):
                    for lab in res[i][1]:
                        st["keep"].setdefault(int(lab), xs[int(lab) - a].copy())
            d = np.concatenate([st["d"], cd], axis=1)
            l = np.concatenate([st["l"], cl], axis=1)
            order = np.lexsort((l, d), axis=1)[:, :k]                                   # (distance, label) ascending
            st["d"], st["l"] = np.take_along_axis(d, order, 1), np.take_along_axis(l, order, 1)
            orc.close()
        del xs
    pool.shutdown()
    out = {}
    for metric, st in state.items():
        final = set(int(x) for x in st["l"].ravel())
        out[metric] = (st["d"], st["l"], {lab: v for lab, v in st["keep"].items() if lab in final})
    return out


def oracle_for_hybrid_at_size(n_docs, dim, csr, points, qtok, rows, metric=None):
    """an oracle index able to answer search_hybrid for the queries `qtok` at full size without holding the 30 GB matrix: the keyword half gets
    the decoded postings of the queries' terms; the vector store gets `rows` = the rows of the oracle's OWN exact top-k over all rows
    (exact_knn_chunked): flat_knn over a superset of the true top-k returns exactly the true top-k. Returns (orc, labels held)."""
    from typesense_amd import synth
    orc = O.OracleIndex(1, 1)
    orc.set_num_docs(n_docs)
    orc.set_sort_dense(0, points)
    for t in np.unique(np.asarray(qtok).ravel()):
        ids, oi, off = synth.csr_term(csr, int(t))
        if ids.size:
            orc.load_posting(0, int(t), ids, oi, off)
    orc.vec_init(dim, O.METRIC_IP if metric is None else metric)
    labs = np.array(sorted(rows.keys()), np.uint32)
    orc.vec_add(labs, np.stack([rows[int(x)] for x in labs]))
    return orc, labs


def oracle_add_rows(orc, n_docs, dim, have, need, **slab_kw):
    """adds the rows `need` (labels) the oracle's vector store does not hold yet (rerank_hybrid_matches computes the distance of keyword-only
    hits BY LABEL, /root/reference/src/index.cpp:8860-8876), regenerated slab by slab on the GPU"""
    import torch
    need = np.unique(np.asarray(need).astype(np.int64))
    need = need[~np.isin(need, np.asarray(have).astype(np.int64))]
    for s0 in range(0, n_docs, VEC_SLAB):
        sel = need[(need >= s0) & (need < s0 + VEC_SLAB)]
        if sel.size:
            x = vector_slab(n_docs, dim, s0, min(n_docs, s0 + VEC_SLAB), **slab_kw)[torch.from_numpy(sel - s0).cuda()].cpu().numpy()
            orc.vec_add(sel.astype(np.uint32), x)
    return need


def facet_csr_of(n_docs, seed=77, n_values=40, max_per_doc=3):
    """a facet field's hash index as CSR by seq_id (doc_ptr [n_docs + 1], hashes): 0..max_per_doc values per document out of n_values (duplicates inside a
    document occur: counted once per document by the walk), deterministic in (n_docs, seed)"""
    rng = np.random.default_rng(seed)
    per = rng.integers(0, max_per_doc + 1, size=n_docs)
    ptr = np.zeros(n_docs + 1, np.uint64)
    ptr[1:] = np.cumsum(per)
    hashes = (rng.integers(0, n_values, size=int(ptr[-1])).astype(np.uint32) * np.uint32(2654435761) >> np.uint32(7)).astype(np.uint32)
    return ptr, hashes


def facet_csr_shard(ptr, hashes, lo, hi):
    """the same index as a shard holds it: the documents outside [lo, hi) have no hashes (global seq_ids kept)"""
    n_docs = ptr.size - 1
    per = np.diff(ptr).astype(np.int64)
    keep = np.zeros(n_docs, bool)
    keep[lo:hi] = True
    per2 = np.where(keep, per, 0)
    ptr2 = np.zeros(n_docs + 1, np.uint64)
    ptr2[1:] = np.cumsum(per2)
    return ptr2, hashes[int(ptr[lo]):int(ptr[hi])].copy()
