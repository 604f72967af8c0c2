"""One rank of tests/test_dist_gloo.py (CPU tier: emulator build, TSGPU_WORKER_LIB = its path) and tests/test_gpu_dist.py (`-m gpu`:
the real libtsgpu.so, the ranks share the one MI355X). Drives the PRODUCT's rank-form exchange — tsgpu_group_create_rank_host +
tsgpu_group_keyword_search_batch / _vec_knn_batch / _hybrid_search_batch — across PROCESSES: the packed exchange blocks, the slice /
all-gather exchanges, the merge kernels and the replicas form are the library's; only the wire is the launcher's (torch.distributed
gloo through the two host-collective callbacks). Every rank compares the merged result with the UNSHARDED oracle bit for bit
(Topster order include/topster.h:146-154). typesense_amd/hostcoll.py only supplies the HOST transport's two callbacks."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import typesense_amd as T                         # noqa: E402
from typesense_amd import _lib as B, hostcoll as D    # noqa: E402
from oracle import oracle_py as O                 # noqa: E402
from tests import helpers as H                    # noqa: E402


def load_shard(g, orc, lo, hi, pts, n_docs, X, dim):
    """postings restricted to [lo, hi), GLOBAL seq_ids kept; the sort column and the vectors of the same range"""
    g.field_create(0, False)
    for term in orc.terms(0):
        ids, oi, off = orc.dump_posting(0, int(term))
        sel = np.nonzero((ids >= lo) & (ids < hi))[0]
        if sel.size == 0:
            continue
        ends = np.append(oi[1:], off.size)
        new_off, new_oi = [], []
        for j in sel:
            new_oi.append(len(new_off))
            new_off.extend(off[oi[j]:ends[j]])
        g.term_upsert(0, int(term), ids[sel], new_oi, new_off)
    g.column_set(0, pts)
    from tests.test_emu_groupby import group_column
    g.column_set(1, group_column(n_docs, seed=3)[0].view(np.int64))              # the group_by column (Index::get_distinct_id per document)
    g.set_num_docs(n_docs)
    g.commit()
    g.set_option("doc_range_lo", lo); g.set_option("doc_range_hi", hi)          # the seq_ids this shard owns (q = * ranks only those)
    fptr, fhash = H.facet_csr_of(n_docs)
    g.facet_set(5, *H.facet_csr_shard(fptr, fhash, lo, hi))
    g.vec_create(1, dim, B.METRIC_IP)
    if hi > lo:
        g.vec_upsert(1, np.arange(lo, hi, dtype=np.uint64), X[lo:hi])


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = os.environ["TSGPU_WORKER_LIB"]
    n_docs, dim, K, k_vec = int(os.environ.get("TSGPU_WORKER_DOCS", "1500")), 24, 40, 12
    docs = H.zipf_docs(n_docs, 80, 10, seed=8)                      # every rank derives the same collection
    pts = H.points_of(n_docs)
    rng = np.random.default_rng(3)
    X = rng.standard_normal((n_docs, dim)).astype(np.float32)
    X[7] = X[900 % n_docs]                                           # duplicate embeddings across shards: distance ties -> smaller label first
    Q = rng.standard_normal((5, dim)).astype(np.float32)
    orc = O.OracleIndex(1, 1)                                        # full-collection oracle (the checker)
    for d in range(n_docs):
        orc.index_plain(d, 0, docs[d])
    orc.set_sort_dense(0, pts)
    orc.vec_init(dim, O.METRIC_IP)
    orc.vec_add(np.arange(n_docs, dtype=np.uint32), X)
    orc.facet_set(5, *H.facet_csr_of(n_docs))
    sort = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
    OSORT = ((O.SORT_TEXT_MATCH, 0, 1), (O.SORT_INT64_COLUMN, 0, 1))
    toks = ([1, 2], [3, 1, 2], [5], [4, 9], [79, 80])                # 5 queries: not a multiple of the world size (padded slices); the last one: the two rarest
                                                                     # terms — some shards hold no posting of one of them (an EMPTY list there, not a dropped token)
    qs = [T.KwQuery(t, sort=sort, topster_size=K) for t in toks]
    qs[3] = T.KwQuery(toks[3], sort=sort, topster_size=K, filter_ids=np.arange(0, n_docs, 2, dtype=np.uint32), excluded_ids=np.array([8, 64], np.uint32))
    ag, a2a = D.torch_collectives()
    ok = True
    log = []

    def check(what, cond):
        nonlocal ok
        if not cond:
            log.append("rank %d: MISMATCH %s" % (rank, what))
            ok = False

    def check_keyword(what, h):
        check(what + " status", (h.status == 0).all())
        for i, q in enumerate(qs):
            ref = H.oracle_keyword(orc, q)
            m = int(h.n_hits[i])
            check("%s q%d keys" % (what, i), m == ref.keys.size and np.array_equal(h.keys[i, :m], ref.keys))
            check("%s q%d scores" % (what, i), m == ref.keys.size and np.array_equal(h.scores[i, :m], ref.scores))
            check("%s q%d num_matched" % (what, i), int(h.num_matched[i]) == int(ref.num_keyword_matches))

    def check_knn(what, dm, lm, cm, allow=None):
        for i in range(Q.shape[0]):
            d, l = orc.flat_knn(Q[i], k_vec, allow_ids=allow) if allow is not None else orc.flat_knn(Q[i], k_vec)
            check("%s q%d" % (what, i), int(cm[i]) == l.size and np.array_equal(lm[i, :l.size].astype(np.uint32), l) and
                  np.array_equal(dm[i, :l.size].view(np.uint32), d.view(np.uint32)))

    # (uneven cut, incl. an EMPTY shard on the last rank) x (slice exchange, literal all-gather)
    cuts = {"uneven": [0] + [int(n_docs * (0.62 + 0.3 * r / world)) for r in range(world - 1)] + [n_docs],
            "empty_last": [0] + [n_docs * (r + 1) // (world - 1) for r in range(world - 1)] + [n_docs]}
    for cut_name, edges in cuts.items():
        lo, hi = edges[rank], edges[rank + 1]
        g = T.GpuIndex(0, lib)
        load_shard(g, orc, lo, hi, pts, n_docs, X, dim)
        grp = T.GpuGroup.join_host(g, rank, world, ag, a2a)
        check("group size", grp.size() == world)
        for slices in (1, 0):
            grp.set_option("kw_exchange_slices", slices)
            vol = {}
            for pruned in (1, 0):                  # bound-pruned exchange (default) and the full top-k exchange: same merged result
                grp.set_option("kw_exchange_pruned", pruned)
                tag = "%s/slices=%d/pruned=%d" % (cut_name, slices, pruned)
                check_keyword("keyword " + tag, grp.keyword_search_batch(qs, K, k_stride=K))
                vol[pruned] = int(grp.timings().hit_exchange_bytes_per_member)
            check("exchange volumes are reported (%s, slices=%d): %s" % (cut_name, slices, vol), vol[1] > 0 and vol[0] > 0)   # (HOST callbacks move equal-sized slices padded to the largest: five queries save nothing here; tests/test_group.py measures the exact-size form)
            grp.set_option("kw_exchange_pruned", 1)
        # kw_own_slice_only: a rank delivers the slice of the batch it merged (queries [rank * per, (rank + 1) * per)) and nothing else
        grp.set_option("kw_exchange_slices", 1)
        grp.set_option("kw_own_slice_only", 1)
        h = grp.keyword_search_batch(qs, K, k_stride=K)
        grp.set_option("kw_own_slice_only", 0)
        per = (len(qs) + world - 1) // world
        for i in range(rank * per, min(len(qs), (rank + 1) * per)):
            ref = H.oracle_keyword(orc, qs[i])
            m = int(h.n_hits[i])
            check("own slice %s q%d" % (cut_name, i), h.status[i] == 0 and m == ref.keys.size and np.array_equal(h.keys[i, :m], ref.keys) and
                  np.array_equal(h.scores[i, :m], ref.scores) and int(h.num_matched[i]) == int(ref.num_keyword_matches))
        # candidate combinations over the shards (tsgpu_group_keyword_search_candidates_batch): per-shard fold, merged Topster, GLOBAL query_index, union counts
        users = [[[1, 2], [1, 3], [79, 80], [1, 2]], [[78, 79], [3], [3, 4]], [[80], [77, 78]]]
        combos = [[T.KwQuery(c, sort=sort, topster_size=K, total_cost=int(j > 0)) for j, c in enumerate(cs)] for cs in users]
        ch, cqi, cfound = grp.keyword_search_candidates_batch(combos, k=K, k_stride=K)
        check("candidates status " + cut_name, (ch.status == 0).all())
        for u, cs in enumerate(combos):
            ref, rqi = H.oracle_candidates(orc, cs, ids_cap=1 << 20)
            m = int(ch.n_hits[u])
            check("candidates %s u%d" % (cut_name, u), m == min(K, ref.keys.size) and np.array_equal(ch.keys[u, :m], ref.keys[:m]) and np.array_equal(ch.scores[u, :m], ref.scores[:m]) and
                  np.array_equal(cqi[u, :m], rqi[:m].astype(np.uint32)) and int(ch.num_matched[u]) == int(ref.num_keyword_matches) and int(cfound[u]) == int(ref.n_result_ids))
        if cut_name == "uneven":
            # a rank that fails BETWEEN the agreement and the sized exchange takes every rank out with it, in the same collective: nobody waits for it
            grp.set_option("kw_exchange_pruned", 2)
            grp.set_option("test_fail_prune_pack_rank", world)                 # the last rank fails
            try:
                grp.keyword_search_batch(qs, K, k_stride=K)
                check("injected failure " + cut_name, False)
            except B.TsgpuError as e:
                check("injected failure " + cut_name, ("injected" in str(e)) == (rank == world - 1) and (rank == world - 1 or "another rank failed" in str(e)))
            grp.set_option("test_fail_prune_pack_rank", 0)
            grp.set_option("kw_exchange_pruned", 1)
            check_keyword("after the injected failure " + cut_name, grp.keyword_search_batch(qs, K, k_stride=K))
        # facet counts over the ranks (tsgpu_group_facet_count_batch): two gathers (counts, entries), merged on every rank
        id_lists = [np.arange(n_docs, dtype=np.uint32), np.arange(1, n_docs, 3, dtype=np.uint32), np.array([], np.uint32)]
        for cap, sample_mod in ((512, 1), (5, 1), (512, 4)):
            got = grp.facet_count_batch(5, id_lists, cap=cap, sample_mod=sample_mod)
            for i, ids in enumerate(id_lists):
                rh, rc_, rd, rp, rn = orc.facet_count(5, ids, sample_mod=sample_mod, cap=cap)
                h, c, d, p, n = got[i]
                check("facets %s cap %d mod %d q%d" % (cut_name, cap, sample_mod, i), np.array_equal(h, rh) and np.array_equal(c, rc_) and np.array_equal(d, rd) and np.array_equal(p, rp) and
                      ((n == rn) if rn <= cap else (n > cap)))
        edges = np.linspace(int(pts.min()), int(pts.max()) + 1, 5).astype(np.int64)
        ranges = [(int(edges[r + 1]), int(edges[r])) for r in range(4) if edges[r + 1] > edges[r]]
        got = grp.facet_range_count_batch(5, 0, ranges, id_lists)
        st = grp.facet_stats_batch(5, B.FACET_INT32, id_lists)
        for i, ids in enumerate(id_lists):
            k, c, d, p, n = orc.facet_count_ex(5, ids, ranges=ranges, doc_vals=pts)
            m = {int(a): int(b) for a, b in zip(k.view(np.int64), c)}
            check("range facets %s q%d" % (cut_name, i), [m.get(int(up), 0) for up, _ in ranges] == got[i].tolist())
            mn, mx, sm, cnt = orc.facet_stats(5, ids, B.FACET_INT32)
            check("facet stats %s q%d" % (cut_name, i), st[i][:4] == (mn, mx, sm, cnt))
        # wildcard over the shards (tsgpu_group_wildcard_search_batch): every rank ranks the ids of its range
        fl = np.arange(2, n_docs, 7, dtype=np.uint32)
        wq = [T.KwQuery([], sort=((B.SORT_INT64_COLUMN, 1, 0), (B.SORT_SEQ_ID, 1, 0)), topster_size=K),
              T.KwQuery([], sort=((B.SORT_INT64_COLUMN, -1, 0), (B.SORT_SEQ_ID, -1, 0)), topster_size=K, filter_ids=fl),
              T.KwQuery([], sort=((B.SORT_SEQ_ID, 1, 0),), topster_size=K, excluded_ids=fl[::3]),
              T.KwQuery([], sort=((B.SORT_INT64_COLUMN, 1, 0), (B.SORT_SEQ_ID, 1, 0)), topster_size=K, filter_ids=fl[:2])]
        wh = grp.wildcard_search_batch(wq, k=K, k_stride=K)
        for i, q in enumerate(wq):
            ref = H.oracle_wildcard(orc, q)
            m = int(wh.n_hits[i])
            check("wildcard %s q%d" % (cut_name, i), wh.status[i] == 0 and m == min(K, ref.keys.size) and np.array_equal(wh.keys[i, :m], ref.keys[:m]) and
                  np.array_equal(wh.scores[i, :m], ref.scores[:m]) and int(wh.num_matched[i]) == int(ref.num_keyword_matches))
        # group_by over the ranks (tsgpu_group_keyword_search_grouped_batch): two rounds of gathers (heads, then the given groups' counts and KVs), merged on every rank
        from tests.test_emu_groupby import group_column, check_query, oracle_grouped, oracle_grouped_wildcard
        distinct, has_value = group_column(n_docs, seed=3)
        gqs = [T.KwQuery([1, 2], sort=sort, topster_size=250), T.KwQuery([3], sort=sort, topster_size=4), T.KwQuery([79, 80], sort=sort, topster_size=K),
               T.KwQuery([4, 9], sort=sort, topster_size=6, filter_ids=np.arange(0, n_docs, 2, dtype=np.uint32))]
        glim = [3, 2, 3, 1]
        for first_pass in (1, 0):
            grs = [(glim[i], 1, first_pass, 0, 0) for i in range(len(gqs))] + [(2, 1, first_pass, 0, 1)]
            try:
                h, gh = grp.keyword_search_grouped_batch(gqs + [wq[1]], grs, k_stride=750, g_stride=250, want_registers=bool(first_pass))
                for i, q in enumerate(gqs):
                    check_query(h, gh, i, oracle_grouped(orc, q, distinct, has_value, glim[i], first_pass), first_pass, glim[i], "group_by", check_total=False)
                check_query(h, gh, len(gqs), oracle_grouped_wildcard(wq[1], n_docs, pts, distinct, 2, first_pass), first_pass, 2, "group_by wildcard", check_total=False)
            except AssertionError as e:
                check("group_by %s pass %d: %s" % (cut_name, first_pass, str(e)[:300]), False)
            # ... and over candidate combinations: the ranks' own folds, query_index from the OR of the ranks' pass masks
            gusers = [[[1, 2], [1, 3], [79, 80], [1, 2]], [[78, 79], [3], [3, 4]], [[80], [77, 78]]]
            gcombos = [[T.KwQuery(c, sort=sort, topster_size=K, total_cost=int(j > 0)) for j, c in enumerate(cs)] for cs in gusers]
            try:
                h, gh, gqi = grp.keyword_search_grouped_candidates_batch(gcombos, [(3, 1, first_pass, 0, 0)] * len(gusers), k_stride=750, g_stride=250, want_registers=bool(first_pass))
                for u, cs in enumerate(gcombos):
                    ref, rqi = orc.search_candidates_grouped([H.oracle_query(orc, q) for q in cs], distinct, 3, first_pass, has_value=has_value, ids_cap=1 << 20)
                    check_query(h, gh, u, ref, first_pass, 3, "grouped candidates", check_total=False)
                    ng = int(gh.n_groups[u])
                    if first_pass:
                        assert {int(h.keys[u, r]): int(gqi[u, r]) for r in range(ng)} == {int(k): int(q) for k, q in zip(ref.keys, rqi)}, "query_index"
                    else:
                        for r in range(ng):
                            n = int(ref.group_size[r])
                            assert np.array_equal(gqi[u, r * 3:r * 3 + n], rqi[ref.begin[r]:ref.begin[r + 1]].astype(np.uint32)), "query_index"
            except AssertionError as e:
                check("grouped candidates %s pass %d: %s" % (cut_name, first_pass, str(e)[:300]), False)
        dm, lm, cm = grp.vec_knn_batch(1, Q, k_vec)
        check_knn("knn " + cut_name, dm, lm, cm)
        allow = np.arange(3, n_docs, 5, dtype=np.uint32)
        dm, lm, cm = grp.vec_knn_batch(1, Q, k_vec, allow_ids=allow)
        check_knn("knn allow " + cut_name, dm, lm, cm, allow=allow)
        fused = grp.hybrid_search_batch(qs, 1, B.METRIC_IP, Q, k=k_vec, fetch_size=10, alpha=0.3, k_stride=K)
        check("hybrid status", (fused.status == 0).all())
        for i, q in enumerate(qs):
            kw = {}
            if q.filter_ids is not None:
                kw["filter_ids"] = q.filter_ids
            if q.excluded_ids is not None:
                kw["excluded_ids"] = q.excluded_ids
            ref = orc.search_hybrid(orc.make_query(q.tokens, sort=OSORT, fetch_size=10, topster_size=K, **kw), Q[i], k=k_vec, alpha=0.3)
            m = int(fused.n_hits[i])
            check("hybrid %s q%d" % (cut_name, i), m == ref.keys.size and np.array_equal(fused.keys[i, :m], ref.keys) and np.array_equal(fused.scores[i, :m], ref.scores) and
                  np.array_equal(fused.vector_distance[i, :m].view(np.uint32), ref.vector_distance.view(np.uint32)))
        # rerank_hybrid_matches over the ranks: the rank whose shard owns a one-sided hit supplies its missing score (gathered), every rank re-fuses
        fused = grp.hybrid_search_batch(qs, 1, B.METRIC_IP, Q, k=k_vec, fetch_size=10, alpha=0.3, k_stride=K, rerank=True)
        check("hybrid rerank status", (fused.status == 0).all())
        for i, q in enumerate(qs):
            kw = {}
            if q.filter_ids is not None:
                kw["filter_ids"] = q.filter_ids
            if q.excluded_ids is not None:
                kw["excluded_ids"] = q.excluded_ids
            ref = orc.search_hybrid(orc.make_query(q.tokens, sort=OSORT, fetch_size=10, topster_size=K, **kw), Q[i], k=k_vec, alpha=0.3, rerank=True)
            m = int(fused.n_hits[i])
            check("hybrid rerank %s q%d" % (cut_name, i), m == ref.keys.size and np.array_equal(fused.keys[i, :m], ref.keys) and np.array_equal(fused.scores[i, :m], ref.scores) and
                  np.array_equal(fused.text_match[i, :m], ref.text_match) and np.array_equal(fused.vector_distance[i, :m].view(np.uint32), ref.vector_distance.view(np.uint32)))
        if cut_name == "uneven":
            # agreement before the collectives: ONE rank hands a bad k -> every rank returns the error, nobody hangs in a collective
            try:
                grp.keyword_search_batch(qs, K + 1 if rank == world - 1 else K, k_stride=K)
                check("bad k on one rank must fail everywhere", False)
            except T.TsgpuError as e:
                check("bad k: code 400 on every rank", e.code == 400)
                check("bad k: the other ranks learn who failed", rank == world - 1 or "rank %d failed" % (world - 1) in str(e))
            # ... and ranks called with DIFFERENT (individually valid) arguments fail together with 400
            try:
                grp.vec_knn_batch(1, Q, k_vec - 1 if rank == 0 else k_vec)
                check("differing k must fail everywhere", False)
            except T.TsgpuError as e:
                check("differing arguments: 400", e.code == 400 and "different arguments" in str(e))
            # ... and so does a rank that is called with an EMPTY batch while the others bring queries (it used to return before the agreement
            # step and leave the others waiting in it); everybody empty is fine
            try:
                grp.keyword_search_batch([] if rank == 0 else qs, K, k_stride=K)
                check("an empty batch on one rank must fail everywhere", False)
            except T.TsgpuError as e:
                check("empty batch on one rank: 400", e.code == 400 and "different arguments" in str(e))
            grp.keyword_search_batch([], K, k_stride=K)
            check_keyword("keyword after the failed calls", grp.keyword_search_batch(qs, K, k_stride=K))
        grp.close()
        g.close()

    # replicas form: every rank mirrors the WHOLE collection, the batch is cut into query slices, no merge
    g = T.GpuIndex(0, lib)
    load_shard(g, orc, 0, n_docs, pts, n_docs, X, dim)
    grp = T.GpuGroup.join_host(g, rank, world, ag, a2a)
    grp.set_option("replicas", 1)
    check_keyword("keyword replicas", grp.keyword_search_batch(qs, K, k_stride=K))
    wq = [T.KwQuery([], sort=((B.SORT_INT64_COLUMN, -1, 0), (B.SORT_SEQ_ID, -1, 0)), topster_size=K, filter_ids=np.arange(1, n_docs, 5, dtype=np.uint32))]
    wh = grp.wildcard_search_batch(wq, k=K, k_stride=K)
    ref = H.oracle_wildcard(orc, wq[0])
    m = int(wh.n_hits[0])
    check("wildcard replicas", wh.status[0] == 0 and m == min(K, ref.keys.size) and np.array_equal(wh.keys[0, :m], ref.keys[:m]) and np.array_equal(wh.scores[0, :m], ref.scores[:m]))
    dm, lm, cm = grp.vec_knn_batch(1, Q, k_vec)
    check_knn("knn replicas", dm, lm, cm)
    grp.close()
    g.close()

    for line in log:
        print(line, flush=True)
    print("rank %d: %s" % (rank, "DIST_OK" if ok else "DIST_MISMATCH"), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
