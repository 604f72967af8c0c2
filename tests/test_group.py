"""tsgpu_group: the doc-range shard exchange behind the C-ABI (include/tsgpu.h "multi-GPU group"; SURVEY §8e, BASELINE config 5).
CPU tier: G members on the SIMT emulator, TSGPU_XCHG_COPY transport (a memcpy "all-gather"): keyword, k-NN and hybrid results of
the group equal the UNSHARDED oracle bit for bit, for G = 2 and 3, uneven shards, per-query Topster capacities, filters.
GPU tier (one MI355X): the same with members sharing the device at 2M docs, and the RCCL transport in its one-rank form
(ncclCommInitRank / ncclAllGather / merge on a real stream; more ranks need more GPUs: the driver's scaling run)."""
import numpy as np
import pytest

import typesense_amd as T
from typesense_amd import _lib as B
from oracle import oracle_py as O
from tests import helpers as H
from tests.test_emu_groupby import group_column, check_query, oracle_grouped, oracle_grouped_wildcard

GROUP_COL = 1
SORT = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
OSORT = ((O.SORT_TEXT_MATCH, 0, 1), (O.SORT_INT64_COLUMN, 0, 1))


def build_group(lib, cuts, n_docs, dim, transport, seed=8):
    docs = H.zipf_docs(n_docs, 80, 10, seed=seed)
    pts = H.points_of(n_docs)
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n_docs, dim)).astype(np.float32)
    orc = O.OracleIndex(1, 1)
    for d in range(n_docs):
        orc.index_plain(d, 0, docs[d])
    orc.set_num_docs(n_docs)
    orc.set_sort_dense(0, pts)
    orc.vec_init(dim, O.METRIC_IP)
    orc.vec_add(np.arange(n_docs, dtype=np.uint32), X)
    fptr, fhash = H.facet_csr_of(n_docs)
    orc.facet_set(5, fptr, fhash)
    members = []
    distinct, _ = group_column(n_docs, seed=3)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        g = T.GpuIndex(0, lib)
        H.load_shard(g, orc, lo, hi, n_docs, pts)
        g.column_set(GROUP_COL, distinct.view(np.int64))                             # Index::get_distinct_id per document (the group_by column; global seq_ids)
        g.facet_set(5, *H.facet_csr_shard(fptr, fhash, lo, hi))                      # the facet hash index of the shard's own documents
        g.set_option("doc_range_lo", lo); g.set_option("doc_range_hi", hi)          # the seq_ids this shard OWNS (q = * ranks only those)
        g.vec_create(1, dim, B.METRIC_IP)
        if hi > lo:
            g.vec_upsert(1, np.arange(lo, hi, dtype=np.uint64), X[lo:hi])
        members.append(g)
    return orc, members, T.GpuGroup(members, transport), rng


def check_group(orc, grp, rng, n_docs, dim):
    # ---- keyword: global Topster order, num_matched = sum, per-query Topster capacity, a query no shard can serve (501) ----
    filt = np.sort(rng.choice(n_docs, size=n_docs // 3, replace=False)).astype(np.uint32)
    qs = [T.KwQuery([1, 2], sort=SORT, topster_size=40), T.KwQuery([3, 1, 2], sort=SORT, topster_size=40), T.KwQuery([5], sort=SORT, topster_size=12),
          T.KwQuery([4, 9], sort=SORT, topster_size=40, filter_ids=filt), T.KwQuery([1], sort=SORT, topster_size=0), T.KwQuery([77, 78], sort=SORT, topster_size=40),
          T.KwQuery([2, 3], sort=SORT, topster_size=40, excluded_ids=filt[::2]),
          # the rarest terms: a small shard holds no posting of some of them. The reference DROPS a token that matches no field (get_field_token_its,
          # src/index.cpp:5651-5655) — of the whole collection: a token missing on one shard and present on another is an EMPTY list there, not a dropped token
          # (round 6: before, such a shard answered the query without the token and contributed documents the collection's answer does not hold)
          T.KwQuery([79, 78], sort=SORT, topster_size=40), T.KwQuery([80, 1], sort=SORT, topster_size=40), T.KwQuery([76, 77, 2], sort=SORT, topster_size=40),
          T.KwQuery([9999, 80], sort=SORT, topster_size=40),                                    # 9999 exists NOWHERE: dropped on every shard, the query is [80]
          T.KwQuery([1, 2], sort=((B.SORT_INT64_COLUMN, 1, 77),), topster_size=40)]           # unknown sort column -> 501 on every shard
    hits = grp.keyword_search_batch(qs, k=250, k_stride=250)
    for i, q in enumerate(qs[:-1]):
        assert hits.status[i] == 0
        H.assert_hits_equal(hits, i, H.oracle_keyword(orc, q), "group keyword")
    assert hits.status[len(qs) - 1] == B.ERR_UNSUPPORTED and hits.n_hits[len(qs) - 1] == 0
    # a smaller exchange (k = 10 of the Topster's 40): the global top-10
    h10 = grp.keyword_search_batch(qs[:3], k=10, k_stride=16)
    for i, q in enumerate(qs[:3]):
        ref = H.oracle_keyword(orc, q)
        n = min(10, ref.keys.size)
        assert int(h10.n_hits[i]) == n and np.array_equal(h10.keys[i, :n], ref.keys[:n]) and np.array_equal(h10.scores[i, :n], ref.scores[:n])
        assert int(h10.num_matched[i]) == int(ref.num_keyword_matches)
    # ---- candidate combinations (Index::search_all_candidates over the shards): per-shard fold, merged Topster, GLOBAL query_index, union counts ----
    users = [[[1, 2], [1, 3], [2, 3], [1, 2], [77, 78]],          # a repeated combination (the later pass wins ties), a pass without matches anywhere
             [[79, 80], [3], [4], [3, 4]],                        # the FIRST pass matches nothing: every later pass's query_index starts at 0
             [[2, 1, 3]],
             [[77], [78, 79]],                                     # no pass matches anything
             [[60], [61], [62], [63], [64], [65], [5], [6]]]       # rare terms: passes that match on SOME shards only (the masks differ between shards)
    tsz = [40, 7, 40, 40, 250]
    combos = [[T.KwQuery(c, sort=SORT, topster_size=tsz[u], total_cost=int(j > 0)) for j, c in enumerate(cs)] for u, cs in enumerate(users)]
    for kk in (250, 10):
        ch, cqi, cfound = grp.keyword_search_candidates_batch(combos, k=kk, k_stride=250)
        assert (ch.status == 0).all()
        for u, cs in enumerate(combos):
            ref, rqi = H.oracle_candidates(orc, cs, ids_cap=1 << 20)
            n = min(kk, ref.keys.size)
            assert int(ch.n_hits[u]) == n, (kk, u, int(ch.n_hits[u]), n)
            assert np.array_equal(ch.keys[u, :n], ref.keys[:n]) and np.array_equal(ch.scores[u, :n], ref.scores[:n]) and np.array_equal(ch.text_match[u, :n], ref.text_match[:n]), (kk, u)
            assert np.array_equal(cqi[u, :n], rqi[:n].astype(np.uint32)), (kk, u, cqi[u, :n][:12], rqi[:12])
            assert int(ch.num_matched[u]) == int(ref.num_keyword_matches) and int(cfound[u]) == int(ref.n_result_ids), (kk, u)
    # ---- wildcard (Index::search_wildcard over the shards): every member ranks the ids it owns; filter / excluded ids, ascending and descending keys ----
    wq = [T.KwQuery([], sort=((B.SORT_INT64_COLUMN, 1, 0), (B.SORT_SEQ_ID, 1, 0)), topster_size=40),
          T.KwQuery([], sort=((B.SORT_INT64_COLUMN, -1, 0), (B.SORT_SEQ_ID, -1, 0)), topster_size=25, filter_ids=filt),
          T.KwQuery([], sort=((B.SORT_SEQ_ID, 1, 0),), topster_size=40, excluded_ids=filt[::2]),
          T.KwQuery([], sort=((B.SORT_INT64_COLUMN, 1, 0), (B.SORT_SEQ_ID, 1, 0)), topster_size=40, filter_ids=filt[:3])]       # three documents: most shards own none of them
    wh = grp.wildcard_search_batch(wq, k=250, k_stride=250)
    assert (wh.status == 0).all()
    for i, q in enumerate(wq):
        H.assert_hits_equal(wh, i, H.oracle_wildcard(orc, q), "group wildcard")
    check_group_by(orc, grp, rng, n_docs, filt)
    # ---- facet counts over the shards (do_facets' hash-index branch): counts add up, doc_id / array_pos of the greatest document; a small cap (truncated lists
    #      merge exactly for the first cap values), a facet query's allowed hashes, estimate_facets' sampling of the WHOLE list ----
    id_lists = [np.arange(n_docs, dtype=np.uint32), filt, filt[::5], np.array([], np.uint32), np.array([3, n_docs - 1], np.uint32)]
    allowed = np.unique(orc.facet_count(5, np.arange(n_docs, dtype=np.uint32))[0])[::2]
    for cap, sample_mod, allow in ((1024, 1, None), (7, 1, None), (1024, 3, None), (1024, 1, allowed)):
        got = grp.facet_count_batch(5, id_lists, cap=cap, sample_mod=sample_mod, allowed_hashes=allow)
        for i, ids in enumerate(id_lists):
            rh, rc_, rd, rp, rn = orc.facet_count(5, ids, sample_mod=sample_mod, allowed_hashes=allow, cap=cap)
            h, c, d, p, n = got[i]
            assert np.array_equal(h, rh) and np.array_equal(c, rc_) and np.array_equal(d, rd) and np.array_equal(p, rp), ("facets", cap, sample_mod, i)
            assert (n == rn) if rn <= cap else (n > cap), ("facet n_values", cap, sample_mod, i, n, rn)
    # ---- range facets and facet stats over the shards: the counts add up; min / max / count / sum merge ----
    pts = H.points_of(n_docs)
    lo_v, hi_v = int(pts.min()), int(pts.max()) + 1
    edges = np.linspace(lo_v, hi_v, 6).astype(np.int64)
    ranges = [(int(edges[r + 1]), int(edges[r])) for r in range(5) if edges[r + 1] > edges[r]]
    for sample_mod in (1, 3):
        got = grp.facet_range_count_batch(5, 0, ranges, id_lists, sample_mod=sample_mod)
        for i, ids in enumerate(id_lists):
            k, c, d, p, n = orc.facet_count_ex(5, ids, ranges=ranges, doc_vals=pts, sample_mod=sample_mod)
            m = {int(a): int(b) for a, b in zip(k.view(np.int64), c)}
            assert [m.get(int(up), 0) for up, _ in ranges] == got[i].tolist(), ("range facets", sample_mod, i)
        st = grp.facet_stats_batch(5, B.FACET_INT32, id_lists, sample_mod=sample_mod)
        for i, ids in enumerate(id_lists):
            mn, mx, sm, cnt = orc.facet_stats(5, ids, B.FACET_INT32, sample_mod=sample_mod)
            assert st[i][:4] == (mn, mx, sm, cnt) and st[i][4] == 1, ("facet stats", sample_mod, i, st[i], (mn, mx, sm, cnt))
    # ---- k-NN: closest first, ties -> smaller label; allow list ----
    Q = rng.standard_normal((5, dim)).astype(np.float32)
    for k in (7, 30):
        dist, lab, cnt = grp.vec_knn_batch(1, Q, k)
        for i in range(Q.shape[0]):
            d, l = orc.flat_knn(Q[i], k)
            assert cnt[i] == d.size and np.array_equal(lab[i, :d.size].astype(np.uint32), l)
            assert np.array_equal(dist[i, :d.size].view(np.uint32), d.view(np.uint32))
    allow = filt[:50]
    dist, lab, cnt = grp.vec_knn_batch(1, Q[:2], 9, allow_ids=allow)
    for i in range(2):
        d, l = orc.flat_knn(Q[i], 9, allow_ids=allow)
        assert cnt[i] == d.size and np.array_equal(lab[i, :d.size].astype(np.uint32), l) and np.array_equal(dist[i, :d.size].view(np.uint32), d.view(np.uint32))
    # ---- hybrid: fused AFTER the merge (global ranks), incl. a filtered and a curated query ----
    hq = [T.KwQuery([1, 2], sort=SORT, topster_size=0), T.KwQuery([3], sort=SORT, topster_size=0), T.KwQuery([2, 5], sort=SORT, topster_size=0, filter_ids=filt),
          T.KwQuery([59, 1], sort=SORT, topster_size=0, excluded_ids=filt[::3]), T.KwQuery([7, 7], sort=SORT, topster_size=0)]
    fused = grp.hybrid_search_batch(hq, 1, B.METRIC_IP, Q, k=0, fetch_size=10, alpha=0.3, k_stride=250)
    assert (fused.status == 0).all()
    for i, q in enumerate(hq):
        kw = {}
        if q.filter_ids is not None:
            kw["filter_ids"] = q.filter_ids
        if q.excluded_ids is not None:
            kw["excluded_ids"] = q.excluded_ids
        ref = orc.search_hybrid(orc.make_query(q.tokens, sort=OSORT, fetch_size=10, **kw), Q[i], k=0, alpha=0.3)
        n = int(fused.n_hits[i])
        assert n == ref.keys.size, (i, n, ref.keys.size)
        assert np.array_equal(fused.keys[i, :n], ref.keys) and np.array_equal(fused.scores[i, :n], ref.scores), i
        assert np.array_equal(fused.text_match[i, :n], ref.text_match), i
        assert np.array_equal(fused.vector_distance[i, :n].view(np.uint32), ref.vector_distance.view(np.uint32)), i
    # rerank_hybrid_matches over the shards (Index::compute_aux_scores): the shard that owns a one-sided hit supplies its missing score, every rank re-fuses
    fused = grp.hybrid_search_batch(hq, 1, B.METRIC_IP, Q, k=0, fetch_size=10, alpha=0.3, k_stride=250, rerank=True)
    assert (fused.status == 0).all()
    one_sided = 0
    for i, q in enumerate(hq):
        kw = {}
        if q.filter_ids is not None:
            kw["filter_ids"] = q.filter_ids
        if q.excluded_ids is not None:
            kw["excluded_ids"] = q.excluded_ids
        ref = orc.search_hybrid(orc.make_query(q.tokens, sort=OSORT, fetch_size=10, **kw), Q[i], k=0, alpha=0.3, rerank=True)
        n = int(fused.n_hits[i])
        assert n == ref.keys.size, (i, n, ref.keys.size)
        assert np.array_equal(fused.keys[i, :n], ref.keys) and np.array_equal(fused.scores[i, :n], ref.scores), ("rerank", i)
        assert np.array_equal(fused.text_match[i, :n], ref.text_match), ("rerank", i)
        assert np.array_equal(fused.vector_distance[i, :n].view(np.uint32), ref.vector_distance.view(np.uint32)), ("rerank", i)
        one_sided += int((ref.text_match == 0).sum())


def check_group_by(orc, grp, rng, n_docs, filt):
    """group_by over the shards (tsgpu_group_keyword_search_grouped_batch): both passes, keyword and q = *, against the UNSHARDED oracle's distinct Topster.
    Small Topsters: a group the collection selects is missing from the best groups of most shards (its documents there still count and compete for its KVs)."""
    distinct, has_value = group_column(n_docs, seed=3)
    pts = H.points_of(n_docs)
    wsort = ((B.SORT_INT64_COLUMN, 1, 0), (B.SORT_SEQ_ID, 1, 0))
    qs = [T.KwQuery([1, 2], sort=SORT, topster_size=250), T.KwQuery([3], sort=SORT, topster_size=5), T.KwQuery([2], sort=SORT, topster_size=3),
          T.KwQuery([4, 9], sort=SORT, topster_size=250, filter_ids=filt), T.KwQuery([2, 3], sort=SORT, topster_size=7, excluded_ids=filt[::2]),
          T.KwQuery([79, 78], sort=SORT, topster_size=40), T.KwQuery([80, 1], sort=SORT, topster_size=40),        # rare terms: an empty list on some shards
          T.KwQuery([77, 78, 79], sort=SORT, topster_size=40),                                                    # matches nothing anywhere
          T.KwQuery([1], sort=SORT, topster_size=0)]
    limits = [3, 1, 2, 3, 3, 2, 3, 3, 5]
    wq = [T.KwQuery([], sort=wsort, topster_size=40), T.KwQuery([], sort=((B.SORT_INT64_COLUMN, -1, 0), (B.SORT_SEQ_ID, -1, 0)), topster_size=6, filter_ids=filt),
          T.KwQuery([], sort=wsort, topster_size=4, excluded_ids=filt[::2]), T.KwQuery([], sort=wsort, topster_size=40, filter_ids=filt[:3])]
    wlimits = [3, 2, 4, 3]
    for first_pass in (1, 0):
        for gmv in (0, 1):
            groups = [(limits[i], GROUP_COL, first_pass, gmv, 0) for i in range(len(qs))] + [(wlimits[i], GROUP_COL, first_pass, gmv, 1) for i in range(len(wq))]
            h, gh = grp.keyword_search_grouped_batch(qs + wq, groups, k_stride=250 * 5, g_stride=250, want_registers=bool(first_pass))
            for i, q in enumerate(qs):
                ref = oracle_grouped(orc, q, distinct, has_value, limits[i], first_pass, gmv=bool(gmv))
                check_query(h, gh, i, ref, first_pass, limits[i], "group group_by kw pass%d gmv%d" % (first_pass, gmv), check_total=False)
            for j, q in enumerate(wq):
                ref = oracle_grouped_wildcard(q, n_docs, pts, distinct, wlimits[j], first_pass, gmv=bool(gmv))
                check_query(h, gh, len(qs) + j, ref, first_pass, wlimits[j], "group group_by wildcard pass%d gmv%d" % (first_pass, gmv), check_total=False)
    # ---- the same over candidate combinations (tsgpu_group_keyword_search_grouped_candidates_batch): the shards' own folds, GLOBAL query_index from the shards' pass masks ----
    users = [[[1, 2], [1, 3], [2, 3], [1, 2], [77, 78]],          # a repeated combination (the later pass wins ties), a pass that matches nothing anywhere
             [[79, 80], [3], [4], [3, 4]],                        # the FIRST pass matches nothing anywhere: the later passes' query_index starts at 0
             [[2, 1, 3]],
             [[77], [78, 79]],
             [[60], [61], [62], [63], [64], [65], [5], [6]]]       # rare terms: passes that match on SOME shards only (the masks differ between shards)
    tsz = [40, 5, 40, 40, 250]
    combos = [[T.KwQuery(c, sort=SORT, topster_size=tsz[u], total_cost=int(j > 0)) for j, c in enumerate(cs)] for u, cs in enumerate(users)]
    for first_pass in (1, 0):
        h, gh, qidx = grp.keyword_search_grouped_candidates_batch(combos, [(3, GROUP_COL, first_pass, 0, 0)] * len(users), k_stride=750, g_stride=250, want_registers=bool(first_pass))
        assert (h.status == 0).all()
        for u, cs in enumerate(combos):
            ref, rqi = orc.search_candidates_grouped([H.oracle_query(orc, q) for q in cs], distinct, 3, first_pass, has_value=has_value, ids_cap=1 << 20)
            check_query(h, gh, u, ref, first_pass, 3, "group grouped candidates u%d pass%d" % (u, first_pass), check_total=False)
            ng = int(gh.n_groups[u])
            if first_pass:        # (the reference's heap order is not the library's: compare query_index per key)
                assert {int(h.keys[u, r]): int(qidx[u, r]) for r in range(ng)} == {int(k): int(q) for k, q in zip(ref.keys, rqi)}, u
            else:
                for r in range(ng):
                    n = int(ref.group_size[r])
                    assert np.array_equal(qidx[u, r * 3:r * 3 + n], rqi[ref.begin[r]:ref.begin[r + 1]].astype(np.uint32)), (u, r)
    # a bad query next to good ones (unknown group column -> 404 on every shard), strides too small for the capacity (400), groups_total asked for across shards (501)
    h, gh = grp.keyword_search_grouped_batch([qs[0], qs[1], qs[0]], [(3, GROUP_COL, 0, 0, 0), (1, 77, 0, 0, 0), (3, GROUP_COL, 0, 0, 0)], k_stride=300, g_stride=250)
    assert h.status.tolist() == [B.ERR_INVALID, B.ERR_NOT_FOUND, B.ERR_INVALID] and h.n_hits.sum() == 0
    h, gh = grp.keyword_search_grouped_batch([qs[1], qs[1]], [(1, GROUP_COL, 0, 0, 0), (1, 77, 1, 0, 0)], k_stride=300, g_stride=250)
    assert h.status.tolist() == [0, B.ERR_NOT_FOUND]
    check_query(h, gh, 0, oracle_grouped(orc, qs[1], distinct, has_value, 1, 0), 0, 1, "next to a bad query", check_total=False)
    # a batch whose only query matches nothing anywhere (no groups: the second round still runs, with nothing given)
    for first_pass in (1, 0):
        h, gh = grp.keyword_search_grouped_batch([qs[7]], [(3, GROUP_COL, first_pass, 0, 0)], k_stride=750, g_stride=250)
        assert int(h.status[0]) == 0 and int(gh.n_groups[0]) == 0 and int(h.n_hits[0]) == 0 and int(h.num_matched[0]) == 0
    if grp.size() > 1 and not getattr(grp, "replicas_form", False):
        with pytest.raises(T.TsgpuError):
            grp.keyword_search_grouped_batch(qs[:1], [(3, GROUP_COL, 1, 0, 0)], k_stride=750, g_stride=250, want_totals=True)


def device_output_equals_host_output(grp, n_docs):
    """`out` in device memory (of member 0): the merged slices are replicated there; same content as the host delivery"""
    qs = [T.KwQuery(t, sort=SORT, topster_size=40) for t in ([1, 2], [3], [2, 5], [4, 9, 1], [6])]
    host = grp.keyword_search_batch(qs, k=30, k_stride=32)
    dev = T.Hits(len(qs), 32)
    hs = dev.c_struct()
    hs.mem = B.MEM_DEVICE                      # (the test tiers' "device" arrays: numpy memory on the emulator, hipMalloc'd memory below on the GPU)
    import ctypes as C
    L = grp.L
    if "emu" not in grp.members[0].lib_path:
        import torch
        t = {name: torch.zeros(getattr(dev, name).shape, dtype=getattr(torch, str(getattr(dev, name).dtype).replace("uint64", "int64").replace("uint32", "int32")), device="cuda")
             for name in ("keys", "scores", "text_match", "n_hits", "num_matched", "status")}
        for name, v in t.items():
            setattr(hs, name, v.data_ptr())
        hs.vector_distance = hs.match_score_index = hs.search_cutoff = None
        torch.cuda.synchronize()
    arr = T.index.make_query_array(qs)
    B.check(L, L.tsgpu_group_keyword_search_batch(grp.h, C.cast(arr, C.c_void_p), len(qs), 30, C.byref(hs)))
    if "emu" not in grp.members[0].lib_path:
        got = {name: v.cpu().numpy() for name, v in t.items()}
    else:
        got = {name: getattr(dev, name) for name in ("keys", "scores", "text_match", "n_hits", "num_matched", "status")}
    for i in range(len(qs)):
        n = int(host.n_hits[i])
        assert int(got["n_hits"][i]) == n and int(got["num_matched"][i]) == int(host.num_matched[i]) and int(got["status"][i]) == 0
        assert np.array_equal(got["keys"][i, :n].astype(np.uint64), host.keys[i, :n]) and np.array_equal(got["scores"][i, :n], host.scores[i, :n])
        assert np.array_equal(got["text_match"][i, :n], host.text_match[i, :n])


@pytest.mark.parametrize("cuts,slices", [((0, 700, 1500), 1), ((0, 100, 1100, 1500), 1), ((0, 1500, 1500), 1), ((0, 400, 900, 1500), 0)])
def test_group_over_copy_transport_equals_the_unsharded_oracle(cuts, slices):
    orc, members, grp, rng = build_group(H.emu_lib_path(), cuts, 1500, 24, B.XCHG_COPY)
    try:
        assert grp.size() == len(cuts) - 1
        grp.set_option("kw_exchange_slices", slices)
        check_group(orc, grp, rng, 1500, 24)
        t = grp.timings()
        assert t.exchange_bytes_per_member > 0
        device_output_equals_host_output(grp, 1500)
    finally:
        grp.close()
        for g in members:
            g.close()


def test_replicas_form_cuts_the_batch_into_query_slices():
    """option "replicas": every member mirrors the whole collection, member i answers the i-th slice of the batch; same results"""
    lib = H.emu_lib_path()
    n_docs, dim = 1200, 16
    orc, members, grp, rng = build_group(lib, (0, n_docs), n_docs, dim, B.XCHG_COPY)       # one full mirror ...
    grp.close()
    orc2, more, grp2, _ = build_group(lib, (0, n_docs), n_docs, dim, B.XCHG_COPY)          # ... and a second one
    grp2.close()
    mirrors = members + more
    grp = T.GpuGroup(mirrors + [mirrors[0]], B.XCHG_COPY)                                    # 3 replicas (two share a context: allowed)
    try:
        grp.set_option("replicas", 1)
        grp.replicas_form = True
        check_group(orc, grp, rng, n_docs, dim)
        device_output_equals_host_output(grp, n_docs)
    finally:
        grp.close()
        for g in mirrors:
            g.close()


def test_group_by_over_shards_with_big_groups_takes_the_chunked_path_in_the_given_groups_round():
    """a group_by field with a handful of values over many matches, cut into two doc ranges: in round 2 (the groups GIVEN) a shard's groups of more than 4096 members
    take gb_chunk_kernel, a singleton group lives on ONE shard (the other answers it empty), group_limit 50 merges two shards' lists of 50; q = * needs no postings"""
    n = 20000
    lib = H.emu_lib_path()
    rng = np.random.default_rng(3)
    points = H.points_of(n)
    u = rng.random(n)
    col = np.where(u < 0.7, 111, np.where(u < 0.99, 222, 333)).astype(np.uint64)
    col[rng.choice(n, 40, replace=False)] = np.arange(40, dtype=np.uint64) + np.uint64(10**12)       # singletons
    members = []
    for lo, hi in ((0, 9000), (9000, n)):
        g = T.GpuIndex(0, lib)
        g.field_create(0, False)
        g.set_num_docs(n)
        g.column_set(0, points)
        g.column_set(GROUP_COL, col.view(np.int64))
        g.commit()
        g.set_option("doc_range_lo", lo); g.set_option("doc_range_hi", hi)
        members.append(g)
    grp = T.GpuGroup(members, B.XCHG_COPY)
    try:
        excl = np.sort(rng.choice(n, 500, replace=False)).astype(np.uint32)
        filt = np.sort(rng.choice(n, 14000, replace=False)).astype(np.uint32)
        sort = ((B.SORT_INT64_COLUMN, 1, 0), (B.SORT_SEQ_ID, -1, 0))
        qs = [T.KwQuery([], sort=sort, topster_size=250), T.KwQuery([], sort=sort, topster_size=30, filter_ids=filt, excluded_ids=excl),
              T.KwQuery([], sort=((B.SORT_SEQ_ID, 1, 0),), topster_size=2)]
        for limit, passes in ((3, (False, True)), (50, (False,))):
            for first_pass in passes:
                h, gh = grp.keyword_search_grouped_batch(qs, [(limit, GROUP_COL, int(first_pass), 0, 1)] * len(qs), k_stride=250 * limit, g_stride=250, want_registers=True)
                for i, q in enumerate(qs):
                    check_query(h, gh, i, oracle_grouped_wildcard(q, n, points, col, limit, first_pass), first_pass, limit, "sharded big groups limit %d" % limit, check_total=False)
        assert int(gh.group_found[0, :int(gh.n_groups[0])].max()) > 8192          # (a group of two chunks on the larger shard)
    finally:
        grp.close()
        for g in members:
            g.close()


def test_group_argument_errors_are_reported_not_crashed():
    lib = H.emu_lib_path()
    g = T.GpuIndex(0, lib)
    g2 = T.GpuIndex(0, lib)
    two = T.GpuGroup([g, g2], B.XCHG_COPY)
    with pytest.raises(T.TsgpuError):             # q = * grouped over shards without doc ranges: every shard would group every document
        two.keyword_search_grouped_batch([T.KwQuery([], sort=((B.SORT_SEQ_ID, 1, 0),), topster_size=5)], [(3, 0, 1, 0, 1)], k_stride=16, g_stride=8)
    two.close()
    g2.close()
    with pytest.raises(T.TsgpuError):
        T.GpuGroup([g, g], B.XCHG_RCCL)           # RCCL needs one GPU per member
    grp = T.GpuGroup([g], B.XCHG_COPY)
    with pytest.raises(T.TsgpuError):
        grp.keyword_search_batch([T.KwQuery([1], sort=SORT)], k=2000, k_stride=2000)
    grp.close()
    g.close()


@pytest.mark.gpu
def test_group_on_one_mi355x_members_share_the_device():
    orc, members, grp, rng = build_group(H.gpu_lib_path(), (0, 9000, 20000, 30000), 30000, 64, B.XCHG_COPY, seed=5)
    try:
        check_group(orc, grp, rng, 30000, 64)
        device_output_equals_host_output(grp, 30000)
        grp.set_option("kw_exchange_slices", 0)
        check_group(orc, grp, rng, 30000, 64)
    finally:
        grp.close()
        for g in members:
            g.close()


@pytest.mark.gpu
def test_rccl_transport_one_rank_form_runs_the_real_collective():
    """ncclGetUniqueId -> ncclCommInitRank(1 rank) -> ncclAllGather on the context's stream -> merge: equals the plain call"""
    lib = H.gpu_lib_path()
    docs = H.zipf_docs(20000, 200, 10, seed=3)
    orc, g = H.build_pair(docs, lib)
    rng = np.random.default_rng(1)
    X = rng.standard_normal((20000, 48)).astype(np.float32)
    g.vec_create(1, 48, B.METRIC_IP)
    g.vec_upsert(1, np.arange(20000, dtype=np.uint64), X)
    uid = T.GpuGroup.unique_id(g.L)
    grp = T.GpuGroup.join(g, uid, 0, 1)
    try:
        qs = [T.KwQuery(t, sort=SORT, topster_size=250) for t in ([1, 2], [3, 4, 5], [9], [150, 2])]
        plain = g.keyword_search_batch(qs, k_stride=250)
        for mode, pruned in ((1, 1), (2, 0), (2, 2), (0, 2), (0, 0)):
            # slices = 2 / pruned = 2: the forms forced on one rank — ncclAllToAll, in-place ncclAllGather, the bounds ncclAllGather, the totals
            # ncclAllGather and the grouped ncclSend / ncclRecv pairs of the bound-pruned exchange run for real
            grp.set_option("kw_exchange_slices", mode)
            grp.set_option("kw_exchange_pruned", pruned)
            got = grp.keyword_search_batch(qs, k=100, k_stride=100)
            for i in range(len(qs)):
                n = min(100, int(plain.n_hits[i]))
                assert int(got.n_hits[i]) == n and np.array_equal(got.keys[i, :n], plain.keys[i, :n]) and np.array_equal(got.scores[i, :n], plain.scores[i, :n]), (mode, pruned, i)
                assert int(got.num_matched[i]) == int(plain.num_matched[i])
        grp.set_option("kw_exchange_slices", 1)
        grp.set_option("kw_exchange_pruned", 1)
        device_output_equals_host_output(grp, 20000)      # rank form + device outputs: the replication collectives
        Q = rng.standard_normal((6, 48)).astype(np.float32)
        d0, l0, c0 = g.vec_knn_batch(1, Q, 20)
        d1, l1, c1 = grp.vec_knn_batch(1, Q, 20)
        assert np.array_equal(l0, l1) and np.array_equal(d0.view(np.uint32), d1.view(np.uint32)) and np.array_equal(c0, c1)
        # round 6's group calls through the same transport (their gathers are ncclAllGather on the member's stream): candidate combinations, q = *,
        # rerank_hybrid_matches, facet counts — each equals the plain call on the one context
        users = [[T.KwQuery(c, sort=SORT, topster_size=250, total_cost=int(j > 0)) for j, c in enumerate(cs)] for cs in ([[1, 2], [1, 3], [150, 2]], [[9], [3, 4]])]
        ph, pqi, pf = g.keyword_search_candidates_batch(users, k_stride=250)
        ch, cqi, cf = grp.keyword_search_candidates_batch(users, k=100, k_stride=100)
        for u in range(len(users)):
            n = min(100, int(ph.n_hits[u]))
            assert int(ch.n_hits[u]) == n and np.array_equal(ch.keys[u, :n], ph.keys[u, :n]) and np.array_equal(ch.scores[u, :n], ph.scores[u, :n]) and np.array_equal(cqi[u, :n], pqi[u, :n])
            assert int(cf[u]) == int(pf[u]) and int(ch.num_matched[u]) == int(ph.num_matched[u])
        wq = [T.KwQuery([], sort=((B.SORT_INT64_COLUMN, 1, 0), (B.SORT_SEQ_ID, 1, 0)), topster_size=250, filter_ids=np.arange(3, 20000, 7, dtype=np.uint32))]
        pw = g.wildcard_search_batch(wq, k_stride=250)
        gw = grp.wildcard_search_batch(wq, k=100, k_stride=100)
        assert int(gw.n_hits[0]) == 100 and np.array_equal(gw.keys[0, :100], pw.keys[0, :100]) and np.array_equal(gw.scores[0, :100], pw.scores[0, :100]) and int(gw.num_matched[0]) == int(pw.num_matched[0])
        hq = [T.KwQuery([1, 2], sort=SORT, topster_size=0), T.KwQuery([9], sort=SORT, topster_size=0)]
        pf_ = g.hybrid_search_batch(hq, 1, Q[:2], k=0, fetch_size=10, alpha=0.3, k_stride=250, rerank=True)
        gf = grp.hybrid_search_batch(hq, 1, B.METRIC_IP, Q[:2], k=0, fetch_size=10, alpha=0.3, k_stride=250, rerank=True)
        for i in range(2):
            n = int(pf_.n_hits[i])
            assert int(gf.n_hits[i]) == n and np.array_equal(gf.keys[i, :n], pf_.keys[i, :n]) and np.array_equal(gf.scores[i, :n], pf_.scores[i, :n]) and np.array_equal(gf.text_match[i, :n], pf_.text_match[i, :n])
        # group_by through the same transport (two rounds of ncclAllGather'd host blocks): equals the plain grouped call on the one context
        distinct, _ = group_column(20000, seed=3)
        g.column_set(GROUP_COL, distinct.view(np.int64))
        gq = [T.KwQuery([1, 2], sort=SORT, topster_size=250), T.KwQuery([9], sort=SORT, topster_size=5), wq[0]]
        for first_pass in (1, 0):
            grs = [(3, GROUP_COL, first_pass, 0, 0), (2, GROUP_COL, first_pass, 0, 0), (2, GROUP_COL, first_pass, 0, 1)]
            ph_, pg = g.keyword_search_grouped_batch(gq, grs, k_stride=750, g_stride=250, want_registers=bool(first_pass))
            sh_, sg = grp.keyword_search_grouped_batch(gq, grs, k_stride=750, g_stride=250, want_registers=bool(first_pass))
            for i in range(len(gq)):
                ng = int(pg.n_groups[i])
                assert int(sh_.status[i]) == 0 and int(sg.n_groups[i]) == ng and int(sh_.n_hits[i]) == int(ph_.n_hits[i]) and int(sh_.num_matched[i]) == int(ph_.num_matched[i]), (first_pass, i)
                assert np.array_equal(sg.distinct_key[i, :ng], pg.distinct_key[i, :ng]) and np.array_equal(sg.group_found[i, :ng], pg.group_found[i, :ng]) and np.array_equal(sg.group_size[i, :ng], pg.group_size[i, :ng])
                L = 1 if first_pass else grs[i][0]
                for r in range(ng):
                    n = int(pg.group_size[i, r])
                    assert np.array_equal(sh_.keys[i, r * L:r * L + n], ph_.keys[i, r * L:r * L + n]) and np.array_equal(sh_.scores[i, r * L:r * L + n], ph_.scores[i, r * L:r * L + n]), (first_pass, i, r)
                assert int(sg.groups_count[i]) == int(pg.groups_count[i]) and (not first_pass or np.array_equal(sg.loglog_registers[i], pg.loglog_registers[i]))
        g.facet_set(5, *H.facet_csr_of(20000))
        id_lists = [np.arange(20000, dtype=np.uint32), np.arange(1, 20000, 3, dtype=np.uint32)]
        for cap in (512, 6):
            a, b = g.facet_count_batch(5, id_lists, cap=cap), grp.facet_count_batch(5, id_lists, cap=cap)
            for x, y in zip(a, b):
                assert all(np.array_equal(x[j], y[j]) for j in range(4)) and ((x[4] == y[4]) if x[4] <= cap else (y[4] > cap))
        ranges = [(250, 0), (500, 250), (1000, 600)]
        assert np.array_equal(g.facet_range_count_batch(5, 0, ranges, id_lists), grp.facet_range_count_batch(5, 0, ranges, id_lists))
        assert g.facet_stats_batch(5, B.FACET_INT32, id_lists) == grp.facet_stats_batch(5, B.FACET_INT32, id_lists)
    finally:
        grp.close()
        g.close()


@pytest.mark.parametrize("slices", [1, 0])
def test_bound_pruned_exchange_equals_the_unpruned_one_and_the_oracle_under_skew(slices):
    """option kw_exchange_pruned (default 1): every shard sends only its entries at or above B[q] = the greatest of the shards' kq-th
    best entries (KV::is_greater order, /root/reference/include/topster.h:146-154). Adversarial data: the documents that win every query live in
    ONE shard (that shard supplies the bound, the others send nothing or almost nothing), thousands of documents TIE on (text_match, points) around
    the bound (the key decides), queries with fewer hits than the Topster holds (no bound: nothing may be pruned), per-query Topster capacities,
    a failing query, an empty shard. Both forms = the unsharded oracle bit for bit, and the pruned form moves fewer bytes."""
    n_docs, dim = 1500, 8
    docs = H.zipf_docs(n_docs, 60, 8, seed=21)
    pts = np.where(np.arange(n_docs) < 500, 1000 + (np.arange(n_docs) % 2), np.arange(n_docs) % 3).astype(np.int64)   # shard 0 wins; two- and three-way ties everywhere
    orc = O.OracleIndex(1, 1)
    for d in range(n_docs):
        orc.index_plain(d, 0, docs[d])
    orc.set_num_docs(n_docs)
    orc.set_sort_dense(0, pts)
    cuts = (0, 500, 1000, 1000, 1500)                        # four members, the third one empty
    members = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        g = T.GpuIndex(0, H.emu_lib_path())
        H.load_shard(g, orc, lo, hi, n_docs, pts)
        members.append(g)
    grp = T.GpuGroup(members, B.XCHG_COPY)
    try:
        grp.set_option("kw_exchange_slices", slices)
        filt = np.arange(0, n_docs, 3, dtype=np.uint32)
        qs = [T.KwQuery([1], sort=SORT, topster_size=0), T.KwQuery([1, 2], sort=SORT, topster_size=40), T.KwQuery([2], sort=SORT, topster_size=12),
              T.KwQuery([3, 1, 2], sort=SORT, topster_size=40), T.KwQuery([55, 56], sort=SORT, topster_size=40), T.KwQuery([4], sort=SORT, topster_size=40, filter_ids=filt),
              T.KwQuery([2, 3], sort=((B.SORT_INT64_COLUMN, -1, 0), (B.SORT_TEXT_MATCH, 1, 0)), topster_size=40),     # ascending points: the winners are NOT in shard 0
              T.KwQuery([1, 2], sort=((B.SORT_INT64_COLUMN, 1, 77),), topster_size=40), T.KwQuery([9999], sort=SORT, topster_size=40), T.KwQuery([5], sort=SORT, topster_size=1)]
        out = {}
        for pruned in (1, 0):
            grp.set_option("kw_exchange_pruned", pruned)
            for k in (250, 40, 7):
                out[(pruned, k)] = (grp.keyword_search_batch(qs, k=k, k_stride=250), grp.timings().hit_exchange_bytes_per_member)
        for k in (250, 40, 7):
            (hp, bytes_p), (hu, bytes_u) = out[(1, k)], out[(0, k)]
            for name in ("keys", "scores", "text_match", "n_hits", "num_matched", "status"):
                assert np.array_equal(getattr(hp, name), getattr(hu, name)), (name, k)
            assert k < 40 or bytes_p < bytes_u, (k, bytes_p, bytes_u)     # (at k = 7 the 32-byte bounds outweigh what they save)
            for i, q in enumerate(qs):
                if i == 7:
                    assert hp.status[i] == B.ERR_UNSUPPORTED and hp.n_hits[i] == 0
                    continue
                ref = H.oracle_keyword(orc, q)
                n = min(k, ref.keys.size)
                assert hp.status[i] == 0 and int(hp.n_hits[i]) == n, (i, k, int(hp.n_hits[i]), n)
                assert np.array_equal(hp.keys[i, :n], ref.keys[:n]) and np.array_equal(hp.scores[i, :n], ref.scores[:n]), (i, k)
                assert int(hp.num_matched[i]) == int(ref.num_keyword_matches)
        # the winners of the default sort live in shard 0: at k = 40 the other shards' entries all fall below its 40th entry. What crosses the wire
        # is the bounds (32 B per query and shard) + per slice the header pairs + the LARGEST (source, destination) total of entries
        assert int(out[(1, 250)][0].n_hits[0]) == 250 and out[(1, 40)][1] * 2 < out[(0, 40)][1], (out[(1, 40)][1], out[(0, 40)][1])
    finally:
        grp.close()
        for g in members:
            g.close()


RCCL_TWO_RANKS_ONE_DEVICE = r"""
import sys, threading, numpy as np
sys.path.insert(0, %(root)r)
import typesense_amd as T
from typesense_amd import _lib as B
from tests import helpers as H
from oracle import oracle_py as O
SORT = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
n_docs = 6000
docs = H.zipf_docs(n_docs, 100, 10, seed=5)
pts = H.points_of(n_docs)
orc = O.OracleIndex(1, 1)
for d in range(n_docs):
    orc.index_plain(d, 0, docs[d])
orc.set_num_docs(n_docs); orc.set_sort_dense(0, pts)
lib = H.gpu_lib_path()
members = []
for lo, hi in ((0, 2500), (2500, n_docs)):
    g = T.GpuIndex(0, lib)
    H.load_shard(g, orc, lo, hi, n_docs, pts)
    members.append(g)
uid = T.GpuGroup.unique_id(members[0].L)
grp, err = [None, None], [None, None]
def join(r):
    try:
        grp[r] = T.GpuGroup.join(members[r], uid, r, 2)
    except Exception as e:
        err[r] = repr(e)
th = [threading.Thread(target=join, args=(r,)) for r in (0, 1)]
[t.start() for t in th]; [t.join(60) for t in th]
if any(t.is_alive() for t in th):
    print("RCCL_REFUSED ncclCommInitRank with two ranks on one device did not return within 60 s"); sys.stdout.flush(); import os; os._exit(0)
if err[0] or err[1]:
    print("RCCL_REFUSED " + str(err[0] or err[1])); sys.exit(0)
qs = [T.KwQuery(t, sort=SORT, topster_size=250) for t in ([1, 2], [3, 4, 5], [9], [50, 2], [7])]
out = [None, None]
def run(r):
    res = {}
    for slices in (1, 0):
        for pruned in (1, 0):
            grp[r].set_option("kw_exchange_slices", slices); grp[r].set_option("kw_exchange_pruned", pruned)
            res[(slices, pruned)] = grp[r].keyword_search_batch(qs, k=100, k_stride=100)
    out[r] = res
th = [threading.Thread(target=run, args=(r,)) for r in (0, 1)]
[t.start() for t in th]; [t.join(120) for t in th]
if any(t.is_alive() for t in th):
    print("RCCL_HUNG the two-rank collectives did not complete"); sys.stdout.flush(); import os; os._exit(3)
bad = 0
for r in (0, 1):
    for key, h in out[r].items():
        for i, q in enumerate(qs):
            ref = H.oracle_keyword(orc, q)
            n = min(100, ref.keys.size)
            if int(h.n_hits[i]) != n or not np.array_equal(h.keys[i, :n], ref.keys[:n]) or not np.array_equal(h.scores[i, :n], ref.scores[:n]) or int(h.num_matched[i]) != int(ref.num_keyword_matches):
                bad += 1
print("RCCL_TWO_RANKS_OK" if bad == 0 else "RCCL_TWO_RANKS_MISMATCH %%d" %% bad)
"""


@pytest.mark.gpu
def test_rccl_with_two_ranks_on_the_one_device_if_rccl_permits_it():
    """ncclAllToAll / ncclSend / ncclRecv / in-place ncclAllGather with MORE THAN ONE rank have only run with n = 1 (one GPU per
    box here). Two contexts on device 0, two threads, ncclCommInitRank(2 ranks): RCCL normally refuses duplicate devices — then the refusal is
    recorded in the skip reason; if it permits them, both exchange forms (pruned / full, slices / all-gather) must equal the unsharded oracle.
    Runs in a subprocess: a communicator that cannot form must not take the test session with it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
    try:
        p = subprocess.run([sys.executable, "-c", RCCL_TWO_RANKS_ONE_DEVICE % {"root": root}], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    except subprocess.TimeoutExpired:
        pytest.skip("RCCL with two ranks on one device: the subprocess did not finish within 600 s (communicator set-up hangs on duplicate devices)")
    tail = (p.stdout + p.stderr)[-1500:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("RCCL_")]
    if not lines:
        pytest.skip("RCCL with two ranks on one device: no verdict (rc %d): %s" % (p.returncode, tail))
    if lines[-1].startswith("RCCL_REFUSED"):
        pytest.skip("RCCL refuses two ranks on one device: " + lines[-1][len("RCCL_REFUSED "):] + " | " + " ".join(l for l in tail.splitlines() if "NCCL WARN" in l)[-400:])
    assert lines[-1] == "RCCL_TWO_RANKS_OK", tail
