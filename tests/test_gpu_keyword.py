"""`-m gpu`: the keyword hot path on a real MI355X through the C-ABI (libtsgpu.so), bit-exact against the oracle
(doc-id sets, num_keyword_matches, top-K order and every score). Collections follow SURVEY §8(d) config 1
(100K docs, V=20K, 16 tokens/doc) plus a 2M-doc collection for multi-work-item queries; the full 10M-doc size
is covered by size-independent properties (sortedness, subset/superset relations, idempotence, shard merge)."""
import numpy as np
import pytest

import typesense_amd as T
from typesense_amd import _lib as B, synth
from oracle import oracle_py as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _real_library(monkeypatch):
    """bodies shared with tests/test_emu_keyword.py ask for the emulator build: give them the real library here"""
    monkeypatch.setattr(H, "emu_lib_path", lambda *a, **k: H.gpu_lib_path())

SORT = ((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0))
OSORT = ((O.SORT_TEXT_MATCH, 0, 1), (O.SORT_INT64_COLUMN, 0, 1))


class Corpus:
    def __init__(self, n_docs, vocab, tpd, seed):
        self.n_docs = n_docs
        self.csr = synth.zipf_corpus_csr(n_docs, vocab, tpd, seed)
        self.pts = synth.points_column(n_docs)
        self.g = T.GpuIndex(0)
        self.g.field_create(0, False)
        c = self.csr
        self.g.terms_load_csr(0, c["term_ids"], c["ids_ptr"], c["ids"], c["offset_index"], c["off_ptr"], c["offsets"])
        self.g.column_set(0, self.pts)
        self.g.set_num_docs(n_docs)
        self.g.commit()
        self.orc = O.OracleIndex(1, 1)
        self.orc.set_num_docs(n_docs)
        self.orc.set_sort_dense(0, self.pts)
        self.loaded = set()

    def need(self, terms):
        for t in np.unique(np.asarray(terms).ravel()):
            t = int(t)
            if t in self.loaded or t < 1 or t > self.csr["term_ids"].size:
                continue
            ids, oi, off = synth.csr_term(self.csr, t)
            if ids.size:
                self.orc.load_posting(0, t, ids, oi, off)
            self.loaded.add(t)

    def oracle(self, q, ids_cap=0):
        self.need(q.tokens)
        return H.oracle_keyword(self.orc, q, ids_cap=ids_cap)


@pytest.fixture(scope="module")
def c100k():
    c = Corpus(100_000, 20_000, 16, seed=1)
    yield c
    c.g.close()


@pytest.fixture(scope="module")
def c2m():
    c = Corpus(2_000_000, 50_000, 24, seed=7)
    yield c
    c.g.close()


def test_format_roundtrip_on_device(c100k):
    for t in (1, 2, 50, 1234, 19999):
        ids, oi, off = synth.csr_term(c100k.csr, t)
        if not ids.size:
            continue
        gi, go, gf = c100k.g.term_download(0, t)
        assert np.array_equal(ids, gi) and np.array_equal(oi, go) and np.array_equal(off, gf)


def test_config1_three_term_and_top10_bit_exact(c100k):
    """BASELINE config 1: 1 000 queries, 3 distinct terms, ranks log-uniform [5,500], per_page=10 -> Topster 250"""
    qtok = synth.keyword_queries(1000, 3, 5, 500, seed=11)
    qs = [T.KwQuery(q, sort=SORT, topster_size=250) for q in qtok]
    hits = c100k.g.keyword_search_batch(qs, k_stride=250)
    assert (hits.status == 0).all()
    nonempty = 0
    for i, q in enumerate(qs):
        H.assert_hits_equal(hits, i, c100k.oracle(q), "config1")
        nonempty += int(hits.n_hits[i] > 0)
    assert nonempty > 100


@pytest.mark.parametrize("n_tok", [1, 2, 4, 6, 10])
def test_token_counts_bit_exact(c100k, n_tok):
    qtok = synth.keyword_queries(40, n_tok, 1, 60 if n_tok > 3 else 300, seed=20 + n_tok)
    qs = [T.KwQuery(q, sort=SORT, topster_size=250) for q in qtok]
    hits = c100k.g.keyword_search_batch(qs, k_stride=250)
    for i, q in enumerate(qs):
        H.assert_hits_equal(hits, i, c100k.oracle(q), "T=%d" % n_tok)


def test_flags_sorts_and_duplicates_bit_exact(c100k):
    base = dict(topster_size=250)
    qs = [
        T.KwQuery([3, 3], **base), T.KwQuery([2, 5, 2], **base), T.KwQuery([1, 999999, 4], **base), T.KwQuery([999999], **base),
        T.KwQuery([1, 2], prioritize_token_position=True, **base), T.KwQuery([6], prioritize_token_position=True, **base),
        T.KwQuery([1, 2, 3], prioritize_exact_match=False, **base), T.KwQuery([1, 4], prioritize_num_matching_fields=False, **base),
        T.KwQuery([2, 3], match_type=B.MAX_WEIGHT, weight=7, **base), T.KwQuery([2, 3], match_type=B.MAX_WEIGHT, weight=0, **base),
        T.KwQuery([2, 3], match_type=B.SUM_SCORE, weight=3, **base), T.KwQuery([5, 1], total_cost=3, **base), T.KwQuery([8], **base),
        T.KwQuery([1, 2], sort=((B.SORT_INT64_COLUMN, -1, 0), (B.SORT_TEXT_MATCH, 1, 0), (B.SORT_SEQ_ID, -1, 0)), **base),
        T.KwQuery([1, 2], sort=((B.SORT_TEXT_MATCH, -1, 0), (B.SORT_SEQ_ID, 1, 0)), **base),
        T.KwQuery([1, 3], sort=((B.SORT_SEQ_ID, 1, 0),), **base),
        T.KwQuery([1, 2], topster_size=5, sort=SORT), T.KwQuery([1], topster_size=3), T.KwQuery([2, 1, 3], topster_size=1),
        T.KwQuery([1, 2], topster_size=1000, sort=SORT), T.KwQuery([4], topster_size=600, sort=SORT),
    ]
    hits = c100k.g.keyword_search_batch(qs, k_stride=1000)
    assert (hits.status == 0).all()
    for i, q in enumerate(qs):
        H.assert_hits_equal(hits, i, c100k.oracle(q), "flags")


def test_result_ids_and_excluded_ids(c100k):
    g = c100k.g
    g.keep_result_ids(True)
    try:
        q0 = T.KwQuery([1, 2], topster_size=250, sort=SORT)
        ref0 = c100k.oracle(q0, ids_cap=200000)
        excl = np.sort(ref0.result_ids[::3])
        qs = [q0, T.KwQuery([1, 2], topster_size=250, sort=SORT, excluded_ids=excl), T.KwQuery([7], topster_size=250, sort=SORT)]
        hits = g.keyword_search_batch(qs, k_stride=250)
        for i, q in enumerate(qs):
            ref = c100k.oracle(q, ids_cap=200000)
            H.assert_hits_equal(hits, i, ref, "ids")
            assert np.array_equal(g.result_ids(i), ref.result_ids)
    finally:
        g.keep_result_ids(False)


def test_two_million_docs_multi_chunk_bit_exact(c2m):
    """lists of up to ~1M ids: drivers span many 64-block work items; partial top-K merge on device"""
    qtok = np.concatenate([synth.keyword_queries(60, 3, 1, 200, seed=31), synth.keyword_queries(20, 2, 1, 50, seed=32)[:, [0, 1, 1]]])
    qs = [T.KwQuery(q[:3] if i < 60 else q[:2], sort=SORT, topster_size=250) for i, q in enumerate(qtok)]
    qs += [T.KwQuery([1], sort=SORT, topster_size=250), T.KwQuery([2], sort=((B.SORT_SEQ_ID, 1, 0),), topster_size=250)]
    assert c2m.g.term_num_ids(0, 1) > 64 * 256 * 4
    hits = c2m.g.keyword_search_batch(qs, k_stride=250)
    assert (hits.status == 0).all()
    for i, q in enumerate(qs):
        H.assert_hits_equal(hits, i, c2m.oracle(q), "2M")


def test_size_independent_properties(c2m):
    """properties that hold at any size (used again at 10M docs by bench.py --check)"""
    g = c2m.g
    qtok = synth.keyword_queries(300, 3, 1, 400, seed=41)
    q3 = [T.KwQuery(q, sort=SORT, topster_size=250) for q in qtok]
    q2 = [T.KwQuery(q[:2], sort=SORT, topster_size=250) for q in qtok]
    h3 = g.keyword_search_batch(q3, k_stride=250)
    h2 = g.keyword_search_batch(q2, k_stride=250)
    again = g.keyword_search_batch(q3, k_stride=250)
    for i in range(len(q3)):
        n = int(h3.n_hits[i])
        # idempotence
        assert n == again.n_hits[i] and np.array_equal(h3.keys[i, :n], again.keys[i, :n]) and np.array_equal(h3.scores[i, :n], again.scores[i, :n])
        # Topster::sort() order: (s0, s1, s2, key) strictly descending
        tup = [tuple(h3.scores[i, j]) + (int(h3.keys[i, j]),) for j in range(n)]
        assert all(tup[j] > tup[j + 1] for j in range(n - 1))
        # adding a token can only shrink the match set
        assert h3.num_matched[i] <= h2.num_matched[i]
        assert n == min(250, int(h3.num_matched[i]))
        # text_match of a 3-token hit encodes 3 tokens matched, 1 field
        if n:
            tm = h3.text_match[i, :n].astype(np.uint64)
            assert ((tm >> np.uint64(59)) == 3).all() and ((tm & np.uint64(7)) == 1).all()


# ---- the emulator bodies against the real library (filter ids inside the AND loop, several query_by fields) ----
from tests import test_emu_keyword as EK


@pytest.fixture(scope="module")
def pair():
    docs = H.zipf_docs(3000, 300, 12, seed=1)
    orc, g = H.build_pair(docs, H.gpu_lib_path())
    yield orc, g, docs
    g.close()


@pytest.fixture(scope="module")
def pair3():
    rng = np.random.default_rng(41)
    title = H.zipf_docs(2500, 120, 6, seed=11)
    body = H.zipf_docs(2500, 120, 14, seed=12)
    tags = H.zipf_docs(2500, 120, 4, seed=13)
    title[rng.random(title.shape) < 0.3] = 0
    tags[rng.random(tags.shape) < 0.6] = 0
    body[rng.random(2500) < 0.1] = 0
    orc, g = H.build_pair_fields([title, body, tags], H.gpu_lib_path())
    yield orc, g
    g.close()


test_keyword_filter_ids_hits_ids_and_the_reference_match_count = EK.test_keyword_filter_ids_hits_ids_and_the_reference_match_count
test_keyword_filter_ids_with_excluded_ids = EK.test_keyword_filter_ids_with_excluded_ids
test_multi_field_union_per_token_and_field_aggregation = EK.test_multi_field_union_per_token_and_field_aggregation
test_wildcard_search_ranks_filter_ids_by_sort_keys = EK.test_wildcard_search_ranks_filter_ids_by_sort_keys
test_edge_cases_empty_index_missing_tokens_and_degenerate_topsters = EK.test_edge_cases_empty_index_missing_tokens_and_degenerate_topsters
test_candidate_combinations_fold_like_the_shared_topster_and_id_buff = EK.test_candidate_combinations_fold_like_the_shared_topster_and_id_buff
test_two_kernel_form_and_fused_kernel_agree_with_the_oracle = EK.test_two_kernel_form_and_fused_kernel_agree_with_the_oracle
test_dropped_tokens_are_scored_when_present_and_never_required = EK.test_dropped_tokens_are_scored_when_present_and_never_required
test_pair_find_kernel_matches_the_oracle = EK.test_pair_find_kernel_matches_the_oracle
test_synonym_passes_score_like_score_results2 = EK.test_synonym_passes_score_like_score_results2
test_parallel_planning_of_a_batch_gives_the_serial_plan = EK.test_parallel_planning_of_a_batch_gives_the_serial_plan
test_device_side_planner_equals_the_host_planner_and_the_oracle = EK.test_device_side_planner_equals_the_host_planner_and_the_oracle
test_multi_field_block_merge_windows_wide_runs_and_exhaustion = EK.test_multi_field_block_merge_windows_wide_runs_and_exhaustion
test_long_work_items_reload_the_driver_metadata_window = EK.test_long_work_items_reload_the_driver_metadata_window
test_pipelined_two_field_find_kernel_equals_the_block_at_a_time_kernel_and_the_oracle = EK.test_pipelined_two_field_find_kernel_equals_the_block_at_a_time_kernel_and_the_oracle
test_find_kernel_counts_the_bytes_it_requests_without_changing_the_results = EK.test_find_kernel_counts_the_bytes_it_requests_without_changing_the_results


def test_device_shard_merge_on_cuda_tensors(c100k):
    """the array form of the shard merge on the GPU: gathered CUDA tensors -> kw_shard_merge_kernel (tsgpu_merge_shard_hits_device) == a numpy lexsort merge"""
    import ctypes as C
    import torch
    gen = torch.Generator(device="cpu").manual_seed(3)
    G, Bq, K = 8, 300, 250
    sc = torch.randint(0, 5, (G, Bq, K, 3), generator=gen, dtype=torch.int64)
    keys = (torch.rand((G, Bq, K), generator=gen) * 1e6).to(torch.int64) * 8 + torch.arange(G)[:, None, None]      # shard-unique keys
    n_hits = torch.randint(0, K + 1, (G, Bq), generator=gen, dtype=torch.int32)
    # sort every per-shard list into Topster order
    flat = torch.stack([sc[..., 0], sc[..., 1], sc[..., 2], keys], -1).reshape(G * Bq, K, 4).numpy()
    for r in range(G * Bq):
        flat[r] = flat[r][np.lexsort((flat[r][:, 3], flat[r][:, 2], flat[r][:, 1], flat[r][:, 0]))[::-1]]
    flat = torch.from_numpy(flat.copy()).reshape(G, Bq, K, 4)
    sc, keys = flat[..., :3].contiguous(), flat[..., 3].contiguous()
    num = torch.randint(0, 10000, (G, Bq), generator=gen, dtype=torch.int64)
    ref = H.reference_shard_merge(keys.numpy(), sc.numpy(), n_hits.numpy(), 250)
    g = {"keys": keys.cuda(), "scores": sc.cuda(), "n_hits": n_hits.cuda(), "num_matched": num.cuda()}
    out = dict(keys=torch.empty((Bq, 250), dtype=torch.int64, device="cuda"), scores=torch.empty((Bq, 250, 3), dtype=torch.int64, device="cuda"),
               n_hits=torch.empty(Bq, dtype=torch.int32, device="cuda"), num_matched=torch.empty(Bq, dtype=torch.int64, device="cuda"))
    hin, hout = B.HitsC(), B.HitsC()
    hin.mem = hout.mem = B.MEM_DEVICE
    hin.k_stride, hout.k_stride = K, 250
    for name in ("keys", "scores", "n_hits", "num_matched"):
        setattr(hin, name, g[name].data_ptr())
        setattr(hout, name, out[name].data_ptr())
    torch.cuda.synchronize()
    B.check(c100k.g.L, c100k.g.L.tsgpu_merge_shard_hits_device(c100k.g.h, C.byref(hin), G, Bq, 250, C.byref(hout)))
    k_, s_, n_ = out["keys"].cpu().numpy(), out["scores"].cpu().numpy(), out["n_hits"].cpu().numpy()
    for q in range(Bq):
        n = ref[q][0].size
        assert int(n_[q]) == n and np.array_equal(k_[q, :n], ref[q][0]) and np.array_equal(s_[q, :n], ref[q][1])
    assert torch.equal(out["num_matched"].cpu(), num.sum(0))
test_string_array_fields_match_per_element_and_mix_with_plain_fields = EK.test_string_array_fields_match_per_element_and_mix_with_plain_fields


@pytest.fixture(scope="module")
def pair_arr():
    orc, g = EK.make_pair_arr(H.gpu_lib_path())       # the REAL library, resolved explicitly (round-2 leak: the emulator ran here)
    assert "emu" not in g.lib_path
    yield orc, g
    g.close()


def test_wildcard_over_2m_docs(c2m):
    """q = "*" over every document and over a 30 % filter: many work items, partial top-K merge, exact order"""
    g = c2m.g
    rng = np.random.default_rng(8)
    filt = np.unique(rng.integers(0, 2_000_000, size=600_000)).astype(np.uint32)
    qs = [T.KwQuery([], sort=((B.SORT_INT64_COLUMN, 1, 0), (B.SORT_SEQ_ID, -1, 0)), topster_size=250),
          T.KwQuery([], sort=((B.SORT_INT64_COLUMN, -1, 0), (B.SORT_SEQ_ID, 1, 0)), topster_size=100, filter_ids=filt),
          T.KwQuery([], sort=((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_INT64_COLUMN, 1, 0)), topster_size=250, filter_ids=filt,
                    excluded_ids=filt[::2])]
    hits = g.wildcard_search_batch(qs, k_stride=250)
    assert (hits.status == 0).all()
    for i, q in enumerate(qs):
        H.assert_hits_equal(hits, i, H.oracle_wildcard(c2m.orc, q), "wildcard 2M")


def test_filter_ids_on_2m_docs_multi_chunk(c2m):
    """filters of very different selectivity against lists of ~1M ids: hits, match counts (chained over many work items), ids"""
    g = c2m.g
    rng = np.random.default_rng(5)
    g.keep_result_ids(True)
    try:
        qs = []
        for sel in (0.5, 0.01, 0.0001):
            filt = np.unique(rng.integers(0, 2_000_000, size=int(2_000_000 * sel))).astype(np.uint32)
            for toks in ([1, 2], [3, 1, 2], [2]):
                qs.append(T.KwQuery(toks, sort=SORT, topster_size=250, filter_ids=filt))
        hits = g.keyword_search_batch(qs, k_stride=250)
        assert (hits.status == 0).all()
        for i, q in enumerate(qs):
            ref = c2m.oracle(q, ids_cap=2_000_000)
            H.assert_hits_equal(hits, i, ref, "2M filter")
            assert np.array_equal(g.result_ids(i), ref.result_ids)
    finally:
        g.keep_result_ids(False)


def test_candidate_combinations_on_2m_docs(c2m):
    """search_all_candidates at size: 10 combinations of frequent tokens per user query (multi-work-item passes, ~1M-id unions)"""
    g = c2m.g
    rng = np.random.default_rng(9)
    groups = []
    for shape in [(2, 5), (3, 3), (2, 2, 2), (1, 4)]:
        cands = [rng.choice(np.arange(1, 40), size=c, replace=False) for c in shape]
        combos = []
        for x in range(min(int(np.prod(shape)), 10)):
            toks, r = [], x
            for pos in range(len(shape) - 1, -1, -1):
                toks.insert(0, int(cands[pos][r % shape[pos]])); r //= shape[pos]
            combos.append(T.KwQuery(toks, sort=SORT, topster_size=250, total_cost=int(x % 3)))
        groups.append(combos)
    hits, qidx, found = g.keyword_search_candidates_batch(groups, k_stride=250)
    assert (hits.status == 0).all()
    for gi, combos in enumerate(groups):
        for q in combos:
            c2m.need(q.tokens)
        ref, ref_qi = H.oracle_candidates(c2m.orc, combos, ids_cap=2_000_000)
        H.assert_hits_equal(hits, gi, ref, "candidates 2M g%d" % gi)
        assert np.array_equal(qidx[gi, :int(hits.n_hits[gi])], ref_qi)
        assert int(found[gi]) == int(ref.n_result_ids)
        assert np.array_equal(g.candidates_result_ids(gi), ref.result_ids)


@pytest.mark.parametrize("select_min", [2, 17, 0])
def test_many_work_items_two_level_merge_on_2m_docs(c2m, select_min):
    """kw_chunk_blocks = 1: the frequent terms' queries are cut into hundreds of work items -> merged by selection (kw_select_partials; past
    384 lists or 1 024 candidates: folded), or with kw_merge_select_min = 0 by kw_merge_groups_kernel + kw_merge_kernel"""
    c2m.g.set_option("kw_chunk_blocks", 1)
    c2m.g.set_option("kw_merge_select_min", select_min)
    try:
        qtok = synth.keyword_queries(24, 3, 1, 60, seed=41)
        qs = [T.KwQuery(q, sort=SORT, topster_size=250) for q in qtok] + [T.KwQuery([3], sort=SORT, topster_size=100)] + \
             [T.KwQuery(q, sort=SORT, topster_size=ts) for q, ts in zip(qtok[:6], (400, 510, 300, 64, 500, 7))]      # (2k beyond half the LDS buffer: sorted, not tree-merged)
        hits = c2m.g.keyword_search_batch(qs, k_stride=512)
        assert (hits.status == 0).all()
        for i, q in enumerate(qs):
            H.assert_hits_equal(hits, i, c2m.oracle(q), "two-level merge 2M")
    finally:
        c2m.g.set_option("kw_chunk_blocks", 0)
        c2m.g.set_option("kw_merge_select_min", 2)


def test_deadline_in_flight_partial_hits_on_2m_docs(c2m):
    """a batch of 6 000 queries takes milliseconds; with a few hundred us of budget queries are cut off ON THE DEVICE: status 0,
    search_cutoff 1, the hits returned are a subset of the full result with identical scores (408 only for queries already late when
    the batch was planned). The budget that lands between planning and the end of the launch depends on the box: several are tried."""
    import ctypes as C
    import time
    qtok = synth.keyword_queries(6000, 3, 1, 3000, seed=43)         # heavy and light queries (the light ones start last: heaviest work first)
    full = c2m.g.keyword_search_batch([T.KwQuery(q, sort=SORT, topster_size=250) for q in qtok], k_stride=250)
    assert (full.status == 0).all() and (full.search_cutoff == 0).all()
    arr = T.index.make_query_array([T.KwQuery(q, sort=SORT, topster_size=250, deadline_us=1) for q in qtok])
    # the deadline is stamped into the prebuilt query array in one vectorised store right before the call (filling 6 000 structs from
    # Python takes longer than the budget)
    dl = np.ndarray((len(qtok),), dtype=np.uint64, buffer=arr, offset=B.KwQueryC.deadline_us.offset, strides=(C.sizeof(B.KwQueryC),))
    checked = in_flight = 0
    seen = []
    for budget in (300, 500, 800, 1200, 2000, 3500, 6000):
        dl[:] = int(time.time() * 1e6) + budget
        hits = c2m.g.keyword_search_batch(arr, k_stride=250)
        cut, ok = hits.search_cutoff == 1, hits.status == 0
        assert (hits.status[~ok] == B.ERR_DEADLINE).all() and cut[~ok].all()
        seen.append((budget, int(cut.sum()), int((cut & ok).sum())))
        in_flight += int((cut & ok).sum())
        for i in np.nonzero(ok)[0]:
            n, nf = int(hits.n_hits[i]), int(full.n_hits[i])
            if not cut[i]:
                assert n == nf and np.array_equal(hits.keys[i, :n], full.keys[i, :n]) and np.array_equal(hits.scores[i, :n], full.scores[i, :n])
            elif int(full.num_matched[i]) <= 250:          # the full Topster holds every match: a partial result must be a subset of it
                ref = {int(k): tuple(int(x) for x in s) for k, s in zip(full.keys[i, :nf], full.scores[i, :nf])}
                assert all(int(k) in ref and ref[int(k)] == tuple(int(x) for x in s) for k, s in zip(hits.keys[i, :n], hits.scores[i, :n]))
                assert int(hits.num_matched[i]) <= int(full.num_matched[i])
                checked += 1
    assert in_flight > 0 and checked > 0, "no budget cut a query in flight: (budget us, cut, cut in flight) = %s" % seen


def test_device_side_planner_on_2m_docs_many_work_items(c2m):
    """the device planner at size: 3 000 three-term queries over 2M docs (several work items per query, both launch tables, heaviest-first
    layout, hit offsets of a multi-GB hit buffer) = the host planner's results; a sample against the oracle"""
    rng = np.random.default_rng(5)
    qtok = synth.keyword_queries(3000, 3, 5, 1500, seed=12)
    qs = [T.KwQuery(qtok[i], sort=SORT, topster_size=250) for i in range(3000)]
    qs += [T.KwQuery(rng.choice(np.arange(5, 400), size=5, replace=False), sort=SORT, topster_size=250) for _ in range(40)]
    g = c2m.g
    try:
        g.set_option("kw_device_plan_min_queries", 0)
        host = g.keyword_search_batch(qs, k_stride=250)
        g.set_option("kw_device_plan_min_queries", 512)
        n0 = g.counter("kw_device_plans")
        dev = g.keyword_search_batch(qs, k_stride=250)
        assert g.counter("kw_device_plans") > n0
    finally:
        g.set_option("kw_device_plan_min_queries", 512)
    assert np.array_equal(dev.status, host.status) and np.array_equal(dev.n_hits, host.n_hits) and np.array_equal(dev.num_matched, host.num_matched)
    for i in range(len(qs)):
        n = int(host.n_hits[i])
        assert np.array_equal(dev.keys[i, :n], host.keys[i, :n]) and np.array_equal(dev.scores[i, :n], host.scores[i, :n]), i
    for i in list(range(0, 3000, 250)) + [3001, 3020]:
        H.assert_hits_equal(dev, i, c2m.oracle(qs[i]), "device plan at 2M docs")


@pytest.mark.parametrize("n_fields", [2, 3, 4])
def test_pipelined_find_kernel_on_1m_docs_several_fields(n_fields):
    """kw_find_mf2_kernel<., 2> / <., 4> at size: 1M documents, two to four string fields, 400 three-term queries over all of them (driver lists of hundreds of blocks: window
    slides and re-centring, runs in both tile sizes and wider than the tile, many work items per query) — identical to kw_search_mf_kernel on
    every output array, and 24 of them identical to the oracle's or_iterator_t union (/root/reference/src/or_iterator.cpp:95-171)"""
    n_docs = 1_000_000
    csr = [synth.zipf_corpus_csr(n_docs, 20_000, (20, 10, 6, 12)[f], seed=71 + f) for f in range(n_fields)]
    pts = synth.points_column(n_docs)
    g = T.GpuIndex(0)
    for f, c in enumerate(csr):
        g.field_create(f, False)
        g.terms_load_csr(f, c["term_ids"], c["ids_ptr"], c["ids"], c["offset_index"], c["off_ptr"], c["offsets"])
    g.column_set(0, pts)
    g.set_num_docs(n_docs)
    g.commit()
    try:
        qtok = np.concatenate([synth.keyword_queries(300, 3, 1, 300, seed=81), synth.keyword_queries(100, 3, 2, 4000, seed=82)])
        fields = tuple((f, 15 - f) for f in range(n_fields))
        qs = [T.KwQuery(q, sort=SORT, topster_size=250, fields=fields if i % 3 else fields[::-1]) for i, q in enumerate(qtok)]
        outs = []
        for pipelined in (1, 0):
            g.set_option("kw_mf_pipelined", pipelined)
            n0 = g.counter("kw_mf_pipelined_launches")
            h = g.keyword_search_batch(qs, k_stride=250)
            assert (h.status == 0).all()
            assert (g.counter("kw_mf_pipelined_launches") > n0) == bool(pipelined)
            outs.append(h)
        for name in ("keys", "scores", "n_hits", "num_matched"):
            assert np.array_equal(getattr(outs[0], name), getattr(outs[1], name)), name
        assert int(outs[0].n_hits.sum()) > 20_000
        orc = O.OracleIndex(n_fields, 1)
        orc.set_num_docs(n_docs)
        orc.set_sort_dense(0, pts)
        for t in np.unique(qtok[:24]):
            for f, c in enumerate(csr):
                ids, oi, off = synth.csr_term(c, int(t))
                if ids.size:
                    orc.load_posting(f, int(t), ids, oi, off)
        for i in range(24):
            H.assert_hits_equal(outs[0], i, H.oracle_keyword(orc, qs[i]), "1M docs, %d fields, pipelined find kernel" % n_fields)
        orc.close()
    finally:
        g.close()
