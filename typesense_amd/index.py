"""Host-side mirror of the reference's operator surface for the hot path, over the C-ABI (include/tsgpu.h).

GpuIndex wraps one tsgpu context = one GPU's shard of: the posting lists of the query_by fields
(Index::search_index, include/index.h), the numeric sort index (include/index.h:442) and the vector fields
(hnsw_index_t, include/index.h:356-389). Every method is a 1:1 call into libtsgpu.so; nothing is computed
in Python.
"""
import ctypes as C
import os
import numpy as np

from . import _lib as B


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class KwQuery:
    """One search_across_fields call (src/index.cpp:5385): tokens of one candidate combination + ranking params."""

    def __init__(self, tokens, field=0, weight=15, sort=((B.SORT_TEXT_MATCH, 1, 0), (B.SORT_SEQ_ID, 1, 0)),
                 topster_size=0, match_type=B.MAX_SCORE, prioritize_exact_match=True, prioritize_token_position=False,
                 prioritize_num_matching_fields=True, total_cost=0, excluded_ids=None, filter_ids=None, deadline_us=0,
                 n_fields=None, fields=None, dropped_tokens=(), syn_orig_num_tokens=-1, orig_num_tokens=0, is_synonym_query=False,
                 demote_synonym_match=False):
        self.tokens = list(tokens)
        self.dropped_tokens = list(dropped_tokens)      # scored when present, never required (drop_tokens passes)
        self.syn_orig_num_tokens, self.orig_num_tokens = syn_orig_num_tokens, orig_num_tokens       # synonym passes
        self.is_synonym_query, self.demote_synonym_match = is_synonym_query, demote_synonym_match
        self.field, self.weight, self.sort = field, weight, tuple(sort)     # sort: (kind, order, column)
        self.fields = [(int(f), int(w)) for f, w in fields] if fields else [(field, weight)]     # query_by fields: (field id, weight)
        self.topster_size = topster_size
        self.match_type = match_type
        self.prioritize_exact_match = prioritize_exact_match
        self.prioritize_token_position = prioritize_token_position
        self.prioritize_num_matching_fields = prioritize_num_matching_fields
        self.total_cost = total_cost
        self.excluded_ids = None if excluded_ids is None else _u32(excluded_ids)
        self.filter_ids = None if filter_ids is None else _u32(filter_ids)
        self.deadline_us = deadline_us
        self.n_fields = len(self.fields) if n_fields is None else n_fields

    def fill(self, c):
        c.n_tokens = len(self.tokens)
        for i, t in enumerate(self.tokens[:B.MAX_QUERY_TOKENS]):
            c.term_ids[i] = int(t)
        c.n_fields = self.n_fields
        for i, (f, w) in enumerate(self.fields[:4]):
            c.field_ids[i] = f
            c.field_weights[i] = w
        c.match_type = self.match_type
        c.prioritize_exact_match = int(self.prioritize_exact_match)
        c.prioritize_token_position = int(self.prioritize_token_position)
        c.prioritize_num_matching_fields = int(self.prioritize_num_matching_fields)
        c.total_cost = self.total_cost
        c.n_sort = len(self.sort)
        for i, s in enumerate(self.sort[:3]):
            c.sort[i].kind, c.sort[i].order, c.sort[i].column = s
        c.topster_size = self.topster_size
        if self.excluded_ids is not None and self.excluded_ids.size:
            c.excluded_ids = self.excluded_ids.ctypes.data_as(C.POINTER(C.c_uint32))
            c.n_excluded = self.excluded_ids.size
        if self.filter_ids is not None and self.filter_ids.size:
            c.filter_ids = self.filter_ids.ctypes.data_as(C.POINTER(C.c_uint32))
            c.n_filter = self.filter_ids.size
        c.deadline_us = self.deadline_us
        c.is_synonym_query, c.demote_synonym_match = int(self.is_synonym_query), int(self.demote_synonym_match)
        c.syn_orig_num_tokens_p1, c.orig_num_tokens = self.syn_orig_num_tokens + 1, self.orig_num_tokens
        c.n_dropped = len(self.dropped_tokens)
        for i, t in enumerate(self.dropped_tokens[:4]):
            c.dropped_term_ids[i] = int(t)


class Hits:
    """Host copy of tsgpu_hits for a batch (numpy, [n_queries, k_stride, ...])."""

    def __init__(self, n_queries, k_stride):
        self.n_queries, self.k_stride = n_queries, k_stride
        self.keys = np.zeros((n_queries, k_stride), np.uint64)
        self.scores = np.zeros((n_queries, k_stride, 3), np.int64)
        self.text_match = np.zeros((n_queries, k_stride), np.int64)
        self.vector_distance = np.zeros((n_queries, k_stride), np.float32)
        self.match_score_index = np.zeros((n_queries, k_stride), np.int8)
        self.n_hits = np.zeros(n_queries, np.uint32)
        self.num_matched = np.zeros(n_queries, np.uint64)
        self.status = np.zeros(n_queries, np.int32)
        self.search_cutoff = np.zeros(n_queries, np.int32)

    def c_struct(self, seam_arrays_only=False):
        """seam_arrays_only: what the B1 shim asks for when the query sorts on _text_match (typesense_amd/csrc/host/tsgpu_keyword_shim.h) — keys, scores
        and match_score_index; KV::text_match_score is scores[match_score_index] and KV::vector_distance its default there, so those two
        arrays are not requested (NULL: the library does not deliver them)."""
        h = B.HitsC()
        h.mem = B.MEM_HOST
        h.k_stride = self.k_stride
        h.keys, h.scores, h.text_match = self.keys.ctypes.data, self.scores.ctypes.data, (None if seam_arrays_only else self.text_match.ctypes.data)
        h.vector_distance, h.match_score_index = (None if seam_arrays_only else self.vector_distance.ctypes.data), self.match_score_index.ctypes.data
        h.n_hits, h.num_matched = self.n_hits.ctypes.data, self.num_matched.ctypes.data
        h.status, h.search_cutoff = self.status.ctypes.data, self.search_cutoff.ctypes.data
        return h


class GroupedHits:
    """Host copy of tsgpu_grouped_hits (numpy, [n_queries, g_stride])."""

    def __init__(self, n_queries, g_stride, registers=False, totals=True):
        self.n_queries, self.g_stride, self.totals = n_queries, g_stride, totals
        self.n_groups = np.zeros(n_queries, np.uint32)
        self.distinct_key = np.zeros((n_queries, g_stride), np.uint64)
        self.group_size = np.zeros((n_queries, g_stride), np.uint32)
        self.group_found = np.zeros((n_queries, g_stride), np.uint32)
        self.groups_total = np.zeros(n_queries, np.uint64)
        self.groups_count = np.zeros(n_queries, np.uint64)
        self.loglog_registers = np.zeros((n_queries, 16384), np.uint8) if registers else None

    def c_struct(self):
        g = B.GroupedHitsC()
        g.g_stride = self.g_stride
        g.n_groups, g.distinct_key, g.group_size, g.group_found = self.n_groups.ctypes.data, self.distinct_key.ctypes.data, self.group_size.ctypes.data, self.group_found.ctypes.data
        g.groups_total, g.groups_count = (self.groups_total.ctypes.data if self.totals else None), self.groups_count.ctypes.data
        g.loglog_registers = self.loglog_registers.ctypes.data if self.loglog_registers is not None else None
        return g


def make_query_array(queries):
    """list[KwQuery] (or a prebuilt ctypes array) -> contiguous tsgpu_kw_query[n]"""
    if isinstance(queries, C.Array):
        return queries
    arr = (B.KwQueryC * len(queries))()
    for i, q in enumerate(queries):
        q.fill(arr[i])
    arr._keep = queries
    return arr


class GpuIndex:
    def __init__(self, device=0, lib_path=None):
        self.L = B.lib(lib_path)
        self.lib_path = os.path.realpath(lib_path or B.LIB_PATH)
        h = C.c_void_p()
        B.check(self.L, self.L.tsgpu_create(device, C.byref(h)))
        self.h = h
        self.vec_dim = {}

    def close(self):
        if getattr(self, "h", None):
            self.L.tsgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        B.check(self.L, rc)

    # ---- keyword index mirror ----
    def field_create(self, field_id, is_array=False):
        self._ck(self.L.tsgpu_field_create(self.h, field_id, int(is_array)))

    def term_upsert(self, field_id, term_id, ids, offset_index, offsets):
        ids, offset_index, offsets = _u32(ids), _u32(offset_index), _u32(offsets)
        self._ck(self.L.tsgpu_term_upsert(self.h, field_id, term_id, _vp(ids), _vp(offset_index), _vp(offsets), ids.size, offsets.size))

    def posting_upsert(self, field_id, term_id, seq_id, offsets):
        off = _u32(offsets)
        self._ck(self.L.tsgpu_posting_upsert(self.h, field_id, term_id, seq_id, _vp(off), off.size))

    def posting_erase(self, field_id, term_id, seq_id):
        self._ck(self.L.tsgpu_posting_erase(self.h, field_id, term_id, seq_id))

    def index_plain_doc(self, seq_id, field_id, tokens):
        """what Index::tokenize_string + posting_t::upsert do for one plain string field of one document (src/index.cpp:1323-1348):
        per distinct token its positions + 1, and a trailing 0 for the token that ends the field"""
        toks = [int(t) for t in tokens]
        per = {}
        for pos, t in enumerate(toks):
            per.setdefault(t, []).append(pos + 1)
        if toks:
            per[toks[-1]].append(0)
        for t, off in per.items():
            self.posting_upsert(field_id, t, seq_id, off)

    def remove_plain_doc(self, seq_id, field_id, tokens):
        for t in set(int(x) for x in tokens):
            self.posting_erase(field_id, t, seq_id)

    def terms_load_csr(self, field_id, term_ids, ids_ptr, ids, offset_index, off_ptr, offsets):
        term_ids, ids, offsets = _u32(term_ids), _u32(ids), _u32(offsets)
        ids_ptr = np.ascontiguousarray(ids_ptr, dtype=np.uint64)
        off_ptr = np.ascontiguousarray(off_ptr, dtype=np.uint64)
        offset_index = np.ascontiguousarray(offset_index, dtype=np.uint64)
        self._ck(self.L.tsgpu_terms_load_csr(self.h, field_id, term_ids.size, _vp(term_ids), _vp(ids_ptr), _vp(ids),
                                             _vp(offset_index), _vp(off_ptr), _vp(offsets)))

    def column_set(self, column_id, values, present=None):
        v = np.ascontiguousarray(values, dtype=np.int64)
        p = None if present is None else np.ascontiguousarray(present, dtype=np.uint8)
        self._ck(self.L.tsgpu_column_set(self.h, column_id, _vp(v), _vp(p) if p is not None else None, v.size, B.MEM_HOST))

    def column_set_device(self, column_id, data_ptr, n):
        self._ck(self.L.tsgpu_column_set(self.h, column_id, C.c_void_p(data_ptr), None, n, B.MEM_DEVICE))

    def set_num_docs(self, n):
        self._ck(self.L.tsgpu_set_num_docs(self.h, n))

    def commit(self):
        self._ck(self.L.tsgpu_commit(self.h))

    def term_num_ids(self, field_id, term_id):
        return self.L.tsgpu_term_num_ids(self.h, field_id, term_id)

    def term_download(self, field_id, term_id):
        n = self.term_num_ids(field_id, term_id)
        no = C.c_uint32(0)
        self._ck(self.L.tsgpu_term_download(self.h, field_id, term_id, None, None, None, C.byref(no)))
        ids, oi, off = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(max(no.value, 1), np.uint32)
        self._ck(self.L.tsgpu_term_download(self.h, field_id, term_id, _vp(ids), _vp(oi), off.ctypes.data_as(C.c_void_p), C.byref(no)))
        return ids, oi, off[:no.value]

    def device_bytes(self):
        return self.L.tsgpu_device_bytes(self.h)

    def set_option(self, name, value):
        self._ck(self.L.tsgpu_set_option(self.h, name.encode(), int(value)))

    def counter(self, name):
        v = C.c_uint64(0)
        self._ck(self.L.tsgpu_get_counter(self.h, name.encode(), C.byref(v)))
        return int(v.value)

    def set_stream(self, stream_ptr):
        self._ck(self.L.tsgpu_set_stream(self.h, C.c_void_p(stream_ptr)))

    # ---- keyword search (seam B1) ----
    def keyword_search_batch(self, queries, k_stride=250, hits=None):
        arr = make_query_array(queries)
        n = len(arr)
        hits = hits or Hits(n, k_stride)
        hs = hits.c_struct()
        self._ck(self.L.tsgpu_keyword_search_batch(self.h, C.cast(arr, C.c_void_p), n, C.byref(hs)))
        return hits

    def wildcard_search_batch(self, queries, k_stride=250, hits=None):
        """q = "*": rank the filter ids (all documents without a filter) by the sort keys (Index::search_wildcard)"""
        arr = make_query_array(queries)
        n = len(arr)
        hits = hits or Hits(n, k_stride)
        hs = hits.c_struct()
        self._ck(self.L.tsgpu_wildcard_search_batch(self.h, C.cast(arr, C.c_void_p), n, C.byref(hs)))
        return hits

    def keyword_search_candidates_batch(self, groups, k_stride=250, want_found=True):
        """groups: per user query the list of candidate-token combinations (KwQuery, pass order) — Index::search_all_candidates.
        Returns (Hits [n_groups], query_index [n_groups, k_stride] u32, found [n_groups] u64 or None)."""
        flat = [q for g in groups for q in g]
        begin = np.zeros(len(groups) + 1, np.uint32)
        begin[1:] = np.cumsum([len(g) for g in groups])
        arr = make_query_array(flat) if flat else None
        hits = Hits(len(groups), k_stride)
        hs = hits.c_struct()
        qi = np.zeros((len(groups), k_stride), np.uint32)
        found = np.zeros(len(groups), np.uint64) if want_found else None
        self._ck(self.L.tsgpu_keyword_search_candidates_batch(self.h, C.cast(arr, C.c_void_p) if flat else None, _vp(begin), len(groups), C.byref(hs),
                                                              _vp(qi), _vp(found) if want_found else None))
        return hits, qi, found

    def keyword_search_candidates_batch_raw(self, arr, begin, n_groups, hs, qi, found):
        """prebuilt ctypes query array (all combinations, pass order), begin[n_groups + 1], tsgpu_hits struct, query_index / found arrays (bench loop)"""
        self._ck(self.L.tsgpu_keyword_search_candidates_batch(self.h, C.cast(arr, C.c_void_p), _vp(begin), n_groups, C.byref(hs), _vp(qi), _vp(found) if found is not None else None))

    def candidates_result_ids(self, group):
        n = self.L.tsgpu_candidates_result_ids(self.h, group, None, 0)
        out = np.zeros(max(n, 1), np.uint32)
        if n:
            self.L.tsgpu_candidates_result_ids(self.h, group, _vp(out), n)
        return out[:n]

    def keyword_search_batch_raw(self, arr, n, hs):
        """prebuilt ctypes query array + tsgpu_hits struct (device or host outputs); no allocation (bench loop)"""
        self._ck(self.L.tsgpu_keyword_search_batch(self.h, C.cast(arr, C.c_void_p), n, C.byref(hs)))

    def keyword_search_batch_ids(self, queries, k_stride=250, hits=None):
        """search + the matched ids of every query as a list that belongs to this call (safe with concurrent callers)"""
        arr = make_query_array(queries)
        n = len(arr)
        hits = hits or Hits(n, k_stride)
        hs = hits.c_struct()
        lists = C.c_void_p()
        self._ck(self.L.tsgpu_keyword_search_batch_ids(self.h, C.cast(arr, C.c_void_p), n, C.byref(hs), C.byref(lists)))
        try:
            ids = []
            for q in range(n):
                cnt = self.L.tsgpu_id_lists_count(lists, q)
                ids.append(np.ctypeslib.as_array(self.L.tsgpu_id_lists_ids(lists, q), shape=(cnt,)).copy() if cnt else np.zeros(0, np.uint32))
        finally:
            self.L.tsgpu_id_lists_free(lists)
        return hits, ids

    # ---- facet counting over matched ids (do_facets, hash-index branch) ----
    def keyword_search_grouped_batch(self, queries, groups, k_stride, g_stride=250, want_ids=False, want_registers=False):
        """tsgpu_keyword_search_grouped_batch: groups = [(group_limit, column, first_pass, group_missing_values, wildcard), ...] per query.
        Returns (Hits, GroupedHits[, id lists])."""
        arr = make_query_array(queries)
        n = len(arr)
        ga = (B.GroupByC * n)()
        for i, g in enumerate(groups):
            ga[i].group_limit, ga[i].column, ga[i].first_pass, ga[i].group_missing_values, ga[i].wildcard = int(g[0]), int(g[1]), int(g[2]), int(g[3]), int(g[4])
        h = Hits(n, k_stride)
        gh = GroupedHits(n, g_stride, want_registers)
        hs, gs = h.c_struct(), gh.c_struct()
        handle = C.c_void_p()
        self._ck(self.L.tsgpu_keyword_search_grouped_batch(self.h, arr, ga, n, C.byref(hs), C.byref(gs), C.byref(handle) if want_ids else None))
        if not want_ids:
            return h, gh
        lists = []
        try:
            for q in range(n):
                cnt = int(self.L.tsgpu_id_lists_count(handle, q))
                lists.append(np.ctypeslib.as_array(self.L.tsgpu_id_lists_ids(handle, q), shape=(cnt,)).copy() if cnt else np.zeros(0, np.uint32))
        finally:
            self.L.tsgpu_id_lists_free(handle)
        return h, gh, lists

    def keyword_search_grouped_candidates_batch(self, user_combos, groups, k_stride, g_stride=250, want_ids=False, want_registers=False):
        """tsgpu_keyword_search_grouped_candidates_batch: user_combos = [[KwQuery, ...], ...] (the candidate combinations of every user query, pass order),
        groups as in keyword_search_grouped_batch (one per user query). Returns (Hits, GroupedHits, query_index[n_user, k_stride][, id lists])."""
        flat = [q for combos in user_combos for q in combos]
        arr = make_query_array(flat)
        n = len(user_combos)
        begin = np.zeros(n + 1, np.uint32)
        begin[1:] = np.cumsum([len(c) for c in user_combos])
        ga = (B.GroupByC * n)()
        for i, g in enumerate(groups):
            ga[i].group_limit, ga[i].column, ga[i].first_pass, ga[i].group_missing_values, ga[i].wildcard = int(g[0]), int(g[1]), int(g[2]), int(g[3]), int(g[4])
        h = Hits(n, k_stride)
        gh = GroupedHits(n, g_stride, want_registers)
        qidx = np.zeros((n, k_stride), np.uint32)
        hs, gs = h.c_struct(), gh.c_struct()
        handle = C.c_void_p()
        self._ck(self.L.tsgpu_keyword_search_grouped_candidates_batch(self.h, arr, begin.ctypes.data, ga, n, C.byref(hs), C.byref(gs), qidx.ctypes.data,
                                                                      C.byref(handle) if want_ids else None))
        if not want_ids:
            return h, gh, qidx
        lists = []
        try:
            for q in range(n):
                cnt = int(self.L.tsgpu_id_lists_count(handle, q))
                lists.append(np.ctypeslib.as_array(self.L.tsgpu_id_lists_ids(handle, q), shape=(cnt,)).copy() if cnt else np.zeros(0, np.uint32))
        finally:
            self.L.tsgpu_id_lists_free(handle)
        return h, gh, qidx, lists

    def facet_set(self, field_id, doc_ptr, hashes):
        doc_ptr = np.ascontiguousarray(doc_ptr, dtype=np.uint64)
        hashes = _u32(hashes)
        self._ck(self.L.tsgpu_facet_set(self.h, field_id, _vp(doc_ptr), hashes.ctypes.data_as(C.c_void_p), doc_ptr.size - 1))

    def facet_count_batch(self, field_id, id_lists, cap=1024, sample_mod=1, allowed_hashes=None, group_column=None, group_missing_values=False):
        """id_lists: per query an ascending uint32 id array -> per query (hash, count, doc_id, array_pos) arrays in ascending hash order;
        group_column: the facets of a grouped search (count = number of groups the value was seen in)"""
        lists = [_u32(x) for x in id_lists]
        n = len(lists)
        ptrs = (C.c_void_p * n)(*[x.ctypes.data if x.size else None for x in lists])
        cnts = np.array([x.size for x in lists], np.uint64)
        out = B.FacetCountsC()
        h, c, d, p = (np.empty((n, cap), np.uint32) for _ in range(4))              # (only the first n_values[q] entries of a row are written and returned)
        nv = np.zeros(n, np.uint32)
        out.cap, out.hash, out.count, out.doc_id, out.array_pos, out.n_values = cap, h.ctypes.data, c.ctypes.data, d.ctypes.data, p.ctypes.data, nv.ctypes.data
        a = None if allowed_hashes is None else _u32(allowed_hashes)
        if group_column is None:
            self._ck(self.L.tsgpu_facet_count_batch(self.h, field_id, C.cast(ptrs, C.c_void_p), _vp(cnts), n, sample_mod,
                                                    _vp(a) if a is not None else None, a.size if a is not None else 0, C.byref(out)))
        else:
            self._ck(self.L.tsgpu_facet_count_grouped_batch(self.h, field_id, C.cast(ptrs, C.c_void_p), _vp(cnts), n, sample_mod,
                                                            _vp(a) if a is not None else None, a.size if a is not None else 0, int(group_column), int(group_missing_values), C.byref(out)))
        return [(h[q, :min(nv[q], cap)].copy(), c[q, :min(nv[q], cap)].copy(), d[q, :min(nv[q], cap)].copy(), p[q, :min(nv[q], cap)].copy(), int(nv[q])) for q in range(n)]

    def facet_range_count_batch(self, field_id, value_column, ranges, id_lists, sample_mod=1, group_column=None, group_missing_values=False):
        """ranges: [(upper, lower), ...] in ascending upper order (facet_range_map) -> counts uint32 [n_queries][n_ranges] (0 = not in result_map)"""
        lists = [_u32(x) for x in id_lists]
        n = len(lists)
        ptrs = (C.c_void_p * max(n, 1))(*[x.ctypes.data if x.size else None for x in lists])
        cnts = np.array([x.size for x in lists], np.uint64)
        up = np.ascontiguousarray([r[0] for r in ranges], dtype=np.int64)
        lo = np.ascontiguousarray([r[1] for r in ranges], dtype=np.int64)
        counts = np.zeros((n, len(ranges)), np.uint32)
        self._ck(self.L.tsgpu_facet_range_count_batch(self.h, field_id, int(value_column), _vp(up), _vp(lo), len(ranges), C.cast(ptrs, C.c_void_p), _vp(cnts), n, sample_mod,
                                                      B.NO_COLUMN if group_column is None else int(group_column), int(group_missing_values), _vp(counts)))
        return counts

    def facet_stats_batch(self, field_id, value_type, id_lists, sample_mod=1, int64_map=None):
        """-> per query (fvmin, fvmax, fvsum, fvcount, sum_exact); int64_map = (sorted hashes uint32[], values int64[]) for int64 fields"""
        lists = [_u32(x) for x in id_lists]
        n = len(lists)
        ptrs = (C.c_void_p * n)(*[x.ctypes.data if x.size else None for x in lists])
        cnts = np.array([x.size for x in lists], np.uint64)
        out = (B.FacetStatsC * n)()
        mh = mv = None
        if int64_map is not None:
            mh, mv = _u32(int64_map[0]), np.ascontiguousarray(int64_map[1], dtype=np.int64)
        self._ck(self.L.tsgpu_facet_stats_batch(self.h, field_id, value_type, C.cast(ptrs, C.c_void_p), _vp(cnts), n, sample_mod,
                                                _vp(mh) if mh is not None else None, _vp(mv) if mv is not None else None, mh.size if mh is not None else 0, C.cast(out, C.c_void_p)))
        return [(o.fvmin, o.fvmax, o.fvsum, int(o.fvcount), int(o.sum_exact)) for o in out]

    def facet_value_set(self, field_id, value_ptr, seq_ids, total_counts):
        value_ptr = np.ascontiguousarray(value_ptr, dtype=np.uint64)
        seq_ids, total_counts = _u32(seq_ids), _u32(total_counts)
        self._ck(self.L.tsgpu_facet_value_set(self.h, field_id, _vp(value_ptr), _vp(seq_ids) if seq_ids.size else None, _vp(total_counts) if total_counts.size else None, value_ptr.size - 1))

    def facet_value_count_batch(self, field_id, id_lists, max_facets, wildcard_no_filter=False, estimate=False, sample_interval=1, order=None, cap=None):
        """-> per query (value_index[], count[], doc_id[]) of the first max_facets values (visiting order) with a non-zero count"""
        lists = [_u32(x) for x in id_lists]
        n = len(lists)
        cap = cap or max(int(max_facets), 1)
        ptrs = (C.c_void_p * n)(*[x.ctypes.data if x.size else None for x in lists])
        cnts = np.array([x.size for x in lists], np.uint64)
        out = B.FacetValueCountsC()
        v, c, d, nf = np.zeros((n, cap), np.uint32), np.zeros((n, cap), np.uint32), np.zeros((n, cap), np.uint32), np.zeros(n, np.uint32)
        out.cap, out.value_index, out.count, out.doc_id, out.n_found = cap, v.ctypes.data, c.ctypes.data, d.ctypes.data, nf.ctypes.data
        o = None if order is None else _u32(order)
        self._ck(self.L.tsgpu_facet_value_count_batch(self.h, field_id, C.cast(ptrs, C.c_void_p), _vp(cnts), n, int(max_facets), int(wildcard_no_filter), int(estimate),
                                                      int(sample_interval), _vp(o) if o is not None else None, C.byref(out)))
        return [(v[q, :min(nf[q], cap)].copy(), c[q, :min(nf[q], cap)].copy(), d[q, :min(nf[q], cap)].copy()) for q in range(n)]

    def keep_result_ids(self, keep=True):
        self._ck(self.L.tsgpu_keep_result_ids(self.h, int(keep)))

    def result_ids(self, q):
        n = self.L.tsgpu_result_ids(self.h, q, None, 0)
        out = np.zeros(max(n, 1), np.uint32)
        if n:
            self.L.tsgpu_result_ids(self.h, q, _vp(out), n)
        return out[:n]

    def timings(self):
        t = B.TimingsC()
        self._ck(self.L.tsgpu_last_timings(self.h, C.byref(t)))
        return t

    def aux_timings(self):
        """tsgpu_last_aux_timings: kernel time / algorithmic bytes of the last group_by batch and the last facet-count batch"""
        t = B.AuxTimingsC()
        self._ck(self.L.tsgpu_last_aux_timings(self.h, C.byref(t)))
        return t

    def kw_touched(self):
        """bytes the find kernel counted itself during the last keyword batch run under option kw_count_touched (measurement)"""
        t = B.KwTouchedC()
        self._ck(self.L.tsgpu_kw_last_touched(self.h, C.byref(t)))
        return {n: int(getattr(t, n)) for n, _ in t._fields_}

    def kw_lists_footprint(self, field_ids, term_ids):
        """what the DISTINCT posting lists of the (field, term) pairs occupy in the mirror (measurement)"""
        f = np.ascontiguousarray(field_ids, np.uint32)
        t = np.ascontiguousarray(term_ids, np.uint32)
        out = B.KwFootprintC()
        self._ck(self.L.tsgpu_kw_lists_footprint(self.h, f.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), int(t.size), C.byref(out)))
        return {n: int(getattr(out, n)) for n, _ in out._fields_}

    # ---- vector index (seam B2) ----
    def vec_create(self, field_id, dim, metric=B.METRIC_IP, capacity_hint=0):
        self._ck(self.L.tsgpu_vec_create(self.h, field_id, dim, metric, capacity_hint))
        self.vec_dim[field_id] = dim

    def vec_upsert(self, field_id, labels, data):
        labels = np.ascontiguousarray(labels, dtype=np.uint64)
        data = np.ascontiguousarray(data, dtype=np.float32).reshape(labels.size, self.vec_dim[field_id])
        self._ck(self.L.tsgpu_vec_upsert(self.h, field_id, _vp(labels), _vp(data), labels.size, B.MEM_HOST))

    def vec_upsert_device(self, field_id, labels_ptr, data_ptr, n):
        self._ck(self.L.tsgpu_vec_upsert(self.h, field_id, C.c_void_p(labels_ptr), C.c_void_p(data_ptr), n, B.MEM_DEVICE))

    def vec_delete(self, field_id, label):
        self._ck(self.L.tsgpu_vec_delete(self.h, field_id, label))

    def vec_get(self, field_id, label):
        out = np.zeros(self.vec_dim[field_id], np.float32)
        rc = self.L.tsgpu_vec_get(self.h, field_id, label, _vp(out))
        if rc == B.ERR_NOT_FOUND:
            return None
        self._ck(rc)
        return out

    def vec_count(self, field_id):
        return self.L.tsgpu_vec_count(self.h, field_id)

    def vec_knn_batch(self, field_id, Q, k, allow_ids=None, excluded_ids=None):
        Q = np.ascontiguousarray(Q, dtype=np.float32).reshape(-1, self.vec_dim[field_id])
        n = Q.shape[0]
        dist = np.zeros((n, k), np.float32)
        lab = np.zeros((n, k), np.uint64)
        cnt = np.zeros(n, np.uint32)
        a = None if allow_ids is None else _u32(allow_ids)
        e = None if excluded_ids is None else _u32(excluded_ids)
        self._ck(self.L.tsgpu_vec_knn_batch(self.h, field_id, _vp(Q), B.MEM_HOST, n, k,
                                            _vp(a) if a is not None else None, a.size if a is not None else 0,
                                            _vp(e) if e is not None else None, e.size if e is not None else 0,
                                            _vp(dist), _vp(lab), _vp(cnt), B.MEM_HOST))
        return dist, lab, cnt

    def vec_knn_batch_raw(self, field_id, q_ptr, mem_q, n, k, dist_ptr, lab_ptr, cnt_ptr, mem_out):
        self._ck(self.L.tsgpu_vec_knn_batch(self.h, field_id, C.c_void_p(q_ptr), mem_q, n, k, None, 0, None, 0,
                                            C.c_void_p(dist_ptr), C.c_void_p(lab_ptr), C.c_void_p(cnt_ptr), mem_out))

    def vec_hnsw_load(self, field_id, graph):
        """graph: dict(M, maxlevel, enterpoint, link0[n, 1+2M], upper_ptr[n+1], upper_links[n_upper, 1+M]) — hnswlib's link lists"""
        l0 = np.ascontiguousarray(graph["link0"], dtype=np.uint32)
        up = np.ascontiguousarray(graph["upper_ptr"], dtype=np.uint64)
        ul = np.ascontiguousarray(graph["upper_links"], dtype=np.uint32)
        self._ck(self.L.tsgpu_vec_hnsw_load(self.h, field_id, int(graph["M"]), int(graph["maxlevel"]), int(graph["enterpoint"]), _vp(l0), _vp(up),
                                            _vp(ul) if ul.size else None, l0.shape[0]))

    def vec_hnsw_enable(self, field_id, M=16, ef_construction=200, seed=100, threads=1):
        """build the field's HNSW graph inside the library (hnswlib's addPoint on every new label; include/index.h:365-367 defaults)"""
        self._ck(self.L.tsgpu_vec_hnsw_enable(self.h, field_id, M, ef_construction, seed, threads))

    def vec_hnsw_build(self, field_id, M=16, ef_construction=200, seed=100, threads=1, seed_min=0, max_batch=0):
        """tsgpu_vec_hnsw_build: the graph over the rows the field holds now, built in batches on the device -> dict of tsgpu_hnsw_build_info"""
        bi = B.HnswBuildInfoC()
        self._ck(self.L.tsgpu_vec_hnsw_build(self.h, field_id, M, ef_construction, seed, threads, seed_min, max_batch, C.byref(bi)))
        return {n: getattr(bi, n) for n, _ in bi._fields_}

    def vec_hnsw_export(self, field_id):
        """-> dict(n, maxlevel, enterpoint, M, levels[n], link0[n, 1+2M], upper_ptr[n+1], upper_links[n_upper, 1+M]) (the oracle's hnsw_export keys)"""
        info = np.zeros(4, np.int32)
        nu = C.c_uint64(0)
        self._ck(self.L.tsgpu_vec_hnsw_export(self.h, field_id, _vp(info), None, None, None, None, C.byref(nu)))
        n, M, n_upper = int(info[0]), int(info[3]), int(nu.value)
        levels = np.zeros(n, np.uint32); link0 = np.zeros((n, 1 + 2 * M), np.uint32)
        upper_ptr = np.zeros(n + 1, np.uint64); upper = np.zeros((max(n_upper, 1), 1 + M), np.uint32)
        self._ck(self.L.tsgpu_vec_hnsw_export(self.h, field_id, _vp(info), _vp(levels), _vp(link0), _vp(upper_ptr), _vp(upper), C.byref(nu)))
        return dict(n=n, maxlevel=int(info[1]), enterpoint=int(np.uint32(info[2])), M=M, levels=levels, link0=link0, upper_ptr=upper_ptr, upper_links=upper[:n_upper])

    def vec_hnsw_search_batch(self, field_id, Q, k, ef, allow_ids=None, excluded_ids=None, functor_present=True):
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        n = Q.shape[0]
        dist = np.zeros((n, k), np.float32); lab = np.zeros((n, k), np.uint64); cnt = np.zeros(n, np.uint32)
        a = None if allow_ids is None else _u32(allow_ids)
        e = None if excluded_ids is None else _u32(excluded_ids)
        self._ck(self.L.tsgpu_vec_hnsw_search_batch(self.h, field_id, _vp(Q), B.MEM_HOST, n, k, ef, int(functor_present),
                                                    _vp(a) if a is not None else None, a.size if a is not None else 0,
                                                    _vp(e) if e is not None else None, e.size if e is not None else 0,
                                                    _vp(dist), _vp(lab), _vp(cnt), B.MEM_HOST))
        return dist, lab, cnt

    def vec_hnsw_search_batch_raw(self, field_id, q_ptr, mem_q, n, k, ef, dist_ptr, lab_ptr, cnt_ptr, mem_out, functor_present=True):
        self._ck(self.L.tsgpu_vec_hnsw_search_batch(self.h, field_id, C.c_void_p(q_ptr), mem_q, n, k, ef, int(functor_present), None, 0, None, 0,
                                                    C.c_void_p(dist_ptr), C.c_void_p(lab_ptr), C.c_void_p(cnt_ptr), mem_out))

    def vec_distances(self, field_id, q, labels):
        q = np.ascontiguousarray(q, dtype=np.float32)
        labels = np.ascontiguousarray(labels, dtype=np.uint64)
        out = np.zeros(labels.size, np.float32)
        self._ck(self.L.tsgpu_vec_distances(self.h, field_id, _vp(q), _vp(labels), labels.size, _vp(out)))
        return out

    def vector_search_batch(self, field_id, Q, k=0, fetch_size=10, distance_threshold=B.FLT_MAX,
                            sort=((B.SORT_VECTOR_DISTANCE, -1, 0), (B.SORT_SEQ_ID, 1, 0)), topster_size=0, k_stride=250,
                            filter_ids=None, excluded_ids=None, flat_search_cutoff=0, query_doc=None, want_ids=False):
        """the vector branch of Index::search (src/index.cpp:3645-3732). filter_ids = what filter_by matched (None = no filter_by); fewer
        of them than flat_search_cutoff -> the FLAT branch (every filter id through the Topster, num_matched = found). query_doc = X of
        `vec:([], id: X)` (Q = X's stored vector). want_ids: also return all_result_ids per query -> (hits, [ids])."""
        Q = np.ascontiguousarray(Q, dtype=np.float32).reshape(-1, self.vec_dim[field_id])
        p = B.VecQueryC()
        p.k, p.fetch_size, p.distance_threshold, p.n_sort, p.topster_size = k, fetch_size, distance_threshold, len(sort), topster_size
        for i, s in enumerate(sort):
            p.sort[i].kind, p.sort[i].order, p.sort[i].column = s
        f = _u32(filter_ids) if filter_ids is not None else None
        e = _u32(excluded_ids) if excluded_ids is not None else None
        p.filter_by_provided = 1 if f is not None else 0
        p.filter_ids, p.n_filter = (f.ctypes.data if f is not None and f.size else None), (f.size if f is not None else 0)
        p.excluded_ids, p.n_excluded = (e.ctypes.data if e is not None and e.size else None), (e.size if e is not None else 0)
        p.flat_search_cutoff = int(flat_search_cutoff)
        p.query_doc_given, p.query_seq_id = (1, int(query_doc)) if query_doc is not None else (0, 0)
        hits = Hits(Q.shape[0], k_stride)
        hs = hits.c_struct()
        if not want_ids:
            self._ck(self.L.tsgpu_vector_search_batch(self.h, field_id, C.byref(p), _vp(Q), B.MEM_HOST, Q.shape[0], C.byref(hs)))
            return hits
        lists = C.c_void_p()
        self._ck(self.L.tsgpu_vector_search_batch_ids(self.h, field_id, C.byref(p), _vp(Q), B.MEM_HOST, Q.shape[0], C.byref(hs), C.byref(lists)))
        try:
            ids = []
            for q in range(Q.shape[0]):
                cnt = self.L.tsgpu_id_lists_count(lists, q)
                ids.append(np.ctypeslib.as_array(self.L.tsgpu_id_lists_ids(lists, q), shape=(cnt,)).copy() if cnt else np.zeros(0, np.uint32))
        finally:
            self.L.tsgpu_id_lists_free(lists)
        return hits, ids

    def keyword_aux_scores(self, queries, item_query, item_seq_id):
        """compute_aux_scores' text half: text_match of given documents for the queries' tokens -> int64[n_items]"""
        arr = make_query_array(queries)
        iq, ii = _u32(item_query), _u32(item_seq_id)
        out = np.zeros(iq.size, np.int64)
        self._ck(self.L.tsgpu_keyword_aux_scores(self.h, C.cast(arr, C.c_void_p), len(arr), _vp(iq), _vp(ii), iq.size, _vp(out)))
        return out

    def hybrid_search_batch(self, queries, field_id, Q, k=0, fetch_size=10, alpha=0.3, distance_threshold=B.FLT_MAX, k_stride=250, rerank=False):
        arr = make_query_array(queries)
        Q = np.ascontiguousarray(Q, dtype=np.float32).reshape(-1, self.vec_dim[field_id])
        p = B.HybridParamsC()
        p.k, p.fetch_size, p.alpha, p.distance_threshold = k, fetch_size, alpha, distance_threshold
        p.rerank_hybrid_matches = 1 if rerank else 0
        hits = Hits(len(arr), k_stride)
        hs = hits.c_struct()
        self._ck(self.L.tsgpu_hybrid_search_batch(self.h, C.cast(arr, C.c_void_p), field_id, C.byref(p), _vp(Q), B.MEM_HOST, len(arr), C.byref(hs)))
        return hits

    def hybrid_fuse_batch(self, queries, kw_hits, knn_dist, knn_labels, knn_cnt, metric, k=0, fetch_size=10, alpha=0.3,
                          distance_threshold=B.FLT_MAX, k_stride=250):
        """fusion step alone on already-computed (e.g. shard-merged) results; kw_hits: Hits (host)"""
        arr = make_query_array(queries)
        p = B.HybridParamsC()
        p.k, p.fetch_size, p.alpha, p.distance_threshold = k, fetch_size, alpha, distance_threshold
        p.rerank_hybrid_matches = 0
        d = np.ascontiguousarray(knn_dist, dtype=np.float32)
        l = np.ascontiguousarray(knn_labels, dtype=np.uint64)
        c = np.ascontiguousarray(knn_cnt, dtype=np.uint32)
        hits = Hits(len(arr), k_stride)
        hs, ks = hits.c_struct(), kw_hits.c_struct()
        self._ck(self.L.tsgpu_hybrid_fuse_batch(self.h, C.cast(arr, C.c_void_p), C.byref(p), metric, C.byref(ks), _vp(d), _vp(l), _vp(c),
                                                d.shape[1], len(arr), C.byref(hs)))
        return hits


class GpuGroup:
    """tsgpu_group: doc-range shards behind the C-ABI (include/tsgpu.h "multi-GPU group"). Local form: GpuGroup(members=[GpuIndex, ...],
    transport=B.XCHG_RCCL | B.XCHG_COPY); rank form (one process per GPU): GpuGroup.join(index, unique_id, rank, n_ranks)."""

    def __init__(self, members=None, transport=B.XCHG_RCCL, _handle=None, _lib=None, _members=None):
        if _handle is not None:
            self.h, self.L, self.members = _handle, _lib, _members
            return
        self.members = list(members)
        self.L = self.members[0].L
        arr = (C.c_void_p * len(self.members))(*[m.h for m in self.members])
        h = C.c_void_p()
        B.check(self.L, self.L.tsgpu_group_create_local(C.cast(arr, C.c_void_p), len(self.members), transport, C.byref(h)))
        self.h = h

    @staticmethod
    def unique_id(L):
        buf = (C.c_uint8 * 128)()
        B.check(L, L.tsgpu_group_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    @classmethod
    def join(cls, index, unique_id, rank, n_ranks):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        h = C.c_void_p()
        B.check(index.L, index.L.tsgpu_group_create_rank(index.h, C.cast(buf, C.c_void_p), rank, n_ranks, C.byref(h)))
        return cls(_handle=h, _lib=index.L, _members=[index])

    @classmethod
    def join_host(cls, index, rank, n_ranks, all_gather, all_to_all):
        """rank form over the CALLER's collectives on host memory (tsgpu_group_create_rank_host, TSGPU_XCHG_HOST): all_gather(send, recv,
        bytes) / all_to_all(send, recv, bytes) get numpy uint8 views of the library's pinned staging buffers (send: bytes resp. n_ranks x
        bytes; recv: n_ranks x bytes) and follow ncclAllGather / ncclAllToAll. typesense_amd.hostcoll.torch_collectives() backs them with
        torch.distributed (gloo on CPU tensors)."""
        def wrap(fn, send_slices):
            def cb(_user, send, recv, nbytes):
                try:
                    ns = nbytes * (n_ranks if send_slices else 1)
                    sv = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(ns,)) if ns else np.zeros(0, np.uint8)
                    rv = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(nbytes * n_ranks,)) if nbytes else np.zeros(0, np.uint8)
                    fn(sv, rv, int(nbytes))
                    return 0
                except Exception as e:       # an exception must not unwind through the C frames: report it as a failed collective
                    import sys, traceback
                    traceback.print_exc(file=sys.stderr)
                    return 1
            return B.HOST_COLLECTIVE_FN(cb)
        coll = B.HostCollectivesC(None, wrap(all_gather, False), wrap(all_to_all, True))
        h = C.c_void_p()
        B.check(index.L, index.L.tsgpu_group_create_rank_host(index.h, C.byref(coll), rank, n_ranks, C.byref(h)))
        grp = cls(_handle=h, _lib=index.L, _members=[index])
        grp._keepalive = coll               # the CFUNCTYPE thunks must outlive the group
        return grp

    def close(self):
        if getattr(self, "h", None):
            self.L.tsgpu_group_destroy(self.h)
            self.h = None

    def size(self):
        return int(self.L.tsgpu_group_size(self.h))

    def set_option(self, name, value):
        B.check(self.L, self.L.tsgpu_group_set_option(self.h, name.encode(), int(value)))

    def keyword_search_batch(self, queries, k, k_stride=None):
        arr = make_query_array(queries)
        hits = Hits(len(arr), k_stride or k)
        hs = hits.c_struct()
        B.check(self.L, self.L.tsgpu_group_keyword_search_batch(self.h, C.cast(arr, C.c_void_p), len(arr), k, C.byref(hs)))
        return hits

    def keyword_search_batch_raw(self, arr, n, k, hs):
        B.check(self.L, self.L.tsgpu_group_keyword_search_batch(self.h, C.cast(arr, C.c_void_p), n, k, C.byref(hs)))

    def wildcard_search_batch(self, queries, k, k_stride=None):
        """tsgpu_group_wildcard_search_batch: q = * over the shards (every member ranks the ids of its doc range)"""
        arr = make_query_array(queries)
        hits = Hits(len(arr), k_stride or k)
        hs = hits.c_struct()
        B.check(self.L, self.L.tsgpu_group_wildcard_search_batch(self.h, C.cast(arr, C.c_void_p), len(arr), k, C.byref(hs)))
        return hits

    def keyword_search_grouped_batch(self, queries, groups, k_stride, g_stride=250, want_registers=False, want_totals=False):
        """tsgpu_group_keyword_search_grouped_batch: group_by over the shards (keyed exchange in two rounds); groups as in GpuIndex.keyword_search_grouped_batch.
        groups_total is not computed across shards (want_totals=True asks for it anyway: 501 unless the group is in replicas form)"""
        arr = make_query_array(queries)
        n = len(arr)
        ga = (B.GroupByC * max(n, 1))()
        for i, g in enumerate(groups):
            ga[i].group_limit, ga[i].column, ga[i].first_pass, ga[i].group_missing_values, ga[i].wildcard = int(g[0]), int(g[1]), int(g[2]), int(g[3]), int(g[4])
        h = Hits(n, k_stride)
        gh = GroupedHits(n, g_stride, want_registers, totals=want_totals)
        hs, gs = h.c_struct(), gh.c_struct()
        B.check(self.L, self.L.tsgpu_group_keyword_search_grouped_batch(self.h, C.cast(arr, C.c_void_p), ga, n, C.byref(hs), C.byref(gs)))
        return h, gh

    def keyword_search_grouped_candidates_batch(self, user_combos, groups, k_stride, g_stride=250, want_registers=False):
        """tsgpu_group_keyword_search_grouped_candidates_batch: GpuIndex.keyword_search_grouped_candidates_batch over the shards. Returns (Hits, GroupedHits, query_index)"""
        flat = [q for combos in user_combos for q in combos]
        arr = make_query_array(flat)
        n = len(user_combos)
        begin = np.zeros(n + 1, np.uint32)
        begin[1:] = np.cumsum([len(c) for c in user_combos])
        ga = (B.GroupByC * max(n, 1))()
        for i, g in enumerate(groups):
            ga[i].group_limit, ga[i].column, ga[i].first_pass, ga[i].group_missing_values, ga[i].wildcard = int(g[0]), int(g[1]), int(g[2]), int(g[3]), int(g[4])
        h = Hits(n, k_stride)
        gh = GroupedHits(n, g_stride, want_registers, totals=False)
        qidx = np.zeros((n, k_stride), np.uint32)
        hs, gs = h.c_struct(), gh.c_struct()
        B.check(self.L, self.L.tsgpu_group_keyword_search_grouped_candidates_batch(self.h, C.cast(arr, C.c_void_p) if flat else None, begin.ctypes.data, ga, n, C.byref(hs), C.byref(gs), qidx.ctypes.data))
        return h, gh, qidx

    def facet_count_batch(self, field_id, id_lists, cap=1024, sample_mod=1, allowed_hashes=None):
        """tsgpu_group_facet_count_batch: GpuIndex.facet_count_batch over the shards (GLOBAL ascending id lists)"""
        lists = [_u32(x) for x in id_lists]
        n = len(lists)
        ptrs = (C.c_void_p * max(n, 1))(*[x.ctypes.data if x.size else None for x in lists])
        cnts = np.array([x.size for x in lists], np.uint64)
        out = B.FacetCountsC()
        h, c, d, p = (np.empty((n, cap), np.uint32) for _ in range(4))
        nv = np.zeros(n, np.uint32)
        out.cap, out.hash, out.count, out.doc_id, out.array_pos, out.n_values = cap, h.ctypes.data, c.ctypes.data, d.ctypes.data, p.ctypes.data, nv.ctypes.data
        a = None if allowed_hashes is None else _u32(allowed_hashes)
        B.check(self.L, self.L.tsgpu_group_facet_count_batch(self.h, field_id, C.cast(ptrs, C.c_void_p), _vp(cnts), n, sample_mod,
                                                             _vp(a) if a is not None else None, a.size if a is not None else 0, C.byref(out)))
        return [(h[q, :min(nv[q], cap)].copy(), c[q, :min(nv[q], cap)].copy(), d[q, :min(nv[q], cap)].copy(), p[q, :min(nv[q], cap)].copy(), int(nv[q])) for q in range(n)]

    def facet_range_count_batch(self, field_id, value_column, ranges, id_lists, sample_mod=1):
        """tsgpu_group_facet_range_count_batch: ranges = [(upper, lower), ...] in ascending upper order -> counts uint32 [n_queries][n_ranges]"""
        lists = [_u32(x) for x in id_lists]
        n = len(lists)
        ptrs = (C.c_void_p * max(n, 1))(*[x.ctypes.data if x.size else None for x in lists])
        cnts = np.array([x.size for x in lists], np.uint64)
        up = np.ascontiguousarray([r[0] for r in ranges], dtype=np.int64)
        lo = np.ascontiguousarray([r[1] for r in ranges], dtype=np.int64)
        counts = np.zeros((n, len(ranges)), np.uint32)
        B.check(self.L, self.L.tsgpu_group_facet_range_count_batch(self.h, field_id, value_column, _vp(up), _vp(lo), len(ranges), C.cast(ptrs, C.c_void_p), _vp(cnts), n, sample_mod, _vp(counts)))
        return counts

    def facet_stats_batch(self, field_id, value_type, id_lists, sample_mod=1, int64_map=None):
        """tsgpu_group_facet_stats_batch -> per query (fvmin, fvmax, fvsum, fvcount, sum_exact)"""
        lists = [_u32(x) for x in id_lists]
        n = len(lists)
        ptrs = (C.c_void_p * max(n, 1))(*[x.ctypes.data if x.size else None for x in lists])
        cnts = np.array([x.size for x in lists], np.uint64)
        out = (B.FacetStatsC * n)()
        mh = mv = None
        if int64_map is not None:
            mh, mv = _u32(int64_map[0]), np.ascontiguousarray(int64_map[1], dtype=np.int64)
        B.check(self.L, self.L.tsgpu_group_facet_stats_batch(self.h, field_id, value_type, C.cast(ptrs, C.c_void_p), _vp(cnts), n, sample_mod,
                                                             _vp(mh) if mh is not None else None, _vp(mv) if mv is not None else None, mh.size if mh is not None else 0, C.cast(out, C.c_void_p)))
        return [(o.fvmin, o.fvmax, o.fvsum, int(o.fvcount), int(o.sum_exact)) for o in out]

    def keyword_search_candidates_batch(self, groups, k, k_stride=None, want_found=True):
        """tsgpu_group_keyword_search_candidates_batch: groups = per user query the list of candidate-token combinations (KwQuery, pass order).
        Returns (Hits [n_groups], query_index [n_groups, k_stride] u32, found [n_groups] u64 or None) — Index::search_all_candidates over the shards."""
        flat = [q for g in groups for q in g]
        begin = np.zeros(len(groups) + 1, np.uint32)
        begin[1:] = np.cumsum([len(g) for g in groups])
        arr = make_query_array(flat) if flat else None
        hits = Hits(len(groups), k_stride or k)
        hs = hits.c_struct()
        qi = np.zeros((len(groups), k_stride or k), np.uint32)
        found = np.zeros(len(groups), np.uint64) if want_found else None
        B.check(self.L, self.L.tsgpu_group_keyword_search_candidates_batch(self.h, C.cast(arr, C.c_void_p) if flat else None, _vp(begin), len(groups), k, C.byref(hs),
                                                                            _vp(qi), _vp(found) if want_found else None))
        return hits, qi, found

    def vec_knn_batch(self, field_id, Q, k, allow_ids=None, excluded_ids=None):
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        n = Q.shape[0]
        dist = np.zeros((n, k), np.float32); lab = np.zeros((n, k), np.uint64); cnt = np.zeros(n, np.uint32)
        a = None if allow_ids is None else _u32(allow_ids)
        e = None if excluded_ids is None else _u32(excluded_ids)
        B.check(self.L, self.L.tsgpu_group_vec_knn_batch(self.h, field_id, _vp(Q), B.MEM_HOST, n, k, _vp(a) if a is not None else None, a.size if a is not None else 0,
                                                         _vp(e) if e is not None else None, e.size if e is not None else 0, _vp(dist), _vp(lab), _vp(cnt), B.MEM_HOST))
        return dist, lab, cnt

    def vec_knn_batch_raw(self, field_id, q_ptr, mem_q, n, k, dist_ptr, lab_ptr, cnt_ptr, mem_out):
        B.check(self.L, self.L.tsgpu_group_vec_knn_batch(self.h, field_id, C.c_void_p(q_ptr), mem_q, n, k, None, 0, None, 0,
                                                         C.c_void_p(dist_ptr), C.c_void_p(lab_ptr), C.c_void_p(cnt_ptr), mem_out))

    def hybrid_search_batch(self, queries, field_id, metric, Q, k=0, fetch_size=10, alpha=0.3, distance_threshold=B.FLT_MAX, k_stride=250, mem_q=B.MEM_HOST, q_ptr=None, dim=None,
                            rerank=False):
        arr = make_query_array(queries)
        p = B.HybridParamsC()
        p.k, p.fetch_size, p.alpha, p.distance_threshold, p.rerank_hybrid_matches = k, fetch_size, alpha, distance_threshold, int(bool(rerank))
        hits = Hits(len(arr), k_stride)
        hs = hits.c_struct()
        if q_ptr is None:
            Q = np.ascontiguousarray(Q, dtype=np.float32)
            q_ptr, dim = Q.ctypes.data, Q.shape[1]
        B.check(self.L, self.L.tsgpu_group_hybrid_search_batch(self.h, C.cast(arr, C.c_void_p), field_id, metric, C.byref(p), C.c_void_p(q_ptr), mem_q, dim, len(arr), C.byref(hs)))
        return hits

    def timings(self):
        t = B.GroupTimingsC()
        B.check(self.L, self.L.tsgpu_group_last_timings(self.h, C.byref(t)))
        return t
