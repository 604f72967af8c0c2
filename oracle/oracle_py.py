"""ORACLE — TEST INFRASTRUCTURE ONLY. ctypes binding over oracle/_build/liboracle.so (oracle_capi.cpp)."""
import ctypes as C
import os
import subprocess
import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_DIR, "_build", "liboracle.so")
_REF = os.path.join(_DIR, "_ref", "libref_match.so")

SORT_TEXT_MATCH, SORT_SEQ_ID, SORT_INT64_COLUMN, SORT_VECTOR_DISTANCE = 0, 1, 2, 3
METRIC_IP, METRIC_COSINE = 0, 1
MAX_SCORE, MAX_WEIGHT, SUM_SCORE = 0, 1, 2


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(_LIB) or not os.path.exists(os.path.join(_DIR, "_build", "golden_tests")):
        subprocess.check_call(["make", "-C", _DIR, "--no-print-directory"], stdout=subprocess.DEVNULL)
    return _LIB


class KwQuery(C.Structure):
    _fields_ = [("tokens", C.POINTER(C.c_uint32)), ("n_tokens", C.c_uint32),
                ("field_ids", C.POINTER(C.c_uint32)), ("field_weights", C.POINTER(C.c_int64)), ("n_fields", C.c_uint32),
                ("match_type", C.c_int32),
                ("prioritize_exact_match", C.c_int32), ("prioritize_token_position", C.c_int32),
                ("prioritize_num_matching_fields", C.c_int32),
                ("total_cost", C.c_uint32),
                ("sort_kind", C.c_int32 * 3), ("sort_column", C.c_int32 * 3), ("sort_order", C.c_int32 * 3), ("n_sort", C.c_uint32),
                ("fetch_size", C.c_uint32),
                ("excluded_ids", C.POINTER(C.c_uint32)), ("n_excluded", C.c_uint32),
                ("filter_ids", C.POINTER(C.c_uint32)), ("n_filter", C.c_uint32), ("topster_size", C.c_uint32),
                ("dropped_tokens", C.POINTER(C.c_uint32)), ("n_dropped", C.c_uint32),
                ("syn_orig_num_tokens", C.c_int32), ("orig_num_tokens", C.c_int32), ("is_synonym_query", C.c_int32), ("demote_synonym_match", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("cap", C.c_uint32), ("n", C.c_uint32),
                ("keys", C.POINTER(C.c_uint64)), ("scores", C.POINTER(C.c_int64)), ("text_match", C.POINTER(C.c_int64)),
                ("vector_distance", C.POINTER(C.c_float)), ("match_score_index", C.POINTER(C.c_int8)),
                ("num_keyword_matches", C.c_uint64), ("n_result_ids", C.c_uint64),
                ("result_ids", C.POINTER(C.c_uint32)), ("result_ids_cap", C.c_uint64), ("search_cutoff", C.c_int32)]


class Grouped(C.Structure):
    _fields_ = [("group_cap", C.c_uint32), ("kv_cap", C.c_uint32), ("n_groups", C.c_uint32),
                ("group_size", C.POINTER(C.c_uint32)), ("group_found", C.POINTER(C.c_uint32)), ("distinct_key", C.POINTER(C.c_uint64)),
                ("keys", C.POINTER(C.c_uint64)), ("scores", C.POINTER(C.c_int64)),
                ("groups_count", C.c_uint64), ("groups_exact", C.c_uint64), ("loglog", C.POINTER(C.c_uint8)),
                ("missing_ids", C.POINTER(C.c_uint32)), ("missing_cap", C.c_uint64), ("n_missing", C.c_uint64),
                ("num_keyword_matches", C.c_uint64), ("n_result_ids", C.c_uint64),
                ("result_ids", C.POINTER(C.c_uint32)), ("result_ids_cap", C.c_uint64)]


class GroupedHits:
    """Decoded Grouped: groups in the collector's order; group g = keys[begin[g]:begin[g+1]] (+ scores rows), its distinct key and found count."""
    def __init__(self, g, b):
        n = g.n_groups
        self.n_groups = n
        self.group_size = b["gsize"][:n].copy()
        self.group_found = b["gfound"][:n].copy()
        self.distinct_key = b["dkey"][:n].copy()
        self.begin = np.concatenate([[0], np.cumsum(self.group_size)]).astype(np.int64)
        tot = int(self.begin[-1])
        self.keys = b["keys"][:tot].copy()
        self.scores = b["scores"][:tot * 3].reshape(tot, 3).copy()
        self.groups_count, self.groups_exact = int(g.groups_count), int(g.groups_exact)
        self.loglog = b["loglog"].copy()
        self.missing_ids = b["missing"][:min(g.n_missing, g.missing_cap)].copy()
        self.num_keyword_matches = int(g.num_keyword_matches)
        self.result_ids = b["ids"][:min(g.n_result_ids, g.result_ids_cap)].copy()


def _alloc_grouped(group_cap, kv_cap, ids_cap=0):
    g = Grouped()
    b = dict(gsize=np.zeros(group_cap, np.uint32), gfound=np.zeros(group_cap, np.uint32), dkey=np.zeros(group_cap, np.uint64),
             keys=np.zeros(kv_cap, np.uint64), scores=np.zeros(kv_cap * 3, np.int64), loglog=np.zeros(16384, np.uint8),
             missing=np.zeros(max(ids_cap, 1), np.uint32), ids=np.zeros(max(ids_cap, 1), np.uint32))
    g.group_cap, g.kv_cap = group_cap, kv_cap
    g.group_size = b["gsize"].ctypes.data_as(C.POINTER(C.c_uint32)); g.group_found = b["gfound"].ctypes.data_as(C.POINTER(C.c_uint32))
    g.distinct_key = b["dkey"].ctypes.data_as(C.POINTER(C.c_uint64))
    g.keys = b["keys"].ctypes.data_as(C.POINTER(C.c_uint64)); g.scores = b["scores"].ctypes.data_as(C.POINTER(C.c_int64))
    g.loglog = b["loglog"].ctypes.data_as(C.POINTER(C.c_uint8))
    if ids_cap:
        g.missing_ids = b["missing"].ctypes.data_as(C.POINTER(C.c_uint32)); g.missing_cap = ids_cap
        g.result_ids = b["ids"].ctypes.data_as(C.POINTER(C.c_uint32)); g.result_ids_cap = ids_cap
    return g, b


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_set_num_docs.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_num_docs.restype = C.c_uint32
        L.orc_num_docs.argtypes = [C.c_void_p]
        L.orc_index_plain.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        L.orc_index_array.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_load_posting.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.orc_dump_posting.restype = C.c_uint32
        L.orc_dump_posting.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        L.orc_list_terms.restype = C.c_uint32
        L.orc_list_terms.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.orc_field_is_array.restype = C.c_int32
        L.orc_field_is_array.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_set_sort.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int64]
        L.orc_set_sort_dense.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.orc_vec_init.argtypes = [C.c_void_p, C.c_uint32, C.c_int32]
        L.orc_vec_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_vec_get.restype = C.c_int32
        L.orc_vec_get.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_ip_distance.restype = C.c_float
        L.orc_ip_distance.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_set_ip_lanes.restype = C.c_int32
        L.orc_set_ip_lanes.argtypes = [C.c_int32]
        L.orc_get_ip_lanes.restype = C.c_int32
        L.orc_flat_knn.restype = C.c_uint32
        L.orc_flat_knn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_search_keyword.argtypes = [C.c_void_p, C.POINTER(KwQuery), C.POINTER(Result)]
        L.orc_search_wildcard.argtypes = [C.c_void_p, C.POINTER(KwQuery), C.POINTER(Result)]
        L.orc_search_keyword_grouped.argtypes = [C.c_void_p, C.POINTER(KwQuery), C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_uint32, C.c_int32, C.POINTER(Grouped)]
        L.orc_search_candidates_grouped.argtypes = [C.c_void_p, C.POINTER(KwQuery), C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_uint32, C.c_int32, C.POINTER(Grouped), C.c_void_p]
        L.orc_group_topster_run.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Grouped)]
        L.orc_hash_wy.restype = C.c_uint64
        L.orc_hash_wy.argtypes = [C.c_char_p, C.c_uint64]
        L.orc_hash_combine.restype = C.c_uint64
        L.orc_hash_combine.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_loglog_of_keys.restype = C.c_uint64
        L.orc_loglog_of_keys.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_loglog_cardinality.restype = C.c_uint64
        L.orc_loglog_cardinality.argtypes = [C.c_void_p]
        L.orc_distinct_ids.restype = None
        L.orc_distinct_ids.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.orc_search_candidates.argtypes = [C.c_void_p, C.POINTER(KwQuery), C.c_uint32, C.POINTER(Result), C.c_void_p]
        L.orc_hnsw_build.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_hnsw_bulk_build.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_hnsw_bulk_build.restype = None
        L.orc_hnsw_free.argtypes = [C.c_void_p]
        L.orc_hnsw_mark_deleted.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_hnsw_add_new_rows.argtypes = [C.c_void_p]
        L.orc_hnsw_add_new_rows.restype = None
        L.orc_hnsw_add_point_replace.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_hnsw_add_point_replace.restype = C.c_int32
        L.orc_hnsw_labels.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_hnsw_labels.restype = None
        L.orc_hnsw_export.restype = C.c_uint64
        L.orc_hnsw_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_hnsw_import.restype = C.c_int32
        L.orc_hnsw_import.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_hnsw_search_batch.restype = None
        L.orc_hnsw_search_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_hnsw_search.restype = C.c_uint32
        L.orc_hnsw_search.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_search_vector.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(Result)]
        L.orc_search_vector2.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                         C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_uint32, C.c_void_p]
        L.orc_search_hybrid.argtypes = [C.c_void_p, C.POINTER(KwQuery), C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.POINTER(Result)]
        L.orc_search_hybrid_rerank.argtypes = [C.c_void_p, C.POINTER(KwQuery), C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_int32, C.POINTER(Result)]
        L.orc_facet_set.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_facet_count.restype = C.c_uint32
        L.orc_facet_count.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_facet_stats.restype = None
        L.orc_facet_stats.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_facet_value_set.restype = None
        L.orc_facet_value_set.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_facet_value_count.restype = C.c_uint32
        L.orc_facet_value_count.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_bench_keyword.restype = C.c_double
        L.orc_bench_keyword.argtypes = [C.c_void_p, C.POINTER(KwQuery), C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64)]
        L.orc_bench_vector.restype = C.c_double
        L.orc_bench_vector.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_match.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p]
        L.orc_match_score.restype = C.c_uint64
        L.orc_match_score.argtypes = [C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint32, C.c_uint32, C.c_uint8]
        L.orc_float_to_int64.restype = C.c_int64
        L.orc_float_to_int64.argtypes = [C.c_float]
        L.orc_int64_to_float.restype = C.c_float
        L.orc_int64_to_float.argtypes = [C.c_int64]
        _lib = L
    return _lib


def set_ip_lanes(lanes):
    """process-wide summation order of the oracle's distance function: 4 = SSE (stock reference build, default), 8 = AVX, 16 = AVX-512"""
    if lib().orc_set_ip_lanes(int(lanes)) != 0:
        raise ValueError("ip lanes must be 4, 8 or 16")


def get_ip_lanes():
    return int(lib().orc_get_ip_lanes())


def ref_match_lib():
    """oracle/_ref/libref_match.so: the REFERENCE's own match_score.h (None if it was never built)."""
    if not os.path.exists(_REF):
        return None
    R = C.CDLL(_REF)
    R.ref_match.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p]
    R.ref_match_score.restype = C.c_uint64
    R.ref_match_score.argtypes = [C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint32, C.c_uint32, C.c_uint8]
    return R


_REF_TOPSTER = os.path.join(_DIR, "_ref", "libref_topster.so")


def ref_topster_lib():
    """oracle/_ref/libref_topster.so: the REFERENCE's own topster.h / loglogbeta.h / wyhash_v5.h (None if it was never built)."""
    if not os.path.exists(_REF_TOPSTER):
        return None
    R = C.CDLL(_REF_TOPSTER)
    R.ref_hash_wy.restype = C.c_uint64
    R.ref_hash_wy.argtypes = [C.c_char_p, C.c_uint64]
    R.ref_hash_combine.restype = C.c_uint64
    R.ref_hash_combine.argtypes = [C.c_uint64, C.c_uint64]
    R.ref_loglog_of_keys.restype = C.c_uint64
    R.ref_loglog_of_keys.argtypes = [C.c_void_p, C.c_uint64]
    R.ref_topster_run.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return R


def group_topster_run(capacity, distinct, first_pass, keys, dkeys, scores):
    """the restated distinct Topster fed (key, distinct_key, scores[3]) in order -> (add() return values, GroupedHits)"""
    keys = np.ascontiguousarray(keys, np.uint64); dkeys = np.ascontiguousarray(dkeys, np.uint64); scores = np.ascontiguousarray(scores, np.int64)
    n = keys.size
    ret = np.zeros(max(n, 1), np.int32)
    g, b = _alloc_grouped(max(n, 1), max(n, 1))
    lib().orc_group_topster_run(capacity, distinct, int(first_pass), n, _ptr(keys), _ptr(dkeys), _ptr(scores), _ptr(ret), C.byref(g))
    return ret[:n], GroupedHits(g, b)


def ref_group_topster_run(R, capacity, distinct, first_pass, keys, dkeys, scores):
    """the same through the reference's own Topster<KV> (ref_topster_lib()) -> (ret, group_size, distinct_key, keys, scores[n,3], groups_count)"""
    keys = np.ascontiguousarray(keys, np.uint64); dkeys = np.ascontiguousarray(dkeys, np.uint64); scores = np.ascontiguousarray(scores, np.int64)
    n = keys.size
    cap = max(n, 1)
    ret = np.zeros(cap, np.int32); gsize = np.zeros(cap, np.uint32); dk = np.zeros(cap, np.uint64); ok = np.zeros(cap, np.uint64); osc = np.zeros(cap * 3, np.int64)
    ng = C.c_uint32(0); gc = C.c_uint64(0)
    R.ref_topster_run(capacity, distinct, int(first_pass), n, _ptr(keys), _ptr(dkeys), _ptr(scores), _ptr(ret), cap, cap, C.byref(ng), _ptr(gsize), _ptr(dk),
                      _ptr(ok), _ptr(osc), C.byref(gc))
    tot = int(gsize[:ng.value].sum())
    return ret[:n], gsize[:ng.value].copy(), dk[:ng.value].copy(), ok[:tot].copy(), osc[:tot * 3].reshape(tot, 3).copy(), int(gc.value)


def distinct_ids(n_docs, fields, group_missing_values=False):
    """Index::get_distinct_id per document over the group_by fields: fields = [(doc_ptr uint64[n_docs+1], hashes uint32[]), ...] (facet hash indexes, CSR)
    -> (distinct uint64[n_docs], has_value uint8[n_docs])"""
    ptrs = [np.ascontiguousarray(f[0], np.uint64) for f in fields]
    hs = [np.ascontiguousarray(f[1] if len(f[1]) else [0], np.uint32) for f in fields]
    pa = (C.c_void_p * len(fields))(*[p.ctypes.data for p in ptrs])
    ha = (C.c_void_p * len(fields))(*[h.ctypes.data for h in hs])
    out = np.zeros(n_docs, np.uint64); hv = np.zeros(n_docs, np.uint8)
    lib().orc_distinct_ids(n_docs, len(fields), pa, ha, int(group_missing_values), _ptr(out), _ptr(hv))
    return out, hv


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class HitList:
    """Decoded Result: numpy arrays in topster order."""
    def __init__(self, keys, scores, text_match, vdist, msi, num_keyword_matches, result_ids, n_result_ids, cutoff):
        self.keys, self.scores, self.text_match, self.vector_distance = keys, scores, text_match, vdist
        self.match_score_index = msi
        self.num_keyword_matches, self.result_ids, self.n_result_ids = num_keyword_matches, result_ids, n_result_ids
        self.search_cutoff = cutoff


class OracleIndex:
    def __init__(self, n_fields=1, n_columns=1):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_create(n_fields, n_columns))
        self.dim = 0

    def close(self):
        if self.h:
            self.L.orc_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- index time ----
    def index_plain(self, seq_id, field, tokens):
        t = _u32(tokens)
        self.L.orc_index_plain(self.h, seq_id, field, _ptr(t), t.size)

    def index_array(self, seq_id, field, elems):
        lens = _u32([len(e) for e in elems])
        flat = _u32([t for e in elems for t in e])
        self.L.orc_index_array(self.h, seq_id, field, _ptr(flat), _ptr(lens), lens.size)

    def load_posting(self, field, term, ids, offset_index, offsets):
        ids, offset_index, offsets = _u32(ids), _u32(offset_index), _u32(offsets)
        self.L.orc_load_posting(self.h, field, term, _ptr(ids), _ptr(offset_index), _ptr(offsets), ids.size, offsets.size)

    def dump_posting(self, field, term):
        no = C.c_uint32(0)
        n = self.L.orc_dump_posting(self.h, field, term, None, None, None, C.byref(no))
        ids = np.zeros(n, np.uint32); oi = np.zeros(n, np.uint32); off = np.zeros(no.value, np.uint32)
        if n:
            self.L.orc_dump_posting(self.h, field, term, _ptr(ids), _ptr(oi), off.ctypes.data_as(C.c_void_p), C.byref(no))
        return ids, oi, off

    def terms(self, field):
        n = self.L.orc_list_terms(self.h, field, None, 0)
        t = np.zeros(n, np.uint32)
        if n:
            self.L.orc_list_terms(self.h, field, _ptr(t), n)
        return np.sort(t)

    def set_num_docs(self, n):
        self.L.orc_set_num_docs(self.h, n)

    def num_docs(self):
        return self.L.orc_num_docs(self.h)

    def set_sort_dense(self, column, vals):
        v = np.ascontiguousarray(vals, dtype=np.int64)
        self.L.orc_set_sort_dense(self.h, column, _ptr(v), v.size)

    def set_sort(self, column, seq_id, v):
        self.L.orc_set_sort(self.h, column, seq_id, int(v))

    def vec_init(self, dim, metric=METRIC_IP):
        self.dim = dim
        self.L.orc_vec_init(self.h, dim, metric)

    def vec_add(self, labels, data):
        labels = _u32(labels)
        data = np.ascontiguousarray(data, dtype=np.float32).reshape(labels.size, self.dim)
        self.L.orc_vec_add(self.h, _ptr(labels), _ptr(data), labels.size)

    def vec_get(self, label):
        out = np.zeros(self.dim, np.float32)
        rc = self.L.orc_vec_get(self.h, label, _ptr(out))
        return out if rc == 0 else None

    # ---- query time ----
    def make_query(self, tokens, fields=((0, 15),), sort=((SORT_TEXT_MATCH, 0, 1), (SORT_SEQ_ID, 0, 1)), fetch_size=10,
                   match_type=MAX_SCORE, prioritize_exact_match=True, prioritize_token_position=False,
                   prioritize_num_matching_fields=True, total_cost=0, excluded_ids=None, filter_ids=None, topster_size=0, dropped_tokens=None,
                   syn_orig_num_tokens=-1, orig_num_tokens=0, is_synonym_query=False, demote_synonym_match=False):
        q = KwQuery()
        q.syn_orig_num_tokens, q.orig_num_tokens = syn_orig_num_tokens, orig_num_tokens
        q.is_synonym_query, q.demote_synonym_match = int(is_synonym_query), int(demote_synonym_match)
        keep = []
        t = _u32(tokens); keep.append(t)
        q.tokens = t.ctypes.data_as(C.POINTER(C.c_uint32)); q.n_tokens = t.size
        fid = _u32([f[0] for f in fields]); fw = np.ascontiguousarray([f[1] for f in fields], dtype=np.int64)
        keep += [fid, fw]
        q.field_ids = fid.ctypes.data_as(C.POINTER(C.c_uint32)); q.field_weights = fw.ctypes.data_as(C.POINTER(C.c_int64))
        q.n_fields = fid.size
        q.match_type = match_type
        q.prioritize_exact_match = int(prioritize_exact_match)
        q.prioritize_token_position = int(prioritize_token_position)
        q.prioritize_num_matching_fields = int(prioritize_num_matching_fields)
        q.total_cost = total_cost
        for i, s in enumerate(sort):
            q.sort_kind[i], q.sort_column[i], q.sort_order[i] = s
        q.n_sort = len(sort)
        q.fetch_size = fetch_size
        q.topster_size = topster_size
        if excluded_ids is not None and len(excluded_ids):
            e = _u32(excluded_ids); keep.append(e)
            q.excluded_ids = e.ctypes.data_as(C.POINTER(C.c_uint32)); q.n_excluded = e.size
        if filter_ids is not None and len(filter_ids):
            f = _u32(filter_ids); keep.append(f)
            q.filter_ids = f.ctypes.data_as(C.POINTER(C.c_uint32)); q.n_filter = f.size
        if dropped_tokens is not None and len(dropped_tokens):
            d = _u32(dropped_tokens); keep.append(d)
            q.dropped_tokens = d.ctypes.data_as(C.POINTER(C.c_uint32)); q.n_dropped = d.size
        q._keep = keep
        return q

    @staticmethod
    def _alloc(cap, ids_cap):
        r = Result()
        bufs = dict(keys=np.zeros(cap, np.uint64), scores=np.zeros(cap * 3, np.int64), tm=np.zeros(cap, np.int64),
                    vd=np.zeros(cap, np.float32), msi=np.zeros(cap, np.int8), ids=np.zeros(max(ids_cap, 1), np.uint32))
        r.cap = cap
        r.keys = bufs["keys"].ctypes.data_as(C.POINTER(C.c_uint64))
        r.scores = bufs["scores"].ctypes.data_as(C.POINTER(C.c_int64))
        r.text_match = bufs["tm"].ctypes.data_as(C.POINTER(C.c_int64))
        r.vector_distance = bufs["vd"].ctypes.data_as(C.POINTER(C.c_float))
        r.match_score_index = bufs["msi"].ctypes.data_as(C.POINTER(C.c_int8))
        if ids_cap:
            r.result_ids = bufs["ids"].ctypes.data_as(C.POINTER(C.c_uint32))
            r.result_ids_cap = ids_cap
        return r, bufs

    @staticmethod
    def _decode(r, b):
        n = r.n
        m = min(r.n_result_ids, r.result_ids_cap)
        return HitList(b["keys"][:n].copy(), b["scores"][:n * 3].reshape(n, 3).copy(), b["tm"][:n].copy(), b["vd"][:n].copy(),
                       b["msi"][:n].copy(), r.num_keyword_matches, b["ids"][:m].copy(), r.n_result_ids, bool(r.search_cutoff))

    def search_keyword(self, q, cap=1024, ids_cap=0):
        r, b = self._alloc(cap, ids_cap)
        self.L.orc_search_keyword(self.h, C.byref(q), C.byref(r))
        return self._decode(r, b)

    def search_keyword_grouped(self, q, distinct, group_limit, first_pass, has_value=None, group_missing_values=False, group_cap=4096, kv_cap=65536, ids_cap=0):
        """one pass of a group_by search (oracle_index.h search_keyword_grouped): distinct[seq_id] = get_distinct_id's result"""
        d = np.ascontiguousarray(distinct, np.uint64)
        hv = np.ascontiguousarray(has_value, np.uint8) if has_value is not None else None
        g, b = _alloc_grouped(group_cap, kv_cap, ids_cap)
        self.L.orc_search_keyword_grouped(self.h, C.byref(q), _ptr(d), _ptr(hv) if hv is not None else None, d.size, int(group_missing_values), group_limit,
                                          int(first_pass), C.byref(g))
        return GroupedHits(g, b)

    def search_candidates_grouped(self, combos, distinct, group_limit, first_pass, has_value=None, group_missing_values=False, group_cap=4096, kv_cap=65536, ids_cap=0):
        """Index::search_all_candidates with group_by: one grouped pass per combination over ONE collector -> (GroupedHits, query_index per KV)"""
        d = np.ascontiguousarray(distinct, np.uint64)
        hv = np.ascontiguousarray(has_value, np.uint8) if has_value is not None else None
        g, b = _alloc_grouped(group_cap, kv_cap, ids_cap)
        arr = (KwQuery * len(combos))(*combos)
        qi = np.zeros(kv_cap, np.uint16)
        self.L.orc_search_candidates_grouped(self.h, arr, len(combos), _ptr(d), _ptr(hv) if hv is not None else None, d.size, int(group_missing_values), group_limit,
                                             int(first_pass), C.byref(g), _ptr(qi))
        gh = GroupedHits(g, b)
        return gh, qi[:gh.keys.size].copy()

    def search_wildcard(self, q, cap=1024, ids_cap=0):
        r, b = self._alloc(cap, ids_cap)
        self.L.orc_search_wildcard(self.h, C.byref(q), C.byref(r))
        return self._decode(r, b)

    def search_candidates(self, combos, cap=1024, ids_cap=0):
        """Index::search_all_candidates over the candidate-token combinations `combos` (make_query results, pass order);
        returns (HitList, query_index[n])"""
        r, b = self._alloc(cap, ids_cap)
        arr = (KwQuery * len(combos))(*combos)
        qi = np.zeros(cap, np.uint16)
        self.L.orc_search_candidates(self.h, arr, len(combos), C.byref(r), _ptr(qi))
        return self._decode(r, b), qi[:r.n].copy()

    def search_vector(self, qvec, k=0, fetch_size=10, sort=((SORT_VECTOR_DISTANCE, 0, -1), (SORT_SEQ_ID, 0, 1)),
                      distance_threshold=3.4028234663852886e38, filter_ids=None, cap=1024, ids_cap=0, excluded_ids=None,
                      flat_search_cutoff=0, query_doc=None):
        """the vector branch of Index::search (src/index.cpp:3645-3732): filter_ids = what filter_by matched (None = no filter_by);
        fewer filter ids than flat_search_cutoff -> the FLAT branch (every filter id into the Topster, found = the ids kept);
        query_doc = seq_id of `vec:([], id: X)` (the caller passes X's stored vector as qvec)"""
        r, b = self._alloc(cap, ids_cap)
        qv = np.ascontiguousarray(qvec, dtype=np.float32)
        sk = np.array([s[0] for s in sort], np.int32); sc = np.array([s[1] for s in sort], np.int32); so = np.array([s[2] for s in sort], np.int32)
        f = _u32(filter_ids) if filter_ids is not None else None
        e = _u32(excluded_ids) if excluded_ids is not None else None
        self.L.orc_search_vector2(self.h, _ptr(qv), k, C.c_float(distance_threshold), _ptr(sk), _ptr(sc), _ptr(so), len(sort), fetch_size,
                                  _ptr(f) if f is not None and f.size else None, f.size if f is not None else 0, 1 if f is not None else 0,
                                  _ptr(e) if e is not None and e.size else None, e.size if e is not None else 0,
                                  C.c_uint64(int(flat_search_cutoff)), 1 if query_doc is not None else 0, int(query_doc or 0), C.byref(r))
        return self._decode(r, b)

    def search_hybrid(self, q, qvec, k=0, alpha=0.3, distance_threshold=3.4028234663852886e38, cap=1024, ids_cap=0, rerank=False):
        r, b = self._alloc(cap, ids_cap)
        qv = np.ascontiguousarray(qvec, dtype=np.float32)
        self.L.orc_search_hybrid_rerank(self.h, C.byref(q), _ptr(qv), k, alpha, distance_threshold, int(bool(rerank)), C.byref(r))
        return self._decode(r, b)

    # ---- facets (oracle/facet_count.h) ----
    def facet_set(self, field, doc_ptr, hashes):
        dp = np.ascontiguousarray(doc_ptr, dtype=np.uint64)
        hs = _u32(hashes)
        self.L.orc_facet_set(self.h, field, _ptr(dp), hs.ctypes.data_as(C.c_void_p), dp.size - 1)

    def facet_count(self, field, ids, sample_mod=1, allowed_hashes=None, cap=65536):
        ids = _u32(ids)
        a = _u32(allowed_hashes) if allowed_hashes is not None else None
        h, c, d, p = (np.zeros(cap, np.uint32) for _ in range(4))
        n = self.L.orc_facet_count(self.h, field, ids.ctypes.data_as(C.c_void_p), ids.size, sample_mod, a.ctypes.data_as(C.c_void_p) if a is not None else None,
                                   a.size if a is not None else 0, _ptr(h), _ptr(c), _ptr(d), _ptr(p), cap)
        m = min(n, cap)
        return h[:m].copy(), c[:m].copy(), d[:m].copy(), p[:m].copy(), n

    def facet_count_ex(self, field, ids, sample_mod=1, allowed_hashes=None, ranges=None, doc_vals=None, distinct_ids=None, group_missing_values=False, cap=65536):
        """the grouped (distinct_ids = the per-document distinct id column) and range (ranges = [(upper, lower), ...], doc_vals = the sort index as a dense
        column) forms of the walk -> (keys uint64 ascending, count, doc_id, array_pos, n)"""
        ids = _u32(ids)
        a = _u32(allowed_hashes) if allowed_hashes is not None else None
        up = np.ascontiguousarray([r[0] for r in ranges], dtype=np.int64) if ranges else None
        lo = np.ascontiguousarray([r[1] for r in ranges], dtype=np.int64) if ranges else None
        dv = np.ascontiguousarray(doc_vals, dtype=np.int64) if doc_vals is not None else None
        di = np.ascontiguousarray(distinct_ids, dtype=np.uint64) if distinct_ids is not None else None
        k = np.zeros(cap, np.uint64)
        c, d, p = (np.zeros(cap, np.uint32) for _ in range(3))
        fn = self.L.orc_facet_count_ex
        fn.restype = C.c_uint32
        fn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64,
                       C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        vp = lambda x: x.ctypes.data_as(C.c_void_p) if x is not None else None
        n = fn(self.h, field, vp(ids), ids.size, sample_mod, vp(a), a.size if a is not None else 0, vp(up), vp(lo), len(ranges) if ranges else 0, vp(dv),
               dv.size if dv is not None else 0, vp(di), di.size if di is not None else 0, int(group_missing_values), vp(k), vp(c), vp(d), vp(p), cap)
        m = min(n, cap)
        return k[:m].copy(), c[:m].copy(), d[:m].copy(), p[:m].copy(), n

    def facet_stats(self, field, ids, value_type, sample_mod=1, int64_map=None):
        ids = _u32(ids)
        mh = _u32(int64_map[0]) if int64_map is not None else None
        mv = np.ascontiguousarray(int64_map[1], dtype=np.int64) if int64_map is not None else None
        out = np.zeros(4, np.float64)
        self.L.orc_facet_stats(self.h, field, ids.ctypes.data_as(C.c_void_p), ids.size, sample_mod, value_type, _ptr(mh) if mh is not None else None,
                               _ptr(mv) if mv is not None else None, mh.size if mh is not None else 0, _ptr(out))
        return float(out[0]), float(out[1]), float(out[2]), int(out[3])

    def facet_value_set(self, field, value_ptr, seq_ids, total_counts):
        vp = np.ascontiguousarray(value_ptr, dtype=np.uint64)
        si, tc = _u32(seq_ids), _u32(total_counts)
        self.L.orc_facet_value_set(self.h, field, _ptr(vp), si.ctypes.data_as(C.c_void_p), tc.ctypes.data_as(C.c_void_p), vp.size - 1)

    def facet_value_count(self, field, ids, max_facets, wildcard_no_filter=False, estimate=False, sample_interval=1, order=None, cap=65536):
        ids = _u32(ids)
        o = _u32(order) if order is not None else None
        v, c, d = (np.zeros(cap, np.uint32) for _ in range(3))
        n = self.L.orc_facet_value_count(self.h, field, ids.ctypes.data_as(C.c_void_p), ids.size, int(max_facets), int(wildcard_no_filter), int(estimate), int(sample_interval),
                                         _ptr(o) if o is not None else None, _ptr(v), _ptr(c), _ptr(d), cap)
        return v[:n].copy(), c[:n].copy(), d[:n].copy()

    # ---- HNSW (oracle/hnsw_graph.h) ----
    def hnsw_build(self, M=16, ef_construction=200, seed=100):
        """HierarchicalNSW(space, 16, M, ef_construction, 100, true) + addPoint per row in insertion order (include/index.h:365-367)"""
        self.L.orc_hnsw_build(self.h, M, ef_construction, seed)

    def hnsw_bulk_build(self, M=16, ef_construction=200, seed=100, seed_min=0, max_batch=0):
        """the batched bulk build of tsgpu_vec_hnsw_build over the rows added with vec_add (hnsw_graph_t::bulk_build)"""
        self.L.orc_hnsw_bulk_build(self.h, M, ef_construction, seed, seed_min, max_batch)

    def hnsw_add(self, labels, X):
        """vec_add + hnswlib addPoint for rows that arrive after hnsw_build (insertion order = row order)"""
        self.vec_add(labels, X)
        self.L.orc_hnsw_add_new_rows(self.h)

    def hnsw_mark_deleted(self, label):
        return self.L.orc_hnsw_mark_deleted(self.h, int(label))

    def hnsw_upsert(self, label, x):
        """addPoint(vec, label, replace_deleted=True) the way the reference calls it (src/index.cpp:1052-1054): the flat store takes the vector
        (vec_add), the graph updates a live label in place, re-uses a deleted slot, or appends. Returns the graph's internal id."""
        self.vec_add(np.array([label], np.uint32), np.ascontiguousarray(x, np.float32).reshape(1, -1))
        return int(self.L.orc_hnsw_add_point_replace(self.h, int(label)))

    def hnsw_labels(self, n):
        out = np.zeros(n, np.uint64)
        self.L.orc_hnsw_labels(self.h, _ptr(out))
        return out

    def hnsw_export(self):
        """-> dict(n, maxlevel, enterpoint, M, levels[n], link0[n, 1+2M], upper_ptr[n+1], upper_links[n_upper, 1+M])"""
        info = np.zeros(4, np.int32)
        n_upper = self.L.orc_hnsw_export(self.h, _ptr(info), None, None, None, None)
        n, M = int(info[0]), int(info[3])
        levels = np.zeros(n, np.uint32); link0 = np.zeros((n, 1 + 2 * M), np.uint32)
        upper_ptr = np.zeros(n + 1, np.uint64); upper = np.zeros((max(n_upper, 1), 1 + M), np.uint32)
        self.L.orc_hnsw_export(self.h, _ptr(info), _ptr(levels), _ptr(link0), _ptr(upper_ptr), _ptr(upper))
        return dict(n=n, maxlevel=int(info[1]), enterpoint=int(info[2]), M=M, levels=levels, link0=link0, upper_ptr=upper_ptr,
                    upper_links=upper[:n_upper])

    def hnsw_import(self, graph):
        """adopt a graph in the flat mirror form (hnsw_export's keys) over the rows added with vec_add (row i = internal id i)"""
        l0 = np.ascontiguousarray(graph["link0"], dtype=np.uint32)
        up = np.ascontiguousarray(graph["upper_ptr"], dtype=np.uint64)
        ul = np.ascontiguousarray(graph["upper_links"], dtype=np.uint32)
        if ul.size == 0:
            ul = np.zeros((1, 1 + int(graph["M"])), np.uint32)
        rc = self.L.orc_hnsw_import(self.h, int(graph["M"]), int(graph["maxlevel"]), int(graph["enterpoint"]), _ptr(l0), _ptr(up), _ptr(ul))
        assert rc == 0, "orc_hnsw_import: malformed graph"

    def hnsw_search_batch(self, Q, k, ef, threads=1, functor_present=True):
        """searchKnnCloserFirst for every row of Q on `threads` host threads -> (dist[n, k], labels[n, k], counts[n])"""
        Q = np.ascontiguousarray(Q, dtype=np.float32)
        n = Q.shape[0]
        dist = np.zeros((n, k), np.float32); lab = np.zeros((n, k), np.uint64); cnt = np.zeros(n, np.uint32)
        self.L.orc_hnsw_search_batch(self.h, _ptr(Q), n, k, ef, int(functor_present), int(threads), _ptr(dist), _ptr(lab), _ptr(cnt))
        return dist, lab, cnt

    def hnsw_search(self, qvec, k, ef, allow_ids=None, functor_present=True):
        qv = np.ascontiguousarray(qvec, dtype=np.float32)
        d = np.zeros(max(k, 1), np.float32); l = np.zeros(max(k, 1), np.uint64); nd = np.zeros(1, np.uint64)
        a = _u32(allow_ids) if allow_ids is not None else None
        n = self.L.orc_hnsw_search(self.h, _ptr(qv), k, ef, int(functor_present), _ptr(a) if a is not None else None, a.size if a is not None else 0,
                                   _ptr(d), _ptr(l), _ptr(nd))
        return d[:n], l[:n], int(nd[0])

    def flat_knn(self, qvec, k, allow_ids=None):
        qv = np.ascontiguousarray(qvec, dtype=np.float32)
        d = np.zeros(k, np.float32); l = np.zeros(k, np.uint32)
        a = _u32(allow_ids) if allow_ids is not None else None
        n = self.L.orc_flat_knn(self.h, _ptr(qv), k, _ptr(a) if a is not None else None, a.size if a is not None else 0, _ptr(d), _ptr(l))
        return d[:n], l[:n]

    def bench_keyword(self, base_query, tokens, n_threads):
        t = np.ascontiguousarray(tokens, dtype=np.uint32)
        nq = t.shape[0]
        per = np.zeros(nq, np.float64)
        chk = C.c_uint64(0)
        wall = self.L.orc_bench_keyword(self.h, C.byref(base_query), _ptr(t), nq, n_threads, _ptr(per), C.byref(chk))
        return wall, per

    def bench_vector(self, queries, k, n_threads):
        qv = np.ascontiguousarray(queries, dtype=np.float32)
        per = np.zeros(qv.shape[0], np.float64)
        wall = self.L.orc_bench_vector(self.h, _ptr(qv), qv.shape[0], k, n_threads, _ptr(per))
        return wall, per
