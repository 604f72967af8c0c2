"""ctypes declarations of include/tsgpu.h. Loads typesense_amd/libtsgpu.so (built by typesense_amd/build.py)."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TSGPU_LIB") or os.path.join(HERE, "libtsgpu.so")     # (TSGPU_LIB: a variant build under typesense_amd/variants/, tools/ experiments only)

TSGPU_OK, ERR_INVALID, ERR_NOT_FOUND, ERR_DEADLINE, ERR_DEVICE, ERR_UNSUPPORTED, ERR_NO_MEMORY = 0, 400, 404, 408, 500, 501, 507
MEM_HOST, MEM_DEVICE = 0, 1
SORT_TEXT_MATCH, SORT_SEQ_ID, SORT_INT64_COLUMN, SORT_VECTOR_DISTANCE = 0, 1, 2, 3
METRIC_IP, METRIC_COSINE = 0, 1
MAX_SCORE, MAX_WEIGHT, SUM_SCORE = 0, 1, 2
MAX_QUERY_TOKENS = 10
FLT_MAX = 3.4028234663852886e38


class TsgpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("tsgpu error %d: %s" % (code, msg))
        self.code = code


class SortBy(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("order", C.c_int8), ("column", C.c_uint16)]


class KwQueryC(C.Structure):
    _fields_ = [("n_tokens", C.c_uint32), ("term_ids", C.c_uint32 * MAX_QUERY_TOKENS),
                ("n_fields", C.c_uint32), ("field_ids", C.c_uint32 * 4), ("field_weights", C.c_int32 * 4),
                ("match_type", C.c_uint8), ("prioritize_exact_match", C.c_uint8), ("prioritize_token_position", C.c_uint8),
                ("prioritize_num_matching_fields", C.c_uint8),
                ("total_cost", C.c_uint32), ("n_sort", C.c_uint32), ("sort", SortBy * 3), ("topster_size", C.c_uint32),
                ("excluded_ids", C.POINTER(C.c_uint32)), ("n_excluded", C.c_uint32),
                ("filter_ids", C.POINTER(C.c_uint32)), ("n_filter", C.c_uint32),
                ("deadline_us", C.c_uint64), ("n_dropped", C.c_uint32), ("dropped_term_ids", C.c_uint32 * 4),
                ("is_synonym_query", C.c_uint8), ("demote_synonym_match", C.c_uint8), ("syn_orig_num_tokens_p1", C.c_uint8), ("orig_num_tokens", C.c_uint8)]


class HitsC(C.Structure):
    _fields_ = [("mem", C.c_int), ("k_stride", C.c_uint32),
                ("keys", C.c_void_p), ("scores", C.c_void_p), ("text_match", C.c_void_p), ("vector_distance", C.c_void_p),
                ("match_score_index", C.c_void_p), ("n_hits", C.c_void_p), ("num_matched", C.c_void_p),
                ("status", C.c_void_p), ("search_cutoff", C.c_void_p)]


class GroupByC(C.Structure):
    _fields_ = [("group_limit", C.c_uint32), ("column", C.c_uint16), ("first_pass", C.c_uint8), ("group_missing_values", C.c_uint8),
                ("wildcard", C.c_uint8), ("pad", C.c_uint8 * 3)]


class GroupedHitsC(C.Structure):
    _fields_ = [("g_stride", C.c_uint32), ("n_groups", C.c_void_p), ("distinct_key", C.c_void_p), ("group_size", C.c_void_p), ("group_found", C.c_void_p),
                ("groups_total", C.c_void_p), ("groups_count", C.c_void_p), ("loglog_registers", C.c_void_p)]


class VecQueryC(C.Structure):
    _fields_ = [("k", C.c_uint32), ("fetch_size", C.c_uint32), ("distance_threshold", C.c_float),
                ("n_sort", C.c_uint32), ("sort", SortBy * 3), ("topster_size", C.c_uint32),
                ("filter_by_provided", C.c_uint32), ("filter_ids", C.c_void_p), ("n_filter", C.c_uint32), ("n_excluded", C.c_uint32),
                ("excluded_ids", C.c_void_p), ("flat_search_cutoff", C.c_uint64), ("query_doc_given", C.c_uint32), ("query_seq_id", C.c_uint32)]


class HybridParamsC(C.Structure):
    _fields_ = [("k", C.c_uint32), ("fetch_size", C.c_uint32), ("alpha", C.c_float), ("distance_threshold", C.c_float), ("rerank_hybrid_matches", C.c_uint32)]


NO_COLUMN = 0xFFFFFFFF


class FacetCountsC(C.Structure):
    _fields_ = [("cap", C.c_uint32), ("hash", C.c_void_p), ("count", C.c_void_p), ("doc_id", C.c_void_p), ("array_pos", C.c_void_p), ("n_values", C.c_void_p)]


class TimingsC(C.Structure):
    _fields_ = [("kw_search_ms", C.c_float), ("kw_merge_ms", C.c_float), ("vec_knn_ms", C.c_float), ("vec_merge_ms", C.c_float),
                ("total_ms", C.c_float), ("vec_scan_ms", C.c_float), ("kw_algorithmic_bytes", C.c_uint64), ("vec_flops", C.c_uint64),
                ("vec_scan_bytes", C.c_uint64), ("kw_find_ms", C.c_float)]


class AuxTimingsC(C.Structure):
    _fields_ = [("gb_id_pass_ms", C.c_float), ("gb_kernels_ms", C.c_float), ("gb_fold_ms", C.c_float), ("gb_select_ms", C.c_float),
                ("gb_matched_ids", C.c_uint64), ("gb_table_slots", C.c_uint64), ("gb_algorithmic_bytes", C.c_uint64),
                ("facet_kernels_ms", C.c_float), ("facet_count_ms", C.c_float), ("facet_ids", C.c_uint64), ("facet_table_slots", C.c_uint64),
                ("facet_algorithmic_bytes", C.c_uint64)]


class HnswBuildInfoC(C.Structure):
    _fields_ = [("n", C.c_uint32), ("n_seed", C.c_uint32), ("n_batches", C.c_uint32), ("unlinked", C.c_uint32), ("maxlevel", C.c_int32), ("enterpoint", C.c_uint32),
                ("seed_seconds", C.c_double), ("device_seconds", C.c_double), ("search_seconds", C.c_double), ("link_seconds", C.c_double)]


class KwTouchedC(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("find_requested_bytes", "find_driver_ids", "find_metadata", "find_tile_dma", "find_probes", "find_records",
                                          "find_work_items", "find_hit_records", "score_requested_bytes")]


class KwFootprintC(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_lists", "n_ids", "ids_bytes", "block_metadata_bytes", "directory_bytes", "payload_bytes")]


class FacetStatsC(C.Structure):
    _fields_ = [("fvmin", C.c_double), ("fvmax", C.c_double), ("fvsum", C.c_double), ("fvcount", C.c_uint64), ("sum_exact", C.c_int32), ("pad", C.c_int32)]


class FacetValueCountsC(C.Structure):
    _fields_ = [("cap", C.c_uint32), ("value_index", C.c_void_p), ("count", C.c_void_p), ("doc_id", C.c_void_p), ("n_found", C.c_void_p)]


FACET_INT32, FACET_INT64, FACET_FLOAT = 0, 1, 2


class GroupTimingsC(C.Structure):
    _fields_ = [("local_ms", C.c_float), ("exchange_merge_ms", C.c_float), ("exchange_bytes_per_member", C.c_uint64), ("hit_exchange_bytes_per_member", C.c_uint64), ("exchange_kernels_ms", C.c_float)]


XCHG_RCCL, XCHG_COPY, XCHG_HOST = 0, 1, 2

HOST_COLLECTIVE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)     # (user, send, recv, bytes) -> 0 on success


class HostCollectivesC(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_gather", HOST_COLLECTIVE_FN), ("all_to_all", HOST_COLLECTIVE_FN)]


EXPORTS = [
    "tsgpu_abi_version", "tsgpu_create", "tsgpu_destroy", "tsgpu_last_error", "tsgpu_set_stream", "tsgpu_set_option", "tsgpu_get_counter", "tsgpu_device_bytes",
    "tsgpu_field_create", "tsgpu_term_upsert", "tsgpu_posting_upsert", "tsgpu_posting_erase", "tsgpu_terms_load_csr", "tsgpu_column_set", "tsgpu_set_num_docs", "tsgpu_commit",
    "tsgpu_term_num_ids", "tsgpu_term_download", "tsgpu_keyword_search_batch", "tsgpu_wildcard_search_batch", "tsgpu_keyword_search_candidates_batch", "tsgpu_candidates_result_ids", "tsgpu_keep_result_ids", "tsgpu_result_ids",
    "tsgpu_keyword_search_batch_ids", "tsgpu_keyword_search_grouped_batch", "tsgpu_keyword_search_grouped_candidates_batch", "tsgpu_id_lists_count", "tsgpu_id_lists_ids", "tsgpu_id_lists_free", "tsgpu_facet_set", "tsgpu_facet_count_batch", "tsgpu_facet_count_grouped_batch", "tsgpu_facet_range_count_batch", "tsgpu_facet_stats_batch", "tsgpu_facet_value_set", "tsgpu_facet_value_count_batch",
    "tsgpu_vec_create", "tsgpu_vec_upsert", "tsgpu_vec_delete", "tsgpu_vec_get", "tsgpu_vec_count", "tsgpu_vec_knn_batch",
    "tsgpu_vec_hnsw_load", "tsgpu_vec_hnsw_enable", "tsgpu_vec_hnsw_build", "tsgpu_vec_hnsw_export", "tsgpu_vec_hnsw_search_batch", "tsgpu_vec_distances", "tsgpu_ip_distance", "tsgpu_vector_search_batch", "tsgpu_vector_search_batch_ids", "tsgpu_hybrid_search_batch", "tsgpu_hybrid_fuse_batch", "tsgpu_keyword_aux_scores", "tsgpu_merge_shard_hits", "tsgpu_merge_shard_hits_device", "tsgpu_last_timings", "tsgpu_last_aux_timings", "tsgpu_kw_last_touched", "tsgpu_kw_lists_footprint",
    "tsgpu_group_create_local", "tsgpu_group_unique_id", "tsgpu_group_create_rank", "tsgpu_group_create_rank_host", "tsgpu_group_destroy", "tsgpu_group_size", "tsgpu_group_keyword_search_batch", "tsgpu_group_keyword_search_candidates_batch", "tsgpu_group_wildcard_search_batch", "tsgpu_group_keyword_search_grouped_batch", "tsgpu_group_keyword_search_grouped_candidates_batch", "tsgpu_group_facet_count_batch", "tsgpu_group_facet_range_count_batch", "tsgpu_group_facet_stats_batch",
    "tsgpu_group_vec_knn_batch", "tsgpu_group_hybrid_search_batch", "tsgpu_group_last_timings", "tsgpu_group_set_option",
]

_libs = {}


def lib(path=None):
    """Load the C-ABI library. Raises (never falls back to a CPU path) when it is missing."""
    path = path or LIB_PATH
    if path in _libs:
        return _libs[path]
    # a process that also uses torch on the GPU: torch's bundled HIP runtime must open the device before /opt/rocm's (which
    # libtsgpu.so links) does, or torch.cuda's lazy initialisation later reports "No HIP GPUs are available"
    import sys
    if "torch" in sys.modules:
        try:
            _t = sys.modules["torch"]
            if _t.cuda.is_available():
                _t.cuda.init()
        except Exception:
            pass
    if not os.path.exists(path):
        raise TsgpuError(ERR_DEVICE, "%s not found: build it with `python -m typesense_amd.build` (hipcc, gfx950). "
                                     "There is no CPU fallback." % path)
    L = C.CDLL(path)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    L.tsgpu_abi_version.restype = i32
    L.tsgpu_create.argtypes = [i32, C.POINTER(vp)]
    L.tsgpu_destroy.argtypes = [vp]
    L.tsgpu_destroy.restype = None
    L.tsgpu_last_error.restype = C.c_char_p
    L.tsgpu_set_stream.argtypes = [vp, vp]
    L.tsgpu_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.tsgpu_get_counter.argtypes = [vp, C.c_char_p, C.POINTER(u64)]
    L.tsgpu_device_bytes.argtypes = [vp]
    L.tsgpu_device_bytes.restype = u64
    L.tsgpu_field_create.argtypes = [vp, u32, i32]
    L.tsgpu_term_upsert.argtypes = [vp, u32, u32, vp, vp, vp, u32, u32]
    L.tsgpu_posting_upsert.argtypes = [vp, u32, u32, u32, vp, u32]
    L.tsgpu_posting_erase.argtypes = [vp, u32, u32, u32]
    L.tsgpu_terms_load_csr.argtypes = [vp, u32, u32, vp, vp, vp, vp, vp, vp]
    L.tsgpu_column_set.argtypes = [vp, u32, vp, vp, u32, i32]
    L.tsgpu_set_num_docs.argtypes = [vp, u32]
    L.tsgpu_commit.argtypes = [vp]
    L.tsgpu_term_num_ids.argtypes = [vp, u32, u32]
    L.tsgpu_term_num_ids.restype = u32
    L.tsgpu_term_download.argtypes = [vp, u32, u32, vp, vp, vp, C.POINTER(u32)]
    L.tsgpu_keyword_search_batch.argtypes = [vp, vp, u32, C.POINTER(HitsC)]
    L.tsgpu_wildcard_search_batch.argtypes = [vp, vp, u32, C.POINTER(HitsC)]
    L.tsgpu_keyword_search_candidates_batch.argtypes = [vp, vp, vp, u32, C.POINTER(HitsC), vp, vp]
    L.tsgpu_candidates_result_ids.argtypes = [vp, u32, vp, u64]
    L.tsgpu_candidates_result_ids.restype = u64
    L.tsgpu_keep_result_ids.argtypes = [vp, i32]
    L.tsgpu_result_ids.argtypes = [vp, u32, vp, u64]
    L.tsgpu_result_ids.restype = u64
    L.tsgpu_keyword_search_batch_ids.argtypes = [vp, vp, u32, C.POINTER(HitsC), C.POINTER(vp)]
    L.tsgpu_keyword_search_grouped_batch.argtypes = [vp, vp, vp, u32, C.POINTER(HitsC), C.POINTER(GroupedHitsC), C.POINTER(vp)]
    L.tsgpu_keyword_search_grouped_candidates_batch.argtypes = [vp, vp, vp, vp, u32, C.POINTER(HitsC), C.POINTER(GroupedHitsC), vp, C.POINTER(vp)]
    L.tsgpu_id_lists_count.argtypes = [vp, u32]
    L.tsgpu_id_lists_count.restype = u64
    L.tsgpu_id_lists_ids.argtypes = [vp, u32]
    L.tsgpu_id_lists_ids.restype = C.POINTER(C.c_uint32)
    L.tsgpu_id_lists_free.argtypes = [vp]
    L.tsgpu_id_lists_free.restype = None
    L.tsgpu_facet_set.argtypes = [vp, u32, vp, vp, u32]
    L.tsgpu_facet_count_batch.argtypes = [vp, u32, vp, vp, u32, u32, vp, u32, C.POINTER(FacetCountsC)]
    L.tsgpu_facet_count_grouped_batch.argtypes = [vp, u32, vp, vp, u32, u32, vp, u32, u32, i32, C.POINTER(FacetCountsC)]
    L.tsgpu_facet_range_count_batch.argtypes = [vp, u32, u32, vp, vp, u32, vp, vp, u32, u32, u32, i32, vp]
    L.tsgpu_vec_create.argtypes = [vp, u32, u32, i32, u64]
    L.tsgpu_vec_upsert.argtypes = [vp, u32, vp, vp, u32, i32]
    L.tsgpu_vec_delete.argtypes = [vp, u32, u64]
    L.tsgpu_vec_get.argtypes = [vp, u32, u64, vp]
    L.tsgpu_vec_count.argtypes = [vp, u32]
    L.tsgpu_vec_count.restype = u64
    L.tsgpu_vec_knn_batch.argtypes = [vp, u32, vp, i32, u32, u32, vp, u32, vp, u32, vp, vp, vp, i32]
    L.tsgpu_vec_hnsw_load.argtypes = [vp, u32, u32, C.c_int32, u32, vp, vp, vp, u32]
    L.tsgpu_vec_hnsw_enable.argtypes = [vp, u32, u32, u32, u32, u32]
    L.tsgpu_vec_hnsw_build.argtypes = [vp, u32, u32, u32, u32, u32, u32, u32, C.POINTER(HnswBuildInfoC)]
    L.tsgpu_vec_hnsw_export.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp]
    L.tsgpu_vec_hnsw_search_batch.argtypes = [vp, u32, vp, i32, u32, u32, u32, i32, vp, u32, vp, u32, vp, vp, vp, i32]
    L.tsgpu_vec_distances.argtypes = [vp, u32, vp, vp, u32, vp]
    L.tsgpu_ip_distance.argtypes = [vp, vp, u32, i32]
    L.tsgpu_ip_distance.restype = C.c_float
    L.tsgpu_keyword_aux_scores.argtypes = [vp, vp, u32, vp, vp, u32, vp]
    L.tsgpu_vector_search_batch.argtypes = [vp, u32, C.POINTER(VecQueryC), vp, i32, u32, C.POINTER(HitsC)]
    L.tsgpu_vector_search_batch_ids.argtypes = [vp, u32, C.POINTER(VecQueryC), vp, i32, u32, C.POINTER(HitsC), C.POINTER(vp)]
    L.tsgpu_hybrid_search_batch.argtypes = [vp, vp, u32, C.POINTER(HybridParamsC), vp, i32, u32, C.POINTER(HitsC)]
    L.tsgpu_hybrid_fuse_batch.argtypes = [vp, vp, C.POINTER(HybridParamsC), i32, C.POINTER(HitsC), vp, vp, vp, u32, u32, C.POINTER(HitsC)]
    L.tsgpu_merge_shard_hits.argtypes = [vp, vp, u32, u32, u32, C.POINTER(HitsC)]
    L.tsgpu_merge_shard_hits_device.argtypes = [vp, C.POINTER(HitsC), u32, u32, u32, C.POINTER(HitsC)]
    L.tsgpu_last_timings.argtypes = [vp, C.POINTER(TimingsC)]
    L.tsgpu_kw_last_touched.argtypes = [vp, C.POINTER(KwTouchedC)]
    L.tsgpu_last_aux_timings.argtypes = [vp, C.POINTER(AuxTimingsC)]
    L.tsgpu_kw_lists_footprint.argtypes = [vp, vp, vp, u32, C.POINTER(KwFootprintC)]
    L.tsgpu_facet_stats_batch.argtypes = [vp, u32, i32, vp, vp, u32, u32, vp, vp, u32, vp]
    L.tsgpu_facet_value_set.argtypes = [vp, u32, vp, vp, vp, u32]
    L.tsgpu_facet_value_count_batch.argtypes = [vp, u32, vp, vp, u32, u32, i32, i32, u32, vp, C.POINTER(FacetValueCountsC)]
    L.tsgpu_group_create_local.argtypes = [vp, u32, i32, vp]
    L.tsgpu_group_unique_id.argtypes = [vp]
    L.tsgpu_group_create_rank.argtypes = [vp, vp, u32, u32, vp]
    L.tsgpu_group_create_rank_host.argtypes = [vp, C.POINTER(HostCollectivesC), u32, u32, vp]
    L.tsgpu_group_destroy.argtypes = [vp]
    L.tsgpu_group_destroy.restype = None
    L.tsgpu_group_size.argtypes = [vp]
    L.tsgpu_group_size.restype = u32
    L.tsgpu_group_keyword_search_batch.argtypes = [vp, vp, u32, u32, C.POINTER(HitsC)]
    L.tsgpu_group_keyword_search_candidates_batch.argtypes = [vp, vp, vp, u32, u32, C.POINTER(HitsC), vp, vp]
    L.tsgpu_group_wildcard_search_batch.argtypes = [vp, vp, u32, u32, C.POINTER(HitsC)]
    L.tsgpu_group_keyword_search_grouped_batch.argtypes = [vp, vp, vp, u32, C.POINTER(HitsC), C.POINTER(GroupedHitsC)]
    L.tsgpu_group_keyword_search_grouped_candidates_batch.argtypes = [vp, vp, vp, vp, u32, C.POINTER(HitsC), C.POINTER(GroupedHitsC), vp]
    L.tsgpu_group_facet_count_batch.argtypes = [vp, u32, vp, vp, u32, u32, vp, u32, C.POINTER(FacetCountsC)]
    L.tsgpu_group_facet_range_count_batch.argtypes = [vp, u32, u32, vp, vp, u32, vp, vp, u32, u32, vp]
    L.tsgpu_group_facet_stats_batch.argtypes = [vp, u32, i32, vp, vp, u32, u32, vp, vp, u32, vp]
    L.tsgpu_group_vec_knn_batch.argtypes = [vp, u32, vp, i32, u32, u32, vp, u32, vp, u32, vp, vp, vp, i32]
    L.tsgpu_group_hybrid_search_batch.argtypes = [vp, vp, u32, i32, C.POINTER(HybridParamsC), vp, i32, u32, u32, C.POINTER(HitsC)]
    L.tsgpu_group_last_timings.argtypes = [vp, C.POINTER(GroupTimingsC)]
    L.tsgpu_group_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    _libs[path] = L
    return L


def check(L, rc):
    if rc != TSGPU_OK:
        raise TsgpuError(rc, L.tsgpu_last_error().decode("utf-8", "replace"))
