/*
 * tsgpu.h — C-ABI of libtsgpu.so: the MI355X (gfx950) implementation of Typesense's query-time
 * scoring hot path. Plain C, plain pointers and sizes; no torch / C++ types cross this boundary.
 *
 * What each entry point replaces in the reference (typesense/typesense, paths relative to the
 * reference root; see SURVEY.md §8b and INTEGRATION.md for the call-site patches):
 *
 *   B1 keyword seam — the body of Index::search_across_fields from get_field_token_its to the end
 *   of the scoring lambda (src/index.cpp:5468-5551), i.e. or_iterator_t::intersect
 *   (include/or_iterator.h:61-182) + compute_aggregated_score (src/index.cpp:5227-5383) +
 *   score_results2 (src/index.cpp:6966-7098) + Match (include/match_score.h:129-275) +
 *   compute_sort_scores (src/index.cpp:5662-5907) + Topster::add/sort (include/topster.h:321-473):
 *       tsgpu_keyword_search_batch()
 *   fed by an index mirror built from the decoded posting blocks (posting_list_t::block_t,
 *   include/posting_list.h:56-77) and the numeric sort index (include/index.h:442):
 *       tsgpu_field_create() tsgpu_term_upsert() tsgpu_column_set() tsgpu_commit()
 *
 *   B2 vector seam — the methods Typesense calls on hnswlib::HierarchicalNSW<float> /
 *   InnerProductSpace (include/index.h:356-370; src/index.cpp:1003-1054, 3355-3386, 7423):
 *       tsgpu_vec_create()      ~ HierarchicalNSW ctor            (include/index.h:365-367)
 *       tsgpu_vec_upsert()      ~ addPoint(data, label, true)     (src/index.cpp:1052-1054)
 *       tsgpu_vec_delete()      ~ markDelete(label)               (src/index.cpp:7423)
 *       tsgpu_vec_get()         ~ getDataByLabel<float>(label)    (src/index.cpp:3355-3359)
 *       tsgpu_vec_count()       ~ getCurrentElementCount()        (src/index.cpp:1004-1006)
 *       tsgpu_vec_knn_batch()   ~ searchKnnCloserFirst(q,k,ef,f)  (src/index.cpp:3384-3386), exact
 *       tsgpu_vec_distances()   ~ process_results_bruteforce's per-id dist_func loop (src/index.cpp:3345-3374)
 *
 *   B3 hybrid — keyword pass, vector pass, reciprocal-rank fusion exactly as src/index.cpp:4036-4221:
 *       tsgpu_hybrid_search_batch()
 *
 * Error convention mirrors Option<T> (include/option.h): every call returns a code (0 = ok, else an
 * HTTP-like code as the reference uses: 400 bad request, 404 not found, 408 deadline, 500 device
 * failure, 501 valid request this library does not accelerate -> caller takes its CPU path for THAT
 * query) and tsgpu_last_error() returns the message for the calling thread. Never throws, never aborts.
 *
 * Threading: any number of request threads may call the search entry points concurrently on one
 * context (the reference calls the seam once per query from its request threads under a shared_lock on
 * Index::mutex, src/index.cpp:3488, src/http_server.cpp:827-832). Small concurrent calls (host outputs,
 * <= "batch_max_queries" queries) are COALESCED inside the library into one launch per round (micro-batcher,
 * csrc/tsgpu_batcher.h) and every caller gets exactly its own results; larger calls run on one of two
 * execution lanes (own stream + scratch each), so one batch is planned and uploaded while another runs.
 * tsgpu_commit() publishes a new immutable snapshot (RCU): a search keeps the snapshot it started on, a commit
 * never waits for searches and searches never wait for a commit. Everything else that mutates (term / vector
 * upserts, deletes, column_set) follows the unique_lock side of Index::mutex: externally serialised against
 * searches.
 */
#ifndef TSGPU_H
#define TSGPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSGPU_ABI_VERSION 6
#define TSGPU_MAX_DROPPED_TOKENS 4

/* limits of the accelerated path (anything beyond -> TSGPU_ERR_UNSUPPORTED for that query) */
#define TSGPU_MAX_QUERY_TOKENS 10   /* = WINDOW_SIZE, include/match_score.h:11 */
#define TSGPU_MAX_SORT_KEYS 3       /* include/topster.h:26 scores[3] */
#define TSGPU_MAX_TOPK 1024         /* Topster capacity max(fetch_size, 250), src/index.cpp:3506 */
#define TSGPU_DEFAULT_TOPSTER_SIZE 250 /* include/index.h:679 */
#define TSGPU_BLOCK_IDS 256         /* posting_t::MAX_BLOCK_ELEMENTS, include/posting.h:46 */

enum tsgpu_status {
    TSGPU_OK = 0,
    TSGPU_ERR_INVALID = 400,
    TSGPU_ERR_NOT_FOUND = 404,
    TSGPU_ERR_DEADLINE = 408,
    TSGPU_ERR_DEVICE = 500,
    TSGPU_ERR_UNSUPPORTED = 501,
    TSGPU_ERR_NO_MEMORY = 507
};

enum tsgpu_mem_kind { TSGPU_MEM_HOST = 0, TSGPU_MEM_DEVICE = 1 };

/* text_match_type_t, include/index.h (max_score default) */
enum tsgpu_match_type { TSGPU_MAX_SCORE = 0, TSGPU_MAX_WEIGHT = 1, TSGPU_SUM_SCORE = 2 };

/* what a sort_by slot reads (the sentinel maps of src/index.cpp:5722-5725, 5835-5836, 5865-5866) */
enum tsgpu_sort_kind {
    TSGPU_SORT_TEXT_MATCH = 0,      /* text_match_sentinel_value */
    TSGPU_SORT_SEQ_ID = 1,          /* seq_id_sentinel_value */
    TSGPU_SORT_INT64_COLUMN = 2,    /* sort_index[field]->find(seq_id), missing -> INT64_MIN */
    TSGPU_SORT_VECTOR_DISTANCE = 3  /* vector_distance_sentinel_value -> float_to_int64_t(d) */
};

/* vector_distance_type_t, include/field.h:92-95 */
enum tsgpu_metric { TSGPU_METRIC_IP = 0, TSGPU_METRIC_COSINE = 1 };

typedef struct tsgpu_ctx tsgpu_ctx;

/* ------------------------------------------------------------------ lifecycle */
int tsgpu_abi_version(void);
/* device_ordinal: HIP device index (one context per process per GPU). */
int tsgpu_create(int device_ordinal, tsgpu_ctx** out);
void tsgpu_destroy(tsgpu_ctx* ctx);
/* message of the last failing call on this thread ("" if none) */
const char* tsgpu_last_error(void);
/* run the library's kernels on a caller-owned hipStream_t (NULL = the context's own stream) */
int tsgpu_set_stream(tsgpu_ctx* ctx, void* hip_stream);
/* "doc_range_lo" / "doc_range_hi" (default 0 / 0 = the whole collection): the seq_ids [lo, hi) this context OWNS when it is a doc-range shard of a
 * group (tsgpu_group_wildcard_search_batch ranks only those; posting lists need no range: they hold what was fed).
 * tuning knobs (all optional): "kw_chunk_blocks" = driver posting blocks per keyword work item (default 0 = sized per batch; 1..256),
 * "kw_pair_blocks" = 1 (default): the find kernel serves two blocks of the shortest list per iteration (kw_find2_kernel; 0 = one),
 * "kw_mf_pipelined" = 1 (default): launches with multi-field queries run the PIPELINED find kernel (kw_find_mf2_kernel<., 2> when no query
 * of the launch has more than two query_by fields, <., 4> otherwise: the second token's lists of all fields merged block-wise through
 * double-buffered LDS tiles, requested one driver block ahead; 0 = kw_search_mf_kernel, block at a time); counter "kw_mf_pipelined_launches",
 * "kw_iddir_min_ids" (default 256) / "kw_iddir_density_div" (default 64) / "kw_iddir_budget_mb" (default 4096): posting lists of at least
 * max(min_ids, S / density_div) ids (S = the doc-id range the context's lists cover: num_docs, or a shard's range) carry an ID DIRECTORY in HBM — 8 bytes per 32 doc ids, {posting position, bits} — that answers
 * "is id x in the list, and where" (the probes of a query's third.. lists, of runs wider than the find kernel's tile, of the multi-field and
 * candidate kernels) with one load instead of two searches; built on the device by tsgpu_commit for the lists a write batch touched, within
 * the budget (longest lists first). min_ids = 0 switches them off (identical results). Any change takes effect at the next commit, which
 * re-packs the index. Counters: "kw_iddir_lists" (lists of the current snapshot that carry one), "kw_iddir_built" (built by commits so far),
 * "kw_two_kernels" = 1 (default): single-field queries run as a find kernel + a score kernel with hit records (seq_id + one
 * posting position per token) between them, 0: one fused kernel (identical results); "kw_hit_buffer_mb" = budget of that hit
 * buffer (default 20480; 16 bytes (<= 3 tokens) or 44 bytes per driver posting of the batch are reserved; a table of work items
 * that needs more than two buffer-sized groups runs fused);
 * "kw_sort_work" = 1 (default): work items launched heaviest first,
 * "vec_rows_per_slab" = base rows per k-NN workgroup slab (default: automatic), "vec_sample_tiles" = 128-row tiles of
 * the k-NN threshold sample (default 512), "vec_cand_cap" = candidate slots per query of the filtered pass (0 = auto),
 * "vec_ip_lanes" = 4 (default) / 8 / 16: the order every exact distance is summed in = the SIMD level hnswlib is compiled for in the
 * server (4 = SSE, what the reference's stock build flags give; 8 = -mavx; 16 = -mavx512f): hnswlib's InnerProductSpace accumulates
 * element i in lane i % lanes and adds the lanes left to right, so the bits of a distance depend on it;
 * "vec_prefilter" = 1 (default): bf16 bracket scan + exact fp32 re-score of the survivors, 0: fp32 MFMA scan of every row
 * (identical result sets either way), "vec_count_rescored" = 1: keep the vec_rescored_rows counter (costs one sync);
 * micro-batcher: "batch_max_queries" = calls with at most this many queries are coalesced with concurrent callers (default 64,
 * 0 = never), "batch_window_us" = how long a round's leader waits for the other threads that are inside the entry point to
 * park (default 10: a lane that is free should not idle; while every lane is busy callers keep parking and the rounds size
 * themselves), "batch_round_queries" = queries per coalesced round at most (default 1024);
 * "hybrid_overlap" = 1 (default): tsgpu_hybrid_search_batch runs its keyword pass on a second host thread and its own lane while
 * the vector pass runs (0: one after the other; identical results);
 * "vec_batch_post_window_us" (default 300) = a coalesced VECTOR round's leader, once the executor is free, waits this long for the
 * callers of the round that just finished to call again (a scan's cost hardly depends on the number of queries it serves);
 * "kw_merge_select_min" = from this many per-work-item Topsters per query the merge selects (threshold of the k-th largest of a
 * prefix union, then the candidates above it; both ordered by a tree of pairwise rank merges in LDS) instead of folding one list
 * after the other (default 2; 0 = always fold; identical results);
 * "kw_zero_copy_max_queries" = keyword batches with host output and at most this many queries have their merge kernel write the
 * result image straight into the lane's pinned host buffer, no device-to-host copy (default 256; 0 = always copy);
 * "kw_host_split_queries" (default 1000) / "kw_host_split_first_pct" (default 85) / "kw_host_split_tail_slices" (default 1; 2) /
 * "kw_host_split_device_plan" (default 1) = a keyword batch with HOST output of at least four times kw_host_split_queries queries is
 * served in two slices (85 % / 15 %), each on a lane and host thread of its own, enqueued in slice order: the large slice — planned on the
 * device — runs first, and its hit arrays cross PCIe while the small slice computes (10 000 queries: 8.1 -> 7.5 ms; identical results;
 * kw_host_split_queries = 0: never);
 * "kw_timing_min_queries" = keyword batches below this many queries record no phase events (tsgpu_last_timings then reports 0 ms
 * for them; default 64: the four marker packets cost a 1-query call ~20 us of its ~80; 0 = always record; coalesced rounds never
 * record);
 * host side: "plan_threads" (default 8) / "plan_parallel_min_queries" (default 2048) = the work table of a batch with at least
 * that many queries is planned in slices on the context's parked host threads, "fuse_threads" (default 32) = host threads of the
 * hybrid fusion, "blocking_sync_min_callers" (default 48) = from this many threads inside the keyword entry point a round is
 * awaited SLEEPING (most of the lane's usual wait, then event polls between 15 us naps) instead of with a spinning stream wait (the
 * request threads need the cores; hipEventSynchronize on a blocking-sync event still spins inside the runtime);
 * "hnsw_visited_hash" = 1 (default): an HNSW traversal keeps the ids it has visited in a per-query hash set (32-256 KB, whatever the
 * row count; up to 4096 queries traverse at once), 0: 16-bit tags per row and concurrent query, capped by
 * "hnsw_visited_max_gib" (default 64, 1..128: 2 bytes x rows x concurrent queries — 41 GB for 2048 queries at 10M rows) */
int tsgpu_set_option(tsgpu_ctx* ctx, const char* name, int64_t value);
/* introspection counters (tests / bench): "vec_overflow_rounds", "vec_prefilter_groups", "vec_prefilter_fallbacks",
 * "vec_rescored_rows", "kw_last_hit_groups" (find+score groups of the last keyword batch, 0 = fused), "kw_last_hit_records",
 * "batch_rounds" / "batch_coalesced_calls" (micro-batcher: rounds executed / calls they served; "gb_batch_rounds" / "gb_batch_coalesced_calls": of grouped calls), host phase totals in us over all
 * keyword batches ("kw_batches", "kw_plan_us", "kw_upload_us", "kw_launch_us", "kw_wait_us", "kw_book_us", "batch_exec_us",
 * "batch_scatter_us") and over all coalesced calls ("kw_queue_us" = parked -> its round starts, "kw_wake_us" = results ready ->
 * the caller runs again) */
int tsgpu_get_counter(tsgpu_ctx* ctx, const char* name, uint64_t* out);
/* bytes of HBM held by the context's index mirrors */
uint64_t tsgpu_device_bytes(tsgpu_ctx* ctx);

/* ------------------------------------------------------------------ keyword index mirror */
/* declare a query_by field; is_array = field is string[] (offset format of src/index.cpp:1351-1395) */
int tsgpu_field_create(tsgpu_ctx* ctx, uint32_t field_id, int is_array);

/* Replace the posting list of (field, term) with the DECODED content of its posting_list_t blocks
 * (include/posting_list.h:56-77): ids ascending; offset_index[i] = start of doc i's run in offsets[];
 * offsets in the reference's encoding (position+1, trailing 0 = last token, src/index.cpp:1323-1348).
 * n_ids == 0 removes the term. Host pointers. Takes effect at the next tsgpu_commit(). */
int tsgpu_term_upsert(tsgpu_ctx* ctx, uint32_t field_id, uint32_t term_id,
                      const uint32_t* ids, const uint32_t* offset_index, const uint32_t* offsets,
                      uint32_t n_ids, uint32_t n_offsets);

/* Bulk form of tsgpu_term_upsert for n_terms lists in CSR layout: list t owns
 * ids[ids_ptr[t] .. ids_ptr[t+1]) ; offset_index entries are absolute indices into offsets[] and list t's
 * offsets end at off_ptr[t+1]. Host pointers. */
int tsgpu_terms_load_csr(tsgpu_ctx* ctx, uint32_t field_id, uint32_t n_terms, const uint32_t* term_ids,
                         const uint64_t* ids_ptr, const uint32_t* ids, const uint64_t* offset_index,
                         const uint64_t* off_ptr, const uint32_t* offsets);

/* Single-document mutation, the calls the reference makes on an ART leaf's posting object while indexing / removing a document
 * (posting_t::upsert(obj, id, offsets) / posting_t::erase(obj, id), src/posting.cpp:247-330, from Index::index_field_in_memory and
 * Index::remove_field, under the unique_lock of Index::mutex): document `id` gets `offsets` (the reference's encoding) in the list of
 * (field, term) — inserted in id order, or its run replaced — / leaves the list. ONE block of the list changes; tsgpu_commit then
 * uploads only the changed blocks (+ the list's small descriptor arrays): a write batch costs O(changed blocks), not O(index). */
int tsgpu_posting_upsert(tsgpu_ctx* ctx, uint32_t field_id, uint32_t term_id, uint32_t id, const uint32_t* offsets, uint32_t n_offsets);
int tsgpu_posting_erase(tsgpu_ctx* ctx, uint32_t field_id, uint32_t term_id, uint32_t id);

/* Dense numeric sort column (the reference's sort_index[field], include/index.h:442): values[seq_id];
 * present == NULL means every seq_id < n has a value, else present[seq_id] != 0. mem = tsgpu_mem_kind. */
int tsgpu_column_set(tsgpu_ctx* ctx, uint32_t column_id, const int64_t* values, const uint8_t* present,
                     uint32_t n, int mem);

/* number of documents (num_seq_ids(), bounds the Topster capacity, src/index.cpp:3510) */
int tsgpu_set_num_docs(tsgpu_ctx* ctx, uint32_t num_docs);

/* Publish all pending posting-list changes as ONE new immutable snapshot (RCU): searches that started before keep the snapshot they
 * run on, searches never wait for a commit, a failing commit leaves the previous snapshot in place. Incremental: re-written blocks
 * and the touched lists' descriptors are appended at the tails of the device arenas and a new descriptor table is swapped in;
 * everything is re-packed (compaction) only at the first commit, when the tails run out of room, when the arenas' garbage (words no
 * list of the newest snapshot refers to) outweighs their live words (and exceeds option "index_compact_min_words", default 2^20), or
 * after tsgpu_set_option(ctx, "commit_full", 1). Cost of an incremental commit: the changed blocks + the descriptor table (48 B per
 * list, re-uploaded whole) + a copy of the term map when a term appeared or disappeared. Device memory of replaced snapshots is freed
 * here, by the committing thread, never by a search. Counters: "commit_last_us", "commit_last_uploaded_bytes", "commit_full_count",
 * "commit_incremental_count", "commit_compactions", "commit_failed_count", "index_used_words", "index_live_words". */
int tsgpu_commit(tsgpu_ctx* ctx);

/* introspection (tests): number of ids of a term, 0 if absent */
uint32_t tsgpu_term_num_ids(tsgpu_ctx* ctx, uint32_t field_id, uint32_t term_id);
/* decode a committed list back from the device format (tests: format round trip). Buffers sized by
 * tsgpu_term_num_ids / *n_offsets (call once with NULL buffers to size). */
int tsgpu_term_download(tsgpu_ctx* ctx, uint32_t field_id, uint32_t term_id, uint32_t* ids,
                        uint32_t* offset_index, uint32_t* offsets, uint32_t* n_offsets);

/* ------------------------------------------------------------------ keyword search (seam B1) */
typedef struct tsgpu_sort_by {
    uint8_t kind;      /* tsgpu_sort_kind */
    int8_t order;      /* 1 = DESC, -1 = ASC (sort_order[], src/index.cpp:5901-5903) */
    uint16_t column;   /* for TSGPU_SORT_INT64_COLUMN */
} tsgpu_sort_by;

typedef struct tsgpu_kw_query {
    /* one candidate-token combination, i.e. ONE call of search_across_fields (src/index.cpp:5385) */
    uint32_t n_tokens;                               /* query_tokens.size() */
    uint32_t term_ids[TSGPU_MAX_QUERY_TOKENS];       /* tokens absent from the index are skipped (src/index.cpp:5651-5655) */
    uint32_t n_fields;                               /* v1: 1 */
    uint32_t field_ids[4];
    int32_t field_weights[4];                        /* the_fields[i].weight (0..15) */
    uint8_t match_type;                              /* tsgpu_match_type */
    uint8_t prioritize_exact_match;
    uint8_t prioritize_token_position;
    uint8_t prioritize_num_matching_fields;
    uint32_t total_cost;                             /* sum(2*typo_cost + is_prefix), src/index.cpp:7233-7235 */
    uint32_t n_sort;
    tsgpu_sort_by sort[TSGPU_MAX_SORT_KEYS];
    uint32_t topster_size;                           /* Topster capacity, src/index.cpp:3506-3512 (0 = max(250, ...) by the library) */
    const uint32_t* excluded_ids;                    /* sorted, host; may be NULL */
    uint32_t n_excluded;
    const uint32_t* filter_ids;                      /* sorted, host; NULL = no filter */
    uint32_t n_filter;
    uint64_t deadline_us;                            /* absolute epoch us (system clock) after which search_cutoff is raised; 0 = none. A query
                                                        already late when its batch is planned reports 408 and no hits; one that runs out of
                                                        time on the device stops scanning (every work item checks the device clock every 16
                                                        driver blocks) and returns the PARTIAL hits found so far with status 0 and
                                                        search_cutoff = 1, like the reference (include/or_iterator.h:148-153) */
    uint32_t n_dropped;                              /* dropped_tokens.size() (the drop_tokens_threshold passes, src/index.cpp:5427-5464): tokens that */
    uint32_t dropped_term_ids[TSGPU_MAX_DROPPED_TOKENS]; /* take no part in the AND but are scored when the document holds them (compute_aggregated_score,
                                                        :5271-5290: their postings join the field's tokens after the query's own, query_len counts them);
                                                        n_tokens + n_dropped <= TSGPU_MAX_QUERY_TOKENS. Such queries take the general (per-candidate
                                                        probe) kernel. */
    /* synonym passes (the query is a synonym's expansion; src/index.cpp:5292-5294, 6989-6994, 7024-7060). All zero = not a synonym pass. */
    uint8_t is_synonym_query;
    uint8_t demote_synonym_match;
    uint8_t syn_orig_num_tokens_p1;                  /* syn_orig_num_tokens + 1 (0 = the reference's -1: none) */
    uint8_t orig_num_tokens;
} tsgpu_kw_query;

/* Results, structure-of-arrays, slot q*k_stride+i = i-th best hit of query q in Topster::sort() order
 * (descending (scores[0],scores[1],scores[2],key), include/topster.h:146-149,469-473). The host shim turns
 * each slot into KV{key, distinct_key=key, scores, match_score_index, text_match_score} and calls topster->add.
 * With mem = HOST the arrays text_match, vector_distance, match_score_index, num_matched and search_cutoff are OPTIONAL (NULL: not delivered;
 * text_match equals scores[match_score_index] whenever _text_match is a sort key); device outputs need every array. */
typedef struct tsgpu_hits {
    int mem;                 /* tsgpu_mem_kind of every pointer below */
    uint32_t k_stride;       /* slots per query (>= the largest topster_size in the batch) */
    uint64_t* keys;          /* [n_queries*k_stride] seq_id */
    int64_t* scores;         /* [n_queries*k_stride*3] */
    int64_t* text_match;     /* [n_queries*k_stride] aggregated text score (KV::text_match_score) */
    float* vector_distance;  /* [n_queries*k_stride] -1.0f unless set by the vector / hybrid path */
    int8_t* match_score_index; /* [n_queries*k_stride] */
    uint32_t* n_hits;        /* [n_queries] */
    uint64_t* num_matched;   /* [n_queries] num_keyword_matches (keyword) / hits returned (vector) */
    int32_t* status;         /* [n_queries] per-query tsgpu_status (501 -> run that query on the CPU path) */
    int32_t* search_cutoff;  /* [n_queries] 1 if the deadline passed (thread_local search_cutoff) */
} tsgpu_hits;

int tsgpu_keyword_search_batch(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_queries, tsgpu_hits* out);

/* Wildcard search, q = "*" without a vector query (Index::search_wildcard, src/index.cpp:6616-6818): every id of
 * queries[i].filter_ids (every seq_id < num_docs when n_filter == 0) minus excluded_ids is ranked by the sort keys alone — the
 * _text_match slot is the constant 100 the reference passes to compute_sort_scores — into a Topster of topster_size. Only
 * sort / topster_size / excluded_ids / filter_ids / deadline_us of the query are read. num_matched = ids ranked. */
int tsgpu_wildcard_search_batch(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_queries, tsgpu_hits* out);

/* Candidate-token combinations (Index::search_all_candidates, src/index.cpp:1794-1894). With prefix / typo
 * candidates the reference runs one search_across_fields pass per combination of candidate tokens over ONE Topster and ONE
 * id_buff. Here the combinations of user query g are combos[group_begin[g] .. group_begin[g+1]) in the reference's pass order
 * (each with its own term_ids and total_cost; at most 16 per group and 16 * out->k_stride <= 4096; group_begin is a host array
 * of n_groups + 1 entries); all of them run as one keyword batch and are folded on the device exactly like the shared Topster:
 * a key found by several passes keeps its greatest (scores[0..2]) — the later pass when equal (include/topster.h:392-406) —
 * and out holds the top topster_size of those in sort() order. out: [n_groups][k_stride];
 *   query_index [n_groups][k_stride] (nullable): KV::query_index of each hit = number of earlier passes of the group that
 *     matched anything (searched_queries.size() at the time of the pass, :5511, :5580-5585) — add the caller's base;
 *   found [n_groups] (nullable): all_result_ids_len = distinct ids matched by any pass (id_buff -> sort + unique + or_scalar);
 *     the ids themselves: tsgpu_candidates_result_ids(ctx, g, ...) until the next call. Needs n_groups * num_docs / 8 bytes
 *     (<= 8 GiB) of scratch;
 *   num_matched[g] = the LAST pass's count (the reference assigns num_keyword_matches per pass, :5553).
 * query_index / found live where out->mem says. A group with a failing combination reports that status and no hits. */
int tsgpu_keyword_search_candidates_batch(tsgpu_ctx* ctx, const tsgpu_kw_query* combos, const uint32_t* group_begin, uint32_t n_groups,
                                          tsgpu_hits* out, uint32_t* query_index, uint64_t* found);
uint64_t tsgpu_candidates_result_ids(tsgpu_ctx* ctx, uint32_t group, uint32_t* out_host, uint64_t cap);

/* Keyword search that also returns every matched id (id_buff -> all_result_ids, src/index.cpp:5549, 5565: facets, group-by and
 * `found` read them): *ids_out receives a list object that belongs to THIS call — safe with any number of concurrent callers —
 * holding, per query, the ascending ids that passed exclusion / filter. Free it with tsgpu_id_lists_free. */
typedef struct tsgpu_id_lists tsgpu_id_lists;
int tsgpu_keyword_search_batch_ids(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_queries, tsgpu_hits* out, tsgpu_id_lists** ids_out);
uint64_t tsgpu_id_lists_count(const tsgpu_id_lists* lists, uint32_t q);
const uint32_t* tsgpu_id_lists_ids(const tsgpu_id_lists* lists, uint32_t q);
void tsgpu_id_lists_free(tsgpu_id_lists* lists);

/* SINGLE-CALLER form of the same (tests, tools): all matched ids of the LAST keyword batch for query q, ascending. Only available
 * when tsgpu_keep_result_ids(ctx, 1) was set before the batch; such batches are never coalesced and always run on lane 0. The
 * state is context-global: with concurrent callers use tsgpu_keyword_search_batch_ids. Returns count; copies min(count, cap). */
int tsgpu_keep_result_ids(tsgpu_ctx* ctx, int keep);
uint64_t tsgpu_result_ids(tsgpu_ctx* ctx, uint32_t q, uint32_t* out_host, uint64_t cap);

/* ------------------------------------------------------------------ group_by: the DISTINCT Topster */
/* One pass of the reference's two-pass grouped search (Index::run_search, src/index.cpp:2488-2760) through search_across_fields with
 * group_limit != 0 (src/index.cpp:5511-5520, 5546-5549) — or Index::search_wildcard's grouped loop (:6736-6760) — into
 * Topster(capacity, distinct = group_limit, is_group_by_first_pass) (include/topster.h:266-296, add :321-466) and, for the second
 * pass, populate_result_kvs' grouped branch (src/index.cpp:8962-9011).
 *   column: a column set with tsgpu_column_set(ctx, column, values, NULL, n, mem) whose values[seq_id] are the bits of the uint64
 *     Index::get_distinct_id yields for the request's group_by fields (src/index.cpp:7100-7142: hash_combine over the fields' facet
 *     hashes; a document without any value: its seq_id, or 1 with group_missing_values) — the server computes it once per
 *     (collection, group_by fields, group_missing_values) from facet_index_v4's hash indexes (host helper: tsgpu_groupby_shim.h). A
 *     seq_id >= n has no value. (With SEVERAL group_by fields the reference's result for a document that lies beyond the last id of
 *     a later field's hash index depends on which documents the query visited before — its iterator is then already exhausted,
 *     :7104-7111 —; the column holds the value of a fresh iterator.)
 *   first_pass = 1: out holds ONE hit per group — the group's greatest KV — for the topster_size groups with the greatest such
 *     KVs, best first (the reference's heap holds the same KVs in heap-array order; its consumers read them as a set:
 *     Index::get_group_by_values, src/index.cpp:7144-7170). groups_count = Topster::getGroupsCount() = the LogLogBeta estimate over
 *     every distinct key of the pass (include/loglogbeta.h; the registers themselves on request: mergeGroupsCount).
 *   first_pass = 0: the groups populate_result_kvs returns, best head first; group r's KVs — the group Topster's content in sort()
 *     order, group_size[r] <= group_limit of them — occupy hit slots [r * group_limit, r * group_limit + group_size[r]) of the query;
 *     out->k_stride >= topster_size * group_limit. n_hits = the sum of the group sizes.
 *   group_found = groups_processed[distinct_key] (:5546-5549: the group's matched documents); groups_total = the exact number of
 *     distinct keys among the matched documents (not a reference quantity).
 *   wildcard = 1: q = "*" (only sort / topster_size / excluded_ids / filter_ids of the query are read, as in tsgpu_wildcard_search_batch).
 * `sort_by: _group_found` (the count-min sketch, include/topster.h:327-340), curated hits and Union_KV are not covered: 501.
 * out and gout are HOST arrays. ids_out (nullable): the matched ids of every query (all_result_ids; group_by_missing_value_ids = those
 * of them without a value: the caller knows which). The call holds the context's index lock like a search holds Index::mutex; small calls from
 * concurrent request threads (<= "batch_max_queries" queries each) are coalesced into one grouped batch like the plain keyword calls. */
#define TSGPU_MAX_GROUP_LIMIT 256
typedef struct tsgpu_group_by {
    uint32_t group_limit;            /* Topster's `distinct`, 1..TSGPU_MAX_GROUP_LIMIT */
    uint16_t column;                 /* distinct-key column */
    uint8_t first_pass;              /* is_group_by_first_pass */
    uint8_t group_missing_values;    /* distinct key of a seq_id beyond the column: 1 (true) or the seq_id (false) */
    uint8_t wildcard;                /* the query is q = "*" */
    uint8_t pad[3];
} tsgpu_group_by;
typedef struct tsgpu_grouped_hits {
    uint32_t g_stride;               /* group slots per query (>= the largest topster_size of the batch) */
    uint32_t* n_groups;              /* [n_queries] groups returned */
    uint64_t* distinct_key;          /* [n_queries * g_stride] */
    uint32_t* group_size;            /* [n_queries * g_stride] KVs of the group in out (first pass: 1) */
    uint32_t* group_found;           /* [n_queries * g_stride] */
    uint64_t* groups_total;          /* [n_queries], nullable */
    uint64_t* groups_count;          /* [n_queries], nullable: first pass = LogLogBeta::cardinality(); second pass 0 */
    uint8_t* loglog_registers;       /* [n_queries * 16384], nullable: the first pass' sketch registers */
} tsgpu_grouped_hits;
int tsgpu_keyword_search_grouped_batch(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, const tsgpu_group_by* groups, uint32_t n_queries,
                                       tsgpu_hits* out, tsgpu_grouped_hits* gout, tsgpu_id_lists** ids_out);
/* The same over candidate-token combinations (Index::search_all_candidates with group_limit != 0, src/index.cpp:1794-1894 over :5511-5549): the combinations of user
 * query u — combos[group_begin[u] .. group_begin[u + 1]), at most 16, in the reference's pass order, groups[u] its group_by — are one search_across_fields pass each
 * over ONE distinct Topster and ONE groups_processed, folded on the device:
 *   first pass : a group's KV = the greatest over all combinations' KVs of its documents (an equal KV of a later combination replaces the earlier one,
 *     include/topster.h:392-406: its query_index is the later pass'); group_found counts every add, as the reference's first pass does; the sketch sees every key;
 *   second pass: a document met by several combinations counts ONCE (group_doc_seq_ids -> ret == 2, src/index.cpp:5546-5549) and contributes its greatest KV (the
 *     later combination's on ties) to its group's Topster.
 * query_index [n_user * out->k_stride] (nullable): KV::query_index per hit slot = the earlier combinations of the user query that matched anything (add the caller's
 * searched_queries.size()); num_matched = the LAST combination's count; ids_out = per user query the ascending union of its combinations' ids (all_result_ids).
 * A user query with a failing combination reports that status and nothing else. Not coalesced across callers (a call is a batch already). */
int tsgpu_keyword_search_grouped_candidates_batch(tsgpu_ctx* ctx, const tsgpu_kw_query* combos, const uint32_t* group_begin, const tsgpu_group_by* groups, uint32_t n_user,
                                                  tsgpu_hits* out, tsgpu_grouped_hits* gout, uint32_t* query_index, tsgpu_id_lists** ids_out);

/* ------------------------------------------------------------------ facet counting over matched ids */
/* The hash-index branch of Index::do_facets (src/index.cpp:1659-1771): for every matched id (ascending, e.g. the id list of
 * tsgpu_keyword_search_batch_ids) the field's facet hash index (facet_index_v4's hash index, a posting list seq_id -> value hashes)
 * is looked up and result_map[hash] is bumped once per DISTINCT hash of the document; doc_id / array_pos = the last (greatest)
 * document that carried the value and the value's position in it.
 *   tsgpu_facet_set: the mirror of one field's facet hash index, CSR by seq_id: doc_ptr[d] .. doc_ptr[d+1] = the hashes of document d
 *     in field order (empty = no value; documents >= n_docs have none). Host arrays; replaces the field's mirror.
 *   tsgpu_facet_count_batch: result_ids[q] = host array of n_result_ids[q] ascending ids. sample_mod > 1 = estimate_facets (only the
 *     ids at positions i % sample_mod == 0, :1683-1687); allowed_hashes (sorted, NULL = all) = fquery_hashes of a facet query (:1742).
 *     out: [n_queries][cap] in ascending hash order; n_values[q] = distinct values found (may exceed cap: the first cap are returned).
 * Facet stats (tsgpu_facet_stats_batch) and the value-index ("intersect") branch (tsgpu_facet_value_set / tsgpu_facet_value_count_batch) follow below.
 * The facets of a grouped search (hash_groups: tsgpu_facet_count_grouped_batch) and range facets (tsgpu_facet_range_count_batch) follow them.
 * Left to the caller, being lookups by what these calls return: sort_field_val (the sort-index value of the returned doc_id, :1765-1767) and
 * hash_tokens (fquery_hashes.at(hash), :1761-1764). */
typedef struct tsgpu_facet_counts {
    uint32_t cap;            /* slots per query */
    uint32_t* hash;          /* [n_queries * cap] facet value hash */
    uint32_t* count;         /* facet_count_t::count */
    uint32_t* doc_id;        /* facet_count_t::doc_id */
    uint32_t* array_pos;     /* facet_count_t::array_pos */
    uint32_t* n_values;      /* [n_queries] */
} tsgpu_facet_counts;
int tsgpu_facet_set(tsgpu_ctx* ctx, uint32_t facet_field_id, const uint64_t* doc_ptr, const uint32_t* hashes, uint32_t n_docs);
int tsgpu_facet_count_batch(tsgpu_ctx* ctx, uint32_t facet_field_id, const uint32_t* const* result_ids, const uint64_t* n_result_ids, uint32_t n_queries,
                            uint32_t sample_mod, const uint32_t* allowed_hashes, uint32_t n_allowed, tsgpu_facet_counts* out);

/* facets of a GROUPED search (group_limit != 0, src/index.cpp:1747-1749, 1756-1758): instead of counting documents the walk records
 * hash_groups[value].emplace(distinct_id) and a value's count becomes the number of groups it was seen in (:4455-4458; hash_groups holds
 * uint32_t: the distinct id is truncated, include/field.h:791). group_column = the distinct-id column of tsgpu_keyword_search_grouped_batch
 * (get_distinct_id per document; beyond its length: 1 with group_missing_values, else the seq_id). Everything else as tsgpu_facet_count_batch. */
int tsgpu_facet_count_grouped_batch(tsgpu_ctx* ctx, uint32_t facet_field_id, const uint32_t* const* result_ids, const uint64_t* n_result_ids, uint32_t n_queries,
                                    uint32_t sample_mod, const uint32_t* allowed_hashes, uint32_t n_allowed, uint32_t group_column, int group_missing_values,
                                    tsgpu_facet_counts* out);

/* range facets (a_facet.is_range_query in the same walk, src/index.cpp:1738-1750): per result document the facet hash index holds, once per
 * DISTINCT hash of the document, the field's sort-index value (value_column: dense int64, INT64_MAX where the sort index has no entry —
 * get_doc_val_from_sort_index, :1470-1482; float fields hold Index::float_to_int64_t keys like the sort index) is looked up in facet_range_map
 * (facet::get_range, include/field.h:820-838): range r = [range_lower[r], range_upper[r]) with range_upper strictly ascending (the map's key);
 * the first range whose upper bound is greater than the value is taken if the value reaches its lower bound. counts[q * n_ranges + r] =
 * result_map[range_upper[r]].count; 0 = the range is not in result_map. group_column != TSGPU_NO_COLUMN: the grouped form (count = number of
 * groups, sets keyed by the range id's low 32 bits like hash_groups). */
#define TSGPU_NO_COLUMN 0xFFFFFFFFu
int tsgpu_facet_range_count_batch(tsgpu_ctx* ctx, uint32_t facet_field_id, uint32_t value_column, const int64_t* range_upper, const int64_t* range_lower, uint32_t n_ranges,
                                  const uint32_t* const* result_ids, const uint64_t* n_result_ids, uint32_t n_queries, uint32_t sample_mod,
                                  uint32_t group_column, int group_missing_values, uint32_t* counts);

/* numeric facet stats of the same walk (should_compute_stats, src/index.cpp:1730-1741 -> compute_facet_stats :1430-1460): every
 * (document, distinct hash) contributes its VALUE — the hash itself for int32 fields, its bits as a float for float fields, the
 * fhash_int64_map entry (sorted int64_map_hashes -> int64_map_values; a missing hash = INT64_MAX) for int64 fields. min / max / count
 * are exact; fvsum is exact (= the reference's double accumulation) for integer fields while count * max|value| < 2^53 (sum_exact),
 * and for float fields equals the reference's in-order sum up to double rounding (sum_exact = 0). Untouched stats keep the
 * reference's initial values (include/field.h:765-770). */
#define TSGPU_FACET_INT32 0
#define TSGPU_FACET_INT64 1
#define TSGPU_FACET_FLOAT 2
typedef struct tsgpu_facet_stats { double fvmin, fvmax, fvsum; uint64_t fvcount; int32_t sum_exact; int32_t pad; } tsgpu_facet_stats;
int tsgpu_facet_stats_batch(tsgpu_ctx* ctx, uint32_t facet_field_id, int value_type, const uint32_t* const* result_ids, const uint64_t* n_result_ids,
                            uint32_t n_queries, uint32_t sample_mod, const uint32_t* int64_map_hashes, const int64_t* int64_map_values, uint32_t n_map,
                            tsgpu_facet_stats* out);

/* value-index branch of do_facets ("Using intersection to find facets", src/index.cpp:1596-1657 -> facet_index_t::intersect,
 * src/facet_index.cpp:230-353). tsgpu_facet_value_set mirrors facet_index_v4's fvalue_seq_ids: value v owns the ascending seq_ids
 * seq_ids[value_ptr[v] .. value_ptr[v+1]) and its total count; values are given in the reference's visiting order (counter_list:
 * by count, ties as the multiset holds them). A count call visits the values in that order (or in `order`, a permutation: the
 * alphabetical walks of sort_by _alpha), counts |ids(v) ∩ result_ids| exactly — or, with estimate_facets and more than 300 ids, with
 * the reference's strided walk (src/id_list.cpp:725-766); is_wildcard_no_filter_query: the stored total — and returns the first
 * max_facets values with a non-zero count: value_index / count / doc_id (= the value's first seq_id) [n_queries][cap], n_found. */
typedef struct tsgpu_facet_value_counts { uint32_t cap; uint32_t* value_index; uint32_t* count; uint32_t* doc_id; uint32_t* n_found; } tsgpu_facet_value_counts;
int tsgpu_facet_value_set(tsgpu_ctx* ctx, uint32_t facet_field_id, const uint64_t* value_ptr, const uint32_t* seq_ids, const uint32_t* total_counts, uint32_t n_values);
int tsgpu_facet_value_count_batch(tsgpu_ctx* ctx, uint32_t facet_field_id, const uint32_t* const* result_ids, const uint64_t* n_result_ids, uint32_t n_queries,
                                  uint32_t max_facets, int is_wildcard_no_filter_query, int estimate_facets, uint32_t facet_sample_interval, const uint32_t* order,
                                  tsgpu_facet_value_counts* out);

/* ------------------------------------------------------------------ vector index (seam B2) */
int tsgpu_vec_create(tsgpu_ctx* ctx, uint32_t vec_field_id, uint32_t dim, int metric, uint64_t capacity_hint);
/* addPoint: cosine fields are L2-normalised on insert like src/index.cpp:1049-1052. data: [n][dim] fp32. */
int tsgpu_vec_upsert(tsgpu_ctx* ctx, uint32_t vec_field_id, const uint64_t* labels, const float* data,
                     uint32_t n, int mem);
int tsgpu_vec_delete(tsgpu_ctx* ctx, uint32_t vec_field_id, uint64_t label);
/* TSGPU_ERR_NOT_FOUND where getDataByLabel would throw */
int tsgpu_vec_get(tsgpu_ctx* ctx, uint32_t vec_field_id, uint64_t label, float* out_host);
uint64_t tsgpu_vec_count(tsgpu_ctx* ctx, uint32_t vec_field_id);

/* exact k nearest by dist = 1 - <q,x> (cosine: q normalised first, src/index.cpp:3381-3384), closest first,
 * ties -> smaller label first. allow_ids (sorted, NULL = all) plays VectorFilterFunctor (include/index.h:325-354).
 * Q: [n_q][dim] fp32 (mem_q). Outputs [n_q][k] (mem_out), n_out[n_q] = hits written per query. */
int tsgpu_vec_knn_batch(tsgpu_ctx* ctx, uint32_t vec_field_id, const float* Q, int mem_q, uint32_t n_q, uint32_t k,
                        const uint32_t* allow_ids, uint32_t n_allow, const uint32_t* excluded_ids, uint32_t n_excluded,
                        float* dist_out, uint64_t* label_out, uint32_t* n_out, int mem_out);

/* HNSW graph search (hnswlib::HierarchicalNSW<float>::searchKnnCloserFirst(q, k, ef, filter) of the Typesense fork, call site
 * src/index.cpp:3376-3445). The graph is a MIRROR of the server's hnswlib index (its construction stays in the reference's
 * indexing path): rows of the field = hnswlib internal ids = insertion order; link0[i] = (count, up to 2M neighbour ids) of
 * level 0 (hnswlib get_linklist0), the lists of node i for levels 1..level(i) are upper_links[(upper_ptr[i] + level - 1)] =
 * (count, up to M ids) (hnswlib linkLists_), maxlevel / enterpoint = maxlevel_ / enterpoint_node_. Host arrays; M <= 31.
 * Re-load after the rows change. */
int tsgpu_vec_hnsw_load(tsgpu_ctx* ctx, uint32_t vec_field_id, uint32_t M, int32_t maxlevel, uint32_t enterpoint, const uint32_t* link0,
                        const uint64_t* upper_ptr, const uint32_t* upper_links, uint32_t n);
/* The graph BUILT inside the library — hnswlib's incremental addPoint (level draw from default_random_engine(seed), mult = 1 / ln M,
 * greedy descent, ef_construction-bounded beam per layer, getNeighborsByHeuristic2, reverse links, markDelete), the way the reference
 * inserts (src/index.cpp:1002-1075; ctor arguments include/index.h:365-367: M 16, ef_construction 200, seed 100) — so that the B2
 * typedef swap leaves a graph to search. From this call on every NEW label given to tsgpu_vec_upsert is inserted into
 * the graph as well (rows already in the field are inserted first, in row order); tsgpu_vec_delete = markDelete; the graph search uploads
 * the lists when they changed. n_threads: host threads inserting the rows of ONE upsert call concurrently with hnswlib's locking
 * (1 = the sequential algorithm, deterministic: equal, link for link, to the oracle's restatement; the reference itself indexes on four
 * threads). Round 5: the reference's addPoint(vec, seq_id, replace_deleted = true) on an index built with allow_replace_deleted = true
 * (include/index.h:367) is followed — tsgpu_vec_upsert of a LIVE label runs hnswlib's updatePoint on its row; of any other label while deleted
 * rows exist RE-USES the most recently deleted row (the label moves there, unmarkDeleted, updatePoint); only without one the row is appended —
 * so the graph stays searchable through updates (Typesense's update = markDelete + addPoint lands in the document's own row). Two choices
 * hnswlib leaves to std::unordered_set are fixed (which deleted slot; candidate order on equal distances): csrc/tsgpu_hnsw_build.h.
 * PARITY UNPINNED (hnswlib is not in the reference tree). M <= 31. */
int tsgpu_vec_hnsw_enable(tsgpu_ctx* ctx, uint32_t vec_field_id, uint32_t M, uint32_t ef_construction, uint32_t seed, uint32_t n_threads);
/* BULK construction of the graph over the rows the field holds NOW, on the device — what a restart or an import of a collection needs (the reference re-runs
 * addPoint per document from its indexing threads, src/index.cpp:1002-1075: minutes to hours at BASELINE config 3's size). Levels as hnswlib draws them
 * (default_random_engine(seed), label order); the rows with level >= 2 (one in M^2) and the first seed_min rows (0 = 1024) are inserted by the sequential
 * algorithm on n_threads host threads; the rest (levels 0 and 1) in batches of at most max_batch (0 = 65 536) and a sixteenth of what is linked: per row and
 * layer the ef_construction beam on the device, hnswlib's neighbour heuristic, reverse links applied per node for the whole batch (csrc/vec_hnsw_build.hip.h).
 * 10M x 768 rows, M 16, ef_construction 200: about a minute on one MI355X (the row-by-row insertion: 8 K rows/s on 16 host threads). Deterministic for
 * n_threads = 1 (oracle: hnsw_graph_t::bulk_build). The graph lives on the device like a loaded mirror (tsgpu_vec_hnsw_search_batch serves it,
 * tsgpu_vec_hnsw_export reads it back); rows added afterwards need a new build. PARITY UNPINNED like the search. M <= 31, ef_construction <= 1024. */
typedef struct tsgpu_hnsw_build_info {
    uint32_t n, n_seed, n_batches, unlinked;         /* rows, rows inserted on the host, device batches, rows whose beam overflowed (left without links; 0 in practice) */
    int32_t maxlevel; uint32_t enterpoint;
    double seed_seconds, device_seconds, search_seconds, link_seconds;      /* host insertion of the seed set; the batches: their beams, their links */
} tsgpu_hnsw_build_info;
int tsgpu_vec_hnsw_build(tsgpu_ctx* ctx, uint32_t vec_field_id, uint32_t M, uint32_t ef_construction, uint32_t seed, uint32_t n_threads, uint32_t seed_min, uint32_t max_batch,
                         tsgpu_hnsw_build_info* info);
/* the built graph in tsgpu_vec_hnsw_load's flat form (tests, persistence). info = {n, maxlevel, enterpoint, M}; *n_upper = upper lists.
 * Arrays may be NULL (sizes only): levels[n], link0[n][1 + 2M], upper_ptr[n + 1], upper_links[n_upper][1 + M]. */
int tsgpu_vec_hnsw_export(tsgpu_ctx* ctx, uint32_t vec_field_id, int32_t info[4], uint32_t* levels, uint32_t* link0, uint64_t* upper_ptr,
                          uint32_t* upper_links, uint64_t* n_upper);
/* up to k (distance, label) per query, closest first, found by greedy descent + the ef-bounded best-first search of layer 0
 * (max(ef, k) candidates). functor_present: the caller passes a filter functor (Typesense always does) — it selects hnswlib's
 * stricter stop rule; allow_ids (sorted, NULL = all) / excluded_ids / deleted labels are what the functor and isMarkedDeleted
 * reject. One wavefront per query, up to 4 096 queries in flight (16-bit visited tags per query slot, 16 GiB at most); the heaps
 * live in LDS sized by max(ef, k) (<= 128 / 512 / 1024; max(ef, k) > 1024 -> 501), a batch in which a candidate heap outgrows a small
 * size runs again on the largest. n_out[q] == 0xFFFFFFFF: the candidate heap outgrew even that (4 096 entries) — run the query
 * with tsgpu_vec_knn_batch. Counters "hnsw_last_expansions" / "hnsw_last_distances": layer-0 totals of the last batch. */
int tsgpu_vec_hnsw_search_batch(tsgpu_ctx* ctx, uint32_t vec_field_id, const float* Q, int mem_q, uint32_t n_q, uint32_t k, uint32_t ef,
                                int functor_present, const uint32_t* allow_ids, uint32_t n_allow, const uint32_t* excluded_ids,
                                uint32_t n_excluded, float* dist_out, uint64_t* label_out, uint32_t* n_out, int mem_out);

/* distances of one query to explicit labels (flat scan over filter ids, src/index.cpp:3345-3374);
 * missing labels get NaN (the reference `continue`s on the throw). Host pointers. */
int tsgpu_vec_distances(tsgpu_ctx* ctx, uint32_t vec_field_id, const float* q, const uint64_t* labels, uint32_t n,
                        float* dist_out);

/* ONE pair on the host: space->get_dist_func()(a, b, &dim) of hnswlib's InnerProductSpace = 1 - <a, b> (src/index.cpp:3365, :5842,
 * :8868), summed in the order of the SIMD level hnswlib is compiled for in the server: simd_lanes = 4 (SSE: the reference's stock
 * flags, CMakeLists.txt:8 / BUILD:81-88 pass no -march), 8 (AVX) or 16 (AVX-512); anything else = 4. The same value as option
 * "vec_ip_lanes" makes this function, tsgpu_vec_distances and the k-NN entry points return the same bits. */
float tsgpu_ip_distance(const float* a, const float* b, uint32_t dim, int simd_lanes);

/* pure vector search (q="*", src/index.cpp:3645-3732): the vector branch of Index::search with BOTH of its sub-branches, then
 * abs() for cosine / distance_threshold / sort scores / Topster order. One parameter block per batch (the queries share the filter).
 *   FLAT branch (filter_by_provided && n_filter < flat_search_cutoff, :3664-3665 -> process_results_bruteforce :3345-3374): EVERY filter id
 *     that has a vector gets its exact distance and goes to the Topster — no k cut; the excluded ids are not consulted (the reference
 *     does not consult them there either); ties are the Topster's (default sort: larger seq_id first). num_matched = ids the threshold
 *     kept = `found`. Device: one distance matrix [n_q][n_filter] + the wildcard ranking kernels (LDS Topster per 16K-id work item).
 *   k-cut branch (otherwise, :3666-3670 -> process_results_hnsw_index): the exact k nearest among the ids the VectorFilterFunctor passes
 *     (filter ids minus excluded ids; include/index.h:325-354), then the Topster. num_matched = hits kept (<= k).
 *   query_doc_given (`vec:([], id: X)`): Q holds X's stored vector; k grows by one when X passes the functor (:3651-3654) and X itself is
 *     left out (:3686). */
typedef struct tsgpu_vec_query {
    uint32_t k;                     /* vector_query.k (0 -> fetch_size) */
    uint32_t fetch_size;
    float distance_threshold;       /* FLT_MAX = none */
    uint32_t n_sort;
    tsgpu_sort_by sort[TSGPU_MAX_SORT_KEYS];
    uint32_t topster_size;          /* 0 = the reference's: max(fetch_size, 250) capped by n_filter (else the collection size), :3506-3512 */
    /* ---- ABI 4 ---- */
    uint32_t filter_by_provided;    /* a filter_by clause exists; filter_ids = the seq_ids it matched (sorted, host) */
    const uint32_t* filter_ids;
    uint32_t n_filter;
    uint32_t n_excluded;
    const uint32_t* excluded_ids;   /* sorted, host; hidden / curated ids (k-cut branch only) */
    uint64_t flat_search_cutoff;    /* vector_query.flat_search_cutoff (default 0: never flat), include/vector_query_ops.h:13 */
    uint32_t query_doc_given;
    uint32_t query_seq_id;
} tsgpu_vec_query;
int tsgpu_vector_search_batch(tsgpu_ctx* ctx, uint32_t vec_field_id, const tsgpu_vec_query* params,
                              const float* Q, int mem_q, uint32_t n_q, tsgpu_hits* out);
/* the same + all_result_ids of every query (sorted seq_ids that went to the Topster, :3727-3732; facets and `found` read them):
 * *ids_out is a list object the caller frees with tsgpu_id_lists_free (tsgpu_id_lists_ids / tsgpu_id_lists_count per query) */
int tsgpu_vector_search_batch_ids(tsgpu_ctx* ctx, uint32_t vec_field_id, const tsgpu_vec_query* params,
                                  const float* Q, int mem_q, uint32_t n_q, tsgpu_hits* out, tsgpu_id_lists** ids_out);

/* ------------------------------------------------------------------ hybrid (B3) */
typedef struct tsgpu_hybrid_params {
    uint32_t k;                     /* vector_query.k (0 -> max(fetch_size, 100), src/index.cpp:4060-4063) */
    uint32_t fetch_size;
    float alpha;                    /* VECTOR_SEARCH_WEIGHT, default 0.3 (include/vector_query_ops.h:19) */
    float distance_threshold;
    uint32_t rerank_hybrid_matches; /* search_params->rerank_hybrid_matches: Index::compute_aux_scores (src/index.cpp:8793-8923) after the fusion — hits
                                     * found by one side only get the other side's score (text_match of the document for the query's tokens; exact
                                     * distance by label), then every hit is re-ranked on both and re-fused. tsgpu_hybrid_search_batch only. */
} tsgpu_hybrid_params;
/* Q: [n_queries][dim] host or device; queries[i] pairs with Q[i]. Output = Topster after fusion + sort. */
int tsgpu_hybrid_search_batch(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t vec_field_id,
                              const tsgpu_hybrid_params* params, const float* Q, int mem_q,
                              uint32_t n_queries, tsgpu_hits* out);

/* Index::compute_aux_scores' text half (src/index.cpp:8800-8846): the aggregated text_match score (compute_aggregated_score with
 * total_cost 0) of GIVEN documents for a query's tokens — every token's lists are positioned on the document (skip_to), the tokens it
 * holds are scored per field, a document holding none scores 0. Item i = (queries[item_query[i]], document item_seq_id[i]);
 * scores_out[n_items]. Host arrays. */
int tsgpu_keyword_aux_scores(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_queries, const uint32_t* item_query,
                             const uint32_t* item_seq_id, uint32_t n_items, int64_t* scores_out);

/* Fusion step alone (src/index.cpp:4094-4211) on already-computed results, e.g. after a shard merge where ranks must be
 * global: kw_hits = keyword Topster content per query in sort() order, knn_* = [n_queries][knn_k] nearest neighbours
 * (closest first) with knn_cnt[q] valid entries; metric = tsgpu_metric of the vector field. Host arrays only. */
int tsgpu_hybrid_fuse_batch(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, const tsgpu_hybrid_params* params, int metric,
                            const tsgpu_hits* kw_hits, const float* knn_dist, const uint64_t* knn_labels, const uint32_t* knn_cnt,
                            uint32_t knn_k, uint32_t n_queries, tsgpu_hits* out);

/* ------------------------------------------------------------------ multi-GPU (doc-range shards) */
/* Merge G per-shard hit lists (already gathered, e.g. by an RCCL all-gather) into the global Topster
 * order. in[g] are host-resident tsgpu_hits for the same n_queries; key_offset[g] is added to shard g's keys
 * (0 if shards keep global seq_ids). Pure host code, exact (same comparator as include/topster.h:146-149). */
int tsgpu_merge_shard_hits(const tsgpu_hits* in, const uint64_t* key_offset, uint32_t n_shards,
                           uint32_t n_queries, uint32_t k, tsgpu_hits* out);

/* The same merge on the GPU, for hit lists that an RCCL all-gather left in device memory: every array of `gathered` is laid
 * out [shard][query][gathered->k_stride] (n_hits / num_matched: [shard][query]); text_match / vector_distance /
 * match_score_index may be NULL on either side. n_shards * gathered->k_stride <= 4096. Keys must be global seq_ids. */
int tsgpu_merge_shard_hits_device(tsgpu_ctx* ctx, const tsgpu_hits* gathered, uint32_t n_shards, uint32_t n_queries,
                                  uint32_t k, tsgpu_hits* out);

/* ------------------------------------------------------------------ multi-GPU group: the shard exchange behind the C-ABI */
/* G contexts, one per GPU, each mirroring the postings / sort columns / vectors of ITS seq_id range (global seq_ids kept). A group
 * call scores the whole batch on every member, exchanges the per-member top-k ONCE (RCCL ncclAllGather over xGMI) and merges exactly
 * (Topster order, include/topster.h:146-149; k-NN: distance, then label); hybrid fuses AFTER the merge (src/index.cpp:4036-4221: the
 * reciprocal ranks are ranks in the global lists). SURVEY §8(e), BASELINE config 5. The reference has no counterpart. */
typedef struct tsgpu_group tsgpu_group;
#define TSGPU_XCHG_RCCL 0   /* ncclAllGather on the members' streams; librccl.so.1 is resolved with dlopen at group creation */
#define TSGPU_XCHG_COPY 1   /* device-to-device copies into member 0 (local form only; members may share a device) */
#define TSGPU_XCHG_HOST 2   /* rank form over the CALLER's collectives on host memory (tsgpu_group_create_rank_host) */
/* ONE process owns all members (the C++ server): members run on their own host threads inside every group call */
int tsgpu_group_create_local(tsgpu_ctx* const* members, uint32_t n_members, int transport, tsgpu_group** out);
/* one process per GPU: rank 0 calls tsgpu_group_unique_id, the launcher broadcasts the 128 bytes, every rank joins with its context */
int tsgpu_group_unique_id(uint8_t id[128]);
int tsgpu_group_create_rank(tsgpu_ctx* ctx, const uint8_t id[128], uint32_t rank, uint32_t n_ranks, tsgpu_group** out);
/* Rank form over the caller's own channel (MPI, gloo, the server's RPC between nodes — where there is no xGMI): the exchange blocks are
 * staged through pinned host memory and handed to two callbacks with the semantics of ncclAllGather / ncclAllToAll on HOST buffers:
 *   all_gather(user, send, recv, bytes): recv[r * bytes .. (r+1) * bytes) = rank r's `send` (bytes each), on every rank;
 *   all_to_all(user, send, recv, bytes): recv[j * bytes ..) = the slice [rank * bytes ..) of rank j's `send` (n_ranks slices of `bytes`).
 * Both return 0 on success, are called from the thread that made the group call, and in the same order on every rank. Ranks need not own
 * a GPU each (two ranks may share a device). The struct is copied. Same blocks, merge kernels and results as the RCCL transport.
 * EVERY rank-form call (all transports): the ranks must pass the same n_queries, k, k_stride, options and the same SET of optional output
 * arrays; before any data collective they exchange {return code of the local phase, signature of those arguments}: a rank whose shard
 * failed makes the call fail on every rank (nobody is left inside a collective), differing arguments fail with 400 everywhere. */
typedef struct tsgpu_host_collectives {
    void* user;
    int (*all_gather)(void* user, const void* send, void* recv, size_t bytes_per_rank);
    int (*all_to_all)(void* user, const void* send, void* recv, size_t bytes_per_slice);
} tsgpu_host_collectives;
int tsgpu_group_create_rank_host(tsgpu_ctx* ctx, const tsgpu_host_collectives* coll, uint32_t rank, uint32_t n_ranks, tsgpu_group** out);
void tsgpu_group_destroy(tsgpu_group* g);        /* (the member contexts stay the caller's) */
uint32_t tsgpu_group_size(const tsgpu_group* g);
/* global top-k of every query, Topster order; `out` = host memory, or device memory of member 0 / of this rank (then keys, scores,
 * text_match, n_hits, num_matched, status are filled). k <= out->k_stride, members * k <= 4096; num_matched = sum over the shards;
 * a query's list never exceeds its own Topster capacity (topster_size, src/index.cpp:3506-3512). */
int tsgpu_group_keyword_search_batch(tsgpu_group* g, const tsgpu_kw_query* queries, uint32_t n_queries, uint32_t k, tsgpu_hits* out);
/* Candidate-token combinations over the shards (Index::search_all_candidates, src/index.cpp:1794-1894 — the reference's default `prefix = true` request):
 * tsgpu_keyword_search_candidates_batch's arguments and results with `out`, query_index ([n_groups][k_stride]) and found ([n_groups]) in host memory or in device
 * memory of member 0 / of this rank. Every shard folds its own passes (the fold is per key and a document lives in one shard), the folded Topsters take the
 * keyword exchange (top k per user query, num_matched = the last pass's counts added up, found = the union counts added up); KV::query_index — the earlier passes
 * that matched ANYTHING, :5511, :5580-5585 — is taken from the OR of the shards' pass masks (one more all-gather of 16 bytes per user query). At most 16
 * combinations per user query; the replicas form answers on one member; 501 with "kw_own_slice_only"; tsgpu_candidates_result_ids is per member (the ids of ITS shard). */
int tsgpu_group_keyword_search_candidates_batch(tsgpu_group* g, const tsgpu_kw_query* combos, const uint32_t* group_begin, uint32_t n_groups, uint32_t k, tsgpu_hits* out,
                                                uint32_t* query_index, uint64_t* found);
/* Wildcard search (q = "*", Index::search_wildcard, src/index.cpp:6616-6818) over the shards: tsgpu_wildcard_search_batch's queries; every member ranks the ids it
 * OWNS — tsgpu_set_option(member, "doc_range_lo" / "doc_range_hi", ...) when the shard is loaded (hi exclusive; without them a context ranks every seq_id below
 * num_docs, which is what a single GPU wants) — and the per-shard Topsters take the keyword exchange; num_matched = the ids ranked, added up. 400 when a member of
 * a group of several has no range. */
int tsgpu_group_wildcard_search_batch(tsgpu_group* g, const tsgpu_kw_query* queries, uint32_t n_queries, uint32_t k, tsgpu_hits* out);
/* group_by over doc-range shards: tsgpu_keyword_search_grouped_batch (one combination per query; first or second pass, keyword or q = *) when a group's documents live on
 * several shards. Round 1: every shard's `capacity` best groups with their greatest KVs -> per distinct key the greatest head, the collection's `capacity` best groups,
 * best first (a group the collection selects is among the best of the shard that holds its head); round 2: every shard answers for exactly those groups (slot r = group r:
 * its member count and its group_limit greatest KVs on that shard) -> group_found adds up, a group's KVs merge to its group_limit greatest, groups_count = cardinality of the
 * shards' LogLogBeta registers merged by their maxima (first pass), num_matched adds up. Results are those of the one-GPU call on the whole collection
 * (Topster(capacity, distinct, first_pass), include/topster.h:266-466; populate_result_kvs, src/index.cpp:8962-9011). gout->groups_total must be NULL (the exact distinct-key
 * count is not computed across shards: 501); matched-id lists are not offered. Every member needs its doc range for q = * queries. out / gout: HOST arrays. */
int tsgpu_group_keyword_search_grouped_batch(tsgpu_group* g, const tsgpu_kw_query* queries, const tsgpu_group_by* groups, uint32_t n_queries, tsgpu_hits* out, tsgpu_grouped_hits* gout);
/* ... and over candidate-token combinations (tsgpu_keyword_search_grouped_candidates_batch over the shards: combos / group_begin / groups / query_index as there, 1..16
 * combinations per user query): both rounds run the shards' own folds (a document lives in one shard); KV::query_index — the earlier passes that matched ANYTHING — comes from
 * the OR of the shards' pass masks; num_matched = the last pass' counts added up. Same restrictions as the call above. */
int tsgpu_group_keyword_search_grouped_candidates_batch(tsgpu_group* g, const tsgpu_kw_query* combos, const uint32_t* group_begin, const tsgpu_group_by* groups, uint32_t n_user,
                                                        tsgpu_hits* out, tsgpu_grouped_hits* gout, uint32_t* query_index);
/* "kw_exchange_slices" = 1 (default): the keyword exchange is an ncclAllToAll of query slices (member j receives only the records of
 * the 1/G of the batch it merges), a slice merge per member, and in-place ncclAllGathers of the merged lists (rank form / device outputs;
 * the local form with host outputs delivers every slice over its own GPU's PCIe link); 0: ONE ncclAllGather of the per-GPU top-k
 * blocks and a full merge. Identical results; at 8 GPUs and 10 000 queries 56 MB instead of 224 MB arrive per GPU. */
/* "kw_own_slice_only" = 1 (rank form, slice exchange): a rank delivers only the query slice it merged — queries [rank * per, (rank + 1) * per),
 * per = ceil(n_queries / n_ranks) — into its output arrays at those queries' slots; the all-gather of the merged lists is skipped (at 8
 * GPUs and 10 000 queries 35 MB instead of 70 MB arrive per GPU). For deployments where every rank serves its own callers, and what the local
 * form does with host outputs anyway. Every rank must set it (it is part of the call's agreed signature). */
/* "replicas" = 1: every member mirrors the WHOLE collection (small corpora: 10M documents = 6 GB of postings + 46 GB of vectors fit a
 * 288 GB GPU several times); a batch is cut into G query slices, member i answers slice i, the slices are delivered / replicated as
 * above — no merge, and the per-batch fixed costs (planning, launches) shrink with the slice. Default 0 = doc-range shards. */
int tsgpu_group_set_option(tsgpu_group* g, const char* name, int64_t value);
/* exact k-NN over all shards (tsgpu_vec_knn_batch per member + the exchange); labels must be seq_ids (< 2^32); members * k <= 8192 */
int tsgpu_group_vec_knn_batch(tsgpu_group* g, uint32_t vec_field_id, const float* Q, int mem_q, uint32_t n_queries, uint32_t k,
                              const uint32_t* allow_ids, uint32_t n_allow, const uint32_t* excluded_ids, uint32_t n_excluded,
                              float* dist_out, uint64_t* label_out, uint32_t* n_out, int mem_out);
/* hybrid: merged keyword Topsters (capacity out->k_stride) + merged k nearest, then tsgpu_hybrid_fuse_batch. Host outputs; metric / dim
 * = the vector field's (TSGPU_METRIC_*, num_dim); rerank_hybrid_matches (compute_aux_scores): every shard scores the one-sided hits it owns, the answers are gathered,
 * every rank re-fuses */
int tsgpu_group_hybrid_search_batch(tsgpu_group* g, const tsgpu_kw_query* queries, uint32_t vec_field_id, int metric, const tsgpu_hybrid_params* p,
                                    const float* Q, int mem_q, uint32_t dim, uint32_t n_queries, tsgpu_hits* out);
/* facet counts (tsgpu_facet_count_batch's arguments and output; the hash-index branch of Index::do_facets, src/index.cpp:1659-1771) over doc-range shards: every
 * member holds the facet mirror of ITS documents (tsgpu_facet_set with global seq_ids: the documents of other shards have no hashes), walks the matched ids of its
 * doc range (the whole lists with sample_mod > 1, or when a member has no doc_range_lo / _hi), and the per-shard lists are gathered and merged: counts add up,
 * doc_id / array_pos = the greatest document's. Exact for the first `cap` values; n_values is exact while no shard had more than cap, else a lower bound > cap. */
int tsgpu_group_facet_count_batch(tsgpu_group* g, uint32_t facet_field_id, const uint32_t* const* result_ids, const uint64_t* n_result_ids, uint32_t n_queries,
                                  uint32_t sample_mod, const uint32_t* allowed_hashes, uint32_t n_allowed, tsgpu_facet_counts* out);
/* ... the range facets (tsgpu_facet_range_count_batch without a group column: the counts add up; the grouped form is not sharded) and the facet stats
 * (tsgpu_facet_stats_batch: min / max / count / sum merged, sum_exact recomputed from the merged values) over the shards */
int tsgpu_group_facet_range_count_batch(tsgpu_group* g, uint32_t facet_field_id, uint32_t value_column, const int64_t* range_upper, const int64_t* range_lower, uint32_t n_ranges,
                                        const uint32_t* const* result_ids, const uint64_t* n_result_ids, uint32_t n_queries, uint32_t sample_mod, uint32_t* counts);
int tsgpu_group_facet_stats_batch(tsgpu_group* g, uint32_t facet_field_id, int value_type, const uint32_t* const* result_ids, const uint64_t* n_result_ids, uint32_t n_queries,
                                  uint32_t sample_mod, const uint32_t* int64_map_hashes, const int64_t* int64_map_values, uint32_t n_map, tsgpu_facet_stats* out);
typedef struct tsgpu_group_timings {
    float local_ms;                      /* host wall: every member's own batch + pack (members run concurrently) */
    float exchange_merge_ms;             /* host wall: the exchange, the merge and the delivery of the merged result */
    uint64_t exchange_bytes_per_member;  /* bytes one member RECEIVES from the others per call (all collectives of the call) */
    uint64_t hit_exchange_bytes_per_member;  /* (ABI 5) the part of it that carries the shards' hits to their mergers: the bounds + the (pruned) slices /
                                              * blocks — without the replication of the merged lists, which does not depend on how the hits travelled */
    float exchange_kernels_ms;               /* (ABI 5) device time (HIP events, member 0's stream) of the exchange's own kernels: pack or bounds, count + pruned pack, slice merge */
} tsgpu_group_timings;
int tsgpu_group_last_timings(tsgpu_group* g, tsgpu_group_timings* out);

/* ------------------------------------------------------------------ measurement hooks */
/* device time (ms, HIP events on the launch stream) of the dominant kernel(s) of the last batch call */
typedef struct tsgpu_timings {
    float kw_search_ms;    /* keyword intersect+score+select kernel */
    float kw_merge_ms;     /* per-query partial merge kernel */
    float vec_knn_ms;      /* MFMA distance + running top-k kernel */
    float vec_merge_ms;
    float total_ms;        /* first launch -> last kernel done */
    float vec_scan_ms;     /* the full-index scan kernel alone (vec_hscan_kernel / vec_scan_kernel), 0 for small indexes */
    uint64_t kw_algorithmic_bytes;   /* SURVEY §8(d) bytes of the last keyword batch */
    uint64_t vec_flops;              /* 2*N*D*B of the last knn batch */
    uint64_t vec_scan_bytes;         /* bytes the scan kernel must stream per launch: the row matrix once + the queries */
    float kw_find_ms;                /* two-kernel form, one group: the find kernel alone (kw_search_ms spans find + score); else 0 */
} tsgpu_timings;
int tsgpu_last_timings(tsgpu_ctx* ctx, tsgpu_timings* out);

/* Kernel time and algorithmic bytes of the last group_by batch (tsgpu_keyword_search_grouped[_candidates]_batch) and of the last facet-count batch
 * (tsgpu_facet_count[_grouped]_batch) of this context (measurement; bench.py `general_kernels.{group_by, facets}.roofline`). HIP events on the
 * library's own stream around the launches named below; the id pass of a grouped batch (the keyword kernels that produce the matched ids) is timed on
 * the host clock. Bytes: per matched id 4 (id) + 32 (its record: three sort keys + distinct key) and per table slot 20 (key, best record, rank, count:
 * what gb_select_kernel walks); facets: per id 4 + 16 (its doc_ptr pair) + 4 per value it holds, per table slot 20 (cleared + compacted). */
typedef struct tsgpu_aux_timings {
    float gb_id_pass_ms;             /* host clock: tsgpu_keyword_search_batch_ids of the combinations (0 for q = * over the whole collection) */
    float gb_kernels_ms;             /* gb_iota .. gb_chunk: everything between the id pass and the delivery */
    float gb_fold_ms;                /* ... of which gb_score + gb_dedupe + gb_insert (one thread per matched id) */
    float gb_select_ms;              /* ... gb_select_kernel (one workgroup per query) */
    uint64_t gb_matched_ids, gb_table_slots, gb_algorithmic_bytes;
    float facet_kernels_ms;          /* facet_count + facet_compact + facet_sort */
    float facet_count_ms;            /* ... facet_count_kernel alone */
    uint64_t facet_ids, facet_table_slots, facet_algorithmic_bytes;
} tsgpu_aux_timings;
int tsgpu_last_aux_timings(tsgpu_ctx* ctx, tsgpu_aux_timings* out);

/* Bytes the keyword kernels REQUEST, counted by the find kernel itself (measurement; bench.py `roofline.touched_bytes_per_launch`).
 * tsgpu_set_option("kw_count_touched", 1) makes keyword batches launch a second instantiation of the pair-find kernel
 * (kw_find2_kernel<TMAX, COUNT = true>) in which every lane adds the width of each of its own loads / LDS-DMA words / stores to
 * per-thread counters (one reduction + atomics per wavefront at the end) — the default instantiation carries none of it, so the
 * timed kernel is unchanged; results are identical. Host-planned batches only (set "kw_device_plan_min_queries" above the batch).
 * score_requested_bytes = hit records x the score kernel's fixed request sizes (host outputs; 0 otherwise). */
typedef struct tsgpu_kw_touched {
    uint64_t find_requested_bytes;   /* sum of the five classes below */
    uint64_t find_driver_ids;        /* the shortest list's ids, one aligned 2/4-byte load per slot */
    uint64_t find_metadata;          /* BlockIds windows of both lists, descriptors, query records, block-search loads */
    uint64_t find_tile_dma;          /* second-list runs copied into the LDS tile (2 or 7 slabs of 1 KB per pair of driver blocks) */
    uint64_t find_probes;            /* third.. lists (and wide / broken runs of the second): directory entries, guided searches */
    uint64_t find_records;           /* hit records written for the score kernel */
    uint64_t find_work_items, find_hit_records;
    uint64_t score_requested_bytes;
} tsgpu_kw_touched;
int tsgpu_kw_last_touched(tsgpu_ctx* ctx, tsgpu_kw_touched* out);
/* What the DISTINCT posting lists of n (field, term) pairs occupy in the mirror (terms not in the index are skipped): the working set a
 * batch's requested / fetched bytes are compared with (`roofline.l2_refetch`). Exact — block records are read back from the device. */
typedef struct tsgpu_kw_footprint {
    uint64_t n_lists, n_ids;
    uint64_t ids_bytes;              /* ids arena words of the lists' blocks (fixed-width deltas + guard words) */
    uint64_t block_metadata_bytes;   /* BlockIds (16) + blk_last (4) + BlockMeta (32) per block */
    uint64_t directory_bytes;        /* id directories of the long lists */
    uint64_t payload_bytes;          /* offset_index + offsets (the score kernel's side) */
} tsgpu_kw_footprint;
int tsgpu_kw_lists_footprint(tsgpu_ctx* ctx, const uint32_t* field_ids, const uint32_t* term_ids, uint32_t n, tsgpu_kw_footprint* out);

#ifdef __cplusplus
}
#endif
#endif /* TSGPU_H */
