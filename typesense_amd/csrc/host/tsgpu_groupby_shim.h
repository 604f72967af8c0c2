// tsgpu_groupby_shim.h — the patch body for the GROUPED branch of seam B1: what replaces the scoring loop of the reference's
// Index::search_across_fields (src/index.cpp:5468-5551) — or Index::search_wildcard's per-thread loop (:6700-6760) — when group_limit != 0 and the
// index is mirrored in a tsgpu context. One call = one pass of Index::run_search's two passes (src/index.cpp:2488-2760).
//
// Written against the reference's own types through template parameters, like tsgpu_keyword_shim.h:
//     KV       : KV(uint16_t query_index, uint64_t key, uint64_t distinct_key, int8_t match_score_index, const int64_t* scores) + text_match_score
//     TopsterT : int add(KV*), bool is_group_by_first_pass, std::unique_ptr<LogLogBeta> loglog_counter with void addHash(uint64_t)   (include/topster.h:238-296)
//     GroupsProcessed : operator[](uint64_t) -> uint32_t&                                                                             (spp::sparse_hash_map<uint64_t, uint32_t>)
// What the library returns is ALREADY the pass' outcome (first pass: the greatest KV of each of the best groups; second pass: populate_result_kvs'
// groups with their KVs). The shim re-adds exactly those KVs to the caller's Topster, which thereby holds the state the reference's consumers read:
//   first pass : the same set of (distinct key -> greatest KV) — consumers take it as a set (Index::get_group_by_values, src/index.cpp:7144-7170);
//                loglog_counter receives EVERY distinct key of the pass through the sketch registers the device built (getGroupsCount() as Index::run_search reads it, :2691);
//   second pass: group_kv_map holds the returned groups with their group_limit greatest KVs each, so populate_result_kvs (src/index.cpp:8962-9011)
//                yields what it would have yielded over all matched documents;
//   groups_processed[distinct_key] = the group's matched documents (:5546-5549), id_buff / num_keyword_matches / search_cutoff as in the plain shim,
//   group_by_missing_value_ids (first pass, :7107-7109 / :7132-7134) = the matched ids without a value in some group_by field.
// On 501 (sort_by _group_found, > 4 query_by fields, ...) NOTHING is touched and the caller runs its unchanged CPU body.
#pragma once
#include <cstdint>
#include <set>
#include <vector>
#include "../../../include/tsgpu.h"

namespace tsgpu {

// StringUtils::hash_combine (include/string_utils.h:323-326)
inline uint64_t groupby_hash_combine(uint64_t combined, uint64_t hash) {
    combined ^= hash + 0x517cc1b727220a95ull + (combined << 6) + (combined >> 2);
    return combined;
}

// The distinct-key column of one (group_by fields, group_missing_values) combination: what Index::get_distinct_id (src/index.cpp:7100-7142) yields per
// seq_id from the fields' facet hash indexes, given as CSR (doc_ptr[f][d] .. doc_ptr[f][d + 1] = the hashes of document d in field f: one for a plain
// field, one per element for an array — the layout tsgpu_facet_set takes). has_value[d] = every field held a value. Upload with
// tsgpu_column_set(ctx, column, values.data(), NULL, n_docs, TSGPU_MEM_HOST); rebuild when the collection's documents change.
inline void build_distinct_column(uint32_t n_docs, const std::vector<const uint64_t*>& doc_ptr, const std::vector<const uint32_t*>& hashes,
                                  bool group_missing_values, std::vector<int64_t>& values, std::vector<uint8_t>& has_value) {
    values.resize(n_docs);
    has_value.assign(n_docs, 1);
    for (uint32_t d = 0; d < n_docs; d++) {
        uint64_t distinct_id = 1;
        for (size_t f = 0; f < doc_ptr.size(); f++) {
            const uint64_t b = doc_ptr[f][d], e = doc_ptr[f][d + 1];
            if (b == e) has_value[d] = 0;
            for (uint64_t j = b; j < e; j++) distinct_id = groupby_hash_combine(distinct_id, hashes[f][j]);
            if (distinct_id == 1 && !group_missing_values) distinct_id = d;            // (:7137-7139, evaluated per field like the reference's call per field)
        }
        values[d] = (int64_t)distinct_id;
    }
}

struct GroupByShimArgs {
    tsgpu_ctx* ctx = nullptr;
    tsgpu_kw_query query{};              // as in KeywordShimArgs
    tsgpu_group_by group{};              // group_limit, column, first_pass, group_missing_values, wildcard
    uint16_t query_index = 0;
    const uint8_t* has_value = nullptr;  // build_distinct_column's has_value (nullable: no missing ids are reported)
    uint32_t n_has_value = 0;
};

// groups_processed receives the groups the call RETURNS (at most topster_size of them), the reference fills it for every distinct key of the pass. Index::search
// reads groups_processed.size() as results_count for the typo_tokens_threshold / drop_tokens_threshold decisions (src/index.cpp:5095, 3617): with a threshold
// above topster_size use *groups_total — the exact number of distinct keys of the pass — in its place.
template <class KV, class TopsterT, class GroupsProcessed>
int search_across_fields_grouped_gpu(const GroupByShimArgs& a, TopsterT* topster, GroupsProcessed& groups_processed, std::vector<uint32_t>& id_buff,
                                     size_t& num_keyword_matches, bool& search_cutoff, std::set<uint32_t>* group_by_missing_value_ids, uint64_t* groups_total = nullptr) {
    const uint32_t K = a.query.topster_size ? a.query.topster_size : TSGPU_DEFAULT_TOPSTER_SIZE;
    const uint32_t L = a.group.first_pass ? 1u : a.group.group_limit;
    const size_t slots = (size_t)K * (L ? L : 1);
    std::vector<uint64_t> keys(slots), dkeys(K);
    std::vector<int64_t> scores(slots * 3), text_match(slots);
    std::vector<int8_t> msi(slots);
    std::vector<uint32_t> gsize(K), gfound(K);
    std::vector<uint8_t> regs(a.group.first_pass ? 16384 : 0);
    uint32_t n_hits = 0, n_groups = 0;
    uint64_t num_matched = 0;
    int32_t status = 0, cutoff = 0;
    tsgpu_hits h{};
    h.mem = TSGPU_MEM_HOST; h.k_stride = (uint32_t)slots;
    h.keys = keys.data(); h.scores = scores.data(); h.text_match = text_match.data(); h.match_score_index = msi.data();
    h.n_hits = &n_hits; h.num_matched = &num_matched; h.status = &status; h.search_cutoff = &cutoff;
    tsgpu_grouped_hits g{};
    g.g_stride = K; g.n_groups = &n_groups; g.distinct_key = dkeys.data(); g.group_size = gsize.data(); g.group_found = gfound.data();
    g.loglog_registers = a.group.first_pass ? regs.data() : nullptr;
    uint64_t gtotal = 0;
    g.groups_total = &gtotal;
    tsgpu_id_lists* ids = nullptr;
    const int rc = tsgpu_keyword_search_grouped_batch(a.ctx, &a.query, &a.group, 1, &h, &g, &ids);
    if (rc != TSGPU_OK) return rc;
    if (status != TSGPU_OK) { tsgpu_id_lists_free(ids); search_cutoff = search_cutoff || cutoff != 0; return status; }
    for (uint32_t r = 0; r < n_groups; r++) {
        for (uint32_t j = 0; j < gsize[r]; j++) {
            const size_t o = (size_t)r * L + j;
            KV kv(a.query_index, keys[o], dkeys[r], msi[o], &scores[o * 3]);
            kv.text_match_score = text_match[o];
            topster->add(&kv);
        }
        groups_processed[dkeys[r]] += gfound[r];
    }
    if (a.group.first_pass && topster->loglog_counter) {
        // every distinct key of the pass, through the registers: x is any hash that lands in register k with rho = v (include/loglogbeta.h:86-99)
        for (uint32_t k = 0; k < 16384; k++) {
            const uint32_t v = regs[k];
            if (v == 0) continue;
            topster->loglog_counter->addHash(((uint64_t)k << 50) | (v <= 50 ? 1ull << (50 - v) : 0ull));
        }
    }
    if (ids) {
        const uint64_t n = tsgpu_id_lists_count(ids, 0);
        const uint32_t* p = tsgpu_id_lists_ids(ids, 0);
        id_buff.insert(id_buff.end(), p, p + n);
        if (a.group.first_pass && group_by_missing_value_ids)
            for (uint64_t i = 0; i < n; i++) if (p[i] >= a.n_has_value || (a.has_value && !a.has_value[p[i]])) group_by_missing_value_ids->insert(p[i]);
        tsgpu_id_lists_free(ids);
    }
    num_keyword_matches = (size_t)num_matched;
    search_cutoff = search_cutoff || cutoff != 0;
    if (groups_total) *groups_total = gtotal;
    return TSGPU_OK;
}

// The candidate loop of Index::search_all_candidates with group_limit != 0 (src/index.cpp:1794-1894): `combos` = the candidate-token combinations in the reference's
// pass order (each filled like KeywordShimArgs::query, its own term_ids / total_cost); ONE call folds them like the one Topster and the one groups_processed the reference
// passes through the loop. base_query_index = searched_queries.size() before the loop; matched_combos (out) = which combinations matched anything (the reference appends
// exactly those to searched_queries, :5580-5585). Everything else as in search_across_fields_grouped_gpu; id_buff receives the ascending UNION of the passes' ids.
struct GroupByCandidatesShimArgs {
    tsgpu_ctx* ctx = nullptr;
    std::vector<tsgpu_kw_query> combos;
    tsgpu_group_by group{};
    uint16_t base_query_index = 0;
    const uint8_t* has_value = nullptr;
    uint32_t n_has_value = 0;
};

template <class KV, class TopsterT, class GroupsProcessed>
int search_all_candidates_grouped_gpu(const GroupByCandidatesShimArgs& a, TopsterT* topster, GroupsProcessed& groups_processed, std::vector<uint32_t>& id_buff,
                                      size_t& num_keyword_matches, bool& search_cutoff, std::set<uint32_t>* group_by_missing_value_ids, uint64_t* groups_total = nullptr) {
    if (a.combos.empty()) return TSGPU_OK;
    const uint32_t K = a.combos[0].topster_size ? a.combos[0].topster_size : TSGPU_DEFAULT_TOPSTER_SIZE;
    const uint32_t L = a.group.first_pass ? 1u : a.group.group_limit;
    const size_t slots = (size_t)K * (L ? L : 1);
    std::vector<uint64_t> keys(slots), dkeys(K);
    std::vector<int64_t> scores(slots * 3), text_match(slots);
    std::vector<int8_t> msi(slots);
    std::vector<uint32_t> gsize(K), gfound(K), qidx(slots);
    std::vector<uint8_t> regs(a.group.first_pass ? 16384 : 0);
    uint32_t n_hits = 0, n_groups = 0;
    uint64_t num_matched = 0;
    int32_t status = 0, cutoff = 0;
    tsgpu_hits h{};
    h.mem = TSGPU_MEM_HOST; h.k_stride = (uint32_t)slots;
    h.keys = keys.data(); h.scores = scores.data(); h.text_match = text_match.data(); h.match_score_index = msi.data();
    h.n_hits = &n_hits; h.num_matched = &num_matched; h.status = &status; h.search_cutoff = &cutoff;
    tsgpu_grouped_hits g{};
    g.g_stride = K; g.n_groups = &n_groups; g.distinct_key = dkeys.data(); g.group_size = gsize.data(); g.group_found = gfound.data();
    g.loglog_registers = a.group.first_pass ? regs.data() : nullptr;
    uint64_t gtotal = 0;
    g.groups_total = &gtotal;
    const uint32_t begin[2] = {0, (uint32_t)a.combos.size()};
    tsgpu_id_lists* ids = nullptr;
    const int rc = tsgpu_keyword_search_grouped_candidates_batch(a.ctx, a.combos.data(), begin, &a.group, 1, &h, &g, qidx.data(), &ids);
    if (rc != TSGPU_OK) return rc;
    if (status != TSGPU_OK) { tsgpu_id_lists_free(ids); search_cutoff = search_cutoff || cutoff != 0; return status; }
    for (uint32_t r = 0; r < n_groups; r++) {
        for (uint32_t j = 0; j < gsize[r]; j++) {
            const size_t o = (size_t)r * L + j;
            KV kv((uint16_t)(a.base_query_index + qidx[o]), keys[o], dkeys[r], msi[o], &scores[o * 3]);
            kv.text_match_score = text_match[o];
            topster->add(&kv);
        }
        groups_processed[dkeys[r]] += gfound[r];
    }
    if (a.group.first_pass && topster->loglog_counter) {
        for (uint32_t k = 0; k < 16384; k++) {
            const uint32_t v = regs[k];
            if (v) topster->loglog_counter->addHash(((uint64_t)k << 50) | (v <= 50 ? 1ull << (50 - v) : 0ull));
        }
    }
    if (ids) {
        const uint64_t n = tsgpu_id_lists_count(ids, 0);
        const uint32_t* p = tsgpu_id_lists_ids(ids, 0);
        id_buff.insert(id_buff.end(), p, p + n);
        if (a.group.first_pass && group_by_missing_value_ids)
            for (uint64_t i = 0; i < n; i++) if (p[i] >= a.n_has_value || (a.has_value && !a.has_value[p[i]])) group_by_missing_value_ids->insert(p[i]);
        tsgpu_id_lists_free(ids);
    }
    num_keyword_matches = (size_t)num_matched;
    search_cutoff = search_cutoff || cutoff != 0;
    if (groups_total) *groups_total = gtotal;
    return TSGPU_OK;
}

}  // namespace tsgpu
