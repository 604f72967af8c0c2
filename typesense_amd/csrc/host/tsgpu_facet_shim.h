// tsgpu_facet_shim.h — the patch body for the hash-index branch of Index::do_facets (src/index.cpp:1659-1771, "Using hashing to find facets") for ONE
// facet of a search whose facet hash index (tsgpu_facet_set) — and, for a range facet, sort index (a column) — is mirrored in a tsgpu context.
//
// Written against the reference's own types through a template parameter, like the other shims:
//     Facet : result_map            operator[](uint64_t) -> { count, doc_id, array_pos }            (spp::sparse_hash_map<uint64_t, facet_count_t>, include/field.h:783)
//             facet_range_map       ordered: begin() / end(), ->first = upper bound (the range id), ->second.lower_range   (std::map<int64_t, range_specs_t>, :772-779, :816)
//             is_range_query        bool
// One call walks ALL result ids of the search (the reference cuts them into per-thread windows and merges the windows' facets in aggregate_facet,
// src/index.cpp:4628-4660; for counts that merge is a sum, for a grouped search a union of group sets — the device does the whole walk at once, so there
// is nothing to merge). What the shim leaves in a_facet.result_map is what the reference holds AFTER the merge and the group step (:4452-4463):
//   plain           : count / doc_id / array_pos per value hash (:1751-1760),
//   group_limit != 0: count = the number of groups the value was seen in = hash_groups[hash].size() (:1756-1758, :4455-4458). hash_groups itself is NOT
//                     filled: the caller skips its `count = hash_groups[...].size()` overwrite for a facet counted here,
//   range facet     : result_map[range_id].count for the ranges that met a document (:1738-1750); grouped likewise.
// estimate_facets' final scaling (count * 100 / facet_sample_percent, :4460-4462) stays with the caller, as do hash_tokens (= fquery_hashes.at(hash), :1761-1764)
// and sort_field_val (= the sort-index value of the entry's doc_id, :1765-1767). Stats: tsgpu_facet_stats_batch. On any error NOTHING is touched.
#pragma once
#include <cstdint>
#include <vector>
#include "../../../include/tsgpu.h"

namespace tsgpu {

struct FacetShimArgs {
    tsgpu_ctx* ctx = nullptr;
    uint32_t facet_field_id = 0;                 // tsgpu_facet_set
    const uint32_t* result_ids = nullptr;        // ascending
    uint64_t results_size = 0;
    bool estimate_facets = false;
    uint32_t facet_sample_mod_value = 1;         // (:1683-1687)
    bool use_facet_query = false;
    const uint32_t* fquery_hashes = nullptr;     // the KEYS of fquery_hashes, ascending (:1742)
    uint32_t n_fquery_hashes = 0;
    uint64_t group_limit = 0;                    // != 0: the facets of a grouped search
    uint32_t group_column = 0;                   // the distinct-id column (build_distinct_column, tsgpu_groupby_shim.h)
    bool group_missing_values = true;
    uint32_t value_column = 0;                   // range facets: the field's sort index as a column (INT64_MAX where it has no entry)
    uint32_t values_hint = 4096;                 // distinct values expected (the call is repeated once with the exact number when there are more)
};

template <class Facet>
int do_facets_hash_gpu(const FacetShimArgs& a, Facet& a_facet) {
    const uint32_t* ids = a.result_ids;
    const uint64_t n = a.results_size;
    const uint32_t mod = a.estimate_facets ? a.facet_sample_mod_value : 1;
    if (a_facet.is_range_query) {
        std::vector<int64_t> upper, lower;
        for (auto it = a_facet.facet_range_map.begin(); it != a_facet.facet_range_map.end(); ++it) { upper.push_back((int64_t)it->first); lower.push_back((int64_t)it->second.lower_range); }
        if (upper.empty()) return TSGPU_OK;                                       // "Facet range is not defined": get_range finds nothing
        std::vector<uint32_t> counts(upper.size());
        const int rc = tsgpu_facet_range_count_batch(a.ctx, a.facet_field_id, a.value_column, upper.data(), lower.data(), (uint32_t)upper.size(), &ids, &n, 1, mod,
                                                     a.group_limit ? a.group_column : TSGPU_NO_COLUMN, a.group_missing_values ? 1 : 0, counts.data());
        if (rc != TSGPU_OK) return rc;
        for (size_t r = 0; r < upper.size(); r++)
            if (counts[r]) a_facet.result_map[(uint64_t)upper[r]].count = counts[r];
        return TSGPU_OK;
    }
    std::vector<uint32_t> h, c, d, p;
    uint32_t n_values = 0;
    for (uint32_t cap = a.values_hint ? a.values_hint : 1;;) {
        h.resize(cap); c.resize(cap); d.resize(cap); p.resize(cap);
        tsgpu_facet_counts out;
        out.cap = cap; out.hash = h.data(); out.count = c.data(); out.doc_id = d.data(); out.array_pos = p.data(); out.n_values = &n_values;
        const uint32_t* fq = a.use_facet_query ? a.fquery_hashes : nullptr;
        const uint32_t nfq = a.use_facet_query ? a.n_fquery_hashes : 0;
        if (a.use_facet_query && nfq == 0) return TSGPU_OK;                       // nothing can pass the facet query
        const int rc = a.group_limit ? tsgpu_facet_count_grouped_batch(a.ctx, a.facet_field_id, &ids, &n, 1, mod, fq, nfq, a.group_column, a.group_missing_values ? 1 : 0, &out)
                                     : tsgpu_facet_count_batch(a.ctx, a.facet_field_id, &ids, &n, 1, mod, fq, nfq, &out);
        if (rc != TSGPU_OK) return rc;
        if (n_values <= cap) break;
        cap = n_values;                                                           // more values than expected: once more with room for all of them
    }
    for (uint32_t i = 0; i < n_values; i++) {
        auto& fc = a_facet.result_map[(uint64_t)h[i]];
        fc.count = c[i]; fc.doc_id = d[i]; fc.array_pos = p[i];
    }
    return TSGPU_OK;
}

}  // namespace tsgpu
