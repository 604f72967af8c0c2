// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/for_codec.h header).
// Restates the hash-index branch of Index::do_facets, /root/reference/src/index.cpp:1659-1771 ("Using hashing to find facets"),
// (plain counts: count(); the group_by and range-facet forms of the same walk: count_ex()): result ids are walked in ascending order; a
// document absent from the facet hash index is skipped (:1696-1698) and the walk stops once the index is exhausted (:1692-1694);
// per document every DISTINCT hash (unique_facet_hashes, :1719-1728) bumps result_map[hash].count and records doc_id / array_pos
// (:1744-1752); estimate_facets skips ids whose position is not a multiple of facet_sample_mod_value (:1683-1687); with a facet
// query only hashes in fquery_hashes are counted (:1742). The facet hash index itself (facet_index_v4: a posting list
// seq_id -> value hashes, scalar fields one hash = offset(), array fields the offsets list) is held as a map here.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <set>
#include <vector>

namespace oracle {

struct facet_count_t { uint32_t count = 0; uint32_t doc_id = 0; uint32_t array_pos = 0; };

struct FacetHashIndex {
    std::map<uint32_t, std::vector<uint32_t>> docs;           // seq_id -> hashes in field order (documents without a value are absent)

    // should_compute_stats (src/index.cpp:1730-1741) + compute_facet_stats(a_facet, int64_t raw_value, type) (:1430-1460): the same walk,
    // every (document, distinct hash) BEFORE the facet-query filter; value_type 0 = int32 (val = (int32) hash), 1 = int64 (fhash_int64_map,
    // a missing hash -> INT64_MAX), 2 = float (the hash's bits). Sequential double accumulation in document order, as the reference.
    struct stats_t { double fvmin = std::numeric_limits<double>::max(), fvmax = -std::numeric_limits<double>::max(), fvcount = 0, fvsum = 0; };   // include/field.h:765-770
    stats_t stats(const uint32_t* result_ids, size_t results_size, size_t facet_sample_mod_value, int value_type,
                  const std::map<uint32_t, int64_t>* fhash_int64_map) const {
        stats_t st;
        for (size_t i = 0; i < results_size; i++) {
            if (facet_sample_mod_value > 1 && i % facet_sample_mod_value != 0) continue;
            auto it = docs.lower_bound(result_ids[i]);
            if (it == docs.end()) break;
            if (it->first != result_ids[i]) continue;
            const std::vector<uint32_t>& facet_hashes = it->second;
            std::set<uint32_t> unique_facet_hashes;
            for (size_t j = 0; j < facet_hashes.size(); j++) {
                const uint32_t fhash = facet_hashes[j];
                if (facet_hashes.size() > 1) {
                    if (unique_facet_hashes.count(fhash) != 0) continue;
                    unique_facet_hashes.insert(fhash);
                }
                int64_t val = fhash;                                   // (the reference widens the uint32 hash; the int32 branch truncates it again)
                if (value_type == 1) { auto m = fhash_int64_map ? fhash_int64_map->find(fhash) : decltype(fhash_int64_map->find(fhash))(); val = (fhash_int64_map && m != fhash_int64_map->end()) ? m->second : INT64_MAX; }
                if (value_type == 0) { int32_t v = (int32_t)val; if (v < st.fvmin) st.fvmin = v; if (v > st.fvmax) st.fvmax = v; st.fvsum += v; st.fvcount++; }
                else if (value_type == 1) { int64_t v = val; if (v < st.fvmin) st.fvmin = v; if (v > st.fvmax) st.fvmax = v; st.fvsum += v; st.fvcount++; }
                else { float v; uint32_t b = fhash; memcpy(&v, &b, 4); if (v < st.fvmin) st.fvmin = v; if (v > st.fvmax) st.fvmax = v; st.fvsum += v; st.fvcount++; }
            }
        }
        return st;
    }

    std::map<uint32_t, facet_count_t> count(const uint32_t* result_ids, size_t results_size, size_t facet_sample_mod_value,
                                             const std::set<uint32_t>* fquery_hashes) const {
        std::map<uint32_t, facet_count_t> result_map;
        auto it = docs.begin();                                 // facet_index_it: only moves forward (skip_to)
        for (size_t i = 0; i < results_size; i++) {
            if (facet_sample_mod_value > 1 && i % facet_sample_mod_value != 0) continue;
            const uint32_t doc_seq_id = result_ids[i];
            it = docs.lower_bound(doc_seq_id);                  // skip_to(doc_seq_id)
            if (it == docs.end()) break;
            if (it->first != doc_seq_id) continue;
            const std::vector<uint32_t>& facet_hashes = it->second;
            std::set<uint32_t> unique_facet_hashes;
            for (size_t j = 0; j < facet_hashes.size(); j++) {
                const uint32_t fhash = facet_hashes[j];
                if (facet_hashes.size() > 1) {
                    if (unique_facet_hashes.count(fhash) != 0) continue;
                    unique_facet_hashes.insert(fhash);
                }
                if (fquery_hashes && fquery_hashes->find(fhash) == fquery_hashes->end()) continue;
                facet_count_t& fc = result_map[fhash];
                fc.doc_id = doc_seq_id;
                fc.array_pos = (uint32_t)j;
                fc.count += 1;
            }
        }
        return result_map;
    }

    // The same walk with the two forms count() leaves out (src/index.cpp:1738-1760):
    //  * range facets (a_facet.is_range_query): per DISTINCT hash of the document (the branch sits inside the hash loop) doc_val =
    //    get_doc_val_from_sort_index (:1470-1482, INT64_MAX when the sort index has no entry) goes through facet::get_range
    //    (include/field.h:820-838 over facet_range_map: upper bound -> {label, lower_range}; range_specs_t::is_in_range :776-778) and
    //    result_map[range_id].count += 1 (doc_id / array_pos untouched);
    //  * group_limit != 0: hash_groups[key].emplace(distinct_id) — spp::sparse_hash_map<uint32_t, sparse_hash_set<uint32_t>> (:791), both truncated —
    //    plain values do NOT count documents then (:1756-1760), ranges do; afterwards every result_map entry's count = hash_groups[key].size()
    //    (:4455-4458).
    // doc_val / distinct_id: the per-document values the caller read from the sort index / get_distinct_id.
    template <class DocVal, class DistinctId>
    std::map<uint64_t, facet_count_t> count_ex(const uint32_t* result_ids, size_t results_size, size_t facet_sample_mod_value, const std::set<uint32_t>* fquery_hashes,
                                                const std::map<int64_t, int64_t>* facet_range_map, DocVal doc_val, bool group_limit, DistinctId distinct_id_of) const {
        std::map<uint64_t, facet_count_t> result_map;
        std::map<uint32_t, std::set<uint32_t>> hash_groups;
        for (size_t i = 0; i < results_size; i++) {
            if (facet_sample_mod_value > 1 && i % facet_sample_mod_value != 0) continue;
            const uint32_t doc_seq_id = result_ids[i];
            auto it = docs.lower_bound(doc_seq_id);
            if (it == docs.end()) break;
            if (it->first != doc_seq_id) continue;
            const std::vector<uint32_t>& facet_hashes = it->second;
            const uint64_t distinct_id = group_limit ? distinct_id_of(doc_seq_id) : 0;
            std::set<uint32_t> unique_facet_hashes;
            for (size_t j = 0; j < facet_hashes.size(); j++) {
                const uint32_t fhash = facet_hashes[j];
                if (facet_hashes.size() > 1) {
                    if (unique_facet_hashes.count(fhash) != 0) continue;
                    unique_facet_hashes.insert(fhash);
                }
                if (facet_range_map) {
                    const int64_t key = doc_val(doc_seq_id);
                    auto rit = facet_range_map->lower_bound(key);                              // facet::get_range
                    if (rit != facet_range_map->end() && rit->first == key) rit++;
                    if (rit != facet_range_map->end() && key >= rit->second) {
                        const int64_t range_id = rit->first;
                        result_map[(uint64_t)range_id].count += 1;
                        if (group_limit) hash_groups[(uint32_t)range_id].emplace((uint32_t)distinct_id);
                    }
                } else if (!fquery_hashes || fquery_hashes->find(fhash) != fquery_hashes->end()) {
                    facet_count_t& fc = result_map[fhash];
                    fc.doc_id = doc_seq_id;
                    fc.array_pos = (uint32_t)j;
                    if (group_limit) hash_groups[fhash].emplace((uint32_t)distinct_id);
                    else fc.count += 1;
                }
            }
        }
        if (group_limit) for (auto& kv : result_map) kv.second.count = (uint32_t)hash_groups[(uint32_t)kv.first].size();
        return result_map;
    }
};

// Value-index branch ("Using intersection to find facets", src/index.cpp:1596-1657): facet_index_t::intersect (src/facet_index.cpp:230-353)
// with ids_t::intersect_count (src/ids_t.cpp:148-169, 368-377; src/id_list.cpp:725-766). Values are held in the reference's visiting order
// (counter_list); `order` = the alphabetical walks. docid_count_t = {first_id(ids), count}.
struct FacetValueIndex {
    std::vector<std::vector<uint32_t>> ids;                     // value -> ascending seq_ids
    std::vector<uint32_t> total;                                // facet_count_it->count

    static size_t intersect_count(const std::vector<uint32_t>& list, const uint32_t* res_ids, size_t res_ids_len, bool estimate_facets, size_t interval) {
        size_t count = 0, res_index = 0, i = 0;
        const bool compact = list.size() < 64;                  // ids_t::COMPACT_LIST_THRESHOLD_LENGTH: compact lists are never estimated
        if (estimate_facets && !compact) {
            while (i < list.size() && res_index < res_ids_len) {
                if (list[i] == res_ids[res_index]) { count++; i += interval; res_index += interval; }
                else if (list[i] < res_ids[res_index]) i += interval;
                else res_index += interval;
            }
            count = count * interval * interval;
        } else {
            while (i < list.size() && res_index < res_ids_len) {
                if (list[i] == res_ids[res_index]) { count++; i++; res_index++; }
                else if (list[i] < res_ids[res_index]) i = std::lower_bound(list.begin() + i, list.end(), res_ids[res_index]) - list.begin();    // skip_to
                else res_index = std::lower_bound(res_ids + res_index, res_ids + res_ids_len, list[i]) - res_ids;
            }
        }
        return std::min<size_t>(list.size(), count);
    }

    struct found_t { uint32_t value, doc_id, count; };
    std::vector<found_t> intersect(const uint32_t* result_ids, size_t results_size, size_t max_facets, bool is_wildcard_no_filter_query, bool estimate_facets,
                                   size_t facet_sample_interval, const std::vector<uint32_t>* order) const {
        std::vector<found_t> found;
        if (results_size == 0) return found;                    // do_facets returns before any facet is looked at (:1531-1533)
        for (size_t p = 0; p < ids.size(); p++) {
            const uint32_t v = order ? (*order)[p] : (uint32_t)p;
            uint32_t count;
            if (is_wildcard_no_filter_query) count = total[v];
            else {
                const bool estimate_facet_count = estimate_facets && ids[v].size() > 300;
                count = (uint32_t)intersect_count(ids[v], result_ids, results_size, estimate_facet_count, facet_sample_interval);
            }
            if (count) found.push_back({v, ids[v][0], count});
            if (found.size() == max_facets) break;
        }
        return found;
    }
};

}  // namespace oracle
