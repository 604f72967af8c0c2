// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/for_codec.h header).
// Restates the hash-index branch of Index::do_facets, /root/reference/src/index.cpp:1659-1771 ("Using hashing to find facets"),
// for the plain case the GPU path covers (no group_by, no range facet, no stats): result ids are walked in ascending order; a
// document absent from the facet hash index is skipped (:1696-1698) and the walk stops once the index is exhausted (:1692-1694);
// per document every DISTINCT hash (unique_facet_hashes, :1719-1728) bumps result_map[hash].count and records doc_id / array_pos
// (:1744-1752); estimate_facets skips ids whose position is not a multiple of facet_sample_mod_value (:1683-1687); with a facet
// query only hashes in fquery_hashes are counted (:1742). The facet hash index itself (facet_index_v4: a posting list
// seq_id -> value hashes, scalar fields one hash = offset(), array fields the offsets list) is held as a map here.
#pragma once
#include <cstdint>
#include <map>
#include <set>
#include <vector>

namespace oracle {

struct facet_count_t { uint32_t count = 0; uint32_t doc_id = 0; uint32_t array_pos = 0; };

struct FacetHashIndex {
    std::map<uint32_t, std::vector<uint32_t>> docs;           // seq_id -> hashes in field order (documents without a value are absent)

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
};

}  // namespace oracle
