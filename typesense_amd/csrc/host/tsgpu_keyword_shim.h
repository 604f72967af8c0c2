// tsgpu_keyword_shim.h — the patch body for seam B1: what replaces lines 5468-5551 of the reference's
// Index::search_across_fields (src/index.cpp) when the index is mirrored in a tsgpu context.
//
// It is written against the reference's own types through template parameters so that it compiles both inside
// the server (KV = ::KV, TopsterT = Topster<KV>) and in this repository's tests (any type with the same members):
//     KV      : KV(uint16_t query_index, uint64_t key, uint64_t distinct_key, int8_t match_score_index, const int64_t* scores)
//               + members scores[3], text_match_score, vector_distance           (include/topster.h:20-47)
//     TopsterT: int add(KV*)                                                      (include/topster.h:321)
// Contract kept from the reference: the caller owns `topster`; every emitted seq_id is appended to `id_buff`
// (all_result_ids feed `found` and facets, src/index.cpp:5565); `num_keyword_matches` is set; on an unsupported
// query (501) NOTHING is touched and the caller runs its unchanged CPU body for that call.
#pragma once
#include <cstdint>
#include <vector>
#include "../../../include/tsgpu.h"

namespace tsgpu {

struct KeywordShimArgs {
    tsgpu_ctx* ctx = nullptr;
    tsgpu_kw_query query{};          // filled from query_tokens / the_fields / sort_fields_std / flags by the call site
    uint16_t query_index = 0;        // searched_queries.size() at the time of the pass (src/index.cpp:5539)
    bool want_result_ids = true;     // false when neither facets nor group-by need all_result_ids (count is always exact)
};

// returns a tsgpu_status; TSGPU_ERR_UNSUPPORTED means "run the CPU body"
template <class KV, class TopsterT>
int search_across_fields_gpu(const KeywordShimArgs& a, TopsterT* topster, std::vector<uint32_t>& id_buff,
                             size_t& num_keyword_matches, bool& search_cutoff) {
    const uint32_t K = a.query.topster_size ? a.query.topster_size : TSGPU_DEFAULT_TOPSTER_SIZE;
    std::vector<uint64_t> keys(K);
    std::vector<int64_t> scores((size_t)K * 3), text_match(K);
    std::vector<float> vdist(K);
    std::vector<int8_t> msi(K);
    uint32_t n_hits = 0;
    uint64_t num_matched = 0;
    int32_t status = 0, cutoff = 0;
    tsgpu_hits h;
    h.mem = TSGPU_MEM_HOST;
    h.k_stride = K;
    // KV::text_match_score = scores[match_score_index] whenever _text_match is one of the sort keys (src/index.cpp:5541-5544), and a keyword
    // KV keeps its default vector_distance: neither array is requested then — a quarter of the bytes a large batch sends across PCIe
    bool sorts_on_text_match = false;
    for (uint32_t s = 0; s < a.query.n_sort; s++) sorts_on_text_match = sorts_on_text_match || a.query.sort[s].kind == TSGPU_SORT_TEXT_MATCH;
    h.keys = keys.data(); h.scores = scores.data(); h.text_match = sorts_on_text_match ? nullptr : text_match.data(); h.vector_distance = nullptr;
    h.match_score_index = msi.data(); h.n_hits = &n_hits; h.num_matched = &num_matched; h.status = &status; h.search_cutoff = &cutoff;
    // the matched ids come back as a list that belongs to THIS call (tsgpu_keyword_search_batch_ids): request threads share the
    // context, and the library may have coalesced this call with other threads' calls
    tsgpu_id_lists* ids = nullptr;
    int rc = a.want_result_ids ? tsgpu_keyword_search_batch_ids(a.ctx, &a.query, 1, &h, &ids) : tsgpu_keyword_search_batch(a.ctx, &a.query, 1, &h);
    if (rc != TSGPU_OK) return rc;
    if (status != TSGPU_OK) { tsgpu_id_lists_free(ids); search_cutoff = search_cutoff || cutoff != 0; return status; }
    if (topster != nullptr) {
        for (uint32_t i = 0; i < n_hits; i++) {
            KV kv(a.query_index, keys[i], keys[i], msi[i], &scores[(size_t)i * 3]);
            kv.text_match_score = sorts_on_text_match ? (msi[i] >= 0 ? scores[(size_t)i * 3 + (size_t)msi[i]] : 0) : text_match[i];
            topster->add(&kv);
        }
    }
    if (ids) {
        const uint64_t n = tsgpu_id_lists_count(ids, 0);
        const uint32_t* p = tsgpu_id_lists_ids(ids, 0);
        id_buff.insert(id_buff.end(), p, p + n);
        tsgpu_id_lists_free(ids);
    }
    num_keyword_matches = (size_t)num_matched;
    search_cutoff = search_cutoff || cutoff != 0;
    return TSGPU_OK;
}

}  // namespace tsgpu
