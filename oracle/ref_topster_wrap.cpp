// ORACLE — TEST INFRASTRUCTURE ONLY.
// Thin C wrapper that compiles the REFERENCE's own Topster — /root/reference/include/topster.h with loglogbeta.h, count_min_sketch.h,
// sparsepp.h and wyhash_v5.h, where they lie (never copied) — into oracle/_ref/libref_topster.so, so that the restated group-by collector
// (oracle/group_topster.h) and the restated wyhash / LogLogBeta can be checked against the real thing on arbitrary inputs.
// Shims: oracle/ref_shim/topster/{field.h, filter_result_iterator.h} (what they stand in for is said there).
// populate_result_kvs lives in src/index.cpp (not compilable here): its grouped branch (src/index.cpp:8962-9011) is spelled out below
// over the reference's Topster objects.
#include "inc/topster.h"
#include <cstring>
#include <string>
#include <vector>

extern "C" {

uint64_t ref_hash_wy(const void* key, uint64_t len) { return StringUtils::hash_wy(key, len); }
uint64_t ref_hash_combine(uint64_t a, uint64_t b) { return StringUtils::hash_combine(a, b); }

uint64_t ref_loglog_of_keys(const uint64_t* dkeys, uint64_t n) {
    LogLogBeta c;
    for (uint64_t i = 0; i < n; i++) c.add(std::to_string(dkeys[i]));
    return c.cardinality();
}

// Feeds n KVs (key, distinct_key, scores[3]) to Topster<KV>(capacity, distinct, first_pass) in order.
// ret_out[i] = add()'s return value. First pass: out_* = the heap array as it lies (size entries), *groups_count = getGroupsCount().
// Second pass: the groups in populate_result_kvs order; group_size[g] KVs each, concatenated in keys / scores.
int32_t ref_topster_run(uint32_t capacity, uint32_t distinct, int32_t first_pass, uint32_t n, const uint64_t* keys, const uint64_t* dkeys,
                        const int64_t* scores, int32_t* ret_out, uint32_t group_cap, uint32_t kv_cap, uint32_t* n_groups, uint32_t* group_size,
                        uint64_t* distinct_key, uint64_t* out_keys, int64_t* out_scores, uint64_t* groups_count) {
    Topster<KV> topster(capacity, distinct, first_pass != 0);
    for (uint32_t i = 0; i < n; i++) {
        KV kv(0, keys[i], dkeys[i], 0, scores + (size_t)i * 3);
        const int r = topster.add(&kv);
        if (ret_out) ret_out[i] = r;
    }
    topster.sort();
    uint32_t ng = 0;
    size_t at = 0;
    auto put = [&](KV* kv) {
        if (at >= kv_cap) return;
        out_keys[at] = kv->key;
        for (int j = 0; j < 3; j++) out_scores[at * 3 + j] = kv->scores[j];
        at++;
    };
    if (topster.distinct && !first_pass) {
        Topster<KV> gtopster(topster.MAX_SIZE);
        for (auto& group_topster : topster.group_kv_map) {
            group_topster.second->sort();
            if (group_topster.second->size != 0) gtopster.add(group_topster.second->getKV(0));
        }
        gtopster.sort();
        for (size_t i = 0; i < gtopster.size && ng < group_cap; i++) {
            KV* kv = gtopster.getKV(i);
            auto* g = topster.group_kv_map[kv->distinct_key];
            group_size[ng] = g->size;
            distinct_key[ng] = kv->distinct_key;
            for (uint32_t j = 0; j < g->size; j++) put(g->kvs[j]);
            ng++;
        }
        *groups_count = 0;
    } else {
        for (uint32_t t = 0; t < topster.size && ng < group_cap; t++) {
            group_size[ng] = 1;
            distinct_key[ng] = topster.getDistinctKeyAt(t);
            put(topster.getKV(t));
            ng++;
        }
        *groups_count = topster.getGroupsCount();
    }
    *n_groups = ng;
    return 0;
}
}
