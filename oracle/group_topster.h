// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/for_codec.h header).
// Restates the GROUP-BY form of the reference's bounded top-K collector and what sits around it on the scoring path:
//   Topster(capacity, distinct, is_group_by_first_pass)  : /root/reference/include/topster.h:266-296
//   Topster::add, first pass (one KV per distinct key)   : :342-349 (drop below the heap minimum, the key still reaches the counter),
//                                                           :378-428 (keyed by get_distinct_key; a smaller KV of a known group is dropped)
//   Topster::add, second pass (group_kv_map)             : :355-376 (ret = 2 for a seq_id seen before; one Topster(distinct) per group)
//   Topster::sort                                        : :469-473 (no-op when distinct != 0)
//   getGroupsCount / LogLogBeta                          : :491-493, /root/reference/include/loglogbeta.h:17-140
//   StringUtils::hash_wy / hash_combine                  : /root/reference/include/string_utils.h:316-326 (wyhash v5, include/wyhash_v5.h:69-94)
//   Index::populate_result_kvs, grouped branch           : /root/reference/src/index.cpp:8962-9011
//   groups_processed                                     : /root/reference/src/index.cpp:5546-5549
//   Index::get_distinct_id                               : /root/reference/src/index.cpp:7100-7142
// NOT restated: the count-min sketch of `sort_by: _group_found` (topster.h:327-340; such queries stay on the CPU path) and Union_KV.
// Pinned by oracle/golden_tests.cpp against TopsterTest.DistinctIntValues (test/topster_test.cpp:181-262) and, where the reference tree
// exists, by tests/test_oracle_groupby.py against oracle/_ref/libref_topster.so = the reference's own topster.h / loglogbeta.h / wyhash_v5.h.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include "topk_heap.h"

namespace oracle {

// ---- wyhash v5 with the default secret, as StringUtils::hash_wy calls it (seed 0) ----
namespace wy {
static const uint64_t P[6] = {0xa0761d6478bd642full, 0xe7037ed1a0b428dbull, 0x8ebc6af09c88c6e3ull,
                              0x589965cc75374cc3ull, 0x1d8e4e27c47d124full, 0x72b22b96e169b471ull};
inline uint64_t mum(uint64_t A, uint64_t B) { __uint128_t r = A; r *= B; return (uint64_t)(r >> 64) ^ (uint64_t)r; }
inline uint64_t mix(uint64_t A, uint64_t B) { return A ^ B ^ mum(A, B); }
inline uint64_t r8(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline uint64_t r4(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint64_t r3(const uint8_t* p, unsigned k) { return ((uint64_t)p[0] << 16) | ((uint64_t)p[k >> 1] << 8) | p[k - 1]; }
inline uint64_t core(const uint8_t* p, uint64_t len, uint64_t seed) {          // _wyhash, include/wyhash_v5.h:69-92
    uint64_t i = len;
    seed ^= P[4];
    if (i > 64) {
        uint64_t see1 = seed, see2 = seed, see3 = seed;
        for (; i > 64; i -= 64, p += 64) {
            seed = mix(r8(p) ^ P[0], r8(p + 8) ^ seed); see1 = mix(r8(p + 16) ^ P[1], r8(p + 24) ^ see1);
            see2 = mix(r8(p + 32) ^ P[2], r8(p + 40) ^ see2); see3 = mix(r8(p + 48) ^ P[3], r8(p + 56) ^ see3);
        }
        seed ^= see1 ^ see2 ^ see3;
    }
    if (i >= 8) {
        if (i <= 16) return mix(r8(p) ^ P[0], r8(p + i - 8) ^ seed);
        if (i <= 32) return mix(r8(p) ^ P[0], r8(p + 8) ^ seed) ^ mix(r8(p + i - 16) ^ P[1], r8(p + i - 8) ^ seed);
        return mix(r8(p) ^ P[0], r8(p + 8) ^ seed) ^ mix(r8(p + 16) ^ P[1], r8(p + 24) ^ seed)
               ^ mix(r8(p + i - 32) ^ P[2], r8(p + i - 24) ^ seed) ^ mix(r8(p + i - 16) ^ P[3], r8(p + i - 8) ^ seed);
    }
    if (i >= 4) return mix(r4(p) ^ P[0], r4(p + i - 4) ^ seed);
    return mix((i ? r3(p, (unsigned)i) : 0) ^ P[0], seed);
}
inline uint64_t hash(const void* key, uint64_t len, uint64_t seed = 0) { return mum(core((const uint8_t*)key, len, seed) ^ len, P[5]); }
}  // namespace wy

inline uint64_t hash_wy(const void* key, uint64_t len) {                        // string_utils.h:316-320
    const uint64_t h = wy::hash(key, len, 0);
    return h != std::numeric_limits<uint64_t>::max() ? h : (std::numeric_limits<uint64_t>::max() - 1);
}
inline uint64_t hash_combine(uint64_t combined, uint64_t hash) {                // string_utils.h:323-326
    combined ^= hash + 0x517cc1b727220a95ull + (combined << 6) + (combined >> 2);
    return combined;
}

// Index::get_distinct_id over the group_by fields of one document (src/index.cpp:5511-5520 + :7100-7142): hashes[f] = the facet hashes the
// field's hash index holds for the document (one for a plain field, every element's for an array; empty = no value).
// `missing` is set when some field has no value (the first pass collects those seq_ids, :7107-7109 / :7132-7134).
// exhausted[f] (optional) = field f's facet iterator was ALREADY invalid when the document was reached, i.e. an earlier visited document lay
// beyond the index's last id: the reference then overwrites whatever the earlier fields combined with seq_id (:7104-7111) — a state that
// depends on which documents the query visited before; with one group_by field both branches give seq_id.
inline uint64_t distinct_id_of(uint32_t seq_id, const std::vector<std::vector<uint32_t>>& hashes, bool group_missing_values, bool* missing = nullptr,
                               const std::vector<bool>* exhausted = nullptr) {
    uint64_t distinct_id = 1;
    bool miss = false;
    for (size_t f = 0; f < hashes.size(); f++) {
        const auto& hs = hashes[f];
        if (exhausted && (*exhausted)[f]) { if (!group_missing_values) distinct_id = seq_id; miss = true; continue; }
        if (hs.empty()) miss = true;
        for (uint32_t h : hs) distinct_id = hash_combine(distinct_id, h);
        if (distinct_id == 1 && !group_missing_values) distinct_id = seq_id;    // (evaluated after EVERY field, like the reference's call per field)
    }
    if (missing) *missing = miss;
    return distinct_id;
}

// ---- LogLogBeta, include/loglogbeta.h ----
class LogLogBeta {
public:
    static constexpr int PRECISION = 14;
    static constexpr uint32_t M = 1u << PRECISION;
    static constexpr int MAX_SHIFT = 64 - PRECISION;
    static constexpr uint64_t MAX_X = std::numeric_limits<uint64_t>::max() >> MAX_SHIFT;
    std::array<uint8_t, M> registers_;
    LogLogBeta() { registers_.fill(0); }
    void addHash(uint64_t x) {
        const uint32_t k = (uint32_t)(x >> MAX_SHIFT);
        const uint64_t shifted = (x << PRECISION) ^ MAX_X;
        const uint8_t val = (uint8_t)((shifted == 0 ? 64 : __builtin_clzll(shifted)) + 1);
        if (registers_[k] < val) registers_[k] = val;
    }
    void add(const std::string& value) { addHash(hash_wy(value.c_str(), value.size())); }
    static uint64_t cardinality_of(const uint8_t* regs) {
        const double ALPHA = 0.7213 / (1.0 + 1.079 / (double)M);
        double sum = 0.0, ez = 0.0;
        for (uint32_t i = 0; i < M; i++) {
            if (regs[i] == 0) ez += 1.0;
            sum += std::ldexp(1.0, -(int)regs[i]);
        }
        const double zl = std::log(ez + 1.0);
        const double beta = -0.370393911 * ez + 0.070471823 * zl + 0.17393686 * std::pow(zl, 2) + 0.16339839 * std::pow(zl, 3)
                            - 0.09237745 * std::pow(zl, 4) + 0.03738027 * std::pow(zl, 5) - 0.005384159 * std::pow(zl, 6)
                            + 0.00042419 * std::pow(zl, 7);
        const double m = (double)M;
        double estimate = ALPHA * m * (m - ez) / (beta + sum);
        if (estimate < 0.0) estimate = 0.0;
        return (uint64_t)estimate;
    }
    uint64_t cardinality() const { return cardinality_of(registers_.data()); }
};

// ---- the Topster with distinct != 0 ----
struct GroupTopster {
    const uint32_t MAX_SIZE;
    uint32_t size = 0;
    KV* data;
    KV** kvs;
    std::unordered_map<uint64_t, KV*> map;
    size_t distinct;
    std::unordered_set<uint64_t> group_doc_seq_ids;
    std::unordered_map<uint64_t, Topster*> group_kv_map;                         // (the per-group collectors are plain Topsters: Topster(distinct, 0, false))
    const bool is_group_by_first_pass;
    std::unique_ptr<LogLogBeta> loglog_counter;

    GroupTopster(size_t capacity, size_t distinct, bool first_pass)
        : MAX_SIZE((uint32_t)capacity), distinct(distinct), is_group_by_first_pass(first_pass) {
        data = new KV[capacity];
        kvs = new KV*[capacity];
        for (size_t i = 0; i < capacity; i++) { data[i].array_index = (uint16_t)i; kvs[i] = &data[i]; }
        if (first_pass) loglog_counter = std::make_unique<LogLogBeta>();
    }
    ~GroupTopster() { delete[] data; delete[] kvs; for (auto& g : group_kv_map) delete g.second; }
    GroupTopster(const GroupTopster&) = delete;

    int add(KV* kv) {
        int ret = 1;
        const bool less_than_min_heap = (size >= MAX_SIZE) && KV::is_smaller(kv, kvs[0]);
        size_t heap_op_index = 0;
        const bool second_pass = distinct && !is_group_by_first_pass;
        if (!second_pass && less_than_min_heap) {
            if (is_group_by_first_pass && loglog_counter) loglog_counter->add(std::to_string(kv->distinct_key));
            return 0;
        }
        bool SIFT_DOWN = true;
        if (second_pass) {
            if (group_doc_seq_ids.count(kv->key)) ret = 2;
            group_doc_seq_ids.emplace(kv->key);
            auto it = group_kv_map.find(kv->distinct_key);
            if (it != group_kv_map.end()) it->second->add(kv);
            else { auto* g = new Topster(distinct); g->add(kv); group_kv_map.insert({kv->distinct_key, g}); }
            return ret;
        }
        const uint64_t key = is_group_by_first_pass ? kv->distinct_key : kv->key;
        const auto found_it = map.find(key);
        auto key_of = [&](const KV* x) { return is_group_by_first_pass ? x->distinct_key : x->key; };
        if (found_it != map.end()) {
            KV* existing = found_it->second;
            if (KV::is_smaller(kv, existing)) return 0;
            heap_op_index = existing->array_index;
            map.erase(key_of(kvs[heap_op_index]));
        } else {
            if (is_group_by_first_pass && loglog_counter) loglog_counter->add(std::to_string(key));
            if (size < MAX_SIZE) { SIFT_DOWN = false; heap_op_index = size; size++; }
            else { heap_op_index = 0; map.erase(key_of(kvs[heap_op_index])); }
        }
        map.emplace(key, kvs[heap_op_index]);
        kv->array_index = (uint16_t)heap_op_index;
        *kvs[heap_op_index] = *kv;
        if (SIFT_DOWN) {
            while ((2 * heap_op_index + 1) < size) {
                uint32_t next = (uint32_t)(2 * heap_op_index + 1);
                if (next + 1 < size && KV::is_greater(kvs[next], kvs[next + 1])) next++;
                if (KV::is_greater(kvs[heap_op_index], kvs[next])) Topster::swapMe(&kvs[heap_op_index], &kvs[next]);
                else break;
                heap_op_index = next;
            }
        } else {
            while (heap_op_index > 0) {
                const uint32_t parent = (uint32_t)((heap_op_index - 1) / 2);
                if (KV::is_greater(kvs[parent], kvs[heap_op_index])) { Topster::swapMe(&kvs[heap_op_index], &kvs[parent]); heap_op_index = parent; }
                else break;
            }
        }
        return ret;
    }
    void sort() { if (!distinct) std::stable_sort(kvs, kvs + size, KV::is_greater); }
    size_t getGroupsCount() const { return loglog_counter ? loglog_counter->cardinality() : 0; }
};

// what a grouped keyword pass hands back
struct grouped_result_t {
    // first pass: one entry per group the Topster holds, in the heap's ARRAY order (sort() is a no-op; the reference's consumers read them as a set:
    // Index::get_group_by_values collects kvs.front()->key into a sorted vector, src/index.cpp:7144-7170)
    // second pass: result_kvs of populate_result_kvs — the groups in gtopster order, each with its KVs in the group Topster's sort() order
    std::vector<std::vector<KV>> groups;
    std::vector<uint32_t> group_found;                    // groups_processed[distinct_key] of each returned group
    uint64_t groups_count = 0;                            // first pass: getGroupsCount() (the LogLogBeta estimate)
    uint64_t groups_exact = 0;                            // distinct keys among the matched documents (not a reference quantity; the sketch's input size)
    std::array<uint8_t, LogLogBeta::M> loglog{};          // first pass: the sketch's registers
};

// populate_result_kvs (src/index.cpp:8962-9011) for a second-pass Topster; first pass: the heap as it lies
inline void populate_grouped(GroupTopster& topster, const std::unordered_map<uint64_t, uint32_t>& groups_processed, grouped_result_t& out) {
    if (topster.distinct && !topster.is_group_by_first_pass) {
        Topster gtopster(topster.MAX_SIZE);
        for (auto& g : topster.group_kv_map) {
            g.second->sort();
            if (g.second->size != 0) gtopster.add(g.second->getKV(0));
        }
        gtopster.sort();
        for (uint32_t i = 0; i < gtopster.size; i++) {
            const KV* head = gtopster.getKV(i);
            Topster* g = topster.group_kv_map[head->distinct_key];
            std::vector<KV> v;
            for (uint32_t j = 0; j < g->size; j++) v.push_back(*g->kvs[j]);
            out.groups.push_back(std::move(v));
            const auto it = groups_processed.find(head->distinct_key);
            out.group_found.push_back(it == groups_processed.end() ? 0 : it->second);
        }
        return;
    }
    for (uint32_t t = 0; t < topster.size; t++) {
        out.groups.push_back({*topster.kvs[t]});
        const auto it = groups_processed.find(topster.kvs[t]->distinct_key);
        out.group_found.push_back(it == groups_processed.end() ? 0 : it->second);
    }
    out.groups_count = topster.getGroupsCount();
    if (topster.loglog_counter) out.loglog = topster.loglog_counter->registers_;
}

}  // namespace oracle
