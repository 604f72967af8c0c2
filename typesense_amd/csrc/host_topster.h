// host_topster.h — the product's host-side bounded top-K collector with the semantics the reference's callers rely
// on (include/topster.h: KV :20-168, Topster::add :321-466, sort :469-473; non-group-by form). Used only where
// the reference itself works on <= a few hundred already-scored hits on the host: the rank-fusion step of hybrid
// search (src/index.cpp:4094-4211 keeps calling add() on the *sorted* array), the wildcard-vector branch
// (src/index.cpp:3682-3725) and the multi-GPU shard merge. Scoring/intersection never runs here.
#pragma once
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <unordered_map>
#include <vector>

namespace tsgpu {

struct HostKV {
    int8_t match_score_index = 0;
    uint16_t array_index = 0;
    uint64_t key = 0;
    int64_t scores[3] = {0, 0, 0};
    float vector_distance = -1.0f;
    int64_t text_match_score = 0;
};

inline bool kv_greater(const HostKV* a, const HostKV* b) {
    if (a->scores[0] != b->scores[0]) return a->scores[0] > b->scores[0];
    if (a->scores[1] != b->scores[1]) return a->scores[1] > b->scores[1];
    if (a->scores[2] != b->scores[2]) return a->scores[2] > b->scores[2];
    return a->key > b->key;
}
inline bool kv_smaller(const HostKV* a, const HostKV* b) {
    if (a->scores[0] != b->scores[0]) return a->scores[0] < b->scores[0];
    if (a->scores[1] != b->scores[1]) return a->scores[1] < b->scores[1];
    if (a->scores[2] != b->scores[2]) return a->scores[2] < b->scores[2];
    return a->key < b->key;
}

class HostTopster {
public:
    const uint32_t max_size;
    uint32_t size = 0;
    std::vector<HostKV> data;
    std::vector<HostKV*> kvs;
    std::unordered_map<uint64_t, HostKV*> map;

    explicit HostTopster(uint32_t capacity) : max_size(capacity), data(capacity), kvs(capacity) {
        for (uint32_t i = 0; i < capacity; i++) { data[i].array_index = (uint16_t)i; kvs[i] = &data[i]; }
    }

    // adopt a list that is already in sort() order (what the device returns): the state right after Topster::sort()
    void adopt_sorted(const HostKV* sorted, uint32_t n) {
        size = std::min(n, max_size);
        map.clear();
        for (uint32_t i = 0; i < size; i++) {
            const uint16_t keep = kvs[i]->array_index;
            *kvs[i] = sorted[i];
            kvs[i]->array_index = keep;
            map.emplace(sorted[i].key, kvs[i]);
        }
    }

    int add(HostKV* kv) {   // include/topster.h:341-465, distinct == 0
        if (size >= max_size && kv_smaller(kv, kvs[0])) return 0;
        size_t at = 0;
        bool sift_down = true;
        auto found = map.find(kv->key);
        if (found != map.end()) {
            HostKV* existing = found->second;
            if (kv_smaller(kv, existing)) return 0;
            at = existing->array_index;
            map.erase(kvs[at]->key);
        } else if (size < max_size) {
            sift_down = false;
            at = size++;
        } else {
            at = 0;
            map.erase(kvs[at]->key);
        }
        map.emplace(kv->key, kvs[at]);
        kv->array_index = (uint16_t)at;
        *kvs[at] = *kv;
        if (sift_down) {
            while (2 * at + 1 < size) {
                size_t next = 2 * at + 1;
                if (next + 1 < size && kv_greater(kvs[next], kvs[next + 1])) next++;
                if (!kv_greater(kvs[at], kvs[next])) break;
                swap_slots(at, next);
                at = next;
            }
        } else {
            while (at > 0) {
                size_t parent = (at - 1) / 2;
                if (!kv_greater(kvs[parent], kvs[at])) break;
                swap_slots(at, parent);
                at = parent;
            }
        }
        return 1;
    }

    void sort() { std::stable_sort(kvs.begin(), kvs.begin() + size, kv_greater); }

private:
    void swap_slots(size_t a, size_t b) {
        std::swap(kvs[a], kvs[b]);
        std::swap(kvs[a]->array_index, kvs[b]->array_index);
    }
};

// Index::float_to_int64_t / int64_t_to_float, src/index.cpp:266-284
inline int64_t float_to_int64(float f) { int32_t i; memcpy(&i, &f, 4); if (i < 0) i ^= INT32_MAX; return i; }
inline float int64_to_float(int64_t n) { int32_t i = (int32_t)n; if (i < 0) i ^= INT32_MAX; float f; memcpy(&f, &i, 4); return f; }

}  // namespace tsgpu
