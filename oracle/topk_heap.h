// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/for_codec.h header).
// Restates the bounded top-K collector of the reference (non-group-by form):
//   KV                 : /root/reference/include/topster.h:20-168 (is_greater/is_smaller :146-154)
//   Topster ctor       : :266-296      swapMe :311-319
//   Topster::add       : :321-466 (distinct == 0 path)       sort :469-473
// Group-by (distinct > 0), loglog counter and count-min sketch are not restated (out of scope,
// SURVEY §2b). array_index is deliberately NOT refreshed by sort(), exactly as in the reference:
// the hybrid branch (src/index.cpp:4094-4211) keeps calling add() on the sorted array.
#pragma once
#include <cstdint>
#include <tuple>
#include <algorithm>
#include <unordered_map>

namespace oracle {

struct KV {
    int8_t match_score_index = 0;
    uint16_t query_index = 0;
    uint16_t array_index = 0;
    uint64_t key = 0;
    uint64_t distinct_key = 0;
    int64_t scores[3] = {0, 0, 0};
    float vector_distance = -1.0f;
    int64_t text_match_score = 0;

    KV() = default;
    KV(uint16_t query_index, uint64_t key, uint64_t distinct_key, int8_t match_score_index, const int64_t* s,
       float vector_distance = -1.0f)
        : match_score_index(match_score_index), query_index(query_index), array_index(0), key(key),
          distinct_key(distinct_key), vector_distance(vector_distance) {
        scores[0] = s[0]; scores[1] = s[1]; scores[2] = s[2];
        if (match_score_index >= 0) text_match_score = s[match_score_index];
    }

    static bool is_greater(const KV* i, const KV* j) {
        return std::tie(i->scores[0], i->scores[1], i->scores[2], i->key) >
               std::tie(j->scores[0], j->scores[1], j->scores[2], j->key);
    }
    static bool is_smaller(const KV* i, const KV* j) {
        return std::tie(i->scores[0], i->scores[1], i->scores[2], i->key) <
               std::tie(j->scores[0], j->scores[1], j->scores[2], j->key);
    }
};

struct Topster {
    const uint32_t MAX_SIZE;
    uint32_t size = 0;
    KV* data;
    KV** kvs;
    std::unordered_map<uint64_t, KV*> map;

    explicit Topster(size_t capacity) : MAX_SIZE((uint32_t)capacity) {
        data = new KV[capacity];
        kvs = new KV*[capacity];
        for (size_t i = 0; i < capacity; i++) {
            data[i].match_score_index = 0;
            data[i].query_index = 0;
            data[i].array_index = (uint16_t)i;
            data[i].key = 0;
            data[i].distinct_key = 0;
            kvs[i] = &data[i];
        }
    }
    ~Topster() { delete[] data; delete[] kvs; }
    Topster(const Topster&) = delete;

    static void swapMe(KV** a, KV** b) {
        KV* t = *a; *a = *b; *b = t;
        uint16_t ai = (*a)->array_index;
        (*a)->array_index = (*b)->array_index;
        (*b)->array_index = ai;
    }

    int add(KV* kv) {
        bool less_than_min_heap = (size >= MAX_SIZE) && KV::is_smaller(kv, kvs[0]);
        if (less_than_min_heap) return 0;

        size_t heap_op_index = 0;
        bool SIFT_DOWN = true;
        const uint64_t key = kv->key;
        const auto found_it = map.find(key);
        if (found_it != map.end()) {
            KV* existing = found_it->second;
            if (KV::is_smaller(kv, existing)) return 0;
            SIFT_DOWN = true;
            heap_op_index = existing->array_index;
            map.erase(kvs[heap_op_index]->key);
        } else {
            if (size < MAX_SIZE) {
                SIFT_DOWN = false;
                heap_op_index = size;
                size++;
            } else {
                SIFT_DOWN = true;
                heap_op_index = 0;
                map.erase(kvs[heap_op_index]->key);
            }
        }
        map.emplace(key, kvs[heap_op_index]);

        kv->array_index = (uint16_t)heap_op_index;
        *kvs[heap_op_index] = *kv;

        if (SIFT_DOWN) {
            while ((2 * heap_op_index + 1) < size) {
                uint32_t next = (uint32_t)(2 * heap_op_index + 1);
                if (next + 1 < size && KV::is_greater(kvs[next], kvs[next + 1])) next++;
                if (KV::is_greater(kvs[heap_op_index], kvs[next])) swapMe(&kvs[heap_op_index], &kvs[next]);
                else break;
                heap_op_index = next;
            }
        } else {
            while (heap_op_index > 0) {
                uint32_t parent = (uint32_t)((heap_op_index - 1) / 2);
                if (KV::is_greater(kvs[parent], kvs[heap_op_index])) {
                    swapMe(&kvs[heap_op_index], &kvs[parent]);
                    heap_op_index = parent;
                } else break;
            }
        }
        return 1;
    }

    void sort() { std::stable_sort(kvs, kvs + size, KV::is_greater); }
    void clear() { size = 0; }
    uint64_t getKeyAt(uint32_t i) { return kvs[i]->key; }
    KV* getKV(uint32_t i) { return kvs[i]; }
};

}  // namespace oracle
