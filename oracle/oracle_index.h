// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/for_codec.h header).
// A minimal in-memory index + the query-time scoring path of the reference, restated:
//   offset encoding at index time  : /root/reference/src/index.cpp:1323-1348 (plain string), :1351-1395 (string[])
//   compact -> full at query time  : src/index.cpp:5637-5646, include/posting.h:45-46
//   get_field_token_its            : src/index.cpp:5598-5660
//   search_across_fields (lambda)  : src/index.cpp:5385-5596
//   compute_aggregated_score       : src/index.cpp:5227-5383
//   score_results2                 : src/index.cpp:6966-7098
//   compute_sort_scores (subset)   : src/index.cpp:5662-5907  (text_match / seq_id / int64 column / vector_distance)
//   float_to_int64_t / int64_t_to_float : src/index.cpp:266-284
//   topster sizing                 : src/index.cpp:3506-3512
//   flat vector scan               : src/index.cpp:3345-3374 ; normalize_vector include/index.h:379-388
//   wildcard vector branch         : src/index.cpp:3645-3732
//   hybrid rank fusion             : src/index.cpp:4036-4221 ; alpha default include/vector_query_ops.h:19
// Terms are integer ids: the ART dictionary (token string -> posting pointer) is upstream of the
// path and out of scope. typos>0, group-by, joins,
// geo/str/eval sorts are not restated.
//
// Vector distance follows hnswlib's InnerProductSpace (typesense fork pinned at
// cmake/hnsw.cmake:3 / WORKSPACE:186-191, source NOT under /root/reference): dist = 1 - sum(q_i*x_i),
// restated with hnswlib's published 16-lane accumulate + sequential horizontal add. The HNSW graph
// traversal (approximate) is NOT restated: "parity unpinned" for which approximate neighbours come
// back; the oracle is the exact flat scan (what process_results_bruteforce computes).
#pragma once
#include <string>
#include <cstring>
#include <cmath>
#include <cfloat>
#include <unordered_map>
#include <memory>
#include <iterator>
#include "token_or_iter.h"
#include "topk_heap.h"
#include "group_topster.h"

namespace oracle {

static inline int64_t float_to_int64_t(float f) {  // index.cpp:266-274
    int32_t i;
    memcpy(&i, &f, sizeof i);
    if (i < 0) i ^= INT32_MAX;
    return i;
}
static inline float int64_t_to_float(int64_t n) {  // index.cpp:276-285
    int32_t i = (int32_t)n;
    if (i < 0) i ^= INT32_MAX;
    float f;
    memcpy(&f, &i, sizeof f);
    return f;
}

enum text_match_type_t { max_score = 0, max_weight = 1, sum_score = 2 };  // include/index.h text_match_type_t
enum sort_kind_t { SORT_TEXT_MATCH = 0, SORT_SEQ_ID = 1, SORT_INT64_COLUMN = 2, SORT_VECTOR_DISTANCE = 3 };
enum vector_distance_type_t { ip = 0, cosine = 1 };                       // include/field.h:92-95

struct sort_by_t { int kind = SORT_TEXT_MATCH; int column = 0; int order = 1; /* 1 = DESC, -1 = ASC */ };

static const size_t COMPACT_LIST_THRESHOLD_LENGTH = 64;  // include/posting.h:45
static const size_t MAX_BLOCK_ELEMENTS = 256;            // include/posting.h:46
static const size_t DEFAULT_TOPSTER_SIZE = 250;          // include/index.h:679
static const int FIELD_MAX_WEIGHT = 15;                  // include/index.h:667

// tagged posting value, like art_leaf::values (include/posting.h:9-12)
struct posting_value_t {
    posting_list_t* full = nullptr;
    compact_posting_list_t* compact = nullptr;
    ~posting_value_t() { delete full; delete compact; }
    size_t num_ids() const { return full ? full->num_ids() : (compact ? compact->num_ids() : 0); }
};

struct field_index_t {
    bool is_array = false;
    std::unordered_map<uint32_t, std::unique_ptr<posting_value_t>> terms;
};

struct search_field_t { uint32_t field = 0; int64_t weight = FIELD_MAX_WEIGHT; };

struct keyword_query_t {
    std::vector<uint32_t> tokens;            // term ids in query order
    std::vector<search_field_t> fields;      // query_by fields (+ weights)
    int match_type = max_score;
    bool prioritize_exact_match = true;
    bool prioritize_token_position = false;
    bool prioritize_num_matching_fields = true;
    uint32_t total_cost = 0;                 // sum(2*typo_cost + is_prefix), index.cpp:7233-7235
    std::vector<sort_by_t> sort;             // <= 3
    size_t fetch_size = 10;                  // offset + per_page
    std::vector<uint32_t> excluded_ids;      // sorted
    std::vector<uint32_t> filter_ids;        // sorted; empty = no filter
    uint64_t search_stop_us = UINT64_MAX;
    size_t topster_size = 0;                 // 0 = the reference's sizing rule; >0 = explicit Topster capacity (tests)
    std::vector<uint32_t> dropped_tokens;    // dropped_tokens of search_across_fields (index.cpp:5427-5464): scored when present, never required
    // synonym passes (index.cpp:5292-5294, 6989-6994, 7024-7060): the query is a synonym's expansion
    int syn_orig_num_tokens = -1;            // tokens of the phrase the synonym stands for (-1 = not a synonym pass)
    int orig_num_tokens = 0;                 // tokens of the user's query
    bool is_synonym_query = false;
    bool demote_synonym_match = false;
};

struct keyword_result_t {
    std::vector<KV> kvs;                     // topster->sort() order
    std::vector<uint32_t> result_ids;        // every emitted id, ascending (id_buff / all_result_ids)
    size_t num_keyword_matches = 0;
    bool search_cutoff = false;
};

struct vector_query_t {
    std::vector<float> values;
    size_t k = 0;
    float distance_threshold = FLT_MAX;
    float alpha = 0.3f;                      // include/vector_query_ops.h:19
    bool rerank_hybrid_matches = false;      // search_params->rerank_hybrid_matches -> compute_aux_scores (index.cpp:4234-4240)
    size_t flat_search_cutoff = 0;           // include/vector_query_ops.h:13: a filter matching FEWER ids takes the flat branch (index.cpp:3664)
    uint32_t seq_id = 0;                     // vec:([], id: X): the query is document X's stored vector ...
    bool query_doc_given = false;            // ... and X itself is left out of the results (index.cpp:3651-3654, :3686)
};

struct vec_hit_t { float dist; uint32_t seq_id; };

class Index {
public:
    std::vector<field_index_t> fields;
    std::vector<std::unordered_map<uint32_t, int64_t>> sort_index;  // numeric columns: seq_id -> value (index.h:442)
    uint32_t num_docs = 0;

    // vector field
    size_t num_dim = 0;
    int distance_type = ip;
    std::vector<float> vec_store;                        // row-major [n_rows][num_dim] (hnswlib level-0 data)
    std::vector<uint32_t> vec_labels;                    // row -> label (= seq_id, index.cpp:1052-1054)
    std::unordered_map<uint32_t, uint32_t> vec_row_of;   // label_lookup_: label -> row

    explicit Index(size_t n_fields = 1, size_t n_columns = 1) : fields(n_fields), sort_index(n_columns) {}

    // ---------------- index time ----------------
    // one document's field = token id sequence (plain string field), index.cpp:1323-1348
    void index_plain_field(uint32_t seq_id, uint32_t field, const std::vector<uint32_t>& tokens) {
        std::unordered_map<uint32_t, std::vector<uint32_t>> token_to_offsets;
        std::vector<uint32_t> order;
        for (size_t i = 0; i < tokens.size(); i++) {
            if (!token_to_offsets.count(tokens[i])) order.push_back(tokens[i]);
            token_to_offsets[tokens[i]].push_back((uint32_t)i + 1);
        }
        if (!tokens.empty()) token_to_offsets[tokens.back()].push_back(0);
        for (uint32_t t : order) upsert_posting(field, t, seq_id, token_to_offsets[t]);
        if (seq_id + 1 > num_docs) num_docs = seq_id + 1;
    }

    // string[] field, index.cpp:1351-1395
    void index_array_field(uint32_t seq_id, uint32_t field, const std::vector<std::vector<uint32_t>>& elems) {
        fields[field].is_array = true;
        std::unordered_map<uint32_t, std::vector<uint32_t>> token_positions;
        std::vector<uint32_t> order;
        for (size_t array_index = 0; array_index < elems.size(); array_index++) {
            std::unordered_map<uint32_t, bool> seen_here;
            std::vector<uint32_t> here_order;
            uint32_t last_token = 0;
            bool any = false;
            for (size_t i = 0; i < elems[array_index].size(); i++) {
                uint32_t tok = elems[array_index][i];
                if (!token_positions.count(tok)) order.push_back(tok);
                if (!seen_here.count(tok)) { seen_here[tok] = true; here_order.push_back(tok); }
                token_positions[tok].push_back((uint32_t)i + 1);
                last_token = tok;
                any = true;
            }
            for (uint32_t tok : here_order) {
                auto& v = token_positions[tok];
                v.push_back(v.back());                 // repeat last position: end of this element
                v.push_back((uint32_t)array_index);
                if (any && tok == last_token) v.push_back(0);  // token is this element's last token
            }
        }
        for (uint32_t t : order) upsert_posting(field, t, seq_id, token_positions[t]);
        if (seq_id + 1 > num_docs) num_docs = seq_id + 1;
    }

    // posting_t::upsert (src/posting.cpp:247-288): compact while <= 64 cells, else full
    void upsert_posting(uint32_t field, uint32_t term, uint32_t seq_id, const std::vector<uint32_t>& offsets) {
        auto& slot = fields[field].terms[term];
        if (!slot) slot.reset(new posting_value_t);
        if (slot->full) { slot->full->upsert(seq_id, offsets); return; }
        if (!slot->compact) slot->compact = new compact_posting_list_t;
        slot->compact->upsert(seq_id, offsets.data(), (uint32_t)offsets.size());
        if (slot->compact->id_offsets.size() > COMPACT_LIST_THRESHOLD_LENGTH) {
            slot->full = slot->compact->to_full_posting_list((uint16_t)MAX_BLOCK_ELEMENTS);
            delete slot->compact;
            slot->compact = nullptr;
        }
    }

    // bulk: whole posting list for one term (ids ascending), as produced by sequential upserts
    void load_posting(uint32_t field, uint32_t term, const uint32_t* ids, const uint32_t* offset_index,
                      const uint32_t* offsets, uint32_t n, uint32_t n_offsets) {
        auto& slot = fields[field].terms[term];
        slot.reset(new posting_value_t);
        if ((size_t)n_offsets + 2 * (size_t)n <= COMPACT_LIST_THRESHOLD_LENGTH) {
            slot->compact = compact_posting_list_t::create(n, ids, offset_index, n_offsets, offsets);
        } else {
            slot->full = new posting_list_t((uint16_t)MAX_BLOCK_ELEMENTS);
            slot->full->load_sorted(ids, offset_index, offsets, n, n_offsets);
        }
        if (n && ids[n - 1] + 1 > num_docs) num_docs = ids[n - 1] + 1;
    }

    void set_sort_value(uint32_t column, uint32_t seq_id, int64_t v) { sort_index[column][seq_id] = v; }

    // ---------------- vector index (flat) ----------------
    static void normalize_vector(const std::vector<float>& src, std::vector<float>& norm_dest) {  // include/index.h:379-388
        float norm = 0.0f;
        for (float v : src) norm += v * v;
        norm = 1.0f / (sqrtf(norm) + 1e-30f);
        for (size_t i = 0; i < src.size(); i++) norm_dest[i] = src[i] * norm;
    }

    void vec_init(size_t dim, int dist_type) { num_dim = dim; distance_type = dist_type; }
    void vec_add(uint32_t label, const float* v) {  // index.cpp:1040-1054
        std::vector<float> x(v, v + num_dim);
        if (distance_type == cosine) { std::vector<float> n(num_dim); normalize_vector(x, n); x.swap(n); }
        auto it = vec_row_of.find(label);
        if (it != vec_row_of.end()) {                       // addPoint on an existing label updates in place
            std::copy(x.begin(), x.end(), vec_store.begin() + (size_t)it->second * num_dim);
        } else {
            vec_row_of.emplace(label, (uint32_t)vec_labels.size());
            vec_labels.push_back(label);
            vec_store.insert(vec_store.end(), x.begin(), x.end());
        }
        if (label + 1 > num_docs) num_docs = label + 1;
    }
    const float* vec_get(uint32_t label) const {            // getDataByLabel; nullptr = "throws"
        auto it = vec_row_of.find(label);
        return it == vec_row_of.end() ? nullptr : vec_store.data() + (size_t)it->second * num_dim;
    }

    // hnswlib InnerProductSpace distance: 1 - <a,b> (space_ip.h of the pinned fork; NOT under /root/reference — restated from the
    // published hnswlib 0.7/0.8 source). The summation ORDER depends on the SIMD level hnswlib was COMPILED for,
    // because the header is compiled into Typesense's own translation units with Typesense's flags:
    //   ip_lanes = 4  (SSE)    the reference's stock build: CMakeLists.txt:8 / BUILD:81-88 pass no -march/-mavx, so only __SSE__ is
    //                          defined on x86-64 -> InnerProductSIMD16ExtSSE / SIMD4ExtSSE: ONE __m128 accumulator, element i goes to
    //                          lane i % 4, final sum T0+T1+T2+T3. DEFAULT.
    //   ip_lanes = 8  (AVX)    -mavx builds: SIMD16ExtAVX = one __m256 (lane i % 8, T0+..+T7); SIMD4ExtAVX = __m256 over the 16-multiples,
    //                          folded lo+hi into an __m128 that takes the remaining 4-groups.
    //   ip_lanes = 16 (AVX512) -mavx512f builds: SIMD16ExtAVX512 = one __m512 (lane i % 16, T0+..+T15); SIMD4Ext = the AVX form.
    // Residual forms (dim % 4 != 0): SIMD part + scalar tail, 1 - (res + tail). Multiply and add are rounded separately everywhere
    // (intrinsics _mm*_mul_ps + _mm*_add_ps; compiled with -ffp-contract=off here). Pinned by the reference's own distance known
    // answers at dims 4 and 5 (golden_tests.cpp: test_vector_reference_pins), where the three orders coincide.
    static int& ip_lanes() { static int lanes = 4; return lanes; }
    // lanes[l] += a[i + l] * b[i + l] for i = from, from + L, ..: written on 4-float vector registers (mulps + addps, as the
    // intrinsics compile to) so that the CPU baseline pays what hnswlib's loop pays; lane-wise identical to the scalar statement
    typedef float ip_v4sf __attribute__((vector_size(16), aligned(4), may_alias));
    template <int L>
    static void ip_accumulate(float* lanes, const float* a, const float* b, size_t from, size_t to) {
        ip_v4sf acc[L / 4];
        for (int v = 0; v < L / 4; v++) acc[v] = *(const ip_v4sf*)(lanes + 4 * v);
        for (size_t i = from; i < to; i += L)
            for (int v = 0; v < L / 4; v++) acc[v] = acc[v] + *(const ip_v4sf*)(a + i + 4 * v) * *(const ip_v4sf*)(b + i + 4 * v);
        for (int v = 0; v < L / 4; v++) *(ip_v4sf*)(lanes + 4 * v) = acc[v];
    }
    static float ip_simd16ext(const float* a, const float* b, size_t qty) {              // qty % 16 == 0
        float lanes[16] = {0};
        float sum = 0;
        const int L = ip_lanes();
        if (L == 16) ip_accumulate<16>(lanes, a, b, 0, qty); else if (L == 8) ip_accumulate<8>(lanes, a, b, 0, qty); else ip_accumulate<4>(lanes, a, b, 0, qty);
        if (L == 4) return lanes[0] + lanes[1] + lanes[2] + lanes[3];
        for (int l = 0; l < L; l++) sum += lanes[l];
        return sum;
    }
    static float ip_simd4ext(const float* a, const float* b, size_t qty) {               // qty % 4 == 0
        float lanes[8] = {0};
        if (ip_lanes() == 4) { ip_accumulate<4>(lanes, a, b, 0, qty); return lanes[0] + lanes[1] + lanes[2] + lanes[3]; }
        const size_t q16 = qty / 16 * 16;                                                 // InnerProductSIMD4ExtAVX
        ip_accumulate<8>(lanes, a, b, 0, q16);
        float s4[4];
        for (int l = 0; l < 4; l++) s4[l] = lanes[l] + lanes[l + 4];
        ip_accumulate<4>(s4, a, b, q16, qty);
        return s4[0] + s4[1] + s4[2] + s4[3];
    }
    static float ip_scalar(const float* a, const float* b, size_t n) {
        float r = 0;
        for (size_t i = 0; i < n; i++) r += a[i] * b[i];
        return r;
    }
    static float ip_distance(const float* a, const float* b, size_t dim) {
        if (dim % 16 == 0) return 1.0f - ip_simd16ext(a, b, dim);
        if (dim % 4 == 0) return 1.0f - ip_simd4ext(a, b, dim);
        if (dim > 16) { size_t q = dim >> 4 << 4; return 1.0f - (ip_simd16ext(a, b, q) + ip_scalar(a + q, b + q, dim - q)); }
        if (dim > 4) { size_t q = dim >> 2 << 2; return 1.0f - (ip_simd4ext(a, b, q) + ip_scalar(a + q, b + q, dim - q)); }
        return 1.0f - ip_scalar(a, b, dim);
    }

    // exact k nearest: k smallest (dist, label) pairs, closest first (what searchKnnCloserFirst returns
    // when the graph search is exact). allow = optional sorted id whitelist (filter / VectorFilterFunctor).
    std::vector<vec_hit_t> flat_knn(const std::vector<float>& q_in, size_t k, const std::vector<uint32_t>* allow = nullptr,
                                    const std::vector<uint32_t>* excluded = nullptr) const {
        std::vector<float> q = q_in;
        if (distance_type == cosine) { std::vector<float> n(q.size()); normalize_vector(q_in, n); q.swap(n); }
        auto cmp = [](const vec_hit_t& a, const vec_hit_t& b) { return a.dist < b.dist || (a.dist == b.dist && a.seq_id < b.seq_id); };
        std::vector<vec_hit_t> heap;                          // max-heap of the k best (dist, label) pairs
        heap.reserve(k + 1);
        for (size_t r = 0; r < vec_labels.size(); r++) {
            uint32_t label = vec_labels[r];
            if (allow && !std::binary_search(allow->begin(), allow->end(), label)) continue;
            if (excluded && std::binary_search(excluded->begin(), excluded->end(), label)) continue;
            vec_hit_t h{ip_distance(q.data(), vec_store.data() + r * num_dim, num_dim), label};
            if (heap.size() < k) { heap.push_back(h); std::push_heap(heap.begin(), heap.end(), cmp); }
            else if (k > 0 && cmp(h, heap.front())) { std::pop_heap(heap.begin(), heap.end(), cmp); heap.back() = h; std::push_heap(heap.begin(), heap.end(), cmp); }
        }
        std::sort(heap.begin(), heap.end(), cmp);
        return heap;
    }

    // ---------------- query time: keyword ----------------
    size_t topster_size(size_t fetch_size, size_t n_filter) const {  // index.cpp:3506-3512
        size_t s = std::max<size_t>(fetch_size, DEFAULT_TOPSTER_SIZE);
        if (n_filter != 0) s = std::min<size_t>(s, n_filter); else s = std::min<size_t>(s, num_docs);
        return std::max<size_t>(1, s);
    }

    void compute_sort_scores(const std::vector<sort_by_t>& sort, uint32_t seq_id, int64_t max_field_match_score,
                             int64_t* scores, int64_t& match_score_index, float vector_distance) const {  // index.cpp:5662-5907
        const int64_t default_score = INT64_MIN;
        for (size_t i = 0; i < sort.size(); i++) {
            switch (sort[i].kind) {
                case SORT_TEXT_MATCH: scores[i] = max_field_match_score; match_score_index = (int64_t)i; break;
                case SORT_SEQ_ID: scores[i] = seq_id; break;
                case SORT_VECTOR_DISTANCE: scores[i] = float_to_int64_t(vector_distance); break;
                default: {
                    const auto& col = sort_index[sort[i].column];
                    auto it = col.find(seq_id);
                    scores[i] = (it == col.end()) ? default_score : it->second;
                }
            }
            if (sort[i].order == -1) scores[i] = -scores[i];
        }
    }

    void score_results2(bool field_is_array, uint32_t total_cost, int64_t& match_score, uint32_t seq_id,
                        bool prioritize_exact_match, bool single_exact_query_token, bool prioritize_token_position,
                        size_t num_query_tokens, int syn_orig_num_tokens, int orig_num_tokens, bool is_synonym_query, bool demote_synonym_match,
                        const std::vector<posting_list_t::iterator_t>& posting_lists) const {  // index.cpp:6966-7098
        if (posting_lists.size() <= 1) {
            const uint8_t is_verbatim = uint8_t(prioritize_exact_match && single_exact_query_token &&
                                                posting_list_t::is_single_token_verbatim_match(posting_lists[0], field_is_array));
            size_t words_present = (num_query_tokens == 1 && is_synonym_query) ? syn_orig_num_tokens : 1;
            size_t distance = (num_query_tokens == 1 && is_synonym_query) ? syn_orig_num_tokens - 1 : 0;
            size_t max_offset = prioritize_token_position ? posting_list_t::get_last_offset(posting_lists[0], field_is_array) : 255;
            uint8_t synonym_score = (is_synonym_query && demote_synonym_match) ? 0 : 1;
            Match m((uint8_t)words_present, (uint8_t)distance, (uint8_t)max_offset, is_verbatim);
            match_score = (int64_t)m.get_match_score(total_cost, (uint32_t)words_present, synonym_score);
            return;
        }
        std::map<size_t, std::vector<token_positions_t>> array_token_positions;
        posting_list_t::get_offsets(posting_lists, array_token_positions);
        for (const auto& kv : array_token_positions) {
            const std::vector<token_positions_t>& token_positions = kv.second;
            if (token_positions.empty()) continue;
            const Match match(seq_id, token_positions, false, prioritize_exact_match);
            uint8_t synonym_score_in = (is_synonym_query && demote_synonym_match) ? 0 : 1;
            uint64_t s = match.get_match_score(total_cost, (uint32_t)posting_lists.size(), synonym_score_in);
            auto this_words_present = ((s >> 40) & 0xFF);
            auto unique_words = field_is_array ? this_words_present : ((s >> 32) & 0xFF);
            auto typo_score = ((s >> 24) & 0xFF);
            auto proximity = ((s >> 16) & 0xFF);
            auto verbatim = ((s >> 12) & 0xF);
            auto offset_score = prioritize_token_position ? ((s >> 4) & 0xFF) : 0;
            auto synonym_score = ((s >> 0) & 0xF);
            if (is_synonym_query && num_query_tokens == posting_lists.size()) {
                unique_words = syn_orig_num_tokens;
                this_words_present = syn_orig_num_tokens;
            }
            if (is_synonym_query && syn_orig_num_tokens > 0 && orig_num_tokens > 0) {
                double rel_factor = double(orig_num_tokens) / double(syn_orig_num_tokens);
                auto scale_component = [&](uint64_t v) -> uint64_t {
                    double scaled = double(v) * rel_factor;
                    if (scaled > 255.0) scaled = 255.0;
                    return (uint64_t)scaled;
                };
                this_words_present = scale_component(this_words_present);
                unique_words = scale_component(unique_words);
                auto reversed_typo_score = 255 - typo_score;
                reversed_typo_score = scale_component(reversed_typo_score);
                typo_score = 255 - reversed_typo_score;
                auto reversed_proximity = 100 - proximity;
                reversed_proximity = scale_component(reversed_proximity);
                proximity = 100 - reversed_proximity;
                auto reversed_offset_score = 255 - offset_score;
                reversed_offset_score = scale_component(reversed_offset_score);
                offset_score = prioritize_token_position ? 255 - reversed_offset_score : 0;
            }
            uint64_t mod = ((int64_t(this_words_present) << 40) | (int64_t(unique_words) << 32) | (int64_t(typo_score) << 24) |
                            (int64_t(proximity) << 16) | (int64_t(verbatim) << 12) | (int64_t(offset_score) << 4) |
                            (int64_t(synonym_score) << 0));
            if (mod > (uint64_t)match_score) match_score = (int64_t)mod;
        }
    }

    int64_t compute_aggregated_score(const std::vector<or_iterator_t>& its, const keyword_query_t& q, uint32_t seq_id,
                                     std::vector<or_iterator_t>* dropped_token_its = nullptr) const {  // index.cpp:5227-5383
        const size_t num_search_fields = q.fields.size();
        std::vector<std::vector<posting_list_t::iterator_t>> field_to_tokens(num_search_fields);
        size_t query_len = 0;
        for (size_t ti = 0; ti < its.size(); ti++) {
            const auto& field_iters = its[ti].get_its();
            bool found_token = false;
            for (size_t fi = 0; fi < field_iters.size(); fi++) {
                const auto& field_iter = field_iters[fi];
                if (field_iter.valid() && field_iter.id() == seq_id && field_iter.get_field_id() < num_search_fields) {
                    field_to_tokens[field_iter.get_field_id()].push_back(field_iter.clone());
                    found_token = true;
                }
            }
            if (found_token) query_len++;
        }
        // check if seq_id exists in any of the dropped_token iters (:5271-5290)
        for (size_t ti = 0; dropped_token_its && ti < dropped_token_its->size(); ti++) {
            or_iterator_t& token_fields_iters = (*dropped_token_its)[ti];
            if (token_fields_iters.skip_to(seq_id) && token_fields_iters.id() == seq_id) {
                const auto& field_iters = token_fields_iters.get_its();
                bool found_token = false;
                for (size_t fi = 0; fi < field_iters.size(); fi++) {
                    const auto& field_iter = field_iters[fi];
                    if (field_iter.id() == seq_id && field_iter.get_field_id() < num_search_fields) {
                        field_to_tokens[field_iter.get_field_id()].push_back(field_iter.clone());
                        found_token = true;
                    }
                }
                if (found_token) query_len++;
            }
        }
        if (q.syn_orig_num_tokens != -1) query_len = q.syn_orig_num_tokens;      // :5292-5294
        int64_t best_field_match_score = 0, best_field_weight = 0, sum_field_weighted_score = 0;
        uint32_t num_matching_fields = 0;
        for (size_t fi = 0; fi < field_to_tokens.size(); fi++) {
            const auto& token_postings = field_to_tokens[fi];
            if (token_postings.empty()) continue;
            const int64_t field_weight = q.fields[fi].weight;
            const bool field_is_array = fields[q.fields[fi].field].is_array;
            int64_t field_match_score = 0;
            bool single_exact_query_token = (q.total_cost == 0 && q.tokens.size() == 1);
            score_results2(field_is_array, q.total_cost, field_match_score, seq_id, q.prioritize_exact_match,
                           single_exact_query_token, q.prioritize_token_position, q.tokens.size(), q.syn_orig_num_tokens, q.orig_num_tokens,
                           q.is_synonym_query, q.demote_synonym_match, token_postings);
            if (q.match_type == max_score && field_match_score > best_field_match_score) {
                best_field_match_score = field_match_score; best_field_weight = field_weight;
            }
            if (q.match_type == max_weight && field_weight > best_field_weight) {
                best_field_weight = field_weight; best_field_match_score = field_match_score;
            }
            if (q.match_type == sum_score) sum_field_weighted_score += (field_weight * field_match_score);
            num_matching_fields++;
        }
        query_len = (best_field_match_score == 0) ? 0 : std::min<size_t>(15, query_len);
        auto max_field_weight = std::min<size_t>(FIELD_MAX_WEIGHT, (size_t)best_field_weight);
        num_matching_fields = (uint32_t)std::min<size_t>(7, num_matching_fields);
        if (!q.prioritize_num_matching_fields) num_matching_fields = 0;
        uint64_t agg;
        if (q.match_type == max_score)
            agg = ((int64_t(query_len) << 59) | (int64_t(best_field_match_score) << 11) | (int64_t(max_field_weight) << 3) | int64_t(num_matching_fields));
        else if (q.match_type == max_weight)
            agg = ((int64_t(query_len) << 59) | (int64_t(max_field_weight) << 51) | (int64_t(best_field_match_score) << 3) | int64_t(num_matching_fields));
        else
            agg = ((int64_t(query_len) << 59) | (int64_t(sum_field_weighted_score) << 3) | int64_t(num_matching_fields));
        return (int64_t)agg;
    }

    // get_field_token_its, index.cpp:5598-5660
    void get_field_token_its(const keyword_query_t& q, std::vector<or_iterator_t>& token_its, std::vector<posting_list_t*>& expanded_plists,
                             const std::vector<uint32_t>* tokens_in = nullptr, bool keep_empty = false) const {
        const std::vector<uint32_t>& toks = tokens_in ? *tokens_in : q.tokens;
        for (size_t ti = 0; ti < toks.size(); ti++) {
            std::vector<posting_list_t::iterator_t> its;
            for (size_t i = 0; i < q.fields.size(); i++) {
                const auto& fidx = fields[q.fields[i].field];
                auto leaf = fidx.terms.find(toks[ti]);
                if (leaf == fidx.terms.end()) continue;
                if (leaf->second->compact) {
                    posting_list_t* fl = leaf->second->compact->to_full_posting_list((uint16_t)MAX_BLOCK_ELEMENTS);
                    expanded_plists.push_back(fl);
                    its.push_back(fl->new_iterator(nullptr, nullptr, (uint32_t)i));
                } else {
                    its.push_back(leaf->second->full->new_iterator(nullptr, nullptr, (uint32_t)i));
                }
            }
            if (its.empty() && !keep_empty) continue;  // token absent from every field: silently skipped (:5651-5655)
            or_iterator_t token_fields(its);
            token_its.push_back(std::move(token_fields));
        }
    }

    // search_across_fields, index.cpp:5385-5596 (topster owned by the caller, like the reference)
    void search_across_fields(const keyword_query_t& q, Topster* topster, keyword_result_t& out, uint16_t query_index = 0) const {
        std::vector<or_iterator_t> token_its;
        std::vector<posting_list_t*> expanded_plists;
        get_field_token_its(q, token_its, expanded_plists);
        // one or_iterator per dropped token (:5427-5464; a token no field holds never matches a document: left out)
        std::vector<or_iterator_t> dropped_token_its;
        if (!q.dropped_tokens.empty()) get_field_token_its(q, dropped_token_its, expanded_plists, &q.dropped_tokens);

        result_iter_state_t istate(q.excluded_ids.data(), q.excluded_ids.size(), q.filter_ids.data(), q.filter_ids.size());
        deadline_t dl;
        dl.search_begin_us = deadline_t::now_us();
        dl.search_stop_us = q.search_stop_us;

        or_iterator_t::intersect(token_its, istate, dl, [&](single_filter_result_t& fr, const std::vector<or_iterator_t>& its) {
            uint32_t seq_id = fr.seq_id;
            if (topster == nullptr) { out.result_ids.push_back(seq_id); return; }
            int64_t aggregated_score = compute_aggregated_score(its, q, seq_id, &dropped_token_its);
            int64_t scores[3] = {0, 0, 0};
            int64_t match_score_index = -1;
            compute_sort_scores(q.sort, seq_id, aggregated_score, scores, match_score_index, 0);
            KV kv(query_index, seq_id, seq_id, (int8_t)match_score_index, scores);
            if (match_score_index != -1) {
                kv.scores[match_score_index] = aggregated_score;
                kv.text_match_score = aggregated_score;
            }
            topster->add(&kv);
            out.result_ids.push_back(seq_id);
        });

        out.num_keyword_matches = istate.num_keyword_matches;
        out.search_cutoff = dl.search_cutoff;
        for (auto* p : expanded_plists) delete p;
    }

    keyword_result_t search_keyword(const keyword_query_t& q) const {
        keyword_result_t out;
        Topster topster(q.topster_size ? q.topster_size : topster_size(q.fetch_size, q.filter_ids.size()));
        search_across_fields(q, &topster, out);
        topster.sort();
        for (uint32_t i = 0; i < topster.size; i++) out.kvs.push_back(*topster.getKV(i));
        return out;
    }

    // ---------------- group_by: search_across_fields with group_limit != 0 (index.cpp:5511-5520, 5546-5549) ----------------
    // distinct_ids / has_value: per seq_id, what Index::get_distinct_id yields for the query's group_by fields and whether every field
    // held a value (oracle::distinct_id_of computes both from the facet hashes; has_value only feeds missing_ids). One pass of the reference's two-pass protocol
    // (Index::run_search, index.cpp:2488-2760): first_pass = the Topster keyed by distinct key + the LogLogBeta counter; else the
    // second pass' group_kv_map followed by populate_result_kvs. missing_ids = group_by_missing_value_ids of a first pass.
    // one grouped pass of search_across_fields into the caller's collector (the reference's pass over the caller's Topster)
    void grouped_pass(const keyword_query_t& q, const std::vector<uint64_t>& distinct_ids, const std::vector<uint8_t>& has_value, bool group_missing_values,
                      GroupTopster& topster, std::unordered_map<uint64_t, uint32_t>& groups_processed, std::unordered_set<uint64_t>& seen, keyword_result_t& out,
                      uint16_t query_index, std::vector<uint32_t>* missing_ids) const {
        std::vector<or_iterator_t> token_its;
        std::vector<posting_list_t*> expanded_plists;
        get_field_token_its(q, token_its, expanded_plists);
        std::vector<or_iterator_t> dropped_token_its;
        if (!q.dropped_tokens.empty()) get_field_token_its(q, dropped_token_its, expanded_plists, &q.dropped_tokens);
        result_iter_state_t istate(q.excluded_ids.data(), q.excluded_ids.size(), q.filter_ids.data(), q.filter_ids.size());
        deadline_t dl;
        dl.search_begin_us = deadline_t::now_us();
        dl.search_stop_us = q.search_stop_us;
        const bool first_pass = topster.is_group_by_first_pass;
        or_iterator_t::intersect(token_its, istate, dl, [&](single_filter_result_t& fr, const std::vector<or_iterator_t>& its) {
            const uint32_t seq_id = fr.seq_id;
            const int64_t aggregated_score = compute_aggregated_score(its, q, seq_id, &dropped_token_its);
            // distinct_ids[seq_id] = get_distinct_id's result (a document without any value: seq_id, or 1 with group_missing_values, :7105-7106, :7137-7139)
            const uint64_t distinct_id = seq_id < distinct_ids.size() ? distinct_ids[seq_id] : (group_missing_values ? 1 : (uint64_t)seq_id);
            const bool valued = seq_id < distinct_ids.size() && (has_value.empty() || has_value[seq_id]);
            if (!valued && first_pass && missing_ids) missing_ids->push_back(seq_id);
            int64_t scores[3] = {0, 0, 0};
            int64_t match_score_index = -1;
            compute_sort_scores(q.sort, seq_id, aggregated_score, scores, match_score_index, 0);
            KV kv(query_index, seq_id, distinct_id, (int8_t)match_score_index, scores);
            if (match_score_index != -1) { kv.scores[match_score_index] = aggregated_score; kv.text_match_score = aggregated_score; }
            const int ret = topster.add(&kv);
            if (ret < 2) groups_processed[distinct_id]++;                       // :5546-5549
            seen.insert(distinct_id);
            out.result_ids.push_back(seq_id);
        });
        out.num_keyword_matches = istate.num_keyword_matches;
        out.search_cutoff = dl.search_cutoff;
        for (auto* p : expanded_plists) delete p;
    }

    keyword_result_t search_keyword_grouped(const keyword_query_t& q, const std::vector<uint64_t>& distinct_ids, const std::vector<uint8_t>& has_value,
                                            bool group_missing_values, size_t group_limit, bool first_pass, grouped_result_t& gout,
                                            std::vector<uint32_t>* missing_ids = nullptr) const {
        keyword_result_t out;
        GroupTopster topster(q.topster_size ? q.topster_size : topster_size(q.fetch_size, q.filter_ids.size()), group_limit, first_pass);
        std::unordered_map<uint64_t, uint32_t> groups_processed;
        std::unordered_set<uint64_t> seen;
        grouped_pass(q, distinct_ids, has_value, group_missing_values, topster, groups_processed, seen, out, 0, missing_ids);
        populate_grouped(topster, groups_processed, gout);
        gout.groups_exact = seen.size();
        return out;
    }

    // Index::search_all_candidates with group_limit != 0 (index.cpp:1794-1894 over :5511-5549): one grouped pass per candidate combination over ONE collector
    // and ONE groups_processed; KV::query_index = searched_queries.size() at the time of the pass; a second pass counts a document once (ret == 2 afterwards),
    // a first pass counts every add; all_result_ids = the union of the passes' ids.
    keyword_result_t search_candidates_grouped(const std::vector<keyword_query_t>& combos, const std::vector<uint64_t>& distinct_ids, const std::vector<uint8_t>& has_value,
                                               bool group_missing_values, size_t group_limit, bool first_pass, grouped_result_t& gout) const {
        keyword_result_t out;
        if (combos.empty()) return out;
        const keyword_query_t& q0 = combos[0];
        GroupTopster topster(q0.topster_size ? q0.topster_size : topster_size(q0.fetch_size, q0.filter_ids.size()), group_limit, first_pass);
        std::unordered_map<uint64_t, uint32_t> groups_processed;
        std::unordered_set<uint64_t> seen;
        std::vector<uint32_t> all_ids;
        uint16_t searched_queries = 0;
        for (const keyword_query_t& q : combos) {
            keyword_result_t pass;
            grouped_pass(q, distinct_ids, has_value, group_missing_values, topster, groups_processed, seen, pass, searched_queries, nullptr);
            out.num_keyword_matches = pass.num_keyword_matches;
            out.search_cutoff = out.search_cutoff || pass.search_cutoff;
            all_ids.insert(all_ids.end(), pass.result_ids.begin(), pass.result_ids.end());
            if (!pass.result_ids.empty()) searched_queries++;
        }
        std::sort(all_ids.begin(), all_ids.end());
        all_ids.erase(std::unique(all_ids.begin(), all_ids.end()), all_ids.end());
        out.result_ids = all_ids;
        populate_grouped(topster, groups_processed, gout);
        gout.groups_exact = seen.size();
        return out;
    }

    // ---------------- query time: candidate-token combinations, Index::search_all_candidates, index.cpp:1794-1894 ----------------
    // One search_across_fields pass per combination (each with its own tokens and total_cost) over ONE shared Topster — a key seen
    // again replaces its KV unless the new one is_smaller (include/topster.h:392-406: equal scores -> the LATER pass wins) — and one
    // shared id_buff that ends as the sorted-unique all_result_ids (index.cpp:5565-5578, 5081-5090). KV::query_index =
    // searched_queries.size() at the time of the pass; a pass is pushed to searched_queries iff it emitted ids (:5580-5585).
    // num_keyword_matches is ASSIGNED by every pass (:5553): the last pass's count survives.
    keyword_result_t search_candidates(const std::vector<keyword_query_t>& combos) const {
        keyword_result_t out;
        if (combos.empty()) return out;
        const keyword_query_t& q0 = combos[0];
        Topster topster(q0.topster_size ? q0.topster_size : topster_size(q0.fetch_size, q0.filter_ids.size()));
        std::vector<uint32_t> id_buff, all_ids;
        uint16_t searched_queries = 0;
        auto flush = [&]() {
            std::sort(id_buff.begin(), id_buff.end());
            id_buff.erase(std::unique(id_buff.begin(), id_buff.end()), id_buff.end());
            std::vector<uint32_t> merged;
            std::set_union(all_ids.begin(), all_ids.end(), id_buff.begin(), id_buff.end(), std::back_inserter(merged));   // ArrayUtils::or_scalar
            all_ids.swap(merged);
            id_buff.clear();
        };
        for (const keyword_query_t& q : combos) {
            keyword_result_t pass;
            search_across_fields(q, &topster, pass, searched_queries);
            out.num_keyword_matches = pass.num_keyword_matches;
            out.search_cutoff = out.search_cutoff || pass.search_cutoff;
            id_buff.insert(id_buff.end(), pass.result_ids.begin(), pass.result_ids.end());
            if (id_buff.size() > 100000) flush();
            if (!pass.result_ids.empty()) searched_queries++;
        }
        flush();
        out.result_ids = all_ids;
        topster.sort();
        for (uint32_t i = 0; i < topster.size; i++) out.kvs.push_back(*topster.getKV(i));
        return out;
    }

    // ---------------- query time: wildcard (q="*" without a vector query), Index::search_wildcard, index.cpp:6616-6818 ----------------
    // Every filter id (every seq_id when there is no filter) minus the excluded ids gets a KV whose text-match slot is the
    // constant 100 (compute_sort_scores(..., max_field_match_score = 100, ...), :6728-6730 — no :5541 override here) and goes
    // through Topster::add; the reference splits the ids over threads and merges the per-thread Topsters (aggregate_topster),
    // which keeps exactly the global top-K. result ids = the ids processed (all_result_ids = the filter id array).
    keyword_result_t search_wildcard(const keyword_query_t& q) const {
        keyword_result_t out;
        Topster topster(q.topster_size ? q.topster_size : topster_size(q.fetch_size, q.filter_ids.size()));
        const size_t n = q.filter_ids.empty() ? num_docs : q.filter_ids.size();
        for (size_t i = 0; i < n; i++) {
            const uint32_t seq_id = q.filter_ids.empty() ? (uint32_t)i : q.filter_ids[i];
            if (!q.excluded_ids.empty() && std::binary_search(q.excluded_ids.begin(), q.excluded_ids.end(), seq_id)) continue;
            int64_t scores[3] = {0, 0, 0};
            int64_t match_score_index = -1;
            compute_sort_scores(q.sort, seq_id, 100, scores, match_score_index, 0);
            KV kv(0, seq_id, seq_id, (int8_t)match_score_index, scores);
            if (match_score_index >= 0) kv.text_match_score = scores[match_score_index];      // KV ctor, include/topster.h:38-48
            topster.add(&kv);
            out.result_ids.push_back(seq_id);
            out.num_keyword_matches++;
        }
        topster.sort();
        for (uint32_t i = 0; i < topster.size; i++) out.kvs.push_back(*topster.getKV(i));
        return out;
    }

    // ---------------- query time: pure vector (q="*"), index.cpp:3645-3732 ----------------
    // filter_ids = what filter_by matched (sorted; nullptr = no filter_by), excluded_ids = hidden / curated ids (sorted).
    // Two branches, index.cpp:3664-3670:
    //   * FLAT (`filter_by` given and its id count < vector_query.flat_search_cutoff): process_results_bruteforce, :3345-3374 — EVERY
    //     filter id that has a vector gets its exact distance and goes to the Topster: no k, and the VectorFilterFunctor (hence the
    //     excluded ids) is not consulted. Which ids survive is decided by the Topster alone (capacity :3506-3512, order
    //     include/topster.h:146-154: with the default sort [vector_distance asc, seq_id desc] equal distances keep the LARGER seq_id).
    //     result_ids (-> `found`) = every id kept by the threshold, not just the Topster's content.
    //   * otherwise process_results_hnsw_index, :3376-3445: searchKnnCloserFirst(k, VectorFilterFunctor) — restated as the EXACT k nearest
    //     (hnswlib's result heap orders (distance, internal id) pairs: at the k cut equal distances keep the SMALLER id), label order (:3389).
    keyword_result_t search_vector(const vector_query_t& vq, const std::vector<sort_by_t>& sort, size_t fetch_size,
                                   const std::vector<uint32_t>* filter_ids = nullptr, const std::vector<uint32_t>* excluded_ids = nullptr) const {
        keyword_result_t out;
        Topster topster(topster_size(fetch_size, filter_ids ? filter_ids->size() : 0));
        size_t k = vq.k == 0 ? std::max<size_t>(vq.k, fetch_size) : vq.k;                                  // :3646
        auto functor = [&](uint32_t id) {                                                                   // include/index.h:339-353
            const size_t nf = filter_ids ? filter_ids->size() : 0, ne = excluded_ids ? excluded_ids->size() : 0;
            if (nf == 0 && ne == 0) return true;
            if (ne > 0 && std::binary_search(excluded_ids->begin(), excluded_ids->end(), id)) return false;
            if (nf == 0) return true;
            return std::binary_search(filter_ids->begin(), filter_ids->end(), id);
        };
        if (vq.query_doc_given && functor(vq.seq_id)) k++;                                                  // :3651-3654
        const bool filter_by_provided = filter_ids != nullptr;
        const size_t filter_id_count = filter_ids ? filter_ids->size() : 0;
        std::vector<vec_hit_t> hits;
        if (filter_by_provided && filter_id_count < vq.flat_search_cutoff) {
            std::vector<float> q = vq.values;
            if (distance_type == cosine) { std::vector<float> n(q.size()); normalize_vector(vq.values, n); q.swap(n); }   // :3362-3364 (re-done per id there)
            std::unordered_map<uint32_t, size_t> row_of;
            for (size_t r = 0; r < vec_labels.size(); r++) row_of[vec_labels[r]] = r;
            for (uint32_t seq_id : *filter_ids) {                                                           // the filter iterator: ascending seq_ids
                auto it = row_of.find(seq_id);
                if (it == row_of.end()) continue;                                                           // getDataByLabel throws: "likely not found" :3355-3360
                hits.push_back({ip_distance(q.data(), vec_store.data() + it->second * num_dim, num_dim), seq_id});
            }
        } else {
            hits = flat_knn(vq.values, k, (filter_ids && !filter_ids->empty()) ? filter_ids : nullptr, (excluded_ids && !excluded_ids->empty()) ? excluded_ids : nullptr);
            std::sort(hits.begin(), hits.end(), [](const vec_hit_t& a, const vec_hit_t& b) { return a.seq_id < b.seq_id; });  // :3389
        }
        std::vector<uint32_t> nearest_ids;
        for (const auto& h : hits) {
            if (vq.query_doc_given && vq.seq_id == h.seq_id) continue;                                      // :3686
            float d = (distance_type == cosine) ? std::abs(h.dist) : h.dist;
            if (d > vq.distance_threshold) continue;
            int64_t scores[3] = {0, 0, 0};
            int64_t match_score_index = -1;
            compute_sort_scores(sort, h.seq_id, 0, scores, match_score_index, d);
            KV kv(0, h.seq_id, h.seq_id, (int8_t)match_score_index, scores);
            kv.vector_distance = d;
            topster.add(&kv);
            nearest_ids.push_back(h.seq_id);
        }
        std::sort(nearest_ids.begin(), nearest_ids.end());
        out.result_ids = nearest_ids;
        out.num_keyword_matches = nearest_ids.size();          // (all_result_ids_len -> `found`, :3727-3732)
        topster.sort();
        for (uint32_t i = 0; i < topster.size; i++) out.kvs.push_back(*topster.getKV(i));
        return out;
    }

    // ---------------- query time: hybrid, index.cpp:4036-4221 ----------------
    keyword_result_t search_hybrid(const keyword_query_t& q, const vector_query_t& vq) const {
        keyword_result_t out;
        Topster topster(q.topster_size ? q.topster_size : topster_size(q.fetch_size, q.filter_ids.size()));
        search_across_fields(q, &topster, out);

        const float VECTOR_SEARCH_WEIGHT = vq.alpha;
        const float TEXT_MATCH_WEIGHT = 1.0 - VECTOR_SEARCH_WEIGHT;

        size_t default_k = 100;
        size_t k = vq.k == 0 ? std::max<size_t>(q.fetch_size, default_k) : vq.k;
        const std::vector<uint32_t>* allow = q.filter_ids.empty() ? nullptr : &q.filter_ids;
        const std::vector<uint32_t>* excl = q.excluded_ids.empty() ? nullptr : &q.excluded_ids;
        std::vector<vec_hit_t> dist_results;
        {   // process_results_hnsw_index non-wildcard tail, index.cpp:3414-3437
            std::vector<vec_hit_t> pairs = flat_knn(vq.values, k, allow, excl);
            std::sort(pairs.begin(), pairs.end(), [](const vec_hit_t& a, const vec_hit_t& b) { return a.seq_id < b.seq_id; });
            for (const auto& p : pairs) {
                float s = (distance_type == cosine) ? std::abs(p.dist) : p.dist;
                if (s > vq.distance_threshold) continue;
                dist_results.push_back(p);
            }
            // reference: std::sort by distance (unstable); restated as a stable sort on the label-ordered
            // sequence — differs only for bit-equal distances
            std::stable_sort(dist_results.begin(), dist_results.end(), [](const vec_hit_t& a, const vec_hit_t& b) { return a.dist < b.dist; });
        }
        std::unordered_map<uint32_t, uint32_t> seq_id_to_rank;
        for (size_t i = 0; i < dist_results.size(); i++) seq_id_to_rank.emplace(dist_results[i].seq_id, (uint32_t)i);
        std::sort(dist_results.begin(), dist_results.end(), [](const vec_hit_t& a, const vec_hit_t& b) { return a.seq_id < b.seq_id; });

        topster.sort();
        int64_t text_rank = 0;
        int64_t last_text_match_score = INT64_MAX;
        for (uint32_t i = 0; i < topster.size; i++) {
            KV* r = topster.getKV(i);
            if (r->match_score_index < 0 || r->match_score_index > 2) continue;
            r->text_match_score = r->scores[r->match_score_index];
            if (r->text_match_score < last_text_match_score) ++text_rank;
            last_text_match_score = r->text_match_score;
            r->scores[r->match_score_index] = float_to_int64_t((1.0 / (text_rank)) * TEXT_MATCH_WEIGHT);
        }

        std::vector<uint32_t> vec_search_ids;
        for (size_t ri = 0; ri < dist_results.size(); ri++) {
            const auto& dr = dist_results[ri];
            uint32_t seq_id = dr.seq_id;
            KV* found_kv = nullptr;
            auto it = topster.map.find(seq_id);
            if (it != topster.map.end()) found_kv = it->second;
            if (found_kv) {
                if (found_kv->match_score_index < 0 || found_kv->match_score_index > 2) continue;
                found_kv->vector_distance = dr.dist;
                int64_t match_score = float_to_int64_t((int64_t_to_float(found_kv->scores[found_kv->match_score_index])) +
                                                       ((1.0 / (seq_id_to_rank[seq_id] + 1)) * VECTOR_SEARCH_WEIGHT));
                int64_t match_score_index = -1;
                int64_t scores[3] = {0, 0, 0};
                compute_sort_scores(q.sort, seq_id, match_score, scores, match_score_index, dr.dist);
                for (int i = 0; i < 3; i++) found_kv->scores[i] = scores[i];
                found_kv->match_score_index = (int8_t)match_score_index;
            } else {
                int64_t scores[3] = {0, 0, 0};
                int64_t match_score = float_to_int64_t((1.0 / (seq_id_to_rank[seq_id] + 1)) * VECTOR_SEARCH_WEIGHT);
                int64_t match_score_index = -1;
                compute_sort_scores(q.sort, seq_id, match_score, scores, match_score_index, dr.dist);
                KV kv(0, seq_id, seq_id, (int8_t)match_score_index, scores);
                kv.text_match_score = 0;
                kv.vector_distance = dr.dist;
                topster.add(&kv);
                vec_search_ids.push_back(seq_id);
            }
        }
        if (!vec_search_ids.empty()) {  // all_result_ids = or_scalar(all_result_ids, vec_search_ids), :4206-4212
            std::vector<uint32_t> merged;
            std::set_union(out.result_ids.begin(), out.result_ids.end(), vec_search_ids.begin(), vec_search_ids.end(),
                           std::back_inserter(merged));
            out.result_ids.swap(merged);
        }
        if (vq.rerank_hybrid_matches) compute_aux_scores(&topster, q, vq);          // index.cpp:4234-4236
        topster.sort();
        for (uint32_t i = 0; i < topster.size; i++) out.kvs.push_back(*topster.getKV(i));
        return out;
    }

    // compute_aux_scores, index.cpp:8793-8923: hits found by one side only get the other side's score, then every hit is re-ranked on
    // both and re-fused (topster->map is an unordered_map in the reference: both sorts below are total / stable on a total order, so
    // its iteration order does not matter)
    void compute_aux_scores(Topster* topster, const keyword_query_t& q, const vector_query_t& vq) const {
        std::vector<KV*> text_match_ids;
        for (auto& kv : topster->map) {
            if (kv.second->text_match_score == 0) {
                text_match_ids.push_back(kv.second);                                    // only found via vector distance
            } else if (kv.second->vector_distance == -1.0f) {                         // only found via text match
                const float* x = vec_get((uint32_t)kv.second->key);
                if (!x) continue;                                                       // getDataByLabel throws: likely not found
                float dist;
                if (distance_type == cosine) {
                    std::vector<float> normalized_q(vq.values.size());
                    normalize_vector(vq.values, normalized_q);
                    dist = ip_distance(normalized_q.data(), x, num_dim);
                } else {
                    dist = ip_distance(vq.values.data(), x, num_dim);
                }
                kv.second->vector_distance = dist;
            }
        }
        if (!text_match_ids.empty()) {
            std::sort(text_match_ids.begin(), text_match_ids.end(), [](const KV* a, const KV* b) { return a->key < b->key; });
            // compute_text_match_aux_score: one iterator per token, positioned on every id in ascending order
            std::vector<or_iterator_t> token_its;
            std::vector<posting_list_t*> expanded_plists;
            get_field_token_its(q, token_its, expanded_plists);
            keyword_query_t q0 = q;
            q0.total_cost = 0;                                                           // compute_aggregated_score(..., total_cost = 0, syn_orig_num_tokens = -1, ...)
            q0.syn_orig_num_tokens = -1; q0.orig_num_tokens = (int)q.tokens.size(); q0.is_synonym_query = false; q0.demote_synonym_match = false;
            for (KV* kv : text_match_ids) {
                const uint32_t seq_id = (uint32_t)kv->key;
                for (size_t i = 0; i < token_its.size(); i++) token_its[i].skip_to(seq_id);
                kv->text_match_score = compute_aggregated_score(token_its, q0, seq_id);
            }
            for (auto* p : expanded_plists) delete p;
        }
        std::vector<KV*> kvs;
        for (const auto& kv : topster->map) kvs.push_back(kv.second);
        std::unordered_map<uint64_t, int32_t> semantic_seq_id_ranks, keyword_seq_id_ranks;
        std::stable_sort(kvs.begin(), kvs.end(), [](const KV* a, const KV* b) { return std::tie(a->text_match_score, a->key) > std::tie(b->text_match_score, b->key); });
        for (size_t i = 0; i < kvs.size(); ++i) keyword_seq_id_ranks.emplace(kvs[i]->key, (int32_t)i + 1);
        std::stable_sort(kvs.begin(), kvs.end(), [](const KV* a, const KV* b) { return a->vector_distance < b->vector_distance; });
        for (size_t i = 0; i < kvs.size(); ++i) semantic_seq_id_ranks.emplace(kvs[i]->key, (int32_t)i + 1);
        for (auto& kv : topster->map) {
            const uint64_t seq_id = kv.second->key;
            if (kv.second->match_score_index < 0 || kv.second->match_score_index > 2) continue;   // (the reference indexes scores[] unguarded)
            kv.second->scores[kv.second->match_score_index] = float_to_int64_t((1.0 / keyword_seq_id_ranks[seq_id]) * (1.0 - vq.alpha) +
                                                                               (1.0 / semantic_seq_id_ranks[seq_id]) * vq.alpha);
        }
    }
};

}  // namespace oracle
