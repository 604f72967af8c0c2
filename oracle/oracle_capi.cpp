// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/for_codec.h header).
// Flat C entry points over the oracle so that tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg can drive it through ctypes. Nothing here is part of the product ABI
// (that is include/tsgpu.h).
#include <thread>
#include <atomic>
#include <chrono>
#include "oracle_index.h"
#include "facet_count.h"
#include "hnsw_graph.h"
#include <map>

using namespace oracle;

extern "C" {

struct orc_kw_query {
    const uint32_t* tokens; uint32_t n_tokens;
    const uint32_t* field_ids; const int64_t* field_weights; uint32_t n_fields;
    int32_t match_type;
    int32_t prioritize_exact_match, prioritize_token_position, prioritize_num_matching_fields;
    uint32_t total_cost;
    int32_t sort_kind[3]; int32_t sort_column[3]; int32_t sort_order[3]; uint32_t n_sort;
    uint32_t fetch_size;
    const uint32_t* excluded_ids; uint32_t n_excluded;
    const uint32_t* filter_ids; uint32_t n_filter;
    uint32_t topster_size;   // 0 = reference rule
    const uint32_t* dropped_tokens; uint32_t n_dropped;
    int32_t syn_orig_num_tokens, orig_num_tokens, is_synonym_query, demote_synonym_match;   // syn_orig_num_tokens: -1 = not a synonym pass
};

struct orc_result {
    uint32_t cap;            // in: capacity of the per-hit arrays
    uint32_t n;              // out: hits written (topster order)
    uint64_t* keys;          // [cap]
    int64_t* scores;         // [cap*3]
    int64_t* text_match;     // [cap]
    float* vector_distance;  // [cap]
    int8_t* match_score_index;  // [cap]
    uint64_t num_keyword_matches;
    uint64_t n_result_ids;   // out: total emitted ids
    uint32_t* result_ids;    // nullable, [result_ids_cap]
    uint64_t result_ids_cap;
    int32_t search_cutoff;
};

static keyword_query_t to_query(const orc_kw_query* q) {
    keyword_query_t k;
    k.tokens.assign(q->tokens, q->tokens + q->n_tokens);
    for (uint32_t i = 0; i < q->n_fields; i++) k.fields.push_back({q->field_ids[i], q->field_weights[i]});
    k.match_type = q->match_type;
    k.prioritize_exact_match = q->prioritize_exact_match != 0;
    k.prioritize_token_position = q->prioritize_token_position != 0;
    k.prioritize_num_matching_fields = q->prioritize_num_matching_fields != 0;
    k.total_cost = q->total_cost;
    for (uint32_t i = 0; i < q->n_sort; i++) k.sort.push_back({q->sort_kind[i], q->sort_column[i], q->sort_order[i]});
    k.fetch_size = q->fetch_size;
    if (q->n_excluded) k.excluded_ids.assign(q->excluded_ids, q->excluded_ids + q->n_excluded);
    if (q->n_filter) k.filter_ids.assign(q->filter_ids, q->filter_ids + q->n_filter);
    k.topster_size = q->topster_size;
    if (q->n_dropped) k.dropped_tokens.assign(q->dropped_tokens, q->dropped_tokens + q->n_dropped);
    k.syn_orig_num_tokens = q->syn_orig_num_tokens; k.orig_num_tokens = q->orig_num_tokens;
    k.is_synonym_query = q->is_synonym_query != 0; k.demote_synonym_match = q->demote_synonym_match != 0;
    return k;
}

static void fill(const keyword_result_t& r, orc_result* out) {
    uint32_t n = (uint32_t)std::min<size_t>(r.kvs.size(), out->cap);
    out->n = n;
    for (uint32_t i = 0; i < n; i++) {
        out->keys[i] = r.kvs[i].key;
        for (int j = 0; j < 3; j++) out->scores[i * 3 + j] = r.kvs[i].scores[j];
        out->text_match[i] = r.kvs[i].text_match_score;
        out->vector_distance[i] = r.kvs[i].vector_distance;
        if (out->match_score_index) out->match_score_index[i] = r.kvs[i].match_score_index;
    }
    out->num_keyword_matches = r.num_keyword_matches;
    out->n_result_ids = r.result_ids.size();
    if (out->result_ids) {
        size_t m = std::min<size_t>(r.result_ids.size(), out->result_ids_cap);
        std::copy(r.result_ids.begin(), r.result_ids.begin() + m, out->result_ids);
    }
    out->search_cutoff = r.search_cutoff ? 1 : 0;
}

void* orc_create(uint32_t n_fields, uint32_t n_columns) { return new Index(n_fields, n_columns); }
void orc_free(void* h) { delete (Index*)h; }
void orc_set_num_docs(void* h, uint32_t n) { ((Index*)h)->num_docs = n; }
uint32_t orc_num_docs(void* h) { return ((Index*)h)->num_docs; }

void orc_index_plain(void* h, uint32_t seq_id, uint32_t field, const uint32_t* toks, uint32_t n) {
    ((Index*)h)->index_plain_field(seq_id, field, std::vector<uint32_t>(toks, toks + n));
}
void orc_index_array(void* h, uint32_t seq_id, uint32_t field, const uint32_t* toks, const uint32_t* elem_lens, uint32_t n_elems) {
    std::vector<std::vector<uint32_t>> elems;
    size_t p = 0;
    for (uint32_t e = 0; e < n_elems; e++) { elems.emplace_back(toks + p, toks + p + elem_lens[e]); p += elem_lens[e]; }
    ((Index*)h)->index_array_field(seq_id, field, elems);
}
void orc_load_posting(void* h, uint32_t field, uint32_t term, const uint32_t* ids, const uint32_t* offset_index,
                      const uint32_t* offsets, uint32_t n, uint32_t n_offsets) {
    ((Index*)h)->load_posting(field, term, ids, offset_index, offsets, n, n_offsets);
}
// ---- group_by (oracle/group_topster.h) ----
struct orc_grouped {
    uint32_t group_cap, kv_cap;     // in: capacities
    uint32_t n_groups;              // out
    uint32_t* group_size;           // [group_cap] KVs of each returned group
    uint32_t* group_found;          // [group_cap] groups_processed[distinct_key]
    uint64_t* distinct_key;         // [group_cap]
    uint64_t* keys;                 // [kv_cap] group-major
    int64_t* scores;                // [kv_cap*3]
    uint64_t groups_count, groups_exact;
    uint8_t* loglog;                // nullable, [16384]
    uint32_t* missing_ids; uint64_t missing_cap, n_missing;   // group_by_missing_value_ids of a first pass (ascending)
    uint64_t num_keyword_matches, n_result_ids;
    uint32_t* result_ids; uint64_t result_ids_cap;
};
static void fill_grouped(const grouped_result_t& g, orc_grouped* out) {
    out->n_groups = (uint32_t)std::min<size_t>(g.groups.size(), out->group_cap);
    size_t at = 0;
    for (uint32_t i = 0; i < out->n_groups; i++) {
        const auto& v = g.groups[i];
        out->group_size[i] = (uint32_t)v.size();
        out->group_found[i] = g.group_found[i];
        out->distinct_key[i] = v.empty() ? 0 : v[0].distinct_key;
        for (const KV& kv : v) {
            if (at >= out->kv_cap) break;
            out->keys[at] = kv.key;
            for (int j = 0; j < 3; j++) out->scores[at * 3 + j] = kv.scores[j];
            at++;
        }
    }
    out->groups_count = g.groups_count; out->groups_exact = g.groups_exact;
    if (out->loglog) memcpy(out->loglog, g.loglog.data(), g.loglog.size());
}
int32_t orc_search_keyword_grouped(void* h, const orc_kw_query* q, const uint64_t* distinct_ids, const uint8_t* has_value, uint32_t n_distinct,
                                   int32_t group_missing_values, uint32_t group_limit, int32_t first_pass, orc_grouped* out) {
    std::vector<uint64_t> d(distinct_ids, distinct_ids + n_distinct);
    std::vector<uint8_t> hv;
    if (has_value) hv.assign(has_value, has_value + n_distinct);
    grouped_result_t g;
    std::vector<uint32_t> missing;
    const keyword_result_t r = ((Index*)h)->search_keyword_grouped(to_query(q), d, hv, group_missing_values != 0, group_limit, first_pass != 0, g, &missing);
    fill_grouped(g, out);
    out->n_missing = missing.size();
    if (out->missing_ids) std::copy(missing.begin(), missing.begin() + std::min<size_t>(missing.size(), out->missing_cap), out->missing_ids);
    out->num_keyword_matches = r.num_keyword_matches;
    out->n_result_ids = r.result_ids.size();
    if (out->result_ids) std::copy(r.result_ids.begin(), r.result_ids.begin() + std::min<size_t>(r.result_ids.size(), out->result_ids_cap), out->result_ids);
    return 0;
}
int32_t orc_search_candidates_grouped(void* h, const orc_kw_query* combos, uint32_t n_combos, const uint64_t* distinct_ids, const uint8_t* has_value, uint32_t n_distinct,
                                      int32_t group_missing_values, uint32_t group_limit, int32_t first_pass, orc_grouped* out, uint16_t* query_index_out) {
    std::vector<keyword_query_t> qs;
    for (uint32_t i = 0; i < n_combos; i++) qs.push_back(to_query(combos + i));
    std::vector<uint64_t> d(distinct_ids, distinct_ids + n_distinct);
    std::vector<uint8_t> hv;
    if (has_value) hv.assign(has_value, has_value + n_distinct);
    grouped_result_t g;
    const keyword_result_t r = ((Index*)h)->search_candidates_grouped(qs, d, hv, group_missing_values != 0, group_limit, first_pass != 0, g);
    fill_grouped(g, out);
    if (query_index_out) {                       // per KV, in the order of out->keys
        size_t at = 0;
        for (uint32_t i = 0; i < out->n_groups; i++) for (const KV& kv : g.groups[i]) { if (at < out->kv_cap) query_index_out[at] = kv.query_index; at++; }
    }
    out->n_missing = 0;
    out->num_keyword_matches = r.num_keyword_matches;
    out->n_result_ids = r.result_ids.size();
    if (out->result_ids) std::copy(r.result_ids.begin(), r.result_ids.begin() + std::min<size_t>(r.result_ids.size(), out->result_ids_cap), out->result_ids);
    return 0;
}
// the distinct Topster alone, fed a sequence of KVs (key, distinct_key, scores[3]) in order: what add() returned per KV + the collector's content
int32_t orc_group_topster_run(uint32_t capacity, uint32_t distinct, int32_t first_pass, uint32_t n, const uint64_t* keys, const uint64_t* dkeys,
                              const int64_t* scores, int32_t* ret_out, orc_grouped* out) {
    GroupTopster t(capacity, distinct, first_pass != 0);
    std::unordered_map<uint64_t, uint32_t> processed;
    std::unordered_set<uint64_t> seen;
    for (uint32_t i = 0; i < n; i++) {
        KV kv(0, keys[i], dkeys[i], 0, scores + (size_t)i * 3);
        const int r = t.add(&kv);
        if (ret_out) ret_out[i] = r;
        if (r < 2) processed[dkeys[i]]++;
        seen.insert(dkeys[i]);
    }
    grouped_result_t g;
    populate_grouped(t, processed, g);
    g.groups_exact = seen.size();
    fill_grouped(g, out);
    return 0;
}
uint64_t orc_hash_wy(const void* key, uint64_t len) { return hash_wy(key, len); }
uint64_t orc_hash_combine(uint64_t a, uint64_t b) { return hash_combine(a, b); }
uint64_t orc_loglog_of_keys(const uint64_t* dkeys, uint64_t n, uint8_t* registers_out) {
    LogLogBeta c;
    for (uint64_t i = 0; i < n; i++) c.add(std::to_string(dkeys[i]));
    if (registers_out) memcpy(registers_out, c.registers_.data(), c.registers_.size());
    return c.cardinality();
}
uint64_t orc_loglog_cardinality(const uint8_t* registers) { return LogLogBeta::cardinality_of(registers); }
// Index::get_distinct_id over n_fields facet hash indexes in CSR form (field f: doc d owns hashes[f][ptr[f][d] .. ptr[f][d+1]))
void orc_distinct_ids(uint32_t n_docs, uint32_t n_fields, const uint64_t* const* ptr, const uint32_t* const* hashes, int32_t group_missing_values,
                      uint64_t* distinct_out, uint8_t* has_value_out) {
    std::vector<std::vector<uint32_t>> hs(n_fields);
    for (uint32_t d = 0; d < n_docs; d++) {
        for (uint32_t f = 0; f < n_fields; f++) hs[f].assign(hashes[f] + ptr[f][d], hashes[f] + ptr[f][d + 1]);
        bool missing = false;
        distinct_out[d] = distinct_id_of(d, hs, group_missing_values != 0, &missing);
        if (has_value_out) has_value_out[d] = missing ? 0 : 1;
    }
}

// decoded dump of one posting list (for building the GPU index from oracle-built postings in tests)
// returns n ids; when buffers are null only sizes are reported
uint32_t orc_dump_posting(void* h, uint32_t field, uint32_t term, uint32_t* ids, uint32_t* offset_index, uint32_t* offsets,
                          uint32_t* n_offsets_out) {
    Index* idx = (Index*)h;
    auto it = idx->fields[field].terms.find(term);
    if (it == idx->fields[field].terms.end()) { if (n_offsets_out) *n_offsets_out = 0; return 0; }
    posting_list_t* pl = it->second->full;
    std::unique_ptr<posting_list_t> tmp;
    if (!pl) { tmp.reset(it->second->compact->to_full_posting_list((uint16_t)MAX_BLOCK_ELEMENTS)); pl = tmp.get(); }
    uint32_t n = 0, no = 0;
    auto iter = pl->new_iterator();
    while (iter.valid()) {
        auto* blk = iter.block();
        uint32_t ci = iter.index();
        uint32_t s = iter.offset_index[ci];
        uint32_t e = (ci == blk->size() - 1) ? blk->offsets.getLength() : iter.offset_index[ci + 1];
        if (ids) { ids[n] = iter.id(); offset_index[n] = no; for (uint32_t j = s; j < e; j++) offsets[no + (j - s)] = iter.offsets[j]; }
        no += e - s;
        n++;
        iter.next();
    }
    if (n_offsets_out) *n_offsets_out = no;
    return n;
}
uint32_t orc_list_terms(void* h, uint32_t field, uint32_t* terms, uint32_t cap) {
    Index* idx = (Index*)h;
    uint32_t n = 0;
    for (auto& kv : idx->fields[field].terms) { if (terms && n < cap) terms[n] = kv.first; n++; }
    return n;
}
int32_t orc_field_is_array(void* h, uint32_t field) { return ((Index*)h)->fields[field].is_array ? 1 : 0; }

void orc_set_sort(void* h, uint32_t column, uint32_t seq_id, int64_t v) { ((Index*)h)->set_sort_value(column, seq_id, v); }
void orc_set_sort_dense(void* h, uint32_t column, const int64_t* vals, uint32_t n) {
    auto& m = ((Index*)h)->sort_index[column];
    m.reserve(n);
    for (uint32_t i = 0; i < n; i++) m[i] = vals[i];
}

void orc_vec_init(void* h, uint32_t dim, int32_t metric) { ((Index*)h)->vec_init(dim, metric); }
void orc_vec_add(void* h, const uint32_t* labels, const float* data, uint32_t n) {
    Index* idx = (Index*)h;
    idx->vec_store.reserve(idx->vec_store.size() + (size_t)n * idx->num_dim);
    for (uint32_t i = 0; i < n; i++) idx->vec_add(labels[i], data + (size_t)i * idx->num_dim);
}
int32_t orc_vec_get(void* h, uint32_t label, float* out) {
    Index* idx = (Index*)h;
    const float* p = idx->vec_get(label);
    if (!p) return -1;
    std::copy(p, p + idx->num_dim, out);
    return 0;
}
float orc_ip_distance(const float* a, const float* b, uint32_t dim) { return Index::ip_distance(a, b, dim); }
// summation order of the distance function = the SIMD level hnswlib was compiled for: 4 (SSE, stock reference build; default), 8 (AVX), 16 (AVX-512)
int32_t orc_set_ip_lanes(int32_t lanes) { if (lanes != 4 && lanes != 8 && lanes != 16) return -1; Index::ip_lanes() = lanes; return 0; }
int32_t orc_get_ip_lanes() { return Index::ip_lanes(); }

// exact k nearest, closest first; returns hits written
uint32_t orc_flat_knn(void* h, const float* q, uint32_t k, const uint32_t* allow_ids, uint32_t n_allow,
                      float* dist_out, uint32_t* label_out) {
    Index* idx = (Index*)h;
    std::vector<float> qv(q, q + idx->num_dim);
    std::vector<uint32_t> allow;
    if (n_allow) allow.assign(allow_ids, allow_ids + n_allow);
    auto hits = idx->flat_knn(qv, k, n_allow ? &allow : nullptr);
    for (size_t i = 0; i < hits.size(); i++) { dist_out[i] = hits[i].dist; label_out[i] = hits[i].seq_id; }
    return (uint32_t)hits.size();
}

// ---- HNSW (hnsw_graph.h): graph over the index's vectors in row (= insertion) order; one graph per oracle index ----
static std::map<void*, hnsw_graph_t*>& hnsw_of() { static std::map<void*, hnsw_graph_t*> m; return m; }
static float hnsw_dist(const float* a, const float* b, size_t dim) { return Index::ip_distance(a, b, dim); }

void orc_hnsw_build(void* h, uint32_t M, uint32_t ef_construction, uint32_t seed) {
    Index* idx = (Index*)h;
    auto& slot = hnsw_of()[h];
    delete slot;
    slot = new hnsw_graph_t;
    slot->init(idx->num_dim, M, ef_construction, seed, hnsw_dist);
    for (size_t r = 0; r < idx->vec_labels.size(); r++) slot->addPoint(idx->vec_store.data() + r * idx->num_dim, idx->vec_labels[r]);
}
// the batched bulk build the library runs on the device (tsgpu_vec_hnsw_build), restated: hnsw_graph_t::bulk_build
void orc_hnsw_bulk_build(void* h, uint32_t M, uint32_t ef_construction, uint32_t seed, uint32_t seed_min, uint32_t max_batch) {
    Index* idx = (Index*)h;
    auto& slot = hnsw_of()[h];
    delete slot;
    slot = new hnsw_graph_t;
    slot->init(idx->num_dim, M, ef_construction, seed, hnsw_dist);
    std::vector<uint64_t> lab(idx->vec_labels.begin(), idx->vec_labels.end());
    slot->bulk_build(idx->vec_store.data(), lab.data(), lab.size(), seed_min ? seed_min : 1024, max_batch ? max_batch : 65536);
}
// incremental addPoint after orc_hnsw_build: the rows that orc_vec_add appended since (row order = insertion order)
void orc_hnsw_add_new_rows(void* h) {
    Index* idx = (Index*)h;
    hnsw_graph_t* g = hnsw_of()[h];
    for (size_t r = g->size(); r < idx->vec_labels.size(); r++) g->addPoint(idx->vec_store.data() + r * idx->num_dim, idx->vec_labels[r]);
}
void orc_hnsw_free(void* h) { auto it = hnsw_of().find(h); if (it != hnsw_of().end()) { delete it->second; hnsw_of().erase(it); } }
int32_t orc_hnsw_mark_deleted(void* h, uint32_t label) {
    hnsw_graph_t* g = hnsw_of()[h];
    return g->markDelete(label) ? 0 : -1;
}
// addPoint(vec, label, replace_deleted = true) for a label whose vector orc_vec_add has just stored / overwritten: a live label is updated in place, a
// vacant slot re-used, else a new element appended (hnsw_graph_t::addPointReplace). Returns the internal id, -1 when the label has no vector.
int32_t orc_hnsw_add_point_replace(void* h, uint32_t label) {
    Index* idx = (Index*)h;
    hnsw_graph_t* g = hnsw_of()[h];
    const float* v = idx->vec_get(label);
    if (!g || !v) return -1;
    return (int32_t)g->addPointReplace(v, label);
}
// labels[n] of the graph's internal ids (slot re-use moves labels between rows)
void orc_hnsw_labels(void* h, uint64_t* out) {
    hnsw_graph_t* g = hnsw_of()[h];
    for (size_t i = 0; i < g->labels.size(); i++) out[i] = g->labels[i];
}
// graph in the flat form the GPU mirror takes: levels[n]; link0[n][1 + 2M] = (count, ids..); upper lists of node i (level >= 1,
// ascending) at upper_links[(upper_ptr[i] + level - 1) * (1 + M)] = (count, ids..). Call with NULL arrays to size: returns the
// number of upper lists. info = {n, maxlevel, enterpoint, M}
uint64_t orc_hnsw_export(void* h, int32_t* info, uint32_t* levels, uint32_t* link0, uint64_t* upper_ptr, uint32_t* upper_links) {
    hnsw_graph_t* g = hnsw_of()[h];
    const size_t n = g->size(), M = g->M, S0 = 1 + 2 * M, SU = 1 + M;
    if (info) { info[0] = (int32_t)n; info[1] = g->maxlevel; info[2] = (int32_t)g->enterpoint; info[3] = (int32_t)M; }
    uint64_t n_upper = 0;
    for (size_t i = 0; i < n; i++) {
        if (upper_ptr) upper_ptr[i] = n_upper;
        if (levels) levels[i] = (uint32_t)g->levels[i];
        if (link0) {
            link0[i * S0] = (uint32_t)g->link0[i].size();
            for (size_t j = 0; j < g->link0[i].size(); j++) link0[i * S0 + 1 + j] = g->link0[i][j];
        }
        for (int l = 1; l <= g->levels[i]; l++) {
            if (upper_links) {
                const auto& nb = g->linkU[i][l - 1];
                upper_links[n_upper * SU] = (uint32_t)nb.size();
                for (size_t j = 0; j < nb.size(); j++) upper_links[n_upper * SU + 1 + j] = nb[j];
            }
            n_upper++;
        }
    }
    if (upper_ptr) upper_ptr[n] = n_upper;
    return n_upper;
}
// The inverse of orc_hnsw_export: adopt a graph given in the flat mirror form over the rows already added with orc_vec_add (row i =
// internal id i). Used by the bench / tests to run the oracle's traversal on graphs that were not built by orc_hnsw_build.
int32_t orc_hnsw_import(void* h, uint32_t M, int32_t maxlevel, uint32_t enterpoint, const uint32_t* link0, const uint64_t* upper_ptr, const uint32_t* upper_links) {
    Index* idx = (Index*)h;
    auto& slot = hnsw_of()[h];
    delete slot;
    slot = new hnsw_graph_t;
    hnsw_graph_t* g = slot;
    g->init(idx->num_dim, M, 200, 100, hnsw_dist);
    const size_t n = idx->vec_labels.size(), S0 = 1 + 2 * (size_t)M, SU = 1 + (size_t)M;
    g->data = idx->vec_store;
    g->labels.assign(idx->vec_labels.begin(), idx->vec_labels.end());
    for (size_t i = 0; i < g->labels.size(); i++) g->label_lookup[g->labels[i]] = (hnsw_graph_t::tableint)i;
    g->deleted.assign(n, 0);
    g->levels.assign(n, 0);
    g->link0.resize(n);
    g->linkU.resize(n);
    for (size_t i = 0; i < n; i++) {
        const uint32_t* l0 = link0 + i * S0;
        if (l0[0] > 2 * M) return -1;
        g->link0[i].assign(l0 + 1, l0 + 1 + l0[0]);
        const uint64_t nl = upper_ptr[i + 1] - upper_ptr[i];
        g->levels[i] = (int)nl;
        g->linkU[i].resize(nl);
        for (uint64_t l = 0; l < nl; l++) {
            const uint32_t* lu = upper_links + (upper_ptr[i] + l) * SU;
            if (lu[0] > M) return -1;
            g->linkU[i][l].assign(lu + 1, lu + 1 + lu[0]);
        }
    }
    g->maxlevel = maxlevel;
    g->enterpoint = enterpoint;
    return 0;
}
// searchKnnCloserFirst for a batch of queries on `threads` host threads (pooled visited tags per thread, like hnswlib's
// VisitedListPool): the CPU baseline of the HNSW bench leg. dist_out / label_out: [n_q][k]; n_out: [n_q]
void orc_hnsw_search_batch(void* h, const float* Q, uint32_t n_q, uint32_t k, uint32_t ef, int32_t functor_present, uint32_t threads,
                           float* dist_out, uint64_t* label_out, uint32_t* n_out) {
    Index* idx = (Index*)h;
    hnsw_graph_t* g = hnsw_of()[h];
    std::vector<uint8_t> allow;
    if (functor_present) allow.assign(g->size(), 1);
    int has_del = 0;
    for (uint8_t d : g->deleted) if (d) { has_del = 1; break; }
    std::atomic<uint32_t> next{0};
    auto work = [&]() {
        hnsw_graph_t::visited_tags_t tags;
        std::vector<float> qv(idx->num_dim), nrm(idx->num_dim);
        for (;;) {
            const uint32_t i = next.fetch_add(1);
            if (i >= n_q) break;
            qv.assign(Q + (size_t)i * idx->num_dim, Q + (size_t)(i + 1) * idx->num_dim);
            if (idx->distance_type == cosine) { Index::normalize_vector(qv, nrm); qv.swap(nrm); }
            auto res = g->searchKnnCloserFirst(qv.data(), k, ef, allow.empty() ? nullptr : allow.data(), nullptr, &tags, has_del);
            n_out[i] = (uint32_t)res.size();
            for (size_t j = 0; j < res.size(); j++) { dist_out[(size_t)i * k + j] = res[j].first; label_out[(size_t)i * k + j] = res[j].second; }
        }
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < std::max<uint32_t>(threads, 1); t++) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
}
// searchKnnCloserFirst(q, k, ef, filter): allow_ids = sorted label whitelist (VectorFilterFunctor) or NULL
uint32_t orc_hnsw_search(void* h, const float* q, uint32_t k, uint32_t ef, int32_t functor_present, const uint32_t* allow_ids, uint32_t n_allow,
                         float* dist_out, uint64_t* label_out, uint64_t* n_dist) {
    Index* idx = (Index*)h;
    hnsw_graph_t* g = hnsw_of()[h];
    std::vector<float> qv(q, q + idx->num_dim);
    if (idx->distance_type == cosine) { std::vector<float> nrm(qv.size()); Index::normalize_vector(qv, nrm); qv.swap(nrm); }
    std::vector<uint8_t> allow;                      // Typesense always passes a VectorFilterFunctor (src/index.cpp:3379-3386): it may allow everything
    if (allow_ids || functor_present) {
        allow.assign(g->size(), allow_ids ? 0 : 1);
        if (allow_ids) for (size_t i = 0; i < g->size(); i++) allow[i] = std::binary_search(allow_ids, allow_ids + n_allow, (uint32_t)g->labels[i]) ? 1 : 0;
    }
    uint64_t nd = 0;
    auto res = g->searchKnnCloserFirst(qv.data(), k, ef, allow.empty() ? nullptr : allow.data(), &nd);
    for (size_t i = 0; i < res.size(); i++) { dist_out[i] = res[i].first; label_out[i] = res[i].second; }
    if (n_dist) *n_dist = nd;
    return (uint32_t)res.size();
}

int32_t orc_search_keyword(void* h, const orc_kw_query* q, orc_result* out) {
    fill(((Index*)h)->search_keyword(to_query(q)), out);
    return 0;
}

// candidate-token combinations of one user query (Index::search_all_candidates); query_index_out: [out->cap], nullable
int32_t orc_search_candidates(void* h, const orc_kw_query* combos, uint32_t n_combos, orc_result* out, uint16_t* query_index_out) {
    std::vector<keyword_query_t> qs;
    for (uint32_t i = 0; i < n_combos; i++) qs.push_back(to_query(combos + i));
    keyword_result_t r = ((Index*)h)->search_candidates(qs);
    fill(r, out);
    if (query_index_out) for (uint32_t i = 0; i < out->n; i++) query_index_out[i] = r.kvs[i].query_index;
    return 0;
}

int32_t orc_search_wildcard(void* h, const orc_kw_query* q, orc_result* out) {
    fill(((Index*)h)->search_wildcard(to_query(q)), out);
    return 0;
}

// the vector branch of Index::search with both of its sub-branches (flat / k-cut), excluded ids and `vec:([], id: X)`
int32_t orc_search_vector2(void* h, const float* qvec, uint32_t k, float distance_threshold, const int32_t* sort_kind,
                           const int32_t* sort_column, const int32_t* sort_order, uint32_t n_sort, uint32_t fetch_size,
                           const uint32_t* filter_ids, uint32_t n_filter, int32_t filter_by_provided, const uint32_t* excluded_ids, uint32_t n_excluded,
                           uint64_t flat_search_cutoff, int32_t query_doc_given, uint32_t query_seq_id, orc_result* out) {
    Index* idx = (Index*)h;
    vector_query_t vq;
    vq.values.assign(qvec, qvec + idx->num_dim);
    vq.k = k;
    vq.distance_threshold = distance_threshold;
    vq.flat_search_cutoff = flat_search_cutoff;
    vq.query_doc_given = query_doc_given != 0;
    vq.seq_id = query_seq_id;
    std::vector<sort_by_t> sort;
    for (uint32_t i = 0; i < n_sort; i++) sort.push_back({sort_kind[i], sort_column[i], sort_order[i]});
    std::vector<uint32_t> filt, excl;
    if (n_filter) filt.assign(filter_ids, filter_ids + n_filter);
    if (n_excluded) excl.assign(excluded_ids, excluded_ids + n_excluded);
    fill(idx->search_vector(vq, sort, fetch_size, filter_by_provided ? &filt : nullptr, n_excluded ? &excl : nullptr), out);
    return 0;
}

int32_t orc_search_vector(void* h, const float* qvec, uint32_t k, float distance_threshold, const int32_t* sort_kind,
                          const int32_t* sort_column, const int32_t* sort_order, uint32_t n_sort, uint32_t fetch_size,
                          const uint32_t* filter_ids, uint32_t n_filter, orc_result* out) {
    return orc_search_vector2(h, qvec, k, distance_threshold, sort_kind, sort_column, sort_order, n_sort, fetch_size, filter_ids, n_filter, n_filter ? 1 : 0,
                              nullptr, 0, 0, 0, 0, out);
}

int32_t orc_search_hybrid_rerank(void* h, const orc_kw_query* q, const float* qvec, uint32_t k, float alpha,
                                 float distance_threshold, int32_t rerank_hybrid_matches, orc_result* out) {
    Index* idx = (Index*)h;
    vector_query_t vq;
    vq.values.assign(qvec, qvec + idx->num_dim);
    vq.k = k;
    vq.alpha = alpha;
    vq.distance_threshold = distance_threshold;
    vq.rerank_hybrid_matches = rerank_hybrid_matches != 0;
    fill(idx->search_hybrid(to_query(q), vq), out);
    return 0;
}
int32_t orc_search_hybrid(void* h, const orc_kw_query* q, const float* qvec, uint32_t k, float alpha,
                          float distance_threshold, orc_result* out) {
    Index* idx = (Index*)h;
    vector_query_t vq;
    vq.values.assign(qvec, qvec + idx->num_dim);
    vq.k = k;
    vq.alpha = alpha;
    vq.distance_threshold = distance_threshold;
    fill(idx->search_hybrid(to_query(q), vq), out);
    return 0;
}

// ---- facet counting over result ids (do_facets, hash-index branch) ----
static std::map<std::pair<void*, uint32_t>, oracle::FacetHashIndex>& facets_of() { static std::map<std::pair<void*, uint32_t>, oracle::FacetHashIndex> m; return m; }
void orc_facet_set(void* h, uint32_t field, const uint64_t* doc_ptr, const uint32_t* hashes, uint32_t n_docs) {
    oracle::FacetHashIndex& f = facets_of()[{h, field}];
    f.docs.clear();
    for (uint32_t d = 0; d < n_docs; d++)
        if (doc_ptr[d + 1] > doc_ptr[d]) f.docs[d].assign(hashes + doc_ptr[d], hashes + doc_ptr[d + 1]);
}
uint32_t orc_facet_count(void* h, uint32_t field, const uint32_t* ids, uint64_t n_ids, uint32_t sample_mod, const uint32_t* allowed, uint32_t n_allowed,
                         uint32_t* out_hash, uint32_t* out_count, uint32_t* out_doc, uint32_t* out_pos, uint32_t cap) {
    const oracle::FacetHashIndex& f = facets_of()[{h, field}];
    std::set<uint32_t> fq(allowed, allowed + n_allowed);
    const auto m = f.count(ids, n_ids, sample_mod, allowed ? &fq : nullptr);
    uint32_t i = 0;
    for (const auto& kv : m) {
        if (i < cap) { out_hash[i] = kv.first; out_count[i] = kv.second.count; out_doc[i] = kv.second.doc_id; out_pos[i] = kv.second.array_pos; }
        i++;
    }
    return i;
}

// the grouped / range forms of the walk (FacetHashIndex::count_ex). n_ranges != 0: keys are the ranges' upper bounds; doc_vals = the sort index as a dense column
// (beyond n_doc_vals: INT64_MAX). distinct_ids != null: group_limit != 0, the per-document distinct id (beyond n_distinct: 1 with group_missing_values, else seq_id).
uint32_t orc_facet_count_ex(void* h, uint32_t field, const uint32_t* ids, uint64_t n_ids, uint32_t sample_mod, const uint32_t* allowed, uint32_t n_allowed,
                            const int64_t* range_upper, const int64_t* range_lower, uint32_t n_ranges, const int64_t* doc_vals, uint64_t n_doc_vals,
                            const uint64_t* distinct_ids, uint64_t n_distinct, int32_t group_missing_values,
                            uint64_t* out_key, uint32_t* out_count, uint32_t* out_doc, uint32_t* out_pos, uint32_t cap) {
    const oracle::FacetHashIndex& f = facets_of()[{h, field}];
    std::set<uint32_t> fq(allowed, allowed + n_allowed);
    std::map<int64_t, int64_t> ranges;
    for (uint32_t r = 0; r < n_ranges; r++) ranges[range_upper[r]] = range_lower[r];
    const auto m = f.count_ex(ids, n_ids, sample_mod, allowed ? &fq : nullptr, n_ranges ? &ranges : nullptr,
                              [&](uint32_t d) -> int64_t { return d < n_doc_vals ? doc_vals[d] : INT64_MAX; }, distinct_ids != nullptr,
                              [&](uint32_t d) -> uint64_t { return d < n_distinct ? distinct_ids[d] : (group_missing_values ? 1ull : (uint64_t)d); });
    uint32_t i = 0;
    for (const auto& kv : m) {
        if (i < cap) { out_key[i] = kv.first; out_count[i] = kv.second.count; out_doc[i] = kv.second.doc_id; out_pos[i] = kv.second.array_pos; }
        i++;
    }
    return i;
}

// stats of the hash-index walk; out = {fvmin, fvmax, fvsum, fvcount}
void orc_facet_stats(void* h, uint32_t field, const uint32_t* ids, uint64_t n_ids, uint32_t sample_mod, int32_t value_type, const uint32_t* map_hash, const int64_t* map_val,
                     uint32_t n_map, double* out) {
    const oracle::FacetHashIndex& f = facets_of()[{h, field}];
    std::map<uint32_t, int64_t> m;
    for (uint32_t i = 0; i < n_map; i++) m[map_hash[i]] = map_val[i];
    const auto st = f.stats(ids, n_ids, sample_mod, value_type, &m);
    out[0] = st.fvmin; out[1] = st.fvmax; out[2] = st.fvsum; out[3] = st.fvcount;
}
// ---- value-index branch ----
static std::map<std::pair<void*, uint32_t>, oracle::FacetValueIndex>& facet_values_of() { static std::map<std::pair<void*, uint32_t>, oracle::FacetValueIndex> m; return m; }
void orc_facet_value_set(void* h, uint32_t field, const uint64_t* value_ptr, const uint32_t* seq_ids, const uint32_t* total, uint32_t n_values) {
    oracle::FacetValueIndex& f = facet_values_of()[{h, field}];
    f.ids.assign(n_values, {}); f.total.assign(total, total + n_values);
    for (uint32_t v = 0; v < n_values; v++) f.ids[v].assign(seq_ids + value_ptr[v], seq_ids + value_ptr[v + 1]);
}
uint32_t orc_facet_value_count(void* h, uint32_t field, const uint32_t* ids, uint64_t n_ids, uint32_t max_facets, int32_t wildcard_no_filter, int32_t estimate, uint32_t interval,
                               const uint32_t* order, uint32_t* out_value, uint32_t* out_count, uint32_t* out_doc, uint32_t cap) {
    const oracle::FacetValueIndex& f = facet_values_of()[{h, field}];
    std::vector<uint32_t> ord;
    if (order) ord.assign(order, order + f.ids.size());
    const auto found = f.intersect(ids, n_ids, max_facets, wildcard_no_filter != 0, estimate != 0, interval, order ? &ord : nullptr);
    for (size_t i = 0; i < found.size() && i < cap; i++) { out_value[i] = found[i].value; out_count[i] = found[i].count; out_doc[i] = found[i].doc_id; }
    return (uint32_t)found.size();
}

// ---- CPU baseline drivers: one query per thread, like the reference server (thread-per-request) ----
// tokens: [nq][n_tokens]; per_query_us: [nq] (nullable). Returns wall seconds for the whole batch.
double orc_bench_keyword(void* h, const orc_kw_query* base, const uint32_t* tokens, uint32_t nq, uint32_t n_threads,
                         double* per_query_us, uint64_t* checksum) {
    Index* idx = (Index*)h;
    std::atomic<uint32_t> next(0);
    std::atomic<uint64_t> sum(0);
    auto t0 = std::chrono::steady_clock::now();
    auto work = [&]() {
        for (;;) {
            uint32_t i = next.fetch_add(1);
            if (i >= nq) break;
            orc_kw_query q = *base;
            q.tokens = tokens + (size_t)i * base->n_tokens;
            auto q0 = std::chrono::steady_clock::now();
            keyword_result_t r = idx->search_keyword(to_query(&q));
            auto q1 = std::chrono::steady_clock::now();
            if (per_query_us) per_query_us[i] = std::chrono::duration<double, std::micro>(q1 - q0).count();
            uint64_t c = r.num_keyword_matches;
            for (auto& kv : r.kvs) c = c * 1315423911ull + kv.key;
            sum.fetch_add(c);
        }
    };
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < n_threads; t++) th.emplace_back(work);
    for (auto& t : th) t.join();
    auto t1 = std::chrono::steady_clock::now();
    if (checksum) *checksum = sum.load();
    return std::chrono::duration<double>(t1 - t0).count();
}

// queries: [nq][dim]; exact flat scan per query, one query per thread
double orc_bench_vector(void* h, const float* queries, uint32_t nq, uint32_t k, uint32_t n_threads, double* per_query_us) {
    Index* idx = (Index*)h;
    std::atomic<uint32_t> next(0);
    auto t0 = std::chrono::steady_clock::now();
    auto work = [&]() {
        for (;;) {
            uint32_t i = next.fetch_add(1);
            if (i >= nq) break;
            std::vector<float> qv(queries + (size_t)i * idx->num_dim, queries + (size_t)(i + 1) * idx->num_dim);
            auto q0 = std::chrono::steady_clock::now();
            auto hits = idx->flat_knn(qv, k);
            auto q1 = std::chrono::steady_clock::now();
            if (per_query_us) per_query_us[i] = std::chrono::duration<double, std::micro>(q1 - q0).count();
            (void)hits;
        }
    };
    std::vector<std::thread> th;
    for (uint32_t t = 0; t < n_threads; t++) th.emplace_back(work);
    for (auto& t : th) t.join();
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

// ---- direct hooks for unit tests of the restated Match (vs oracle/_ref) ----
void orc_match(const uint16_t* positions, const uint32_t* lens, const uint8_t* last, uint32_t n_tokens,
               int32_t check_exact, uint8_t* out) {
    std::vector<token_positions_t> tp(n_tokens);
    size_t p = 0;
    for (uint32_t t = 0; t < n_tokens; t++) {
        tp[t].last_token = last[t] != 0;
        tp[t].positions.assign(positions + p, positions + p + lens[t]);
        p += lens[t];
    }
    Match m(0, tp, false, check_exact != 0);
    out[0] = m.words_present; out[1] = m.distance; out[2] = m.max_offset; out[3] = m.exact_match;
}
uint64_t orc_match_score(uint8_t words_present, uint8_t distance, uint8_t max_offset, uint8_t exact_match,
                         uint32_t total_cost, uint32_t unique_words, uint8_t synonym_score) {
    Match m(words_present, distance, max_offset, exact_match);
    return m.get_match_score(total_cost, unique_words, synonym_score);
}
int64_t orc_float_to_int64(float f) { return float_to_int64_t(f); }
float orc_int64_to_float(int64_t v) { return int64_t_to_float(v); }

}  // extern "C"
