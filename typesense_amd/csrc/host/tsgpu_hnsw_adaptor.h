// tsgpu_hnsw_adaptor.h — header-only C++ adaptor with the method names Typesense calls on
// hnswlib::HierarchicalNSW<float> / hnswlib::InnerProductSpace, implemented over the C-ABI (include/tsgpu.h).
// With it, `struct hnsw_index_t` (reference include/index.h:356-370) changes by two typedefs:
//
//     tsgpu::InnerProductSpace*        space;    // was hnswlib::InnerProductSpace*
//     tsgpu::HierarchicalNSW<float>*   vecdex;   // was hnswlib::HierarchicalNSW<float>*
//
// Call sites that keep compiling unchanged (reference file:line):
//     new HierarchicalNSW<float>(space, init, M, ef_c, 100, true)            include/index.h:367
//     vecdex->getCurrentElementCount(), getMaxElements(), resizeIndex(n)     src/index.cpp:1004-1006
//     vecdex->addPoint(vec.data(), (size_t)seq_id, true)                     src/index.cpp:1052-1054
//     vecdex->markDelete(seq_id)                                             src/index.cpp:7423
//     vecdex->getDataByLabel<float>(seq_id)   (throws when missing)          src/index.cpp:3355-3359, 5840, 8860
//     vecdex->searchKnnCloserFirst(q, k, ef, &filterFunctor)                 src/index.cpp:3384-3386
//     space->get_dist_func()(a, b, &dim)                                     src/index.cpp:3365
// Differences, all deliberate: by default this adaptor's search is EXACT (ef, M, ef_construction are accepted and ignored); with
// graph_threads > 0 the library builds hnswlib's graph itself from the addPoint calls and searches it (tsgpu_hnsw_build.h); a graph built
// by a real hnswlib can be mirrored with mirror_hnsw_graph() at the end of this file. cosine normalisation stays where the reference does it (caller side, src/index.cpp:1049-1052, 3381-3384),
// and the filter functor (a pure predicate over seq_ids) is never ignored (filter_by, hidden and excluded hits depend on it,
// include/index.h:325-354): at the reference's unmodified call site it is applied to the results of an over-fetching exact search
// (2k labels asked for a functor that rejects little), with candidate ids (the filter's id array) it becomes an allow-list, and only a
// selective functor WITHOUT candidate ids is swept over every live label (searchKnnCloserFirst below).
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <algorithm>
#include <unordered_set>
#include <vector>
#include "../../../include/tsgpu.h"

namespace tsgpu {

typedef size_t labeltype;

class BaseFilterFunctor {
public:
    virtual bool operator()(labeltype) { return true; }
    virtual ~BaseFilterFunctor() {}
};

typedef float (*DISTFUNC)(const void*, const void*, const void*);

// hnswlib's InnerProductSpace::get_dist_func contract for ONE pair on the host (by-id paths); the batched
// scans go through tsgpu_vec_knn_batch / tsgpu_vec_distances on the GPU. TSGPU_IP_LANES = the SIMD level the server's own
// hnswlib would have been compiled for (4 = SSE: the stock flags; 8 = -mavx; 16 = -mavx512f) = option "vec_ip_lanes".
#ifndef TSGPU_IP_LANES
#define TSGPU_IP_LANES 4
#endif
inline float InnerProductDistance(const void* a, const void* b, const void* dim_ptr) {
    return tsgpu_ip_distance((const float*)a, (const float*)b, (uint32_t)*(const size_t*)dim_ptr, TSGPU_IP_LANES);
}

class InnerProductSpace {
    size_t dim_;
public:
    explicit InnerProductSpace(size_t dim) : dim_(dim) {}
    DISTFUNC get_dist_func() { return InnerProductDistance; }
    void* get_dist_func_param() { return &dim_; }
    size_t get_data_size() { return dim_ * sizeof(float); }
    size_t dim() const { return dim_; }
};

template <typename dist_t>
class HierarchicalNSW {
    tsgpu_ctx* ctx_;
    uint32_t field_;
    size_t dim_;
    size_t max_elements_;
    std::unordered_set<uint32_t> live_;               // labels a filter functor can be asked about (mutated under the server's unique_lock)
    bool graph_ = false;                              // the library builds and searches hnswlib's graph (constructor: graph_threads > 0)

    static void check(int rc, const char* what) {
        if (rc != TSGPU_OK) throw std::runtime_error(std::string(what) + ": " + tsgpu_last_error());
    }

public:
    // ctx / field_id select the tsgpu vector field this index mirrors; the remaining arguments are hnswlib's
    // graph_threads > 0: the library also BUILDS hnswlib's graph from the addPoint calls (tsgpu_vec_hnsw_enable: M, ef_construction, seed as
    // given; graph_threads host threads per addPoint batch) and searchKnnCloserFirst answers from it (approximate, like hnswlib; parity
    // unpinned) instead of from the exact scan. 0 (default) = exact search, no graph.
    HierarchicalNSW(tsgpu_ctx* ctx, uint32_t field_id, InnerProductSpace* s, size_t max_elements, size_t M = 16,
                    size_t ef_construction = 200, size_t random_seed = 100, bool /*allow_replace_deleted*/ = false, unsigned graph_threads = 0)
        : ctx_(ctx), field_(field_id), dim_(s->dim()), max_elements_(max_elements), graph_(graph_threads > 0) {
        check(tsgpu_vec_create(ctx_, field_, (uint32_t)dim_, TSGPU_METRIC_IP, max_elements), "tsgpu_vec_create");
        if (graph_) check(tsgpu_vec_hnsw_enable(ctx_, field_, (uint32_t)M, (uint32_t)ef_construction, (uint32_t)random_seed, graph_threads), "tsgpu_vec_hnsw_enable");
    }

    size_t getCurrentElementCount() { return (size_t)tsgpu_vec_count(ctx_, field_); }
    size_t getMaxElements() { return max_elements_; }
    void resizeIndex(size_t n) { max_elements_ = n; }       // storage grows on demand inside the library
    void repair_zero_indegree() {}                          // (fork-only call, src/index.cpp:8367; its source is not under /root/reference: nothing to restate)

    void addPoint(const void* data, labeltype label, bool /*replace_deleted*/ = false) {
        uint64_t l = (uint64_t)label;
        check(tsgpu_vec_upsert(ctx_, field_, &l, (const float*)data, 1, TSGPU_MEM_HOST), "tsgpu_vec_upsert");
        live_.insert((uint32_t)label);
        if (getCurrentElementCount() > max_elements_) max_elements_ = getCurrentElementCount();
    }

    void markDelete(labeltype label) {
        check(tsgpu_vec_delete(ctx_, field_, (uint64_t)label), "tsgpu_vec_delete");
        live_.erase((uint32_t)label);
    }

    template <typename data_t>
    std::vector<data_t> getDataByLabel(labeltype label) {
        std::vector<float> v(dim_);
        int rc = tsgpu_vec_get(ctx_, field_, (uint64_t)label, v.data());
        if (rc != TSGPU_OK) throw std::runtime_error("Label not found");
        return std::vector<data_t>(v.begin(), v.end());
    }

    // closest first; filter == nullptr -> whole index.
    // With a functor and NO candidate ids — the reference's unmodified call, searchKnnCloserFirst(q, k, ef, &filterFunctor)
    // (src/index.cpp:3384-3386) — the predicate is applied to the RESULTS of an over-fetching exact search instead of to every live label:
    // the exact k' = 2k nearest are fetched, the functor is asked about those k' labels only, and k' doubles until k of them pass or the
    // index is exhausted. Exact: a passing label outside the k' nearest is farther than every label inside them, and the k'-nearest lists are
    // prefixes of each other (ties -> smaller label). Cost for a functor that rejects nothing / little (no filter_by, a few hidden hits):
    // ONE scan and 2k predicate calls — not live_.size() virtual calls plus a sort per query. A selective functor (fewer than ~k / 1024 of
    // the labels pass) ends in the sweep below; call sites that know the filter's id array should pass it (candidate_ids): the allow-list
    // form is one scan whatever the selectivity.
    std::vector<std::pair<dist_t, labeltype>> searchKnnCloserFirst(const void* query, size_t k, size_t ef = 0,
                                                                   BaseFilterFunctor* filter = nullptr,
                                                                   const uint32_t* candidate_ids = nullptr, uint32_t n_candidates = 0) {
        std::vector<std::pair<dist_t, labeltype>> out;
        if (k == 0) return out;
        std::vector<float> dist;
        std::vector<uint64_t> lab;
        bool use_graph = graph_ && !candidate_ids;
        auto knn = [&](size_t kk, const uint32_t* allow_ptr, uint32_t n_allow) -> uint32_t {
            dist.resize(kk); lab.resize(kk);
            uint32_t n = 0;
            if (use_graph && !allow_ptr) {
                // hnswlib's own search on the graph the library built: max(ef, kk) candidates at layer 0 (the functor, when there is one, is applied
                // to an over-fetched result below instead of inside the traversal). A stale graph (501) / an outgrown heap: the exact scan answers.
                const int rc = tsgpu_vec_hnsw_search_batch(ctx_, field_, (const float*)query, TSGPU_MEM_HOST, 1, (uint32_t)kk, (uint32_t)std::max(ef, kk), filter ? 1 : 0,
                                                           nullptr, 0, nullptr, 0, dist.data(), lab.data(), &n, TSGPU_MEM_HOST);
                if (rc == TSGPU_OK && n != 0xFFFFFFFFu) return n;
                use_graph = false;
            }
            check(tsgpu_vec_knn_batch(ctx_, field_, (const float*)query, TSGPU_MEM_HOST, 1, (uint32_t)kk, allow_ptr, n_allow, nullptr, 0,
                                      dist.data(), lab.data(), &n, TSGPU_MEM_HOST), "tsgpu_vec_knn_batch");
            return n;
        };
        if (!filter) {
            const uint32_t n = knn(k, nullptr, 0);
            for (uint32_t i = 0; i < n; i++) out.emplace_back((dist_t)dist[i], (labeltype)lab[i]);
            return out;
        }
        if (!candidate_ids && k <= TSGPU_MAX_TOPK) {
            std::vector<uint64_t> asked_lab;                    // labels already asked about, in list order (the lists are prefixes of each other) ...
            std::vector<uint8_t> verdict;                       // ... and what the functor said
            for (size_t kk = std::min<size_t>(std::max<size_t>(2 * k, 16), TSGPU_MAX_TOPK);;) {
                const uint32_t n = knn(kk, nullptr, 0);
                size_t passed = 0;
                out.clear();
                for (uint32_t i = 0; i < n && passed < k; i++) {
                    if (i < asked_lab.size() && asked_lab[i] != lab[i]) { asked_lab.resize(i); verdict.resize(i); }      // (never expected: the prefix property)
                    if (i >= asked_lab.size()) { asked_lab.push_back(lab[i]); verdict.push_back((*filter)((labeltype)lab[i]) ? 1 : 0); }
                    if (verdict[i]) { out.emplace_back((dist_t)dist[i], (labeltype)lab[i]); passed++; }
                }
                if (passed >= k) return out;
                if (n < kk) {                                   // fewer than asked for: the whole index has been seen — or the graph search found no more
                    if (!use_graph) return out;
                    use_graph = false; asked_lab.clear(); verdict.clear();      // (approximate lists are not prefixes of the exact ones)
                    continue;
                }
                // what it would take at the pass rate seen so far; beyond the largest supported k the predicate has to be swept
                const size_t need = passed ? (k * (size_t)n + passed - 1) / passed * 3 / 2 : kk * 4;
                if (kk >= TSGPU_MAX_TOPK || need > TSGPU_MAX_TOPK) break;
                kk = std::min<size_t>(std::max<size_t>(need, kk * 2), TSGPU_MAX_TOPK);
            }
            out.clear();
        }
        std::vector<uint32_t> allow;
        // the predicate evaluated into an allow-list: over the ids the call site hands over (the filter's sorted id array), or over every
        // live label (selective functors without a candidate list). NEVER skipped: a dropped functor would return filtered-out / hidden documents.
        if (candidate_ids) {
            for (uint32_t i = 0; i < n_candidates; i++) if ((*filter)(candidate_ids[i])) allow.push_back(candidate_ids[i]);
        } else {
            allow.reserve(live_.size());
            for (uint32_t l : live_) if ((*filter)((labeltype)l)) allow.push_back(l);
        }
        std::sort(allow.begin(), allow.end());
        allow.erase(std::unique(allow.begin(), allow.end()), allow.end());
        if (allow.empty()) return {};
        const uint32_t n = knn(k, allow.data(), (uint32_t)allow.size());
        for (uint32_t i = 0; i < n; i++) out.emplace_back((dist_t)dist[i], (labeltype)lab[i]);
        return out;
    }
};

// Graph search instead of the exact scan: mirror the adjacency of a REAL hnswlib index (the reference keeps building it in
// its indexing path) into HBM, then tsgpu_vec_hnsw_search_batch replays searchKnnCloserFirst on it. H = hnswlib::HierarchicalNSW<float>
// (its members are public: data_level0_memory_, size_data_per_element_, offsetLevel0_, linkLists_, element_levels_,
// size_links_per_element_, maxlevel_, enterpoint_node_, cur_element_count, M_); a template so that this header compiles
// without hnswlib. The tsgpu field must hold the same vectors in the same insertion order (internal id == row).
template <class H>
inline int mirror_hnsw_graph(tsgpu_ctx* ctx, uint32_t field_id, const H& h) {
    const size_t n = h.cur_element_count, M = h.M_, s0 = 1 + 2 * M, su = 1 + M;
    std::vector<uint32_t> link0(n * s0, 0), upper;
    std::vector<uint64_t> upper_ptr(n + 1, 0);
    for (size_t i = 0; i < n; i++) {
        const unsigned int* ll0 = (const unsigned int*)(h.data_level0_memory_ + i * h.size_data_per_element_ + h.offsetLevel0_);
        const unsigned cnt0 = *((const unsigned short*)ll0);                  // getListCount: low 16 bits of the first word
        link0[i * s0] = cnt0;
        for (unsigned j = 0; j < cnt0; j++) link0[i * s0 + 1 + j] = ll0[1 + j];
        upper_ptr[i] = upper.size() / su;
        for (int level = 1; level <= h.element_levels_[i]; level++) {
            const unsigned int* ll = (const unsigned int*)(h.linkLists_[i] + (size_t)(level - 1) * h.size_links_per_element_);
            const unsigned cnt = *((const unsigned short*)ll);
            const size_t at = upper.size();
            upper.resize(at + su, 0);
            upper[at] = cnt;
            for (unsigned j = 0; j < cnt; j++) upper[at + 1 + j] = ll[1 + j];
        }
    }
    upper_ptr[n] = upper.size() / su;
    return tsgpu_vec_hnsw_load(ctx, field_id, (uint32_t)M, (int32_t)h.maxlevel_, (uint32_t)h.enterpoint_node_, link0.data(), upper_ptr.data(),
                               upper.empty() ? nullptr : upper.data(), (uint32_t)n);
}

}  // namespace tsgpu
