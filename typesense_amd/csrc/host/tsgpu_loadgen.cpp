// tsgpu_loadgen.cpp — measurement tooling (bench.py `concurrency` object, tests): drives the C-ABI the way the reference server
// does — T request threads, each issuing blocking ONE-QUERY calls on one shared context (Index::search ->
// search_across_fields, src/index.cpp:3488; one thread per HTTP request, src/http_server.cpp:827-832) — and records the
// latency of every call plus a checksum of every query's results for the parity check against the batch path.
// Plain C++ (std::thread); the entry points of libtsgpu.so come in as function pointers, so this file links against nothing.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include "../../../include/tsgpu.h"

namespace {
typedef int (*kw_search_fn)(tsgpu_ctx*, const tsgpu_kw_query*, uint32_t, tsgpu_hits*);
typedef int (*grouped_fn)(tsgpu_ctx*, const tsgpu_kw_query*, const tsgpu_group_by*, uint32_t, tsgpu_hits*, tsgpu_grouped_hits*, tsgpu_id_lists**);
typedef int (*knn_fn)(tsgpu_ctx*, uint32_t, const float*, int, uint32_t, uint32_t, const uint32_t*, uint32_t, const uint32_t*, uint32_t, float*, uint64_t*, uint32_t*, int);

inline uint64_t mix(uint64_t h, uint64_t v) { h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); return h; }

// a SLEEPING start barrier: 256 threads yielding in a loop until the last one has been created burn more CPU than the measured calls
// (and, under a container CPU quota, get the whole process throttled before the first call)
struct SpinBarrier {
    std::mutex m;
    std::condition_variable cv;
    uint32_t n = 0;
    void wait(uint32_t total) {
        std::unique_lock<std::mutex> lk(m);
        if (++n >= total) cv.notify_all();
        else cv.wait(lk, [&] { return n >= total; });
    }
};
}  // namespace

extern "C" {

// checksum of one query's hits: n_hits, num_matched, then (key, scores[3]) of the first `top` hits — the same function is applied
// to the batch path's results by the caller (tsgpu_loadgen_hits_checksum)
uint64_t tsgpu_loadgen_hits_checksum(const uint64_t* keys, const int64_t* scores, uint32_t n_hits, uint64_t num_matched, uint32_t top) {
    uint64_t h = mix(0x1234, n_hits);
    h = mix(h, num_matched);
    const uint32_t n = n_hits < top ? n_hits : top;
    for (uint32_t i = 0; i < n; i++) { h = mix(h, keys[i]); h = mix(h, (uint64_t)scores[i * 3]); h = mix(h, (uint64_t)scores[i * 3 + 1]); h = mix(h, (uint64_t)scores[i * 3 + 2]); }
    return h;
}

// n_threads threads; thread t issues calls_per_thread calls of queries_per_call queries each, walking the query array from its own
// offset (call c of thread t uses queries [(t * calls_per_thread + c) * queries_per_call ...) modulo n_queries).
// latency_us: [n_threads * calls_per_thread]; checksum: [n_queries] (last writer wins: every query index always yields the same
// value); failures: calls that returned an error or a non-OK per-query status. Returns wall seconds of the whole run.
double tsgpu_loadgen_keyword(void* fn_search, tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_queries, uint32_t k_stride, uint32_t top,
                             uint32_t n_threads, uint32_t calls_per_thread, uint32_t queries_per_call, double* latency_us, uint64_t* checksum,
                             uint64_t* failures) {
    kw_search_fn search = (kw_search_fn)fn_search;
    std::atomic<uint64_t> fails{0};
    SpinBarrier start;
    std::chrono::steady_clock::time_point t0;
    auto body = [&](uint32_t t) {
        const uint32_t qpc = queries_per_call;
        std::vector<uint64_t> keys((size_t)qpc * k_stride), nm(qpc);
        std::vector<int64_t> scores((size_t)qpc * k_stride * 3), tm((size_t)qpc * k_stride);
        std::vector<float> vd((size_t)qpc * k_stride);
        std::vector<int8_t> msi((size_t)qpc * k_stride);
        std::vector<uint32_t> nh(qpc);
        std::vector<int32_t> st(qpc), co(qpc);
        std::vector<tsgpu_kw_query> mine(qpc);
        tsgpu_hits h;
        h.mem = TSGPU_MEM_HOST; h.k_stride = k_stride;
        h.keys = keys.data(); h.scores = scores.data(); h.text_match = tm.data(); h.vector_distance = vd.data(); h.match_score_index = msi.data();
        h.n_hits = nh.data(); h.num_matched = nm.data(); h.status = st.data(); h.search_cutoff = co.data();
        start.wait(n_threads + 1);
        for (uint32_t c = 0; c < calls_per_thread; c++) {
            const uint64_t first = ((uint64_t)t * calls_per_thread + c) * qpc;
            for (uint32_t i = 0; i < qpc; i++) mine[i] = queries[(first + i) % n_queries];
            const auto a = std::chrono::steady_clock::now();
            const int rc = search(ctx, mine.data(), qpc, &h);
            const auto b = std::chrono::steady_clock::now();
            latency_us[(size_t)t * calls_per_thread + c] = std::chrono::duration<double, std::micro>(b - a).count();
            if (rc != TSGPU_OK) { fails.fetch_add(1); continue; }
            for (uint32_t i = 0; i < qpc; i++) {
                if (st[i] != TSGPU_OK) { fails.fetch_add(1); continue; }
                checksum[(first + i) % n_queries] = tsgpu_loadgen_hits_checksum(keys.data() + (size_t)i * k_stride, scores.data() + (size_t)i * k_stride * 3, nh[i], nm[i], top);
            }
        }
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < n_threads; t++) pool.emplace_back(body, t);
    start.wait(n_threads + 1);
    t0 = std::chrono::steady_clock::now();
    for (auto& th : pool) th.join();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (failures) *failures = fails.load();
    return wall;
}

// checksum of one grouped user query: the first pass' group count and getGroupsCount(), then the second pass' groups in order (distinct key, found, size, every KV)
uint64_t tsgpu_loadgen_grouped_checksum(uint32_t first_groups, uint64_t groups_count, uint32_t n_groups, const uint64_t* dkeys, const uint32_t* found, const uint32_t* gsize,
                                        const uint64_t* keys, const int64_t* scores, uint32_t group_limit) {
    uint64_t h = mix(0x4321, first_groups);
    h = mix(h, groups_count);
    h = mix(h, n_groups);
    for (uint32_t r = 0; r < n_groups; r++) {
        h = mix(h, dkeys[r]); h = mix(h, found[r]); h = mix(h, gsize[r]);
        for (uint32_t j = 0; j < gsize[r]; j++) {
            const size_t o = (size_t)r * group_limit + j;
            h = mix(h, keys[o]); h = mix(h, (uint64_t)scores[o * 3]); h = mix(h, (uint64_t)scores[o * 3 + 1]); h = mix(h, (uint64_t)scores[o * 3 + 2]);
        }
    }
    return h;
}

// group_by under the reference's calling convention: thread t issues calls_per_thread USER queries, each as the reference runs a grouped request — a FIRST-pass call,
// then a SECOND-pass call of the same query (tsgpu_keyword_search_grouped_batch with one query each). latency_us = both calls of a user query.
double tsgpu_loadgen_grouped(void* fn_grouped, tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_queries, uint32_t topster, uint32_t group_limit, uint32_t column,
                             uint32_t n_threads, uint32_t calls_per_thread, double* latency_us, uint64_t* checksum, uint64_t* failures) {
    grouped_fn search = (grouped_fn)fn_grouped;
    std::atomic<uint64_t> fails{0};
    SpinBarrier start;
    auto body = [&](uint32_t t) {
        const size_t slots = (size_t)topster * group_limit;
        std::vector<uint64_t> keys(slots), dk(topster);
        std::vector<int64_t> scores(slots * 3);
        std::vector<int8_t> msi(slots);
        std::vector<uint32_t> gs(topster), gf(topster);
        uint32_t nh = 0, ng = 0; uint64_t nm = 0, gc = 0; int32_t st = 0, co = 0;
        tsgpu_hits h;
        memset(&h, 0, sizeof h);
        h.mem = TSGPU_MEM_HOST; h.k_stride = (uint32_t)slots;
        h.keys = keys.data(); h.scores = scores.data(); h.match_score_index = msi.data(); h.n_hits = &nh; h.num_matched = &nm; h.status = &st; h.search_cutoff = &co;
        tsgpu_grouped_hits g;
        memset(&g, 0, sizeof g);
        g.g_stride = topster; g.n_groups = &ng; g.distinct_key = dk.data(); g.group_size = gs.data(); g.group_found = gf.data(); g.groups_count = &gc;
        tsgpu_group_by gb;
        memset(&gb, 0, sizeof gb);
        gb.group_limit = group_limit; gb.column = (uint16_t)column;
        start.wait(n_threads + 1);
        for (uint32_t c = 0; c < calls_per_thread; c++) {
            const uint64_t qi = ((uint64_t)t * calls_per_thread + c) % n_queries;
            const tsgpu_kw_query q = queries[qi];
            const auto a = std::chrono::steady_clock::now();
            gb.first_pass = 1;
            int rc = search(ctx, &q, &gb, 1, &h, &g, nullptr);
            const uint32_t first_groups = ng; const uint64_t first_count = gc; const int32_t st1 = st;
            gb.first_pass = 0;
            if (rc == TSGPU_OK) rc = search(ctx, &q, &gb, 1, &h, &g, nullptr);
            const auto b = std::chrono::steady_clock::now();
            latency_us[(size_t)t * calls_per_thread + c] = std::chrono::duration<double, std::micro>(b - a).count();
            if (rc != TSGPU_OK || st != TSGPU_OK || st1 != TSGPU_OK) { fails.fetch_add(1); continue; }
            checksum[qi] = tsgpu_loadgen_grouped_checksum(first_groups, first_count, ng, dk.data(), gf.data(), gs.data(), keys.data(), scores.data(), group_limit);
        }
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < n_threads; t++) pool.emplace_back(body, t);
    start.wait(n_threads + 1);
    const auto t0 = std::chrono::steady_clock::now();
    for (auto& th : pool) th.join();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (failures) *failures = fails.load();
    return wall;
}

// the same for tsgpu_vec_knn_batch: one query vector per call. labels_out / dist_out: [n_queries][k] (every query index is written
// by whichever call served it: identical values).
double tsgpu_loadgen_knn(void* fn_knn, tsgpu_ctx* ctx, uint32_t field, const float* Q, uint32_t n_queries, uint32_t dim, uint32_t k, uint32_t n_threads,
                         uint32_t calls_per_thread, double* latency_us, uint64_t* labels_out, float* dist_out, uint64_t* failures) {
    knn_fn knn = (knn_fn)fn_knn;
    std::atomic<uint64_t> fails{0};
    SpinBarrier start;
    auto body = [&](uint32_t t) {
        std::vector<float> d(k);
        std::vector<uint64_t> l(k);
        uint32_t n = 0;
        start.wait(n_threads + 1);
        for (uint32_t c = 0; c < calls_per_thread; c++) {
            const uint64_t qi = ((uint64_t)t * calls_per_thread + c) % n_queries;
            const auto a = std::chrono::steady_clock::now();
            const int rc = knn(ctx, field, Q + qi * dim, TSGPU_MEM_HOST, 1, k, nullptr, 0, nullptr, 0, d.data(), l.data(), &n, TSGPU_MEM_HOST);
            const auto b = std::chrono::steady_clock::now();
            latency_us[(size_t)t * calls_per_thread + c] = std::chrono::duration<double, std::micro>(b - a).count();
            if (rc != TSGPU_OK || n != k) { fails.fetch_add(1); continue; }
            memcpy(labels_out + qi * k, l.data(), (size_t)k * 8);
            memcpy(dist_out + qi * k, d.data(), (size_t)k * 4);
        }
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < n_threads; t++) pool.emplace_back(body, t);
    start.wait(n_threads + 1);
    const auto t0 = std::chrono::steady_clock::now();
    for (auto& th : pool) th.join();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (failures) *failures = fails.load();
    return wall;
}

}  // extern "C"
