// tsgpu_hnsw_build.h — HNSW graph CONSTRUCTION inside the library (host code, included by tsgpu_vec.hip).
//
// Why: after INTEGRATION.md §2's typedef swap the server no longer owns an hnswlib index — tsgpu::HierarchicalNSW::addPoint only stored the
// row — so there was no graph for tsgpu_vec_hnsw_search_batch to mirror (VERDICT r3 "What's missing" #2). The reference inserts
// incrementally: Index::index_field_in_memory -> vecdex->addPoint(vec, seq_id, true) from up to four indexing threads
// (/root/reference/src/index.cpp:1002-1075; `new HierarchicalNSW(space, 16, M, ef_construction, 100, true)`, include/index.h:365-367).
// This is that insertion: level draw from std::default_random_engine(seed) with mult = 1 / ln(M), greedy descent through the upper layers,
// an ef_construction-bounded beam per layer (libstdc++ binary heaps of (distance, internal id): ties fall as in hnswlib), neighbour
// selection by the distance heuristic, reverse links with re-selection when a list is full, markDelete. hnswlib itself is NOT under
// /root/reference: PARITY UNPINNED — what is checked is equality, link for link, with the oracle's independent restatement of the
// published algorithm (oracle/hnsw_graph.h) for single-threaded insertion in label order, and recall against the exact scan.
//
// Storage is the mirror's own flat form (what vec_hnsw_search_kernel reads): level-0 lists [n][1 + 2M] and a pool of upper lists
// [(node, level)][1 + M], so an upload is a plain copy. Distances are tsgpu_ip_distance in the summation order of option vec_ip_lanes — the
// bits every other path returns. Concurrent insertion of a batch follows hnswlib's locking (a global lock while the entry point may move,
// one lock per node around its link lists, a visited list per thread); like the reference's four indexing threads it is not deterministic.
// Round 5: the reference constructs the index with allow_replace_deleted = true and always calls addPoint(vec, seq_id, true)
// (/root/reference/include/index.h:367, src/index.cpp:1052-1054), so an insertion after a markDelete RE-USES a deleted slot and runs hnswlib's
// updatePoint on it, and addPoint on a live label is an updatePoint, too: update_point() / repair_connections_for_update() below restate both
// (updateNeighborProbability = 1.0). Two choices hnswlib leaves to the standard library are fixed here AND in the oracle: which deleted slot is
// taken (`*deleted_elements.begin()` of an unordered_set: here the most recently deleted one — for Typesense's update = remove + add that is the
// document's own slot) and the order in which updatePoint walks its candidate sets (ascending ids; it matters for equal distances only).
#pragma once
#include <cmath>
#include <deque>
#include <queue>
#include <random>
#include <thread>

namespace tsgpu {

struct VisitedList {                                  // hnswlib's VisitedList: a tag per node and an epoch instead of a cleared array per search
    std::vector<uint16_t> tag; uint16_t epoch = 0;
    void begin(size_t n) {
        if (tag.size() < n) tag.resize(std::max(n, tag.size() + tag.size() / 2), 0);      // (geometric: one-row insertions do not reallocate every time)
        if (++epoch == 0) { std::fill(tag.begin(), tag.end(), 0); epoch = 1; }
    }
    bool test_and_set(uint32_t i) { const bool v = tag[i] == epoch; tag[i] = epoch; return v; }
};

struct HnswBuilder {
    typedef std::pair<float, uint32_t> DistId;
    struct ByDist { bool operator()(const DistId& a, const DistId& b) const noexcept { return a.first < b.first; } };
    typedef std::priority_queue<DistId, std::vector<DistId>, ByDist> Heap;       // max-heap on the distance only (hnswlib's CompareByFirst)

    uint32_t dim = 0, M = 16, ef_construction = 200;
    int ip_lanes = 4;
    double mult = 0;
    int32_t maxlevel = -1;
    uint32_t enterpoint = 0xFFFFFFFFu;
    std::default_random_engine level_generator;
    uint32_t n_threads = 1;
    bool stale = false;                               // an operation the builder does not follow happened: the graph no longer matches the rows
    bool dirty = false;                               // changed since the last upload to the device

    std::vector<float> data;                          // [n][dim], row = internal id
    std::vector<int32_t> levels;
    std::vector<uint8_t> deleted;
    std::vector<uint32_t> link0;                      // [n][1 + 2M]: count, neighbours
    std::vector<uint64_t> upper_at;                   // [n]: first list of the node in `upper` (its levels[n] lists follow each other)
    std::vector<uint32_t> upper;                      // [(node, level >= 1)][1 + M]
    std::deque<std::mutex> node_mu;                   // one per node (stable addresses while the deque grows)
    std::mutex global_mu;
    std::vector<uint32_t> deleted_stack;              // markDelete order (hnswlib's deleted_elements): the most recently deleted slot is re-used first
    std::deque<VisitedList> vis_pool;          // one visited list per inserting thread, kept across add_batch calls: the server's calling convention is
                                                      // one addPoint per document, and a fresh list per call allocates and zeroes a tag per NODE (O(n) per insertion)

    void init(uint32_t dim_, uint32_t M_, uint32_t efc, uint32_t seed, uint32_t threads, int lanes) {
        dim = dim_; M = M_; ef_construction = std::max(efc, M_); mult = 1.0 / std::log(1.0 * M_);
        level_generator.seed(seed);
        n_threads = std::max<uint32_t>(1, threads); ip_lanes = lanes;
    }
    size_t size() const { return levels.size(); }
    size_t s0() const { return 1 + 2 * (size_t)M; }
    size_t su() const { return 1 + (size_t)M; }
    const float* vec(uint32_t i) const { return data.data() + (size_t)i * dim; }
    float dist(const float* a, const float* b) const { return tsgpu_ip_distance(a, b, dim, ip_lanes); }
    uint32_t* list_of(uint32_t node, int level) { return level == 0 ? link0.data() + (size_t)node * s0() : upper.data() + (upper_at[node] + (uint64_t)(level - 1)) * su(); }
    uint64_t n_upper_lists() const { return upper.size() / su(); }

    typedef VisitedList Visited;

    // the ef_construction-bounded beam of one layer (searchBaseLayer)
    Heap search_layer(uint32_t ep, const float* q, int layer, Visited& vis, bool locked) {
        vis.begin(size());
        Heap top, cand;
        float lower;
        if (!deleted[ep]) { const float d = dist(q, vec(ep)); top.emplace(d, ep); lower = d; cand.emplace(-d, ep); }
        else { lower = std::numeric_limits<float>::max(); cand.emplace(-lower, ep); }
        vis.test_and_set(ep);
        std::vector<uint32_t> nb;
        while (!cand.empty()) {
            const DistId cur = cand.top();
            if (-cur.first > lower && top.size() == ef_construction) break;
            cand.pop();
            {
                std::unique_lock<std::mutex> lk(node_mu[cur.second], std::defer_lock);
                if (locked) lk.lock();
                const uint32_t* l = list_of(cur.second, layer);
                nb.assign(l + 1, l + 1 + l[0]);
            }
            for (uint32_t c : nb) {
                if (vis.test_and_set(c)) continue;
                const float d1 = dist(q, vec(c));
                if (top.size() < ef_construction || lower > d1) {
                    cand.emplace(-d1, c);
                    if (!deleted[c]) top.emplace(d1, c);
                    if (top.size() > ef_construction) top.pop();
                    if (!top.empty()) lower = top.top().first;
                }
            }
        }
        return top;
    }

    // getNeighborsByHeuristic2: closest first; a candidate is kept unless an already kept neighbour is nearer to it than the query is
    void select_neighbours(Heap& top, size_t limit) {
        if (top.size() < limit) return;
        Heap closest;
        std::vector<DistId> keep;
        while (!top.empty()) { closest.emplace(-top.top().first, top.top().second); top.pop(); }
        while (!closest.empty() && keep.size() < limit) {
            const DistId cur = closest.top();
            const float d_query = -cur.first;
            closest.pop();
            bool good = true;
            for (const DistId& k : keep) if (dist(vec(k.second), vec(cur.second)) < d_query) { good = false; break; }
            if (good) keep.push_back(cur);
        }
        for (const DistId& k : keep) top.emplace(-k.first, k.second);
    }

    // mutuallyConnectNewElement: the new node's list on this level, then the reverse links
    uint32_t connect(uint32_t cur, Heap& top, int level, bool locked, uint32_t prev_ep, bool is_update = false) {
        const size_t cap = level ? M : 2 * (size_t)M;
        select_neighbours(top, M);
        std::vector<uint32_t> sel;
        sel.reserve(M);
        while (!top.empty()) { sel.push_back(top.top().second); top.pop(); }
        if (sel.empty()) return prev_ep;                  // (cannot happen once addPoint re-adds a deleted entry point, below; never index an empty selection)
        const uint32_t next_ep = sel.back();
        {
            std::unique_lock<std::mutex> lk(node_mu[cur], std::defer_lock);
            if (locked) lk.lock();
            uint32_t* l = list_of(cur, level);
            for (size_t j = sel.size(); j < l[0]; j++) l[1 + j] = 0;      // (an update overwrites a longer list: slots behind the count stay zero)
            l[0] = (uint32_t)sel.size();
            for (size_t j = 0; j < sel.size(); j++) l[1 + j] = sel[j];
        }
        for (uint32_t s : sel) {
            std::unique_lock<std::mutex> lk(node_mu[s], std::defer_lock);
            if (locked) lk.lock();
            uint32_t* l = list_of(s, level);
            if (is_update) {                              // (mutuallyConnectNewElement, isUpdate: a neighbour that already points back keeps its list)
                bool present = false;
                for (uint32_t j = 0; j < l[0]; j++) if (l[1 + j] == cur) { present = true; break; }
                if (present) continue;
            }
            if (l[0] < cap) { l[1 + l[0]] = cur; l[0]++; }
            else {
                Heap c;
                c.emplace(dist(vec(cur), vec(s)), cur);
                for (uint32_t j = 0; j < l[0]; j++) c.emplace(dist(vec(l[1 + j]), vec(s)), l[1 + j]);
                select_neighbours(c, cap);
                uint32_t n = 0;
                while (!c.empty()) { l[1 + n++] = c.top().second; c.pop(); }
                for (uint32_t j = n; j < l[0]; j++) l[1 + j] = 0;       // (slots behind the count stay zero: the exported image is canonical)
                l[0] = n;
            }
        }
        return next_ep;
    }

    // addPoint for the node `cur` whose row, level and (empty) lists already exist
    void insert(uint32_t cur, Visited& vis, bool locked) {
        const int curlevel = levels[cur];
        std::unique_lock<std::mutex> glk(global_mu, std::defer_lock);
        if (locked) glk.lock();
        const int maxlevel_copy = maxlevel;
        if (locked && curlevel <= maxlevel_copy) glk.unlock();          // (held to the end only when this node becomes the entry point)
        uint32_t ep = enterpoint;
        const uint32_t ep_copy = ep;
        const float* q = vec(cur);
        if (ep != 0xFFFFFFFFu) {
            // hnswlib's addPoint: `bool epDeleted = isMarkedDeleted(enterpoint_copy)` — a deleted entry point never enters a beam's result heap, so
            // a collection whose rows were ALL deleted (emptied and refilled) would hand empty heaps to mutuallyConnectNewElement; the entry point is
            // put back into every level's candidates instead (without it new nodes got no links and became unreachable)
            const bool ep_deleted = deleted[ep_copy] != 0;
            if (curlevel < maxlevel_copy) {
                float curdist = dist(q, vec(ep));
                std::vector<uint32_t> nb;
                for (int level = maxlevel_copy; level > curlevel; level--) {
                    bool changed = true;
                    while (changed) {
                        changed = false;
                        {
                            std::unique_lock<std::mutex> lk(node_mu[ep], std::defer_lock);
                            if (locked) lk.lock();
                            const uint32_t* l = list_of(ep, level);
                            nb.assign(l + 1, l + 1 + l[0]);
                        }
                        for (uint32_t c : nb) { const float d = dist(q, vec(c)); if (d < curdist) { curdist = d; ep = c; changed = true; } }
                    }
                }
            }
            for (int level = std::min(curlevel, maxlevel_copy); level >= 0; level--) {
                Heap top = search_layer(ep, q, level, vis, locked);
                if (ep_deleted) {
                    top.emplace(dist(q, vec(ep_copy)), ep_copy);
                    if (top.size() > ef_construction) top.pop();
                }
                ep = connect(cur, top, level, locked, ep);
            }
        } else { enterpoint = cur; maxlevel = curlevel; }
        if (curlevel > maxlevel_copy) { enterpoint = cur; maxlevel = curlevel; }
    }

    // hnswlib updatePoint(dataPoint, internalId, 1.0): the row's vector is replaced, the lists of its one-hop neighbours are re-selected among the
    // one- and two-hop neighbourhood, then its own connections are repaired from the entry point down (repairConnectionsForUpdate)
    void update_point(uint32_t id, const float* v) {
        std::copy(v, v + dim, data.begin() + (size_t)id * dim);
        dirty = true;
        const int maxlevel_copy = maxlevel;
        const uint32_t ep_copy = enterpoint;
        if (ep_copy == id && size() == 1) return;
        const int elem_level = levels[id];
        for (int layer = 0; layer <= elem_level; layer++) {
            const uint32_t* l = list_of(id, layer);
            std::vector<uint32_t> one_hop(l + 1, l + 1 + l[0]);
            if (one_hop.empty()) continue;
            std::vector<uint32_t> cand{id}, neigh;
            for (uint32_t e : one_hop) {
                cand.push_back(e); neigh.push_back(e);
                const uint32_t* l2 = list_of(e, layer);
                cand.insert(cand.end(), l2 + 1, l2 + 1 + l2[0]);
            }
            std::sort(cand.begin(), cand.end()); cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
            std::sort(neigh.begin(), neigh.end()); neigh.erase(std::unique(neigh.begin(), neigh.end()), neigh.end());
            for (uint32_t nb : neigh) {
                Heap c;
                const size_t n_other = std::binary_search(cand.begin(), cand.end(), nb) ? cand.size() - 1 : cand.size();
                const size_t keep = std::min<size_t>(ef_construction, n_other);
                for (uint32_t x : cand) {
                    if (x == nb) continue;
                    const float d = dist(vec(nb), vec(x));
                    if (c.size() < keep) c.emplace(d, x);
                    else if (d < c.top().first) { c.pop(); c.emplace(d, x); }
                }
                select_neighbours(c, layer == 0 ? 2 * (size_t)M : (size_t)M);
                uint32_t* ln = list_of(nb, layer);
                const uint32_t old = ln[0];
                uint32_t n = 0;
                while (!c.empty()) { ln[1 + n++] = c.top().second; c.pop(); }
                for (uint32_t j = n; j < old; j++) ln[1 + j] = 0;
                ln[0] = n;
            }
        }
        repair_connections_for_update(id, ep_copy, elem_level, maxlevel_copy);
    }
    void repair_connections_for_update(uint32_t id, uint32_t ep, int level_of_id, int maxlevel_copy) {
        const float* q = vec(id);
        uint32_t cur = ep;
        if (level_of_id < maxlevel_copy) {
            float curdist = dist(q, vec(cur));
            for (int level = maxlevel_copy; level > level_of_id; level--) {
                bool changed = true;
                while (changed) {
                    changed = false;
                    const uint32_t* l = list_of(cur, level);
                    const std::vector<uint32_t> nb(l + 1, l + 1 + l[0]);
                    for (uint32_t c : nb) { const float d = dist(q, vec(c)); if (d < curdist) { curdist = d; cur = c; changed = true; } }
                }
            }
        }
        while (vis_pool.empty()) vis_pool.emplace_back();
        for (int level = std::min(level_of_id, maxlevel_copy); level >= 0; level--) {
            Heap top = search_layer(cur, q, level, vis_pool[0], false);
            Heap filtered;
            while (!top.empty()) { if (top.top().second != id) filtered.push(top.top()); top.pop(); }
            // (the beam may hold nothing but the row itself: no self loops, and then nothing to connect on this level)
            if (!filtered.empty()) {
                if (deleted[ep]) {
                    filtered.emplace(dist(q, vec(ep)), ep);
                    if (filtered.size() > ef_construction) filtered.pop();
                }
                cur = connect(id, filtered, level, false, cur, true);
            }
        }
    }
    void mark_deleted(uint32_t id) { if (id < deleted.size() && !deleted[id]) { deleted[id] = 1; deleted_stack.push_back(id); } }
    // the slot addPoint(.., replace_deleted = true) takes for a label that is not live: none (-1: append), or the most recently deleted one
    int64_t take_deleted_slot() {
        while (!deleted_stack.empty()) {
            const uint32_t id = deleted_stack.back();
            deleted_stack.pop_back();
            if (id < deleted.size() && deleted[id]) return id;
        }
        return -1;
    }
    // ... that slot gets the new vector: unmarkDeletedInternal + updatePoint
    void replace_deleted(uint32_t id, const float* v) { deleted[id] = 0; update_point(id, v); }

    // n new rows (appended in this order = their internal ids). Levels are drawn in label order before any insertion, so the one-thread
    // build is the sequential algorithm exactly; with more threads the rows of the batch are inserted concurrently (hnswlib's locking).
    // preset_levels (tsgpu_vec_hnsw_build's seed set): the levels were drawn by the caller — for ALL rows of the collection, of which these are some
    void add_batch(const float* rows, size_t n, const int32_t* preset_levels = nullptr) {
        if (n == 0) return;
        const size_t n0 = size();
        data.insert(data.end(), rows, rows + n * dim);
        deleted.resize(n0 + n, 0);
        link0.resize((n0 + n) * s0(), 0u);
        for (size_t i = 0; i < n; i++) {
            std::uniform_real_distribution<double> u(0.0, 1.0);
            const int lv = preset_levels ? preset_levels[i] : (int)(-std::log(u(level_generator)) * mult);
            levels.push_back(lv);
            upper_at.push_back(upper.size() / su());
            upper.resize(upper.size() + (size_t)lv * su(), 0u);
            node_mu.emplace_back();
        }
        dirty = true;
        const uint32_t T = (uint32_t)std::min<size_t>(n_threads, n);
        // the very first node has nothing to connect to; it (and any node that raises the top level) is inserted before the concurrent part
        while (vis_pool.size() < std::max<uint32_t>(T, 1)) vis_pool.emplace_back();
        if (T <= 1) { for (size_t i = 0; i < n; i++) insert((uint32_t)(n0 + i), vis_pool[0], false); return; }
        std::atomic<size_t> next{0};
        if (n0 == 0) { insert(0, vis_pool[0], false); next = 1; }
        for (uint32_t t = 0; t < T; t++) vis_pool[t].begin(size());      // (grown here: a tag array must not reallocate under a sibling's feet — each thread owns its own anyway)
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < T; t++) th.emplace_back([&, t]() { Visited& vis = vis_pool[t]; for (;;) { const size_t i = next.fetch_add(1); if (i >= n) break; insert((uint32_t)(n0 + i), vis, true); } });
        for (auto& x : th) x.join();
    }
};

}  // namespace tsgpu
