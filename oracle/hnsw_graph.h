// TEST INFRASTRUCTURE ONLY (oracle/): CPU restatement of the HNSW index Typesense uses through
// hnswlib::HierarchicalNSW<float> (include/index.h:356-370; call sites src/index.cpp:1003-1054 addPoint,
// :3376-3445 searchKnnCloserFirst with a VectorFilterFunctor, :7423 markDelete).
//
// PARITY UNPINNED: hnswlib is a third-party dependency that is NOT under /root/reference (typesense/hnswlib fork,
// cmake/hnsw.cmake:3 pins 21de18ffabea1a9d1e8b16b49afc6045d7707e4c, WORKSPACE:186-191 pins 687d9817...). This file restates
// the published algorithm of hnswlib 0.7 (Malkov & Yashunin; hnswalg.h: getRandomLevel, addPoint, searchBaseLayer,
// getNeighborsByHeuristic2, mutuallyConnectNewElement, searchKnn, searchBaseLayerST) from upstream knowledge; no reference
// test fixes which approximate neighbours come back, so the GPU traversal is checked against THIS restatement
// on the same graph, and its recall against the exact scan. Distances are the oracle's InnerProductSpace restatement
// (Index::ip_distance), i.e. bit-identical to what the exact path returns.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <queue>
#include <random>
#include <unordered_map>
#include <vector>

namespace oracle {

struct hnsw_graph_t {
    typedef uint32_t tableint;
    typedef std::pair<float, tableint> dist_id_t;
    struct CompareByFirst {
        constexpr bool operator()(const dist_id_t& a, const dist_id_t& b) const noexcept { return a.first < b.first; }
    };
    typedef std::priority_queue<dist_id_t, std::vector<dist_id_t>, CompareByFirst> heap_t;

    size_t dim = 0, M = 16, maxM = 16, maxM0 = 32, ef_construction = 200;
    double mult = 0;
    int maxlevel = -1;
    tableint enterpoint = (tableint)-1;
    std::default_random_engine level_generator;                 // hnswlib: level_generator_.seed(random_seed), seed 100 (include/index.h:367)
    float (*distfn)(const float*, const float*, size_t) = nullptr;

    std::vector<float> data;                                    // [n][dim] (rows in insertion order = internal ids)
    std::vector<uint64_t> labels;
    std::vector<uint8_t> deleted;
    std::vector<int> levels;
    std::vector<std::vector<tableint>> link0;                   // level 0: up to maxM0 neighbours
    std::vector<std::vector<std::vector<tableint>>> linkU;      // [node][level-1]: up to maxM neighbours
    std::unordered_map<uint64_t, tableint> label_lookup;        // hnswlib label_lookup_
    std::vector<tableint> deleted_order;                        // hnswlib deleted_elements (an unordered_set there: WHICH deleted slot addPoint re-uses is the standard
                                                                // library's choice; here, and in the library's builder, the most recently deleted one)

    void init(size_t dim_, size_t M_, size_t efc, size_t seed, float (*fn)(const float*, const float*, size_t)) {
        dim = dim_; M = M_; maxM = M_; maxM0 = 2 * M_; ef_construction = std::max(efc, M_);
        mult = 1 / log(1.0 * M);
        level_generator.seed(seed);
        distfn = fn;
    }
    size_t size() const { return labels.size(); }
    const float* vec(tableint i) const { return data.data() + (size_t)i * dim; }
    float dist(const float* a, const float* b) const { return distfn(a, b, dim); }

    int getRandomLevel(double reverse_size) {
        std::uniform_real_distribution<double> distribution(0.0, 1.0);
        double r = -log(distribution(level_generator)) * reverse_size;
        return (int)r;
    }
    std::vector<tableint>& list_of(tableint node, int level) { return level == 0 ? link0[node] : linkU[node][level - 1]; }
    const std::vector<tableint>& list_of(tableint node, int level) const { return level == 0 ? link0[node] : linkU[node][level - 1]; }

    // searchBaseLayer (construction): ef_construction-bounded beam on one layer
    heap_t searchBaseLayer(tableint ep_id, const float* q, int layer) {
        std::vector<uint8_t> visited(size(), 0);
        heap_t top_candidates, candidateSet;
        float lowerBound;
        if (!deleted[ep_id]) {
            float d = dist(q, vec(ep_id));
            top_candidates.emplace(d, ep_id);
            lowerBound = d;
            candidateSet.emplace(-d, ep_id);
        } else {
            lowerBound = std::numeric_limits<float>::max();
            candidateSet.emplace(-lowerBound, ep_id);
        }
        visited[ep_id] = 1;
        while (!candidateSet.empty()) {
            dist_id_t curr = candidateSet.top();
            if ((-curr.first) > lowerBound && top_candidates.size() == ef_construction) break;
            candidateSet.pop();
            const std::vector<tableint>& nb = list_of(curr.second, layer);
            for (size_t j = 0; j < nb.size(); j++) {
                tableint c = nb[j];
                if (visited[c]) continue;
                visited[c] = 1;
                float d1 = dist(q, vec(c));
                if (top_candidates.size() < ef_construction || lowerBound > d1) {
                    candidateSet.emplace(-d1, c);
                    if (!deleted[c]) top_candidates.emplace(d1, c);
                    if (top_candidates.size() > ef_construction) top_candidates.pop();
                    if (!top_candidates.empty()) lowerBound = top_candidates.top().first;
                }
            }
        }
        return top_candidates;
    }

    void getNeighborsByHeuristic2(heap_t& top_candidates, size_t Mlim) {
        if (top_candidates.size() < Mlim) return;
        heap_t queue_closest;
        std::vector<dist_id_t> return_list;
        while (top_candidates.size() > 0) {
            queue_closest.emplace(-top_candidates.top().first, top_candidates.top().second);
            top_candidates.pop();
        }
        while (queue_closest.size()) {
            if (return_list.size() >= Mlim) break;
            dist_id_t cur = queue_closest.top();
            float dist_to_query = -cur.first;
            queue_closest.pop();
            bool good = true;
            for (const dist_id_t& second : return_list) {
                float curdist = dist(vec(second.second), vec(cur.second));
                if (curdist < dist_to_query) { good = false; break; }
            }
            if (good) return_list.push_back(cur);
        }
        for (const dist_id_t& cur : return_list) top_candidates.emplace(-cur.first, cur.second);
    }

    tableint mutuallyConnectNewElement(const float* q, tableint cur_c, heap_t& top_candidates, int level, tableint prev_entry_point, bool isUpdate = false) {
        size_t Mcurmax = level ? maxM : maxM0;
        getNeighborsByHeuristic2(top_candidates, M);
        std::vector<tableint> selected;
        selected.reserve(M);
        while (top_candidates.size() > 0) { selected.push_back(top_candidates.top().second); top_candidates.pop(); }
        if (selected.empty()) return prev_entry_point;          // (unreachable once addPoint re-adds a deleted entry point; never index an empty selection)
        tableint next_closest_entry_point = selected.back();
        list_of(cur_c, level) = selected;
        for (size_t idx = 0; idx < selected.size(); idx++) {
            std::vector<tableint>& other = list_of(selected[idx], level);
            if (isUpdate) {                                     // "If cur_c is already present in the neighboring connections ... no need to modify any connections"
                bool is_cur_c_present = false;
                for (tableint x : other) if (x == cur_c) { is_cur_c_present = true; break; }
                if (is_cur_c_present) continue;
            }
            if (other.size() < Mcurmax) {
                other.push_back(cur_c);
            } else {
                float d_max = dist(vec(cur_c), vec(selected[idx]));
                heap_t candidates;
                candidates.emplace(d_max, cur_c);
                for (size_t j = 0; j < other.size(); j++) candidates.emplace(dist(vec(other[j]), vec(selected[idx])), other[j]);
                getNeighborsByHeuristic2(candidates, Mcurmax);
                other.clear();
                while (candidates.size() > 0) { other.push_back(candidates.top().second); candidates.pop(); }
            }
        }
        (void)q;
        return next_closest_entry_point;
    }

    // addPoint(data, label, replace_deleted = true) for a NEW label (Typesense removes + re-adds on update; src/index.cpp:1052-1054)
    void addPoint(const float* v, uint64_t label) {
        tableint cur_c = (tableint)size();
        labels.push_back(label);
        label_lookup[label] = cur_c;
        deleted.push_back(0);
        data.insert(data.end(), v, v + dim);
        int curlevel = getRandomLevel(mult);
        levels.push_back(curlevel);
        link0.emplace_back();
        linkU.emplace_back((size_t)curlevel);
        int maxlevelcopy = maxlevel;
        tableint currObj = enterpoint;
        const tableint enterpoint_copy = enterpoint;
        const float* q = vec(cur_c);
        if ((int32_t)currObj != -1) {
            // hnswlib addPoint: `bool epDeleted = isMarkedDeleted(enterpoint_copy);` — a deleted entry point is put back into every level's
            // candidates (searchBaseLayer leaves deleted nodes out of its result heap), so that an emptied and refilled index still links its new nodes
            const bool epDeleted = deleted[enterpoint_copy] != 0;
            if (curlevel < maxlevelcopy) {
                float curdist = dist(q, vec(currObj));
                for (int level = maxlevelcopy; level > curlevel; level--) {
                    bool changed = true;
                    while (changed) {
                        changed = false;
                        const std::vector<tableint>& nb = list_of(currObj, level);
                        for (size_t i = 0; i < nb.size(); i++) {
                            float d = dist(q, vec(nb[i]));
                            if (d < curdist) { curdist = d; currObj = nb[i]; changed = true; }
                        }
                    }
                }
            }
            for (int level = std::min(curlevel, maxlevelcopy); level >= 0; level--) {
                heap_t top_candidates = searchBaseLayer(currObj, q, level);
                if (epDeleted) {
                    top_candidates.emplace(dist(q, vec(enterpoint_copy)), enterpoint_copy);
                    if (top_candidates.size() > ef_construction) top_candidates.pop();
                }
                currObj = mutuallyConnectNewElement(q, cur_c, top_candidates, level, currObj);
            }
        } else {
            enterpoint = 0;
            maxlevel = curlevel;
        }
        if (curlevel > maxlevelcopy) { enterpoint = cur_c; maxlevel = curlevel; }
    }

    // ---- BULK build in batches (the library's tsgpu_vec_hnsw_build; csrc/vec_hnsw_build.hip.h states the algorithm) — NOT hnswlib's insertion order: a
    // deterministic batched variant of it whose pieces are hnswlib's (level draw, searchBaseLayer, getNeighborsByHeuristic2, the reverse-link rule).
    //   1. levels of all rows drawn in label order; 2. the seed set (level >= 2, or id < seed_min; level >= 1 when no such row has an upper level) inserted
    //      by the sequential algorithm in id order; 3. the other rows (levels 0 and 1) in id order, in batches of min(max_batch, max(64, linked / 16)): each
    //      searches the graph as it stood BEFORE the batch — the beam of layer 1 if it has that level, and the beam of layer 0 —, keeps <= M neighbours per
    //      layer by the heuristic over the beam (closest first), and asks each of them for a reverse link; after the batch's beams, layer 1 then layer 0: a
    //      node that was asked takes the requests ordered (distance, id): appended while its list has room for all, else the heuristic over list + the
    //      closest requests (256 candidates at most) ordered (distance, id).
    // the heuristic over candidates given closest first (ids + distances to the centre): kept ids in order
    std::vector<dist_id_t> selectClosestFirst(const std::vector<dist_id_t>& cand, size_t Mlim) {
        if (cand.size() < Mlim) return cand;
        std::vector<dist_id_t> keep;
        for (const dist_id_t& cur : cand) {
            if (keep.size() >= Mlim) break;
            bool good = true;
            for (const dist_id_t& k : keep) if (dist(vec(k.second), vec(cur.second)) < cur.first) { good = false; break; }
            if (good) keep.push_back(cur);
        }
        return keep;
    }
    // addPoint's linking part for a row that is already placed (data, level, empty lists)
    void linkExisting(tableint cur_c) {
        const int curlevel = levels[cur_c];
        int maxlevelcopy = maxlevel;
        tableint currObj = enterpoint;
        const float* q = vec(cur_c);
        if ((int32_t)currObj != -1) {
            if (curlevel < maxlevelcopy) {
                float curdist = dist(q, vec(currObj));
                for (int level = maxlevelcopy; level > curlevel; level--) {
                    bool changed = true;
                    while (changed) {
                        changed = false;
                        const std::vector<tableint>& nb = list_of(currObj, level);
                        for (size_t i = 0; i < nb.size(); i++) { float d = dist(q, vec(nb[i])); if (d < curdist) { curdist = d; currObj = nb[i]; changed = true; } }
                    }
                }
            }
            for (int level = std::min(curlevel, maxlevelcopy); level >= 0; level--) {
                heap_t top_candidates = searchBaseLayer(currObj, q, level);
                currObj = mutuallyConnectNewElement(q, cur_c, top_candidates, level, currObj);
            }
        } else { enterpoint = cur_c; maxlevel = curlevel; }
        if (curlevel > maxlevelcopy) { enterpoint = cur_c; maxlevel = curlevel; }
    }
    // greedy descent from the entry point down to (not into) `stop_above`, then the beam of that layer: closest first
    std::vector<dist_id_t> beamClosestFirst(tableint c, int layer) {
        const float* q = vec(c);
        tableint currObj = enterpoint;
        float curdist = dist(q, vec(currObj));
        for (int level = maxlevel; level > layer; level--) {
            bool changed = true;
            while (changed) {
                changed = false;
                const std::vector<tableint>& nb = list_of(currObj, level);
                for (size_t i = 0; i < nb.size(); i++) { float d = dist(q, vec(nb[i])); if (d < curdist) { curdist = d; currObj = nb[i]; changed = true; } }
            }
        }
        heap_t top = searchBaseLayer(currObj, q, layer);
        std::vector<dist_id_t> beam;
        while (!top.empty()) { beam.push_back(top.top()); top.pop(); }
        std::reverse(beam.begin(), beam.end());                                  // closest first (the order searchKnnCloserFirst hands out)
        return beam;
    }
    // one round of links on `layer`: the new rows' lists = their choices, then every asked node answers all its requests at once
    void linkRound(int layer, const std::vector<tableint>& rows, const std::vector<std::vector<dist_id_t>>& chosen) {
        const size_t TCAP = 256, cap = layer ? maxM : maxM0;
        std::map<tableint, std::vector<dist_id_t>> asked;                       // node -> requests (distance, new row)
        for (size_t r = 0; r < rows.size(); r++) {
            std::vector<tableint>& own = list_of(rows[r], layer);
            own.clear();
            for (const dist_id_t& k : chosen[r]) { own.push_back(k.second); asked[k.second].emplace_back(k.first, rows[r]); }
        }
        for (auto& kv : asked) {
            const tableint s = kv.first;
            std::vector<dist_id_t>& req = kv.second;
            std::sort(req.begin(), req.end());                                   // (distance, id)
            std::vector<tableint>& lst = list_of(s, layer);
            if (lst.size() + req.size() <= cap) { for (const dist_id_t& r : req) lst.push_back(r.second); continue; }
            if (req.size() > TCAP - lst.size()) req.resize(TCAP - lst.size());
            std::vector<dist_id_t> cand;
            for (tableint x : lst) cand.emplace_back(dist(vec(x), vec(s)), x);
            cand.insert(cand.end(), req.begin(), req.end());
            std::sort(cand.begin(), cand.end());
            const std::vector<dist_id_t> keep = selectClosestFirst(cand, cap);
            lst.clear();
            for (const dist_id_t& k : keep) lst.push_back(k.second);
        }
    }
    void bulk_build(const float* rows, const uint64_t* row_labels, size_t n, size_t seed_min, size_t max_batch) {
        data.assign(rows, rows + n * dim);
        labels.assign(row_labels, row_labels + n);
        for (size_t i = 0; i < n; i++) label_lookup[labels[i]] = (tableint)i;
        deleted.assign(n, 0);
        levels.resize(n);
        link0.assign(n, {});
        linkU.resize(n);
        for (size_t i = 0; i < n; i++) { levels[i] = getRandomLevel(mult); linkU[i].assign((size_t)levels[i], {}); }
        int seed_top = 0;
        for (size_t i = 0; i < n; i++) if (levels[i] >= 2 || i < seed_min) seed_top = std::max(seed_top, levels[i]);
        const int host_from = seed_top >= 1 ? 2 : 1;
        std::vector<tableint> rest;
        size_t linked = 0;
        for (size_t i = 0; i < n; i++) {
            if (levels[i] >= host_from || i < seed_min) { linkExisting((tableint)i); linked++; }
            else rest.push_back((tableint)i);
        }
        size_t pos = 0;
        while (pos < rest.size()) {
            const size_t b = std::min(std::min(max_batch, std::max<size_t>(64, linked / 16)), rest.size() - pos);
            std::vector<tableint> rows0(rest.begin() + pos, rest.begin() + pos + b), rows1;
            for (tableint c : rows0) if (levels[c] == 1) rows1.push_back(c);
            std::vector<std::vector<dist_id_t>> chosen0(rows0.size()), chosen1(rows1.size());      // the beams first, on the graph as it stands; the links after both
            for (size_t r = 0; r < rows1.size(); r++) chosen1[r] = selectClosestFirst(beamClosestFirst(rows1[r], 1), M);
            for (size_t r = 0; r < rows0.size(); r++) chosen0[r] = selectClosestFirst(beamClosestFirst(rows0[r], 0), M);
            linkRound(1, rows1, chosen1);
            linkRound(0, rows0, chosen0);
            pos += b; linked += b;
        }
    }

    // markDelete(label) -> markDeletedInternal: with allow_replace_deleted the slot becomes available to a later addPoint
    bool markDelete(uint64_t label) {
        auto it = label_lookup.find(label);
        if (it == label_lookup.end() || deleted[it->second]) return false;
        deleted[it->second] = 1;
        deleted_order.push_back(it->second);
        return true;
    }
    // addPoint(data, label, replace_deleted = true) as Typesense calls it (src/index.cpp:1052-1054; index built with allow_replace_deleted = true,
    // include/index.h:367): a live label is updated in place (hnswlib's inner addPoint -> updatePoint); otherwise a vacant (deleted) slot is re-used —
    // setExternalLabel, label_lookup_ moved, unmarkDeletedInternal, updatePoint(data, slot, 1.0) — and only without one a new element is appended.
    // (Deviation, stated: hnswlib looks for the vacant slot BEFORE it looks the label up, so a LIVE label next to a vacant slot gets a second slot there;
    //  Typesense never does that — an update is remove + add — and this restatement updates the live slot. Returns the internal id.)
    tableint addPointReplace(const float* v, uint64_t label) {
        auto it = label_lookup.find(label);
        if (it != label_lookup.end() && !deleted[it->second]) { updatePoint(v, it->second); return it->second; }
        while (!deleted_order.empty()) {
            const tableint slot = deleted_order.back();
            deleted_order.pop_back();
            if (!deleted[slot]) continue;
            const uint64_t label_replaced = labels[slot];
            auto old = label_lookup.find(label_replaced);
            if (old != label_lookup.end() && old->second == slot) label_lookup.erase(old);
            labels[slot] = label;
            label_lookup[label] = slot;
            deleted[slot] = 0;
            updatePoint(v, slot);
            return slot;
        }
        addPoint(v, label);
        return (tableint)size() - 1;
    }
    // updatePoint(dataPoint, internalId, updateNeighborProbability = 1.0). hnswlib walks two unordered_sets (sCand, sNeigh); the order matters for
    // candidates at EQUAL distance only and is fixed here as ascending ids (as in the library's builder).
    void updatePoint(const float* v, tableint internalId) {
        std::copy(v, v + dim, data.begin() + (size_t)internalId * dim);
        const int maxLevelCopy = maxlevel;
        const tableint entryPointCopy = enterpoint;
        if (entryPointCopy == internalId && size() == 1) return;
        const int elemLevel = levels[internalId];
        for (int layer = 0; layer <= elemLevel; layer++) {
            const std::vector<tableint> listOneHop = list_of(internalId, layer);
            if (listOneHop.empty()) continue;
            std::vector<tableint> sCand{internalId}, sNeigh;
            for (tableint elOneHop : listOneHop) {
                sCand.push_back(elOneHop);
                sNeigh.push_back(elOneHop);
                for (tableint elTwoHop : list_of(elOneHop, layer)) sCand.push_back(elTwoHop);
            }
            std::sort(sCand.begin(), sCand.end()); sCand.erase(std::unique(sCand.begin(), sCand.end()), sCand.end());
            std::sort(sNeigh.begin(), sNeigh.end()); sNeigh.erase(std::unique(sNeigh.begin(), sNeigh.end()), sNeigh.end());
            for (tableint neigh : sNeigh) {
                heap_t candidates;
                const size_t size_ = std::binary_search(sCand.begin(), sCand.end(), neigh) ? sCand.size() - 1 : sCand.size();
                const size_t elementsToKeep = std::min(ef_construction, size_);
                for (tableint cand : sCand) {
                    if (cand == neigh) continue;
                    const float distance = dist(vec(neigh), vec(cand));
                    if (candidates.size() < elementsToKeep) candidates.emplace(distance, cand);
                    else if (distance < candidates.top().first) { candidates.pop(); candidates.emplace(distance, cand); }
                }
                getNeighborsByHeuristic2(candidates, layer == 0 ? maxM0 : maxM);
                std::vector<tableint>& ll = list_of(neigh, layer);
                ll.clear();
                while (candidates.size() > 0) { ll.push_back(candidates.top().second); candidates.pop(); }
            }
        }
        repairConnectionsForUpdate(vec(internalId), entryPointCopy, internalId, elemLevel, maxLevelCopy);
    }
    void repairConnectionsForUpdate(const float* dataPoint, tableint entryPointInternalId, tableint dataPointInternalId, int dataPointLevel, int maxLevel) {
        tableint currObj = entryPointInternalId;
        if (dataPointLevel < maxLevel) {
            float curdist = dist(dataPoint, vec(currObj));
            for (int level = maxLevel; level > dataPointLevel; level--) {
                bool changed = true;
                while (changed) {
                    changed = false;
                    const std::vector<tableint> nb = list_of(currObj, level);
                    for (tableint cand : nb) {
                        const float d = dist(dataPoint, vec(cand));
                        if (d < curdist) { curdist = d; currObj = cand; changed = true; }
                    }
                }
            }
        }
        for (int level = std::min(dataPointLevel, maxLevel); level >= 0; level--) {
            heap_t topCandidates = searchBaseLayer(currObj, dataPoint, level);
            heap_t filteredTopCandidates;
            while (topCandidates.size() > 0) {
                if (topCandidates.top().second != dataPointInternalId) filteredTopCandidates.push(topCandidates.top());
                topCandidates.pop();
            }
            // "there could be cases where topCandidates could just contain the entry point itself. To prevent self loops ... can be empty"
            if (filteredTopCandidates.size() > 0) {
                if (deleted[entryPointInternalId]) {
                    filteredTopCandidates.emplace(dist(dataPoint, vec(entryPointInternalId)), entryPointInternalId);
                    if (filteredTopCandidates.size() > ef_construction) filteredTopCandidates.pop();
                }
                currObj = mutuallyConnectNewElement(dataPoint, dataPointInternalId, filteredTopCandidates, level, currObj, true);
            }
        }
    }

    // searchBaseLayerST<has_deletions, ...>(ep, q, ef, isIdAllowed): allow == nullptr = every row allowed
    // (tags: hnswlib's pooled VisitedList — one 16-bit tag per row and an epoch instead of a fresh zeroed array per query; optional)
    struct visited_tags_t { std::vector<uint16_t> tag; uint16_t epoch = 0; };
    heap_t searchBaseLayerST(tableint ep_id, const float* q, size_t ef, const uint8_t* allow, bool has_deletions, uint64_t* n_dist = nullptr,
                             visited_tags_t* tags = nullptr) const {
        std::vector<uint8_t> visited_own;
        if (!tags) visited_own.assign(size(), 0);
        else {
            if (tags->tag.size() != size()) { tags->tag.assign(size(), 0); tags->epoch = 0; }
            if (++tags->epoch == 0) { std::fill(tags->tag.begin(), tags->tag.end(), 0); tags->epoch = 1; }
        }
        struct visited_view { uint8_t* own; uint16_t* tag; uint16_t epoch;
                              bool test_and_set(tableint i) { if (own) { bool v = own[i]; own[i] = 1; return v; } bool v = tag[i] == epoch; tag[i] = epoch; return v; } };
        visited_view visited{tags ? nullptr : visited_own.data(), tags ? tags->tag.data() : nullptr, tags ? tags->epoch : (uint16_t)0};
        heap_t top_candidates, candidate_set;
        float lowerBound;
        auto ok = [&](tableint i) { return (!has_deletions || !deleted[i]) && (!allow || allow[i]); };
        if (ok(ep_id)) {
            float d = dist(q, vec(ep_id));
            lowerBound = d;
            top_candidates.emplace(d, ep_id);
            candidate_set.emplace(-d, ep_id);
        } else {
            lowerBound = std::numeric_limits<float>::max();
            candidate_set.emplace(-lowerBound, ep_id);
        }
        visited.test_and_set(ep_id);
        while (!candidate_set.empty()) {
            dist_id_t cur = candidate_set.top();
            if ((-cur.first) > lowerBound && (top_candidates.size() == ef || (!allow && !has_deletions))) break;
            candidate_set.pop();
            const std::vector<tableint>& nb = link0[cur.second];
            for (size_t j = 0; j < nb.size(); j++) {
                tableint c = nb[j];
                if (visited.test_and_set(c)) continue;
                float d = dist(q, vec(c));
                if (n_dist) (*n_dist)++;
                if (top_candidates.size() < ef || lowerBound > d) {
                    candidate_set.emplace(-d, c);
                    if (ok(c)) top_candidates.emplace(d, c);
                    if (top_candidates.size() > ef) top_candidates.pop();
                    if (!top_candidates.empty()) lowerBound = top_candidates.top().first;
                }
            }
        }
        return top_candidates;
    }

    // searchKnnCloserFirst(q, k, ef, filter) of the Typesense fork: ef = max(ef, k) candidates at layer 0, closest first
    std::vector<std::pair<float, uint64_t>> searchKnnCloserFirst(const float* q, size_t k, size_t ef, const uint8_t* allow, uint64_t* n_dist = nullptr,
                                                                 visited_tags_t* tags = nullptr, int has_deletions_known = -1) const {
        std::vector<std::pair<float, uint64_t>> result;
        if (size() == 0) return result;
        tableint currObj = enterpoint;
        float curdist = dist(q, vec(enterpoint));
        for (int level = maxlevel; level > 0; level--) {
            bool changed = true;
            while (changed) {
                changed = false;
                const std::vector<tableint>& nb = linkU[currObj][level - 1];
                for (size_t i = 0; i < nb.size(); i++) {
                    float d = dist(q, vec(nb[i]));
                    if (n_dist) (*n_dist)++;
                    if (d < curdist) { curdist = d; currObj = nb[i]; changed = true; }
                }
            }
        }
        bool has_deletions = has_deletions_known > 0;
        if (has_deletions_known < 0) for (uint8_t dlt : deleted) if (dlt) { has_deletions = true; break; }
        heap_t top = searchBaseLayerST(currObj, q, std::max(ef, k), allow, has_deletions, n_dist, tags);
        while (top.size() > k) top.pop();
        result.resize(top.size());
        size_t sz = top.size();
        while (!top.empty()) { result[--sz] = {top.top().first, labels[top.top().second]}; top.pop(); }
        return result;
    }
};

}  // namespace oracle
