// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/for_codec.h header).
// Restates the reference's blocked, FOR-compressed posting list and its iterator:
//   block_t / iterator_t / posting_list_t : /root/reference/include/posting_list.h:56-239
//   block_t::upsert                       : src/posting_list.cpp:9-134
//   insert_and_shift_offset_index         : src/posting_list.cpp:192-209
//   split_block                           : src/posting_list.cpp:374-446
//   posting_list_t::upsert                : src/posting_list.cpp:448-498
//   merge / intersect / take_id           : src/posting_list.cpp:638-809, block_intersect posting_list.h:241-309
//   get_offsets / verbatim / last_offset  : src/posting_list.cpp:832-959, 1899-1951
//   leap-frog helpers                     : src/posting_list.cpp:961-1073
//   iterator_t ctor/next/skip_to/reset    : src/posting_list.cpp:1955-2061, 2101-2109
//   compact_posting_list_t                : include/posting.h:14-46, src/posting.cpp:7-176
// Data-structure costs are kept on purpose (std::map skip index, 3 heap allocations + 3 FOR
// decodes per visited block, linear in-block skip) because this file is also the CPU baseline.
// Not restated: erase / merge_adjacent_blocks (write path, out of scope — SURVEY §2b).
#pragma once
#include <map>
#include <vector>
#include <stdexcept>
#include <cstdlib>
#include "for_arrays.h"
#include "match_window.h"

namespace oracle {

typedef uint32_t last_id_t;

struct result_iter_state_t {  // include/posting_list.h:13-44 (filter_result_iterator_t* path not restated)
    const uint32_t* excluded_result_ids = nullptr;
    size_t excluded_result_ids_size = 0;
    const uint32_t* filter_ids = nullptr;
    size_t filter_ids_length = 0;
    size_t excluded_result_ids_index = 0;
    size_t filter_ids_index = 0;
    size_t num_keyword_matches = 0;

    result_iter_state_t() = default;
    result_iter_state_t(const uint32_t* ex, size_t nex, const uint32_t* fi, size_t nfi)
        : excluded_result_ids(ex), excluded_result_ids_size(nex), filter_ids(fi), filter_ids_length(nfi) {}

    bool is_filter_provided() const { return filter_ids_length > 0; }                 // posting_list.cpp:2172-2174
    bool is_filter_valid() const { return filter_ids_length > 0 && filter_ids_index < filter_ids_length; }  // :2176-2186
    uint32_t get_filter_id() const {                                                  // :2188-2198
        if (filter_ids_length > 0 && filter_ids_index < filter_ids_length) return filter_ids[filter_ids_index];
        return 0;
    }
};

class posting_list_t {
public:
    struct block_t {
        sorted_array ids;
        sorted_array offset_index;
        array offsets;
        block_t* next = nullptr;

        uint32_t size() const { return ids.getLength(); }
        bool contains(uint32_t id) const { return ids.contains(id); }

        void insert_and_shift_offset_index(uint32_t index, uint32_t num_offsets) {  // posting_list.cpp:192-209
            uint32_t existing = offset_index.at(index);
            uint32_t length = offset_index.getLength();
            uint32_t new_length = length + 1;
            uint32_t* cur = offset_index.uncompress(new_length);
            memmove(&cur[index + 1], &cur[index], sizeof(uint32_t) * (length - index));
            cur[index] = existing;
            for (uint32_t i = index + 1; i < new_length; i++) cur[i] += num_offsets;
            offset_index.load(cur, new_length);
            delete[] cur;
        }

        uint32_t upsert(uint32_t id, const std::vector<uint32_t>& positions) {  // posting_list.cpp:9-134
            if (id > ids.last() || ids.getLength() == 0) {
                ids.append(id);
                uint32_t curr_index = offsets.getLength();
                offset_index.append(curr_index);
                for (uint32_t p : positions) offsets.append(p);
                return 1;
            }
            uint32_t id_index = ids.indexOf(id);
            if (id_index == ids.getLength()) {
                size_t inserted_index = ids.append(id);
                uint32_t existing_offset_index = offset_index.at((uint32_t)inserted_index);
                insert_and_shift_offset_index((uint32_t)inserted_index, (uint32_t)positions.size());
                offsets.insert(existing_offset_index, positions.data(), positions.size());
                return 1;
            }
            // id present: replace its offsets
            uint32_t start = offset_index.at(id_index);
            uint32_t end = (id == ids.last()) ? offsets.getLength() - 1 : offset_index.at(id_index + 1) - 1;
            uint32_t num_offsets = (end - start) + 1;
            uint32_t* cur = offsets.uncompress();
            std::vector<uint32_t> nw(cur, cur + start);
            nw.insert(nw.end(), positions.begin(), positions.end());
            nw.insert(nw.end(), cur + end + 1, cur + offsets.getLength());
            delete[] cur;
            uint32_t m = nw.empty() ? 0 : nw[0], M = m;
            for (uint32_t v : nw) { m = std::min(m, v); M = std::max(M, v); }
            int64_t size_diff = int64_t(positions.size()) - num_offsets;
            offsets.load(nw.data(), (uint32_t)nw.size(), m, M);
            if (size_diff != 0) {
                uint32_t* oi = offset_index.uncompress();
                for (size_t i = id_index + 1; i < ids.getLength(); i++) oi[i] = (uint32_t)(oi[i] + size_diff);
                offset_index.load(oi, offset_index.getLength());
                delete[] oi;
            }
            return 0;
        }
    };

    class iterator_t {
        const std::map<last_id_t, block_t*>* id_block_map;
        block_t* curr_block;
        uint32_t curr_index;
        block_t* end_block;
        bool auto_destroy;
        uint32_t field_id;

    public:
        uint32_t* ids = nullptr;
        uint32_t* offset_index = nullptr;
        uint32_t* offsets = nullptr;

        iterator_t(const std::map<last_id_t, block_t*>* m, block_t* start, block_t* end,
                   bool auto_destroy = true, uint32_t field_id = 0)  // posting_list.cpp:1955-1971
            : id_block_map(m), curr_block(start), curr_index(0), end_block(end),
              auto_destroy(auto_destroy), field_id(field_id) {
            if (curr_block != end_block) {
                ids = curr_block->ids.uncompress();
                offset_index = curr_block->offset_index.uncompress();
                offsets = curr_block->offsets.uncompress();
            }
        }
        ~iterator_t() { if (auto_destroy) reset_cache(); }
        iterator_t(const iterator_t&) = delete;
        iterator_t& operator=(const iterator_t&) = delete;
        iterator_t(iterator_t&& r) noexcept { move_from(r); }
        iterator_t& operator=(iterator_t&& r) noexcept { move_from(r); return *this; }

        void move_from(iterator_t& r) {  // posting_list.cpp:2111-2149 (no free of the overwritten cache, as there)
            id_block_map = r.id_block_map; curr_block = r.curr_block; curr_index = r.curr_index;
            end_block = r.end_block; ids = r.ids; offset_index = r.offset_index; offsets = r.offsets;
            auto_destroy = r.auto_destroy; field_id = r.field_id;
            r.id_block_map = nullptr; r.curr_block = nullptr; r.end_block = nullptr;
            r.ids = nullptr; r.offset_index = nullptr; r.offsets = nullptr;
        }

        void reset_cache() {  // :2101-2109
            delete[] ids; delete[] offsets; delete[] offset_index;
            ids = offset_index = offsets = nullptr;
            curr_index = 0;
            curr_block = end_block = nullptr;
        }

        bool valid() const { return (curr_block != end_block) && (curr_index < curr_block->size()); }  // :1973-1975

        void next() {  // :1977-1995
            curr_index++;
            if (curr_index == curr_block->size()) {
                curr_index = 0;
                curr_block = curr_block->next;
                delete[] ids; delete[] offset_index; delete[] offsets;
                ids = offset_index = offsets = nullptr;
                if (curr_block != end_block) {
                    ids = curr_block->ids.uncompress();
                    offset_index = curr_block->offset_index.uncompress();
                    offsets = curr_block->offsets.uncompress();
                }
            }
        }

        uint32_t last_block_id() const { auto s = curr_block->size(); return s == 0 ? 0 : ids[s - 1]; }
        uint32_t id() const { return ids[curr_index]; }
        uint32_t index() const { return curr_index; }
        block_t* block() const { return curr_block; }
        uint32_t get_field_id() const { return field_id; }

        void skip_to(uint32_t id) {  // :2030-2061
            if (id <= this->last_block_id()) {
                while (curr_index < curr_block->size() && this->id() < id) curr_index++;
                return;
            }
            reset_cache();
            const auto it = id_block_map->lower_bound(id);
            if (it == id_block_map->end()) return;
            curr_block = it->second;
            curr_index = 0;
            ids = curr_block->ids.uncompress();
            offset_index = curr_block->offset_index.uncompress();
            offsets = curr_block->offsets.uncompress();
            while (curr_index < curr_block->size() && this->id() < id) curr_index++;
            if (curr_index == curr_block->size()) reset_cache();
        }

        iterator_t clone() const {  // :2155-2167 (shares the decoded arrays, never frees them)
            iterator_t it(nullptr, nullptr, nullptr);
            it.id_block_map = id_block_map; it.curr_block = curr_block; it.curr_index = curr_index;
            it.end_block = end_block; it.ids = ids; it.offsets = offsets; it.offset_index = offset_index;
            it.auto_destroy = false; it.field_id = field_id;
            return it;
        }
    };

    const uint16_t BLOCK_MAX_ELEMENTS;
    uint32_t ids_length = 0;
    block_t root_block;
    std::map<last_id_t, block_t*> id_block_map;

    explicit posting_list_t(uint16_t max_block_elements) : BLOCK_MAX_ELEMENTS(max_block_elements) {
        if (max_block_elements <= 1) throw std::invalid_argument("max_block_elements must be > 1");
    }
    ~posting_list_t() {
        block_t* b = root_block.next;
        while (b != nullptr) { block_t* n = b->next; delete b; b = n; }
    }
    posting_list_t(const posting_list_t&) = delete;

    static void split_block(block_t* src, block_t* dst) {  // posting_list.cpp:374-446
        if (src->size() <= 1) return;
        uint32_t* raw_ids = src->ids.uncompress();
        size_t first = src->size() / 2, second = src->size() - first;
        uint32_t* raw_oi = src->offset_index.uncompress();
        size_t oi_len = src->offset_index.getLength();
        size_t oi_first = oi_len / 2, oi_second = oi_len - oi_first;
        uint32_t* raw_off = src->offsets.uncompress();
        size_t off_len = src->offsets.getLength();

        src->ids.load(raw_ids, (uint32_t)first);
        dst->ids.load(raw_ids + first, (uint32_t)second);
        src->offset_index.load(raw_oi, (uint32_t)oi_first);
        uint32_t base_diff = raw_oi[oi_first];
        for (size_t i = 0; i < oi_second; i++) raw_oi[oi_first + i] -= base_diff;
        dst->offset_index.load(raw_oi + oi_first, (uint32_t)oi_second);

        size_t off_first = base_diff;
        uint32_t mn = raw_off[0], mx = raw_off[0];
        for (size_t i = 0; i < off_first; i++) { mn = std::min(mn, raw_off[i]); mx = std::max(mx, raw_off[i]); }
        src->offsets.load(raw_off, (uint32_t)off_first, mn, mx);
        mn = mx = raw_off[off_first];
        for (size_t i = off_first; i < off_len; i++) { mn = std::min(mn, raw_off[i]); mx = std::max(mx, raw_off[i]); }
        dst->offsets.load(raw_off + off_first, (uint32_t)(off_len - off_first), mn, mx);
        delete[] raw_ids; delete[] raw_oi; delete[] raw_off;
    }

    void upsert(uint32_t id, const std::vector<uint32_t>& offsets) {  // posting_list.cpp:448-498
        block_t* upsert_block;
        last_id_t before_last;
        if (id_block_map.empty()) {
            upsert_block = &root_block;
            before_last = UINT32_MAX;
        } else {
            const auto it = id_block_map.lower_bound(id);
            upsert_block = (it == id_block_map.end()) ? id_block_map.rbegin()->second : it->second;
            before_last = upsert_block->ids.last();
        }
        if (upsert_block->size() < BLOCK_MAX_ELEMENTS) {
            ids_length += upsert_block->upsert(id, offsets);
            last_id_t after_last = upsert_block->ids.last();
            if (before_last != after_last) {
                id_block_map.erase(before_last);
                id_block_map.emplace(after_last, upsert_block);
            }
        } else {
            block_t* new_block = new block_t;
            if (upsert_block->next == nullptr && upsert_block->ids.last() < id) {
                ids_length += new_block->upsert(id, offsets);
            } else {
                ids_length += upsert_block->upsert(id, offsets);
                split_block(upsert_block, new_block);
                last_id_t after_last = upsert_block->ids.last();
                id_block_map.erase(before_last);
                id_block_map.emplace(after_last, upsert_block);
            }
            id_block_map.emplace(new_block->ids.last(), new_block);
            new_block->next = upsert_block->next;
            upsert_block->next = new_block;
        }
    }

    // Bulk construction for ascending ids: produces exactly the structure that |n| sequential
    // upserts of ascending ids produce (blocks fill to BLOCK_MAX_ELEMENTS, then a new block is
    // chained — posting_list.cpp:471-476), without the O(block^2) re-encoding of append().
    void load_sorted(const uint32_t* ids, const uint32_t* offset_index, const uint32_t* offsets,
                     uint32_t n, uint32_t n_offsets) {
        block_t* blk = &root_block;
        for (uint32_t s = 0; s < n; s += BLOCK_MAX_ELEMENTS) {
            uint32_t cnt = std::min<uint32_t>(BLOCK_MAX_ELEMENTS, n - s);
            if (s != 0) { block_t* nb = new block_t; blk->next = nb; blk = nb; }
            uint32_t o0 = offset_index[s];
            uint32_t o1 = (s + cnt == n) ? n_offsets : offset_index[s + cnt];
            std::vector<uint32_t> oi(cnt);
            for (uint32_t i = 0; i < cnt; i++) oi[i] = offset_index[s + i] - o0;
            blk->ids.load(ids + s, cnt);
            blk->offset_index.load(oi.data(), cnt);
            uint32_t mn = offsets[o0], mx = offsets[o0];
            for (uint32_t i = o0; i < o1; i++) { mn = std::min(mn, offsets[i]); mx = std::max(mx, offsets[i]); }
            blk->offsets.load(offsets + o0, o1 - o0, mn, mx);
            id_block_map.emplace(blk->ids.last(), blk);
        }
        ids_length = n;
    }

    block_t* get_root() { return &root_block; }
    size_t num_blocks() const { return id_block_map.size(); }
    size_t num_ids() const { return ids_length; }
    uint32_t first_id() { return ids_length == 0 ? 0 : root_block.ids.at(0); }

    block_t* block_of(uint32_t id) {  // posting_list.cpp:~622-636
        const auto it = id_block_map.lower_bound(id);
        return it == id_block_map.end() ? nullptr : it->second;
    }
    bool contains(uint32_t id) {
        const auto it = id_block_map.lower_bound(id);
        if (it == id_block_map.end()) return false;
        return it->second->contains(id);
    }

    iterator_t new_iterator(block_t* start = nullptr, block_t* end = nullptr, uint32_t field_id = 0) {  // :999-1002
        start = (start == nullptr) ? &root_block : start;
        return iterator_t(&id_block_map, start, end, true, field_id);
    }

    // ---- leap-frog helpers over plain iterators, posting_list.cpp:961-1073 ----
    static bool at_end(const std::vector<iterator_t>& its) { for (const auto& it : its) if (!it.valid()) return true; return false; }
    static bool at_end2(const std::vector<iterator_t>& its) { return !its[0].valid() || !its[1].valid(); }
    static bool all_ended(const std::vector<iterator_t>& its) { for (const auto& it : its) if (it.valid()) return false; return true; }
    static bool all_ended2(const std::vector<iterator_t>& its) { return !its[0].valid() && !its[1].valid(); }
    static bool equals(std::vector<iterator_t>& its) {
        for (int i = 0; i < int(its.size()) - 1; i++) if (its[i].id() != its[i + 1].id()) return false;
        return true;
    }
    static bool equals2(std::vector<iterator_t>& its) { return its[0].id() == its[1].id(); }
    static void advance_all(std::vector<iterator_t>& its) { for (auto& it : its) it.next(); }
    static void advance_all2(std::vector<iterator_t>& its) { its[0].next(); its[1].next(); }
    static void advance_non_largest(std::vector<iterator_t>& its) {
        uint32_t g = 0;
        for (size_t i = 0; i < its.size(); i++) if (its[i].id() > g) g = its[i].id();
        for (size_t i = 0; i < its.size(); i++) if (its[i].id() != g) its[i].skip_to(g);
    }
    static void advance_non_largest2(std::vector<iterator_t>& its) {
        if (its[0].id() > its[1].id()) its[1].skip_to(its[0].id()); else its[0].skip_to(its[1].id());
    }
    static uint32_t advance_smallest(std::vector<iterator_t>& its) {
        uint32_t s = UINT32_MAX;
        for (size_t i = 0; i < its.size(); i++) if (its[i].id() < s) s = its[i].id();
        for (size_t i = 0; i < its.size(); i++) if (its[i].id() == s) its[i].next();
        return s;
    }
    static uint32_t advance_smallest2(std::vector<iterator_t>& its) {
        uint32_t s;
        if (its[0].id() < its[1].id()) { s = its[0].id(); its[0].next(); } else { s = its[1].id(); its[1].next(); }
        return s;
    }

    static void merge(const std::vector<posting_list_t*>& lists, std::vector<uint32_t>& result_ids) {  // :638-705
        std::vector<iterator_t> its;
        its.reserve(lists.size());
        for (auto* pl : lists) its.push_back(pl->new_iterator());
        if (its.size() == 1) {
            auto it = lists[0]->new_iterator();
            while (it.valid()) { result_ids.push_back(it.id()); it.next(); }
            return;
        }
        if (its.size() == 2) {
            while (!at_end2(its)) {
                if (equals2(its)) { result_ids.push_back(its[0].id()); advance_all2(its); }
                else result_ids.push_back(advance_smallest2(its));
            }
            while (its[0].valid()) { result_ids.push_back(its[0].id()); its[0].next(); }
            while (its[1].valid()) { result_ids.push_back(its[1].id()); its[1].next(); }
            return;
        }
        // NB: as in the reference (:687-703) the n>2 branch stops at the first exhausted list and
        // then drains the remaining lists one after the other.
        while (!at_end(its)) {
            if (equals(its)) { result_ids.push_back(its[0].id()); advance_all(its); }
            else result_ids.push_back(advance_smallest(its));
        }
        for (auto& it : its) while (it.valid()) { result_ids.push_back(it.id()); it.next(); }
    }

    static void intersect(const std::vector<posting_list_t*>& lists, std::vector<uint32_t>& result_ids) {  // :708-756
        if (lists.empty()) return;
        if (lists.size() == 1) {
            auto it = lists[0]->new_iterator();
            while (it.valid()) { result_ids.push_back(it.id()); it.next(); }
            return;
        }
        std::vector<iterator_t> its;
        its.reserve(lists.size());
        for (auto* pl : lists) its.push_back(pl->new_iterator());
        if (its.size() == 2) {
            while (!at_end2(its)) {
                if (equals2(its)) { result_ids.push_back(its[0].id()); advance_all2(its); }
                else advance_non_largest2(its);
            }
        } else {
            while (!at_end(its)) {
                if (equals(its)) { result_ids.push_back(its[0].id()); advance_all(its); }
                else advance_non_largest(its);
            }
        }
    }

    static bool take_id(result_iter_state_t& istate, uint32_t id) {  // :794-809
        if (istate.excluded_result_ids_size != 0 &&
            std::binary_search(istate.excluded_result_ids, istate.excluded_result_ids + istate.excluded_result_ids_size, id))
            return false;
        if (istate.filter_ids_length != 0)
            return std::binary_search(istate.filter_ids, istate.filter_ids + istate.filter_ids_length, id);
        return true;
    }

    template <class T>
    static bool block_intersect(std::vector<iterator_t>& its, result_iter_state_t& istate, T func) {  // posting_list.h:241-309
        switch (its.size()) {
            case 0: break;
            case 1:
                while (its[0].valid()) { if (take_id(istate, its[0].id())) func(its[0].id(), its); its[0].next(); }
                break;
            case 2:
                while (!at_end2(its)) {
                    if (equals2(its)) { if (take_id(istate, its[0].id())) func(its[0].id(), its); advance_all2(its); }
                    else advance_non_largest2(its);
                }
                break;
            default:
                while (!at_end(its)) {
                    if (equals(its)) { if (take_id(istate, its[0].id())) func(its[0].id(), its); advance_all(its); }
                    else advance_non_largest(its);
                }
        }
        return false;
    }

    bool contains_atleast_one(const uint32_t* target_ids, size_t n) {  // :~575-620 (semantics only)
        for (size_t i = 0; i < n; i++) if (contains(target_ids[i])) return true;
        return false;
    }

    // ---- per-hit offset decoding ----
    // posting_list.cpp:832-916. Plain string: off1..offn[,0]; array: off1..offn,offn,array_idx[,0]
    static bool get_offsets(const std::vector<iterator_t>& its,
                            std::map<size_t, std::vector<token_positions_t>>& array_token_pos) {
        for (size_t j = 0; j < its.size(); j++) {
            block_t* curr_block = its[j].block();
            uint32_t curr_index = its[j].index();
            if (curr_block == nullptr || curr_index == UINT32_MAX) continue;
            uint32_t* offsets = its[j].offsets;
            uint32_t start_offset = its[j].offset_index[curr_index];
            uint32_t end_offset = (curr_index == curr_block->size() - 1) ? curr_block->offsets.getLength()
                                                                           : its[j].offset_index[curr_index + 1];
            std::vector<uint16_t> positions;
            int prev_pos = -1;
            bool is_last_token = false;
            while (start_offset < end_offset) {
                int pos = offsets[start_offset];
                start_offset++;
                if (pos == 0) { is_last_token = true; start_offset++; continue; }
                if (pos == prev_pos) {
                    if (!positions.empty()) {
                        size_t array_index = (size_t)offsets[start_offset];
                        is_last_token = false;
                        if (start_offset + 1 < end_offset) {
                            size_t next_offset = (size_t)offsets[start_offset + 1];
                            if (next_offset == 0) { is_last_token = true; start_offset++; }
                        }
                        array_token_pos[array_index].push_back(token_positions_t{is_last_token, positions});
                        positions.clear();
                    }
                    start_offset++;
                    prev_pos = -1;
                    continue;
                }
                prev_pos = pos;
                positions.push_back((uint16_t)pos - 1);
            }
            if (!positions.empty()) array_token_pos[0].push_back(token_positions_t{is_last_token, positions});
        }
        return true;
    }

    static bool is_single_token_verbatim_match(const iterator_t& it, bool field_is_array) {  // :918-959
        block_t* curr_block = it.block();
        uint32_t curr_index = it.index();
        if (curr_block == nullptr || curr_index == UINT32_MAX) return false;
        uint32_t* offsets = it.offsets;
        uint32_t start_offset = it.offset_index[curr_index];
        if (!field_is_array && offsets[start_offset] != 1) return false;
        uint32_t end_offset = (curr_index == curr_block->size() - 1) ? curr_block->offsets.getLength()
                                                                       : it.offset_index[curr_index + 1];
        if (field_is_array) {
            int prev_pos = -1;
            while (start_offset < end_offset) {
                int pos = offsets[start_offset];
                start_offset++;
                if (pos == prev_pos && pos == 1 && start_offset + 1 < end_offset && offsets[start_offset + 1] == 0) return true;
                prev_pos = pos;
            }
            return false;
        } else if ((end_offset - start_offset) == 2 && offsets[end_offset - 1] == 0) {
            return true;
        }
        return false;
    }

    static size_t get_last_offset(const iterator_t& it, bool field_is_array) {  // :1899-1951
        block_t* curr_block = it.block();
        uint32_t curr_index = it.index();
        uint32_t* offsets = it.offsets;
        if (curr_block == nullptr || curr_index == UINT32_MAX) return 0;
        uint32_t end_offset = (curr_index == curr_block->size() - 1) ? curr_block->offsets.getLength()
                                                                       : it.offset_index[curr_index + 1];
        if (field_is_array) {
            uint32_t start_offset = it.offset_index[curr_index];
            int prev_pos = -1;
            size_t max_offset = 0;
            while (start_offset < end_offset) {
                int pos = offsets[start_offset];
                start_offset++;
                if ((size_t)pos > max_offset) max_offset = pos;
                if (pos == prev_pos) {
                    if (start_offset + 1 < end_offset) {
                        size_t next_offset = (size_t)offsets[start_offset + 1];
                        if (next_offset == 0) start_offset++;
                    }
                    start_offset++;
                    prev_pos = -1;
                    continue;
                }
                prev_pos = pos;
            }
            return max_offset;
        }
        return offsets[end_offset - 1] == 0 ? offsets[end_offset - 2] : offsets[end_offset - 1];
    }
};

// include/posting.h:14-41, src/posting.cpp:7-176: inline list "[n_off, off.., id] ..." for rare tokens
struct compact_posting_list_t {
    std::vector<uint32_t> id_offsets;
    uint32_t ids_length = 0;

    static compact_posting_list_t* create(uint32_t num_ids, const uint32_t* ids, const uint32_t* offset_index,
                                          uint32_t num_offsets, const uint32_t* offsets) {  // posting.cpp:136-155
        auto* pl = new compact_posting_list_t;
        for (uint32_t i = 0; i < num_ids; i++) {
            uint32_t s = offset_index[i];
            uint32_t e = (i == num_ids - 1) ? num_offsets : offset_index[i + 1];
            pl->upsert(ids[i], offsets + s, e - s);
        }
        return pl;
    }

    void upsert(uint32_t id, const uint32_t* offsets, uint32_t num_offsets) {  // posting.cpp:11-95 (semantics)
        // locate existing entry
        size_t i = 0;
        while (i < id_offsets.size()) {
            size_t n = id_offsets[i];
            uint32_t existing = id_offsets[i + n + 1];
            if (existing == id) {
                id_offsets.erase(id_offsets.begin() + i, id_offsets.begin() + i + n + 2);
                ids_length--;
                break;
            }
            if (existing > id) break;
            i += n + 2;
        }
        std::vector<uint32_t> entry;
        entry.push_back(num_offsets);
        entry.insert(entry.end(), offsets, offsets + num_offsets);
        entry.push_back(id);
        id_offsets.insert(id_offsets.begin() + i, entry.begin(), entry.end());
        ids_length++;
    }

    posting_list_t* to_full_posting_list(uint16_t block_max = 256) const {  // posting.cpp:157-176
        auto* pl = new posting_list_t(block_max);
        size_t i = 0;
        while (i < id_offsets.size()) {
            size_t n = id_offsets[i];
            i++;
            std::vector<uint32_t> offsets(id_offsets.begin() + i, id_offsets.begin() + i + n);
            uint32_t id = id_offsets[i + n];
            pl->upsert(id, offsets);
            i += n + 1;
        }
        return pl;
    }

    uint32_t num_ids() const { return ids_length; }
    uint32_t first_id() const { return id_offsets.empty() ? 0 : id_offsets[id_offsets[0] + 1]; }
    uint32_t last_id() const { return id_offsets.empty() ? UINT32_MAX : id_offsets.back(); }
    bool contains(uint32_t id) const {
        size_t i = 0;
        while (i < id_offsets.size()) {
            size_t n = id_offsets[i];
            if (id_offsets[i + n + 1] == id) return true;
            i += n + 2;
        }
        return false;
    }
};

}  // namespace oracle
