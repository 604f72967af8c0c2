// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/for_codec.h header).
// Restates the per-token OR-of-fields iterator and the conjunctive (AND) driver:
//   or_iterator_t           : /root/reference/include/or_iterator.h:10-59, src/or_iterator.cpp:95-171,274-303
//   equals/advance_* helpers: src/or_iterator.cpp:4-93
//   take_id                 : src/or_iterator.cpp:218-272 (filter-id-array + excluded-ids form)
//   intersect<T>            : include/or_iterator.h:61-182
// The deadline check (or_iterator.h:148-153) is restated with an explicit deadline object instead
// of the thread-locals of include/thread_local_vars.h:7-9.
#pragma once
#include <vector>
#include <chrono>
#include "postings.h"

namespace oracle {

struct deadline_t {
    uint64_t search_begin_us = 0;
    uint64_t search_stop_us = UINT64_MAX;
    bool search_cutoff = false;
    static uint64_t now_us() {
        return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(
                   std::chrono::system_clock::now().time_since_epoch()).count();
    }
    bool expired() const { return search_stop_us != UINT64_MAX && (now_us() - search_begin_us) > search_stop_us; }
};

struct single_filter_result_t { uint32_t seq_id = 0; };

class or_iterator_t {
    std::vector<posting_list_t::iterator_t> its;
    int curr_index = 0;

    void advance_smallest() {  // or_iterator.cpp:120-144
        auto smallest = its[curr_index].id();
        curr_index = 0;
        for (int i = 0; i < int(its.size()); i++) {
            if (its[i].id() == smallest) its[i].next();
            if (!its[i].valid()) { its[i].reset_cache(); its.erase(its.begin() + i); i--; }
        }
        uint32_t new_smallest = UINT32_MAX;
        for (int i = 0; i < int(its.size()); i++) {
            if (its[i].id() < new_smallest) { curr_index = i; new_smallest = its[i].id(); }
        }
    }

public:
    explicit or_iterator_t(std::vector<posting_list_t::iterator_t>& src) : its(std::move(src)) {  // :274-282
        curr_index = 0;
        for (size_t i = 1; i < its.size(); i++) if (its[i].id() < its[curr_index].id()) curr_index = (int)i;
    }
    or_iterator_t(or_iterator_t&& r) noexcept : its(std::move(r.its)), curr_index(r.curr_index) {}
    or_iterator_t& operator=(or_iterator_t&& r) noexcept { its = std::move(r.its); curr_index = r.curr_index; return *this; }
    ~or_iterator_t() { for (auto& it : its) it.reset_cache(); }

    bool valid() const { return !its.empty(); }
    uint32_t id() const { return its[curr_index].id(); }
    const std::vector<posting_list_t::iterator_t>& get_its() const { return its; }

    bool next() {  // :99-118
        switch (its.size()) {
            case 0: break;
            case 2: if (!posting_list_t::all_ended2(its)) advance_smallest(); break;
            default: if (!posting_list_t::all_ended(its)) advance_smallest(); break;
        }
        return !its.empty();
    }

    bool skip_to(uint32_t id) {  // :146-167
        auto current_value = UINT32_MAX;
        curr_index = 0;
        for (size_t i = 0; i < its.size(); i++) {
            auto& it = its[i];
            it.skip_to(id);
            if (!it.valid()) { its[i].reset_cache(); its.erase(its.begin() + i); i--; }
            else if (it.id() < current_value) { curr_index = (int)i; current_value = it.id(); }
        }
        return !its.empty();
    }

    static bool at_end(const std::vector<or_iterator_t>& v) { for (const auto& it : v) if (!it.valid()) return true; return false; }
    static bool at_end2(const std::vector<or_iterator_t>& v) { return !v[0].valid() || !v[1].valid(); }
    static bool equals(std::vector<or_iterator_t>& v) {
        for (int i = 0; i < int(v.size()) - 1; i++) if (v[i].id() != v[i + 1].id()) return false;
        return true;
    }
    static bool equals2(std::vector<or_iterator_t>& v) { return v[0].id() == v[1].id(); }

    static void advance_all(std::vector<or_iterator_t>& v) {  // :35-43 (no index fix-up after erase, as there)
        for (size_t i = 0; i < v.size(); i++) {
            bool ok = v[i].next();
            if (!ok) v.erase(v.begin() + i);
        }
    }
    static void advance_all2(std::vector<or_iterator_t>& v) {  // :45-58
        bool v0 = v[0].next();
        bool v1 = v[1].next();
        if (!v0) { v.erase(v.begin()); if (!v1) v.erase(v.begin()); }
        else if (!v1) v.erase(v.begin() + 1);
    }
    static void advance_non_largest(std::vector<or_iterator_t>& v) {  // :60-79
        uint32_t g = 0;
        for (size_t i = 0; i < v.size(); i++) if (v[i].id() > g) g = v[i].id();
        for (size_t i = 0; i < v.size(); i++) {
            if (v[i].id() != g) {
                bool ok = v[i].skip_to(g);
                if (!ok) { v.erase(v.begin() + i); i--; }
            }
        }
    }
    static void advance_non_largest2(std::vector<or_iterator_t>& v) {  // :81-93
        if (v[0].id() > v[1].id()) { if (!v[1].skip_to(v[0].id())) v.erase(v.begin() + 1); }
        else { if (!v[0].skip_to(v[1].id())) v.erase(v.begin()); }
    }

    static bool take_id(result_iter_state_t& istate, uint32_t id, bool& is_excluded, single_filter_result_t& fr) {  // :218-272
        is_excluded = false;
        if (istate.excluded_result_ids_size != 0 &&
            std::binary_search(istate.excluded_result_ids, istate.excluded_result_ids + istate.excluded_result_ids_size, id)) {
            is_excluded = true;
            return false;
        }
        if (istate.filter_ids_length != 0) {
            if (istate.filter_ids_index >= istate.filter_ids_length) return false;
            size_t found = std::lower_bound(istate.filter_ids + istate.filter_ids_index,
                                            istate.filter_ids + istate.filter_ids_length, id) - istate.filter_ids;
            if (found == istate.filter_ids_length) { istate.filter_ids_index = found + 1; return false; }
            if (istate.filter_ids[found] == id) { fr.seq_id = id; istate.filter_ids_index = found + 1; return true; }
            istate.filter_ids_index = found;
            return false;
        }
        fr.seq_id = id;
        return true;
    }

    template <class T>
    static bool intersect(std::vector<or_iterator_t>& its, result_iter_state_t& istate, deadline_t& dl, T func) {  // or_iterator.h:61-182
        size_t it_size = its.size();
        bool is_excluded;
        size_t num_processed = 0;
        auto skip_all = [&](uint32_t id) { for (auto& it : its) it.skip_to(id); };

        switch (its.size()) {
            case 0: break;
            case 1:
                if (istate.is_filter_provided() && istate.is_filter_valid()) its[0].skip_to(istate.get_filter_id());
                while (its.size() == it_size && its[0].valid()) {
                    num_processed++;
                    if (num_processed % 65536 == 0 && dl.expired()) { dl.search_cutoff = true; break; }
                    auto id = its[0].id();
                    istate.num_keyword_matches++;
                    single_filter_result_t fr;
                    if (take_id(istate, id, is_excluded, fr)) func(fr, its);
                    if (istate.is_filter_provided() && !is_excluded) {
                        if (istate.is_filter_valid()) its[0].skip_to(istate.get_filter_id()); else break;
                    } else its[0].next();
                }
                break;
            case 2:
                if (istate.is_filter_provided() && istate.is_filter_valid()) { its[0].skip_to(istate.get_filter_id()); its[1].skip_to(istate.get_filter_id()); }
                while (its.size() == it_size && !at_end2(its)) {
                    num_processed++;
                    if (num_processed % 65536 == 0 && dl.expired()) { dl.search_cutoff = true; break; }
                    if (equals2(its)) {
                        auto id = its[0].id();
                        istate.num_keyword_matches++;
                        single_filter_result_t fr;
                        if (take_id(istate, id, is_excluded, fr)) func(fr, its);
                        if (istate.is_filter_provided() && !is_excluded) {
                            if (istate.is_filter_valid()) { its[0].skip_to(istate.get_filter_id()); its[1].skip_to(istate.get_filter_id()); }
                            else break;
                        } else advance_all2(its);
                    } else advance_non_largest2(its);
                }
                break;
            default:
                if (istate.is_filter_provided() && istate.is_filter_valid()) skip_all(istate.get_filter_id());
                while (its.size() == it_size && !at_end(its)) {
                    num_processed++;
                    if (num_processed % 65536 == 0 && dl.expired()) { dl.search_cutoff = true; break; }
                    if (equals(its)) {
                        auto id = its[0].id();
                        istate.num_keyword_matches++;
                        single_filter_result_t fr;
                        if (take_id(istate, id, is_excluded, fr)) func(fr, its);
                        if (istate.is_filter_provided() && !is_excluded) {
                            if (istate.is_filter_valid()) skip_all(istate.get_filter_id()); else break;
                        } else advance_all(its);
                    } else advance_non_largest(its);
                }
        }
        return true;
    }
};

}  // namespace oracle
