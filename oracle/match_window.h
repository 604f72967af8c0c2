// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/for_codec.h header).
// Restates /root/reference/include/match_score.h:
//   token_positions_t :14-17, TokenOffset :19-35, get_match_score :56-68, sort2/sort3 :79-111,
//   Match(doc, token_positions, populate_window, check_exact_match) :129-275.
// Validated against the real header (compiled into oracle/_ref/libref_match.so by oracle/Makefile)
// on the reference's MatchScoreV2 vectors and on random position lists: tests/test_oracle_match.py.
#pragma once
#include <cstdint>
#include <vector>
#include <algorithm>
#include <limits>

namespace oracle {

static const size_t WINDOW_SIZE = 10;                                           // match_score.h:11
static const uint16_t MAX_DISPLACEMENT = std::numeric_limits<uint16_t>::max();  // :12

struct token_positions_t {
    bool last_token = false;
    std::vector<uint16_t> positions;
};

struct TokenOffset {
    uint8_t token_id = 0;
    uint16_t offset = MAX_DISPLACEMENT;
    uint32_t offset_index = 0;
    bool operator>(const TokenOffset& a) const { return offset > a.offset; }
    bool operator<(const TokenOffset& a) const { return offset < a.offset; }
};

struct Match {
    uint8_t words_present = 0;
    uint8_t distance = 0;
    uint8_t max_offset = 0;
    uint8_t exact_match = 0;
    std::vector<TokenOffset> offsets;

    Match() {}
    Match(uint8_t words_present, uint8_t distance, uint8_t max_offset, uint8_t exact_match = 0)
        : words_present(words_present), distance(distance), max_offset(max_offset), exact_match(exact_match) {}

    uint64_t get_match_score(const uint32_t total_cost, const uint32_t unique_words, const uint8_t synonym_score) const {
        return (uint64_t)((int64_t(words_present) << 40) | (int64_t(unique_words) << 32) |
                          (int64_t(255 - total_cost) << 24) | (int64_t(100 - distance) << 16) |
                          (int64_t(exact_match) << 12) | (int64_t(255 - max_offset) << 4) |
                          (int64_t(synonym_score) << 0));
    }

    // descending order networks with the reference's exact tie behaviour (:79-111)
    static void sort2(std::vector<TokenOffset>& a) { if (a[0] < a[1]) std::swap(a[0], a[1]); }
    static void sort3(std::vector<TokenOffset>& a) {
        if (a[0] > a[1]) {
            if (a[1] > a[2]) return;
            else if (a[0] > a[2]) std::swap(a[1], a[2]);
            else { TokenOffset t = a[0]; a[0] = a[2]; a[2] = a[1]; a[1] = t; }
        } else {
            if (a[0] > a[2]) std::swap(a[0], a[1]);
            else if (a[2] > a[1]) std::swap(a[0], a[2]);
            else { TokenOffset t = a[0]; a[0] = a[1]; a[1] = a[2]; a[2] = t; }
        }
    }

    Match(uint32_t /*doc_id*/, const std::vector<token_positions_t>& token_offsets,
          bool populate_window = true, bool check_exact_match = false) {
        const size_t tokens_size = std::min(token_offsets.size(), WINDOW_SIZE);
        std::vector<TokenOffset> window(tokens_size);
        for (size_t t = 0; t < tokens_size; t++) {
            window[t].token_id = (uint8_t)t;
            window[t].offset = token_offsets[t].positions[0];
            window[t].offset_index = 0;
        }
        std::vector<TokenOffset> best_window;
        if (populate_window) best_window = window;

        size_t best_num_match = 1;
        size_t best_displacement = MAX_DISPLACEMENT;
        int prev_min_offset = -1;

        while (window.size() > 1) {
            switch (window.size()) {
                case 2: sort2(window); break;
                case 3: sort3(window); break;
                default: std::sort(window.begin(), window.end(), std::greater<TokenOffset>());
            }
            size_t min_offset = window.back().offset;
            if (int(min_offset) < prev_min_offset) break;  // offsets wrapped around (uint16 narrowing)
            prev_min_offset = (int)min_offset;

            size_t this_displacement = 0, this_num_match = 0;
            std::vector<TokenOffset> this_window(tokens_size);
            for (size_t i = 0; i < window.size(); i++) {
                if (populate_window) {
                    this_window[window[i].token_id] = window[i];
                    this_window[window[i].token_id].offset = MAX_DISPLACEMENT;
                }
                if ((window[i].offset - min_offset) <= WINDOW_SIZE) {
                    uint16_t next_offset = (i == window.size() - 1) ? window[i].offset : window[i + 1].offset;
                    this_displacement += window[i].offset - next_offset;
                    this_num_match++;
                    if (populate_window) this_window[window[i].token_id].offset = window[i].offset;
                }
            }
            if ((this_num_match > best_num_match) ||
                (this_num_match == best_num_match && this_displacement < best_displacement)) {
                best_displacement = this_displacement;
                best_num_match = this_num_match;
                max_offset = (uint8_t)std::min((uint16_t)255, window.front().offset);
                if (populate_window) best_window = this_window;
            }
            if (best_num_match == tokens_size && best_displacement == (window.size() - 1)) break;

            const TokenOffset smallest = window.back();
            window.pop_back();
            const std::vector<uint16_t>& toks = token_offsets[smallest.token_id].positions;
            if (smallest.offset == toks.back()) continue;  // token exhausted
            TokenOffset nxt;
            nxt.token_id = smallest.token_id;
            nxt.offset_index = smallest.offset_index + 1;
            nxt.offset = toks[nxt.offset_index];
            window.push_back(nxt);
        }

        if (best_displacement == MAX_DISPLACEMENT) best_displacement = 0;
        words_present = (uint8_t)best_num_match;
        distance = uint8_t(best_displacement);
        if (populate_window) offsets = best_window;
        exact_match = 0;

        if (check_exact_match) {
            if (distance > token_offsets.size() - 1) return;
            int last_token_index = -1;
            size_t total_offsets = 0;
            for (const auto& tp : token_offsets) {
                if (tp.last_token && !tp.positions.empty()) last_token_index = tp.positions.back();
                total_offsets += tp.positions.size();
                if (total_offsets > token_offsets.size() && distance == token_offsets.size() - 1) return;
            }
            if (last_token_index == int(token_offsets.size()) - 1) {
                if (total_offsets == token_offsets.size() && distance == token_offsets.size() - 1) exact_match = 1;
                else if (distance < token_offsets.size() - 1) exact_match = 1;
            }
        }
    }
};

}  // namespace oracle
