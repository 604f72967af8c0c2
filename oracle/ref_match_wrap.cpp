// ORACLE — TEST INFRASTRUCTURE ONLY.
// Thin C wrapper that compiles the REFERENCE's own header /root/reference/include/match_score.h
// (where it lies; never copied) into oracle/_ref/libref_match.so, so the restated Match in
// oracle/match_window.h can be checked against the real thing on arbitrary inputs.
// The only shim is <glog/logging.h> -> a no-op LOG(x) (oracle/ref_shim/glog/logging.h); glog is not installed here.
// Built by oracle/Makefile only when /root/reference exists (build container); the .so travels to the
// GPU box with the snapshot, the sources do not.
#include <match_score.h>

extern "C" {
// positions: concatenated uint16 positions of all tokens; lens[t] = count for token t; last[t] = last_token flag
// out[0..3] = words_present, distance, max_offset, exact_match
void ref_match(const uint16_t* positions, const uint32_t* lens, const uint8_t* last, uint32_t n_tokens,
               int check_exact, uint8_t* out) {
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

uint64_t ref_match_score(uint8_t words_present, uint8_t distance, uint8_t max_offset, uint8_t exact_match,
                         uint32_t total_cost, uint32_t unique_words, uint8_t synonym_score) {
    Match m(words_present, distance, max_offset, exact_match);
    return m.get_match_score(total_cost, unique_words, synonym_score);
}
}
