// ORACLE — TEST INFRASTRUCTURE ONLY. Stand-in for the reference's include/field.h when oracle/Makefile compiles the reference's OWN
// include/topster.h into oracle/_ref/libref_topster.so: field.h drags in S2, ICU, RocksDB and the JSON stack, of which topster.h uses three
// names — reference_filter_result_t (a KV member the keyword path leaves empty), spp::sparse_hash_map/set (the reference's sparsepp.h, included
// where it lies) and StringUtils::hash_wy / hash_combine, whose two bodies are restated here around the reference's own wyhash_v5.h
// (include/string_utils.h:316-326). The Makefile reaches this file through oracle/_ref/inc/, a directory of symlinks to the reference headers,
// because a quoted #include searches the including file's directory first.
#pragma once
#include <cstdint>
#include <limits>
#include <map>
#include <string>
#include "sparsepp.h"
#include "wyhash_v5.h"

struct reference_filter_result_t {};

struct StringUtils {
    static uint64_t hash_wy(const void* key, uint64_t len) {
        uint64_t hash = wyhash(key, len, 0, _wyp);
        return hash != std::numeric_limits<uint64_t>::max() ? hash : (std::numeric_limits<uint64_t>::max() - 1);
    }
    static constexpr uint64_t hash_combine(uint64_t combined, uint64_t hash) {
        combined ^= hash + 0x517cc1b727220a95 + (combined << 6) + (combined >> 2);
        return combined;
    }
};
