// facet_kernels.hip.h — facet value counting over the matched ids of a query (second half): the hash-index
// branch of Index::do_facets (src/index.cpp:1518-1776, "Using hashing to find facets" :1659-1771). The reference walks
// result_ids in ascending order on one thread, looks every document up in the field's facet hash index (a posting list
// seq_id -> value hashes) and, per document, bumps result_map[hash].count for each DISTINCT hash of the document, remembering the
// last document and the hash's position inside it. Here: one thread per result id, one open-addressing table per query in HBM
// (64-bit CAS on the key, atomicAdd on the count, atomicMax on (doc << 32 | position) = "the last document wins" without order).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tsgpu {

static const int FACET_THREADS = 256;

struct FacetQueryDev {
    uint64_t ids_off;        // this query's result ids in the id arena
    uint64_t n_ids;
    uint64_t tab_off;        // its table (slots) in the table arena
    uint32_t tab_mask;       // table size - 1 (a power of two >= 2 x the distinct values it can meet)
    uint32_t first_block;    // first workgroup of this query in the launch
    uint64_t out_off;        // where its compacted (hash, count, doc, pos) entries go
    uint64_t pair_off;       // grouped counting: its (value, group) pair table in the pair arena
    uint32_t pair_mask;      // pair table size - 1 (a power of two >= 2 x the pairs it can meet)
    uint32_t pad;
};

struct FacetArgs {
    const uint64_t* doc_ptr;     // [n_docs + 1] facet hash index: document -> its hashes
    const uint32_t* hashes;
    uint32_t n_docs;
    const uint32_t* ids;         // id arena
    const FacetQueryDev* queries;
    uint32_t n_queries;
    uint32_t sample_mod;         // estimate_facets: only ids whose index i satisfies i % sample_mod == 0 (1 = every id), :1683-1687
    uint32_t ids_per_block;      // result ids one workgroup of the counting launch walks (a multiple of FACET_THREADS)
    const uint32_t* allowed;     // use_facet_query: sorted hashes that may be counted (fquery_hashes, :1742), or null
    uint32_t n_allowed;
    unsigned long long* tab_key; // 0 = empty, else (1 << 32 | hash)
    uint32_t* tab_cnt;
    unsigned long long* tab_last;  // doc_id << 32 | array_pos of the greatest document seen
    // facets of a grouped search (group_limit != 0: hash_groups[value].emplace(distinct_id), src/index.cpp:1747-1749, 1756-1758; the count of a value
    // becomes the number of groups it was seen in, :4455-4458): grouped != 0 -> every NEW (value, (uint32) distinct id) pair bumps tab_gcnt[value]
    uint32_t grouped; uint32_t group_missing_values;
    const long long* group_col; uint32_t group_len;     // distinct id per seq_id (beyond group_len: 1 with group_missing_values, else the seq_id — get_distinct_id, :7100-7142)
    unsigned long long* pair_key;                        // ~0 = empty, else value << 32 | (uint32) distinct id; the pair of all ones lives in pair_ones[query]
    uint32_t* pair_ones; uint32_t* tab_gcnt;
    // compaction
    uint32_t* out_hash; uint32_t* out_cnt; uint32_t* out_doc; uint32_t* out_pos; uint32_t* out_n;
};

__device__ inline uint32_t facet_distinct_id32(const long long* group_col, uint32_t group_len, uint32_t group_missing_values, uint32_t doc) {
    return doc < group_len ? (uint32_t)(unsigned long long)group_col[doc] : (group_missing_values ? 1u : doc);      // (hash_groups holds uint32_t: the id is truncated, include/field.h:791)
}
// hash_groups[value].emplace(distinct_id): true when the pair is new
__device__ inline bool facet_pair_insert(unsigned long long* pair_key, uint32_t* pair_ones, uint32_t qi, uint64_t pair_off, uint32_t pair_mask, uint32_t value, uint32_t did) {
    const unsigned long long key = ((unsigned long long)value << 32) | did;
    if (key == ~0ull) return atomicExch(&pair_ones[qi], 1u) == 0u;
    uint32_t slot = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 32) & pair_mask;
    for (;;) {
        unsigned long long* kp = pair_key + pair_off + slot;
        unsigned long long cur = *kp;
        if (cur == ~0ull) { cur = atomicCAS(kp, ~0ull, key); if (cur == ~0ull) return true; }
        if (cur == key) return false;
        slot = (slot + 1) & pair_mask;
    }
}

// grid = sum over queries of ceil(n_ids / ids_per_block) workgroups; block b belongs to the query q with first_block[q] <= b < first_block[q+1].
// A facet field is typically a few dozen to a few thousand values counted over up to millions of result ids: straight into the query's table
// that is millions of atomics on a handful of addresses, which serialise in L2 (measured over 10M ids, one hash each: 2 values 58 ms, 10 values
// 24 ms, 30 values 23 ms, 1 000 values 3 ms). Two levels of combining in front of the table:
//   * the lanes of a wave that count the same value count it ONCE (by their number; the last document is their maximum),
//   * every workgroup counts into a FACET_LDS_SLOTS-slot table in LDS first and adds each of its values to the query's table once at the end;
//     a value that finds no place within 8 probes of the LDS table goes straight to the query's table (a field of millions of values).
constexpr uint32_t FACET_LDS_SLOTS = 1024;
__global__ __launch_bounds__(FACET_THREADS) void facet_count_kernel(FacetArgs a) {
    __shared__ uint32_t s_q;
    __shared__ unsigned long long s_key[FACET_LDS_SLOTS], s_last[FACET_LDS_SLOTS];
    __shared__ uint32_t s_cnt[FACET_LDS_SLOTS], s_gcnt[FACET_LDS_SLOTS];
    for (uint32_t t = threadIdx.x; t < FACET_LDS_SLOTS; t += FACET_THREADS) { s_key[t] = 0; s_last[t] = 0; s_cnt[t] = 0; s_gcnt[t] = 0; }
    if (threadIdx.x == 0) {
        uint32_t lo = 0, hi = a.n_queries;                      // last query whose first_block <= blockIdx.x
        while (lo + 1 < hi) { const uint32_t mid = (lo + hi) >> 1; if (a.queries[mid].first_block <= blockIdx.x) lo = mid; else hi = mid; }
        s_q = lo;
    }
    __syncthreads();
    const FacetQueryDev q = a.queries[s_q];
    const uint32_t lane = threadIdx.x & 63;
    auto bump_table = [&](uint32_t fh, uint32_t times, uint32_t gtimes, unsigned long long last) {    // result_map[fh].count += times; the greatest (doc, position) wins
        const unsigned long long key = (1ull << 32) | fh;
        uint32_t slot = (fh * 2654435761u) & q.tab_mask;
        for (;;) {
            unsigned long long* kp = a.tab_key + q.tab_off + slot;
            unsigned long long cur = *kp;
            if (cur == 0) cur = atomicCAS(kp, 0ull, key), cur = cur == 0 ? key : cur;
            if (cur == key) break;
            slot = (slot + 1) & q.tab_mask;
        }
        atomicAdd(&a.tab_cnt[q.tab_off + slot], times);
        if (gtimes) atomicAdd(&a.tab_gcnt[q.tab_off + slot], gtimes);
        atomicMax(&a.tab_last[q.tab_off + slot], last);
    };
    auto bump = [&](uint32_t fh, uint32_t times, uint32_t gtimes, unsigned long long last) {
        const unsigned long long key = (1ull << 32) | fh;
        uint32_t slot = ((fh * 2654435761u) >> 16) & (FACET_LDS_SLOTS - 1);
        for (int probe = 0; probe < 8; probe++) {
            unsigned long long cur = s_key[slot];
            if (cur == 0) cur = atomicCAS(&s_key[slot], 0ull, key), cur = cur == 0 ? key : cur;
            if (cur == key) { atomicAdd(&s_cnt[slot], times); if (gtimes) atomicAdd(&s_gcnt[slot], gtimes); atomicMax(&s_last[slot], last); return; }
            slot = (slot + 1) & (FACET_LDS_SLOTS - 1);
        }
        bump_table(fh, times, gtimes, last);
    };
    for (uint32_t rep = 0; rep < a.ids_per_block; rep += FACET_THREADS) {           // (workgroup-uniform)
        const uint64_t i = (uint64_t)(blockIdx.x - q.first_block) * a.ids_per_block + rep + threadIdx.x;
        // (nobody leaves: the wave-wide steps below need every lane)
        bool live = i < q.n_ids && !(a.sample_mod > 1 && (i % a.sample_mod) != 0);
        const uint32_t doc = live ? a.ids[q.ids_off + i] : 0u;
        if (live && doc >= a.n_docs) live = false;                  // beyond the index: facet_index_it is exhausted (:1692-1694)
        const uint64_t h0 = live ? a.doc_ptr[doc] : 0, h1 = live ? a.doc_ptr[doc + 1] : 0;
        for (uint64_t it = 0; __ballot(live && h0 + it < h1 ? 1 : 0) != 0; it++) {
            const uint64_t j = h0 + it;
            bool count_it = live && j < h1;
            uint32_t fh = 0;
            if (count_it) {
                fh = a.hashes[j];
                for (uint64_t p = h0; p < j && count_it; p++) if (a.hashes[p] == fh) count_it = false;       // unique_facet_hashes: a value repeated inside one document counts once (:1722-1728)
                if (count_it && a.allowed) {
                    uint32_t lo = 0, hi = a.n_allowed;
                    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a.allowed[mid] < fh) lo = mid + 1; else hi = mid; }
                    if (lo >= a.n_allowed || a.allowed[lo] != fh) count_it = false;
                }
            }
            const unsigned long long mine_last = ((unsigned long long)doc << 32) | (unsigned long long)it;
            const bool is_new = count_it && a.grouped &&
                                facet_pair_insert(a.pair_key, a.pair_ones, s_q, q.pair_off, q.pair_mask, fh, facet_distinct_id32(a.group_col, a.group_len, a.group_missing_values, doc));
            bool pending = count_it;
            for (int round = 0; round < 16; round++) {
                const unsigned long long rem = __ballot(pending ? 1 : 0);
                if (!rem) break;                                     // (wave-uniform)
                const uint32_t leader = (uint32_t)__ffsll((long long)rem) - 1;
                const uint32_t lfh = __shfl(fh, (int)leader, 64);
                const bool mine = pending && fh == lfh;
                const unsigned long long same = __ballot(mine ? 1 : 0);
                if (__popcll(same) < 3) break;                       // (wave-uniform) the first waiting lane is nearly alone with its value: many values, one by one below
                unsigned long long mx = mine ? mine_last : 0ull;
                for (int d = 32; d > 0; d >>= 1) { const unsigned long long o = __shfl_xor(mx, d, 64); if (o > mx) mx = o; }
                const unsigned long long fresh = __ballot(mine && is_new ? 1 : 0);
                if (lane == leader) bump(fh, (uint32_t)__popcll(same), (uint32_t)__popcll(fresh), mx);
                if (mine) pending = false;
            }
            if (pending) bump(fh, 1u, is_new ? 1u : 0u, mine_last);
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < FACET_LDS_SLOTS; t += FACET_THREADS)
        if (s_key[t] != 0) bump_table((uint32_t)s_key[t], s_cnt[t], s_gcnt[t], s_last[t]);
}

// the occupied slots of every query's table -> a dense list (any order; the host orders by hash). grid = (n_queries, min(ceil(largest table /
// FACET_COMPACT_SLOTS), 4096)): a table of millions of slots is swept by hundreds of workgroups; each wave
// reserves its output range with ONE add to the query's counter (out_n, zeroed by the host).
constexpr uint32_t FACET_COMPACT_PER_THREAD = 16;
constexpr uint32_t FACET_COMPACT_SLOTS = FACET_THREADS * FACET_COMPACT_PER_THREAD;
__global__ __launch_bounds__(FACET_THREADS) void facet_compact_kernel(FacetArgs a) {
    const FacetQueryDev q = a.queries[blockIdx.x];
    const uint32_t lane = threadIdx.x & 63;
    // (workgroup-uniform bounds: a smaller table than the largest of the batch leaves its later workgroups idle)
    for (uint64_t base = (uint64_t)blockIdx.y * FACET_COMPACT_SLOTS; base <= q.tab_mask; base += (uint64_t)gridDim.y * FACET_COMPACT_SLOTS)
    for (uint32_t r = 0; r < FACET_COMPACT_PER_THREAD; r++) {
        const uint64_t s = base + (uint64_t)r * FACET_THREADS + threadIdx.x;
        const unsigned long long k = s <= q.tab_mask ? a.tab_key[q.tab_off + s] : 0ull;
        const unsigned long long occ = __ballot(k != 0 ? 1 : 0);
        if (!occ) continue;                                     // (wave-uniform)
        const uint32_t leader = (uint32_t)__ffsll((long long)occ) - 1;
        uint32_t at = 0;
        if (lane == leader) at = atomicAdd(&a.out_n[blockIdx.x], (uint32_t)__popcll(occ));
        at = __shfl(at, (int)leader, 64);
        if (k == 0) continue;
        at += (uint32_t)__popcll(occ & ((1ull << lane) - 1ull));
        const unsigned long long last = a.tab_last[q.tab_off + s];
        a.out_hash[q.out_off + at] = (uint32_t)k;
        a.out_cnt[q.out_off + at] = a.grouped ? a.tab_gcnt[q.tab_off + s] : a.tab_cnt[q.tab_off + s];       // (grouped: hash_groups[value].size(), src/index.cpp:4455-4458)
        a.out_doc[q.out_off + at] = (uint32_t)(last >> 32);
        a.out_pos[q.out_off + at] = (uint32_t)last;
    }
}

// one workgroup per query: its compacted list in ascending hash order (result_map is keyed by the hash; the caller takes the first `cap` values) — a bitonic
// sort of (hash << 32 | position) in LDS, then the entries gathered into the sorted arrays. 1 000 queries of ~500 values each cost the host 10 ms of
// std::sort + gather before. A query with more than FACET_SORT_MAX values is passed through as it is (the host orders that one).
constexpr uint32_t FACET_SORT_MAX = 4096;
struct FacetSortOut { uint32_t* hash; uint32_t* cnt; uint32_t* doc; uint32_t* pos; };
__global__ __launch_bounds__(FACET_THREADS) void facet_sort_kernel(FacetArgs a, FacetSortOut o) {
    __shared__ unsigned long long keys[FACET_SORT_MAX];
    const FacetQueryDev q = a.queries[blockIdx.x];
    const uint32_t n = a.out_n[blockIdx.x];
    if (n > FACET_SORT_MAX) {                                                   // (workgroup-uniform)
        for (uint32_t i = threadIdx.x; i < n; i += FACET_THREADS) {
            o.hash[q.out_off + i] = a.out_hash[q.out_off + i]; o.cnt[q.out_off + i] = a.out_cnt[q.out_off + i];
            o.doc[q.out_off + i] = a.out_doc[q.out_off + i]; o.pos[q.out_off + i] = a.out_pos[q.out_off + i];
        }
        return;
    }
    uint32_t p2 = 1;
    while (p2 < n) p2 <<= 1;
    for (uint32_t i = threadIdx.x; i < p2; i += FACET_THREADS) keys[i] = i < n ? (((unsigned long long)a.out_hash[q.out_off + i] << 32) | i) : ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= p2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < p2; t += FACET_THREADS) {
                const uint32_t u = t ^ j;
                if (u > t) {
                    const unsigned long long x = keys[t], y = keys[u];
                    if ((x > y) == ((t & k) == 0)) { keys[t] = y; keys[u] = x; }
                }
            }
            __syncthreads();
        }
    for (uint32_t i = threadIdx.x; i < n; i += FACET_THREADS) {
        const uint32_t at = (uint32_t)keys[i];
        o.hash[q.out_off + i] = (uint32_t)(keys[i] >> 32); o.cnt[q.out_off + i] = a.out_cnt[q.out_off + at];
        o.doc[q.out_off + i] = a.out_doc[q.out_off + at]; o.pos[q.out_off + i] = a.out_pos[q.out_off + at];
    }
}

// ------------------------------------------------------------------------------------------------
// Range facets of the hash-index branch (a_facet.is_range_query, src/index.cpp:1738-1750): per result document that the facet hash index holds, ONCE PER DISTINCT
// HASH of the document (the branch sits inside the loop over its hashes), doc_val = the field's sort-index value (get_doc_val_from_sort_index, :1470-1482: INT64_MAX
// when absent) is looked up in facet_range_map (facet::get_range, include/field.h:820-838: the first range whose upper bound is GREATER than the value, taken when
// value >= its lower bound) and result_map[range_id].count += 1; with group_limit also hash_groups[range_id].emplace(distinct_id), the final count being the set's
// size (:4455-4458) — sets keyed by (uint32) range_id, so ranges whose upper bounds agree in their low 32 bits share one (range_set[r] = the first such range).
// Ranges are given in ascending upper-bound order (std::map order). One LDS counter per range and workgroup, one add per range and workgroup at the end.
constexpr uint32_t FACET_MAX_RANGES = 1024;
struct FacetRangeArgs {
    const uint64_t* doc_ptr; const uint32_t* hashes; uint32_t n_docs;
    const uint32_t* ids; const FacetQueryDev* queries; uint32_t n_queries; uint32_t sample_mod; uint32_t ids_per_block;
    const long long* val_col; uint32_t val_len;                                 // the field's sort index as a dense column (beyond val_len: INT64_MAX)
    const long long* range_upper; const long long* range_lower; const uint32_t* range_set; uint32_t n_ranges;
    uint32_t grouped; uint32_t group_missing_values; const long long* group_col; uint32_t group_len;
    unsigned long long* pair_key; uint32_t* pair_ones;
    uint32_t* out_count; uint32_t* out_gcount;                                  // [n_queries][n_ranges], zeroed by the host
};
__global__ __launch_bounds__(FACET_THREADS) void facet_range_kernel(FacetRangeArgs a) {
    __shared__ uint32_t s_q;
    __shared__ uint32_t s_cnt[FACET_MAX_RANGES], s_gcnt[FACET_MAX_RANGES];
    for (uint32_t t = threadIdx.x; t < a.n_ranges; t += FACET_THREADS) { s_cnt[t] = 0; s_gcnt[t] = 0; }
    if (threadIdx.x == 0) {
        uint32_t lo = 0, hi = a.n_queries;
        while (lo + 1 < hi) { const uint32_t mid = (lo + hi) >> 1; if (a.queries[mid].first_block <= blockIdx.x) lo = mid; else hi = mid; }
        s_q = lo;
    }
    __syncthreads();
    const FacetQueryDev q = a.queries[s_q];
    for (uint32_t rep = 0; rep < a.ids_per_block; rep += FACET_THREADS) {
        const uint64_t i = (uint64_t)(blockIdx.x - q.first_block) * a.ids_per_block + rep + threadIdx.x;
        if (i >= q.n_ids || (a.sample_mod > 1 && (i % a.sample_mod) != 0)) continue;
        const uint32_t doc = a.ids[q.ids_off + i];
        if (doc >= a.n_docs) continue;                                          // beyond the index: facet_index_it is exhausted
        const uint64_t h0 = a.doc_ptr[doc], h1 = a.doc_ptr[doc + 1];
        uint32_t times = 0;                                                     // distinct hashes of the document (unique_facet_hashes)
        for (uint64_t j = h0; j < h1; j++) {
            bool dup = false;
            for (uint64_t p = h0; p < j && !dup; p++) dup = a.hashes[p] == a.hashes[j];
            times += dup ? 0u : 1u;
        }
        if (times == 0) continue;                                               // not in the facet hash index
        const long long v = doc < a.val_len ? a.val_col[doc] : 0x7FFFFFFFFFFFFFFFll;
        uint32_t lo = 0, hi = a.n_ranges;                                       // lower_bound(v), stepping over an upper bound equal to v = the first upper bound > v
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a.range_upper[mid] <= v) lo = mid + 1; else hi = mid; }
        if (lo >= a.n_ranges || v < a.range_lower[lo]) continue;
        atomicAdd(&s_cnt[lo], times);
        if (a.grouped) {
            const uint32_t set = a.range_set[lo];
            if (facet_pair_insert(a.pair_key, a.pair_ones, s_q, q.pair_off, q.pair_mask, set, facet_distinct_id32(a.group_col, a.group_len, a.group_missing_values, doc)))
                atomicAdd(&s_gcnt[set], 1u);
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < a.n_ranges; t += FACET_THREADS) {
        if (s_cnt[t]) atomicAdd(&a.out_count[(size_t)s_q * a.n_ranges + t], s_cnt[t]);
        if (s_gcnt[t]) atomicAdd(&a.out_gcount[(size_t)s_q * a.n_ranges + t], s_gcnt[t]);
    }
}

// ------------------------------------------------------------------------------------------------
// Numeric facet stats of the hash-index branch (should_compute_stats: src/index.cpp:1730-1741 -> compute_facet_stats :1430-1460):
// every (document, DISTINCT hash) event of the walk above — before the facet-query filter — contributes its value to min / max / sum /
// count. The hash IS the value for int32 (the integer) and float (its bits) fields; int64 fields go through fhash_int64_map (:1733-1738,
// a missing hash counts as INT64_MAX). min / max / count are order-free; the sum is accumulated as an exact 64-bit integer (integer
// types: equal to the reference's double accumulation whenever every partial sum is below 2^53 — reported as sum_exact) or with
// double atomics (float: the reference adds in document order; a different order moves the last bits of the double only).
struct FacetStatsDev { unsigned long long vmin, vmax, isum, count; double fsum; unsigned long long absmax; };   // vmin / vmax: order-preserving keys
struct FacetStatsArgs {
    const uint64_t* doc_ptr; const uint32_t* hashes; uint32_t n_docs;
    const uint32_t* ids; const FacetQueryDev* queries; uint32_t n_queries; uint32_t sample_mod;
    int value_type;                     // 0 int32, 1 int64, 2 float
    const uint32_t* map_hash; const long long* map_val; uint32_t n_map;      // int64: sorted hashes -> values
    FacetStatsDev* out;                 // [n_queries], vmin pre-set to all ones, the rest to zero
};
__device__ inline unsigned long long facet_i64_key(long long v) { return (unsigned long long)v ^ 0x8000000000000000ull; }
__device__ inline unsigned long long facet_f32_key(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? (uint32_t)~b : (b | 0x80000000u); }
__global__ __launch_bounds__(FACET_THREADS) void facet_stats_kernel(FacetStatsArgs a) {
    __shared__ uint32_t s_q;
    __shared__ unsigned long long s_min, s_max, s_isum, s_cnt, s_abs;
    __shared__ double s_fsum;
    if (threadIdx.x == 0) {
        uint32_t lo = 0, hi = a.n_queries;
        while (lo + 1 < hi) { const uint32_t mid = (lo + hi) >> 1; if (a.queries[mid].first_block <= blockIdx.x) lo = mid; else hi = mid; }
        s_q = lo; s_min = ~0ull; s_max = 0; s_isum = 0; s_cnt = 0; s_abs = 0; s_fsum = 0.0;
    }
    __syncthreads();
    const FacetQueryDev q = a.queries[s_q];
    const uint64_t i = (uint64_t)(blockIdx.x - q.first_block) * FACET_THREADS + threadIdx.x;
    unsigned long long mn = ~0ull, mx = 0, isum = 0, cnt = 0, amax = 0;
    double fsum = 0.0;
    if (i < q.n_ids && !(a.sample_mod > 1 && (i % a.sample_mod) != 0)) {
        const uint32_t doc = a.ids[q.ids_off + i];
        if (doc < a.n_docs) {
            const uint64_t h0 = a.doc_ptr[doc], h1 = a.doc_ptr[doc + 1];
            for (uint64_t j = h0; j < h1; j++) {
                const uint32_t fh = a.hashes[j];
                bool dup = false;
                for (uint64_t p = h0; p < j && !dup; p++) dup = a.hashes[p] == fh;
                if (dup) continue;
                unsigned long long key;
                if (a.value_type == 2) {
                    const float v = __uint_as_float(fh);
                    key = facet_f32_key(v);
                    fsum += (double)v;
                } else {
                    long long v = (long long)(int32_t)fh;
                    if (a.value_type == 1) {
                        uint32_t lo = 0, hi = a.n_map;
                        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a.map_hash[mid] < fh) lo = mid + 1; else hi = mid; }
                        v = (lo < a.n_map && a.map_hash[lo] == fh) ? a.map_val[lo] : 0x7FFFFFFFFFFFFFFFll;
                    }
                    key = facet_i64_key(v);
                    isum += (unsigned long long)v;
                    fsum += (double)v;                                    // (used when the exact 64-bit sum cannot stand for the reference's double sum)
                    const unsigned long long av = v < 0 ? (unsigned long long)(-(v + 1)) + 1ull : (unsigned long long)v;
                    amax = av > amax ? av : amax;
                }
                mn = key < mn ? key : mn; mx = key > mx ? key : mx;
                cnt++;
            }
        }
    }
    if (cnt) {
        atomicMin(&s_min, mn); atomicMax(&s_max, mx); atomicAdd(&s_isum, isum); atomicAdd(&s_cnt, cnt); atomicMax(&s_abs, amax);
        atomicAdd(&s_fsum, fsum);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) {
        FacetStatsDev* o = a.out + s_q;
        atomicMin(&o->vmin, s_min); atomicMax(&o->vmax, s_max); atomicAdd(&o->isum, s_isum); atomicAdd(&o->count, s_cnt); atomicMax(&o->absmax, s_abs);
        atomicAdd(&o->fsum, s_fsum);
    }
}

// ------------------------------------------------------------------------------------------------
// Value-index branch of Index::do_facets ("Using intersection to find facets", src/index.cpp:1596-1657 -> facet_index_t::intersect,
// src/facet_index.cpp:230-353): the field's values are visited in a fixed order (counter_list = by total count, or alphabetical);
// a value's count = |its seq_id list ∩ result_ids| (ids_t::intersect_count; with estimate_facets and more than 300 ids the strided
// walk of id_list_t::intersect_count, src/id_list.cpp:725-766), and the walk stops once max_facets values with a non-zero count have
// been found. Here: one wavefront per (value, query) computes every count at once, then one wavefront per query picks, in visiting
// order, the first max_facets values whose count is non-zero.
struct FacetValueArgs {
    const uint64_t* val_ptr; const uint32_t* val_ids; const uint32_t* val_total; uint32_t n_values;   // value v: val_ids[val_ptr[v] .. val_ptr[v+1]) ascending
    const uint32_t* ids; const FacetQueryDev* queries; uint32_t n_queries;                              // (ids_off, n_ids used)
    const uint32_t* order;              // nullable: visiting order (a permutation of the values)
    uint32_t max_facets; int wildcard_no_filter; int estimate; uint32_t interval;
    uint32_t* counts;                   // [n_queries][n_values]
    uint32_t cap; uint32_t* out_value; uint32_t* out_count; uint32_t* out_doc; uint32_t* out_n;        // [n_queries][cap], [n_queries]
};
__global__ __launch_bounds__(64) void facet_value_count_kernel(FacetValueArgs a) {
    const uint32_t v = blockIdx.x, qi = blockIdx.y, lane = threadIdx.x;
    const FacetQueryDev q = a.queries[qi];
    const uint64_t v0 = a.val_ptr[v], nv = a.val_ptr[v + 1] - v0;
    const uint32_t* __restrict__ A = a.val_ids + v0;
    const uint32_t* __restrict__ R = a.ids + q.ids_off;
    const uint64_t nr = q.n_ids;
    uint32_t count = 0;
    if (a.wildcard_no_filter) {
        count = a.val_total[v];                                             // :305-306
    } else if (a.estimate && nv > 300) {
        if (lane == 0) {                                                    // the reference's strided walk, step by step
            uint64_t i = 0, r = 0, c = 0;
            while (i < nv && r < nr) {
                const uint32_t x = A[i], y = R[r];
                if (x == y) { c++; i += a.interval; r += a.interval; }
                else if (x < y) i += a.interval;
                else r += a.interval;
            }
            c = c * a.interval * a.interval;
            count = (uint32_t)(c < nv ? c : nv);
        }
        count = __shfl(count, 0);
    } else {
        // exact: lanes stride over the shorter list and look each id up in the longer one
        const uint32_t* S = nv <= nr ? A : R; const uint32_t* L = nv <= nr ? R : A;
        const uint64_t ns = nv <= nr ? nv : nr, nl = nv <= nr ? nr : nv;
        for (uint64_t i = lane; i < ns; i += 64) {
            const uint32_t x = S[i];
            uint64_t lo = 0, hi = nl;
            while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (L[mid] < x) lo = mid + 1; else hi = mid; }
            count += (lo < nl && L[lo] == x) ? 1u : 0u;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) count += __shfl_xor(count, off);
    }
    if (lane == 0) a.counts[(size_t)qi * a.n_values + v] = count;
}
__global__ __launch_bounds__(64) void facet_value_select_kernel(FacetValueArgs a) {
    const uint32_t qi = blockIdx.x, lane = threadIdx.x;
    uint32_t found = 0;
    for (uint32_t base = 0; base < a.n_values && found < a.max_facets; base += 64) {
        const uint32_t pos = base + lane;
        const uint32_t v = pos < a.n_values ? (a.order ? a.order[pos] : pos) : 0;
        const uint32_t c = pos < a.n_values ? a.counts[(size_t)qi * a.n_values + v] : 0;
        const unsigned long long m = __ballot(c != 0);
        const uint32_t rank = found + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (c != 0 && rank < a.max_facets && rank < a.cap) {
            const size_t o = (size_t)qi * a.cap + rank;
            a.out_value[o] = v; a.out_count[o] = c; a.out_doc[o] = a.val_ids[a.val_ptr[v]];     // ids_t::first_id(ids), :314
        }
        found += (uint32_t)__popcll(m);
    }
    if (lane == 0) a.out_n[qi] = found < a.max_facets ? found : a.max_facets;
}

}  // namespace tsgpu
