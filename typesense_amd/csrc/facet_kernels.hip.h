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
};

struct FacetArgs {
    const uint64_t* doc_ptr;     // [n_docs + 1] facet hash index: document -> its hashes
    const uint32_t* hashes;
    uint32_t n_docs;
    const uint32_t* ids;         // id arena
    const FacetQueryDev* queries;
    uint32_t n_queries;
    uint32_t sample_mod;         // estimate_facets: only ids whose index i satisfies i % sample_mod == 0 (1 = every id), :1683-1687
    const uint32_t* allowed;     // use_facet_query: sorted hashes that may be counted (fquery_hashes, :1742), or null
    uint32_t n_allowed;
    unsigned long long* tab_key; // 0 = empty, else (1 << 32 | hash)
    uint32_t* tab_cnt;
    unsigned long long* tab_last;  // doc_id << 32 | array_pos of the greatest document seen
    // compaction
    uint32_t* out_hash; uint32_t* out_cnt; uint32_t* out_doc; uint32_t* out_pos; uint32_t* out_n;
};

// grid = sum over queries of ceil(n_ids / 256) workgroups; block b belongs to the query q with first_block[q] <= b < first_block[q+1]
__global__ __launch_bounds__(FACET_THREADS) void facet_count_kernel(FacetArgs a) {
    __shared__ uint32_t s_q;
    if (threadIdx.x == 0) {
        uint32_t lo = 0, hi = a.n_queries;                      // last query whose first_block <= blockIdx.x
        while (lo + 1 < hi) { const uint32_t mid = (lo + hi) >> 1; if (a.queries[mid].first_block <= blockIdx.x) lo = mid; else hi = mid; }
        s_q = lo;
    }
    __syncthreads();
    const FacetQueryDev q = a.queries[s_q];
    const uint64_t i = (uint64_t)(blockIdx.x - q.first_block) * FACET_THREADS + threadIdx.x;
    if (i >= q.n_ids) return;
    if (a.sample_mod > 1 && (i % a.sample_mod) != 0) return;
    const uint32_t doc = a.ids[q.ids_off + i];
    if (doc >= a.n_docs) return;                                // beyond the index: facet_index_it is exhausted (:1692-1694)
    const uint64_t h0 = a.doc_ptr[doc], h1 = a.doc_ptr[doc + 1];
    for (uint64_t j = h0; j < h1; j++) {
        const uint32_t fh = a.hashes[j];
        bool dup = false;                                       // unique_facet_hashes: a value repeated inside one document counts once (:1722-1728)
        for (uint64_t p = h0; p < j && !dup; p++) dup = a.hashes[p] == fh;
        if (dup) continue;
        if (a.allowed) {
            uint32_t lo = 0, hi = a.n_allowed;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a.allowed[mid] < fh) lo = mid + 1; else hi = mid; }
            if (lo >= a.n_allowed || a.allowed[lo] != fh) continue;
        }
        const unsigned long long key = (1ull << 32) | fh;
        uint32_t slot = (fh * 2654435761u) & q.tab_mask;
        for (;;) {
            unsigned long long* kp = a.tab_key + q.tab_off + slot;
            unsigned long long cur = *kp;
            if (cur == 0) cur = atomicCAS(kp, 0ull, key), cur = cur == 0 ? key : cur;
            if (cur == key) break;
            slot = (slot + 1) & q.tab_mask;
        }
        atomicAdd(&a.tab_cnt[q.tab_off + slot], 1u);
        atomicMax(&a.tab_last[q.tab_off + slot], ((unsigned long long)doc << 32) | (unsigned long long)(j - h0));
    }
}

// one workgroup per query: the occupied slots of its table -> a dense list (any order; the host orders by hash)
__global__ __launch_bounds__(FACET_THREADS) void facet_compact_kernel(FacetArgs a) {
    __shared__ uint32_t s_n;
    const FacetQueryDev q = a.queries[blockIdx.x];
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (uint64_t s = threadIdx.x; s <= q.tab_mask; s += FACET_THREADS) {
        const unsigned long long k = a.tab_key[q.tab_off + s];
        if (k == 0) continue;
        const uint32_t at = atomicAdd(&s_n, 1u);
        const unsigned long long last = a.tab_last[q.tab_off + s];
        a.out_hash[q.out_off + at] = (uint32_t)k;
        a.out_cnt[q.out_off + at] = a.tab_cnt[q.tab_off + s];
        a.out_doc[q.out_off + at] = (uint32_t)(last >> 32);
        a.out_pos[q.out_off + at] = (uint32_t)last;
    }
    __syncthreads();
    if (threadIdx.x == 0) a.out_n[blockIdx.x] = s_n;
}

}  // namespace tsgpu
