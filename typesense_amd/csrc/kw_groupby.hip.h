// kw_groupby.hip.h — group_by on the device: the DISTINCT form of the Topster and what the reference wraps around it
// on the scoring path. Included at the end of kw_kernels.hip.h (it scores with the same device functions as the keyword kernels).
//
// Reference (one pass of Index::run_search's two-pass protocol, src/index.cpp:2488-2760):
//   search_across_fields, group_limit != 0   : src/index.cpp:5511-5520 (distinct_id = get_distinct_id over the group_by fields), :5546-5549
//                                              (ret = topster->add(&kv); if (ret < 2) groups_processed[distinct_id]++)
//   Topster(capacity, distinct, first_pass)  : include/topster.h:266-296
//     first pass  (:342-349, :378-428)       : the heap is keyed by distinct key — ONE KV per group, its greatest — and keeps the `capacity`
//                                              groups with the greatest such KVs; every distinct key reaches loglog_counter (LogLogBeta over
//                                              hash_wy(std::to_string(key)), include/loglogbeta.h) -> getGroupsCount() = found
//     second pass (:355-376)                 : group_kv_map[distinct_key] = a Topster of `distinct` (group_limit) KVs per group
//   populate_result_kvs, grouped branch      : src/index.cpp:8962-9011: the groups' heads go through a Topster(capacity); the surviving groups,
//                                              best head first, each with its KVs in sort() order
// Both passes are ORDER-FREE functions of the set of (key, distinct_key, scores) the pass produced, because is_greater / is_smaller
// (include/topster.h:146-154) order KVs totally (the key is unique within a pass) and a document belongs to one group:
//   first pass  = for every group its greatest KV; the `capacity` greatest of those            (a KV evicted from the heap is smaller than every
//                 later heap minimum, so a group that re-enters does so with a KV greater than anything it lost — the heap never ends on a
//                 group's non-greatest KV; oracle/group_topster.h replays the heap itself and agrees on 280 random streams)
//   second pass = the same selection of groups (heads = greatest KVs), each with its min(group_limit, members) greatest KVs, descending;
//                 groups_processed[g] = members of g (every add returns 1: a seq_id is met once per pass)
// That is what the kernels below compute, with one open-addressing table per query in HBM (like facet_kernels.hip.h):
//   gb_score_kernel    one thread per matched id: probes every (token, field) list of the query for the id, scores it exactly like the keyword
//                      kernels (agg_score_mf + sort_scores; the wildcard form: sort_scores with the constant 100), reads its distinct key from
//                      the group column -> (s0, s1, s2, dkey) per matched id
//   gb_insert_kernel   one thread per matched id: claims / finds the slot of its distinct key (64-bit CAS), counts the group's members
//                      (atomicAdd), and raises the group's best record (a CAS loop on a record index: records are immutable, the comparison is
//                      the Topster's) — no ordering between threads is needed
//   gb_select_kernel   one workgroup per query: walks the table; every group's best record goes through the LDS top-K buffer the keyword
//                      kernels use (TopkLds, bitonic compaction) -> the `capacity` best groups in sort order, their rank written back into the
//                      table; first pass: the LogLogBeta registers of ALL distinct keys are built in LDS (wyhash of the decimal string, restated
//                      for the 1..20 bytes a uint64 prints to) and the groups' best KVs are the hits; second pass: member-list offsets
//   gb_scatter_kernel  second pass, one thread per matched id: members of a selected group append themselves to the group's member list
//   gb_members_kernel  second pass, one wave per (query, selected group of <= GB_BIG members): the group's min(group_limit, members) greatest records by
//                      repeated wave-wide extraction of the greatest record below the previous one (group_limit is 3 by default)
//   gb_chunk_kernel    second pass, bigger groups: one workgroup per chunk of GB_CHUNK members (LDS top-K buffer, k = group_limit), the group's last chunk folds
//                      the partial lists and writes the hits
//   gb_dedupe_kernel   candidate combinations (Index::search_all_candidates with group_limit != 0: several passes over ONE collector), second pass only: a
//                      document met by several combinations keeps ONE record — its greatest KV, the later combination on ties (Topster::add replaces unless
//                      smaller, topster.h:392-406; group_doc_seq_ids -> ret == 2: counted once) — in a per-query document table; the others leave the fold.
//                      A user query's items are its combinations' ascending id lists one after the other (GbQuery::first_combo / n_combos, GbArgs::combo_begin).
// Bound: HBM latency / atomics of a random-access table, like the facet kernels; algorithmic bytes per matched id = 4 (id) + 32 (record) + the
// posting probes of gb_score_kernel. Grouped queries are the minority of a server's traffic; the ungrouped keyword path is untouched.
#pragma once
// (inside namespace tsgpu)

static const int GB_THREADS = 256;
constexpr uint32_t GB_WG_SLOTS_LOG2 = 9, GB_WG_SLOTS = 1u << GB_WG_SLOTS_LOG2;      // a workgroup's LDS table over its GB_THREADS items (insert / scatter): twice as many slots, never full
static const unsigned long long GB_EMPTY = ~0ull;     // empty table slot; a distinct key that IS ~0 owns the extra slot tab_mask + 1
static const uint32_t GB_NONE = 0xFFFFFFFFu;
static const uint32_t GB_LOGLOG_M = 16384;            // LogLogBeta::M (PRECISION 14)
static const uint32_t GB_LOGLOG_HIST = 52;            // register values 0..51 (rho <= 50 + 1: the low 14 bits of the shifted hash are ones)
static const uint32_t GB_BIG = 4096;                  // second pass: a selected group with more members than this is cut into chunks (gb_chunk_kernel) instead of being
static const uint32_t GB_CHUNK = 8192;                // walked by one wave (gb_members_kernel): members per chunk

struct GbQuery {
    uint64_t item_begin;      // first matched id of the query in the flat arrays
    uint64_t tab_off;         // first slot of its table
    uint32_t n_items;
    uint32_t tab_mask;        // slots - 1 (a power of two >= 2 x n_items)
    uint32_t k;               // Topster capacity
    uint32_t group_limit;
    uint32_t column;          // the distinct-key column
    uint32_t first_block;     // the query's first workgroup in the per-item launches (ceil(n_items / GB_THREADS) workgroups each: no workgroup spans two queries)
    uint32_t first_sblock;    // ... and in the member scatter (ceil(n_items / GB_SCATTER_ITEMS) workgroups each)
    uint8_t first_pass, group_missing_values, wildcard, run;   // run = 0: the query failed upstream, nothing is produced
    uint8_t iota, dedupe, forced, pad[5];   // forced: the groups to return are GIVEN (forced_begin / n_forced below). iota: q = * over the whole collection, the matched ids are 0 .. n_items - 1 (gb_iota_kernel writes them); dedupe: a SECOND pass over several
                              // candidate combinations — a document met by several of them counts once, with its greatest KV (the later combination on ties)
    uint32_t pw_begin, pw_cap;      // second pass: the query's chunk work list (entries [pw_begin, pw_begin + pw_cap): n_items / GB_CHUNK + n_items / GB_BIG + 1 at most)
    uint64_t pbuf_off;              // ... and its partial top-L buffer (pw_cap x group_limit entries)
    uint32_t forced_begin, n_forced; // forced = 1 (a doc-range shard answering for the groups the whole collection selected, tsgpu_group_keyword_search_grouped_batch): returned group r
                              // IS the key GbArgs::forced_keys[forced_begin + r], r < n_forced <= k — present on this shard or not (then group_found = group_size = 0)
    uint32_t first_combo, n_combos; // the user query's candidate combinations (search_all_candidates: one search_across_fields pass each; 1 = a plain pass): its items are the
                              // combinations' matched ids one after the other, combination c at items [combo_begin[c], combo_begin[c + 1])
};

struct GbArgs {
    const GbQuery* gq; uint32_t n_queries;
    const KwQueryDev* queries; const KwQueryMF* mfs;                       // per COMBINATION
    const unsigned long long* forced_keys;                                     // the given groups of the forced queries (GbQuery::forced_begin)
    const unsigned long long* combo_begin; const uint32_t* qidx_of_combo;   // per combination: first item; KV::query_index of its hits (earlier combinations of the user query that matched)
    uint8_t* pass;                                                             // per matched id: which combination of its user query met it
    uint32_t* dkey32; uint32_t* dbest;                                         // per table slot, dedupe only: the document table (seq_id -> its greatest record)
    uint32_t* out_qidx;                                                        // per hit slot (like out): KV::query_index
    uint32_t n_pw;                                                             // chunk work list: total capacity (= the chunk kernel's grid), entries (group rank | chunk << 10),
    uint32_t* pw_list; uint32_t* pw_count; uint32_t* pw_n;                     // per-query count, entries of each partial; per (query, group): first partial, chunks, ticket
    uint32_t* g_pfirst; uint32_t* g_nchunk; uint32_t* g_ticket;
    int64_t* pb_s0; int64_t* pb_s1; int64_t* pb_s2; int64_t* pb_key;           // partial top-L lists (sorted), group_limit entries per work-list entry
    uint64_t n_items;
    const uint32_t* ids;                                                       // matched ids, ascending per query
    int64_t* s0; int64_t* s1; int64_t* s2; unsigned long long* dkey; uint32_t* rslot;    // per matched id
    unsigned long long* hkey; uint32_t* hcount; uint32_t* hbest; uint32_t* hrank;          // per table slot
    uint32_t* glist; uint32_t* gcount;                                         // the slots in use, per query (at item_begin; <= n_items of them) and how many
    uint32_t* members;                                                         // second pass: per query, the selected groups' members (local item indices)
    uint32_t g_stride;                                                         // group slots per query (>= every k)
    uint32_t* n_groups; unsigned long long* groups_total;                      // per query
    unsigned long long* g_dkey; uint32_t* g_found; uint32_t* g_size; uint32_t* g_mofs; uint32_t* g_mcur;   // per (query, returned group)
    uint8_t* loglog;                                                           // [n_queries][GB_LOGLOG_M] (rows of first-pass queries are written), or null
    uint32_t* loglog_hist;                                                     // [n_queries][GB_LOGLOG_HIST]: registers per value (first-pass queries): all cardinality() needs
    KwOut out;
};

// per-item launches: workgroup b serves the query q with first_block[q] <= b < first_block[q + 1] (one search per workgroup, shared through LDS);
// returns false for the threads behind the query's last item
template <int BLOCK = GB_THREADS>                               // BLOCK-thread workgroups: GB_THREADS / BLOCK of them share one GB_THREADS-item chunk
__device__ inline bool gb_item_of(const GbArgs& a, uint32_t& qi, uint64_t& item) {
    __shared__ uint32_t s_q;
    const uint32_t chunk = blockIdx.x / (GB_THREADS / BLOCK), sub = blockIdx.x % (GB_THREADS / BLOCK);
    if (threadIdx.x == 0) {
        uint32_t lo = 0, hi = a.n_queries;                      // the last query whose first_block <= chunk (queries without items share their successor's)
        while (lo + 1 < hi) { const uint32_t mid = (lo + hi) >> 1; if (a.gq[mid].first_block <= chunk) lo = mid; else hi = mid; }
        s_q = lo;
    }
    __syncthreads();
    qi = s_q;
    const uint32_t local = (chunk - a.gq[qi].first_block) * GB_THREADS + sub * BLOCK + threadIdx.x;
    item = a.gq[qi].item_begin + local;
    return local < a.gq[qi].n_items;
}

__device__ inline bool gb_rec_greater(const GbArgs& a, uint64_t x, uint64_t y) {       // KV::is_greater on records (global item indices); equal KVs of one document
    if (a.s0[x] != a.s0[y]) return a.s0[x] > a.s0[y];                                   // (met by two candidate combinations): the LATER combination wins — Topster::add
    if (a.s1[x] != a.s1[y]) return a.s1[x] > a.s1[y];                                   // replaces a KV unless the new one is smaller (include/topster.h:392-406)
    if (a.s2[x] != a.s2[y]) return a.s2[x] > a.s2[y];
    if (a.ids[x] != a.ids[y]) return a.ids[x] > a.ids[y];
    return a.pass[x] > a.pass[y];
}

// ---- q = * without filter / excluded ids: the id array of the query is 0 .. num_docs - 1 ----
__global__ __launch_bounds__(GB_THREADS) void gb_iota_kernel(GbArgs a) {
    uint32_t qi; uint64_t i;
    if (!gb_item_of(a, qi, i)) return;
    if (a.gq[qi].iota) ((uint32_t*)a.ids)[i] = (uint32_t)(i - a.gq[qi].item_begin);
}

// ---- scoring of every matched id ----
template <int TMAX>                                                          // 3: every query of the batch has <= 3 lists (a fifth of the registers of the 10-token form)
__global__ __launch_bounds__(TMAX <= 3 ? GB_THREADS : 64) void gb_score_kernel(IndexView ix, GbArgs a) {      // (the 10-token form holds 40 positions per thread: one wave per workgroup)
    uint32_t qi; uint64_t i;
    if (!gb_item_of<(TMAX <= 3 ? GB_THREADS : 64)>(a, qi, i)) return;
    const GbQuery g = a.gq[qi];
    uint32_t c = g.first_combo;                                              // the combination that met this id (<= 16 per user query)
    while (c + 1 < g.first_combo + g.n_combos && a.combo_begin[c + 1] <= i) c++;
    a.pass[i] = (uint8_t)(c - g.first_combo);
    const KwQueryDev& q = a.queries[c];
    const uint32_t seq_id = a.ids[i];
    ScoredHit h;
    if (g.wildcard) {
        h = sort_scores(ix, q, seq_id, 100, 0, false, 0.0f);                 // Index::search_wildcard: the text-match slot is the constant 100 (src/index.cpp:6728-6730)
    } else {
        const KwQueryMF& mf = a.mfs[c];
        uint32_t pos[TMAX * KW_MAX_FIELDS];
        uint32_t tokens_found = 0;
#pragma unroll
        for (int t = 0; t < TMAX; t++) {
            bool any = false;
#pragma unroll
            for (int f = 0; f < KW_MAX_FIELDS; f++) {
                uint32_t p = KW_NONE;
                if ((uint32_t)t < q.n_lists && (uint32_t)f < mf.n_fields && mf.list[t][f] != KW_NONE) {
                    uint32_t pp;
                    if (probe_list(ix, ix.lists[mf.list[t][f]], seq_id, pp)) { p = pp; any = true; }
                }
                pos[t * KW_MAX_FIELDS + f] = p;
            }
            tokens_found += any ? 1u : 0u;                                  // every required token (the id matched) + the dropped tokens it holds
        }
        uint32_t off_words = 0;
        const uint64_t agg = agg_score_mf<TMAX>(ix, q, mf, pos, tokens_found, off_words);
        h = sort_scores(ix, q, seq_id, agg, off_words);
    }
    a.s0[i] = h.s0; a.s1[i] = h.s1; a.s2[i] = h.s2;
    const uint32_t col = g.column;
    // the group column holds get_distinct_id's result per seq_id; a document beyond it has no value in any group_by field (src/index.cpp:7104-7111)
    a.dkey[i] = (col < ix.n_columns && seq_id < ix.column_len[col]) ? (unsigned long long)ix.columns[col][seq_id]
                                                                 : (g.group_missing_values ? 1ull : (unsigned long long)seq_id);
}

__device__ inline uint64_t gb_mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// ---- second pass over several candidate combinations: one record per document (its greatest KV, the later combination on ties) ----
__device__ inline uint32_t gb_doc_slot(const GbArgs& a, const GbQuery& g, uint32_t seq_id, bool claim) {
    uint32_t slot = (seq_id * 2654435761u) & g.tab_mask;
    for (;;) {
        uint32_t* kp = a.dkey32 + g.tab_off + slot;
        uint32_t cur = *kp;
        if (cur == GB_NONE && claim) { cur = atomicCAS(kp, GB_NONE, seq_id); if (cur == GB_NONE) cur = seq_id; }
        if (cur == seq_id) return slot;
        slot = (slot + 1) & g.tab_mask;
    }
}
__global__ __launch_bounds__(GB_THREADS) void gb_dedupe_kernel(GbArgs a) {
    uint32_t qi; uint64_t i;
    if (!gb_item_of(a, qi, i)) return;
    const GbQuery g = a.gq[qi];
    if (!g.dedupe) return;
    const uint32_t slot = gb_doc_slot(a, g, a.ids[i], true);
    const uint32_t me = (uint32_t)(i - g.item_begin);
    uint32_t* bp = a.dbest + g.tab_off + slot;
    uint32_t cur = atomicCAS(bp, GB_NONE, me);
    while (cur != GB_NONE && gb_rec_greater(a, i, g.item_begin + cur)) {
        const uint32_t prev = atomicCAS(bp, cur, me);
        if (prev == cur) break;
        cur = prev;
    }
}

// ---- group table: slot per distinct key, member count, best record ----
__global__ __launch_bounds__(GB_THREADS) void gb_insert_kernel(GbArgs a) {
    uint32_t qi; uint64_t i;
    bool active = gb_item_of(a, qi, i);                                      // (nobody leaves: the wave-wide steps below need every lane)
    const GbQuery g = a.gq[qi];
    if (active && g.dedupe && a.dbest[g.tab_off + gb_doc_slot(a, g, a.ids[i], false)] != (uint32_t)(i - g.item_begin)) { a.rslot[i] = GB_NONE; active = false; }   // not its document's record
    uint32_t slot = GB_NONE;
    if (active) {
        const unsigned long long key = a.dkey[i];
        if (key == GB_EMPTY) slot = g.tab_mask + 1;
        else {
            slot = (uint32_t)gb_mix(key) & g.tab_mask;
            for (;;) {
                unsigned long long* kp = a.hkey + g.tab_off + slot;
                unsigned long long cur = *kp;                               // (a slot's key never changes once set; a stale EMPTY is resolved by the CAS)
                if (cur == GB_EMPTY) { cur = atomicCAS(kp, GB_EMPTY, key); if (cur == GB_EMPTY) cur = key; }
                if (cur == key) break;
                slot = (slot + 1) & g.tab_mask;
            }
        }
        a.rslot[i] = slot;
    }
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t me = (uint32_t)(i - g.item_begin);
    // Skewed groups (a field with a handful of values over millions of matches): thousands of threads would hammer ONE slot's counter and best-record word
    // (measured: one group of 10M documents 117 ms in this kernel; same-address atomics retire at ~40 M/s). Two levels of combining in front of the table:
    //   * lanes of a wave that share a slot update it ONCE — the count by their number, the best record by their greatest (a wave-wide reduction of the
    //     Topster's comparison); slot after slot while the groups hold three lanes or more (at most 16), the rest one by one;
    //   * those updates go to a workgroup table in LDS (GB_WG_SLOTS = 2 x the workgroup's items: it cannot fill up), and every slot the workgroup met
    //     is updated in HBM once at the end (10 groups over 10M ids: 10.3 -> see profiles/r05/exp_groupby.txt).
    __shared__ uint32_t s_slot[GB_WG_SLOTS], s_cnt[GB_WG_SLOTS], s_best[GB_WG_SLOTS];
    for (uint32_t t = threadIdx.x; t < GB_WG_SLOTS; t += GB_THREADS) { s_slot[t] = GB_NONE; s_cnt[t] = 0; s_best[t] = GB_NONE; }
    __syncthreads();
    auto wg_put = [&](uint32_t sl, uint32_t times, uint32_t rec) {           // rec: LOCAL item index of the greatest of the `times` records
        uint32_t h = (sl * 2654435761u) >> (32 - GB_WG_SLOTS_LOG2);
        for (;;) {
            uint32_t cur = s_slot[h];
            if (cur == GB_NONE) { cur = atomicCAS(&s_slot[h], GB_NONE, sl); if (cur == GB_NONE) cur = sl; }
            if (cur == sl) break;
            h = (h + 1) & (GB_WG_SLOTS - 1);
        }
        atomicAdd(&s_cnt[h], times);
        uint32_t cur = atomicCAS(&s_best[h], GB_NONE, rec);
        while (cur != GB_NONE && gb_rec_greater(a, g.item_begin + rec, g.item_begin + cur)) {
            const uint32_t prev = atomicCAS(&s_best[h], cur, rec);
            if (prev == cur) break;
            cur = prev;
        }
    };
    bool pending = active;
    for (int round = 0; round < 16; round++) {
        const unsigned long long rem = __ballot(pending ? 1 : 0);
        if (!rem) break;                                                     // (wave-uniform)
        const uint32_t leader = (uint32_t)__ffsll((long long)rem) - 1;
        const uint32_t lslot = __shfl(slot, (int)leader, 64);
        const bool mine = pending && slot == lslot;
        const unsigned long long same = __ballot(mine ? 1 : 0);
        if (__popcll(same) < 3) break;                                       // (wave-uniform) the first waiting lane is nearly alone in its group: many groups, one by one below
        int64_t v0 = 0, v1 = 0, v2 = 0, vk = -1; uint32_t vp = 0, vi = 0;      // this lane's candidate (vk < 0: none)
        if (mine) { v0 = a.s0[i]; v1 = a.s1[i]; v2 = a.s2[i]; vk = (int64_t)a.ids[i]; vp = a.pass[i]; vi = me; }
        for (int d = 32; d > 0; d >>= 1) {
            const int64_t o0 = __shfl_xor(v0, d, 64), o1 = __shfl_xor(v1, d, 64), o2 = __shfl_xor(v2, d, 64), ok = __shfl_xor(vk, d, 64);
            const uint32_t op = __shfl_xor(vp, d, 64), oi = __shfl_xor(vi, d, 64);
            const bool og = ent_greater(o0, o1, o2, ok, v0, v1, v2, vk) || (ok >= 0 && ok == vk && o0 == v0 && o1 == v1 && o2 == v2 && op > vp);
            if (og) { v0 = o0; v1 = o1; v2 = o2; vk = ok; vp = op; vi = oi; }
        }
        if (lane == leader) wg_put(slot, (uint32_t)__popcll(same), vi);
        if (mine) pending = false;
    }
    if (pending) wg_put(slot, 1u, me);
    __syncthreads();
    // every slot this workgroup met: its count and its greatest record into the query's table (hbest holds a LOCAL item index; whoever holds a greater
    // record replaces it); the group's first members list its slot — one counter update per wave (every lane of a workgroup serves the same query)
    for (uint32_t t = threadIdx.x; t < GB_WG_SLOTS; t += GB_THREADS) {       // (workgroup-uniform trip count)
        const uint32_t sl = s_slot[t];
        bool first = false;
        if (sl != GB_NONE) {
            const uint32_t rec = s_best[t];
            first = atomicAdd(&a.hcount[g.tab_off + sl], s_cnt[t]) == 0;
            uint32_t* bp = a.hbest + g.tab_off + sl;
            uint32_t cur = atomicCAS(bp, GB_NONE, rec);
            while (cur != GB_NONE && gb_rec_greater(a, g.item_begin + rec, g.item_begin + cur)) {
                const uint32_t prev = atomicCAS(bp, cur, rec);
                if (prev == cur) break;
                cur = prev;
            }
        }
        const unsigned long long fm = __ballot(first ? 1 : 0);
        if (fm) {
            uint32_t base_at = 0;
            if (lane == (uint32_t)__ffsll((long long)fm) - 1) base_at = atomicAdd(&a.gcount[qi], (uint32_t)__popcll(fm));
            base_at = __shfl(base_at, __ffsll((long long)fm) - 1, 64);
            if (first) a.glist[g.item_begin + base_at + (uint32_t)__popcll(fm & ((1ull << lane) - 1))] = sl;
        }
    }
}

// ---- wyhash v5 (default secret, seed 0) of the decimal string of a uint64 = StringUtils::hash_wy(std::to_string(key)) (include/string_utils.h:316-320,
// include/wyhash_v5.h:69-94; the branches for 1..20 bytes) ----
__device__ inline uint64_t gb_wymum(uint64_t A, uint64_t B) { return __umul64hi(A, B) ^ (A * B); }
__device__ inline uint64_t gb_wymix(uint64_t A, uint64_t B) { return A ^ B ^ gb_wymum(A, B); }
__device__ inline uint64_t gb_hash_wy_decimal(uint64_t v) {
    const uint64_t P0 = 0xa0761d6478bd642full, P1 = 0xe7037ed1a0b428dbull, P4 = 0x1d8e4e27c47d124full, P5 = 0x72b22b96e169b471ull;
    // the string's bytes in three registers, little-endian (byte j of the string = byte j of w2:w1:w0): the digits come out last first, each one shifts
    // the image up by a byte — no byte array (a runtime-indexed one lives in scratch memory)
    uint64_t w0 = 0, w1 = 0, w2 = 0;
    uint32_t len = 0;
    do {
        const uint64_t q = v / 10;
        w2 = (w2 << 8) | (w1 >> 56); w1 = (w1 << 8) | (w0 >> 56); w0 = (w0 << 8) | (uint64_t)('0' + (uint32_t)(v - q * 10));
        v = q; len++;
    } while (v);
    auto rd8 = [&](uint32_t p) -> uint64_t {                               // _wyr8 at string offset p (p + 8 <= len <= 20)
        if (p == 0) return w0;
        if (p < 8) return (w0 >> (8 * p)) | (w1 << (64 - 8 * p));
        if (p == 8) return w1;
        return (w1 >> (8 * (p - 8))) | (w2 << (64 - 8 * (p - 8)));
    };
    const uint64_t seed = P4;                                               // seed 0 ^ secret[4]
    uint64_t h;
    if (len >= 8) {
        if (len <= 16) h = gb_wymix(rd8(0) ^ P0, rd8(len - 8) ^ seed);
        else h = gb_wymix(rd8(0) ^ P0, rd8(8) ^ seed) ^ gb_wymix(rd8(len - 16) ^ P1, rd8(len - 8) ^ seed);
    } else if (len >= 4) h = gb_wymix((w0 & 0xFFFFFFFFull) ^ P0, ((w0 >> (8 * (len - 4))) & 0xFFFFFFFFull) ^ seed);       // _wyr4(p), _wyr4(p + len - 4)
    else h = gb_wymix((((w0 & 0xFF) << 16) | (((w0 >> (8 * (len >> 1))) & 0xFF) << 8) | ((w0 >> (8 * (len - 1))) & 0xFF)) ^ P0, seed);     // _wyr3: p[0], p[len >> 1], p[len - 1]
    h = gb_wymum(h ^ len, P5);
    return h != ~0ull ? h : ~0ull - 1;
}

// ---- selection of the `capacity` best groups; first pass: the hits + the LogLogBeta registers ----
template <int CAP>
__global__ __launch_bounds__(GB_THREADS) void gb_select_kernel(GbArgs a) {
    __shared__ TopkLds<CAP, true> tk;
    __shared__ int64_t thr[4];
    __shared__ uint32_t s_cnt, s_have;
    __shared__ uint32_t regs[GB_LOGLOG_M / 4];                               // LogLogBeta registers, four per word
    __shared__ uint32_t hist[GB_LOGLOG_HIST];                                // ... and how many registers hold each value
    const uint32_t t = threadIdx.x;
    const uint32_t qi = blockIdx.x;
    const GbQuery g = a.gq[qi];
    const size_t gbase = (size_t)qi * a.g_stride;
    if (!g.run) { if (t == 0) { a.n_groups[qi] = 0; a.groups_total[qi] = 0; a.out.n_hits[qi] = 0; a.pw_count[qi] = 0; } return; }
    if (t == 0) { s_cnt = 0; s_have = 0; }
    for (uint32_t w = t; w < GB_LOGLOG_M / 4; w += GB_THREADS) regs[w] = 0;
    __syncthreads();
    const uint32_t n_used = a.gcount[qi];                                    // distinct keys of the pass (gb_insert_kernel listed their slots)
    for (uint32_t base = 0; base < n_used && !g.forced; base += GB_THREADS) {
        const bool have = base + t < n_used;
        uint32_t slot = 0;
        int64_t e0 = 0, e1 = 0, e2 = 0, ek = -1;
        if (have) {
            slot = a.glist[g.item_begin + base + t];
            const uint64_t b = g.item_begin + a.hbest[g.tab_off + slot];
            e0 = a.s0[b]; e1 = a.s1[b]; e2 = a.s2[b]; ek = (int64_t)a.ids[b];
        }
        // only entries that beat the current k-th best are appended; the buffer is compacted when THEY do not fit. Everybody reads the held count
        // BEFORE the counting barrier and appends after it (a count read next to other threads' appends splits the workgroup at the compaction's barriers)
        bool pass = have && (!s_have || ent_greater(e0, e1, e2, ek, thr[0], thr[1], thr[2], thr[3]));
        const uint32_t held = s_cnt;
        const uint32_t n_pass = (uint32_t)__syncthreads_count(pass ? 1 : 0);
        if (held + n_pass > (uint32_t)CAP) {
            topk_compact<CAP, true>(tk, &s_cnt, g.k, thr, &s_have);          // -> <= k entries, CAP >= k + GB_THREADS
            pass = pass && (!s_have || ent_greater(e0, e1, e2, ek, thr[0], thr[1], thr[2], thr[3]));
        }
        if (pass) {
            const uint32_t at = atomicAdd(&s_cnt, 1u);
            tk.s0[at] = e0; tk.s1[at] = e1; tk.s2[at] = e2; tk.key[at] = ek;
        }
        if (have) {
            if (g.first_pass) {
                // loglog_counter->add(std::to_string(distinct_key)) (include/topster.h:346, :408; loglogbeta.h:86-105): every distinct key, once
                const unsigned long long key = slot == g.tab_mask + 1 ? GB_EMPTY : a.hkey[g.tab_off + slot];
                const uint64_t x = gb_hash_wy_decimal(key);
                const uint32_t kreg = (uint32_t)(x >> 50);
                const uint64_t shifted = (x << 14) ^ 0x3FFFull;
                const uint32_t val = (uint32_t)__clzll((long long)shifted) + 1;          // (shifted >= 0x3FFF: never 0)
                uint32_t* w = &regs[kreg >> 2];
                const uint32_t sh = (kreg & 3) * 8;
                uint32_t old = *w;
                for (;;) {
                    if (((old >> sh) & 0xFFu) >= val) break;
                    const uint32_t nw = (old & ~(0xFFu << sh)) | (val << sh);
                    const uint32_t prev = atomicCAS(w, old, nw);
                    if (prev == old) break;
                    old = prev;
                }
            }
        }
        __syncthreads();
    }
    if (!g.forced) topk_compact<CAP, true>(tk, &s_cnt, g.k, thr, &s_have);
    const uint32_t n = g.forced ? g.n_forced : s_cnt;                        // min(k, groups), sorted descending — or the given groups, in the given order
    int msi = -1;
    for (int i = 0; i < 3; i++) if (i < (int)a.queries[g.first_combo].n_sort && a.queries[g.first_combo].sort_kind[i] == 0) msi = i;
    const uint32_t* qids = a.ids + g.item_begin;
    // the given groups (forced): returned group r is the r-th given key — looked up in the table this shard's matched documents built; a key without documents
    // here is returned empty (group_found = group_size = 0, no hit), so that slot r means the same group on every shard
    for (uint32_t r = t; g.forced && r < n; r += GB_THREADS) {
        const unsigned long long dk = a.forced_keys[g.forced_begin + r];
        uint32_t slot = GB_NONE;
        if (g.n_items) {
            if (dk == GB_EMPTY) { if (a.hcount[g.tab_off + g.tab_mask + 1]) slot = g.tab_mask + 1; }
            else {
                for (uint32_t s = (uint32_t)gb_mix(dk) & g.tab_mask;; s = (s + 1) & g.tab_mask) {      // (the table is at most half full: an empty slot ends the probe)
                    const unsigned long long cur = a.hkey[g.tab_off + s];
                    if (cur == dk) { slot = s; break; }
                    if (cur == GB_EMPTY) break;
                }
            }
        }
        a.g_dkey[gbase + r] = dk;
        if (slot == GB_NONE) { a.g_found[gbase + r] = 0; a.g_size[gbase + r] = 0; continue; }
        a.hrank[g.tab_off + slot] = r;
        const uint32_t members = a.hcount[g.tab_off + slot];
        a.g_found[gbase + r] = members;
        a.g_size[gbase + r] = g.first_pass ? 1u : (members < g.group_limit ? members : g.group_limit);
        if (g.first_pass) {
            const uint32_t lo = a.hbest[g.tab_off + slot];
            const uint64_t b = g.item_begin + lo;
            const size_t o = (size_t)qi * a.out.k_stride + r;
            const int64_t e0 = a.s0[b], e1 = a.s1[b], e2 = a.s2[b];
            a.out.keys[o] = a.ids[b];
            a.out.scores[o * 3 + 0] = e0; a.out.scores[o * 3 + 1] = e1; a.out.scores[o * 3 + 2] = e2;
            a.out.text_match[o] = msi == 0 ? e0 : (msi == 1 ? e1 : (msi == 2 ? e2 : 0));
            a.out.vector_distance[o] = -1.0f;
            a.out.match_score_index[o] = (int8_t)msi;
            a.out_qidx[o] = a.qidx_of_combo[g.first_combo + a.pass[b]];
        }
    }
    for (uint32_t r = t; !g.forced && r < n; r += GB_THREADS) {
        const uint32_t key = (uint32_t)tk.key[r];
        // the record of the entry: its id's position in a combination's ascending ids — the one its group's table names as the best record
        uint32_t lo = 0;
        for (uint32_t c = g.first_combo; c < g.first_combo + g.n_combos; c++) {
            const uint32_t b = (uint32_t)(a.combo_begin[c] - g.item_begin);
            uint32_t hi = (uint32_t)(a.combo_begin[c + 1] - g.item_begin);
            const uint32_t end = hi;
            lo = b;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (qids[mid] < key) lo = mid + 1; else hi = mid; }
            if (lo < end && qids[lo] == key) {
                const uint32_t sl = a.rslot[g.item_begin + lo];
                if (sl != GB_NONE && a.hbest[g.tab_off + sl] == lo) break;
            }
        }
        const uint32_t slot = a.rslot[g.item_begin + lo];
        a.hrank[g.tab_off + slot] = r;
        const uint32_t members = a.hcount[g.tab_off + slot];
        a.g_dkey[gbase + r] = a.dkey[g.item_begin + lo];
        a.g_found[gbase + r] = members;
        a.g_size[gbase + r] = g.first_pass ? 1u : (members < g.group_limit ? members : g.group_limit);
        if (g.first_pass) {
            const size_t o = (size_t)qi * a.out.k_stride + r;
            a.out.keys[o] = key;
            a.out.scores[o * 3 + 0] = tk.s0[r]; a.out.scores[o * 3 + 1] = tk.s1[r]; a.out.scores[o * 3 + 2] = tk.s2[r];
            a.out.text_match[o] = msi == 0 ? tk.s0[r] : (msi == 1 ? tk.s1[r] : (msi == 2 ? tk.s2[r] : 0));
            a.out.vector_distance[o] = -1.0f;
            a.out.match_score_index[o] = (int8_t)msi;
            a.out_qidx[o] = a.qidx_of_combo[g.first_combo + a.pass[g.item_begin + lo]];
        }
    }
    if (!g.first_pass) for (uint32_t r = t; r < n; r += GB_THREADS) regs[r] = a.g_found[gbase + r];       // (the register words are free in a second pass; n <= 1024 < 4096)
    __syncthreads();
    if (t == 0) {
        a.n_groups[qi] = n;
        a.groups_total[qi] = n_used;
        uint32_t hits = n;
        if (!g.first_pass) {
            uint32_t ofs = 0, pw = 0;
            hits = 0;
            for (uint32_t r = 0; r < n; r++) {
                const uint32_t members = regs[r];
                a.g_mofs[gbase + r] = ofs; a.g_mcur[gbase + r] = 0;
                ofs += members; hits += members < g.group_limit ? members : g.group_limit;
                // a big group's members are cut into chunks, one workgroup each (gb_chunk_kernel); the work list cannot overflow: a big group has more than
                // GB_BIG members, so there are at most n_items / GB_BIG of them and n_items / GB_CHUNK + that many chunks
                uint32_t nch = 0;
                if (members > GB_BIG) {
                    nch = (members + GB_CHUNK - 1) / GB_CHUNK;
                    if (pw + nch > g.pw_cap) nch = 0;                    // (cannot happen; a group left to the one-wave walk is still answered correctly)
                }
                a.g_pfirst[gbase + r] = pw; a.g_nchunk[gbase + r] = nch;
                for (uint32_t c = 0; c < nch; c++) a.pw_list[g.pw_begin + pw + c] = r | (c << 10);
                pw += nch;
            }
            a.pw_count[qi] = pw;
        } else a.pw_count[qi] = 0;
        a.out.n_hits[qi] = hits;
    }
    if (g.first_pass && !g.forced) {                                         // (a forced pass repeats a pass whose sketch the caller already holds)
        // what LogLogBeta::cardinality() reads: how many registers hold each value (the host sums 2^-value over them), and the registers themselves on request
        if (t < GB_LOGLOG_HIST) hist[t] = 0;
        __syncthreads();
        for (uint32_t w = t; w < GB_LOGLOG_M / 4; w += GB_THREADS) {
            const uint32_t v = regs[w];
            if (v == 0) { atomicAdd(&hist[0], 4u); continue; }
            for (int b = 0; b < 4; b++) { const uint32_t x = (v >> (8 * b)) & 0xFFu; atomicAdd(&hist[x < GB_LOGLOG_HIST ? x : GB_LOGLOG_HIST - 1], 1u); }
            if (a.loglog) ((uint32_t*)(a.loglog + (size_t)qi * GB_LOGLOG_M))[w] = v;          // (the rows are zeroed per batch: only non-zero words travel)
        }
        __syncthreads();
        if (t < GB_LOGLOG_HIST) a.loglog_hist[(size_t)qi * GB_LOGLOG_HIST + t] = hist[t];
    }
}

// ---- second pass: members of the selected groups ----
// Second pass: every record of a selected group takes its place in the group's member list (local item indices; any order — the member kernels rank them).
// One workgroup per GB_SCATTER_ITEMS consecutive items of ONE query (first_sblock). A group of millions of members would mean millions of adds on its
// cursor, and the cursors of a query share a few cache lines whose atomics retire one after the other (measured: 100 groups over 10M ids 5.8 ms in this
// kernel with one add per 256-item workgroup and group): lanes of a wave that share a group reserve together (group after group while they hold three
// lanes or more) in an LDS counter per group rank, the workgroup reserves each of its groups' ranges in HBM ONCE, and every item writes itself at
// range + its place inside the workgroup.
constexpr uint32_t GB_SCATTER_ROUNDS = 8, GB_SCATTER_ITEMS = GB_SCATTER_ROUNDS * GB_THREADS;
constexpr uint32_t GB_RANKS = 1024;                             // group ranks of a query (TSGPU_MAX_TOPK)
__global__ __launch_bounds__(GB_THREADS) void gb_scatter_kernel(GbArgs a) {
    __shared__ uint32_t s_q;
    __shared__ uint32_t s_cnt[GB_RANKS], s_base[GB_RANKS];
    if (threadIdx.x == 0) {
        uint32_t lo = 0, hi = a.n_queries;                      // the last query whose first_sblock <= blockIdx.x (queries without items share their successor's)
        while (lo + 1 < hi) { const uint32_t mid = (lo + hi) >> 1; if (a.gq[mid].first_sblock <= blockIdx.x) lo = mid; else hi = mid; }
        s_q = lo;
    }
    __syncthreads();
    const uint32_t qi = s_q;
    const GbQuery g = a.gq[qi];
    if (g.first_pass || !g.run) return;                                       // (workgroup-uniform)
    const uint32_t ng = a.n_groups[qi] < GB_RANKS ? a.n_groups[qi] : GB_RANKS;
    for (uint32_t t = threadIdx.x; t < ng; t += GB_THREADS) s_cnt[t] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t local0 = (blockIdx.x - g.first_sblock) * GB_SCATTER_ITEMS + threadIdx.x;
    uint32_t rk[GB_SCATTER_ROUNDS], off[GB_SCATTER_ROUNDS];
#pragma unroll
    for (uint32_t k = 0; k < GB_SCATTER_ROUNDS; k++) {
        const uint32_t local = local0 + k * GB_THREADS;
        uint32_t r = GB_NONE;
        if (local < g.n_items) {
            const uint32_t rs = a.rslot[g.item_begin + local];               // (GB_NONE with dedupe: not its document's record)
            if (rs != GB_NONE) r = a.hrank[g.tab_off + rs];                  // (GB_NONE: not a selected group)
        }
        uint32_t at = 0;
        bool pending = r != GB_NONE;
        for (int round = 0; round < 16; round++) {
            const unsigned long long rem = __ballot(pending ? 1 : 0);
            if (!rem) break;                                                 // (wave-uniform)
            const uint32_t leader = (uint32_t)__ffsll((long long)rem) - 1;
            const uint32_t lr = __shfl(r, (int)leader, 64);
            const bool mine = pending && r == lr;
            const unsigned long long same = __ballot(mine ? 1 : 0);
            if (__popcll(same) < 3) break;                                   // (wave-uniform) many groups: one by one below
            uint32_t first = 0;
            if (lane == leader) first = atomicAdd(&s_cnt[lr], (uint32_t)__popcll(same));
            first = __shfl(first, (int)leader, 64);
            if (mine) { at = first + (uint32_t)__popcll(same & ((1ull << lane) - 1)); pending = false; }
        }
        if (pending) at = atomicAdd(&s_cnt[r], 1u);
        rk[k] = r; off[k] = at;
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < ng; t += GB_THREADS)
        if (s_cnt[t]) s_base[t] = atomicAdd(&a.g_mcur[(size_t)qi * a.g_stride + t], s_cnt[t]);
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < GB_SCATTER_ROUNDS; k++)
        if (rk[k] != GB_NONE) a.members[g.item_begin + a.g_mofs[(size_t)qi * a.g_stride + rk[k]] + s_base[rk[k]] + off[k]] = local0 + k * GB_THREADS;
}

// one wave per (query, selected group): the group Topster's content in sort() order = its min(group_limit, members) greatest records, descending.
// Hit slot of the j-th KV of the r-th group: r * group_limit + j.
__global__ __launch_bounds__(GB_THREADS) void gb_members_kernel(GbArgs a) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t pair = (uint64_t)blockIdx.x * (GB_THREADS / 64) + (threadIdx.x >> 6);
    const uint32_t qi = (uint32_t)(pair / a.g_stride), r = (uint32_t)(pair % a.g_stride);
    if (qi >= a.n_queries) return;
    const GbQuery g = a.gq[qi];
    if (!g.run || g.first_pass || r >= a.n_groups[qi]) return;              // (wave-uniform)
    const size_t gi = (size_t)qi * a.g_stride + r;
    if (a.g_nchunk[gi]) return;                                              // a big group: gb_chunk_kernel
    const uint32_t cnt = a.g_found[gi], take = a.g_size[gi];
    const uint32_t* mem = a.members + g.item_begin + a.g_mofs[gi];
    const KwQueryDev& q = a.queries[g.first_combo];
    int msi = -1;
    for (int i = 0; i < 3; i++) if (i < (int)q.n_sort && q.sort_kind[i] == 0) msi = i;
    int64_t p0 = 0, p1 = 0, p2 = 0, pk = -1;                                 // the previous extraction (pk < 0: none yet)
    for (uint32_t j = 0; j < take; j++) {
        int64_t b0 = 0, b1 = 0, b2 = 0, bk = -1;                             // this lane's greatest record below the previous extraction (bk < 0: none)
        uint32_t bx = 0;                                                     // ... and which record it is (local item index)
        for (uint32_t m = lane; m < cnt; m += 64) {
            const uint64_t x = g.item_begin + mem[m];
            const int64_t c0 = a.s0[x], c1 = a.s1[x], c2 = a.s2[x], ck = (int64_t)a.ids[x];
            if (pk >= 0 && !ent_greater(p0, p1, p2, pk, c0, c1, c2, ck)) continue;
            if (ent_greater(c0, c1, c2, ck, b0, b1, b2, bk)) { b0 = c0; b1 = c1; b2 = c2; bk = ck; bx = mem[m]; }
        }
        for (int d = 32; d > 0; d >>= 1) {
            const int64_t o0 = __shfl_xor(b0, d, 64), o1 = __shfl_xor(b1, d, 64), o2 = __shfl_xor(b2, d, 64), ok = __shfl_xor(bk, d, 64);
            const uint32_t ox = __shfl_xor(bx, d, 64);
            if (ent_greater(o0, o1, o2, ok, b0, b1, b2, bk)) { b0 = o0; b1 = o1; b2 = o2; bk = ok; bx = ox; }
        }
        if (lane == 0) {
            const size_t o = (size_t)qi * a.out.k_stride + (size_t)r * g.group_limit + j;
            a.out.keys[o] = (uint64_t)bk;
            a.out.scores[o * 3 + 0] = b0; a.out.scores[o * 3 + 1] = b1; a.out.scores[o * 3 + 2] = b2;
            a.out.text_match[o] = msi == 0 ? b0 : (msi == 1 ? b1 : (msi == 2 ? b2 : 0));
            a.out.vector_distance[o] = -1.0f;
            a.out.match_score_index[o] = (int8_t)msi;
            a.out_qidx[o] = a.qidx_of_combo[g.first_combo + a.pass[g.item_begin + bx]];
        }
        p0 = b0; p1 = b1; p2 = b2; pk = bk;
    }
}

// ---- second pass, big groups: one workgroup per chunk of GB_CHUNK members keeps the chunk's group_limit greatest records (the LDS top-K buffer of the keyword
// kernels, k = group_limit) as a sorted partial list; the workgroup that finishes a group's LAST chunk (a ticket per group, device-scope release / acquire around
// it: the partial lists were written by workgroups on other XCDs) folds the group's partial lists the same way and writes the group's hits. One wave walking a
// group of 10M members cost 0.6 s at group_limit 3 and 7 s at 50 (tools/experiments/giant_group_probe.py); the chunks of such a group run on the whole chip. ----
__device__ inline uint32_t gb_record_of_key(const GbArgs& a, const GbQuery& g, uint32_t key) {       // local item index of the group member with this id
    if (g.n_combos > 1) return a.dbest[g.tab_off + gb_doc_slot(a, g, key, false)];              // (a second pass over several combinations keeps one record per document)
    const uint32_t* qids = a.ids + g.item_begin;
    uint32_t lo = 0, hi = g.n_items;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (qids[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

__global__ __launch_bounds__(GB_THREADS) void gb_chunk_kernel(GbArgs a) {
    __shared__ TopkLds<512, true> tk;
    __shared__ int64_t thr[4];
    __shared__ uint32_t s_cnt, s_have, s_q, s_last;
    const uint32_t t = threadIdx.x;
    if (t == 0) {
        uint32_t lo = 0, hi = a.n_queries;                                   // the last query whose pw_begin <= blockIdx.x
        while (lo + 1 < hi) { const uint32_t mid = (lo + hi) >> 1; if (a.gq[mid].pw_begin <= blockIdx.x) lo = mid; else hi = mid; }
        s_q = lo; s_cnt = 0; s_have = 0;
    }
    __syncthreads();
    const uint32_t qi = s_q;
    const GbQuery g = a.gq[qi];
    const uint32_t w = blockIdx.x - g.pw_begin;
    if (w >= g.pw_cap || w >= a.pw_count[qi]) return;                        // (workgroup-uniform)
    const uint32_t e = a.pw_list[g.pw_begin + w], r = e & 1023u, c = e >> 10;
    const size_t gi = (size_t)qi * a.g_stride + r;
    const uint32_t L = g.group_limit, members = a.g_found[gi];
    const uint32_t* mem = a.members + g.item_begin + a.g_mofs[gi];
    const uint32_t m0 = c * GB_CHUNK, m1 = m0 + GB_CHUNK < members ? m0 + GB_CHUNK : members;
    // stream [first, first + count) entries produced by `get` through the top-L buffer (the select kernel's loop)
    auto fold = [&](uint32_t count, auto get) {
        for (uint32_t base = 0; base < count; base += GB_THREADS) {
            const bool have = base + t < count;
            int64_t e0 = 0, e1 = 0, e2 = 0, ek = -1;
            const bool valid = have && get(base + t, e0, e1, e2, ek);
            bool pass = valid && (!s_have || ent_greater(e0, e1, e2, ek, thr[0], thr[1], thr[2], thr[3]));
            const uint32_t held = s_cnt;
            const uint32_t n_pass = (uint32_t)__syncthreads_count(pass ? 1 : 0);
            if (held + n_pass > 512u) {
                topk_compact<512, true>(tk, &s_cnt, L, thr, &s_have);
                pass = pass && (!s_have || ent_greater(e0, e1, e2, ek, thr[0], thr[1], thr[2], thr[3]));
            }
            if (pass) { const uint32_t at = atomicAdd(&s_cnt, 1u); tk.s0[at] = e0; tk.s1[at] = e1; tk.s2[at] = e2; tk.key[at] = ek; }
            __syncthreads();
        }
        topk_compact<512, true>(tk, &s_cnt, L, thr, &s_have);
    };
    fold(m1 - m0, [&](uint32_t j, int64_t& e0, int64_t& e1, int64_t& e2, int64_t& ek) {
        const uint64_t x = g.item_begin + mem[m0 + j];
        e0 = a.s0[x]; e1 = a.s1[x]; e2 = a.s2[x]; ek = (int64_t)a.ids[x];
        return true;
    });
    const uint32_t np = s_cnt;                                               // min(L, members of the chunk), sorted descending
    const size_t pb = g.pbuf_off + (size_t)w * L;
    for (uint32_t j = t; j < np; j += GB_THREADS) { a.pb_s0[pb + j] = tk.s0[j]; a.pb_s1[pb + j] = tk.s1[j]; a.pb_s2[pb + j] = tk.s2[j]; a.pb_key[pb + j] = tk.key[j]; }
    if (t == 0) a.pw_n[g.pw_begin + w] = np;
    __threadfence();                                                         // release: this chunk's partial list before its ticket
    __syncthreads();
    if (t == 0) s_last = atomicAdd(&a.g_ticket[gi], 1u) + 1 == a.g_nchunk[gi] ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();                                                         // acquire: the other chunks' partial lists
    // ---- the group's last chunk: fold the partial lists, write the hits ----
    const uint32_t pf = a.g_pfirst[gi], nch = a.g_nchunk[gi];
    if (t == 0) { s_cnt = 0; s_have = 0; }
    __syncthreads();
    fold(nch * L, [&](uint32_t j, int64_t& e0, int64_t& e1, int64_t& e2, int64_t& ek) {
        const uint32_t ch = j / L, k = j % L;
        if (k >= a.pw_n[g.pw_begin + pf + ch]) return false;               // (a chunk with fewer than L members)
        const size_t o = g.pbuf_off + (size_t)(pf + ch) * L + k;
        e0 = a.pb_s0[o]; e1 = a.pb_s1[o]; e2 = a.pb_s2[o]; ek = a.pb_key[o];
        return true;
    });
    const uint32_t take = s_cnt;                                             // = g_size: min(L, members)
    const KwQueryDev& q = a.queries[g.first_combo];
    int msi = -1;
    for (int i = 0; i < 3; i++) if (i < (int)q.n_sort && q.sort_kind[i] == 0) msi = i;
    for (uint32_t j = t; j < take && j < L; j += GB_THREADS) {
        const size_t o = (size_t)qi * a.out.k_stride + (size_t)r * L + j;
        a.out.keys[o] = (uint64_t)tk.key[j];
        a.out.scores[o * 3 + 0] = tk.s0[j]; a.out.scores[o * 3 + 1] = tk.s1[j]; a.out.scores[o * 3 + 2] = tk.s2[j];
        a.out.text_match[o] = msi == 0 ? tk.s0[j] : (msi == 1 ? tk.s1[j] : (msi == 2 ? tk.s2[j] : 0));
        a.out.vector_distance[o] = -1.0f;
        a.out.match_score_index[o] = (int8_t)msi;
        a.out_qidx[o] = a.qidx_of_combo[g.first_combo + a.pass[g.item_begin + gb_record_of_key(a, g, (uint32_t)tk.key[j])]];
    }
}
