// kw_plan.hip.h — the batch planner ON THE DEVICE (included by tsgpu.hip after kw_kernels.hip.h).
//
// plan_batch() (tsgpu.hip) turns a keyword batch into the launch tables on the host: per query the posting-list handles of its tokens
// (ART leaf -> list: here a flat term table), the probe order, the cut of the driver list into work items, a launch-order cost key, the
// heaviest-first table layout and the hit-buffer offsets. For the batch shape the server sends most — plain single-field queries without
// filter / hidden ids / dropped tokens / deadline — every one of those steps is data-parallel over the queries, and 10 000 of them cost
// the host ~0.5 ms on eight threads (more where N ranks share a node's cores): three small kernels do the same from a 64-byte record per
// query. ANY valid plan yields the same results (scores depend on the document only, sort keys are total orders: DESIGN.md §3.1), so the
// device planner reproduces the host planner's policy (chunk rule, cost model, heaviest first) but is not required to match it bit for bit.
// Batches with any other query shape keep the host planner.
//
//   kw_plan_resolve_kernel : one thread per query — term ids -> list handles (device mirror of HandleMaps::dense_handle), list lengths,
//                            probe order, the KwQueryDev record; batch totals by atomics (driver blocks for the chunk rule, bytes, max k)
//   kw_plan_chunk_kernel   : one thread per query — chunk length, work-item count, cost key; table totals
//   kw_plan_rank_kernel    : a query's place in the heaviest-first order of its table as a RANK computed against every other query's key (LDS
//                            tiles, the other queries cut into slices over blockIdx.y — no sort, no scan): prefix sums of the work-item counts
//                            and hit-buffer blocks in that order, accumulated with atomics
//   kw_plan_emit_kernel    : one thread per query — first_work / n_work, its work items and their hit offsets
#pragma once

namespace tsgpu {

struct KwPlanIn {                    // what the host pre-scan keeps of one tsgpu_kw_query (76 bytes, pinned staging -> one upload)
    uint32_t term_ids[TSGPU_MAX_QUERY_TOKENS];
    uint32_t field;
    int32_t weight;
    uint32_t k;                      // resolved Topster capacity
    uint32_t total_cost;
    uint8_t n_tokens, match_type, prio_bits /* 1 exact, 2 position, 4 num fields */, n_sort;
    uint8_t sort_kind[3];
    int8_t sort_order[3];
    uint16_t sort_col[3];
    int8_t syn_orig_num_tokens;
    uint8_t orig_num_tokens, is_synonym, demote_synonym;
};

static_assert(sizeof(KwPlanIn) == 76, "the comments (and DESIGN.md) quote the record's size");

struct KwPlanTotals {                // device-resident, read back by the host (twice: after resolve, after layout)
    unsigned long long total_best_blocks;   // sum over the queries of their shortest list's blocks (the auto chunk rule's input)
    unsigned long long list_bytes;          // 4 * sum |L_t|
    unsigned long long n_numeric_sort_q;
    uint32_t max_k, any_s2, fallback;       // fallback: a term id beyond the device term table / a query that needs merge groups
    uint32_t n_work[2];                     // work items of the two tables (<= 3 tokens / up to 10)
    unsigned long long hit_blocks[2];       // driver blocks of the two tables (x 256 = hit records)
    uint32_t pad[2];
};

struct KwPlanScratch {               // per query, between the kernels
    uint32_t* n_blocks;              // driver list's blocks (0: no work)
    uint32_t* len_a;                 // driver list ids
    uint32_t* len_b;                 // second-shortest list ids (0: none)
    uint32_t* cnt;                   // work items
    uint32_t* chunk;                 // driver blocks per work item
    unsigned long long* key;         // table << 32 | (0xFFFFFFFF - cost bits): ascending = table 0 first, heaviest first
};

struct KwPlanParams {
    uint32_t n_queries, num_docs, n_columns;
    uint32_t chunk_blocks_opt;       // kw_chunk_blocks (0 = auto)
    uint32_t max_partials, merge_select_min, max_chunk;
    float cost_fixed, cost_r, cost_probe;
    const uint32_t* dense;           // [64 x {offset, n}] then the per-field term -> handle tables (0xFFFFFFFF = absent)
    const ListDesc* lists;
};

__global__ __launch_bounds__(256) void kw_plan_resolve_kernel(KwPlanParams pp, const KwPlanIn* __restrict__ in, KwQueryDev* __restrict__ qout, KwPlanScratch sc,
                                                               KwPlanTotals* __restrict__ tot) {
    const uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i0 < pp.n_queries;                    // (idle lanes of the last wave re-plan the last query and contribute nothing: the wave reductions below need every lane)
    const uint32_t i = live ? i0 : pp.n_queries - 1;
    const KwPlanIn r = in[i];
    KwQueryDev q;
    {
        uint32_t* z = (uint32_t*)&q;
        for (uint32_t w = 0; w < sizeof(KwQueryDev) / 4; w++) z[w] = 0;
    }
    q.k = r.k;
    q.mf_index = KW_NONE;
    q.n_query_tokens = r.n_tokens;
    q.match_type = r.match_type;
    q.prio_exact = r.prio_bits & 1; q.prio_pos = (r.prio_bits >> 1) & 1; q.prio_nfields = (r.prio_bits >> 2) & 1;
    q.total_cost = r.total_cost;
    q.weight = r.weight;
    q.syn_orig_num_tokens = r.syn_orig_num_tokens; q.orig_num_tokens = r.orig_num_tokens; q.is_synonym = r.is_synonym; q.demote_synonym = r.demote_synonym;
    q.n_sort = r.n_sort;
    uint32_t n_num = 0;
    for (uint32_t s = 0; s < 3; s++) if (s < r.n_sort) {
        q.sort_kind[s] = r.sort_kind[s]; q.sort_order[s] = r.sort_order[s]; q.sort_col[s] = r.sort_col[s];
        if (r.sort_kind[s] == TSGPU_SORT_INT64_COLUMN) n_num++;
    }
    // tokens -> lists (a token the field does not hold is skipped, src/index.cpp:5651-5655)
    const uint32_t f_off = r.field < 64 ? pp.dense[2 * r.field] : 0, f_n = r.field < 64 ? pp.dense[2 * r.field + 1] : 0;
    uint32_t nl = 0, len_of[KW_MAX_TOKENS], nblk_of[KW_MAX_TOKENS];
    unsigned long long bytes = 0;
    bool fallback = false;
    for (uint32_t t = 0; t < r.n_tokens && t < (uint32_t)KW_MAX_TOKENS; t++) {
        const uint32_t term = r.term_ids[t];
        if (term >= (4u << 20)) { fallback = true; continue; }        // (terms beyond the flat tables live in the host's hash map only)
        if (term >= f_n) continue;
        const uint32_t h = pp.dense[128 + f_off + term];
        if (h == 0xFFFFFFFFu) continue;
        const ListDesc d = pp.lists[h];
        q.list[nl] = h;
        len_of[nl] = d.n_ids; nblk_of[nl] = d.n_blocks;
        bytes += 4ull * d.n_ids;
        nl++;
    }
    q.n_lists = nl; q.n_required = nl;
    // probe order: ascending list length, stable (insertion sort of <= 10 entries)
    uint8_t ord[KW_MAX_TOKENS];
    for (uint32_t t = 0; t < nl; t++) {
        uint32_t p = t;
        while (p > 0 && len_of[ord[p - 1]] > len_of[t]) { ord[p] = ord[p - 1]; p--; }
        ord[p] = (uint8_t)t;
    }
    for (uint32_t t = 0; t < nl; t++) q.probe_order[t] = ord[t];
    uint32_t best = 0;
    if (nl) { best = nblk_of[0]; for (uint32_t t = 1; t < nl; t++) best = nblk_of[t] < best ? nblk_of[t] : best; }
    if (live) {
        qout[i] = q;
        sc.n_blocks[i] = nl ? nblk_of[ord[0]] : 0;
        sc.len_a[i] = nl ? len_of[ord[0]] : 0;
        sc.len_b[i] = nl >= 2 ? len_of[ord[1]] : 0;
    }
    // batch totals: one atomic per wavefront and counter
    unsigned long long vb = live ? best : 0, vy = live ? bytes : 0, vn = (live && n_num) ? 1 : 0;
    uint32_t vk = live ? r.k : 0, vs = (live && r.n_sort > 2) ? 1u : 0u, vf = (live && fallback) ? 1u : 0u;
    for (int d = 32; d > 0; d >>= 1) {
        vb += __shfl_down(vb, d, 64); vy += __shfl_down(vy, d, 64); vn += __shfl_down(vn, d, 64);
        const uint32_t ok = __shfl_down(vk, d, 64); vk = ok > vk ? ok : vk;
        vs |= __shfl_down(vs, d, 64); vf |= __shfl_down(vf, d, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&tot->total_best_blocks, vb); atomicAdd(&tot->list_bytes, vy); atomicAdd(&tot->n_numeric_sort_q, vn);
        atomicMax(&tot->max_k, vk);
        if (vs) atomicOr(&tot->any_s2, 1u);
        if (vf) atomicOr(&tot->fallback, 1u);
    }
}

__global__ __launch_bounds__(256) void kw_plan_chunk_kernel(KwPlanParams pp, const KwQueryDev* __restrict__ q, KwPlanScratch sc, KwPlanTotals* __restrict__ tot) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < pp.n_queries;
    // the batch-wide chunk (plan_batch: a few thousand work items, >= 3 per resident workgroup slot, without fragmenting the queries)
    uint32_t CH = pp.chunk_blocks_opt;
    if (CH == 0) {
        const unsigned long long c = tot->total_best_blocks / 3000;
        CH = 16;
        while (CH < pp.max_chunk && (unsigned long long)CH * 2 <= c) CH *= 2;
    }
    uint32_t cnt = 0, nb = 0, tab = 0;
    if (live) {
        nb = sc.n_blocks[i];
        const uint32_t nl = q[i].n_lists;
        tab = nl <= 3 ? 0u : 1u;
        uint32_t chunk_q = CH;
        if (pp.chunk_blocks_opt == 0 && nb) {
            const uint32_t per = (nb + pp.max_partials - 1) / pp.max_partials;
            const uint32_t c2 = per < 256u ? per : 256u;
            chunk_q = chunk_q > c2 ? chunk_q : c2;
        }
        cnt = nb ? (nb + chunk_q - 1) / chunk_q : 0;
        // launch-order key: the query's LARGEST work item = driver blocks x (fixed + |B|/|A| + third-list probes of the stage-1 survivors)
        const float la = (float)(sc.len_a[i] ? sc.len_a[i] : 1u);
        float r = nl >= 2 ? (float)sc.len_b[i] / la : 0.0f;
        r = r < 64.0f ? r : 64.0f;
        const float surv = nl >= 3 ? 256.0f * (float)sc.len_b[i] / (float)(pp.num_docs ? pp.num_docs : 1u) : 0.0f;
        const float cost = (float)(chunk_q < nb ? chunk_q : nb) * (pp.cost_fixed + pp.cost_r * r + pp.cost_probe * surv);
        sc.cnt[i] = cnt;
        sc.chunk[i] = chunk_q;
        sc.key[i] = ((unsigned long long)tab << 32) | (0xFFFFFFFFu - __float_as_uint(cost));
        // more partial lists than the selecting merge takes (or the selecting merge switched off and more than two groups of eight): the
        // host planner's two-level merge groups are needed
        const bool need_groups = cnt > 16u && !(pp.merge_select_min && cnt >= pp.merge_select_min && cnt <= (uint32_t)KW_SEL_PMAX);
        if (need_groups) atomicOr(&tot->fallback, 1u);
    }
    uint32_t c0 = tab == 0 ? cnt : 0, c1 = tab == 1 ? cnt : 0;
    unsigned long long b0 = (live && tab == 0 && cnt) ? nb : 0, b1 = (live && tab == 1 && cnt) ? nb : 0;
    for (int d = 32; d > 0; d >>= 1) { c0 += __shfl_down(c0, d, 64); c1 += __shfl_down(c1, d, 64); b0 += __shfl_down(b0, d, 64); b1 += __shfl_down(b1, d, 64); }
    if ((threadIdx.x & 63) == 0) {
        if (c0) atomicAdd(&tot->n_work[0], c0);
        if (c1) atomicAdd(&tot->n_work[1], c1);
        if (b0) atomicAdd(&tot->hit_blocks[0], b0);
        if (b1) atomicAdd(&tot->hit_blocks[1], b1);
    }
}

// Place of every query in the heaviest-first order of its table, as a RANK: query i sums the work items (and, inside its table, the driver
// blocks) of every query whose key sorts ahead of its own. 10 000 x 10 000 compares: grid = (queries / 256) x KW_PLAN_JPARTS — each block
// takes 256 queries against ONE slice of the other queries (LDS tiles, broadcast reads) and adds its partial sums with two atomics per query;
// a single wave walking all 10 000 alone would be a 0.3 ms dependent loop.
static const int KW_PLAN_JPARTS = 32;
__global__ __launch_bounds__(256) void kw_plan_rank_kernel(KwPlanParams pp, KwPlanScratch sc, uint32_t* __restrict__ fw_acc, unsigned long long* __restrict__ hb_acc) {
    __shared__ unsigned long long s_key[256];
    __shared__ uint32_t s_cnt[256], s_nb[256];
    const uint32_t t = threadIdx.x, i = blockIdx.x * blockDim.x + t;
    const bool live = i < pp.n_queries;
    const unsigned long long my_key = live ? sc.key[i] : ~0ull;
    const uint32_t my_tab = (uint32_t)(my_key >> 32);
    const uint32_t per = ((pp.n_queries + KW_PLAN_JPARTS - 1) / KW_PLAN_JPARTS + 255) & ~255u;
    const uint32_t j_begin = blockIdx.y * per, j_end = j_begin + per < pp.n_queries ? j_begin + per : pp.n_queries;
    uint32_t fw = 0;                       // work items of the queries ahead of this one (both tables: first_work indexes their concatenation)
    unsigned long long hb = 0;             // driver blocks of the queries ahead of it IN ITS TABLE
    for (uint32_t j0 = j_begin; j0 < j_end; j0 += 256) {
        const uint32_t j = j0 + t;
        __syncthreads();
        s_key[t] = j < j_end ? sc.key[j] : ~0ull;
        s_cnt[t] = j < j_end ? sc.cnt[j] : 0u;
        s_nb[t] = j < j_end ? sc.n_blocks[j] : 0u;
        __syncthreads();
        const uint32_t n = j_end - j0 < 256u ? j_end - j0 : 256u;
        for (uint32_t e = 0; e < n; e++) {
            const unsigned long long kj = s_key[e];
            const bool ahead = kj < my_key || (kj == my_key && j0 + e < i);        // ties in query order (stable)
            const uint32_t c = s_cnt[e];
            fw += ahead ? c : 0u;
            hb += (ahead && c && (uint32_t)(kj >> 32) == my_tab) ? s_nb[e] : 0u;
        }
    }
    if (!live) return;
    if (fw) atomicAdd(&fw_acc[i], fw);
    if (hb) atomicAdd(&hb_acc[i], hb);
}

// work: the two tables back to back (table 1 starts at n_work[0]); hoff[w] = first hit record of work item w inside ITS table's hit buffer
__global__ __launch_bounds__(256) void kw_plan_emit_kernel(KwPlanParams pp, KwQueryDev* __restrict__ q, KwPlanScratch sc, const uint32_t* __restrict__ fw_acc,
                                                            const unsigned long long* __restrict__ hb_acc, KwWorkItem* __restrict__ work, unsigned long long* __restrict__ hoff,
                                                            const KwPlanTotals* __restrict__ tot) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pp.n_queries) return;
    const uint32_t cnt = sc.cnt[i], fw = fw_acc[i];
    const unsigned long long hb = hb_acc[i];
    q[i].first_work = fw; q[i].n_work = cnt; q[i].m_first = fw; q[i].m_n = cnt;
    // the id arena (batches that keep the matched ids): a query's segment sits where its hit records sit — driver blocks of the queries ahead of it in
    // its table, table 1 behind table 0 —, its work items' segments at their first block (ids_out_off below), 256 ids per driver block
    q[i].ids_out_off = (((sc.key[i] >> 32) ? tot->hit_blocks[0] : 0ull) + hb) * (unsigned long long)BLOCK_IDS;
    const uint32_t nb = sc.n_blocks[i], chunk = sc.chunk[i];
    for (uint32_t c = 0, b = 0; c < cnt; c++, b += chunk) {
        KwWorkItem w;
        w.query = i; w.blk_begin = b; w.blk_end = b + chunk < nb ? b + chunk : nb; w.ids_out_off = b * BLOCK_IDS;
        work[fw + c] = w;
        hoff[fw + c] = (hb + b) * (unsigned long long)BLOCK_IDS;
    }
}

}  // namespace tsgpu
