// tsgpu_facet.hip — facet counting over matched ids behind include/tsgpu.h: the hash-index branch of
// Index::do_facets (src/index.cpp:1659-1771). The host only sizes the per-query tables and orders the (few) distinct values the
// device found; the per-document work runs in facet_kernels.hip.h.
#include "tsgpu_host.h"
#include <limits>
#include "facet_kernels.hip.h"

using namespace tsgpu;

namespace tsgpu {
struct FacetField {
    DevBuf doc_ptr, hashes;
    uint32_t n_docs = 0;
    uint64_t n_hashes = 0;
    uint32_t n_distinct = 0, max_per_doc = 0;
    DevBuf d_ids, d_queries, d_allowed, d_key, d_cnt, d_last, d_oh, d_oc, d_od, d_op, d_on;      // per-batch scratch
    DevBuf val_ptr, val_ids, val_total;              // value index (tsgpu_facet_value_set): value -> ascending seq_ids, in the reference's visiting order
    uint32_t n_values = 0;
    DevBuf d_stats, d_map_h, d_map_v, d_counts, d_order;
    PinBuf h_in, h_out;                               // staged id lists of a multi-query call (ONE upload), the compacted lists (pinned: no page-by-page DMA)
    DevBuf d_sh, d_sc, d_sd, d_sp;                    // the compacted lists in hash order (facet_sort_kernel)
    DevBuf d_gcnt, d_pair, d_pair_ones, d_rng_up, d_rng_lo, d_rng_set, d_rng_cnt, d_rng_gcnt;      // grouped counting / range facets
    void release() {
        DevBuf* b[] = {&doc_ptr, &hashes, &d_ids, &d_queries, &d_allowed, &d_key, &d_cnt, &d_last, &d_oh, &d_oc, &d_od, &d_op, &d_on,
                       &val_ptr, &val_ids, &val_total, &d_stats, &d_map_h, &d_map_v, &d_counts, &d_order,
                       &d_sh, &d_sc, &d_sd, &d_sp, &d_gcnt, &d_pair, &d_pair_ones, &d_rng_up, &d_rng_lo, &d_rng_set, &d_rng_cnt, &d_rng_gcnt};
        for (auto* x : b) x->release();
        h_in.release(); h_out.release();
    }
};
// the id lists of a call -> the id arena (d_ids reserved by the caller). Many small lists (a keyword batch's per-query ids) are packed into one pinned block
// and cross in ONE copy (1 000 lists: 1 000 copies of ~7 KB cost ~10 ms before); a single list, or lists of megabytes, go straight from the caller's memory.
static int facet_upload_ids(FacetField* f, const std::vector<FacetQueryDev>& qd, const uint32_t* const* result_ids, uint32_t n_queries, uint64_t ids_total, hipStream_t s) {
    if (n_queries > 1 && ids_total && ids_total / n_queries < (256u << 10) && ids_total <= (64u << 20)) {        // (at most 256 MB of pinned staging)
        int rc = f->h_in.reserve(ids_total * 4);
        if (rc) return rc;
        // (every facet call ends with a stream synchronisation under ctx->mu: the block is not feeding an earlier copy any more)
        for (uint32_t q = 0; q < n_queries; q++)
            if (qd[q].n_ids) memcpy(f->h_in.as<uint32_t>() + qd[q].ids_off, result_ids[q], qd[q].n_ids * 4);
        TSGPU_HIP_TRY(hipMemcpyAsync(f->d_ids.p, f->h_in.p, ids_total * 4, hipMemcpyHostToDevice, s));
        return TSGPU_OK;
    }
    for (uint32_t q = 0; q < n_queries; q++)
        if (qd[q].n_ids) TSGPU_HIP_TRY(hipMemcpyAsync(f->d_ids.as<uint32_t>() + qd[q].ids_off, result_ids[q], qd[q].n_ids * 4, hipMemcpyHostToDevice, s));
    return TSGPU_OK;
}
}  // namespace tsgpu

extern "C" {

void tsgpu_facet_destroy_all(tsgpu_ctx* ctx) {
    for (auto& kv : ctx->facet_fields) { kv.second->release(); delete kv.second; }
    ctx->facet_fields.clear();
}

int tsgpu_facet_set(tsgpu_ctx* ctx, uint32_t facet_field_id, const uint64_t* doc_ptr, const uint32_t* hashes, uint32_t n_docs) {
    if (!ctx || !doc_ptr) return fail(TSGPU_ERR_INVALID, "tsgpu_facet_set: NULL argument");
    for (uint32_t d = 0; d < n_docs; d++) if (doc_ptr[d + 1] < doc_ptr[d]) return fail(TSGPU_ERR_INVALID, "tsgpu_facet_set: doc_ptr must be non-decreasing");
    const uint64_t n_h = doc_ptr[n_docs] - doc_ptr[0];
    if (doc_ptr[0] != 0) return fail(TSGPU_ERR_INVALID, "tsgpu_facet_set: doc_ptr[0] must be 0");
    if (n_h && !hashes) return fail(TSGPU_ERR_INVALID, "tsgpu_facet_set: hashes is NULL");
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    try {
        FacetField* f;
        auto it = ctx->facet_fields.find(facet_field_id);
        if (it == ctx->facet_fields.end()) { f = new FacetField; ctx->facet_fields[facet_field_id] = f; } else f = it->second;
        // build next to the old arrays and swap on success
        DevBuf np, nh;
        int rc;
        if ((rc = np.reserve(((size_t)n_docs + 1) * 8)) || (rc = nh.reserve(std::max<uint64_t>(n_h, 1) * 4))) { np.release(); nh.release(); return rc; }
        hipError_t e = hipMemcpy(np.p, doc_ptr, ((size_t)n_docs + 1) * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess && n_h) e = hipMemcpy(nh.p, hashes, n_h * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) { np.release(); nh.release(); return fail(TSGPU_ERR_DEVICE, std::string("tsgpu_facet_set: ") + hipGetErrorString(e)); }
        std::vector<uint32_t> tmp(hashes, hashes + n_h);
        std::sort(tmp.begin(), tmp.end());
        const uint32_t distinct = (uint32_t)(std::unique(tmp.begin(), tmp.end()) - tmp.begin());
        uint32_t mx = 0;
        for (uint32_t d = 0; d < n_docs; d++) mx = std::max<uint32_t>(mx, (uint32_t)std::min<uint64_t>(doc_ptr[d + 1] - doc_ptr[d], 0xFFFFFFFFull));
        f->doc_ptr.release(); f->hashes.release();
        f->doc_ptr = np; f->hashes = nh;
        f->n_docs = n_docs; f->n_hashes = n_h; f->n_distinct = distinct; f->max_per_doc = mx;
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_facet_set: host allocation failed"); }
    return ok();
}

static int facet_count_impl(tsgpu_ctx* ctx, uint32_t facet_field_id, const uint32_t* const* result_ids, const uint64_t* n_result_ids, uint32_t n_queries,
                            uint32_t sample_mod, const uint32_t* allowed_hashes, uint32_t n_allowed, bool grouped, uint32_t group_column, bool group_missing_values,
                            tsgpu_facet_counts* out) {
    if (!ctx || !out || (n_queries && (!result_ids || !n_result_ids))) return fail(TSGPU_ERR_INVALID, "tsgpu_facet_count_batch: NULL argument");
    if (n_queries == 0) return ok();
    if (!out->hash || !out->count || !out->doc_id || !out->array_pos || !out->n_values) return fail(TSGPU_ERR_INVALID, "tsgpu_facet_count_batch: missing output arrays");
    if (n_allowed && !allowed_hashes) return fail(TSGPU_ERR_INVALID, "tsgpu_facet_count_batch: allowed_hashes is NULL");
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    auto it = ctx->facet_fields.find(facet_field_id);
    if (it == ctx->facet_fields.end()) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_facet_count_batch: unknown facet field (tsgpu_facet_set)");
    FacetField* f = it->second;
    hipStream_t s = ctx->stream;
    if (sample_mod == 0) sample_mod = 1;
    if (grouped && group_column >= ctx->columns.size()) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_facet_count_grouped_batch: unknown group column (tsgpu_column_set)");
    try {
        std::vector<FacetQueryDev> qd(n_queries);
        uint64_t ids_total = 0, tab_total = 0, out_total = 0, max_size = 64, pair_total = 0;
        uint32_t blocks = 0;
        // ids per workgroup: the more ids a workgroup folds in LDS before it touches the query's table the fewer atomics meet there, as long as a
        // couple of thousand workgroups remain to fill the 256 CUs
        uint64_t all_ids = 0;
        for (uint32_t q = 0; q < n_queries; q++) all_ids += n_result_ids[q];
        uint32_t ids_per_block = FACET_THREADS;
        while (ids_per_block < 4096 && all_ids / (ids_per_block * 2) >= 2048) ids_per_block *= 2;
        if (ctx->facet_ids_per_block) ids_per_block = ctx->facet_ids_per_block;
        for (uint32_t q = 0; q < n_queries; q++) {
            if (n_result_ids[q] && !result_ids[q]) return fail(TSGPU_ERR_INVALID, "tsgpu_facet_count_batch: result_ids[q] is NULL");
            FacetQueryDev& d = qd[q];
            d.ids_off = ids_total; d.n_ids = n_result_ids[q];
            ids_total += d.n_ids;
            // distinct values this query can meet: every value of the field, or what its (sampled) documents can hold at most
            const uint64_t sampled = (d.n_ids + sample_mod - 1) / sample_mod;
            uint64_t distinct = std::min<uint64_t>(n_allowed ? std::min<uint64_t>(n_allowed, f->n_distinct) : f->n_distinct, sampled * std::max<uint32_t>(f->max_per_doc, 1));
            uint64_t size = 64;
            while (size < 2 * distinct) size <<= 1;
            if (size > (1ull << 31)) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_facet_count_batch: more than 2^30 distinct facet values");
            d.tab_off = tab_total; d.tab_mask = (uint32_t)(size - 1);
            tab_total += size;
            d.out_off = out_total;
            out_total += size / 2 + 1;
            d.first_block = blocks;
            max_size = std::max(max_size, size);
            d.pair_off = pair_total; d.pair_mask = 0;
            if (grouped) {                                       // every (document, hash) of the query may be a pair of its own
                uint64_t psize = 64;
                while (psize < 2 * sampled * std::max<uint32_t>(f->max_per_doc, 1)) psize <<= 1;
                if (psize > (1ull << 31)) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_facet_count_grouped_batch: more than 2^30 (value, group) pairs in one query");
                d.pair_mask = (uint32_t)(psize - 1);
                pair_total += psize;
            }
            const uint64_t nb = (d.n_ids + ids_per_block - 1) / ids_per_block;
            if ((uint64_t)blocks + nb > 0x7FFFFFFFull) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_facet_count_batch: more than 2^31 workgroups");
            blocks += (uint32_t)nb;
        }
        if (tab_total * 24 + pair_total * 8 > (32ull << 30)) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_facet_count_batch: the counting tables of this batch exceed 32 GiB; split it");
        int rc;
        if (grouped && ((rc = f->d_gcnt.reserve(tab_total * 4)) || (rc = f->d_pair.reserve(std::max<uint64_t>(pair_total, 1) * 8)) || (rc = f->d_pair_ones.reserve((size_t)n_queries * 4)))) return rc;
        if ((rc = f->d_ids.reserve(std::max<uint64_t>(ids_total, 1) * 4)) || (rc = f->d_queries.reserve(qd.size() * sizeof(FacetQueryDev))) ||
            (rc = f->d_key.reserve(tab_total * 8)) || (rc = f->d_cnt.reserve(tab_total * 4)) || (rc = f->d_last.reserve(tab_total * 8)) ||
            (rc = f->d_oh.reserve(out_total * 4)) || (rc = f->d_oc.reserve(out_total * 4)) || (rc = f->d_od.reserve(out_total * 4)) ||
            (rc = f->d_op.reserve(out_total * 4)) || (rc = f->d_on.reserve((size_t)n_queries * 4)) || (rc = f->d_allowed.reserve(std::max<uint32_t>(n_allowed, 1) * 4)) ||
            (rc = f->d_sh.reserve(out_total * 4)) || (rc = f->d_sc.reserve(out_total * 4)) || (rc = f->d_sd.reserve(out_total * 4)) || (rc = f->d_sp.reserve(out_total * 4)))
            return rc;
        if ((rc = facet_upload_ids(f, qd, result_ids, n_queries, ids_total, s))) return rc;
        TSGPU_HIP_TRY(hipMemcpyAsync(f->d_queries.p, qd.data(), qd.size() * sizeof(FacetQueryDev), hipMemcpyHostToDevice, s));
        if (n_allowed) TSGPU_HIP_TRY(hipMemcpyAsync(f->d_allowed.p, allowed_hashes, (size_t)n_allowed * 4, hipMemcpyHostToDevice, s));
        TSGPU_HIP_TRY(hipMemsetAsync(f->d_key.p, 0, tab_total * 8, s));
        TSGPU_HIP_TRY(hipMemsetAsync(f->d_cnt.p, 0, tab_total * 4, s));
        TSGPU_HIP_TRY(hipMemsetAsync(f->d_last.p, 0, tab_total * 8, s));
        TSGPU_HIP_TRY(hipMemsetAsync(f->d_on.p, 0, (size_t)n_queries * 4, s));
        if (grouped) {
            TSGPU_HIP_TRY(hipMemsetAsync(f->d_gcnt.p, 0, tab_total * 4, s));
            TSGPU_HIP_TRY(hipMemsetAsync(f->d_pair.p, 0xFF, std::max<uint64_t>(pair_total, 1) * 8, s));
            TSGPU_HIP_TRY(hipMemsetAsync(f->d_pair_ones.p, 0, (size_t)n_queries * 4, s));
        }
        FacetArgs a;
        a.doc_ptr = f->doc_ptr.as<uint64_t>(); a.hashes = f->hashes.as<uint32_t>(); a.n_docs = f->n_docs;
        a.ids = f->d_ids.as<uint32_t>(); a.queries = f->d_queries.as<FacetQueryDev>(); a.n_queries = n_queries; a.sample_mod = sample_mod; a.ids_per_block = ids_per_block;
        a.allowed = n_allowed ? f->d_allowed.as<uint32_t>() : nullptr; a.n_allowed = n_allowed;
        a.tab_key = f->d_key.as<unsigned long long>(); a.tab_cnt = f->d_cnt.as<uint32_t>(); a.tab_last = f->d_last.as<unsigned long long>();
        a.out_hash = f->d_oh.as<uint32_t>(); a.out_cnt = f->d_oc.as<uint32_t>(); a.out_doc = f->d_od.as<uint32_t>(); a.out_pos = f->d_op.as<uint32_t>();
        a.out_n = f->d_on.as<uint32_t>();
        a.grouped = grouped ? 1u : 0u; a.group_missing_values = group_missing_values ? 1u : 0u;
        a.group_col = grouped ? ctx->columns[group_column].data.as<long long>() : nullptr; a.group_len = grouped ? ctx->columns[group_column].n : 0;
        a.pair_key = grouped ? f->d_pair.as<unsigned long long>() : nullptr; a.pair_ones = grouped ? f->d_pair_ones.as<uint32_t>() : nullptr;
        a.tab_gcnt = grouped ? f->d_gcnt.as<uint32_t>() : nullptr;
        for (int e = 0; e < 3; e++) if (!ctx->aux_ev[e]) TSGPU_HIP_TRY(hipEventCreate(&ctx->aux_ev[e]));      // (tsgpu_last_aux_timings)
        TSGPU_HIP_TRY(hipEventRecord(ctx->aux_ev[0], s));
        if (blocks) hipLaunchKernelGGL(facet_count_kernel, dim3(blocks), dim3(FACET_THREADS), 0, s, a);
        TSGPU_HIP_TRY(hipEventRecord(ctx->aux_ev[1], s));
        hipLaunchKernelGGL(facet_compact_kernel, dim3(n_queries, (uint32_t)std::min<uint64_t>((max_size + FACET_COMPACT_SLOTS - 1) / FACET_COMPACT_SLOTS, 4096)), dim3(FACET_THREADS), 0, s, a);
        FacetSortOut so;
        so.hash = f->d_sh.as<uint32_t>(); so.cnt = f->d_sc.as<uint32_t>(); so.doc = f->d_sd.as<uint32_t>(); so.pos = f->d_sp.as<uint32_t>();
        hipLaunchKernelGGL(facet_sort_kernel, dim3(n_queries), dim3(FACET_THREADS), 0, s, a, so);
        TSGPU_HIP_TRY(hipEventRecord(ctx->aux_ev[2], s));
        TSGPU_HIP_TRY(hipGetLastError());
        if ((rc = f->h_out.reserve(((size_t)n_queries + 4 * out_total) * 4))) return rc;
        uint32_t* hn = f->h_out.as<uint32_t>();
        uint32_t *hh = hn + n_queries, *hc = hh + out_total, *hd = hc + out_total, *hp = hd + out_total;
        TSGPU_HIP_TRY(hipMemcpyAsync(hn, f->d_on.p, (size_t)n_queries * 4, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(hh, f->d_sh.p, out_total * 4, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(hc, f->d_sc.p, out_total * 4, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(hd, f->d_sd.p, out_total * 4, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(hp, f->d_sp.p, out_total * 4, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipStreamSynchronize(s));
        {
            float cnt = 0, all = 0;
            (void)hipEventElapsedTime(&cnt, ctx->aux_ev[0], ctx->aux_ev[1]); (void)hipEventElapsedTime(&all, ctx->aux_ev[0], ctx->aux_ev[2]);
            std::lock_guard<std::mutex> tl(ctx->tm_mu);
            tsgpu_aux_timings& A = ctx->aux_timings;
            A.facet_kernels_ms = all; A.facet_count_ms = cnt; A.facet_ids = ids_total; A.facet_table_slots = tab_total;
            // per id: the id + its doc_ptr pair + (on average n_hashes / n_docs) value hashes; per table slot: key 8 + count 4 + last 8 cleared, then compacted
            A.facet_algorithmic_bytes = ids_total * 20ull + (uint64_t)((double)ids_total * (f->n_docs ? (double)f->n_hashes / f->n_docs : 0.0)) * 4ull + tab_total * 20ull;
        }
        // the distinct values of a query in ascending hash order (result_map is keyed by the hash); the first `cap` of them are returned. The device ordered every
        // list of up to FACET_SORT_MAX values; a longer one arrives as the table held it and only its first `cap` hashes are put in order here
        // (hash << 32 | position in the device list)
        std::vector<uint64_t> order;
        for (uint32_t q = 0; q < n_queries; q++) {
            const uint32_t n = hn[q];
            const uint64_t base = qd[q].out_off;
            out->n_values[q] = n;
            const uint32_t m = std::min(n, out->cap);
            const size_t o = (size_t)q * out->cap;
            if (n <= FACET_SORT_MAX) {
                memcpy(out->hash + o, hh + base, (size_t)m * 4); memcpy(out->count + o, hc + base, (size_t)m * 4);
                memcpy(out->doc_id + o, hd + base, (size_t)m * 4); memcpy(out->array_pos + o, hp + base, (size_t)m * 4);
                continue;
            }
            order.resize(n);
            for (uint32_t i = 0; i < n; i++) order[i] = ((uint64_t)hh[base + i] << 32) | i;
            if (m < n) std::nth_element(order.begin(), order.begin() + m, order.end());
            std::sort(order.begin(), order.begin() + m);
            for (uint32_t i = 0; i < m; i++) {
                const uint32_t at = (uint32_t)order[i];
                out->hash[o + i] = hh[base + at]; out->count[o + i] = hc[base + at]; out->doc_id[o + i] = hd[base + at]; out->array_pos[o + i] = hp[base + at];
            }
        }
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_facet_count_batch: host allocation failed"); }
    return ok();
}

int tsgpu_facet_count_batch(tsgpu_ctx* ctx, uint32_t facet_field_id, const uint32_t* const* result_ids, const uint64_t* n_result_ids, uint32_t n_queries,
                            uint32_t sample_mod, const uint32_t* allowed_hashes, uint32_t n_allowed, tsgpu_facet_counts* out) {
    return facet_count_impl(ctx, facet_field_id, result_ids, n_result_ids, n_queries, sample_mod, allowed_hashes, n_allowed, false, 0, false, out);
}

int tsgpu_facet_count_grouped_batch(tsgpu_ctx* ctx, uint32_t facet_field_id, const uint32_t* const* result_ids, const uint64_t* n_result_ids, uint32_t n_queries,
                                    uint32_t sample_mod, const uint32_t* allowed_hashes, uint32_t n_allowed, uint32_t group_column, int group_missing_values,
                                    tsgpu_facet_counts* out) {
    return facet_count_impl(ctx, facet_field_id, result_ids, n_result_ids, n_queries, sample_mod, allowed_hashes, n_allowed, true, group_column, group_missing_values != 0, out);
}

int tsgpu_facet_range_count_batch(tsgpu_ctx* ctx, uint32_t facet_field_id, uint32_t value_column, const int64_t* range_upper, const int64_t* range_lower, uint32_t n_ranges,
                                  const uint32_t* const* result_ids, const uint64_t* n_result_ids, uint32_t n_queries, uint32_t sample_mod,
                                  uint32_t group_column, int group_missing_values, uint32_t* counts) {
    if (!ctx || !counts || !range_upper || !range_lower || (n_queries && (!result_ids || !n_result_ids))) return fail(TSGPU_ERR_INVALID, "tsgpu_facet_range_count_batch: NULL argument");
    if (n_ranges == 0 || n_ranges > FACET_MAX_RANGES) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_facet_range_count_batch: 1 .. 1024 ranges");
    for (uint32_t r = 1; r < n_ranges; r++)
        if (range_upper[r] <= range_upper[r - 1]) return fail(TSGPU_ERR_INVALID, "tsgpu_facet_range_count_batch: range_upper must be strictly ascending (facet_range_map is a std::map)");
    if (n_queries == 0) return ok();
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    auto it = ctx->facet_fields.find(facet_field_id);
    if (it == ctx->facet_fields.end()) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_facet_range_count_batch: unknown facet field (tsgpu_facet_set)");
    FacetField* f = it->second;
    if (value_column >= ctx->columns.size()) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_facet_range_count_batch: unknown value column (tsgpu_column_set)");
    const bool grouped = group_column != TSGPU_NO_COLUMN;
    if (grouped && group_column >= ctx->columns.size()) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_facet_range_count_batch: unknown group column (tsgpu_column_set)");
    hipStream_t s = ctx->stream;
    if (sample_mod == 0) sample_mod = 1;
    try {
        std::vector<FacetQueryDev> qd(n_queries);
        uint64_t ids_total = 0, pair_total = 0, all_ids = 0;
        uint32_t blocks = 0;
        for (uint32_t q = 0; q < n_queries; q++) all_ids += n_result_ids[q];
        uint32_t ids_per_block = FACET_THREADS;
        while (ids_per_block < 4096 && all_ids / (ids_per_block * 2) >= 2048) ids_per_block *= 2;
        if (ctx->facet_ids_per_block) ids_per_block = ctx->facet_ids_per_block;
        for (uint32_t q = 0; q < n_queries; q++) {
            if (n_result_ids[q] && !result_ids[q]) return fail(TSGPU_ERR_INVALID, "tsgpu_facet_range_count_batch: result_ids[q] is NULL");
            FacetQueryDev& d = qd[q];
            memset(&d, 0, sizeof(d));
            d.ids_off = ids_total; d.n_ids = n_result_ids[q];
            ids_total += d.n_ids;
            d.first_block = blocks;
            const uint64_t nb = (d.n_ids + ids_per_block - 1) / ids_per_block;
            if ((uint64_t)blocks + nb > 0x7FFFFFFFull) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_facet_range_count_batch: more than 2^31 workgroups");
            blocks += (uint32_t)nb;
            d.pair_off = pair_total;
            if (grouped) {                                       // one pair per document at most
                const uint64_t sampled = (d.n_ids + sample_mod - 1) / sample_mod;
                uint64_t psize = 64;
                while (psize < 2 * sampled) psize <<= 1;
                d.pair_mask = (uint32_t)(psize - 1);
                pair_total += psize;
            }
        }
        // hash_groups is keyed by (uint32) range_id: ranges whose upper bounds agree in the low 32 bits share one set of groups
        std::vector<uint32_t> rset(n_ranges);
        for (uint32_t r = 0; r < n_ranges; r++) {
            rset[r] = r;
            for (uint32_t p = 0; p < r; p++) if ((uint32_t)(uint64_t)range_upper[p] == (uint32_t)(uint64_t)range_upper[r]) { rset[r] = p; break; }
        }
        const size_t cells = (size_t)n_queries * n_ranges;
        int rc;
        if ((rc = f->d_ids.reserve(std::max<uint64_t>(ids_total, 1) * 4)) || (rc = f->d_queries.reserve(qd.size() * sizeof(FacetQueryDev))) ||
            (rc = f->d_rng_up.reserve((size_t)n_ranges * 8)) || (rc = f->d_rng_lo.reserve((size_t)n_ranges * 8)) || (rc = f->d_rng_set.reserve((size_t)n_ranges * 4)) ||
            (rc = f->d_rng_cnt.reserve(cells * 4)) || (rc = f->d_rng_gcnt.reserve(cells * 4)) ||
            (rc = f->d_pair.reserve(std::max<uint64_t>(pair_total, 1) * 8)) || (rc = f->d_pair_ones.reserve((size_t)n_queries * 4))) return rc;
        if ((rc = facet_upload_ids(f, qd, result_ids, n_queries, ids_total, s))) return rc;
        TSGPU_HIP_TRY(hipMemcpyAsync(f->d_queries.p, qd.data(), qd.size() * sizeof(FacetQueryDev), hipMemcpyHostToDevice, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(f->d_rng_up.p, range_upper, (size_t)n_ranges * 8, hipMemcpyHostToDevice, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(f->d_rng_lo.p, range_lower, (size_t)n_ranges * 8, hipMemcpyHostToDevice, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(f->d_rng_set.p, rset.data(), (size_t)n_ranges * 4, hipMemcpyHostToDevice, s));
        TSGPU_HIP_TRY(hipMemsetAsync(f->d_rng_cnt.p, 0, cells * 4, s));
        TSGPU_HIP_TRY(hipMemsetAsync(f->d_rng_gcnt.p, 0, cells * 4, s));
        if (grouped) {
            TSGPU_HIP_TRY(hipMemsetAsync(f->d_pair.p, 0xFF, std::max<uint64_t>(pair_total, 1) * 8, s));
            TSGPU_HIP_TRY(hipMemsetAsync(f->d_pair_ones.p, 0, (size_t)n_queries * 4, s));
        }
        FacetRangeArgs a;
        a.doc_ptr = f->doc_ptr.as<uint64_t>(); a.hashes = f->hashes.as<uint32_t>(); a.n_docs = f->n_docs;
        a.ids = f->d_ids.as<uint32_t>(); a.queries = f->d_queries.as<FacetQueryDev>(); a.n_queries = n_queries; a.sample_mod = sample_mod; a.ids_per_block = ids_per_block;
        a.val_col = ctx->columns[value_column].data.as<long long>(); a.val_len = ctx->columns[value_column].n;
        a.range_upper = f->d_rng_up.as<long long>(); a.range_lower = f->d_rng_lo.as<long long>(); a.range_set = f->d_rng_set.as<uint32_t>(); a.n_ranges = n_ranges;
        a.grouped = grouped ? 1u : 0u; a.group_missing_values = group_missing_values ? 1u : 0u;
        a.group_col = grouped ? ctx->columns[group_column].data.as<long long>() : nullptr; a.group_len = grouped ? ctx->columns[group_column].n : 0;
        a.pair_key = f->d_pair.as<unsigned long long>(); a.pair_ones = f->d_pair_ones.as<uint32_t>();
        a.out_count = f->d_rng_cnt.as<uint32_t>(); a.out_gcount = f->d_rng_gcnt.as<uint32_t>();
        if (blocks) hipLaunchKernelGGL(facet_range_kernel, dim3(blocks), dim3(FACET_THREADS), 0, s, a);
        TSGPU_HIP_TRY(hipGetLastError());
        std::vector<uint32_t> hc(cells), hg(cells);
        TSGPU_HIP_TRY(hipMemcpyAsync(hc.data(), f->d_rng_cnt.p, cells * 4, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(hg.data(), f->d_rng_gcnt.p, cells * 4, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipStreamSynchronize(s));
        // a range is in result_map when a document fell into it; grouped: its count = the size of the set its (uint32) id names (:4455-4458)
        for (uint32_t q = 0; q < n_queries; q++)
            for (uint32_t r = 0; r < n_ranges; r++) {
                const size_t c = (size_t)q * n_ranges + r;
                counts[c] = hc[c] == 0 ? 0u : (grouped ? hg[(size_t)q * n_ranges + rset[r]] : hc[c]);
            }
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_facet_range_count_batch: host allocation failed"); }
    return ok();
}


namespace {
// the per-query id lists of a batch -> the id arena + FacetQueryDev table (ids_off / n_ids / first_block of a one-thread-per-id launch)
int upload_id_lists(FacetField* f, const uint32_t* const* result_ids, const uint64_t* n_result_ids, uint32_t n_queries, hipStream_t s, std::vector<FacetQueryDev>& qd, uint32_t& blocks) {
    qd.assign(n_queries, FacetQueryDev());
    uint64_t ids_total = 0;
    blocks = 0;
    for (uint32_t q = 0; q < n_queries; q++) {
        if (n_result_ids[q] && !result_ids[q]) return fail(TSGPU_ERR_INVALID, "tsgpu_facet: result_ids[q] is NULL");
        qd[q].ids_off = ids_total; qd[q].n_ids = n_result_ids[q];
        ids_total += n_result_ids[q];
        qd[q].first_block = blocks;
        const uint64_t nb = (n_result_ids[q] + FACET_THREADS - 1) / FACET_THREADS;
        if ((uint64_t)blocks + nb > 0x7FFFFFFFull) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_facet: more than 2^31 workgroups");
        blocks += (uint32_t)nb;
    }
    int rc;
    if ((rc = f->d_ids.reserve(std::max<uint64_t>(ids_total, 1) * 4)) || (rc = f->d_queries.reserve(qd.size() * sizeof(FacetQueryDev)))) return rc;
    if ((rc = facet_upload_ids(f, qd, result_ids, n_queries, ids_total, s))) return rc;
    TSGPU_HIP_TRY(hipMemcpyAsync(f->d_queries.p, qd.data(), qd.size() * sizeof(FacetQueryDev), hipMemcpyHostToDevice, s));
    return TSGPU_OK;
}
}  // namespace

// numeric facet stats of the hash-index branch (should_compute_stats): see facet_stats_kernel
int tsgpu_facet_stats_batch(tsgpu_ctx* ctx, uint32_t facet_field_id, int value_type, const uint32_t* const* result_ids, const uint64_t* n_result_ids, uint32_t n_queries,
                            uint32_t sample_mod, const uint32_t* int64_map_hashes, const int64_t* int64_map_values, uint32_t n_map, tsgpu_facet_stats* out) {
    if (!ctx || !out || (n_queries && (!result_ids || !n_result_ids))) return fail(TSGPU_ERR_INVALID, "tsgpu_facet_stats_batch: NULL argument");
    if (value_type != TSGPU_FACET_INT32 && value_type != TSGPU_FACET_INT64 && value_type != TSGPU_FACET_FLOAT) return fail(TSGPU_ERR_INVALID, "tsgpu_facet_stats_batch: unknown value type");
    if (n_map && (!int64_map_hashes || !int64_map_values)) return fail(TSGPU_ERR_INVALID, "tsgpu_facet_stats_batch: int64 map is NULL");
    for (uint32_t i = 1; i < n_map; i++) if (int64_map_hashes[i] <= int64_map_hashes[i - 1]) return fail(TSGPU_ERR_INVALID, "tsgpu_facet_stats_batch: int64_map_hashes must be strictly ascending");
    if (n_queries == 0) return ok();
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    auto it = ctx->facet_fields.find(facet_field_id);
    if (it == ctx->facet_fields.end()) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_facet_stats_batch: unknown facet field (tsgpu_facet_set)");
    FacetField* f = it->second;
    hipStream_t s = ctx->stream;
    if (sample_mod == 0) sample_mod = 1;
    try {
        std::vector<FacetQueryDev> qd;
        uint32_t blocks = 0;
        int rc = upload_id_lists(f, result_ids, n_result_ids, n_queries, s, qd, blocks);
        if (rc) return rc;
        std::vector<FacetStatsDev> init(n_queries);
        for (auto& d : init) { d.vmin = ~0ull; d.vmax = 0; d.isum = 0; d.count = 0; d.fsum = 0.0; d.absmax = 0; }
        if ((rc = f->d_stats.reserve(init.size() * sizeof(FacetStatsDev))) || (rc = f->d_map_h.reserve(std::max<uint32_t>(n_map, 1) * 4)) || (rc = f->d_map_v.reserve(std::max<uint32_t>(n_map, 1) * 8))) return rc;
        TSGPU_HIP_TRY(hipMemcpyAsync(f->d_stats.p, init.data(), init.size() * sizeof(FacetStatsDev), hipMemcpyHostToDevice, s));
        if (n_map) {
            TSGPU_HIP_TRY(hipMemcpyAsync(f->d_map_h.p, int64_map_hashes, (size_t)n_map * 4, hipMemcpyHostToDevice, s));
            TSGPU_HIP_TRY(hipMemcpyAsync(f->d_map_v.p, int64_map_values, (size_t)n_map * 8, hipMemcpyHostToDevice, s));
        }
        FacetStatsArgs a;
        a.doc_ptr = f->doc_ptr.as<uint64_t>(); a.hashes = f->hashes.as<uint32_t>(); a.n_docs = f->n_docs;
        a.ids = f->d_ids.as<uint32_t>(); a.queries = f->d_queries.as<FacetQueryDev>(); a.n_queries = n_queries; a.sample_mod = sample_mod;
        a.value_type = value_type; a.map_hash = f->d_map_h.as<uint32_t>(); a.map_val = f->d_map_v.as<long long>(); a.n_map = n_map;
        a.out = f->d_stats.as<FacetStatsDev>();
        if (blocks) hipLaunchKernelGGL(facet_stats_kernel, dim3(blocks), dim3(FACET_THREADS), 0, s, a);
        TSGPU_HIP_TRY(hipGetLastError());
        TSGPU_HIP_TRY(hipMemcpyAsync(init.data(), f->d_stats.p, init.size() * sizeof(FacetStatsDev), hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipStreamSynchronize(s));
        for (uint32_t q = 0; q < n_queries; q++) {
            const FacetStatsDev& d = init[q];
            tsgpu_facet_stats& o = out[q];
            o.fvcount = d.count;
            o.fvmin = std::numeric_limits<double>::max(); o.fvmax = -std::numeric_limits<double>::max(); o.fvsum = 0.0; o.sum_exact = 1;      // include/field.h:765-770
            if (d.count == 0) continue;
            if (value_type == TSGPU_FACET_FLOAT) {
                auto unkey = [](unsigned long long k) { uint32_t b = (uint32_t)k; b = (b & 0x80000000u) ? (b & 0x7FFFFFFFu) : ~b; float f; memcpy(&f, &b, 4); return f; };
                o.fvmin = (double)unkey(d.vmin); o.fvmax = (double)unkey(d.vmax); o.fvsum = d.fsum;
                o.sum_exact = 0;                       // the reference adds in document order; this sum differs from it by rounding only (~1e-16 relative per term)
            } else {
                o.fvmin = (double)(long long)(d.vmin ^ 0x8000000000000000ull); o.fvmax = (double)(long long)(d.vmax ^ 0x8000000000000000ull);
                // every partial sum of the reference's double accumulation is an integer below count * max|v|: exact (and equal) under 2^53;
                // beyond that the reference's own sum is a rounded one: the double-atomic sum stands in (same value up to rounding)
                const long double bound = (long double)d.count * (long double)d.absmax;
                o.sum_exact = bound < 9007199254740992.0L ? 1 : 0;
                o.fvsum = o.sum_exact ? (double)(long long)d.isum : d.fsum;
            }
        }
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_facet_stats_batch: host allocation failed"); }
    return ok();
}

// value index of a facet field: the seq_ids of every value, values in the reference's visiting order (counter_list: by total count)
int tsgpu_facet_value_set(tsgpu_ctx* ctx, uint32_t facet_field_id, const uint64_t* value_ptr, const uint32_t* seq_ids, const uint32_t* total_counts, uint32_t n_values) {
    if (!ctx || !value_ptr || (n_values && !total_counts)) return fail(TSGPU_ERR_INVALID, "tsgpu_facet_value_set: NULL argument");
    if (value_ptr[0] != 0) return fail(TSGPU_ERR_INVALID, "tsgpu_facet_value_set: value_ptr[0] must be 0");
    for (uint32_t v = 0; v < n_values; v++) {
        if (value_ptr[v + 1] <= value_ptr[v]) return fail(TSGPU_ERR_INVALID, "tsgpu_facet_value_set: every value needs at least one seq_id");
        for (uint64_t i = value_ptr[v] + 1; i < value_ptr[v + 1]; i++) if (seq_ids[i] <= seq_ids[i - 1]) return fail(TSGPU_ERR_INVALID, "tsgpu_facet_value_set: a value's seq_ids must be strictly ascending");
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    try {
        FacetField* f;
        auto it = ctx->facet_fields.find(facet_field_id);
        if (it == ctx->facet_fields.end()) { f = new FacetField; ctx->facet_fields[facet_field_id] = f; } else f = it->second;
        DevBuf np, ni, nt;
        int rc;
        const uint64_t n_ids = value_ptr[n_values];
        if ((rc = np.reserve(((size_t)n_values + 1) * 8)) || (rc = ni.reserve(std::max<uint64_t>(n_ids, 1) * 4)) || (rc = nt.reserve(std::max<uint32_t>(n_values, 1) * 4))) { np.release(); ni.release(); nt.release(); return rc; }
        hipError_t e = hipMemcpy(np.p, value_ptr, ((size_t)n_values + 1) * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess && n_ids) e = hipMemcpy(ni.p, seq_ids, n_ids * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess && n_values) e = hipMemcpy(nt.p, total_counts, (size_t)n_values * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) { np.release(); ni.release(); nt.release(); return fail(TSGPU_ERR_DEVICE, std::string("tsgpu_facet_value_set: ") + hipGetErrorString(e)); }
        f->val_ptr.release(); f->val_ids.release(); f->val_total.release();
        f->val_ptr = np; f->val_ids = ni; f->val_total = nt; f->n_values = n_values;
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_facet_value_set: host allocation failed"); }
    return ok();
}

// the value-index branch of do_facets for a batch of result-id lists: see facet_value_count_kernel / facet_value_select_kernel
int tsgpu_facet_value_count_batch(tsgpu_ctx* ctx, uint32_t facet_field_id, const uint32_t* const* result_ids, const uint64_t* n_result_ids, uint32_t n_queries,
                                  uint32_t max_facets, int is_wildcard_no_filter_query, int estimate_facets, uint32_t facet_sample_interval, const uint32_t* order,
                                  tsgpu_facet_value_counts* out) {
    if (!ctx || !out || (n_queries && (!result_ids || !n_result_ids))) return fail(TSGPU_ERR_INVALID, "tsgpu_facet_value_count_batch: NULL argument");
    if (!out->value_index || !out->count || !out->doc_id || !out->n_found || out->cap == 0) return fail(TSGPU_ERR_INVALID, "tsgpu_facet_value_count_batch: missing output arrays");
    if (n_queries == 0) return ok();
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    auto it = ctx->facet_fields.find(facet_field_id);
    if (it == ctx->facet_fields.end() || it->second->n_values == 0) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_facet_value_count_batch: the field has no value index (tsgpu_facet_value_set)");
    FacetField* f = it->second;
    hipStream_t s = ctx->stream;
    if (facet_sample_interval == 0) facet_sample_interval = 1;
    if (order) {
        std::vector<uint8_t> seen(f->n_values, 0);
        for (uint32_t i = 0; i < f->n_values; i++) { if (order[i] >= f->n_values || seen[order[i]]) return fail(TSGPU_ERR_INVALID, "tsgpu_facet_value_count_batch: order must be a permutation of the values"); seen[order[i]] = 1; }
    }
    try {
        std::vector<FacetQueryDev> qd;
        uint32_t blocks = 0;
        int rc = upload_id_lists(f, result_ids, n_result_ids, n_queries, s, qd, blocks);
        if (rc) return rc;
        const size_t n_cnt = (size_t)n_queries * f->n_values, n_out = (size_t)n_queries * out->cap;
        if ((rc = f->d_counts.reserve(n_cnt * 4)) || (rc = f->d_oh.reserve(n_out * 4)) || (rc = f->d_oc.reserve(n_out * 4)) || (rc = f->d_od.reserve(n_out * 4)) ||
            (rc = f->d_on.reserve((size_t)n_queries * 4)) || (rc = f->d_order.reserve((size_t)f->n_values * 4))) return rc;
        if (order) TSGPU_HIP_TRY(hipMemcpyAsync(f->d_order.p, order, (size_t)f->n_values * 4, hipMemcpyHostToDevice, s));
        FacetValueArgs a;
        a.val_ptr = f->val_ptr.as<uint64_t>(); a.val_ids = f->val_ids.as<uint32_t>(); a.val_total = f->val_total.as<uint32_t>(); a.n_values = f->n_values;
        a.ids = f->d_ids.as<uint32_t>(); a.queries = f->d_queries.as<FacetQueryDev>(); a.n_queries = n_queries;
        a.order = order ? f->d_order.as<uint32_t>() : nullptr;
        a.max_facets = max_facets; a.wildcard_no_filter = is_wildcard_no_filter_query; a.estimate = estimate_facets; a.interval = facet_sample_interval;
        a.counts = f->d_counts.as<uint32_t>(); a.cap = out->cap;
        a.out_value = f->d_oh.as<uint32_t>(); a.out_count = f->d_oc.as<uint32_t>(); a.out_doc = f->d_od.as<uint32_t>(); a.out_n = f->d_on.as<uint32_t>();
        hipLaunchKernelGGL(facet_value_count_kernel, dim3(f->n_values, n_queries), dim3(64), 0, s, a);
        hipLaunchKernelGGL(facet_value_select_kernel, dim3(n_queries), dim3(64), 0, s, a);
        TSGPU_HIP_TRY(hipGetLastError());
        TSGPU_HIP_TRY(hipMemcpyAsync(out->value_index, f->d_oh.p, n_out * 4, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(out->count, f->d_oc.p, n_out * 4, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(out->doc_id, f->d_od.p, n_out * 4, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(out->n_found, f->d_on.p, (size_t)n_queries * 4, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipStreamSynchronize(s));
        for (uint32_t q = 0; q < n_queries; q++) if (n_result_ids[q] == 0) out->n_found[q] = 0;       // results_size == 0 -> do_facets returns at once (:1531-1533)
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_facet_value_count_batch: host allocation failed"); }
    return ok();
}

}  // extern "C"
