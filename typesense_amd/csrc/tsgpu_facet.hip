// tsgpu_facet.hip — facet counting over matched ids behind include/tsgpu.h: the hash-index branch of
// Index::do_facets (src/index.cpp:1659-1771). The host only sizes the per-query tables and orders the (few) distinct values the
// device found; the per-document work runs in facet_kernels.hip.h.
#include "tsgpu_host.h"
#include "facet_kernels.hip.h"

using namespace tsgpu;

namespace tsgpu {
struct FacetField {
    DevBuf doc_ptr, hashes;
    uint32_t n_docs = 0;
    uint64_t n_hashes = 0;
    uint32_t n_distinct = 0, max_per_doc = 0;
    DevBuf d_ids, d_queries, d_allowed, d_key, d_cnt, d_last, d_oh, d_oc, d_od, d_op, d_on;      // per-batch scratch
    void release() {
        DevBuf* b[] = {&doc_ptr, &hashes, &d_ids, &d_queries, &d_allowed, &d_key, &d_cnt, &d_last, &d_oh, &d_oc, &d_od, &d_op, &d_on};
        for (auto* x : b) x->release();
    }
};
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

int tsgpu_facet_count_batch(tsgpu_ctx* ctx, uint32_t facet_field_id, const uint32_t* const* result_ids, const uint64_t* n_result_ids, uint32_t n_queries,
                            uint32_t sample_mod, const uint32_t* allowed_hashes, uint32_t n_allowed, tsgpu_facet_counts* out) {
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
    try {
        std::vector<FacetQueryDev> qd(n_queries);
        uint64_t ids_total = 0, tab_total = 0, out_total = 0;
        uint32_t blocks = 0;
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
            const uint64_t nb = (d.n_ids + FACET_THREADS - 1) / FACET_THREADS;
            if ((uint64_t)blocks + nb > 0x7FFFFFFFull) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_facet_count_batch: more than 2^31 workgroups");
            blocks += (uint32_t)nb;
        }
        if (tab_total * 20 > (32ull << 30)) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_facet_count_batch: the counting tables of this batch exceed 32 GiB; split it");
        int rc;
        if ((rc = f->d_ids.reserve(std::max<uint64_t>(ids_total, 1) * 4)) || (rc = f->d_queries.reserve(qd.size() * sizeof(FacetQueryDev))) ||
            (rc = f->d_key.reserve(tab_total * 8)) || (rc = f->d_cnt.reserve(tab_total * 4)) || (rc = f->d_last.reserve(tab_total * 8)) ||
            (rc = f->d_oh.reserve(out_total * 4)) || (rc = f->d_oc.reserve(out_total * 4)) || (rc = f->d_od.reserve(out_total * 4)) ||
            (rc = f->d_op.reserve(out_total * 4)) || (rc = f->d_on.reserve((size_t)n_queries * 4)) || (rc = f->d_allowed.reserve(std::max<uint32_t>(n_allowed, 1) * 4)))
            return rc;
        for (uint32_t q = 0; q < n_queries; q++)
            if (qd[q].n_ids) TSGPU_HIP_TRY(hipMemcpyAsync(f->d_ids.as<uint32_t>() + qd[q].ids_off, result_ids[q], qd[q].n_ids * 4, hipMemcpyHostToDevice, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(f->d_queries.p, qd.data(), qd.size() * sizeof(FacetQueryDev), hipMemcpyHostToDevice, s));
        if (n_allowed) TSGPU_HIP_TRY(hipMemcpyAsync(f->d_allowed.p, allowed_hashes, (size_t)n_allowed * 4, hipMemcpyHostToDevice, s));
        TSGPU_HIP_TRY(hipMemsetAsync(f->d_key.p, 0, tab_total * 8, s));
        TSGPU_HIP_TRY(hipMemsetAsync(f->d_cnt.p, 0, tab_total * 4, s));
        TSGPU_HIP_TRY(hipMemsetAsync(f->d_last.p, 0, tab_total * 8, s));
        FacetArgs a;
        a.doc_ptr = f->doc_ptr.as<uint64_t>(); a.hashes = f->hashes.as<uint32_t>(); a.n_docs = f->n_docs;
        a.ids = f->d_ids.as<uint32_t>(); a.queries = f->d_queries.as<FacetQueryDev>(); a.n_queries = n_queries; a.sample_mod = sample_mod;
        a.allowed = n_allowed ? f->d_allowed.as<uint32_t>() : nullptr; a.n_allowed = n_allowed;
        a.tab_key = f->d_key.as<unsigned long long>(); a.tab_cnt = f->d_cnt.as<uint32_t>(); a.tab_last = f->d_last.as<unsigned long long>();
        a.out_hash = f->d_oh.as<uint32_t>(); a.out_cnt = f->d_oc.as<uint32_t>(); a.out_doc = f->d_od.as<uint32_t>(); a.out_pos = f->d_op.as<uint32_t>();
        a.out_n = f->d_on.as<uint32_t>();
        if (blocks) hipLaunchKernelGGL(facet_count_kernel, dim3(blocks), dim3(FACET_THREADS), 0, s, a);
        hipLaunchKernelGGL(facet_compact_kernel, dim3(n_queries), dim3(FACET_THREADS), 0, s, a);
        TSGPU_HIP_TRY(hipGetLastError());
        std::vector<uint32_t> hn(n_queries), hh(out_total), hc(out_total), hd(out_total), hp(out_total);
        TSGPU_HIP_TRY(hipMemcpyAsync(hn.data(), f->d_on.p, (size_t)n_queries * 4, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(hh.data(), f->d_oh.p, out_total * 4, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(hc.data(), f->d_oc.p, out_total * 4, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(hd.data(), f->d_od.p, out_total * 4, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(hp.data(), f->d_op.p, out_total * 4, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipStreamSynchronize(s));
        // the distinct values of a query in ascending hash order (result_map is keyed by the hash); the first `cap` of them are returned
        std::vector<uint32_t> order;
        for (uint32_t q = 0; q < n_queries; q++) {
            const uint32_t n = hn[q];
            const uint64_t base = qd[q].out_off;
            order.resize(n);
            for (uint32_t i = 0; i < n; i++) order[i] = i;
            std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return hh[base + x] < hh[base + y]; });
            out->n_values[q] = n;
            const uint32_t m = std::min(n, out->cap);
            for (uint32_t i = 0; i < m; i++) {
                const size_t o = (size_t)q * out->cap + i;
                out->hash[o] = hh[base + order[i]]; out->count[o] = hc[base + order[i]]; out->doc_id[o] = hd[base + order[i]]; out->array_pos[o] = hp[base + order[i]];
            }
        }
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_facet_count_batch: host allocation failed"); }
    return ok();
}

}  // extern "C"
