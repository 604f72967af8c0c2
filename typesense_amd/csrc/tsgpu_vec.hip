// tsgpu_vec.hip — vector index mirror + batched exact k-NN (seam B2), pure-vector search (src/index.cpp:3645-3732),
// hybrid rank fusion (src/index.cpp:4036-4221) and the multi-GPU shard merge, behind include/tsgpu.h.
// Distances and top-k selection run on the GPU (vec_kernels.hip.h); the host only orders <= k already-scored
// hits per query exactly the way the reference's Topster does.
#include <cmath>
#include <cfloat>
#include <atomic>
#include <thread>
#include <tuple>
#include "tsgpu_host.h"
#include <chrono>
#include "vec_kernels.hip.h"
#include "host_topster.h"

#include "tsgpu_hnsw_build.h"
#include "vec_hnsw_build.hip.h"

using namespace tsgpu;

namespace tsgpu {

struct VecField {
    uint32_t dim = 0;
    int metric = TSGPU_METRIC_IP;
    DevBuf X, labels, row_ok;
    DevBuf Xh, xnorm, tile_nmax;                       // bf16 prefilter mirror: bf16 rows [cap][dimp], inflated row norms, per-tile max norm
    uint32_t dimp = 0;                                 // dim rounded up to a multiple of 64 (zero padded)
    uint64_t cap_rows = 0, n_rows = 0, n_live = 0;
    std::vector<uint64_t> h_labels;
    std::vector<uint8_t> h_ok;
    bool identity = true;                              // label == row for every row so far
    std::unordered_map<uint64_t, uint32_t> row_of;     // materialised when identity breaks
    bool any_deleted = false;
    // scratch
    DevBuf d_dense, d_cand, d_cand_cnt, d_tau, dQ, d_dist, d_lab, d_cnt, d_mask, d_rows, d_q1, d_out1;
    DevBuf d_Qh, d_cq, d_L1, d_lbkey, d_surv, d_surv_cnt, d_gkeys;
    uint64_t seg_cap_hint = 0;                         // the largest (slab, query) candidate-segment capacity this field's data has needed so far
    // HNSW graph mirror (tsgpu_vec_hnsw_load): hnswlib's link lists; rows = hnswlib internal ids
    DevBuf g_link0, g_upper_ptr, g_upper_links, g_visited, g_vhash, g_stat, g_sel;
    uint32_t g_tag_slots = 0;                          // tag mode: concurrent queries the allocated tag array serves (0 = not allocated)
    uint32_t g_M = 0, g_n = 0, g_slots = 0, g_epoch = 1;
    int32_t g_maxlevel = -1;
    uint32_t g_enterpoint = 0;
    bool g_loaded = false;
    std::unique_ptr<HnswBuilder> hb;                   // graph construction inside the library (tsgpu_vec_hnsw_enable); null = the graph is mirrored from outside

    bool find_row(uint64_t label, uint32_t& row) const {
        if (identity) { if (label < n_rows) { row = (uint32_t)label; return true; } return false; }
        auto it = row_of.find(label);
        if (it == row_of.end()) return false;
        row = it->second;
        return true;
    }
    void break_identity() {
        if (!identity) return;
        row_of.reserve(h_labels.size() * 2);
        for (size_t r = 0; r < h_labels.size(); r++) row_of.emplace(h_labels[r], (uint32_t)r);
        identity = false;
    }
    void release() {
        DevBuf* b[] = {&X, &labels, &row_ok, &Xh, &xnorm, &tile_nmax, &d_dense, &d_cand, &d_cand_cnt, &d_tau, &dQ, &d_dist, &d_lab, &d_cnt, &d_mask, &d_rows,
                       &d_q1, &d_out1, &d_Qh, &d_cq, &d_L1, &d_lbkey, &d_surv, &d_surv_cnt, &d_gkeys, &g_link0, &g_upper_ptr, &g_upper_links, &g_visited, &g_vhash, &g_stat, &g_sel};
        for (auto* x : b) x->release();
    }
};

static int vec_reserve_rows(VecField* f, uint64_t rows, hipStream_t s) {
    if (rows <= f->cap_rows) return TSGPU_OK;
    uint64_t want = std::max<uint64_t>(rows, f->cap_rows + f->cap_rows / 2 + 1024);
    DevBuf nx, nl, no, nh, nn, nt;
    int rc;
    auto drop = [&]() { nx.release(); nl.release(); no.release(); nh.release(); nn.release(); nt.release(); };
    if ((rc = nx.reserve((size_t)want * f->dim * 4)) || (rc = nl.reserve((size_t)want * 8)) || (rc = no.reserve((size_t)want)) ||
        (rc = nh.reserve((size_t)((want + VEC_ROWS - 1) / VEC_ROWS) * VEC_ROWS * f->dimp * 2)) || (rc = nn.reserve((size_t)want * 4)) || (rc = nt.reserve((size_t)(want / VEC_ROWS + 2) * 4))) { drop(); return rc; }
    if (f->n_rows) {
        const size_t tiles = (size_t)((f->n_rows + VEC_ROWS - 1) / VEC_ROWS);
        hipError_t e = hipMemcpyAsync(nx.p, f->X.p, (size_t)f->n_rows * f->dim * 4, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(nl.p, f->labels.p, (size_t)f->n_rows * 8, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(no.p, f->row_ok.p, (size_t)f->n_rows, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(nh.p, f->Xh.p, tiles * VEC_ROWS * f->dimp * 2, hipMemcpyDeviceToDevice, s);   // whole tiles (tiled layout)
        if (e == hipSuccess) e = hipMemcpyAsync(nn.p, f->xnorm.p, (size_t)f->n_rows * 4, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(nt.p, f->tile_nmax.p, tiles * 4, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { drop(); return fail(TSGPU_ERR_DEVICE, std::string("vec_reserve_rows: ") + hipGetErrorString(e)); }   // the field keeps its old buffers
    }
    f->X.release(); f->labels.release(); f->row_ok.release(); f->Xh.release(); f->xnorm.release(); f->tile_nmax.release();
    f->X = nx; f->labels = nl; f->row_ok = no; f->Xh = nh; f->xnorm = nn; f->tile_nmax = nt;
    f->cap_rows = want;
    return TSGPU_OK;
}

static VecField* get_field(tsgpu_ctx* ctx, uint32_t id) {
    auto it = ctx->vec_fields.find(id);
    return it == ctx->vec_fields.end() ? nullptr : it->second;
}

// the exact k-NN launch sequence (vec_kernels.hip.h header); caller holds ctx->mu.
// Q_dev: [n_q][dim] on the device (already normalised for cosine). Outputs [n_q][k] on the device.
static int knn_group(tsgpu_ctx* ctx, VecField* f, const float* Q_dev, uint32_t n_q, uint32_t k, const uint8_t* mask_dev,
                     float* dist_dev, uint64_t* label_dev, uint32_t* cnt_dev, bool record_events) {
    hipStream_t s = ctx->stream;
    const uint32_t n_rows = (uint32_t)f->n_rows;
    const uint32_t n_tiles = (n_rows + VEC_ROWS - 1) / VEC_ROWS;
    const bool wide = n_q > 64;                                   // QT = 128 (CB = 2) or 64 (CB = 1)
    const uint32_t QT = wide ? 128 : 64;
    const uint32_t n_qtiles = (n_q + QT - 1) / QT;
    const bool aligned = (f->dim % 4 == 0) && (((uintptr_t)Q_dev & 15) == 0) && (((uintptr_t)f->X.p & 15) == 0);
    const uint32_t sample_tiles = ctx->vec_sample_tiles ? ctx->vec_sample_tiles : 512;
    int rc;
    if ((rc = f->d_tau.reserve((size_t)n_q * 8))) return rc;
    if ((rc = f->d_cand_cnt.reserve((size_t)n_q * 4 + 16))) return rc;
    uint32_t* d_over = f->d_cand_cnt.as<uint32_t>() + n_q;         // overflow counter lives behind the per-query counts

    auto launch_scan = [&](VecScanArgs& a, uint32_t target_wgs) {
        // slabs: a multiple of 8 (XCD mapping), ~target_wgs workgroups in total
        uint32_t n_slabs = std::max<uint32_t>(8, (std::max<uint32_t>(1, target_wgs / n_qtiles) + 7) / 8 * 8);
        uint32_t per = (a.n_ord + n_slabs - 1) / n_slabs;
        if (ctx->vec_rows_per_slab) per = std::max<uint32_t>(1, ctx->vec_rows_per_slab / VEC_ROWS);
        per = std::max<uint32_t>(per, 1);
        n_slabs = (a.n_ord + per - 1) / per;
        n_slabs = std::max<uint32_t>(8, (n_slabs + 7) / 8 * 8);
        a.ord_per_slab = per; a.n_slabs = n_slabs; a.n_qtiles = n_qtiles;
        const dim3 grid(n_slabs * n_qtiles), block(VEC_THREADS);
        if (wide) { if (aligned) hipLaunchKernelGGL((vec_scan_kernel<2, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((vec_scan_kernel<2, false>), grid, block, 0, s, a); }
        else { if (aligned) hipLaunchKernelGGL((vec_scan_kernel<1, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((vec_scan_kernel<1, false>), grid, block, 0, s, a); }
    };
    VecScanArgs base;
    memset(&base, 0, sizeof base);
    base.X = f->X.as<float>(); base.row_ok = mask_dev; base.Q = Q_dev;
    base.n_rows = n_rows; base.dim = f->dim; base.n_q = n_q;

    if (record_events) TSGPU_HIP_TRY(hipEventRecord(ctx->ev[3], s));
    if (n_tiles <= sample_tiles) {
        // small index: one dense pass over every row, then select
        const uint32_t stride = n_tiles * VEC_ROWS;
        if ((rc = f->d_dense.reserve((size_t)n_q * stride * 8))) return rc;
        VecScanArgs a = base;
        a.n_ord = n_tiles; a.tile_stride = 1; a.dense = f->d_dense.as<uint64_t>(); a.dense_stride = stride;
        launch_scan(a, 512);
        if (record_events) TSGPU_HIP_TRY(hipEventRecord(ctx->ev[4], s));
        hipLaunchKernelGGL(vec_select_kernel, dim3(n_q), dim3(VEC_THREADS), 0, s, (const uint64_t*)a.dense, (size_t)stride, (const uint32_t*)nullptr, stride, k, 0,
                           f->labels.as<uint64_t>(), dist_dev, label_dev, cnt_dev, f->d_tau.as<uint64_t>(), d_over);
        if (record_events) TSGPU_HIP_TRY(hipEventRecord(ctx->ev[5], s));
        TSGPU_HIP_TRY(hipGetLastError());
        return TSGPU_OK;
    }
    // pass 1: dense scan of a strided sample of the row tiles -> tau[q] = k-th best key of the sample
    const uint32_t tile_stride = n_tiles / sample_tiles;             // >= 1; sampled tiles = 0, stride, 2*stride, ...
    const uint32_t n_sample = (n_tiles + tile_stride - 1) / tile_stride;
    {
        const uint32_t stride = n_sample * VEC_ROWS;
        if ((rc = f->d_dense.reserve((size_t)n_q * stride * 8))) return rc;
        VecScanArgs a = base;
        a.n_ord = n_sample; a.tile_stride = tile_stride; a.dense = f->d_dense.as<uint64_t>(); a.dense_stride = stride;
        launch_scan(a, 512);
        hipLaunchKernelGGL(vec_select_kernel, dim3(n_q), dim3(VEC_THREADS), 0, s, (const uint64_t*)a.dense, (size_t)stride, (const uint32_t*)nullptr, stride, k, 1,
                           f->labels.as<uint64_t>(), dist_dev, label_dev, cnt_dev, f->d_tau.as<uint64_t>(), d_over);
    }
    // pass 2: every row, filtered by tau; expected survivors per query = k * n_tiles / n_sample
    uint64_t cap = ctx->vec_cand_cap;
    if (!cap) {
        cap = 4ull * k * ((n_tiles + n_sample - 1) / n_sample) + 1024;
        uint64_t p2 = 1024; while (p2 < cap) p2 <<= 1; cap = p2;
        while (cap > 4096 && cap * n_q * 8 > (1ull << 30)) cap >>= 1;   // bound the candidate arena to 1 GiB per query group
    }
    cap = std::max<uint64_t>(cap, 2ull * k);     // a full list must be able to tighten its own threshold
    if ((rc = f->d_cand.reserve((size_t)n_q * cap * 8))) return rc;
    uint32_t h_over = 0;
    for (int round = 0; round < 64; round++) {
        TSGPU_HIP_TRY(hipMemsetAsync(f->d_cand_cnt.p, 0, (size_t)n_q * 4 + 4, s));
        VecScanArgs a = base;
        a.n_ord = n_tiles; a.tile_stride = 1; a.tau = f->d_tau.as<uint64_t>();
        a.cand = f->d_cand.as<uint64_t>(); a.cand_cnt = f->d_cand_cnt.as<uint32_t>(); a.cand_cap = (uint32_t)cap;
        if (record_events && round == 0) TSGPU_HIP_TRY(hipEventRecord(ctx->ev[6], s));
        launch_scan(a, 512);
        if (record_events && round == 0) { TSGPU_HIP_TRY(hipEventRecord(ctx->ev[7], s)); TSGPU_HIP_TRY(hipEventRecord(ctx->ev[4], s)); ctx->scan_events_valid = true; }
        hipLaunchKernelGGL(vec_select_kernel, dim3(n_q), dim3(VEC_THREADS), 0, s, (const uint64_t*)a.cand, (size_t)cap, (const uint32_t*)a.cand_cnt, (uint32_t)cap, k, 0,
                           f->labels.as<uint64_t>(), dist_dev, label_dev, cnt_dev, f->d_tau.as<uint64_t>(), d_over);
        TSGPU_HIP_TRY(hipGetLastError());
        TSGPU_HIP_TRY(hipMemcpyAsync(&h_over, d_over, 4, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipStreamSynchronize(s));
        if (!h_over) break;                 // an overflowing list tightened its own tau: scan again (exactness is data-independent)
        ctx->vec_overflow_rounds++;
    }
    if (h_over) return fail(TSGPU_ERR_DEVICE, "vec knn: candidate lists still overflow after 64 refinement rounds");
    if (record_events) TSGPU_HIP_TRY(hipEventRecord(ctx->ev[5], s));
    return TSGPU_OK;
}


// safety inflation of the stored row norms (fp32 sum-of-squares error << 2^-10)
static const float VEC_NORM_INFLATE = 1.0f + 1.0f / 1024.0f;
// error-radius constant of the bf16 bracket (vec_kernels.hip.h): (2u + u^2) with u = 2^-8, + dim * 2^-21 for the two fp32
// accumulations, + 1 %
static float vec_bracket_c(uint32_t dim) { return ((1.0f / 128.0f + 1.0f / 16384.0f) + (float)dim * (1.0f / 2097152.0f)) * 1.01f; }

// (re)build the bf16 mirror + norms of rows [row0, row0 + n) and the tile maxima they touch; rows must already be
// final in X (normalised for cosine fields). Called with ctx->mu held; enqueues on the stream.
static int vec_refresh_mirror(VecField* f, uint32_t row0, uint32_t n, uint64_t n_rows_after, hipStream_t s) {
    if (n == 0) return TSGPU_OK;
    hipLaunchKernelGGL(vec_to_bf16_kernel, dim3((n + 3) / 4), dim3(256), 0, s, f->X.as<float>(), f->Xh.as<uint16_t>(), f->xnorm.as<float>(), row0, n, f->dim,
                       f->dimp, VEC_NORM_INFLATE, 0u);
    const uint32_t t0 = row0 / VEC_ROWS, t1 = (row0 + n - 1) / VEC_ROWS;
    hipLaunchKernelGGL(vec_tile_nmax_kernel, dim3((t1 - t0 + 1 + 63) / 64), dim3(64), 0, s, f->xnorm.as<float>(), f->tile_nmax.as<float>(), t0, t1 - t0 + 1,
                       (uint32_t)n_rows_after);
    TSGPU_HIP_TRY(hipGetLastError());
    return TSGPU_OK;
}

// the bf16-prefilter k-NN launch sequence (vec_kernels.hip.h, "bf16 PREFILTER path"); caller holds ctx->mu.
// rows per query that may reach the exact re-score (more -> fp32 scan fallback). A query whose k-th neighbour sits in a tight cluster
// has the WHOLE cluster inside its bf16 bracket (unit vectors, 10 000-row clusters: ~9 100 survivors per query; with the 8 192 of rounds
// 1-2 every such group fell back to the fp32 scan: 39 ms instead of 7 per 256 queries). Re-scoring S rows per query costs S x dim x 4 B
// of random row reads: at 64 K x 768 dims x 256 queries ~50 GB = about half the fp32 scan's time, beyond that the fallback is as good.
static const uint32_t VEC_SURV_CAP = 65536;
static const uint32_t VEC_REFINE_GCAP = 4 * VEC_SURV_CAP - VEC_REFINE_LCAP;      // candidates per query beyond the refine kernel's LDS list
static int knn_group_prefilter(tsgpu_ctx* ctx, VecField* f, const float* Q_dev, uint32_t n_q, uint32_t k, const uint8_t* mask_dev,
                               float* dist_dev, uint64_t* label_dev, uint32_t* cnt_dev, bool record_events) {
    hipStream_t s = ctx->stream;
    const uint32_t n_rows = (uint32_t)f->n_rows;
    const uint32_t n_tiles = (n_rows + VEC_ROWS - 1) / VEC_ROWS;
    const uint32_t QT = n_q > 128 ? 256 : (n_q > 64 ? 128 : 64);     // query tile of a workgroup: vec_hscan_kernel<QT / 64>
    const uint32_t n_qtiles = (n_q + QT - 1) / QT;
    const uint32_t sample_tiles = ctx->vec_sample_tiles ? ctx->vec_sample_tiles : 8192;   // bf16 sample pass is cheap: ~1M rows
    const uint32_t n_q_pad = (n_q + 255) / 256 * 256;
    int rc;
    if ((rc = f->d_Qh.reserve((size_t)n_q_pad * f->dimp * 2))) return rc;
    if ((rc = f->d_cq.reserve((size_t)n_q * 4))) return rc;
    if ((rc = f->d_L1.reserve((size_t)n_q * 4))) return rc;
    if ((rc = f->d_surv_cnt.reserve((size_t)n_q * 4 + 16))) return rc;
    if ((rc = f->d_tau.reserve((size_t)n_q * 8))) return rc;
    if ((rc = f->d_surv.reserve((size_t)n_q * VEC_SURV_CAP * 4))) return rc;
    if ((rc = f->d_dense.reserve((size_t)n_q * VEC_SURV_CAP * 8))) return rc;          // exact keys of the survivors
    if ((rc = f->d_gkeys.reserve((size_t)n_q * VEC_REFINE_GCAP * 4))) return rc;        // candidate keys beyond the refine kernel's LDS list
    uint32_t* d_over = f->d_surv_cnt.as<uint32_t>() + n_q;      // [0] overflow, [1] stuck (behind the per-query survivor counts)

    // slab geometry of a scan over n_ord tile ordinals: slabs are a multiple of 8 (XCD mapping), ~target workgroups in total
    auto geometry = [&](uint32_t n_ord, uint32_t target_wgs, uint32_t& per, uint32_t& n_slabs) {
        n_slabs = std::max<uint32_t>(8, (std::max<uint32_t>(1, target_wgs / n_qtiles) + 7) / 8 * 8);
        per = (n_ord + n_slabs - 1) / n_slabs;
        if (ctx->vec_rows_per_slab) per = std::max<uint32_t>(1, ctx->vec_rows_per_slab / VEC_ROWS);
        per = std::max<uint32_t>(per, 1);
        per = std::min<uint32_t>(std::max<uint32_t>(per, 2), VEC_HMAX_PER) / 2 * 2;     // a workgroup steps over ordinal PAIRS; norm maxima of a slab live in LDS
        n_slabs = (n_ord + per - 1) / per;
        n_slabs = std::max<uint32_t>(8, (n_slabs + 7) / 8 * 8);
    };
    auto launch_scan = [&](VecHScanArgs& a, uint32_t target_wgs) {
        geometry(a.n_ord, target_wgs, a.ord_per_slab, a.n_slabs);
        a.n_qtiles = n_qtiles;
        const dim3 grid(a.n_slabs * n_qtiles), block(VEC_HTHREADS);
        if (QT == 256) hipLaunchKernelGGL((vec_hscan_kernel<4>), grid, block, 0, s, a);
        else if (QT == 128) hipLaunchKernelGGL((vec_hscan_kernel<2>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((vec_hscan_kernel<1>), grid, block, 0, s, a);
    };
    if (record_events) TSGPU_HIP_TRY(hipEventRecord(ctx->ev[3], s));
    // queries -> bf16 (chunk-major) + cq = c * ||q||
    hipLaunchKernelGGL(vec_to_bf16_kernel, dim3((n_q + 3) / 4), dim3(256), 0, s, Q_dev, f->d_Qh.as<uint16_t>(), f->d_cq.as<float>(), 0u, n_q, f->dim, f->dimp,
                       vec_bracket_c(f->dim) * VEC_NORM_INFLATE, n_q_pad);
    VecHScanArgs base;
    memset(&base, 0, sizeof base);
    base.Xh = f->Xh.as<uint16_t>(); base.row_ok = mask_dev; base.Qh = f->d_Qh.as<uint16_t>(); base.tile_nmax = f->tile_nmax.as<float>();
    base.n_q_pad = n_q_pad;
    base.cq = f->d_cq.as<float>(); base.n_rows = n_rows; base.dimp = f->dimp; base.n_q = n_q; base.L1 = f->d_L1.as<float>();

    // pass 1: strided sample of the row tiles (all of them for a small index) -> L1[q] = k-th largest group maximum of the
    // sample's lower bounds
    const uint32_t tile_stride = std::max<uint32_t>(1, n_tiles / sample_tiles);
    const uint32_t n_sample = (n_tiles + tile_stride - 1) / tile_stride;
    {
        const uint32_t gstride = n_sample * 4;
        if ((rc = f->d_lbkey.reserve((size_t)n_q * gstride * 4))) return rc;
        VecHScanArgs a = base;
        a.n_ord = n_sample; a.tile_stride = tile_stride; a.mode = 1; a.gmax = f->d_lbkey.as<uint32_t>(); a.gstride = gstride;
        launch_scan(a, 256);
        hipLaunchKernelGGL(vec_thresh_kernel, dim3(n_q), dim3(VEC_THREADS), 0, s, (const uint32_t*)a.gmax, (size_t)gstride, gstride, k, f->d_L1.as<float>());
    }
    // pass 2: every row, kept iff its upper bound reaches L1; candidates land in per-(slab, query) segments
    uint32_t per = 0, n_slabs = 0;
    geometry(n_tiles, 256, per, n_slabs);
    uint64_t seg_cap = ctx->vec_cand_cap;
    if (!seg_cap) {
        // expected candidates per query ~ k * rows / sample rows, times the bracket's widening (x8 head-room), spread over the slabs
        const uint64_t expect = 8ull * k * ((n_tiles + n_sample - 1) / n_sample) / n_slabs;
        seg_cap = 128; while (seg_cap < 4 * expect + 64 && seg_cap < 8192) seg_cap <<= 1;
        while (seg_cap > 128 && seg_cap * n_slabs * n_q * 8 > (1ull << 31)) seg_cap >>= 1;
        // >= 128: a slab of one tile always fits, so an index too small to yield a threshold (fewer than k sample groups) still works
    }
    seg_cap = std::max<uint64_t>(seg_cap, 1);
    const uint64_t arena_limit = 4ull << 30;             // bytes of candidate segments at most (n_slabs x n_q x seg_cap x 8)
    // (a field whose data needed larger segments before starts there: every call would otherwise pay the scan that discovers it)
    if (!ctx->vec_cand_cap) while (seg_cap < f->seg_cap_hint && seg_cap * 2 * n_slabs * n_q * 8 <= arena_limit) seg_cap *= 2;
    if ((rc = f->d_cand.reserve((size_t)n_slabs * n_q * seg_cap * 8))) return rc;
    if ((rc = f->d_cand_cnt.reserve((size_t)n_slabs * n_q * 4))) return rc;
    uint32_t h_over[4] = {0, 0, 0, 0};
    for (int round = 0; round < 10; round++) {
        TSGPU_HIP_TRY(hipMemsetAsync(f->d_cand_cnt.p, 0, (size_t)n_slabs * n_q * 4, s));
        TSGPU_HIP_TRY(hipMemsetAsync(d_over, 0, 16, s));
        VecHScanArgs a = base;
        a.n_ord = n_tiles; a.tile_stride = 1; a.mode = 0;
        a.seg = f->d_cand.as<uint64_t>(); a.seg_cnt = f->d_cand_cnt.as<uint32_t>(); a.seg_cap = (uint32_t)seg_cap;
        if (record_events && round == 0) TSGPU_HIP_TRY(hipEventRecord(ctx->ev[6], s));
        launch_scan(a, 256);
        if (record_events && round == 0) { TSGPU_HIP_TRY(hipEventRecord(ctx->ev[7], s)); TSGPU_HIP_TRY(hipEventRecord(ctx->ev[4], s)); ctx->scan_events_valid = true; }
        hipLaunchKernelGGL(vec_refine_kernel, dim3(n_q), dim3(VEC_THREADS), 0, s, (const uint64_t*)a.seg, (const uint32_t*)a.seg_cnt, a.n_slabs, n_q, a.seg_cap, k,
                           (const float*)f->d_cq.as<float>(), (const float*)f->xnorm.as<float>(), f->d_L1.as<float>(), f->d_surv.as<uint32_t>(), VEC_SURV_CAP,
                           f->d_surv_cnt.as<uint32_t>(), d_over, f->d_gkeys.as<uint32_t>(), VEC_REFINE_GCAP);
        TSGPU_HIP_TRY(hipGetLastError());
        TSGPU_HIP_TRY(hipMemcpyAsync(h_over, d_over, 16, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipStreamSynchronize(s));
        if (!h_over[0] || h_over[1]) break;  // an overflowing list raised its own L1: scan again (unless it is stuck)
        ctx->vec_overflow_rounds++;
        if (h_over[2]) {
            // segments too small for this data (a query's candidates are the cluster around its k-th neighbour, not ~8k rows): grow them
            const uint64_t grown = seg_cap * 4;
            if (grown * n_slabs * n_q * 8 <= arena_limit && !ctx->vec_cand_cap) {
                seg_cap = grown;
                f->seg_cap_hint = std::max<uint64_t>(f->seg_cap_hint, seg_cap);
                if ((rc = f->d_cand.reserve((size_t)n_slabs * n_q * seg_cap * 8))) return rc;
            } else if (h_over[3]) { h_over[1] = 1; break; }                    // cannot grow and the bound cannot move: stuck
        }
    }
    if (h_over[0]) {
        // brackets cannot separate this data (mass ties / more near-duplicates than the survivor arena): the fp32 scan's
        // (distance, row) keys converge on any input — run the group there
        ctx->vec_prefilter_fallbacks++;
        return knn_group(ctx, f, Q_dev, n_q, k, mask_dev, dist_dev, label_dev, cnt_dev, record_events);
    }
    // exact re-score of the survivors (hnswlib's summation order) and final selection
    hipLaunchKernelGGL(vec_rescore_kernel, dim3(n_q, 4), dim3(VEC_THREADS), 0, s, (const float*)f->X.as<float>(), Q_dev, f->dim,
                       (const uint32_t*)f->d_surv.as<uint32_t>(), (const uint32_t*)f->d_surv_cnt.as<uint32_t>(), (size_t)VEC_SURV_CAP, f->d_dense.as<uint64_t>(), ctx->vec_ip_lanes);
    hipLaunchKernelGGL(vec_select_kernel, dim3(n_q), dim3(VEC_THREADS), 0, s, (const uint64_t*)f->d_dense.as<uint64_t>(), (size_t)VEC_SURV_CAP,
                       (const uint32_t*)f->d_surv_cnt.as<uint32_t>(), VEC_SURV_CAP, k, 0, f->labels.as<uint64_t>(), dist_dev, label_dev, cnt_dev,
                       f->d_tau.as<uint64_t>(), d_over);
    TSGPU_HIP_TRY(hipGetLastError());
    ctx->vec_prefilter_groups++;
    if (ctx->vec_count_rescored) {        // introspection (tests / bench): how many rows reached the exact re-score
        std::vector<uint32_t> sc(n_q);
        TSGPU_HIP_TRY(hipMemcpyAsync(sc.data(), f->d_surv_cnt.p, (size_t)n_q * 4, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipStreamSynchronize(s));
        ctx->vec_rescored_rows = 0;
        for (uint32_t i = 0; i < n_q; i++) ctx->vec_rescored_rows += sc[i];
    }
    if (record_events) TSGPU_HIP_TRY(hipEventRecord(ctx->ev[5], s));
    return TSGPU_OK;
}

static int knn_device(tsgpu_ctx* ctx, VecField* f, const float* Q_dev, uint32_t n_q, uint32_t k, const uint8_t* mask_dev,
                      float* dist_dev, uint64_t* label_dev, uint32_t* cnt_dev) {
    // query groups bound the scratch (dense sample + candidate arena); each group streams X once
    const uint32_t GROUP = 512;
    ctx->scan_events_valid = false;
    for (uint32_t q0 = 0; q0 < n_q; q0 += GROUP) {
        const uint32_t nq = std::min<uint32_t>(GROUP, n_q - q0);
        int rc = ctx->vec_prefilter
            ? knn_group_prefilter(ctx, f, Q_dev + (size_t)q0 * f->dim, nq, k, mask_dev, dist_dev + (size_t)q0 * k, label_dev + (size_t)q0 * k, cnt_dev + q0, q0 == 0)
            : knn_group(ctx, f, Q_dev + (size_t)q0 * f->dim, nq, k, mask_dev, dist_dev + (size_t)q0 * k, label_dev + (size_t)q0 * k, cnt_dev + q0, q0 == 0);
        if (rc) return rc;
    }
    ctx->timings.vec_flops = 2ull * f->n_rows * f->dim * std::min<uint32_t>(n_q, GROUP);   // the HIP events bracket the first query group
    // bytes one main-scan launch must stream: the row matrix once (bf16 mirror, or fp32 rows on the fp32 scan) + the queries
    ctx->timings.vec_scan_bytes = ctx->vec_prefilter ? (uint64_t)f->n_rows * f->dimp * 2 + (uint64_t)std::min<uint32_t>(n_q, GROUP) * f->dimp * 2
                                                     : (uint64_t)f->n_rows * f->dim * 4 + (uint64_t)std::min<uint32_t>(n_q, GROUP) * f->dim * 4;
    return TSGPU_OK;
}

static void knn_collect_timings(tsgpu_ctx* ctx) {
    float a = 0, b = 0;
    (void)hipEventElapsedTime(&a, ctx->ev[3], ctx->ev[4]);
    (void)hipEventElapsedTime(&b, ctx->ev[4], ctx->ev[5]);
    float c = 0;
    if (ctx->scan_events_valid) (void)hipEventElapsedTime(&c, ctx->ev[6], ctx->ev[7]);
    std::lock_guard<std::mutex> tl(ctx->tm_mu);
    ctx->timings.vec_knn_ms = a;
    ctx->timings.vec_merge_ms = b;
    ctx->timings.total_ms = a + b;
    ctx->timings.vec_scan_ms = c;
}

// row mask for deleted rows / allow list / excluded ids; returns nullptr (all rows ok) when nothing restricts
static int build_mask(tsgpu_ctx* ctx, VecField* f, const uint32_t* allow_ids, uint32_t n_allow, const uint32_t* excluded_ids,
                      uint32_t n_excluded, const uint8_t** mask_dev) {
    *mask_dev = nullptr;
    if (!f->any_deleted && !allow_ids && n_excluded == 0) return TSGPU_OK;
    if (!allow_ids && n_excluded == 0) { *mask_dev = f->row_ok.as<uint8_t>(); return TSGPU_OK; }
    std::vector<uint8_t> m(f->n_rows, allow_ids ? 0 : 1);
    uint32_t row;
    if (allow_ids) for (uint32_t i = 0; i < n_allow; i++) if (f->find_row(allow_ids[i], row)) m[row] = 1;
    for (uint32_t i = 0; i < n_excluded; i++) if (f->find_row(excluded_ids[i], row)) m[row] = 0;
    if (f->any_deleted) for (size_t r = 0; r < f->n_rows; r++) if (!f->h_ok[r]) m[r] = 0;
    int rc = f->d_mask.reserve(std::max<size_t>(m.size(), 1));
    if (rc) return rc;
    if (!m.empty()) TSGPU_HIP_TRY(hipMemcpyAsync(f->d_mask.p, m.data(), m.size(), hipMemcpyHostToDevice, ctx->stream));
    TSGPU_HIP_TRY(hipStreamSynchronize(ctx->stream));
    *mask_dev = f->d_mask.as<uint8_t>();
    return TSGPU_OK;
}

// queries on the device, normalised for cosine fields (src/index.cpp:3381-3384)
static int stage_queries(tsgpu_ctx* ctx, VecField* f, const float* Q, int mem_q, uint32_t n_q, const float** Q_dev) {
    hipStream_t s = ctx->stream;
    const size_t bytes = (size_t)n_q * f->dim * 4;
    if (mem_q == TSGPU_MEM_DEVICE && f->metric != TSGPU_METRIC_COSINE) { *Q_dev = Q; return TSGPU_OK; }
    int rc = f->dQ.reserve(std::max<size_t>(bytes, 16));
    if (rc) return rc;
    TSGPU_HIP_TRY(hipMemcpyAsync(f->dQ.p, Q, bytes, mem_q == TSGPU_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    if (f->metric == TSGPU_METRIC_COSINE)
        hipLaunchKernelGGL(vec_normalize_rows_kernel, dim3((n_q + 63) / 64), dim3(64), 0, s, f->dQ.as<float>(), n_q, f->dim);
    *Q_dev = f->dQ.as<float>();
    return TSGPU_OK;
}

struct KnnHost { std::vector<float> dist; std::vector<uint64_t> lab; std::vector<uint32_t> cnt; };

// knn with host results (used by the vector-search and hybrid entry points); takes the lock itself
static int knn_to_host(tsgpu_ctx* ctx, uint32_t field_id, const float* Q, int mem_q, uint32_t n_q, uint32_t k,
                       const uint32_t* allow, uint32_t n_allow, const uint32_t* excl, uint32_t n_excl, KnnHost& out) {
    out.dist.assign((size_t)n_q * k, 0.f);
    out.lab.assign((size_t)n_q * k, 0);
    out.cnt.assign(n_q, 0);
    return tsgpu_vec_knn_batch(ctx, field_id, Q, mem_q, n_q, k, allow, n_allow, excl, n_excl, out.dist.data(), out.lab.data(), out.cnt.data(),
                               TSGPU_MEM_HOST);
}

// compute_sort_scores (src/index.cpp:5662-5907) for an already-found hit, on the host mirror of the columns
static void host_sort_scores(tsgpu_ctx* ctx, const tsgpu_sort_by* sort, uint32_t n_sort, uint64_t seq_id, int64_t max_field_match_score,
                             float vector_distance, int64_t* scores, int64_t& match_score_index) {
    for (uint32_t i = 0; i < n_sort && i < 3; i++) {
        int64_t v = 0;
        switch (sort[i].kind) {
            case TSGPU_SORT_TEXT_MATCH: v = max_field_match_score; match_score_index = i; break;
            case TSGPU_SORT_SEQ_ID: v = (int64_t)seq_id; break;
            case TSGPU_SORT_VECTOR_DISTANCE: v = float_to_int64(vector_distance); break;
            default: {
                const uint32_t c = sort[i].column;
                v = (c < ctx->columns.size() && seq_id < ctx->columns[c].host.size()) ? ctx->columns[c].host[seq_id] : INT64_MIN;
            }
        }
        if (sort[i].order == -1) v = (int64_t)(0ull - (uint64_t)v);
        scores[i] = v;
    }
}

static void write_hits(const HostTopster& t, uint32_t q, tsgpu_hits* out) {
    const size_t base = (size_t)q * out->k_stride;
    const uint32_t n = std::min<uint32_t>(t.size, out->k_stride);
    for (uint32_t i = 0; i < n; i++) {
        const HostKV* kv = t.kvs[i];
        out->keys[base + i] = kv->key;
        for (int j = 0; j < 3; j++) out->scores[(base + i) * 3 + j] = kv->scores[j];
        if (out->text_match) out->text_match[base + i] = kv->text_match_score;
        if (out->vector_distance) out->vector_distance[base + i] = kv->vector_distance;
        if (out->match_score_index) out->match_score_index[base + i] = kv->match_score_index;
    }
    out->n_hits[q] = n;
}

}  // namespace tsgpu

extern "C" {

void tsgpu_vec_destroy_all(tsgpu_ctx* ctx) {
    for (auto& kv : ctx->vec_fields) { kv.second->release(); delete kv.second; }
    ctx->vec_fields.clear();
}

uint64_t tsgpu_vec_device_bytes(tsgpu_ctx* ctx) {
    uint64_t b = 0;
    for (auto& kv : ctx->vec_fields) b += kv.second->X.cap + kv.second->labels.cap + kv.second->row_ok.cap + kv.second->Xh.cap + kv.second->xnorm.cap + kv.second->tile_nmax.cap;
    return b;
}

int tsgpu_vec_create(tsgpu_ctx* ctx, uint32_t vec_field_id, uint32_t dim, int metric, uint64_t capacity_hint) {
    if (!ctx) return fail(TSGPU_ERR_INVALID, "ctx is NULL");
    if (dim == 0 || dim > 65536) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_create: bad dim");
    if (metric != TSGPU_METRIC_IP && metric != TSGPU_METRIC_COSINE) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_create: metric must be ip or cosine (include/field.h:92-95)");
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    if (ctx->vec_fields.count(vec_field_id)) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_create: field exists");
    VecField* f = new (std::nothrow) VecField;
    if (!f) return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_vec_create: host allocation failed");
    f->dim = dim;
    f->dimp = (dim + 63) / 64 * 64;
    f->metric = metric;
    int rc = vec_reserve_rows(f, std::max<uint64_t>(capacity_hint, 16), ctx->stream);   // include/index.h:367: init capacity 16
    if (rc) { f->release(); delete f; return rc; }
    ctx->vec_fields[vec_field_id] = f;
    return ok();
}

int tsgpu_vec_upsert(tsgpu_ctx* ctx, uint32_t vec_field_id, const uint64_t* labels, const float* data, uint32_t n, int mem) {
    if (!ctx) return fail(TSGPU_ERR_INVALID, "ctx is NULL");
    if (n == 0) return ok();
    if (!labels || !data) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_upsert: NULL array");
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    VecField* f = get_field(ctx, vec_field_id);
    if (!f) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_vec_upsert: unknown vector field");
    hipStream_t s = ctx->stream;
    try {
        std::vector<uint64_t> hl(n);
        if (mem == TSGPU_MEM_DEVICE) TSGPU_HIP_TRY(hipMemcpy(hl.data(), labels, (size_t)n * 8, hipMemcpyDeviceToHost));
        else std::copy(labels, labels + n, hl.begin());
        // fast path: n brand-new labels continuing the identity numbering -> one bulk append
        bool bulk = true;
        if (f->identity) {
            for (uint32_t i = 0; i < n && bulk; i++) bulk = hl[i] == f->n_rows + i;
            if (!bulk) f->break_identity();        // e.g. a doc-range shard whose labels start at its range's first seq_id
        }
        if (!f->identity) {                         // bulk = every label is new and unique inside the batch
            bulk = true;
            for (uint32_t i = 0; i < n && bulk; i++) { uint32_t row; bulk = !f->find_row(hl[i], row); }
            if (bulk) {
                std::vector<uint64_t> tmp(hl);
                std::sort(tmp.begin(), tmp.end());
                bulk = std::adjacent_find(tmp.begin(), tmp.end()) == tmp.end();
            }
        }
        // the in-library graph follows addPoint(.., replace_deleted = true): while deleted slots exist, new labels RE-USE them (one by one, below)
        if (bulk && f->hb && !f->hb->stale && !f->hb->deleted_stack.empty()) bulk = false;
        if (f->n_rows + n > 0xFFFFFFF0ull) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_vec_upsert: more than 2^32 rows");
        const hipMemcpyKind kind = mem == TSGPU_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        if (bulk) {
            int rc = vec_reserve_rows(f, f->n_rows + n, s);
            if (rc) return rc;
            float* dst = f->X.as<float>() + (size_t)f->n_rows * f->dim;
            TSGPU_HIP_TRY(hipMemcpyAsync(dst, data, (size_t)n * f->dim * 4, kind, s));
            if (f->metric == TSGPU_METRIC_COSINE)   // src/index.cpp:1049-1052
                hipLaunchKernelGGL(vec_normalize_rows_kernel, dim3((n + 63) / 64), dim3(64), 0, s, dst, n, f->dim);
            TSGPU_HIP_TRY(hipMemcpyAsync(f->labels.as<uint64_t>() + f->n_rows, hl.data(), (size_t)n * 8, hipMemcpyHostToDevice, s));
            TSGPU_HIP_TRY(hipMemsetAsync(f->row_ok.as<uint8_t>() + f->n_rows, 1, n, s));
            if ((rc = vec_refresh_mirror(f, (uint32_t)f->n_rows, n, f->n_rows + n, s))) return rc;
            TSGPU_HIP_TRY(hipStreamSynchronize(s));
            if (!f->identity) for (uint32_t i = 0; i < n; i++) f->row_of.emplace(hl[i], (uint32_t)(f->n_rows + i));
            f->h_labels.insert(f->h_labels.end(), hl.begin(), hl.end());
            f->h_ok.insert(f->h_ok.end(), n, 1);
            f->n_rows += n;
            f->n_live += n;
            if (f->hb && !f->hb->stale) {            // hnswlib addPoint for the new rows, in row order, on the rows AS STORED (cosine: normalised)
                std::vector<float> rows((size_t)n * f->dim);
                TSGPU_HIP_TRY(hipMemcpy(rows.data(), dst, rows.size() * 4, hipMemcpyDeviceToHost));
                f->hb->add_batch(rows.data(), n);
            }
        } else {
            f->break_identity();
            const bool graph = f->hb && !f->hb->stale;
            for (uint32_t i = 0; i < n; i++) {
                uint32_t row;
                const bool exists = f->find_row(hl[i], row);
                const bool live = exists && f->h_ok[row];
                // hnswlib addPoint(vec, label, replace_deleted = true), as the reference calls it (src/index.cpp:1052-1054; csrc/tsgpu_hnsw_build.h):
                //   a LIVE label            -> its row is overwritten, the graph runs updatePoint on it;
                //   otherwise, graph on     -> the most recently deleted row is RE-USED (its label moves: the old label loses its mapping), unmarked, updatePoint;
                //                              without a deleted row the new label is appended (addPoint);
                //   otherwise, no graph     -> a deleted label is revived in its own row, a new one appended.
                int mode = live ? 0 : (exists ? 1 : 2);            // 0 overwrite live, 1 revive own row, 2 append, 3 re-use another / own deleted row (graph)
                if (!live && graph) {
                    const int64_t slot = f->hb->take_deleted_slot();
                    if (slot >= 0) {
                        mode = 3;
                        const uint32_t r2 = (uint32_t)slot;
                        const uint64_t old_label = f->h_labels[r2];
                        auto it = f->row_of.find(old_label);
                        if (it != f->row_of.end() && it->second == r2) f->row_of.erase(it);
                        // (the label's own deleted row, when it is not the one re-used, stays deleted: its stale label no longer maps to it)
                        row = r2;
                        f->row_of[hl[i]] = row;
                        f->h_labels[row] = hl[i];
                        TSGPU_HIP_TRY(hipMemcpyAsync(f->labels.as<uint64_t>() + row, &hl[i], 8, hipMemcpyHostToDevice, s));
                    } else mode = 2;                               // (a deleted label always finds at least its own row: only new labels land here)
                }
                if (mode == 2) {
                    int rc = vec_reserve_rows(f, f->n_rows + 1, s);
                    if (rc) return rc;
                    row = (uint32_t)f->n_rows++;
                    f->row_of[hl[i]] = row;
                    f->h_labels.push_back(hl[i]);
                    f->h_ok.push_back(1);
                    f->n_live++;
                    TSGPU_HIP_TRY(hipMemcpyAsync(f->labels.as<uint64_t>() + row, &hl[i], 8, hipMemcpyHostToDevice, s));
                } else if (!f->h_ok[row]) { f->h_ok[row] = 1; f->n_live++; }
                float* dst = f->X.as<float>() + (size_t)row * f->dim;
                TSGPU_HIP_TRY(hipMemcpyAsync(dst, data + (size_t)i * f->dim, (size_t)f->dim * 4, kind, s));
                if (f->metric == TSGPU_METRIC_COSINE)
                    hipLaunchKernelGGL(vec_normalize_rows_kernel, dim3(1), dim3(64), 0, s, dst, 1u, f->dim);
                TSGPU_HIP_TRY(hipMemsetAsync(f->row_ok.as<uint8_t>() + row, 1, 1, s));
                { int rc2 = vec_refresh_mirror(f, row, 1, f->n_rows, s); if (rc2) return rc2; }
                TSGPU_HIP_TRY(hipStreamSynchronize(s));
                if (graph) {
                    std::vector<float> one(f->dim);                // the row AS STORED (cosine: normalised)
                    TSGPU_HIP_TRY(hipMemcpy(one.data(), dst, (size_t)f->dim * 4, hipMemcpyDeviceToHost));
                    if (mode == 2) f->hb->add_batch(one.data(), 1);
                    else if (mode == 3) f->hb->replace_deleted(row, one.data());
                    else if (mode == 0) f->hb->update_point(row, one.data());
                    // (mode 1 cannot happen with the graph on: a deleted label takes the slot path)
                }
            }
        }
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_vec_upsert: host allocation failed"); }
    return ok();
}

int tsgpu_vec_delete(tsgpu_ctx* ctx, uint32_t vec_field_id, uint64_t label) {
    if (!ctx) return fail(TSGPU_ERR_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    VecField* f = get_field(ctx, vec_field_id);
    if (!f) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_vec_delete: unknown vector field");
    uint32_t row;
    if (!f->find_row(label, row) || !f->h_ok[row]) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_vec_delete: label not found");   // markDelete throws
    f->h_ok[row] = 0;
    f->any_deleted = true;
    f->n_live--;
    if (f->hb) f->hb->mark_deleted(row);                                   // markDelete: construction beams no longer keep it as a result; the slot becomes re-usable
    TSGPU_HIP_TRY(hipMemset(f->row_ok.as<uint8_t>() + row, 0, 1));
    return ok();
}

int tsgpu_vec_get(tsgpu_ctx* ctx, uint32_t vec_field_id, uint64_t label, float* out_host) {
    if (!ctx || !out_host) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_get: NULL argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    VecField* f = get_field(ctx, vec_field_id);
    if (!f) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_vec_get: unknown vector field");
    uint32_t row;
    if (!f->find_row(label, row) || !f->h_ok[row]) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_vec_get: label not found");
    TSGPU_HIP_TRY(hipMemcpy(out_host, f->X.as<float>() + (size_t)row * f->dim, (size_t)f->dim * 4, hipMemcpyDeviceToHost));
    return ok();
}

uint64_t tsgpu_vec_count(tsgpu_ctx* ctx, uint32_t vec_field_id) {
    if (!ctx) return 0;
    std::lock_guard<std::mutex> lk(ctx->mu);
    VecField* f = get_field(ctx, vec_field_id);
    return f ? f->n_rows : 0;      // getCurrentElementCount counts deleted slots too
}

}  // extern "C"

namespace tsgpu {
struct VecRequest : ParkedRequest {
    uint32_t field = 0, k = 0;
    const float* Q = nullptr;                        // host, [units][dim]
    float* dist_out = nullptr; uint64_t* label_out = nullptr; uint32_t* n_out = nullptr;   // host
};
}
static int vec_knn_locked(tsgpu_ctx* ctx, uint32_t vec_field_id, const float* Q, int mem_q, uint32_t n_q, uint32_t k, const uint32_t* allow_ids,
                          uint32_t n_allow, const uint32_t* excluded_ids, uint32_t n_excluded, float* dist_out, uint64_t* label_out,
                          uint32_t* n_out, int mem_out);

// A small unfiltered k-NN call from one of several concurrent request threads (the reference: vecdex->searchKnnCloserFirst per
// request thread, src/index.cpp:3384-3386): the parked calls of one (field, k) run as ONE batch — one pass over the row matrix
// serves all of them, which is where the batched scan's throughput comes from.
static int vec_knn_coalesced(tsgpu_ctx* ctx, uint32_t field, const float* Q, uint32_t n_q, uint32_t k, float* dist_out, uint64_t* label_out, uint32_t* n_out) {
    VecRequest me;
    me.units = n_q; me.field = field; me.k = k; me.Q = Q; me.dist_out = dist_out; me.label_out = label_out; me.n_out = n_out;
    typedef std::unique_lock<std::mutex> Lock;
    auto acquire = [&]() { return std::unique_ptr<Lock>(new Lock(ctx->mu)); };
    const uint32_t round_cap = std::max<uint32_t>(ctx->batch_round_queries, n_q);
    auto pick = [&](std::vector<VecRequest*>& pending, std::vector<VecRequest*>& round) {
        const uint32_t f0 = pending[0]->field, k0 = pending[0]->k;
        uint32_t units = 0;
        std::vector<VecRequest*> rest;
        for (VecRequest* r : pending) {
            if (r->field == f0 && r->k == k0 && (round.empty() || units + r->units <= round_cap)) { round.push_back(r); units += r->units; }
            else rest.push_back(r);
        }
        pending.swap(rest);
    };
    auto exec = [&](std::vector<VecRequest*>& round, std::unique_ptr<Lock>&) {
        int rc = TSGPU_OK;
        std::string err;
        try {
            VecField* f = get_field(ctx, round[0]->field);
            if (!f) { rc = TSGPU_ERR_NOT_FOUND; err = "tsgpu_vec_knn_batch: unknown vector field"; }
            else {
                uint32_t total = 0;
                for (VecRequest* r : round) total += r->units;
                const uint32_t kk = round[0]->k;
                std::vector<float> Qall((size_t)total * f->dim), d((size_t)total * kk);
                std::vector<uint64_t> l((size_t)total * kk);
                std::vector<uint32_t> c(total);
                uint32_t at = 0;
                for (VecRequest* r : round) { memcpy(Qall.data() + (size_t)at * f->dim, r->Q, (size_t)r->units * f->dim * 4); at += r->units; }
                rc = vec_knn_locked(ctx, round[0]->field, Qall.data(), TSGPU_MEM_HOST, total, kk, nullptr, 0, nullptr, 0, d.data(), l.data(), c.data(), TSGPU_MEM_HOST);
                if (rc != TSGPU_OK) err = tls_error();
                else {
                    at = 0;
                    for (VecRequest* r : round) {
                        memcpy(r->dist_out, d.data() + (size_t)at * kk, (size_t)r->units * kk * 4);
                        memcpy(r->label_out, l.data() + (size_t)at * kk, (size_t)r->units * kk * 8);
                        memcpy(r->n_out, c.data() + at, (size_t)r->units * 4);
                        at += r->units;
                    }
                }
            }
        } catch (const std::bad_alloc&) { rc = TSGPU_ERR_NO_MEMORY; err = "tsgpu_vec_knn_batch: host allocation failed"; }
        for (VecRequest* r : round) { r->rc = rc; r->err = err; }
    };
    ctx->vec_comb.run(me, ctx->vec_callers, ctx->batch_window_us, acquire, pick, exec, ctx->vec_batch_post_window_us);
    if (me.rc != TSGPU_OK) return fail(me.rc, me.err);
    return ok();
}

extern "C" {

int tsgpu_vec_knn_batch(tsgpu_ctx* ctx, uint32_t vec_field_id, const float* Q, int mem_q, uint32_t n_q, uint32_t k, const uint32_t* allow_ids,
                        uint32_t n_allow, const uint32_t* excluded_ids, uint32_t n_excluded, float* dist_out, uint64_t* label_out,
                        uint32_t* n_out, int mem_out) {
    if (!ctx || !Q || !dist_out || !label_out || !n_out) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_knn_batch: NULL argument");
    if (n_q == 0) return ok();
    if (k == 0) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_knn_batch: k must be > 0");
    if (k > TSGPU_MAX_TOPK) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_vec_knn_batch: k > TSGPU_MAX_TOPK is not accelerated");
    struct CallerCount { std::atomic<int>& c; explicit CallerCount(std::atomic<int>& x) : c(x) { c.fetch_add(1); } ~CallerCount() { c.fetch_sub(1); } } cc(ctx->vec_callers);
    if (mem_q == TSGPU_MEM_HOST && mem_out == TSGPU_MEM_HOST && !allow_ids && n_excluded == 0 && n_q <= ctx->batch_max_queries && ctx->vec_callers.load() > 1)
        return vec_knn_coalesced(ctx, vec_field_id, Q, n_q, k, dist_out, label_out, n_out);
    std::lock_guard<std::mutex> lk(ctx->mu);
    return vec_knn_locked(ctx, vec_field_id, Q, mem_q, n_q, k, allow_ids, n_allow, excluded_ids, n_excluded, dist_out, label_out, n_out, mem_out);
}

}  // extern "C"

// ctx->mu held by the caller
static int vec_knn_locked(tsgpu_ctx* ctx, uint32_t vec_field_id, const float* Q, int mem_q, uint32_t n_q, uint32_t k, const uint32_t* allow_ids,
                          uint32_t n_allow, const uint32_t* excluded_ids, uint32_t n_excluded, float* dist_out, uint64_t* label_out,
                          uint32_t* n_out, int mem_out) {
    (void)hipSetDevice(ctx->device);
    VecField* f = get_field(ctx, vec_field_id);
    if (!f) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_vec_knn_batch: unknown vector field");
    hipStream_t s = ctx->stream;
    try {
        const float* Q_dev = nullptr;
        int rc = stage_queries(ctx, f, Q, mem_q, n_q, &Q_dev);
        if (rc) return rc;
        const uint8_t* mask = nullptr;
        if ((rc = build_mask(ctx, f, allow_ids, n_allow, excluded_ids, n_excluded, &mask))) return rc;
        float* d_dist = dist_out;
        uint64_t* d_lab = label_out;
        uint32_t* d_cnt = n_out;
        if (mem_out != TSGPU_MEM_DEVICE) {
            if ((rc = f->d_dist.reserve((size_t)n_q * k * 4))) return rc;
            if ((rc = f->d_lab.reserve((size_t)n_q * k * 8))) return rc;
            if ((rc = f->d_cnt.reserve((size_t)n_q * 4))) return rc;
            d_dist = f->d_dist.as<float>(); d_lab = f->d_lab.as<uint64_t>(); d_cnt = f->d_cnt.as<uint32_t>();
        }
        if (f->n_rows == 0) {
            TSGPU_HIP_TRY(hipMemsetAsync(d_cnt, 0, (size_t)n_q * 4, s));
        } else if ((rc = knn_device(ctx, f, Q_dev, n_q, k, mask, d_dist, d_lab, d_cnt))) return rc;
        if (mem_out != TSGPU_MEM_DEVICE) {
            TSGPU_HIP_TRY(hipMemcpyAsync(dist_out, d_dist, (size_t)n_q * k * 4, hipMemcpyDeviceToHost, s));
            TSGPU_HIP_TRY(hipMemcpyAsync(label_out, d_lab, (size_t)n_q * k * 8, hipMemcpyDeviceToHost, s));
            TSGPU_HIP_TRY(hipMemcpyAsync(n_out, d_cnt, (size_t)n_q * 4, hipMemcpyDeviceToHost, s));
        }
        TSGPU_HIP_TRY(hipStreamSynchronize(s));
        if (f->n_rows) knn_collect_timings(ctx);
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_vec_knn_batch: host allocation failed"); }
    return ok();
}

extern "C" {

// ---------------------------------------------------------------- HNSW graph mirror + search (seam B2, a18)
// ctx->mu held; the lists have been validated (tsgpu_vec_hnsw_load) or come from the library's own builder
static int hnsw_upload_locked(tsgpu_ctx* ctx, VecField* f, uint32_t M, int32_t maxlevel, uint32_t enterpoint, const uint32_t* link0, const uint64_t* upper_ptr,
                              const uint32_t* upper_links, uint32_t n) {
    hipStream_t s = ctx->stream;
    const uint64_t n_upper = upper_ptr[n];
    int rc;
    if ((rc = f->g_link0.reserve((size_t)std::max<uint32_t>(n, 1) * (1 + 2 * M) * 4))) return rc;
    if ((rc = f->g_upper_ptr.reserve((size_t)(n + 1) * 8))) return rc;
    if ((rc = f->g_upper_links.reserve((size_t)std::max<uint64_t>(n_upper, 1) * (1 + M) * 4))) return rc;
    TSGPU_HIP_TRY(hipMemcpyAsync(f->g_link0.p, link0, (size_t)n * (1 + 2 * M) * 4, hipMemcpyHostToDevice, s));
    TSGPU_HIP_TRY(hipMemcpyAsync(f->g_upper_ptr.p, upper_ptr, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, s));
    if (n_upper) TSGPU_HIP_TRY(hipMemcpyAsync(f->g_upper_links.p, upper_links, (size_t)n_upper * (1 + M) * 4, hipMemcpyHostToDevice, s));
    // visited bookkeeping is allocated by the search (hash sets by default; 16-bit tags with option hnsw_visited_hash = 0)
    const uint32_t slots = 4096;
    f->g_tag_slots = 0;
    if ((rc = f->g_stat.reserve(64))) return rc;
    TSGPU_HIP_TRY(hipStreamSynchronize(s));
    f->g_M = M; f->g_n = n; f->g_slots = slots; f->g_epoch = 1; f->g_maxlevel = maxlevel; f->g_enterpoint = enterpoint; f->g_loaded = true;
    return TSGPU_OK;
}

int tsgpu_vec_hnsw_load(tsgpu_ctx* ctx, uint32_t vec_field_id, uint32_t M, int32_t maxlevel, uint32_t enterpoint, const uint32_t* link0,
                        const uint64_t* upper_ptr, const uint32_t* upper_links, uint32_t n) {
    if (!ctx || !link0 || !upper_ptr) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_hnsw_load: NULL argument");
    if (M < 2 || M > 31) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_vec_hnsw_load: M must be 2..31 (a level-0 list of 2M ids is fetched by one wavefront)");
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    VecField* f = get_field(ctx, vec_field_id);
    if (!f) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_vec_hnsw_load: unknown vector field");
    if (f->hb) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_hnsw_load: this field builds its own graph (tsgpu_vec_hnsw_enable)");
    if (n != f->n_rows) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_hnsw_load: the graph must cover exactly the rows of the field (hnswlib internal id = insertion order)");
    if (n && enterpoint >= n) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_hnsw_load: entry point out of range");
    hipStream_t s = ctx->stream;
    const uint64_t n_upper = upper_ptr[n];
    if (n_upper && !upper_links) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_hnsw_load: upper_links is NULL");
    // the device follows these links unchecked: link counts within 2M / M, neighbour ids inside the graph, upper_ptr monotone
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t* l0 = link0 + (size_t)i * (1 + 2 * M);
        if (l0[0] > 2 * M) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_hnsw_load: a level-0 link count exceeds 2M");
        for (uint32_t j = 0; j < l0[0]; j++) if (l0[1 + j] >= n) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_hnsw_load: a level-0 neighbour id is out of range");
        if (upper_ptr[i + 1] < upper_ptr[i]) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_hnsw_load: upper_ptr must be non-decreasing");
        if ((int64_t)(upper_ptr[i + 1] - upper_ptr[i]) > (int64_t)std::max<int32_t>(maxlevel, 0)) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_hnsw_load: a node has more upper levels than maxlevel");
    }
    for (uint64_t u = 0; u < n_upper; u++) {
        const uint32_t* lu = upper_links + u * (1 + M);
        if (lu[0] > M) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_hnsw_load: an upper-level link count exceeds M");
        for (uint32_t j = 0; j < lu[0]; j++) if (lu[1 + j] >= n) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_hnsw_load: an upper-level neighbour id is out of range");
    }
    if (n && maxlevel >= 0 && (int64_t)(upper_ptr[enterpoint + 1] - upper_ptr[enterpoint]) < (int64_t)maxlevel)
        return fail(TSGPU_ERR_INVALID, "tsgpu_vec_hnsw_load: the entry point does not reach maxlevel");
    const int urc = hnsw_upload_locked(ctx, f, M, maxlevel, enterpoint, link0, upper_ptr, upper_links, n);
    return urc ? urc : ok();
}

int tsgpu_vec_hnsw_enable(tsgpu_ctx* ctx, uint32_t vec_field_id, uint32_t M, uint32_t ef_construction, uint32_t seed, uint32_t n_threads) {
    if (!ctx) return fail(TSGPU_ERR_INVALID, "ctx is NULL");
    if (M < 2 || M > 31) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_vec_hnsw_enable: M must be 2..31 (a level-0 list of 2M ids is fetched by one wavefront)");
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    VecField* f = get_field(ctx, vec_field_id);
    if (!f) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_vec_hnsw_enable: unknown vector field");
    try {
        f->hb.reset(new HnswBuilder);
        f->hb->init(f->dim, M, ef_construction, seed, n_threads, (int)ctx->vec_ip_lanes);
        f->g_loaded = false;
        if (f->n_rows) {                                 // rows that are already there: inserted in row order (= the order they were added in)
            std::vector<float> rows((size_t)f->n_rows * f->dim);
            TSGPU_HIP_TRY(hipMemcpy(rows.data(), f->X.p, rows.size() * 4, hipMemcpyDeviceToHost));
            f->hb->add_batch(rows.data(), f->n_rows);
            for (size_t r = 0; r < f->n_rows; r++) if (!f->h_ok[r]) f->hb->mark_deleted((uint32_t)r);     // (rows deleted before the graph was switched on: in row order)
        }
    } catch (const std::bad_alloc&) { f->hb.reset(); return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_vec_hnsw_enable: host allocation failed"); }
      catch (const std::system_error&) { f->hb.reset(); return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_vec_hnsw_enable: could not start an insertion thread"); }
    return ok();
}

int tsgpu_vec_hnsw_export(tsgpu_ctx* ctx, uint32_t vec_field_id, int32_t info[4], uint32_t* levels, uint32_t* link0, uint64_t* upper_ptr, uint32_t* upper_links,
                          uint64_t* n_upper) {
    if (!ctx || !info) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_hnsw_export: NULL argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    VecField* f = get_field(ctx, vec_field_id);
    if (f && !f->hb && f->g_loaded) {                  // a graph that lives on the device only (tsgpu_vec_hnsw_build, tsgpu_vec_hnsw_load): read back
        (void)hipSetDevice(ctx->device);
        const uint32_t n = f->g_n, M = f->g_M;
        std::vector<uint64_t> up((size_t)n + 1);
        TSGPU_HIP_TRY(hipMemcpy(up.data(), f->g_upper_ptr.p, up.size() * 8, hipMemcpyDeviceToHost));
        info[0] = (int32_t)n; info[1] = f->g_maxlevel; info[2] = (int32_t)f->g_enterpoint; info[3] = (int32_t)M;
        if (n_upper) *n_upper = up[n];
        if (levels) for (uint32_t i = 0; i < n; i++) levels[i] = (uint32_t)(up[i + 1] - up[i]);
        if (link0 && n) TSGPU_HIP_TRY(hipMemcpy(link0, f->g_link0.p, (size_t)n * (1 + 2 * M) * 4, hipMemcpyDeviceToHost));
        if (upper_ptr) std::copy(up.begin(), up.end(), upper_ptr);
        if (upper_links && up[n]) TSGPU_HIP_TRY(hipMemcpy(upper_links, f->g_upper_links.p, (size_t)up[n] * (1 + M) * 4, hipMemcpyDeviceToHost));
        return ok();
    }
    if (!f || !f->hb) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_vec_hnsw_export: no graph for this field (tsgpu_vec_hnsw_enable / tsgpu_vec_hnsw_build / tsgpu_vec_hnsw_load)");
    const HnswBuilder& hb = *f->hb;
    info[0] = (int32_t)hb.size(); info[1] = hb.maxlevel; info[2] = (int32_t)hb.enterpoint; info[3] = (int32_t)hb.M;
    if (n_upper) *n_upper = hb.n_upper_lists();
    if (levels) for (size_t i = 0; i < hb.size(); i++) levels[i] = (uint32_t)hb.levels[i];
    if (link0) std::copy(hb.link0.begin(), hb.link0.end(), link0);
    if (upper_ptr) { for (size_t i = 0; i < hb.size(); i++) upper_ptr[i] = hb.upper_at[i]; upper_ptr[hb.size()] = hb.n_upper_lists(); }
    if (upper_links) std::copy(hb.upper.begin(), hb.upper.end(), upper_links);
    return ok();
}

// the traversal kernel over n_q queries (Q_dev, or the rows q_rows[] of the field itself): LDS tier by max(ef, k), visited bookkeeping, the re-runs when a
// candidate heap or a visited set outgrows its tier. ctx->mu held; results stay on the device. labels = nullptr: internal ids come back.
static int hnsw_search_launch(tsgpu_ctx* ctx, VecField* f, const float* Q_dev, const uint32_t* q_rows, uint32_t n_q, uint32_t k, uint32_t ef, const uint8_t* mask, bool strict,
                              const uint64_t* labels, float* d_dist, uint64_t* d_lab, uint32_t* d_cnt, int build_layer = -1) {
    hipStream_t s = ctx->stream;
    int rc;
    const bool hash_mode = ctx->hnsw_visited_hash != 0;
    uint32_t slots = f->g_slots;
    if (!hash_mode) {
        // tag mode: one uint16 per row and concurrent query (hnswlib's VisitedListPool); option hnsw_visited_max_gib caps the array
        while (slots > 32 && (uint64_t)slots * std::max<uint32_t>(f->g_n, 1) * 2 > ((uint64_t)ctx->hnsw_visited_max_gib << 30)) slots >>= 1;
        if (f->g_tag_slots != slots) {
            if ((rc = f->g_visited.reserve((size_t)slots * std::max<uint32_t>(f->g_n, 1) * 2 + 64))) return rc;
            TSGPU_HIP_TRY(hipMemsetAsync(f->g_visited.p, 0, (size_t)slots * std::max<uint32_t>(f->g_n, 1) * 2 + 64, s));
            f->g_tag_slots = slots; f->g_epoch = 1;
        }
    }
    const uint32_t grid = std::min<uint32_t>(n_q, slots);
    const size_t tag_bytes = ((size_t)slots * f->g_n * 2 + 7) & ~(size_t)7;
    VecHnswArgs a;
    memset(&a, 0, sizeof a);
    a.X = f->X.as<float>(); a.Q = Q_dev; a.q_rows = q_rows; a.base_level = build_layer > 0 ? (uint32_t)build_layer : 0u; a.dim = f->dim; a.n_rows = (uint32_t)f->n_rows; a.n_q = n_q;
    a.link0 = f->g_link0.as<uint32_t>(); a.s0 = 1 + 2 * f->g_M;
    a.upper_ptr = f->g_upper_ptr.as<uint64_t>(); a.upper_links = f->g_upper_links.as<uint32_t>(); a.su = 1 + f->g_M;
    a.maxlevel = f->g_maxlevel; a.enterpoint = f->g_enterpoint;
    a.row_ok = mask; a.strict = strict ? 1u : 0u;
    a.k = k; a.ef = ef; a.ip_lanes = ctx->vec_ip_lanes; a.visited = hash_mode ? nullptr : f->g_visited.as<uint16_t>();
    a.overflow_cnt = f->g_stat.as<uint32_t>();
    a.labels = labels; a.dist_out = d_dist; a.label_out = d_lab; a.n_out = d_cnt;
    // LDS tier by max(ef, k): result heap 128 / 256 / 512 / 1024 entries, candidate heap 1024 / 1024 / 2048 / 4096 = 13.6 / 14.6 / 25 / 45 KB of LDS per query
    // = 11 / 10 / 6 / 3 queries in flight per CU (the 256 tier — round 6 — is the bulk build's: ef_construction 200 ran on the 512 tier's six before). The
    // queries whose candidate heap (or visited set) outgrows their tier — and only those — run again on the largest (round 6: before, one of them sent the
    // whole batch there). Smaller candidate heaps for the upper tiers were measured and lose: at ef 400 / 800 most queries outgrow 1024 / 2048 entries
    // (10M x 768: 105 K -> 48 K q/s, 31 K -> 26 K q/s with the re-runs).
    const uint32_t need = std::max(k, ef);
    int tier = need <= 128 ? 0 : (need <= 256 ? 1 : (need <= 512 ? 2 : 3));
    const int TOP = 3;
    const bool tiny = ctx->hnsw_test_tiny_cand && need <= 128;
    uint32_t vs_boost = 1, grid_now = grid, n_now = n_q;
    uint64_t tot_exp = 0, tot_dist = 0;
    std::vector<uint32_t> h_cnt, h_sel;
    for (;;) {
        const uint32_t grid_t = std::min<uint32_t>(n_now, grid);
        const uint32_t iters_t = (n_now + grid_t - 1) / grid_t;
        if (hash_mode) {
            // per-query visited sets: 64 x the tier's result-heap capacity (8 192 .. 65 536 words), one per concurrent query; a traversal
            // that outgrows the largest tier's set (large ef / k, strict filters: many visited, few admitted) runs again with sets 8x / 64x as
            // large and fewer queries in flight (<= 8 GiB of sets) instead of being reported as overflowed
            const uint32_t vs = (tier == 0 ? 8192u : (tier == 1 ? 16384u : (tier == 2 ? 32768u : 65536u))) * vs_boost;
            grid_now = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(grid_t, (8ull << 30) / ((uint64_t)vs * 4)));
            if ((rc = f->g_vhash.reserve((size_t)grid_now * vs * 4))) return rc;
            a.vhash = f->g_vhash.as<uint32_t>(); a.vhash_slots = vs;
        } else {
            grid_now = grid_t;
            if ((uint64_t)f->g_epoch + iters_t >= 0xFFF0ull) {      // tag space exhausted: clear the tags
                TSGPU_HIP_TRY(hipMemsetAsync(f->g_visited.p, 0, tag_bytes, s));
                f->g_epoch = 1;
            }
        }
        a.epoch_base = f->g_epoch;
        f->g_epoch += (n_now + grid_now - 1) / grid_now;
        a.n_q = n_now;
        TSGPU_HIP_TRY(hipMemsetAsync(a.overflow_cnt, 0, 56, s));
        if (build_layer > 0) {                           // the bulk build's beam on an upper layer
            if (tier == 0) hipLaunchKernelGGL((vec_hnsw_search_kernel<129, 1024, true>), dim3(grid_now), dim3(64), 0, s, a);
            else if (tier == 1) hipLaunchKernelGGL((vec_hnsw_search_kernel<257, 1024, true>), dim3(grid_now), dim3(64), 0, s, a);
            else if (tier == 2) hipLaunchKernelGGL((vec_hnsw_search_kernel<513, 2048, true>), dim3(grid_now), dim3(64), 0, s, a);
            else hipLaunchKernelGGL((vec_hnsw_search_kernel<VEC_HNSW_MAX_EF + 1, VEC_HNSW_CAND_CAP, true>), dim3(grid_now), dim3(64), 0, s, a);
        } else if (tier == 0 && tiny) hipLaunchKernelGGL((vec_hnsw_search_kernel<129, 24>), dim3(grid_now), dim3(64), 0, s, a);
        else if (tier == 0) hipLaunchKernelGGL((vec_hnsw_search_kernel<129, 1024>), dim3(grid_now), dim3(64), 0, s, a);
        else if (tier == 1) hipLaunchKernelGGL((vec_hnsw_search_kernel<257, 1024>), dim3(grid_now), dim3(64), 0, s, a);
        else if (tier == 2) hipLaunchKernelGGL((vec_hnsw_search_kernel<513, 2048>), dim3(grid_now), dim3(64), 0, s, a);
        else hipLaunchKernelGGL((vec_hnsw_search_kernel<VEC_HNSW_MAX_EF + 1, VEC_HNSW_CAND_CAP>), dim3(grid_now), dim3(64), 0, s, a);
        uint32_t h_stat[14] = {0};
        TSGPU_HIP_TRY(hipMemcpyAsync(h_stat, a.overflow_cnt, 56, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipStreamSynchronize(s));
#ifdef TSGPU_HNSW_PROF
        fprintf(stderr, "HNSW_PROF ticks(100MHz)/query: pop+barrier %.0f  links+tags %.0f  distances %.0f  heaps %.0f\n", (double)(h_stat[6] | ((uint64_t)h_stat[7] << 32)) / n_q,
                (double)(h_stat[8] | ((uint64_t)h_stat[9] << 32)) / n_q, (double)(h_stat[10] | ((uint64_t)h_stat[11] << 32)) / n_q, (double)(h_stat[12] | ((uint64_t)h_stat[13] << 32)) / n_q);
#endif
        tot_exp += (uint64_t)h_stat[2] | ((uint64_t)h_stat[3] << 32);
        tot_dist += (uint64_t)h_stat[4] | ((uint64_t)h_stat[5] << 32);
        if (!h_stat[0]) break;
        if (tier == TOP && !(hash_mode && vs_boost < 64)) break;      // (a candidate heap beyond the largest tier: those queries report n_out = 0xFFFFFFFF)
        // which queries: the ones marked 0xFFFFFFFF (of this launch's selection, or of the whole batch the first time)
        h_cnt.resize(n_q);
        TSGPU_HIP_TRY(hipMemcpy(h_cnt.data(), d_cnt, (size_t)n_q * 4, hipMemcpyDeviceToHost));
        std::vector<uint32_t> again;
        if (h_sel.empty()) { for (uint32_t q = 0; q < n_q; q++) if (h_cnt[q] == 0xFFFFFFFFu) again.push_back(q); }
        else for (uint32_t q : h_sel) if (h_cnt[q] == 0xFFFFFFFFu) again.push_back(q);
        if (again.empty()) break;
        h_sel.swap(again);
        if ((rc = f->g_sel.reserve(h_sel.size() * 4))) return rc;
        TSGPU_HIP_TRY(hipMemcpyAsync(f->g_sel.p, h_sel.data(), h_sel.size() * 4, hipMemcpyHostToDevice, s));
        TSGPU_HIP_TRY(hipStreamSynchronize(s));          // (h_sel is reused)
        a.q_sel = f->g_sel.as<uint32_t>();
        n_now = (uint32_t)h_sel.size();
        ctx->hnsw_tier_reruns += n_now;
        if (tier < TOP) tier = TOP; else vs_boost *= 8;
    }
    ctx->hnsw_last_expansions = tot_exp;
    ctx->hnsw_last_distances = tot_dist;
    return TSGPU_OK;
}

// BULK construction on the device (csrc/vec_hnsw_build.hip.h holds the algorithm and the kernels). ctx->mu is held for the whole build.
int tsgpu_vec_hnsw_build(tsgpu_ctx* ctx, uint32_t vec_field_id, uint32_t M, uint32_t ef_construction, uint32_t seed, uint32_t n_threads, uint32_t seed_min, uint32_t max_batch,
                         tsgpu_hnsw_build_info* info) {
    if (!ctx) return fail(TSGPU_ERR_INVALID, "ctx is NULL");
    if (M < 2 || M > 31) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_vec_hnsw_build: M must be 2..31 (a level-0 list of 2M ids is fetched by one wavefront)");
    ef_construction = std::max(ef_construction, M);
    if (ef_construction > VEC_HNSW_MAX_EF) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_vec_hnsw_build: ef_construction > 1024");
    if (seed_min == 0) seed_min = 1024;
    if (max_batch == 0) max_batch = 65536;
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    VecField* f = get_field(ctx, vec_field_id);
    if (!f) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_vec_hnsw_build: unknown vector field");
    if (f->hb) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_hnsw_build: this field inserts incrementally (tsgpu_vec_hnsw_enable)");
    hipStream_t s = ctx->stream;
    tsgpu_hnsw_build_info bi;
    memset(&bi, 0, sizeof bi);
    typedef std::chrono::steady_clock Clock;
    auto secs = [](Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    try {
        const uint32_t n = (uint32_t)f->n_rows, s0 = 1 + 2 * M, su = 1 + M, dim = f->dim;
        f->g_loaded = false;
        // 1. levels of ALL rows, drawn as hnswlib draws them (label order)
        std::vector<int32_t> levels(n);
        {
            std::default_random_engine gen;
            gen.seed(seed);
            const double mult = 1.0 / std::log(1.0 * M);
            for (uint32_t i = 0; i < n; i++) { std::uniform_real_distribution<double> u(0.0, 1.0); levels[i] = (int32_t)(-std::log(u(gen)) * mult); }
        }
        // 2. the seed set on the host: every row with an upper level, and the first seed_min rows
        const auto t0 = Clock::now();
        std::vector<uint32_t> seed_ids, rest_ids;
        int32_t seed_top = 0;
        for (uint32_t i = 0; i < n; i++) if (levels[i] >= 2 || i < seed_min) seed_top = std::max(seed_top, levels[i]);
        const int32_t host_from = seed_top >= 1 ? 2 : 1;          // (no seed row with an upper level: the level-1 rows are inserted on the host, too)
        std::vector<uint32_t> rest1_ids;                        // the level-1 rows among rest_ids (ascending like them)
        for (uint32_t i = 0; i < n; i++) {
            if (levels[i] >= host_from || i < seed_min) seed_ids.push_back(i);
            else { rest_ids.push_back(i); if (levels[i] == 1) rest1_ids.push_back(i); }
        }
        const uint32_t n_seed = (uint32_t)seed_ids.size(), n_rest = (uint32_t)rest_ids.size();
        std::vector<uint64_t> up((size_t)n + 1, 0);
        for (uint32_t i = 0; i < n; i++) up[i + 1] = up[i] + (uint64_t)levels[i];
        int rc;
        DevBuf d_rest1, d_cd1, d_ci1, d_cn1;
        DevBuf d_ids, d_tmp, d_rest, d_cd, d_ci, d_cn, d_req_s, d_req_d, d_cnt, d_fill, d_start, d_touched, d_counters, d_seg_c, d_seg_d;
        struct Scratch { std::vector<DevBuf*> b; ~Scratch() { for (auto* x : b) x->release(); } } scratch;           // (declared after the buffers: released first)
        scratch.b = {&d_rest1, &d_cd1, &d_ci1, &d_cn1, &d_ids, &d_tmp, &d_rest, &d_cd, &d_ci, &d_cn, &d_req_s, &d_req_d, &d_cnt, &d_fill, &d_start, &d_touched, &d_counters, &d_seg_c, &d_seg_d};
        HnswBuilder hs;
        hs.init(dim, M, ef_construction, seed, n_threads, (int)ctx->vec_ip_lanes);
        if ((rc = f->g_link0.reserve((size_t)std::max<uint32_t>(n, 1) * s0 * 4))) return rc;
        if ((rc = f->g_upper_ptr.reserve((size_t)(n + 1) * 8))) return rc;
        if ((rc = f->g_upper_links.reserve((size_t)std::max<uint64_t>(up[n], 1) * su * 4))) return rc;
        if ((rc = f->g_stat.reserve(64))) return rc;
        TSGPU_HIP_TRY(hipMemsetAsync(f->g_link0.p, 0, (size_t)std::max<uint32_t>(n, 1) * s0 * 4, s));
        if (n_seed) {
            if ((rc = d_ids.reserve((size_t)n_seed * 4))) return rc;
            if ((rc = d_tmp.reserve(std::max((size_t)n_seed * dim * 4, (size_t)n_seed * s0 * 4)))) return rc;
            TSGPU_HIP_TRY(hipMemcpyAsync(d_ids.p, seed_ids.data(), (size_t)n_seed * 4, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(vec_hnsw_build_gather_rows_kernel, dim3(4096), dim3(256), 0, s, f->X.as<float>(), d_ids.as<uint32_t>(), n_seed, dim, d_tmp.as<float>());
            {
                std::vector<float> rows((size_t)n_seed * dim);
                std::vector<int32_t> lv(n_seed);
                for (uint32_t i = 0; i < n_seed; i++) lv[i] = levels[seed_ids[i]];
                TSGPU_HIP_TRY(hipMemcpyAsync(rows.data(), d_tmp.p, rows.size() * 4, hipMemcpyDeviceToHost, s));
                TSGPU_HIP_TRY(hipStreamSynchronize(s));
                hs.add_batch(rows.data(), n_seed, lv.data());
            }
            TSGPU_HIP_TRY(hipMemcpyAsync(d_tmp.p, hs.link0.data(), (size_t)n_seed * s0 * 4, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(vec_hnsw_build_place_seed_kernel, dim3(4096), dim3(256), 0, s, d_tmp.as<uint32_t>(), d_ids.as<uint32_t>(), n_seed, s0, f->g_link0.as<uint32_t>());
            // the upper lists of the seed rows move to their places in the pool of ALL rows' upper lists (the device rows' level-1 lists start empty);
            // members map through seed_ids
            std::vector<uint32_t> pool((size_t)up[n] * su, 0u);
            for (uint32_t i = 0; i < n_seed; i++)
                for (int32_t l = 0; l < levels[seed_ids[i]]; l++) {
                    const uint32_t* src = hs.upper.data() + (hs.upper_at[i] + (uint64_t)l) * su;
                    uint32_t* dst = pool.data() + (up[seed_ids[i]] + (uint64_t)l) * su;
                    dst[0] = src[0];
                    for (uint32_t j = 0; j < src[0]; j++) dst[1 + j] = seed_ids[src[1 + j]];
                }
            if (up[n]) TSGPU_HIP_TRY(hipMemcpyAsync(f->g_upper_links.p, pool.data(), pool.size() * 4, hipMemcpyHostToDevice, s));
            TSGPU_HIP_TRY(hipStreamSynchronize(s));
        } else if (up[n]) TSGPU_HIP_TRY(hipMemsetAsync(f->g_upper_links.p, 0, (size_t)up[n] * su * 4, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(f->g_upper_ptr.p, up.data(), up.size() * 8, hipMemcpyHostToDevice, s));
        TSGPU_HIP_TRY(hipStreamSynchronize(s));
        f->g_M = M; f->g_n = n; f->g_slots = 4096; f->g_tag_slots = 0; f->g_epoch = 1;
        f->g_maxlevel = n_seed ? hs.maxlevel : -1;
        f->g_enterpoint = n_seed ? seed_ids[hs.enterpoint] : 0u;
        const auto t1 = Clock::now();
        bi.n = n; bi.n_seed = n_seed; bi.maxlevel = f->g_maxlevel; bi.enterpoint = f->g_enterpoint; bi.seed_seconds = secs(t0, t1);
        // 3. the level-0 rows, in batches against the graph as it stood before the batch
        if (n_rest) {
            const uint32_t B = std::min(max_batch, n_rest), k = ef_construction, n_rest1 = (uint32_t)rest1_ids.size();
            const size_t bm = (size_t)B * M;
            if ((rc = d_rest1.reserve((size_t)std::max<uint32_t>(n_rest1, 1) * 4)) || (rc = d_cd1.reserve((size_t)B * k * 4)) || (rc = d_ci1.reserve((size_t)B * k * 8)) || (rc = d_cn1.reserve((size_t)B * 4))) return rc;
            if (n_rest1) TSGPU_HIP_TRY(hipMemcpyAsync(d_rest1.p, rest1_ids.data(), (size_t)n_rest1 * 4, hipMemcpyHostToDevice, s));
            if ((rc = d_rest.reserve((size_t)n_rest * 4)) || (rc = d_cd.reserve((size_t)B * k * 4)) || (rc = d_ci.reserve((size_t)B * k * 8)) || (rc = d_cn.reserve((size_t)B * 4)) ||
                (rc = d_req_s.reserve(bm * 4)) || (rc = d_req_d.reserve(bm * 4)) || (rc = d_cnt.reserve((size_t)n * 4)) || (rc = d_fill.reserve((size_t)n * 4)) ||
                (rc = d_start.reserve((size_t)n * 4)) || (rc = d_touched.reserve(bm * 4)) || (rc = d_counters.reserve(64)) || (rc = d_seg_c.reserve(bm * 4)) || (rc = d_seg_d.reserve(bm * 4)))
                return rc;
            TSGPU_HIP_TRY(hipMemcpyAsync(d_rest.p, rest_ids.data(), (size_t)n_rest * 4, hipMemcpyHostToDevice, s));
            TSGPU_HIP_TRY(hipMemsetAsync(d_cnt.p, 0, (size_t)n * 4, s));
            TSGPU_HIP_TRY(hipMemsetAsync(d_fill.p, 0, (size_t)n * 4, s));
            TSGPU_HIP_TRY(hipMemsetAsync(d_counters.p, 0, 64, s));
            HnswBuildArgs a;
            memset(&a, 0, sizeof a);
            a.X = f->X.as<float>(); a.dim = dim; a.ip_lanes = ctx->vec_ip_lanes; a.link0 = f->g_link0.as<uint32_t>(); a.s0 = s0; a.M = M; a.k = k;
            a.upper_ptr = f->g_upper_ptr.as<uint64_t>(); a.upper_links = f->g_upper_links.as<uint32_t>(); a.su = su;
            a.req_s = d_req_s.as<uint32_t>(); a.req_d = d_req_d.as<float>(); a.cnt = d_cnt.as<uint32_t>(); a.fill = d_fill.as<uint32_t>(); a.start = d_start.as<uint32_t>();
            a.touched = d_touched.as<uint32_t>(); a.counters = d_counters.as<uint32_t>(); a.seg_c = d_seg_c.as<uint32_t>(); a.seg_d = d_seg_d.as<float>();
            f->g_loaded = true;                          // (the traversal below reads the device graph)
            uint32_t pos = 0, pos1 = 0, inserted = n_seed;
            // one round = the rows' lists on `layer` from their beams, the reverse-link requests, and every asked node's answer
            auto link_round = [&](uint32_t layer, const uint32_t* rows, uint32_t nb, DevBuf& ci, DevBuf& cd, DevBuf& cn) -> int {
                a.layer = layer; a.rows = rows; a.n_batch = nb;
                a.cand_id = ci.as<uint64_t>(); a.cand_d = cd.as<float>(); a.n_cand = cn.as<uint32_t>();
                TSGPU_HIP_TRY(hipMemsetAsync(d_counters.p, 0, 8, s));
                hipLaunchKernelGGL(vec_hnsw_build_select_kernel, dim3(std::min<uint32_t>(nb, 16384)), dim3(64), 0, s, a);
                hipLaunchKernelGGL(vec_hnsw_build_offsets_kernel, dim3(std::min<uint32_t>((nb * M + 255) / 256, 1024)), dim3(256), 0, s, a);
                hipLaunchKernelGGL(vec_hnsw_build_scatter_kernel, dim3(std::min<uint32_t>((nb * M + 255) / 256, 2048)), dim3(256), 0, s, a);
                hipLaunchKernelGGL(vec_hnsw_build_backlink_kernel, dim3(std::min<uint32_t>(nb * M, 16384)), dim3(64), 0, s, a);
                return TSGPU_OK;
            };
            while (pos < n_rest) {
                const uint32_t b = std::min(std::min(max_batch, std::max(64u, inserted / 16)), n_rest - pos);
                uint32_t b1 = 0;                         // the batch's rows with level 1: a run of rest1_ids
                while (pos1 + b1 < n_rest1 && rest1_ids[pos1 + b1] <= rest_ids[pos + b - 1]) b1++;
                const auto ta = Clock::now();
                // the beams first — layer 1 for the rows that have it, layer 0 for all —, on the graph as it stands; the links after both
                if (b1 && (rc = hnsw_search_launch(ctx, f, nullptr, d_rest1.as<uint32_t>() + pos1, b1, k, k, nullptr, true, nullptr, d_cd1.as<float>(), d_ci1.as<uint64_t>(), d_cn1.as<uint32_t>(), 1))) {
                    f->g_loaded = false;
                    return rc;
                }
                if ((rc = hnsw_search_launch(ctx, f, nullptr, d_rest.as<uint32_t>() + pos, b, k, k, nullptr, true, nullptr, d_cd.as<float>(), d_ci.as<uint64_t>(), d_cn.as<uint32_t>()))) {
                    f->g_loaded = false;
                    return rc;
                }
                const auto tb = Clock::now();
                if (b1 && (rc = link_round(1, d_rest1.as<uint32_t>() + pos1, b1, d_ci1, d_cd1, d_cn1))) return rc;
                if ((rc = link_round(0, d_rest.as<uint32_t>() + pos, b, d_ci, d_cd, d_cn))) return rc;
                TSGPU_HIP_TRY(hipStreamSynchronize(s));
                const auto tc = Clock::now();
                bi.search_seconds += secs(ta, tb); bi.link_seconds += secs(tb, tc);
                pos += b; pos1 += b1; inserted += b; bi.n_batches++;
            }
            uint32_t h_counters[4] = {0, 0, 0, 0};
            TSGPU_HIP_TRY(hipMemcpy(h_counters, d_counters.p, 16, hipMemcpyDeviceToHost));
            bi.unlinked = h_counters[2];
            TSGPU_HIP_TRY(hipGetLastError());
        }
        f->g_loaded = true;
        bi.device_seconds = secs(t1, Clock::now());
    } catch (const std::bad_alloc&) { f->g_loaded = false; return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_vec_hnsw_build: host allocation failed"); }
      catch (const std::system_error&) { f->g_loaded = false; return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_vec_hnsw_build: could not start an insertion thread"); }
    if (info) *info = bi;
    return ok();
}

int tsgpu_vec_hnsw_search_batch(tsgpu_ctx* ctx, uint32_t vec_field_id, const float* Q, int mem_q, uint32_t n_q, uint32_t k, uint32_t ef,
                                int functor_present, const uint32_t* allow_ids, uint32_t n_allow, const uint32_t* excluded_ids, uint32_t n_excluded,
                                float* dist_out, uint64_t* label_out, uint32_t* n_out, int mem_out) {
    if (!ctx || !Q || !dist_out || !label_out || !n_out) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_hnsw_search_batch: NULL argument");
    if (n_q == 0) return ok();
    if (k == 0) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_hnsw_search_batch: k must be > 0");
    if (std::max(k, ef) > VEC_HNSW_MAX_EF) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_vec_hnsw_search_batch: max(k, ef) > 1024 is not accelerated");
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    VecField* f = get_field(ctx, vec_field_id);
    if (!f) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_vec_hnsw_search_batch: unknown vector field");
    if (f->hb) {                                       // the graph is built inside the library: bring the device copy up to date
        if (f->hb->stale) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_vec_hnsw_search_batch: the built graph no longer matches the rows: use tsgpu_vec_knn_batch");
        if (f->hb->dirty || !f->g_loaded || f->g_n != f->n_rows) {
            HnswBuilder& hb = *f->hb;
            if (hb.size() != f->n_rows) return fail(TSGPU_ERR_DEVICE, "tsgpu_vec_hnsw_search_batch: the builder and the field disagree on the row count");
            std::vector<uint64_t> up(hb.size() + 1);
            for (size_t i = 0; i < hb.size(); i++) up[i] = hb.upper_at[i];
            up[hb.size()] = hb.n_upper_lists();
            const int urc = hnsw_upload_locked(ctx, f, hb.M, hb.maxlevel, hb.enterpoint == 0xFFFFFFFFu ? 0u : hb.enterpoint, hb.link0.data(), up.data(), hb.upper.data(), (uint32_t)hb.size());
            if (urc) return urc;
            hb.dirty = false;
        }
    }
    if (!f->g_loaded || f->g_n != f->n_rows) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_hnsw_search_batch: no graph loaded for the current rows (tsgpu_vec_hnsw_load / tsgpu_vec_hnsw_enable)");
    hipStream_t s = ctx->stream;
    try {
        const float* Q_dev = nullptr;
        int rc = stage_queries(ctx, f, Q, mem_q, n_q, &Q_dev);
        if (rc) return rc;
        const uint8_t* mask = nullptr;
        if ((rc = build_mask(ctx, f, allow_ids, n_allow, excluded_ids, n_excluded, &mask))) return rc;
        float* d_dist = dist_out; uint64_t* d_lab = label_out; uint32_t* d_cnt = n_out;
        if (mem_out != TSGPU_MEM_DEVICE) {
            if ((rc = f->d_dist.reserve((size_t)n_q * k * 4))) return rc;
            if ((rc = f->d_lab.reserve((size_t)n_q * k * 8))) return rc;
            if ((rc = f->d_cnt.reserve((size_t)n_q * 4))) return rc;
            d_dist = f->d_dist.as<float>(); d_lab = f->d_lab.as<uint64_t>(); d_cnt = f->d_cnt.as<uint32_t>();
        }
        if (f->n_rows == 0) {
            TSGPU_HIP_TRY(hipMemsetAsync(d_cnt, 0, (size_t)n_q * 4, s));
        } else {
            TSGPU_HIP_TRY(hipEventRecord(ctx->ev[3], s));
            if ((rc = hnsw_search_launch(ctx, f, Q_dev, nullptr, n_q, k, ef, mask, functor_present || f->any_deleted, f->labels.as<uint64_t>(), d_dist, d_lab, d_cnt))) return rc;
            TSGPU_HIP_TRY(hipEventRecord(ctx->ev[4], s));
            TSGPU_HIP_TRY(hipEventRecord(ctx->ev[5], s));
            TSGPU_HIP_TRY(hipGetLastError());
        }
        if (mem_out != TSGPU_MEM_DEVICE) {
            TSGPU_HIP_TRY(hipMemcpyAsync(dist_out, d_dist, (size_t)n_q * k * 4, hipMemcpyDeviceToHost, s));
            TSGPU_HIP_TRY(hipMemcpyAsync(label_out, d_lab, (size_t)n_q * k * 8, hipMemcpyDeviceToHost, s));
            TSGPU_HIP_TRY(hipMemcpyAsync(n_out, d_cnt, (size_t)n_q * 4, hipMemcpyDeviceToHost, s));
        }
        TSGPU_HIP_TRY(hipStreamSynchronize(s));
        if (f->n_rows) { ctx->scan_events_valid = false; knn_collect_timings(ctx); }
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_vec_hnsw_search_batch: host allocation failed"); }
    return ok();
}

int tsgpu_vec_distances(tsgpu_ctx* ctx, uint32_t vec_field_id, const float* q, const uint64_t* labels, uint32_t n, float* dist_out) {
    if (!ctx || !q || (n && (!labels || !dist_out))) return fail(TSGPU_ERR_INVALID, "tsgpu_vec_distances: NULL argument");
    if (n == 0) return ok();
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    VecField* f = get_field(ctx, vec_field_id);
    if (!f) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_vec_distances: unknown vector field");
    hipStream_t s = ctx->stream;
    try {
        std::vector<uint32_t> rows(n);
        for (uint32_t i = 0; i < n; i++) { uint32_t r; rows[i] = (f->find_row(labels[i], r) && f->h_ok[r]) ? r : 0xFFFFFFFFu; }
        int rc;
        if ((rc = f->d_rows.reserve((size_t)n * 4))) return rc;
        if ((rc = f->d_q1.reserve((size_t)f->dim * 4))) return rc;
        if ((rc = f->d_out1.reserve((size_t)n * 4))) return rc;
        TSGPU_HIP_TRY(hipMemcpyAsync(f->d_rows.p, rows.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(f->d_q1.p, q, (size_t)f->dim * 4, hipMemcpyHostToDevice, s));
        if (f->metric == TSGPU_METRIC_COSINE)   // the reference re-normalises q inside its loop (src/index.cpp:3362-3366)
            hipLaunchKernelGGL(vec_normalize_rows_kernel, dim3(1), dim3(64), 0, s, f->d_q1.as<float>(), 1u, f->dim);
        hipLaunchKernelGGL(vec_row_distances_kernel, dim3((n + 15) / 16), dim3(256), 0, s, f->X.as<float>(), f->d_q1.as<float>(), f->dim,
                           f->d_rows.as<uint32_t>(), n, f->d_out1.as<float>(), ctx->vec_ip_lanes);
        TSGPU_HIP_TRY(hipMemcpyAsync(dist_out, f->d_out1.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipStreamSynchronize(s));
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_vec_distances: host allocation failed"); }
    return ok();
}

// ONE pair on the host: hnswlib's space->get_dist_func()(a, b, &dim) (src/index.cpp:3365, :5842, :8868) in the summation order of the
// SIMD level hnswlib is compiled for (vec_kernels.hip.h "summation orders"; oracle: ip_simd16ext / ip_simd4ext). Plain host arithmetic on
// two vectors the caller already holds — the batched paths (k-NN, by-id distances, re-ranking) run on the device.
#pragma clang fp contract(off)
extern "C++" {
namespace {
template <int L> inline void ip_host_acc(float* lanes, const float* a, const float* b, size_t from, size_t to) {
    for (size_t i = from; i < to; i += L) for (int l = 0; l < L; l++) { const float p = a[i + l] * b[i + l]; lanes[l] = lanes[l] + p; }
}
inline float ip_host_16ext(const float* a, const float* b, size_t qty, int L) {
    float lanes[16] = {0};
    if (L == 16) ip_host_acc<16>(lanes, a, b, 0, qty); else if (L == 8) ip_host_acc<8>(lanes, a, b, 0, qty); else ip_host_acc<4>(lanes, a, b, 0, qty);
    if (L == 4) return lanes[0] + lanes[1] + lanes[2] + lanes[3];
    float sum = 0;
    for (int l = 0; l < L; l++) sum = sum + lanes[l];
    return sum;
}
inline float ip_host_4ext(const float* a, const float* b, size_t qty, int L) {
    float lanes[8] = {0};
    if (L == 4) { ip_host_acc<4>(lanes, a, b, 0, qty); return lanes[0] + lanes[1] + lanes[2] + lanes[3]; }
    const size_t q16 = qty / 16 * 16;
    ip_host_acc<8>(lanes, a, b, 0, q16);
    float s4[4];
    for (int l = 0; l < 4; l++) s4[l] = lanes[l] + lanes[l + 4];
    ip_host_acc<4>(s4, a, b, q16, qty);
    return s4[0] + s4[1] + s4[2] + s4[3];
}
inline float ip_host_scalar(const float* a, const float* b, size_t n) { float r = 0; for (size_t i = 0; i < n; i++) { const float p = a[i] * b[i]; r = r + p; } return r; }
}  // namespace
}  // extern "C++"
float tsgpu_ip_distance(const float* a, const float* b, uint32_t dim, int simd_lanes) {
    const int L = (simd_lanes == 8 || simd_lanes == 16) ? simd_lanes : 4;
    if (dim % 16 == 0) return 1.0f - ip_host_16ext(a, b, dim, L);
    if (dim % 4 == 0) return 1.0f - ip_host_4ext(a, b, dim, L);
    if (dim > 16) { const size_t q = dim >> 4 << 4; const float r = ip_host_16ext(a, b, q, L), t = ip_host_scalar(a + q, b + q, dim - q); return 1.0f - (r + t); }
    if (dim > 4) { const size_t q = dim >> 2 << 2; const float r = ip_host_4ext(a, b, q, L), t = ip_host_scalar(a + q, b + q, dim - q); return 1.0f - (r + t); }
    return 1.0f - ip_host_scalar(a, b, dim);
}
#pragma clang fp contract(fast)

// pure vector search, src/index.cpp:3645-3732
// FLAT branch (src/index.cpp:3664-3665 -> :3345-3374 -> :3675-3723): the distance matrix [n_q][n_filter] on the device, then the wildcard
// ranking machinery of the keyword module (kw_vector_flat_search) puts EVERY kept id through the Topster — no k cut.
static int vector_search_flat(tsgpu_ctx* ctx, uint32_t vec_field_id, const tsgpu_vec_query* p, const float* Q, int mem_q, uint32_t n_q, tsgpu_hits* out,
                              tsgpu_id_lists** ids_out) {
    const uint32_t nf = p->n_filter;
    const size_t stride = ((size_t)nf + 3) & ~(size_t)3;
    DevBuf d_dist, d_rows, d_q;                        // per call (another flat search may run while this one ranks): parked, not freed, on exit
    struct Park { DevBuf* b[3]; ~Park() { for (DevBuf* x : b) if (x->p) { deferred_frees().park(x->p, x->cap, false); x->p = nullptr; x->cap = 0; } } } park{{&d_dist, &d_rows, &d_q}};
    bool cosine = false;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        (void)hipSetDevice(ctx->device);
        VecField* f = get_field(ctx, vec_field_id);
        if (!f) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_vector_search_batch: unknown vector field");
        cosine = f->metric == TSGPU_METRIC_COSINE;
        hipStream_t s = ctx->stream;
        std::vector<uint32_t> rows(nf);
        for (uint32_t i = 0; i < nf; i++) {              // getDataByLabel throws for an unknown / deleted label: "likely not found", the id is skipped (:3355-3360)
            uint32_t r;
            rows[i] = (f->find_row(p->filter_ids[i], r) && f->h_ok[r]) ? r : 0xFFFFFFFFu;
            if (p->query_doc_given && p->filter_ids[i] == p->query_seq_id) rows[i] = 0xFFFFFFFFu;     // the query document itself is left out (:3686)
        }
        int rc;
        if ((rc = d_dist.reserve(std::max<size_t>((size_t)n_q * stride * 4, 16))) || (rc = d_rows.reserve((size_t)nf * 4)) || (rc = d_q.reserve((size_t)n_q * f->dim * 4))) return rc;
        TSGPU_HIP_TRY(hipMemcpyAsync(d_rows.p, rows.data(), (size_t)nf * 4, hipMemcpyHostToDevice, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(d_q.p, Q, (size_t)n_q * f->dim * 4, mem_q == TSGPU_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
        if (cosine) hipLaunchKernelGGL(vec_normalize_rows_kernel, dim3((n_q + 63) / 64), dim3(64), 0, s, d_q.as<float>(), n_q, f->dim);     // (:3362-3364)
        for (uint32_t q0 = 0; q0 < n_q; q0 += 32768) {
            const uint32_t nq = std::min<uint32_t>(32768, n_q - q0);
            hipLaunchKernelGGL(vec_flat_distances_kernel, dim3((nf + 15) / 16, nq), dim3(256), 0, s, f->X.as<float>(), d_q.as<float>() + (size_t)q0 * f->dim, f->dim,
                               d_rows.as<uint32_t>(), nf, d_dist.as<float>() + (size_t)q0 * stride, stride, ctx->vec_ip_lanes);
        }
        TSGPU_HIP_TRY(hipGetLastError());
        TSGPU_HIP_TRY(hipStreamSynchronize(s));         // the ranking runs on a keyword lane's stream
    }
    std::vector<tsgpu_kw_query> qs(n_q);
    memset(qs.data(), 0, qs.size() * sizeof(tsgpu_kw_query));
    const uint32_t tsz = p->topster_size ? p->topster_size : std::max<uint32_t>(1, std::min<uint32_t>(std::max<uint32_t>(p->fetch_size, TSGPU_DEFAULT_TOPSTER_SIZE), nf));   // :3506-3512
    for (uint32_t i = 0; i < n_q; i++) {
        tsgpu_kw_query& q = qs[i];
        q.n_sort = p->n_sort;
        for (uint32_t j = 0; j < p->n_sort; j++) q.sort[j] = p->sort[j];
        q.topster_size = tsz;
        q.filter_ids = p->filter_ids; q.n_filter = nf;
    }
    KwVFlat vf{d_dist.as<float>(), stride, p->distance_threshold, cosine};
    return kw_vector_flat_search(ctx, qs.data(), n_q, out, &vf, ids_out);
}

static int vector_search_impl(tsgpu_ctx* ctx, uint32_t vec_field_id, const tsgpu_vec_query* p, const float* Q, int mem_q, uint32_t n_q, tsgpu_hits* out,
                              tsgpu_id_lists** ids_out) {
    if (ids_out) *ids_out = nullptr;
    if (!ctx || !p || !Q || !out) return fail(TSGPU_ERR_INVALID, "tsgpu_vector_search_batch: NULL argument");
    if (n_q == 0) { if (ids_out) { *ids_out = new (std::nothrow) tsgpu_id_lists; if (*ids_out) (*ids_out)->begin.assign(1, 0); } return ok(); }
    if (p->n_sort > 3) return fail(TSGPU_ERR_INVALID, "tsgpu_vector_search_batch: more than 3 sort keys");
    if ((p->n_filter && !p->filter_ids) || (p->n_excluded && !p->excluded_ids)) return fail(TSGPU_ERR_INVALID, "tsgpu_vector_search_batch: id array is NULL");
    const bool filter_by_provided = p->filter_by_provided || p->n_filter;
    if (filter_by_provided && p->n_filter != 0 && (uint64_t)p->n_filter < p->flat_search_cutoff) {     // :3664 (an empty filter never gets here: the search returns before)
        if (!out->keys || !out->scores || !out->n_hits || !out->status) return fail(TSGPU_ERR_INVALID, "tsgpu_vector_search_batch: missing output arrays");
        try { return vector_search_flat(ctx, vec_field_id, p, Q, mem_q, n_q, out, ids_out); }
        catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_vector_search_batch: host allocation failed"); }
    }
    if (out->mem != TSGPU_MEM_HOST) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_vector_search_batch: the k-cut branch delivers host outputs only");
    VecField* f;
    uint32_t num_docs;
    { std::lock_guard<std::mutex> lk(ctx->mu); f = get_field(ctx, vec_field_id); num_docs = ctx->num_docs; }
    if (!f) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_vector_search_batch: unknown vector field");
    uint32_t k = p->k == 0 ? std::max<uint32_t>(p->k, p->fetch_size) : p->k;   // :3646
    if (k == 0) return fail(TSGPU_ERR_INVALID, "tsgpu_vector_search_batch: k and fetch_size are both 0");
    try {
        // VectorFilterFunctor (include/index.h:339-353): excluded ids reject, then the filter ids (when there are any) admit
        auto functor = [&](uint32_t id) {
            if (p->n_filter == 0 && p->n_excluded == 0) return true;
            if (p->n_excluded && std::binary_search(p->excluded_ids, p->excluded_ids + p->n_excluded, id)) return false;
            if (p->n_filter == 0) return true;
            return std::binary_search(p->filter_ids, p->filter_ids + p->n_filter, id);
        };
        if (p->query_doc_given && functor(p->query_seq_id)) k++;                // :3651-3654
        if (k > TSGPU_MAX_TOPK) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_vector_search_batch: k > TSGPU_MAX_TOPK is not accelerated");
        KnnHost kh;
        int rc = knn_to_host(ctx, vec_field_id, Q, mem_q, n_q, k, p->n_filter ? p->filter_ids : nullptr, p->n_filter, p->n_excluded ? p->excluded_ids : nullptr, p->n_excluded, kh);
        if (rc) return rc;
        uint32_t tsz = p->topster_size ? p->topster_size : std::max<uint32_t>(p->fetch_size, TSGPU_DEFAULT_TOPSTER_SIZE);
        if (!p->topster_size) tsz = std::max<uint32_t>(1, std::min<uint32_t>(tsz, p->n_filter ? p->n_filter : std::max<uint32_t>(num_docs, (uint32_t)f->n_rows)));   // :3506-3512
        if (out->k_stride < std::min<uint32_t>(tsz, k)) return fail(TSGPU_ERR_INVALID, "tsgpu_vector_search_batch: k_stride too small");
        std::unique_ptr<tsgpu_id_lists> lists;
        if (ids_out) { lists.reset(new tsgpu_id_lists); lists->begin.assign(1, 0); }
        std::vector<std::pair<uint64_t, float>> hits;
        for (uint32_t q = 0; q < n_q; q++) {
            HostTopster topster(tsz);
            hits.clear();
            for (uint32_t i = 0; i < kh.cnt[q]; i++) hits.emplace_back(kh.lab[(size_t)q * k + i], kh.dist[(size_t)q * k + i]);
            std::sort(hits.begin(), hits.end(), [](const auto& a, const auto& b) { return a.first < b.first; });   // :3389
            uint64_t added = 0;
            for (auto& h : hits) {
                if (p->query_doc_given && h.first == p->query_seq_id) continue;                        // :3686
                const float d = f->metric == TSGPU_METRIC_COSINE ? std::fabs(h.second) : h.second;   // :3699
                if (d > p->distance_threshold) continue;                                               // :3702
                HostKV kv;
                int64_t msi = -1;
                host_sort_scores(ctx, p->sort, p->n_sort, h.first, 0, d, kv.scores, msi);
                kv.match_score_index = (int8_t)msi;
                kv.key = h.first;
                kv.vector_distance = d;
                if (msi >= 0) kv.text_match_score = kv.scores[msi];
                topster.add(&kv);
                added++;
                if (lists) lists->ids.push_back((uint32_t)h.first);                                     // nearest_ids (ascending: label order), :3727-3732
            }
            topster.sort();
            write_hits(topster, q, out);
            if (out->num_matched) out->num_matched[q] = added;
            out->status[q] = TSGPU_OK;
            if (out->search_cutoff) out->search_cutoff[q] = 0;
            if (lists) lists->begin.push_back(lists->ids.size());
        }
        if (ids_out) *ids_out = lists.release();
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_vector_search_batch: host allocation failed"); }
    return ok();
}

int tsgpu_vector_search_batch(tsgpu_ctx* ctx, uint32_t vec_field_id, const tsgpu_vec_query* p, const float* Q, int mem_q, uint32_t n_q, tsgpu_hits* out) {
    return vector_search_impl(ctx, vec_field_id, p, Q, mem_q, n_q, out, nullptr);
}
int tsgpu_vector_search_batch_ids(tsgpu_ctx* ctx, uint32_t vec_field_id, const tsgpu_vec_query* p, const float* Q, int mem_q, uint32_t n_q, tsgpu_hits* out,
                                  tsgpu_id_lists** ids_out) {
    if (!ids_out) return fail(TSGPU_ERR_INVALID, "tsgpu_vector_search_batch_ids: ids_out is NULL");
    return vector_search_impl(ctx, vec_field_id, p, Q, mem_q, n_q, out, ids_out);
}

// reciprocal rank fusion of ONE query, exactly src/index.cpp:4094-4211: `kw` = the keyword Topster content in sort()
// order (slots of query q), knn = this query's exact nearest neighbours. Writes the fused + sorted Topster into out.
static void fuse_one_query(tsgpu_ctx* ctx, int metric, const tsgpu_kw_query& kq, const tsgpu_hybrid_params* p, const tsgpu_hits& kw, uint32_t q,
                           const float* knn_dist, const uint64_t* knn_lab, uint32_t knn_n, tsgpu_hits* out) {
    const float VECTOR_SEARCH_WEIGHT = p->alpha;
    const float TEXT_MATCH_WEIGHT = 1.0 - VECTOR_SEARCH_WEIGHT;
    struct VH { float dist; uint64_t seq_id; };
    uint32_t tsz = kq.topster_size ? kq.topster_size : TSGPU_DEFAULT_TOPSTER_SIZE;                      // src/index.cpp:3506-3512
    tsz = std::max<uint32_t>(1, std::min<uint32_t>(tsz, (!kq.topster_size && kq.n_filter) ? kq.n_filter : std::max<uint32_t>(ctx->num_docs, 1)));
    HostTopster topster(tsz);
    const uint32_t nh = kw.n_hits[q];
    std::vector<HostKV> sorted(nh);
    for (uint32_t i = 0; i < nh; i++) {
        const size_t s = (size_t)q * kw.k_stride + i;
        HostKV& kv = sorted[i];
        kv.key = kw.keys[s];
        for (int j = 0; j < 3; j++) kv.scores[j] = kw.scores[s * 3 + j];
        kv.match_score_index = kw.match_score_index ? kw.match_score_index[s] : 0;
        kv.text_match_score = kw.text_match ? kw.text_match[s] : 0;
        kv.vector_distance = -1.0f;
    }
    topster.adopt_sorted(sorted.data(), nh);
    // vector hits: label order, threshold, then distance order (src/index.cpp:3389, 3414-3437)
    std::vector<VH> dist_results;
    for (uint32_t i = 0; i < knn_n; i++) dist_results.push_back({knn_dist[i], knn_lab[i]});
    std::sort(dist_results.begin(), dist_results.end(), [](const VH& a, const VH& b) { return a.seq_id < b.seq_id; });
    {
        std::vector<VH> kept;
        for (auto& r : dist_results) {
            const float sc = metric == TSGPU_METRIC_COSINE ? std::fabs(r.dist) : r.dist;
            if (sc > p->distance_threshold) continue;
            kept.push_back(r);
        }
        std::stable_sort(kept.begin(), kept.end(), [](const VH& a, const VH& b) { return a.dist < b.dist; });
        dist_results.swap(kept);
    }
    std::unordered_map<uint64_t, uint32_t> seq_id_to_rank;
    for (size_t i = 0; i < dist_results.size(); i++) seq_id_to_rank.emplace(dist_results[i].seq_id, (uint32_t)i);
    std::sort(dist_results.begin(), dist_results.end(), [](const VH& a, const VH& b) { return a.seq_id < b.seq_id; });
    // text ranks (dense rank on score ties), :4094-4112
    int64_t text_rank = 0, last_text_match_score = INT64_MAX;
    for (uint32_t i = 0; i < topster.size; i++) {
        HostKV* r = topster.kvs[i];
        if (r->match_score_index < 0 || r->match_score_index > 2) continue;
        r->text_match_score = r->scores[r->match_score_index];
        if (r->text_match_score < last_text_match_score) ++text_rank;
        last_text_match_score = r->text_match_score;
        r->scores[r->match_score_index] = float_to_int64((1.0 / (text_rank)) * TEXT_MATCH_WEIGHT);
    }
    for (auto& dr : dist_results) {   // :4124-4211
        const uint64_t seq_id = dr.seq_id;
        auto it = topster.map.find(seq_id);
        HostKV* found_kv = it == topster.map.end() ? nullptr : it->second;
        if (found_kv) {
            if (found_kv->match_score_index < 0 || found_kv->match_score_index > 2) continue;
            found_kv->vector_distance = dr.dist;
            const int64_t match_score = float_to_int64((int64_to_float(found_kv->scores[found_kv->match_score_index])) +
                                                       ((1.0 / (seq_id_to_rank[seq_id] + 1)) * VECTOR_SEARCH_WEIGHT));
            int64_t match_score_index = -1;
            int64_t sc[3] = {0, 0, 0};
            host_sort_scores(ctx, kq.sort, kq.n_sort, seq_id, match_score, dr.dist, sc, match_score_index);
            for (int j = 0; j < 3; j++) found_kv->scores[j] = sc[j];
            found_kv->match_score_index = (int8_t)match_score_index;
        } else {
            HostKV kv;
            const int64_t match_score = float_to_int64((1.0 / (seq_id_to_rank[seq_id] + 1)) * VECTOR_SEARCH_WEIGHT);
            int64_t match_score_index = -1;
            host_sort_scores(ctx, kq.sort, kq.n_sort, seq_id, match_score, dr.dist, kv.scores, match_score_index);
            kv.match_score_index = (int8_t)match_score_index;
            kv.key = seq_id;
            kv.text_match_score = 0;
            kv.vector_distance = dr.dist;
            topster.add(&kv);
        }
    }
    topster.sort();
    write_hits(topster, q, out);
}

// RRF fusion of already-computed (e.g. shard-merged) keyword hits and k-NN results; host arrays
int tsgpu_hybrid_fuse_batch(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, const tsgpu_hybrid_params* p, int metric, const tsgpu_hits* kw_hits,
                            const float* knn_dist, const uint64_t* knn_labels, const uint32_t* knn_cnt, uint32_t knn_k, uint32_t n_queries,
                            tsgpu_hits* out) {
    if (!ctx || !queries || !p || !kw_hits || !knn_dist || !knn_labels || !knn_cnt || !out) return fail(TSGPU_ERR_INVALID, "tsgpu_hybrid_fuse_batch: NULL argument");
    if (kw_hits->mem != TSGPU_MEM_HOST || out->mem != TSGPU_MEM_HOST) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_hybrid_fuse_batch: host arrays only");
    // the fusion of one query touches only that query's slots: the batch is spread over host threads (the reference fuses inside
    // each request thread; here one call carries a whole batch)
    std::atomic<uint32_t> next{0};
    std::atomic<int> oom{0};
    auto worker = [&]() {
        try {
            for (;;) {
                const uint32_t q = next.fetch_add(1);
                if (q >= n_queries) break;
                const int32_t st = kw_hits->status ? kw_hits->status[q] : TSGPU_OK;
                out->status[q] = st;
                if (out->search_cutoff) out->search_cutoff[q] = kw_hits->search_cutoff ? kw_hits->search_cutoff[q] : 0;
                if (st != TSGPU_OK) { out->n_hits[q] = 0; if (out->num_matched) out->num_matched[q] = 0; continue; }
                fuse_one_query(ctx, metric, queries[q], p, *kw_hits, q, knn_dist + (size_t)q * knn_k, knn_labels + (size_t)q * knn_k, knn_cnt[q], out);
                if (out->num_matched) out->num_matched[q] = kw_hits->num_matched ? kw_hits->num_matched[q] : 0;
            }
        } catch (const std::bad_alloc&) { oom = 1; }
    };
    const uint32_t hw = std::max(1u, std::thread::hardware_concurrency());
    const uint32_t n_threads = std::min<uint32_t>(std::min<uint32_t>(hw, (uint32_t)ctx->fuse_threads), std::max<uint32_t>(1, n_queries / 8));
    if (n_threads <= 1) worker();
    else ctx->host_pool.run(worker, (int)n_threads - 1);          // parked pool threads + this one
    if (oom) return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_hybrid_fuse_batch: host allocation failed");
    return ok();
}

// Index::compute_aux_scores (src/index.cpp:8793-8923; rerank_hybrid_matches) on the fused hits of a batch (`out` = every query's Topster
// content): hits only the vector search found get their text_match score (tsgpu_keyword_aux_scores), hits only the keyword search
// found their exact distance by label; then keyword ranks by (text_match_score, key) descending, semantic ranks by distance
// (stable, on the keyword order), fused score = 1/keyword_rank * (1 - alpha) + 1/semantic_rank * alpha into scores[match_score_index],
// and the Topster order again.
extern "C" int tsgpu_keyword_aux_scores(tsgpu_ctx*, const tsgpu_kw_query*, uint32_t, const uint32_t*, const uint32_t*, uint32_t, int64_t*);
}  // extern "C"
namespace tsgpu {
// exact distances of (query, label) pairs BY LABEL (getDataByLabel + get_dist_func, cosine: the query normalised first), ONE launch for the batch;
// d_out[i] = NaN when the label has no live row in THIS context (a shard that does not own the document; the reference's getDataByLabel throws)
int hybrid_missing_distances(tsgpu_ctx* ctx, uint32_t vec_field_id, const float* Q, int mem_q, uint32_t n_queries, const uint32_t* pair_q, const uint64_t* pair_label, uint32_t n, float* d_out) {
    if (n == 0) return TSGPU_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    VecField* f = get_field(ctx, vec_field_id);
    if (!f) return fail(TSGPU_ERR_NOT_FOUND, "rerank_hybrid_matches: unknown vector field");
    const uint32_t dim = f->dim;
    std::vector<uint32_t> rows, qs, at;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t r;
        d_out[i] = std::numeric_limits<float>::quiet_NaN();
        if (f->find_row(pair_label[i], r) && f->h_ok[r]) { rows.push_back(r); qs.push_back(pair_q[i]); at.push_back(i); }
    }
    const uint32_t m = (uint32_t)rows.size();
    if (m == 0) return TSGPU_OK;
    hipStream_t s = ctx->stream;
    int rc;
    if ((rc = f->d_rows.reserve((size_t)m * 8))) return rc;
    if ((rc = f->d_out1.reserve((size_t)m * 4))) return rc;
    if ((rc = f->d_q1.reserve((size_t)n_queries * dim * 4))) return rc;
    TSGPU_HIP_TRY(hipMemcpyAsync(f->d_rows.p, rows.data(), (size_t)m * 4, hipMemcpyHostToDevice, s));
    TSGPU_HIP_TRY(hipMemcpyAsync((char*)f->d_rows.p + (size_t)m * 4, qs.data(), (size_t)m * 4, hipMemcpyHostToDevice, s));
    TSGPU_HIP_TRY(hipMemcpyAsync(f->d_q1.p, Q, (size_t)n_queries * dim * 4, mem_q == TSGPU_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    if (f->metric == TSGPU_METRIC_COSINE)
        hipLaunchKernelGGL(vec_normalize_rows_kernel, dim3((n_queries + 3) / 4), dim3(256), 0, s, f->d_q1.as<float>(), n_queries, dim);
    hipLaunchKernelGGL(vec_pair_distances_kernel, dim3((m + 15) / 16), dim3(256), 0, s, f->X.as<float>(), f->d_q1.as<float>(), dim,
                       (const uint32_t*)((char*)f->d_rows.p + (size_t)m * 4), f->d_rows.as<uint32_t>(), m, f->d_out1.as<float>(), ctx->vec_ip_lanes);
    std::vector<float> d(m);
    TSGPU_HIP_TRY(hipMemcpyAsync(d.data(), f->d_out1.p, (size_t)m * 4, hipMemcpyDeviceToHost, s));
    TSGPU_HIP_TRY(hipStreamSynchronize(s));
    for (uint32_t i = 0; i < m; i++) d_out[at[i]] = d[i];
    return TSGPU_OK;
}

// which fused hits miss one side's score: item_* = found via vector distance only (text_match == 0), pair_* = via text match only (vector_distance == -1)
void hybrid_missing_items(uint32_t n_queries, const tsgpu_hits* out, std::vector<uint32_t>& item_q, std::vector<uint32_t>& item_id, std::vector<size_t>& item_slot,
                          std::vector<uint32_t>& pair_q, std::vector<uint64_t>& pair_label, std::vector<size_t>& pair_slot) {
    for (uint32_t q = 0; q < n_queries; q++) {
        if (out->status[q] != TSGPU_OK) continue;
        for (uint32_t i = 0; i < out->n_hits[q]; i++) {
            const size_t s = (size_t)q * out->k_stride + i;
            if (out->text_match[s] == 0) { item_q.push_back(q); item_id.push_back((uint32_t)out->keys[s]); item_slot.push_back(s); }
            else if (out->vector_distance[s] == -1.0f) { pair_q.push_back(q); pair_label.push_back(out->keys[s]); pair_slot.push_back(s); }
        }
    }
}

// compute_aux_scores' last step: keyword ranks by (text_match_score, key) descending, semantic ranks by distance (stable, on the keyword order), fused score =
// 1/keyword_rank * (1 - alpha) + 1/semantic_rank * alpha into scores[match_score_index], and the Topster order again
void hybrid_refuse(const tsgpu_hybrid_params* p, uint32_t n_queries, tsgpu_hits* out) {
    struct E { uint64_t key; int64_t sc[3]; int64_t tm; float vd; int8_t msi; uint32_t krank, srank; };
    std::vector<E> es;
    std::vector<uint32_t> order;
    for (uint32_t q = 0; q < n_queries; q++) {
        if (out->status[q] != TSGPU_OK) continue;
        const uint32_t n = out->n_hits[q];
        es.resize(n);
        for (uint32_t i = 0; i < n; i++) {
            const size_t s = (size_t)q * out->k_stride + i;
            E& e = es[i];
            e.key = out->keys[s]; for (int j = 0; j < 3; j++) e.sc[j] = out->scores[s * 3 + j];
            e.tm = out->text_match[s]; e.vd = out->vector_distance[s]; e.msi = out->match_score_index[s];
        }
        order.resize(n);
        for (uint32_t i = 0; i < n; i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return std::tie(es[a].tm, es[a].key) > std::tie(es[b].tm, es[b].key); });
        for (uint32_t i = 0; i < n; i++) es[order[i]].krank = i + 1;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return es[a].vd < es[b].vd; });
        for (uint32_t i = 0; i < n; i++) es[order[i]].srank = i + 1;
        for (uint32_t i = 0; i < n; i++) {
            E& e = es[i];
            if (e.msi < 0 || e.msi > 2) continue;
            e.sc[e.msi] = float_to_int64((1.0 / e.krank) * (1.0 - p->alpha) + (1.0 / e.srank) * p->alpha);
        }
        for (uint32_t i = 0; i < n; i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            const E& x = es[a]; const E& y = es[b];
            return std::tie(x.sc[0], x.sc[1], x.sc[2], x.key) > std::tie(y.sc[0], y.sc[1], y.sc[2], y.key);       // KV::is_greater, include/topster.h:146-149
        });
        for (uint32_t i = 0; i < n; i++) {
            const size_t s = (size_t)q * out->k_stride + i;
            const E& e = es[order[i]];
            out->keys[s] = e.key; for (int j = 0; j < 3; j++) out->scores[s * 3 + j] = e.sc[j];
            out->text_match[s] = e.tm; out->vector_distance[s] = e.vd; out->match_score_index[s] = e.msi;
        }
    }
}
}  // namespace tsgpu
extern "C" {

static int hybrid_rerank(tsgpu_ctx* ctx, VecField* /*f*/, uint32_t vec_field_id, const tsgpu_kw_query* queries, const tsgpu_hybrid_params* p, const float* Q, int mem_q,
                         uint32_t n_queries, tsgpu_hits* out) {
    if (!out->text_match || !out->vector_distance || !out->match_score_index) return fail(TSGPU_ERR_INVALID, "rerank_hybrid_matches needs the text_match, vector_distance and match_score_index outputs");
    // 1) what is missing
    std::vector<uint32_t> item_q, item_id, pair_q;
    std::vector<uint64_t> pair_label;
    std::vector<size_t> item_slot, pair_slot;
    tsgpu::hybrid_missing_items(n_queries, out, item_q, item_id, item_slot, pair_q, pair_label, pair_slot);
    if (!pair_q.empty()) {
        std::vector<float> d(pair_q.size());
        const int rc = tsgpu::hybrid_missing_distances(ctx, vec_field_id, Q, mem_q, n_queries, pair_q.data(), pair_label.data(), (uint32_t)pair_q.size(), d.data());
        if (rc) return rc;
        for (size_t i = 0; i < d.size(); i++) if (d[i] == d[i]) out->vector_distance[pair_slot[i]] = d[i];      // (a label without a vector: getDataByLabel throws, the hit is left as it is)
    }
    // 2) text_match of the vector-only hits; the reference walks them in ascending key order per query (iterators only move forward):
    //    the score of a document does not depend on that order
    if (!item_q.empty()) {
        std::vector<int64_t> sc(item_q.size());
        const int rc = tsgpu_keyword_aux_scores(ctx, queries, n_queries, item_q.data(), item_id.data(), (uint32_t)item_q.size(), sc.data());
        if (rc) return rc;
        for (size_t i = 0; i < sc.size(); i++) out->text_match[item_slot[i]] = sc[i];
    }
    // 3) re-rank and re-fuse, query by query
    tsgpu::hybrid_refuse(p, n_queries, out);
    return TSGPU_OK;
}

// hybrid, src/index.cpp:4036-4221: keyword pass -> exact k-NN -> reciprocal rank fusion on the sorted Topster
int tsgpu_hybrid_search_batch(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t vec_field_id, const tsgpu_hybrid_params* p,
                              const float* Q, int mem_q, uint32_t n_queries, tsgpu_hits* out) {
    if (!ctx || !queries || !p || !Q || !out) return fail(TSGPU_ERR_INVALID, "tsgpu_hybrid_search_batch: NULL argument");
    if (out->mem != TSGPU_MEM_HOST) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_hybrid_search_batch: host outputs only");
    if (n_queries == 0) return ok();
    VecField* f;
    { std::lock_guard<std::mutex> lk(ctx->mu); f = get_field(ctx, vec_field_id); }
    if (!f) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_hybrid_search_batch: unknown vector field");
    const uint32_t k = p->k == 0 ? std::max<uint32_t>(p->fetch_size, 100) : p->k;   // :4060-4063
    try {
        // 1) keyword pass on the GPU -> Topster contents in sort() order
        uint32_t KS = out->k_stride;
        std::vector<uint64_t> keys((size_t)n_queries * KS);
        std::vector<int64_t> scores((size_t)n_queries * KS * 3), tm((size_t)n_queries * KS);
        std::vector<float> vd((size_t)n_queries * KS);
        std::vector<int8_t> msi((size_t)n_queries * KS);
        std::vector<uint32_t> nh(n_queries);
        std::vector<uint64_t> nm(n_queries);
        std::vector<int32_t> st(n_queries), co(n_queries);
        tsgpu_hits kw;
        kw.mem = TSGPU_MEM_HOST; kw.k_stride = KS;
        kw.keys = keys.data(); kw.scores = scores.data(); kw.text_match = tm.data(); kw.vector_distance = vd.data();
        kw.match_score_index = msi.data(); kw.n_hits = nh.data(); kw.num_matched = nm.data(); kw.status = st.data(); kw.search_cutoff = co.data();
        static const bool host_timing = getenv("TSGPU_HOST_TIMING") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        // The keyword pass runs on a second host thread and on a keyword lane of its own (not lane 0: that one shares the vector stream)
        // WHILE the vector pass runs here: its kernels fill the chip during the vector pass's thin phases (query cast, sample pass,
        // threshold select, refine / re-score / select: ~1.2 ms of a 6 ms pass). Neither pass reads the other's results.
        int rc_kw = TSGPU_OK;
        std::string err_kw;
        std::thread kw_thread;
        const bool overlap = ctx->hybrid_overlap && ctx->n_lanes >= 2;
        auto kw_pass = [&]() {
            tls_avoid_lane0() = true;
            rc_kw = tsgpu_keyword_search_batch(ctx, queries, n_queries, &kw);
            tls_avoid_lane0() = false;
            if (rc_kw != TSGPU_OK) err_kw = tls_error();
        };
        struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{kw_thread};      // (every exit path waits for the pass)
        if (overlap) { try { kw_thread = std::thread(kw_pass); } catch (const std::system_error&) { kw_pass(); } }
        else kw_pass();
        if (!overlap && rc_kw) return fail(rc_kw, err_kw);
        const auto t1 = std::chrono::steady_clock::now();
        // 2) vector pass: one batched exact k-NN for the whole batch; a query with filter_by / excluded ids (the VectorFilterFunctor of
        //    process_results_hnsw_index, src/index.cpp:3376-3445) gets its own exact k-NN restricted to its allowed ids afterwards
        KnnHost kh;
        int rc = knn_to_host(ctx, vec_field_id, Q, mem_q, n_queries, k, nullptr, 0, nullptr, 0, kh);
        if (kw_thread.joinable()) kw_thread.join();
        if (rc_kw) return fail(rc_kw, err_kw);
        if (rc) return rc;
        {
            KnnHost one;
            for (uint32_t q = 0; q < n_queries; q++) {
                if (st[q] != TSGPU_OK || (queries[q].n_excluded == 0 && queries[q].n_filter == 0)) continue;
                if ((rc = knn_to_host(ctx, vec_field_id, Q + (size_t)q * f->dim, mem_q, 1, k, queries[q].n_filter ? queries[q].filter_ids : nullptr, queries[q].n_filter,
                                      queries[q].n_excluded ? queries[q].excluded_ids : nullptr, queries[q].n_excluded, one))) return rc;
                std::copy(one.dist.begin(), one.dist.end(), kh.dist.begin() + (size_t)q * k);
                std::copy(one.lab.begin(), one.lab.end(), kh.lab.begin() + (size_t)q * k);
                kh.cnt[q] = one.cnt[0];
            }
        }
        const auto t2 = std::chrono::steady_clock::now();
        // 3) fusion on the host, exactly as the reference
        rc = tsgpu_hybrid_fuse_batch(ctx, queries, p, f->metric, &kw, kh.dist.data(), kh.lab.data(), kh.cnt.data(), k, n_queries, out);
        if (!rc && p->rerank_hybrid_matches) rc = hybrid_rerank(ctx, f, vec_field_id, queries, p, Q, mem_q, n_queries, out);
        if (host_timing) {
            auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return (long long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
            fprintf(stderr, "[tsgpu] hybrid batch %u queries: keyword pass %lld us, k-NN to host %lld us, fusion %lld us\n", n_queries, us(t0, t1), us(t1, t2), us(t2, std::chrono::steady_clock::now()));
        }
        return rc;
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_hybrid_search_batch: host allocation failed"); }
}

// exact merge of per-shard Topster lists (doc-range shards): same comparator as include/topster.h:146-149
int tsgpu_merge_shard_hits(const tsgpu_hits* in, const uint64_t* key_offset, uint32_t n_shards, uint32_t n_queries, uint32_t k, tsgpu_hits* out) {
    if (!in || !out || n_shards == 0) return fail(TSGPU_ERR_INVALID, "tsgpu_merge_shard_hits: NULL argument");
    if (out->k_stride < k) return fail(TSGPU_ERR_INVALID, "tsgpu_merge_shard_hits: k_stride < k");
    for (uint32_t g = 0; g < n_shards; g++) if (in[g].mem != TSGPU_MEM_HOST) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_merge_shard_hits: host inputs only");
    try {
        std::vector<HostKV> all;
        for (uint32_t q = 0; q < n_queries; q++) {
            all.clear();
            uint64_t nm = 0;
            int32_t st = TSGPU_OK, co = 0;
            for (uint32_t g = 0; g < n_shards; g++) {
                const tsgpu_hits& h = in[g];
                if (h.status && h.status[q] != TSGPU_OK) st = h.status[q];
                if (h.search_cutoff && h.search_cutoff[q]) co = 1;
                if (h.num_matched) nm += h.num_matched[q];
                for (uint32_t i = 0; i < h.n_hits[q]; i++) {
                    const size_t s = (size_t)q * h.k_stride + i;
                    HostKV kv;
                    kv.key = h.keys[s] + (key_offset ? key_offset[g] : 0);
                    for (int j = 0; j < 3; j++) kv.scores[j] = h.scores[s * 3 + j];
                    kv.text_match_score = h.text_match ? h.text_match[s] : 0;
                    kv.vector_distance = h.vector_distance ? h.vector_distance[s] : -1.0f;
                    kv.match_score_index = h.match_score_index ? h.match_score_index[s] : 0;
                    all.push_back(kv);
                }
            }
            std::sort(all.begin(), all.end(), [](const HostKV& a, const HostKV& b) { return kv_greater(&a, &b); });
            const uint32_t n = (uint32_t)std::min<size_t>(all.size(), k);
            const size_t base = (size_t)q * out->k_stride;
            for (uint32_t i = 0; i < n; i++) {
                out->keys[base + i] = all[i].key;
                for (int j = 0; j < 3; j++) out->scores[(base + i) * 3 + j] = all[i].scores[j];
                if (out->text_match) out->text_match[base + i] = all[i].text_match_score;
                if (out->vector_distance) out->vector_distance[base + i] = all[i].vector_distance;
                if (out->match_score_index) out->match_score_index[base + i] = all[i].match_score_index;
            }
            out->n_hits[q] = n;
            if (out->num_matched) out->num_matched[q] = nm;
            if (out->status) out->status[q] = st;
            if (out->search_cutoff) out->search_cutoff[q] = co;
        }
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_merge_shard_hits: host allocation failed"); }
    return ok();
}

}  // extern "C"

// ---- tsgpu_group (tsgpu_group.hip): the device-side halves of the k-NN exchange ----
namespace tsgpu {
int group_vec_dim(tsgpu_ctx* ctx, uint32_t vec_field_id, uint32_t* dim) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    VecField* f = get_field(ctx, vec_field_id);
    if (!f) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_group: unknown vector field");
    *dim = f->dim;
    return TSGPU_OK;
}
int group_pack_knn(tsgpu_ctx* ctx, const float* dist_dev, const uint64_t* label_dev, const uint32_t* cnt_dev, uint32_t n_q, uint32_t k, uint64_t* block, uint32_t* bad_dev, hipStream_t s) {
    (void)hipSetDevice(ctx->device);
    const uint64_t n = (uint64_t)n_q * k;
    hipLaunchKernelGGL(vec_group_pack_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, dist_dev, label_dev, cnt_dev, n_q, k, block, bad_dev);
    TSGPU_HIP_TRY(hipGetLastError());
    return TSGPU_OK;
}
int group_merge_knn(tsgpu_ctx* ctx, const uint64_t* gathered, uint64_t shard_stride_words, uint32_t n_shards, uint32_t n_q, uint32_t k,
                    float* dist_dev, uint64_t* label_dev, uint32_t* cnt_dev, hipStream_t s) {
    const uint64_t need = (uint64_t)n_shards * k;
    if (need > 8192) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_group: members * k > 8192");
    (void)hipSetDevice(ctx->device);
    if (need <= 256) hipLaunchKernelGGL((vec_group_merge_kernel<256>), dim3(n_q), dim3(VEC_THREADS), 0, s, gathered, shard_stride_words, n_shards, n_q, k, dist_dev, label_dev, cnt_dev);
    else if (need <= 1024) hipLaunchKernelGGL((vec_group_merge_kernel<1024>), dim3(n_q), dim3(VEC_THREADS), 0, s, gathered, shard_stride_words, n_shards, n_q, k, dist_dev, label_dev, cnt_dev);
    else if (need <= 2048) hipLaunchKernelGGL((vec_group_merge_kernel<2048>), dim3(n_q), dim3(VEC_THREADS), 0, s, gathered, shard_stride_words, n_shards, n_q, k, dist_dev, label_dev, cnt_dev);
    else hipLaunchKernelGGL((vec_group_merge_kernel<8192>), dim3(n_q), dim3(VEC_THREADS), 0, s, gathered, shard_stride_words, n_shards, n_q, k, dist_dev, label_dev, cnt_dev);
    TSGPU_HIP_TRY(hipGetLastError());
    return TSGPU_OK;
}
}  // namespace tsgpu
