// tsgpu.hip — implementation of the C-ABI of include/tsgpu.h: context, index mirror upload, and the batched
// keyword search (seam B1). Host code here only plans the launches (resolve terms, order lists, cut work
// items); all scoring happens in kw_kernels.hip.h on the GPU. There is no CPU fallback: a query this
// library does not accelerate is reported per query as TSGPU_ERR_UNSUPPORTED and left to the caller.
#include "tsgpu_host.h"
#if defined(__linux__)
#include <sys/prctl.h>
#include <time.h>
#endif
#include <cmath>
#include "kw_kernels.hip.h"
#include "kw_plan.hip.h"

using namespace tsgpu;

namespace {


uint64_t now_us() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(
               std::chrono::system_clock::now().time_since_epoch()).count();
}

int upload(DevBuf& dst, const void* src, size_t bytes, hipStream_t s) {
    int rc = dst.reserve(bytes ? bytes : 16);
    if (rc) return rc;
    if (bytes) TSGPU_HIP_TRY(hipMemcpyAsync(dst.p, src, bytes, hipMemcpyHostToDevice, s));
    return TSGPU_OK;
}

int refresh_column_table(tsgpu_ctx* ctx) {
    std::vector<const int64_t*> ptrs(ctx->columns.size());
    std::vector<uint32_t> lens(ctx->columns.size());
    for (size_t i = 0; i < ctx->columns.size(); i++) { ptrs[i] = ctx->columns[i].data.as<int64_t>(); lens[i] = ctx->columns[i].n; }
    int rc = upload(ctx->d_col_ptrs, ptrs.data(), ptrs.size() * sizeof(void*), ctx->stream);
    if (rc) return rc;
    rc = upload(ctx->d_col_len, lens.data(), lens.size() * sizeof(uint32_t), ctx->stream);
    if (rc) return rc;
    TSGPU_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return TSGPU_OK;
}

IndexView make_view(tsgpu_ctx* ctx, const Snapshot& sn) {
    IndexView v;
    v.lists = sn.lists.as<ListDesc>();
    v.blk_last = sn.ar ? sn.ar->blk_last.as<uint32_t>() : nullptr;
    v.blk_ids = sn.ar ? sn.ar->blk_ids.as<BlockIds>() : nullptr;
    v.blk_meta = sn.ar ? sn.ar->blk_meta.as<BlockMeta>() : nullptr;
    v.ids_payload = sn.ar ? sn.ar->ids_payload.as<uint32_t>() : nullptr;
    v.payload = sn.ar ? sn.ar->payload.as<uint32_t>() : nullptr;
    v.columns = ctx->d_col_ptrs.as<const int64_t*>();
    v.column_len = ctx->d_col_len.as<uint32_t>();
    v.n_columns = (uint32_t)ctx->columns.size();
    v.num_docs = ctx->num_docs;
    v.iddir = sn.dir_pool ? sn.dir_pool->buf.as<uint2>() : nullptr;
    v.iddir_slot_entries = sn.dir_pool ? sn.dir_pool->slot_entries : 0u;
    v.iddir_cap_ids = sn.dir_pool ? sn.dir_pool->cap_ids : 0u;
    v.prof = ctx->d_prof.as<unsigned long long>();
    v.touched = nullptr;                              // per batch, option kw_count_touched
    v.mf = nullptr;                                   // per lane: set by the batch
    v.fbits = nullptr;
    v.t0 = nullptr; v.ticks_per_us = 100; v.cutoff = nullptr;
    return v;
}

template <int TMAX, int CAP>
void launch_search(hipStream_t s, uint32_t n_work, const IndexView& v, const KwQueryDev* q, const KwWorkItem* w,
                   const KwPartials& part, const uint32_t* aux, uint32_t* ids_out, bool s2) {
    // s2 = some query of the launch sorts by three keys; the two-key build (CAP 512 only) has a smaller LDS footprint
    if (CAP == 512 && !s2) hipLaunchKernelGGL((kw_search_kernel<TMAX, CAP, CAP != 512>), dim3(n_work), dim3(KW_THREADS), 0, s, v, q, w, part, aux, ids_out, (uint32_t*)nullptr, (const uint64_t*)nullptr);
    else hipLaunchKernelGGL((kw_search_kernel<TMAX, CAP, true>), dim3(n_work), dim3(KW_THREADS), 0, s, v, q, w, part, aux, ids_out, (uint32_t*)nullptr, (const uint64_t*)nullptr);
}

// two-kernel form: find (intersection -> hit records) then score (hit records -> partial top-K)
template <int TMAX>
void launch_find_score(int cap, hipStream_t s, uint32_t n_work, const IndexView& v, const KwQueryDev* q, const KwWorkItem* w, const KwPartials& part,
                       const uint32_t* aux, uint32_t* ids_out, bool s2, uint32_t* hits, const uint64_t* hit_off, hipEvent_t mid_ev = nullptr, bool pair = false, bool plain = false) {
    if (pair && v.touched) hipLaunchKernelGGL((kw_find2_kernel<TMAX, true>), dim3(n_work), dim3(KW_THREADS), 0, s, v, q, w, part, hits, hit_off);   // the byte-counting instantiation (measurement option)
    else if (pair) hipLaunchKernelGGL((kw_find2_kernel<TMAX>), dim3(n_work), dim3(KW_THREADS), 0, s, v, q, w, part, hits, hit_off);
    else hipLaunchKernelGGL((kw_search_kernel<TMAX, 512, true, true>), dim3(n_work), dim3(KW_THREADS), 0, s, v, q, w, part, aux, ids_out, hits, hit_off);
    if (mid_ev) (void)hipEventRecord(mid_ev, s);           // find | score boundary of the batch's first group (tsgpu_timings::kw_find_ms)
    if (cap == 512 && !s2 && plain && !ids_out) hipLaunchKernelGGL((kw_score_kernel<TMAX, 512, false, false, true>), dim3(n_work), dim3(KW_THREADS), 0, s, v, q, w, part, aux, ids_out, hits, hit_off);
    else if (cap == 512 && !s2) hipLaunchKernelGGL((kw_score_kernel<TMAX, 512, false>), dim3(n_work), dim3(KW_THREADS), 0, s, v, q, w, part, aux, ids_out, hits, hit_off);
    else if (cap == 512) hipLaunchKernelGGL((kw_score_kernel<TMAX, 512, true>), dim3(n_work), dim3(KW_THREADS), 0, s, v, q, w, part, aux, ids_out, hits, hit_off);
    else if (cap == 1024) hipLaunchKernelGGL((kw_score_kernel<TMAX, 1024, true>), dim3(n_work), dim3(KW_THREADS), 0, s, v, q, w, part, aux, ids_out, hits, hit_off);
    else hipLaunchKernelGGL((kw_score_kernel<TMAX, 2048, true>), dim3(n_work), dim3(KW_THREADS), 0, s, v, q, w, part, aux, ids_out, hits, hit_off);
}

template <int TMAX>
void launch_search_mf_cap(int cap, hipStream_t s, uint32_t n_work, const IndexView& v, const KwQueryDev* q, const KwWorkItem* w,
                          const KwPartials& part, const uint32_t* aux, uint32_t* ids_out) {
    if (cap == 512) hipLaunchKernelGGL((kw_search_mf_kernel<TMAX, 512>), dim3(n_work), dim3(KW_THREADS), 0, s, v, q, w, part, aux, ids_out, (uint32_t*)nullptr, (const uint64_t*)nullptr);
    else hipLaunchKernelGGL((kw_search_mf_kernel<TMAX, 1024>), dim3(n_work), dim3(KW_THREADS), 0, s, v, q, w, part, aux, ids_out, (uint32_t*)nullptr, (const uint64_t*)nullptr);
}
// two-kernel form of the multi-field search (k + 256 <= 1024 there); oc: queries with filter + excluded ids, counted in id order between the two
struct MfOrderedCount { const uint32_t* jobs = nullptr; uint32_t n_jobs = 0, table_first = 0; const KwWorkItem* work_all = nullptr; KwPartials part_all{}; };
template <int TMAX>
void launch_find_score_mf(int cap, hipStream_t s, uint32_t n_work, const IndexView& v, const KwQueryDev* q, const KwWorkItem* w, const KwPartials& part,
                          const uint32_t* aux, uint32_t* ids_out, uint32_t* hits, const uint64_t* hit_off, const MfOrderedCount& oc, bool plain = false, int pipelined_lists = 0) {
    // pipelined_lists: 2 / 4 = the pipelined find kernel's instantiation that covers every multi-field query of the launch (kw_find_mf2.hip.h); 0 = kw_search_mf_kernel.
    // Same hit records either way.
    if (pipelined_lists == 2) hipLaunchKernelGGL((kw_find_mf2_kernel<TMAX, 2>), dim3(n_work), dim3(KW_THREADS), 0, s, v, q, w, part, hits, hit_off);
    else if (pipelined_lists == 4) hipLaunchKernelGGL((kw_find_mf2_kernel<TMAX, 4>), dim3(n_work), dim3(KW_THREADS), 0, s, v, q, w, part, hits, hit_off);
    else hipLaunchKernelGGL((kw_search_mf_kernel<TMAX, 512, true>), dim3(n_work), dim3(KW_THREADS), 0, s, v, q, w, part, aux, ids_out, hits, hit_off);
    if (oc.n_jobs) hipLaunchKernelGGL((kw_mf_ordered_count_kernel<TMAX>), dim3(oc.n_jobs), dim3(64), 0, s, q, oc.work_all, oc.part_all, hits, hit_off, oc.table_first, aux, oc.jobs);
    if (cap == 512 && plain && !ids_out && !oc.n_jobs) hipLaunchKernelGGL((kw_score_kernel<TMAX, 512, true, true, true>), dim3(n_work), dim3(KW_THREADS), 0, s, v, q, w, part, aux, ids_out, hits, hit_off);
    else if (cap == 512) hipLaunchKernelGGL((kw_score_kernel<TMAX, 512, true, true>), dim3(n_work), dim3(KW_THREADS), 0, s, v, q, w, part, aux, ids_out, hits, hit_off);
    else hipLaunchKernelGGL((kw_score_kernel<TMAX, 1024, true, true>), dim3(n_work), dim3(KW_THREADS), 0, s, v, q, w, part, aux, ids_out, hits, hit_off);
}

template <int TMAX>
void launch_search_cap(int cap, hipStream_t s, uint32_t n_work, const IndexView& v, const KwQueryDev* q, const KwWorkItem* w,
                       const KwPartials& part, const uint32_t* aux, uint32_t* ids_out, bool s2) {
    if (cap == 512) launch_search<TMAX, 512>(s, n_work, v, q, w, part, aux, ids_out, s2);
    else if (cap == 1024) launch_search<TMAX, 1024>(s, n_work, v, q, w, part, aux, ids_out, s2);
    else launch_search<TMAX, 2048>(s, n_work, v, q, w, part, aux, ids_out, s2);
}

void launch_merge(int cap, hipStream_t s, uint32_t n_q, const KwQueryDev* q, const KwPartials& part, const KwOut& out,
                  uint32_t* ids_out, const KwWorkItem* w, uint32_t select_min) {
    if (cap == 512) hipLaunchKernelGGL((kw_merge_kernel<512>), dim3(n_q), dim3(KW_THREADS), 0, s, q, part, out, ids_out, w, select_min);
    else if (cap == 1024) hipLaunchKernelGGL((kw_merge_kernel<1024>), dim3(n_q), dim3(KW_THREADS), 0, s, q, part, out, ids_out, w, select_min);
    else hipLaunchKernelGGL((kw_merge_kernel<2048>), dim3(n_q), dim3(KW_THREADS), 0, s, q, part, out, ids_out, w, select_min);
}

}  // namespace

extern "C" {

int tsgpu_abi_version(void) { return TSGPU_ABI_VERSION; }
const char* tsgpu_last_error(void) { return tls_error().c_str(); }

int tsgpu_create(int device_ordinal, tsgpu_ctx** out) {
    if (!out) return fail(TSGPU_ERR_INVALID, "tsgpu_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(TSGPU_ERR_DEVICE, "tsgpu_create: no HIP device visible (this library has no CPU fallback)");
    if (device_ordinal < 0 || device_ordinal >= n) return fail(TSGPU_ERR_INVALID, "tsgpu_create: bad device ordinal");
    TSGPU_HIP_TRY(hipSetDevice(device_ordinal));
    tsgpu_ctx* ctx = new (std::nothrow) tsgpu_ctx;
    if (!ctx) return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_create: host allocation failed");
    ctx->device = device_ordinal;
    {
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device_ordinal) == hipSuccess && khz >= 1000) ctx->ticks_per_us = (uint32_t)(khz / 1000);
    }
    std::atomic_store(&ctx->snap, std::shared_ptr<const Snapshot>(std::make_shared<Snapshot>()));
    bool good = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess;
    for (auto& ev : ctx->ev) good = good && hipEventCreate(&ev) == hipSuccess;
    for (int l = 0; l < tsgpu_ctx::N_LANES && good; l++) {
        KwLane& L = ctx->lanes[l];
        if (l == 0) { L.stream = ctx->stream; L.own_stream = false; }          // lane 0 shares the vector path's stream
        else good = good && hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) == hipSuccess;
        for (auto& ev : L.ev) good = good && hipEventCreate(&ev) == hipSuccess;
        good = good && hipEventCreateWithFlags(&L.ev_block, hipEventBlockingSync | hipEventDisableTiming) == hipSuccess;
        good = good && hipEventCreateWithFlags(&L.ev_chain, hipEventDisableTiming) == hipSuccess;
    }
    if (!good) { tsgpu_destroy(ctx); return fail(TSGPU_ERR_DEVICE, "tsgpu_create: stream / event creation failed"); }
    *out = ctx;
    return ok();
}

void tsgpu_vec_destroy_all(tsgpu_ctx* ctx);   // tsgpu_vec.hip
void tsgpu_facet_destroy_all(tsgpu_ctx* ctx); // tsgpu_facet.hip
void tsgpu_groupby_destroy(tsgpu_ctx* ctx);   // tsgpu_groupby.inc.h

void tsgpu_destroy(tsgpu_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    tsgpu_vec_destroy_all(ctx);
    tsgpu_facet_destroy_all(ctx);
    tsgpu_groupby_destroy(ctx);
    for (auto& L : ctx->lanes) L.release();
    std::atomic_store(&ctx->snap, std::shared_ptr<const Snapshot>());
    ctx->retire_bin->drain();
    deferred_frees().drain();
    ctx->d_col_ptrs.release(); ctx->d_col_len.release(); ctx->d_prof.release();
    for (auto& c : ctx->columns) c.data.release();
    for (auto& ev : ctx->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : ctx->aux_ev) if (ev) (void)hipEventDestroy(ev);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int tsgpu_set_stream(tsgpu_ctx* ctx, void* hip_stream) {
    if (!ctx) return fail(TSGPU_ERR_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lk(ctx->mu);
    std::lock_guard<std::mutex> lk0(ctx->lanes[0].mu);
    if (ctx->own_stream && ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    if (hip_stream) { ctx->stream = (hipStream_t)hip_stream; ctx->own_stream = false; }
    else {
        TSGPU_HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    ctx->lanes[0].stream = ctx->stream;
    return ok();
}

uint64_t tsgpu_vec_device_bytes(tsgpu_ctx* ctx);   // tsgpu_vec.hip

uint64_t tsgpu_device_bytes(tsgpu_ctx* ctx) {
    if (!ctx) return 0;
    uint64_t b = ctx->snapshot()->bytes;
    for (auto& c : ctx->columns) b += c.data.cap;
    return b + tsgpu_vec_device_bytes(ctx);
}

// ---------------------------------------------------------------- keyword index mirror: tsgpu_index.hip (posting lists), columns here
int tsgpu_column_set(tsgpu_ctx* ctx, uint32_t column_id, const int64_t* values, const uint8_t* present, uint32_t n, int mem) {
    if (!ctx) return fail(TSGPU_ERR_INVALID, "ctx is NULL");
    if (column_id >= 4096) return fail(TSGPU_ERR_INVALID, "tsgpu_column_set: column_id too large");
    if (n && !values) return fail(TSGPU_ERR_INVALID, "tsgpu_column_set: values is NULL");
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    if (ctx->columns.size() <= column_id) ctx->columns.resize(column_id + 1);
    ColumnDev& c = ctx->columns[column_id];
    int rc = c.data.reserve((size_t)std::max<uint32_t>(n, 1) * sizeof(int64_t));
    if (rc) return rc;
    if (mem == TSGPU_MEM_DEVICE) {
        if (present) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_column_set: presence mask with device values is not supported");
        TSGPU_HIP_TRY(hipMemcpyAsync(c.data.p, values, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToDevice, ctx->stream));
        TSGPU_HIP_TRY(hipStreamSynchronize(ctx->stream));
        c.host.resize(n);
        if (n) TSGPU_HIP_TRY(hipMemcpy(c.host.data(), c.data.p, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost));
    } else {
        c.host.assign(values, values + n);
        if (present) for (uint32_t i = 0; i < n; i++) if (!present[i]) c.host[i] = INT64_MIN;   // default_score, src/index.cpp:5696
        if (n) TSGPU_HIP_TRY(hipMemcpy(c.data.p, c.host.data(), (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    c.n = n;
    return refresh_column_table(ctx);
}

int tsgpu_set_num_docs(tsgpu_ctx* ctx, uint32_t num_docs) {
    if (!ctx) return fail(TSGPU_ERR_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->num_docs = num_docs;
    ctx->num_docs_set = true;
    return ok();
}

int tsgpu_set_option(tsgpu_ctx* ctx, const char* name, int64_t value) {
    if (!ctx || !name) return fail(TSGPU_ERR_INVALID, "tsgpu_set_option: NULL argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!strcmp(name, "kw_max_partials")) {
        if (value < 1 || value > 4096) return fail(TSGPU_ERR_INVALID, "kw_max_partials out of range (1..4096)");
        ctx->kw_max_partials = (uint32_t)value; return ok();
    }
    if (!strcmp(name, "kw_cost_r_x10")) { ctx->kw_cost_r_x10 = (uint32_t)std::max<int64_t>(value, 0); return ok(); }
    if (!strcmp(name, "kw_cost_probe_x100")) { ctx->kw_cost_probe_x100 = (uint32_t)std::max<int64_t>(value, 0); return ok(); }
    if (!strcmp(name, "kw_cost_fixed")) { ctx->kw_cost_fixed = (uint32_t)std::max<int64_t>(value, 0); return ok(); }
    if (!strcmp(name, "kw_sort_work")) { ctx->kw_sort_work = value != 0; return ok(); }
    if (!strcmp(name, "kw_two_kernels")) { ctx->kw_two_kernels = value != 0; return ok(); }
    if (!strcmp(name, "kw_device_plan_min_queries")) { ctx->kw_device_plan_min_queries = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 1 << 30); return ok(); }
    if (!strcmp(name, "kw_pair_blocks")) { ctx->kw_pair_blocks = value != 0; return ok(); }
    if (!strcmp(name, "doc_range_lo")) { ctx->doc_range_set = true; ctx->doc_range_lo = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 0xFFFFFFFFll); return ok(); }
    if (!strcmp(name, "doc_range_hi")) { ctx->doc_range_set = true; ctx->doc_range_hi = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 0xFFFFFFFFll); return ok(); }
    if (!strcmp(name, "kw_candidates_rank_fold")) { ctx->kw_candidates_rank_fold = value != 0; return ok(); }
    if (!strcmp(name, "kw_mf_pipelined")) { ctx->kw_mf_pipelined = value != 0; return ok(); }
    if (!strcmp(name, "kw_count_touched")) { ctx->kw_count_touched = value != 0; return ok(); }
    if (!strcmp(name, "kw_iddir_min_ids")) { ctx->kw_iddir_min_ids = value < 0 ? 0 : value; ctx->commit_force_full = true; return ok(); }
    if (!strcmp(name, "kw_iddir_density_div")) { ctx->kw_iddir_density_div = value < 1 ? 1 : value; ctx->commit_force_full = true; return ok(); }
    if (!strcmp(name, "kw_iddir_budget_mb")) { ctx->kw_iddir_budget_mb = value < 0 ? 0 : value; ctx->commit_force_full = true; return ok(); }
    if (!strcmp(name, "kw_host_split_tail_slices")) { ctx->kw_host_split_tail_slices = value >= 2 ? 2 : 1; return ok(); }
    if (!strcmp(name, "kw_host_split_device_plan")) { ctx->kw_host_split_device_plan = value != 0; return ok(); }
    if (!strcmp(name, "kw_host_split_first_pct")) { ctx->kw_host_split_first_pct = (uint32_t)std::min<int64_t>(95, std::max<int64_t>(5, value)); return ok(); }
    if (!strcmp(name, "kw_host_split_queries")) { ctx->kw_host_split_queries = (uint32_t)std::max<int64_t>(0, value); return ok(); }
    if (!strcmp(name, "kw_zero_copy_max_queries")) { ctx->kw_zero_copy_max_queries = (uint32_t)std::max<int64_t>(0, value); return ok(); }
    if (!strcmp(name, "kw_timing_min_queries")) { ctx->kw_timing_min_queries = (uint32_t)std::max<int64_t>(0, value); return ok(); }
    if (!strcmp(name, "kw_merge_select_min")) { ctx->kw_merge_select_min = (uint32_t)std::max<int64_t>(0, value); return ok(); }
    if (!strcmp(name, "hnsw_visited_hash")) { ctx->hnsw_visited_hash = value != 0; return ok(); }
    if (!strcmp(name, "hnsw_test_tiny_cand")) { ctx->hnsw_test_tiny_cand = value != 0; return ok(); }
    if (!strcmp(name, "hnsw_visited_max_gib")) { ctx->hnsw_visited_max_gib = (int)std::min<int64_t>(std::max<int64_t>(1, value), 128); return ok(); }
    if (!strcmp(name, "blocking_sync_min_callers")) { ctx->blocking_sync_min_callers = (int)std::max<int64_t>(0, value); return ok(); }
    if (!strcmp(name, "plan_threads")) { if (value < 1 || value > 64) return fail(TSGPU_ERR_INVALID, "plan_threads: 1..64"); ctx->plan_threads = (int)value; return ok(); }
    if (!strcmp(name, "plan_parallel_min_queries")) { if (value < 0) return fail(TSGPU_ERR_INVALID, "plan_parallel_min_queries >= 0"); ctx->plan_parallel_min_queries = (uint32_t)value; return ok(); }
    if (!strcmp(name, "fuse_threads")) { if (value < 1 || value > 256) return fail(TSGPU_ERR_INVALID, "fuse_threads: 1..256"); ctx->fuse_threads = (int)value; return ok(); }
    if (!strcmp(name, "kw_hit_buffer_records")) {       // exact budget in hit records (tests); 0 = use kw_hit_buffer_mb
        if (value < 0) return fail(TSGPU_ERR_INVALID, "kw_hit_buffer_records must be >= 0");
        ctx->kw_hit_buffer_records = (uint64_t)value; return ok();
    }
    if (!strcmp(name, "kw_hit_buffer_mb")) {
        if (value < 1) return fail(TSGPU_ERR_INVALID, "kw_hit_buffer_mb must be >= 1");
        ctx->kw_hit_buffer_mb = (uint32_t)value; return ok();
    }
    if (!strcmp(name, "kw_chunk_blocks")) {
        if (value < 0 || value > KW_MAX_CHUNK) return fail(TSGPU_ERR_INVALID, "kw_chunk_blocks out of range (0 = auto, 1..KW_MAX_CHUNK)");
        ctx->kw_chunk_blocks = (uint32_t)value;
        return ok();
    }
    if (!strcmp(name, "vec_rows_per_slab")) {
        if (value != 0 && (value < 128 || value > (1 << 24))) return fail(TSGPU_ERR_INVALID, "vec_rows_per_slab out of range");
        ctx->vec_rows_per_slab = (uint32_t)value;
        return ok();
    }
    if (!strcmp(name, "vec_sample_tiles")) {
        if (value < 0 || value > (1 << 20)) return fail(TSGPU_ERR_INVALID, "vec_sample_tiles out of range");
        ctx->vec_sample_tiles = (uint32_t)value;
        return ok();
    }
    if (!strcmp(name, "vec_cand_cap")) {
        if (value < 0 || value > (1 << 24)) return fail(TSGPU_ERR_INVALID, "vec_cand_cap out of range");
        ctx->vec_cand_cap = (uint32_t)value;
        return ok();
    }
    if (!strcmp(name, "vec_count_rescored")) { ctx->vec_count_rescored = value != 0; return ok(); }
    // micro-batcher (tsgpu_batcher.h): concurrent small calls are coalesced into one launch
    if (!strcmp(name, "index_min_slack_words")) { ctx->index_min_slack_words = (uint64_t)std::max<int64_t>(value, 0); return ok(); }   // (tests: small arenas)
    if (!strcmp(name, "index_compact_min_words")) { ctx->index_compact_min_words = (uint64_t)std::max<int64_t>(value, 0); return ok(); }   // (tests: small arenas)
    if (!strcmp(name, "commit_full")) { ctx->commit_force_full = value != 0; return ok(); }      // the NEXT commit re-packs every list (compaction)
    if (!strcmp(name, "kw_lanes")) {                   // execution lanes (stream + scratch each) that concurrent keyword batches spread over
        if (value < 1 || value > tsgpu_ctx::N_LANES) return fail(TSGPU_ERR_INVALID, "kw_lanes out of range (1..8)");
        ctx->n_lanes = (int)value; return ok();
    }
    if (!strcmp(name, "facet_ids_per_block")) {        // result ids per workgroup of the facet counting launch; 0 = chosen per batch (tests: the multi-round walk on small inputs)
        if (value < 0 || value > 4096 || (value % 256) != 0) return fail(TSGPU_ERR_INVALID, "facet_ids_per_block must be 0 or a multiple of 256 up to 4096");
        ctx->facet_ids_per_block = (uint32_t)value; return ok();
    }
    if (!strcmp(name, "hybrid_overlap")) { ctx->hybrid_overlap = value != 0; return ok(); }
    if (!strcmp(name, "vec_batch_post_window_us")) { ctx->vec_batch_post_window_us = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 100000); return ok(); }
    if (!strcmp(name, "batch_window_us")) { ctx->batch_window_us = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 100000); return ok(); }
    if (!strcmp(name, "batch_max_queries")) { ctx->batch_max_queries = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 1 << 20); return ok(); }   // 0 = never coalesce
    if (!strcmp(name, "batch_round_queries")) { ctx->batch_round_queries = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 1), 1 << 20); return ok(); }
    if (!strcmp(name, "vec_ip_lanes")) {               // summation order of every exact distance = the SIMD level hnswlib is compiled for in the server
        if (value != 4 && value != 8 && value != 16) return fail(TSGPU_ERR_INVALID, "vec_ip_lanes must be 4 (SSE, stock build), 8 (AVX) or 16 (AVX-512)");
        ctx->vec_ip_lanes = (uint32_t)value;
        return ok();
    }
    if (!strcmp(name, "vec_prefilter")) {
        if (value < 0 || value > 1) return fail(TSGPU_ERR_INVALID, "vec_prefilter must be 0 (fp32 MFMA scan) or 1 (bf16 bracket scan)");
        ctx->vec_prefilter = (uint32_t)value;
        return ok();
    }
    return fail(TSGPU_ERR_NOT_FOUND, std::string("tsgpu_set_option: unknown option ") + name);
}

int tsgpu_get_counter(tsgpu_ctx* ctx, const char* name, uint64_t* out) {
    if (!ctx || !name || !out) return fail(TSGPU_ERR_INVALID, "tsgpu_get_counter: NULL argument");
    std::lock_guard<std::mutex> lk(ctx->tm_mu);
    if (!strcmp(name, "kw_last_hit_groups")) { *out = ctx->kw_last_hit_groups; return ok(); }      // last keyword batch: find+score groups (0 = fused kernel)
    if (!strcmp(name, "kw_candidates_rank_launches")) { *out = ctx->kw_candidates_rank_launches; return ok(); }
    if (!strcmp(name, "kw_mf_pipelined_launches")) { *out = ctx->kw_mf_pipelined_launches; return ok(); }     // find launches served by kw_find_mf2_kernel
    if (!strcmp(name, "kw_last_hit_records")) { *out = ctx->kw_last_hit_records; return ok(); }    // hit-record capacity the last batch asked for
    if (!strcmp(name, "vec_overflow_rounds")) { *out = ctx->vec_overflow_rounds; return ok(); }
    if (!strcmp(name, "vec_prefilter_fallbacks")) { *out = ctx->vec_prefilter_fallbacks; return ok(); }
    if (!strcmp(name, "vec_prefilter_groups")) { *out = ctx->vec_prefilter_groups; return ok(); }
    if (!strcmp(name, "vec_rescored_rows")) { *out = ctx->vec_rescored_rows; return ok(); }
    if (!strcmp(name, "hnsw_last_expansions")) { *out = ctx->hnsw_last_expansions; return ok(); }
    if (!strcmp(name, "hnsw_tier_reruns")) { *out = ctx->hnsw_tier_reruns; return ok(); }
    if (!strcmp(name, "hnsw_last_distances")) { *out = ctx->hnsw_last_distances; return ok(); }
    if (!strcmp(name, "commit_last_us")) { *out = ctx->commit_last_us; return ok(); }                        // the last tsgpu_commit: wall time, bytes uploaded
    if (!strcmp(name, "commit_last_uploaded_bytes")) { *out = ctx->commit_last_uploaded_bytes; return ok(); }
    if (!strcmp(name, "commit_failed_count")) { *out = ctx->commit_failed_count; return ok(); }
    if (!strcmp(name, "commit_compactions")) { *out = ctx->commit_compactions; return ok(); }             // full commits taken because garbage outweighed the live words
    if (!strcmp(name, "index_live_words")) { const auto sn = ctx->snapshot(); *out = sn && sn->ar ? sn->ar->live_idw + sn->ar->live_pw : 0; return ok(); }
    if (!strcmp(name, "index_used_words")) { const auto sn = ctx->snapshot(); *out = sn && sn->ar ? sn->ar->used_idw + sn->ar->used_pw : 0; return ok(); }
    if (!strcmp(name, "commit_full_count")) { *out = ctx->commit_full_count; return ok(); }                  // commits that re-packed everything / appended at the tails
    if (!strcmp(name, "commit_incremental_count")) { *out = ctx->commit_incremental_count; return ok(); }
    if (!strcmp(name, "kw_batches")) { *out = ctx->kw_batches.load(); return ok(); }                 // host-side phase totals (us) over all keyword batches
    if (!strcmp(name, "kw_plan_us")) { *out = ctx->kw_plan_us.load(); return ok(); }
    if (!strcmp(name, "kw_max_plan_us")) { *out = ctx->kw_max_plan_us.exchange(0); return ok(); }       // (reading resets the maxima)
    if (!strcmp(name, "kw_max_upload_us")) { *out = ctx->kw_max_upload_us.exchange(0); return ok(); }
    if (!strcmp(name, "kw_max_launch_us")) { *out = ctx->kw_max_launch_us.exchange(0); return ok(); }
    if (!strcmp(name, "kw_max_wait_us")) { *out = ctx->kw_max_wait_us.exchange(0); return ok(); }
    if (!strcmp(name, "kw_queue_us")) { *out = ctx->kw_queue_us.load(); return ok(); }
    if (!strcmp(name, "kw_wake_us")) { *out = ctx->kw_wake_us.load(); return ok(); }
    if (!strcmp(name, "kw_max_queue_us")) { *out = ctx->kw_max_queue_us.exchange(0); return ok(); }     // parked -> its round starts executing
    if (!strcmp(name, "kw_max_wake_us")) { *out = ctx->kw_max_wake_us.exchange(0); return ok(); }       // results ready -> the caller runs again
    if (!strcmp(name, "kw_upload_us")) { *out = ctx->kw_upload_us.load(); return ok(); }
    if (!strcmp(name, "kw_launch_us")) { *out = ctx->kw_launch_us.load(); return ok(); }
    if (!strcmp(name, "kw_wait_us")) { *out = ctx->kw_wait_us.load(); return ok(); }
    if (!strcmp(name, "kw_book_us")) { *out = ctx->kw_book_us.load(); return ok(); }
    if (!strcmp(name, "kw_device_plans")) { *out = ctx->kw_device_plans.load(); return ok(); }                     // batches planned on the device / sent back to the host planner
    if (!strcmp(name, "kw_device_plan_fallbacks")) { *out = ctx->kw_device_plan_fallbacks.load(); return ok(); }
    if (!strcmp(name, "kw_iddir_built")) { *out = ctx->kw_iddir_built; return ok(); }                               // id directories (re)built by commits so far
    if (!strcmp(name, "kw_iddir_lists")) { const auto sn = ctx->snapshot(); uint64_t n = 0; if (sn) for (const auto& r : sn->dir_of) n += r ? 1 : 0; *out = n; return ok(); }   // lists of the current snapshot that carry one
    if (!strcmp(name, "batch_exec_us")) { *out = ctx->batch_exec_us.load(); return ok(); }           // coalesced rounds: batch execution / hand-out to the callers
    if (!strcmp(name, "batch_scatter_us")) { *out = ctx->batch_scatter_us.load(); return ok(); }
    if (!strcmp(name, "batch_rounds")) { *out = ctx->kw_comb.rounds + ctx->vec_comb.rounds; return ok(); }               // coalesced rounds executed so far
    if (!strcmp(name, "gb_batch_rounds")) { *out = ctx->gb_comb.rounds; return ok(); }                                          // ... of grouped keyword calls
    if (!strcmp(name, "gb_batch_coalesced_calls")) { *out = ctx->gb_comb.coalesced_calls; return ok(); }
    if (!strcmp(name, "batch_coalesced_calls")) { *out = ctx->kw_comb.coalesced_calls + ctx->vec_comb.coalesced_calls; return ok(); }   // calls served by those rounds
    return fail(TSGPU_ERR_NOT_FOUND, std::string("tsgpu_get_counter: unknown counter ") + name);
}

int tsgpu_keep_result_ids(tsgpu_ctx* ctx, int keep) {
    if (!ctx) return fail(TSGPU_ERR_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->keep_ids = keep != 0;
    return ok();
}

// ---------------------------------------------------------------- keyword search (seam B1)
namespace {
struct Plan {
    std::vector<KwQueryDev> q;
    std::vector<KwWorkItem> work_small, work_big;   // TMAX 3 / TMAX 10 kernels
    std::vector<KwWorkItem> work_mf_small, work_mf_big;   // multi-field kernels, TMAX 3 / TMAX 10
    std::vector<KwWorkItem> work_wild;                    // wildcard scans
    std::vector<KwQueryMF> mf;
    std::vector<KwMergeGroup> groups;                     // first level of the two-level merge (queries with many work items)
    std::vector<uint32_t> ordered_count_q;                // multi-field queries with filter AND excluded ids (kw_mf_ordered_count_kernel)
    uint64_t fbits_words = 0;                             // rank bitmaps of the filtered multi-field queries
    bool any_deadline = false;                            // some query carries a deadline: stamp the batch start, collect cutoff flags
    bool any_s2 = false;              // some query has a third sort key
    bool any_aux = false;             // some query has filter ids or excluded ids (else the score kernel's PLAIN instantiation serves the batch)
    bool any_array = false;           // some multi-field query has a string[] field
    uint32_t mf_max_fields = 0;       // most query_by fields of any multi-field query (<= 2: the pipelined find kernel, kw_find_mf2.hip.h)
    std::vector<uint32_t> aux;
    std::vector<int32_t> status, cutoff;
    uint32_t max_k = 1;
    uint64_t ids_total = 0;
    uint64_t list_bytes = 0;       // 4 * sum |L_t| over the batch
    uint64_t n_numeric_sort_q = 0;
    uint32_t chunk_blocks = 64;
};
}

static uint32_t resolve_topster_size(const tsgpu_ctx* ctx, const tsgpu_kw_query& in) {
    uint32_t k = in.topster_size;
    if (k == 0) {                                                      // src/index.cpp:3506-3512
        k = TSGPU_DEFAULT_TOPSTER_SIZE;
        k = in.n_filter ? std::min<uint32_t>(k, in.n_filter) : std::min<uint32_t>(k, std::max<uint32_t>(ctx->num_docs, 1));
    } else k = std::min<uint32_t>(k, std::max<uint32_t>(ctx->num_docs, 1));
    return std::max<uint32_t>(k, 1);
}

static int plan_batch(tsgpu_ctx* ctx, const Snapshot& snap, const tsgpu_kw_query* queries, uint32_t n_queries, Plan& P, bool keep_ids, bool wildcard, const KwVFlat* vflat = nullptr,
                      const uint16_t* present_elsewhere = nullptr) {
    // driver blocks per work item: fixed by the option, or (0 = auto) sized so that the batch yields a few thousand work
    // items (>= 3 per resident workgroup slot) without fragmenting queries into more partial top-K lists than needed
    uint32_t KW_CHUNK_BLOCKS = ctx->kw_chunk_blocks;
    std::vector<uint32_t> handle_cache;      // single-field queries: list handle per token (KW_NONE - 1 = absent), looked up once
    std::vector<uint8_t> cached(n_queries, 0);
    if (KW_CHUNK_BLOCKS == 0) {
        handle_cache.resize((size_t)n_queries * TSGPU_MAX_QUERY_TOKENS);
        uint64_t total_blocks = 0;
        for (uint32_t i = 0; i < n_queries; i++) {
            const tsgpu_kw_query& in = queries[i];
            if (in.n_tokens == 0 || in.n_tokens > TSGPU_MAX_QUERY_TOKENS || in.n_fields == 0 || in.n_fields > (uint32_t)KW_MAX_FIELDS) continue;
            if (in.n_fields != 1) {
                // several query_by fields: the driver is the token with the fewest postings over all fields, one group of work items per field list
                // of it — its blocks enter the chunk rule, too (before, such batches were always cut with the minimum chunk: 2 000 two-field queries on
                // 10M documents = 101 000 work items of 16 blocks)
                uint64_t best_ids = ~0ull; uint32_t best_blocks = 0;
                for (uint32_t t = 0; t < in.n_tokens; t++) {
                    uint64_t ids = 0; uint32_t blocks = 0; bool found = false;
                    for (uint32_t f = 0; f < in.n_fields; f++) {
                        const uint32_t h = snap.find_handle(in.field_ids[f], in.term_ids[t]);
                        if (h == 0xFFFFFFFFu) continue;
                        found = true; ids += snap.h_lists[h].n_ids; blocks += snap.h_lists[h].n_blocks;
                    }
                    if (found && ids < best_ids) { best_ids = ids; best_blocks = blocks; }
                }
                total_blocks += best_blocks / 4;        // (a multi-field driver block costs ~4x a single-field one — two tile merges, wider records — so these batches are
                                                        //  cut finer; measured, 2 000 two-field queries on 10M documents: 64-block items 20.4 ms, 16 -> 23.2, 256 -> 24.4)
                continue;
            }
            uint32_t best = 0xFFFFFFFFu;
            for (uint32_t t = 0; t < in.n_tokens; t++) {
                const uint32_t h = snap.find_handle(in.field_ids[0], in.term_ids[t]);
                handle_cache[(size_t)i * TSGPU_MAX_QUERY_TOKENS + t] = h != 0xFFFFFFFFu ? h : KW_NONE - 1;     // remembered for the main pass
                if (h != 0xFFFFFFFFu) best = std::min(best, snap.h_lists[h].n_blocks);
            }
            cached[i] = 1;
            if (best != 0xFFFFFFFFu) total_blocks += best;
        }
        uint64_t c = total_blocks / 3000;
        KW_CHUNK_BLOCKS = 16;
        while (KW_CHUNK_BLOCKS < (uint32_t)KW_MAX_CHUNK && KW_CHUNK_BLOCKS * 2 <= c) KW_CHUNK_BLOCKS *= 2;
        // a small batch is as slow as its longest work item, and with the selecting merge (kw_select_partials) the merge no longer grows
        // with the number of partial lists: cut finer (measured on 10M docs: 16 queries 0.256 -> 0.223 ms, 64 queries 0.330 -> 0.305 ms;
        // from 256 queries on the chip is full and coarser items win again)
        if (ctx->kw_merge_select_min && n_queries <= 128) KW_CHUNK_BLOCKS = 8;
    }
    static const bool plan_timing = getenv("TSGPU_HOST_TIMING") != nullptr;
    const uint64_t tp0 = now_us();
    P.chunk_blocks = KW_CHUNK_BLOCKS;
    P.q.resize(n_queries);
    P.status.assign(n_queries, TSGPU_OK);
    P.cutoff.assign(n_queries, 0);
    const uint64_t now = now_us();
    std::vector<uint32_t> q_begin(n_queries, 0), q_cnt(n_queries, 0);
    std::vector<double> item_cost(n_queries, 0.0);
    // The per-query part of the plan (handles, work items, id arenas) is independent per query: big batches are planned in slices on
    // the context's parked host threads, every slice into its own accumulator; the slices are then concatenated in query order and the
    // offsets a query holds into the shared arenas shifted by its slice's base (0.5 ms -> 0.1 ms of a 9 ms step at 10 000 queries).
    struct PlanAcc {
        std::vector<uint32_t> aux, ordered_count_q; std::vector<KwQueryMF> mf; std::vector<KwWorkItem> flat_work;
        uint64_t fbits_words = 0, ids_total = 0, list_bytes = 0;
        uint32_t max_k = 0, n_numeric_sort_q = 0;
        bool any_deadline = false, any_s2 = false, any_aux = false, any_array = false;
        uint32_t mf_max_fields = 0;
    };
    auto plan_range = [&](uint32_t lo, uint32_t hi, PlanAcc& A) {
        A.flat_work.reserve((size_t)(hi - lo) * 4);
        for (uint32_t i = lo; i < hi; i++) {
            const tsgpu_kw_query& in = queries[i];
            KwQueryDev& q = P.q[i];
            memset(&q, 0, sizeof q);
            q.k = 1;
            auto unsupported = [&](const char*) { P.status[i] = TSGPU_ERR_UNSUPPORTED; };
            if (!wildcard && (in.n_tokens == 0 || in.n_tokens > TSGPU_MAX_QUERY_TOKENS)) { unsupported("tokens"); continue; }
            if (!wildcard && (in.n_fields == 0 || in.n_fields > (uint32_t)KW_MAX_FIELDS)) { unsupported("fields"); continue; }
            if (!wildcard) {
                int bad = 0;
                for (uint32_t f = 0; f < in.n_fields && !bad; f++) {
                    if (snap.field_is_array.find(in.field_ids[f]) == snap.field_is_array.end()) bad = TSGPU_ERR_NOT_FOUND;
                }
                if (bad) { P.status[i] = bad; continue; }
            }
            // several fields, or a string[] field: the general kernel (per-candidate probes, per-field scoring incl. the array readers);
            // the block-merge kernel stays free of the array code (it costs 2x the registers)
            bool multi = !wildcard && in.n_fields > 1;
            if (!wildcard && !multi && snap.field_is_array.at(in.field_ids[0])) multi = true;
            // dropped tokens (drop_tokens passes): probed and scored per candidate, never required -> the general kernel
            if (!wildcard && in.n_dropped != 0) {
                if (in.n_dropped > TSGPU_MAX_DROPPED_TOKENS || in.n_tokens + in.n_dropped > TSGPU_MAX_QUERY_TOKENS) { unsupported("dropped tokens"); continue; }
                multi = true;
            }
            // filter ids with several query_by fields: num_keyword_matches has an order-free form only without exclusions (kw_score_stage)
            // filter ids AND excluded ids with several query_by fields: num_keyword_matches needs the intersection in id order — counted by
            // kw_mf_ordered_count_kernel from the find kernel's hit records, i.e. in the two-kernel form only (checked after the tables are laid out)
            const bool ordered_count = multi && in.n_filter != 0 && in.n_excluded != 0;
            if (ordered_count && !ctx->kw_two_kernels) { unsupported("filter ids AND excluded ids with several query_by fields need the two-kernel form"); continue; }
            if (in.n_sort > TSGPU_MAX_SORT_KEYS) { P.status[i] = TSGPU_ERR_INVALID; continue; }
            if (in.n_filter != 0 && !in.filter_ids) { P.status[i] = TSGPU_ERR_INVALID; continue; }
            if (in.match_type > TSGPU_SUM_SCORE) { P.status[i] = TSGPU_ERR_INVALID; continue; }
            bool bad_sort = false;
            for (uint32_t s = 0; s < in.n_sort; s++) {
                if (in.sort[s].kind > TSGPU_SORT_INT64_COLUMN && !(vflat && in.sort[s].kind == TSGPU_SORT_VECTOR_DISTANCE)) bad_sort = true;   // vector_distance belongs to the vector/hybrid entry points
                if (in.sort[s].kind == TSGPU_SORT_INT64_COLUMN && in.sort[s].column >= ctx->columns.size()) bad_sort = true;
                if (in.sort[s].order != 1 && in.sort[s].order != -1) bad_sort = true;
            }
            if (bad_sort) { unsupported("sort"); continue; }
            const uint32_t k = vflat ? std::max<uint32_t>(in.topster_size, 1) : resolve_topster_size(ctx, in);     // (the vector branch resolved it against ITS filter / row count)
            if (k > TSGPU_MAX_TOPK) { unsupported("topster_size"); continue; }
            if (in.deadline_us != 0 && now > in.deadline_us) { P.status[i] = TSGPU_ERR_DEADLINE; P.cutoff[i] = 1; continue; }
            if (in.deadline_us != 0) { q.deadline_rem_us = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(in.deadline_us - now, 1), 0xFFFFFFFFull); A.any_deadline = true; }

            if (wildcard) {
                // Index::search_wildcard (src/index.cpp:6616-6818): rank every filter id (every seq_id without a filter) by its sort keys
                q.mf_index = KW_NONE;
                q.n_sort = (uint8_t)in.n_sort;
                for (uint32_t s = 0; s < in.n_sort; s++) {
                    q.sort_kind[s] = in.sort[s].kind; q.sort_order[s] = in.sort[s].order; q.sort_col[s] = in.sort[s].column;
                    if (in.sort[s].kind == TSGPU_SORT_INT64_COLUMN) A.n_numeric_sort_q++;
                }
                q.k = k;
                A.max_k = std::max(A.max_k, k);
                q.n_excl = in.n_excluded;
                q.n_filt = in.n_filter;
                if (vflat) {
                    // flat vector branch: query i ranks ITS row of the distance matrix (aligned with the filter ids); the queries of a call share
                    // one filter (same host array): uploaded once per planning slice
                    if (in.n_excluded || !in.n_filter) { P.status[i] = TSGPU_ERR_INVALID; continue; }
                    q.vdist = (uint64_t)(uintptr_t)(vflat->dist_dev + (size_t)i * vflat->stride);
                    q.vdist_thr = vflat->threshold; q.vdist_abs = vflat->abs ? 1 : 0;
                }
                if (vflat && i > lo && P.status[i - 1] == TSGPU_OK && queries[i - 1].filter_ids == in.filter_ids && queries[i - 1].n_filter == in.n_filter) q.aux_off = P.q[i - 1].aux_off;
                else {
                    q.aux_off = (uint32_t)A.aux.size();
                    if (in.n_excluded) {
                        if (!in.excluded_ids) { P.status[i] = TSGPU_ERR_INVALID; continue; }
                        A.aux.insert(A.aux.end(), in.excluded_ids, in.excluded_ids + in.n_excluded);
                    }
                    if (in.n_filter) A.aux.insert(A.aux.end(), in.filter_ids, in.filter_ids + in.n_filter);
                }
                uint32_t n_ids = in.n_filter ? in.n_filter : ctx->num_docs;
                if (!vflat && ctx->doc_range_set) {
                    // a doc-range shard ranks the ids it OWNS: the filter ids inside [lo, hi) (a sub-array: they ascend), or lo .. hi - 1
                    const uint32_t lo_id = ctx->doc_range_lo, hi_id = std::min(ctx->doc_range_hi, ctx->num_docs);
                    if (in.n_filter) {
                        const uint32_t* fb = in.filter_ids;
                        const uint32_t a = (uint32_t)(std::lower_bound(fb, fb + in.n_filter, lo_id) - fb), b = (uint32_t)(std::lower_bound(fb, fb + in.n_filter, hi_id) - fb);
                        A.aux.resize(A.aux.size() - in.n_filter);                       // (the filter ids were appended last: keep the sub-array)
                        A.aux.insert(A.aux.end(), fb + a, fb + b);
                        q.n_filt = b - a;
                        n_ids = b - a;
                    } else { n_ids = hi_id > lo_id ? hi_id - lo_id : 0; q.wild_base = lo_id; }
                }
                q.wild_n_ids = n_ids;
                if (n_ids == 0) continue;                                               // (nothing of this query on this shard: zero hits, status 0)
                uint32_t n_num = 0;
                for (uint32_t s = 0; s < in.n_sort; s++) n_num += in.sort[s].kind == TSGPU_SORT_INT64_COLUMN;
                A.list_bytes += 4ull * in.n_filter + 8ull * n_ids * n_num;      // the id array + one column value per id and numeric key
                q.ids_out_off = A.ids_total;
                const uint32_t n_blocks = (n_ids + BLOCK_IDS - 1) / BLOCK_IDS;
                if (keep_ids) A.ids_total += (uint64_t)n_blocks * BLOCK_IDS;
                const uint32_t WCHUNK = 64;                                     // 16K ids per work item
                for (uint32_t b = 0; b < n_blocks; b += WCHUNK) {
                    KwWorkItem w;
                    w.query = i; w.blk_begin = b; w.blk_end = std::min(n_blocks, b + WCHUNK); w.ids_out_off = b * BLOCK_IDS;
                    { if (q_cnt[i] == 0) q_begin[i] = (uint32_t)A.flat_work.size(); A.flat_work.push_back(w); q_cnt[i]++; }
                }
                continue;
            }

            q.n_query_tokens = in.n_tokens;
            q.mf_index = KW_NONE;
            uint32_t nl = 0;
            uint32_t len_of[KW_MAX_TOKENS];
            KwQueryMF mfq;
            if (multi) memset(&mfq, 0xFF, sizeof mfq);                // (only read by the multi-field form)
            bool empty_here = false;
            for (uint32_t t = 0; t < in.n_tokens; t++) {
                // one or_iterator per token = the union of its lists over the fields; a token found in no field is skipped (src/index.cpp:5651-5655)
                uint64_t tot = 0;
                bool found = false;
                for (uint32_t f = 0; f < in.n_fields; f++) {
                    uint32_t handle;
                    if (cached[i]) {                     // (n_fields == 1)
                        handle = handle_cache[(size_t)i * TSGPU_MAX_QUERY_TOKENS + t];
                        if (handle == KW_NONE - 1) continue;
                    } else {
                        handle = snap.find_handle(in.field_ids[f], in.term_ids[t]);
                        if (handle == 0xFFFFFFFFu) continue;
                    }
                    if (!found) q.list[nl] = handle;
                    found = true;
                    mfq.list[nl][f] = handle;
                    tot += snap.h_lists[handle].n_ids;
                    A.list_bytes += 4ull * snap.h_lists[handle].n_ids;
                }
                if (!found) { if (present_elsewhere && ((present_elsewhere[i] >> t) & 1u)) empty_here = true; continue; }      // (exists on another shard: an EMPTY list here)
                len_of[nl] = (uint32_t)std::min<uint64_t>(tot, 0xFFFFFFFFull);
                nl++;
            }
            if (empty_here) nl = 0;               // a required token without postings on this shard: the AND finds nothing here (zero hits below), whatever the others hold
            q.n_required = nl;
            for (uint32_t t = 0; t < in.n_dropped && multi; t++) {        // after the query's own tokens, in their order (:5271-5290)
                bool found = false;
                for (uint32_t f = 0; f < in.n_fields; f++) {
                    const uint32_t handle = snap.find_handle(in.field_ids[f], in.dropped_term_ids[t]);
                    if (handle == 0xFFFFFFFFu) continue;
                    mfq.list[nl][f] = handle;
                    found = true;
                    A.list_bytes += 4ull * snap.h_lists[handle].n_ids;
                }
                if (!found) continue;                                     // an or_iterator without lists: skip_to() is false for every document
                len_of[nl] = 0xFFFFFFFFu;
                nl++;
            }
            q.n_lists = nl;
            q.match_type = in.match_type;
            q.prio_exact = in.prioritize_exact_match ? 1 : 0;
            q.prio_pos = in.prioritize_token_position ? 1 : 0;
            q.prio_nfields = in.prioritize_num_matching_fields ? 1 : 0;
            q.total_cost = in.total_cost;
            q.weight = in.field_weights[0];
            q.syn_orig_num_tokens = (int8_t)((int)in.syn_orig_num_tokens_p1 - 1);
            q.orig_num_tokens = in.orig_num_tokens; q.is_synonym = in.is_synonym_query ? 1 : 0; q.demote_synonym = in.demote_synonym_match ? 1 : 0;
            q.n_sort = (uint8_t)in.n_sort;
            if (in.n_sort > 2) A.any_s2 = true;
            for (uint32_t s = 0; s < in.n_sort; s++) {
                q.sort_kind[s] = in.sort[s].kind; q.sort_order[s] = in.sort[s].order; q.sort_col[s] = in.sort[s].column;
                if (in.sort[s].kind == TSGPU_SORT_INT64_COLUMN) A.n_numeric_sort_q++;
            }
            q.k = k;
            A.max_k = std::max(A.max_k, k);
            q.aux_off = (uint32_t)A.aux.size();
            q.n_excl = in.n_excluded;
            q.n_filt = in.n_filter;
            if (in.n_excluded || in.n_filter) A.any_aux = true;
            if (in.n_excluded) {
                if (!in.excluded_ids) { P.status[i] = TSGPU_ERR_INVALID; continue; }
                A.aux.insert(A.aux.end(), in.excluded_ids, in.excluded_ids + in.n_excluded);
            }
            if (in.n_filter) A.aux.insert(A.aux.end(), in.filter_ids, in.filter_ids + in.n_filter);   // sorted ascending, unique (filter_result_t::docs)
            if (q.n_required == 0) continue;   // no token in the index: zero hits (intersect case 0, or_iterator.h:67-68)
            if (multi) {
                // driver = the token with the fewest postings over all fields; one group of work items per field list of it
                uint32_t td = 0;
                for (uint32_t t = 1; t < q.n_required; t++) if (len_of[t] < len_of[td]) td = t;
                mfq.n_fields = in.n_fields;
                A.mf_max_fields = std::max(A.mf_max_fields, (uint32_t)in.n_fields);
                mfq.driver_token = td;
                mfq.second_token = KW_NONE;             // the required token with the next fewest postings: merged block-wise by the find kernel
                for (uint32_t t = 0; t < q.n_required; t++) if (t != td && (mfq.second_token == KW_NONE || len_of[t] < len_of[mfq.second_token])) mfq.second_token = t;
                for (uint32_t f = 0; f < (uint32_t)KW_MAX_FIELDS; f++) { mfq.is_array[f] = f < in.n_fields && snap.field_is_array.at(in.field_ids[f]) ? 1 : 0; if (mfq.is_array[f]) A.any_array = true; }
                for (uint32_t f = 0; f < (uint32_t)KW_MAX_FIELDS; f++) mfq.weight[f] = f < in.n_fields ? in.field_weights[f] : 0;
                if (k + KW_THREADS > 1024) { unsupported("topster_size with several query_by fields"); continue; }
                q.mf_index = (uint32_t)A.mf.size();
                A.mf.push_back(mfq);
                if (ordered_count) A.ordered_count_q.push_back(i);
                if (in.n_filter) { q.fbits_off = A.fbits_words; A.fbits_words += ((uint64_t)in.n_filter + 31) / 32; }
                q.ids_out_off = A.ids_total;
                uint64_t seg = 0;
                for (uint32_t f = 0; f < in.n_fields; f++) {
                    if (mfq.list[td][f] == KW_NONE) continue;
                    const ListDesc& dF = snap.h_lists[mfq.list[td][f]];
                    for (uint32_t b = 0; b < dF.n_blocks; b += KW_CHUNK_BLOCKS) {
                        KwWorkItem w;
                        w.query = i | (f << 28);
                        w.blk_begin = b;
                        w.blk_end = std::min(dF.n_blocks, b + KW_CHUNK_BLOCKS);
                        w.ids_out_off = (uint32_t)seg;
                        seg += (uint64_t)(w.blk_end - w.blk_begin) * BLOCK_IDS;
                        { if (q_cnt[i] == 0) q_begin[i] = (uint32_t)A.flat_work.size(); A.flat_work.push_back(w); q_cnt[i]++; }
                    }
                }
                if (keep_ids) A.ids_total += seg;
                continue;
            }
            // probe order: ascending list length, stable
            uint8_t ord[KW_MAX_TOKENS];
            for (uint32_t t = 0; t < nl; t++) ord[t] = (uint8_t)t;
            std::stable_sort(ord, ord + nl, [&](uint8_t a, uint8_t b) { return len_of[a] < len_of[b]; });
            for (uint32_t t = 0; t < nl; t++) q.probe_order[t] = ord[t];
            const ListDesc& dA = snap.h_lists[q.list[ord[0]]];
            q.ids_out_off = A.ids_total;
            if (keep_ids) A.ids_total += (uint64_t)dA.n_blocks * BLOCK_IDS;
            // at most 8..64 partial top-K lists per query: kw_merge_kernel folds a query's partials one after the other, and a small
            // batch (auto chunk 16) would otherwise cut a long driver list into hundreds of work items
            // (small batches: a few thousand work items fill the chip, more only lengthen the per-query merge chain; measured on the
            // 10M-doc collection: 100 queries 1.47 -> 1.15 ms, while a cap of 8 at 1 000+ queries unbalances the search kernel)
            uint32_t chunk_q = KW_CHUNK_BLOCKS;
            // small batches are as slow as their heaviest query: its longest work item (~3 us per driver block when the chip is not full)
            // plus the chain of partial folds in kw_merge_kernel (~4.5 us each) -> the item count that balances the two, ~sqrt(blocks / 1.5)
            uint32_t max_partials = ctx->kw_max_partials;
            // (with the two-level merge a chain of P folds costs G + P / G, G = 8: the balance moves to ~sqrt(2.7 x blocks) items)
            if (n_queries < 512) max_partials = std::max(max_partials, std::min<uint32_t>(384, (uint32_t)std::sqrt((double)dA.n_blocks * 2.7)));
            if (ctx->kw_merge_select_min && n_queries <= 128) max_partials = (uint32_t)KW_SEL_PMAX;        // (the selecting merge: see the chunk rule above)
            // ... but never longer than 256 blocks (only the batch-wide chunk of a very large batch goes beyond, up to KW_MAX_CHUNK): the batch is as slow as its longest work item (a 16K-block driver list cut in 16
            // would run 1 000 blocks in sequence), and folding 64 sorted partials costs kw_merge_kernel ~0.3 ms
            if (ctx->kw_chunk_blocks == 0) chunk_q = std::max(chunk_q, std::min<uint32_t>((dA.n_blocks + max_partials - 1) / max_partials, 256u));
            {   // launch-order key: estimated cost of the query's LARGEST work item = driver blocks x (fixed cost + second-list ids per
                // driver id); the work table is laid out heaviest first so that the long items do not start last (tail of the launch)
                const double r = nl >= 2 ? (double)len_of[ord[1]] / (double)std::max<uint32_t>(len_of[ord[0]], 1) : 0.0;
                // + third-list probes: every stage-1 survivor (256 |B| / N per driver block) costs a two-level global binary search
                const double surv = nl >= 3 ? 256.0 * (double)len_of[ord[1]] / (double)std::max<uint32_t>(ctx->num_docs, 1) : 0.0;
                item_cost[i] = (double)std::min(chunk_q, dA.n_blocks) * ((double)ctx->kw_cost_fixed + 0.1 * ctx->kw_cost_r_x10 * std::min(r, 64.0) + 0.01 * ctx->kw_cost_probe_x100 * surv);
            }
            for (uint32_t b = 0; b < dA.n_blocks; b += chunk_q) {
                KwWorkItem w;
                w.query = i;
                w.blk_begin = b;
                w.blk_end = std::min(dA.n_blocks, b + chunk_q);
                w.ids_out_off = b * BLOCK_IDS;
                { if (q_cnt[i] == 0) q_begin[i] = (uint32_t)A.flat_work.size(); A.flat_work.push_back(w); q_cnt[i]++; }
            }
        }
    };
    std::vector<KwWorkItem> flat_work;                       // every query's items, contiguous, in query order
    {
        const uint32_t min_par = ctx->plan_parallel_min_queries;
        const uint32_t min_slice = std::max<uint32_t>(8, min_par / 8);                      // (256 queries per slice at the default threshold)
        const uint32_t n_thr = (min_par && n_queries >= min_par) ? std::min<uint32_t>((uint32_t)std::max(1, ctx->plan_threads), std::max<uint32_t>(1, n_queries / min_slice)) : 1;
        // (four slices per thread, handed out dynamically: a parked thread that wakes late still finds work, the caller never idles)
        const uint32_t n_parts = n_thr == 1 ? 1 : std::min<uint32_t>(4 * n_thr, std::max<uint32_t>(1, n_queries / std::max<uint32_t>(8, min_slice / 4)));
        std::vector<PlanAcc> parts(n_parts);
        auto bound = [&](uint32_t k) { return (uint32_t)((uint64_t)n_queries * k / n_parts); };
        if (n_parts == 1) plan_range(0, n_queries, parts[0]);
        else {
            std::atomic<uint32_t> next{0};
            std::atomic<int> oom{0};
            const std::function<void()> job = [&]() {
                try { for (;;) { const uint32_t k = next.fetch_add(1); if (k >= n_parts) break; plan_range(bound(k), bound(k + 1), parts[k]); } } catch (const std::bad_alloc&) { oom = 1; }
            };
            ctx->host_pool.run(job, (int)n_thr - 1);
            if (oom) throw std::bad_alloc();
        }
        for (uint32_t k = 0; k < n_parts; k++) {
            PlanAcc& A = parts[k];
            const uint32_t aux_base = (uint32_t)P.aux.size(), mf_base = (uint32_t)P.mf.size(), work_base = (uint32_t)flat_work.size();
            const uint64_t ids_base = P.ids_total, fbits_base = P.fbits_words;
            if (k) {
                for (uint32_t i = bound(k); i < bound(k + 1); i++) {
                    KwQueryDev& q = P.q[i];
                    q.aux_off += aux_base; q.ids_out_off += ids_base; q.fbits_off += fbits_base;
                    if (q.mf_index != KW_NONE && P.status[i] == TSGPU_OK && !q.wild_n_ids) q.mf_index += mf_base;
                    q_begin[i] += work_base;
                }
            }
            if (k == 0) { P.aux.swap(A.aux); P.mf.swap(A.mf); flat_work.swap(A.flat_work); }
            else {
                P.aux.insert(P.aux.end(), A.aux.begin(), A.aux.end());
                P.mf.insert(P.mf.end(), A.mf.begin(), A.mf.end());
                flat_work.insert(flat_work.end(), A.flat_work.begin(), A.flat_work.end());
            }
            P.ordered_count_q.insert(P.ordered_count_q.end(), A.ordered_count_q.begin(), A.ordered_count_q.end());
            P.ids_total += A.ids_total; P.fbits_words += A.fbits_words; P.list_bytes += A.list_bytes;
            P.max_k = std::max(P.max_k, A.max_k); P.n_numeric_sort_q += A.n_numeric_sort_q;
            P.any_deadline = P.any_deadline || A.any_deadline; P.any_s2 = P.any_s2 || A.any_s2; P.any_aux = P.any_aux || A.any_aux; P.any_array = P.any_array || A.any_array; P.mf_max_fields = std::max(P.mf_max_fields, A.mf_max_fields);
        }
    }
    // work tables: one launch per kernel flavour (single-field T<=3, single-field generic, multi-field T<=3, multi-field generic);
    // a query's items stay contiguous and first_work indexes the concatenation of the four tables
    const uint64_t tp1 = now_us();
    std::vector<uint32_t> by_cost(n_queries);
    for (uint32_t i = 0; i < n_queries; i++) by_cost[i] = i;
    if (ctx->kw_sort_work) {
        // descending cost, ties in query order: LSD radix sort of the float bits (costs are non-negative, so the bits order like the
        // values), two stable counting passes of 16 bits (std::sort of 10 000 keys was 0.33 ms of a 0.95 ms plan; a coarser one-pass
        // bucket order measurably lengthened the find kernel's tail)
        // (the two 65 536-entry histograms are a FIXED ~30 us: a small round — the 1-query calling convention — sorts by comparison)
        std::vector<uint32_t> key(n_queries);
        for (uint32_t i = 0; i < n_queries; i++) { const float c = (float)item_cost[i]; uint32_t bits; memcpy(&bits, &c, 4); key[i] = 0xFFFFFFFFu - bits; }
        const bool radix = n_queries >= 2048;
        if (!radix && n_queries > 1) std::stable_sort(by_cost.begin(), by_cost.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
        std::vector<uint32_t> tmp(radix ? n_queries : 0), hist(radix ? 65537 : 0);
        for (int pass = 0; radix && pass < 2; pass++) {
            const int sh = pass * 16;
            std::fill(hist.begin(), hist.end(), 0u);
            for (uint32_t i = 0; i < n_queries; i++) hist[((key[by_cost[i]] >> sh) & 0xFFFFu) + 1]++;
            for (uint32_t k = 0; k < 65536; k++) hist[k + 1] += hist[k];
            for (uint32_t i = 0; i < n_queries; i++) { const uint32_t q = by_cost[i]; tmp[hist[(key[q] >> sh) & 0xFFFFu]++] = q; }
            by_cost.swap(tmp);
        }
    }
    const uint64_t tp2 = now_us();
    {
        std::vector<KwWorkItem>* tabs[5] = {&P.work_small, &P.work_big, &P.work_mf_small, &P.work_mf_big, &P.work_wild};
        auto flavour_of = [&](uint32_t i) { return P.q[i].wild_n_ids ? 4 : (P.q[i].mf_index != KW_NONE ? 2 : 0) + (P.q[i].n_lists <= 3 ? 0 : 1); };
        size_t total[5] = {0, 0, 0, 0, 0}, base[5];
        for (uint32_t i = 0; i < n_queries; i++) if (q_cnt[i]) total[flavour_of(i)] += q_cnt[i];
        size_t acc = 0;
        for (int k = 0; k < 5; k++) { base[k] = acc; acc += total[k]; tabs[k]->reserve(total[k]); }
        for (uint32_t oi = 0; oi < n_queries; oi++) {              // heaviest first inside every table
            const uint32_t i = by_cost[oi];
            if (q_cnt[i] == 0) continue;
            auto& dst = *tabs[flavour_of(i)];
            P.q[i].first_work = (uint32_t)(base[flavour_of(i)] + dst.size());
            P.q[i].n_work = q_cnt[i];
            dst.insert(dst.end(), flat_work.begin() + q_begin[i], flat_work.begin() + q_begin[i] + q_cnt[i]);
        }
        // merge sources: the work items' lists, or — more than 2 x 8 of them — group lists of 8 folded in parallel first (slots behind
        // the work items' own)
        const uint32_t G = 8;
        uint32_t slot = (uint32_t)acc;
        for (uint32_t i = 0; i < n_queries; i++) {
            KwQueryDev& q = P.q[i];
            q.m_first = q.first_work; q.m_n = q.n_work;
            if (q.n_work <= 2 * G) continue;
            // many work items: kw_merge_kernel SELECTS the top k from their lists (kw_select_partials; cost independent of their number) —
            // no groups; beyond its capacity the lists are folded in two levels as before
            if (ctx->kw_merge_select_min && q.n_work >= ctx->kw_merge_select_min && q.n_work <= (uint32_t)KW_SEL_PMAX) continue;
            q.m_first = slot;
            q.m_n = (q.n_work + G - 1) / G;
            for (uint32_t a = 0; a < q.n_work; a += G) P.groups.push_back({i, q.first_work + a, std::min(G, q.n_work - a), slot++});
        }
    }
    if (plan_timing) fprintf(stderr, "[tsgpu] plan: per-query loop %llu us, cost sort %llu us, table layout %llu us\n", (unsigned long long)(tp1 - tp0),
                             (unsigned long long)(tp2 - tp1), (unsigned long long)(now_us() - tp2));
    return TSGPU_OK;
}

// ---- the same plan made ON THE DEVICE (kw_plan.hip.h) for batches of plain single-field queries ----
namespace {
struct DevPlan {
    bool on = false;
    const KwQueryDev* dq = nullptr; const KwWorkItem* dw = nullptr; const uint32_t* daux = nullptr; const uint64_t* hoff = nullptr;
    uint32_t n_work[2] = {0, 0};
    uint64_t hit_blocks[2] = {0, 0};
};
}
// returns TSGPU_OK (DP.on tells whether the device made the plan; false = this batch needs the host planner) or an error
static int plan_batch_device(tsgpu_ctx* ctx, KwLane& L, const Snapshot& snap, const tsgpu_kw_query* queries, uint32_t n_queries, Plan& P, DevPlan& DP, hipStream_t s) {
    DP.on = false;
    if (!snap.maps || !snap.maps->d_dense.p || !snap.lists.p) return TSGPU_OK;
    // host pre-scan: eligibility + the 76-byte record per query, straight into the pinned staging buffer
    int rc;
    if ((rc = L.h_plan.reserve((size_t)n_queries * sizeof(KwPlanIn) + 64))) return rc;
    KwPlanIn* hin = (KwPlanIn*)L.h_plan.p;
    const uint32_t n_columns = (uint32_t)ctx->columns.size();
    std::atomic<int> bad{0};
    auto scan = [&](uint32_t lo, uint32_t hi) {
        for (uint32_t i = lo; i < hi; i++) {
            const tsgpu_kw_query& in = queries[i];
            bool ok = in.n_fields == 1 && in.n_tokens >= 1 && in.n_tokens <= TSGPU_MAX_QUERY_TOKENS && in.n_dropped == 0 && in.n_filter == 0 && in.n_excluded == 0 &&
                      in.deadline_us == 0 && in.n_sort <= TSGPU_MAX_SORT_KEYS && in.match_type <= TSGPU_SUM_SCORE && in.field_ids[0] < 64;
            if (ok) { auto it = snap.field_is_array.find(in.field_ids[0]); ok = it != snap.field_is_array.end() && !it->second; }
            for (uint32_t k = 0; ok && k < in.n_sort; k++)
                ok = in.sort[k].kind <= TSGPU_SORT_INT64_COLUMN && (in.sort[k].kind != TSGPU_SORT_INT64_COLUMN || in.sort[k].column < n_columns) && (in.sort[k].order == 1 || in.sort[k].order == -1);
            const uint32_t k = ok ? resolve_topster_size(ctx, in) : 0;
            if (!ok || k > TSGPU_MAX_TOPK) { bad.store(1); return; }
            KwPlanIn& r = hin[i];
            for (uint32_t t = 0; t < (uint32_t)TSGPU_MAX_QUERY_TOKENS; t++) r.term_ids[t] = t < in.n_tokens ? in.term_ids[t] : 0;
            r.field = in.field_ids[0]; r.weight = in.field_weights[0]; r.k = k; r.total_cost = in.total_cost;
            r.n_tokens = (uint8_t)in.n_tokens; r.match_type = in.match_type;
            r.prio_bits = (uint8_t)((in.prioritize_exact_match ? 1 : 0) | (in.prioritize_token_position ? 2 : 0) | (in.prioritize_num_matching_fields ? 4 : 0));
            r.n_sort = (uint8_t)in.n_sort;
            for (uint32_t k2 = 0; k2 < 3; k2++) { r.sort_kind[k2] = k2 < in.n_sort ? in.sort[k2].kind : 0; r.sort_order[k2] = k2 < in.n_sort ? in.sort[k2].order : 0; r.sort_col[k2] = k2 < in.n_sort ? in.sort[k2].column : 0; }
            r.syn_orig_num_tokens = (int8_t)((int)in.syn_orig_num_tokens_p1 - 1); r.orig_num_tokens = in.orig_num_tokens;
            r.is_synonym = in.is_synonym_query ? 1 : 0; r.demote_synonym = in.demote_synonym_match ? 1 : 0;
        }
    };
    {
        const uint32_t min_par = ctx->plan_parallel_min_queries;
        const uint32_t n_thr = (min_par && n_queries >= min_par) ? std::min<uint32_t>((uint32_t)std::max(1, ctx->plan_threads), std::max<uint32_t>(1, n_queries / 512)) : 1;
        if (n_thr <= 1) scan(0, n_queries);
        else {
            const uint32_t n_parts = 4 * n_thr;
            std::atomic<uint32_t> next{0};
            const std::function<void()> job = [&]() { for (;;) { const uint32_t k = next.fetch_add(1); if (k >= n_parts || bad.load()) break; scan((uint32_t)((uint64_t)n_queries * k / n_parts), (uint32_t)((uint64_t)n_queries * (k + 1) / n_parts)); } };
            ctx->host_pool.run(job, (int)n_thr - 1);
        }
    }
    if (bad.load()) return TSGPU_OK;
    // device buffers: [KwQueryDev x n | one aux word | totals | six scratch arrays] in d_plan, the input records in d_plan_in
    size_t bytes = 0;
    auto place = [&](size_t b) { const size_t at = (bytes + 63) & ~(size_t)63; bytes = at + b; return at; };
    const size_t at_q = place((size_t)n_queries * sizeof(KwQueryDev)), at_aux = place(64), at_tot = place(sizeof(KwPlanTotals)), at_nb = place((size_t)n_queries * 4),
                 at_la = place((size_t)n_queries * 4), at_lb = place((size_t)n_queries * 4), at_cnt = place((size_t)n_queries * 4), at_ch = place((size_t)n_queries * 4),
                 at_key = place((size_t)n_queries * 8), at_fw = place((size_t)n_queries * 4), at_hb = place((size_t)n_queries * 8);
    if ((rc = L.d_plan.reserve(bytes + 64)) || (rc = L.d_plan_in.reserve((size_t)n_queries * sizeof(KwPlanIn))) || (rc = L.h_plan_tot.reserve(2 * sizeof(KwPlanTotals)))) return rc;
    uint8_t* const dp = (uint8_t*)L.d_plan.p;
    KwPlanTotals* const d_tot = (KwPlanTotals*)(dp + at_tot);
    KwPlanTotals* const h_tot = (KwPlanTotals*)L.h_plan_tot.p;
    TSGPU_HIP_TRY(hipMemcpyAsync(L.d_plan_in.p, hin, (size_t)n_queries * sizeof(KwPlanIn), hipMemcpyHostToDevice, s));
    TSGPU_HIP_TRY(hipMemsetAsync(dp + at_aux, 0, 64 + sizeof(KwPlanTotals) + 64, s));
    KwPlanParams pp;
    pp.n_queries = n_queries; pp.num_docs = ctx->num_docs; pp.n_columns = n_columns;
    pp.chunk_blocks_opt = ctx->kw_chunk_blocks; pp.max_partials = std::max<uint32_t>(ctx->kw_max_partials, 1); pp.merge_select_min = ctx->kw_merge_select_min; pp.max_chunk = (uint32_t)KW_MAX_CHUNK;
    pp.cost_fixed = (float)ctx->kw_cost_fixed; pp.cost_r = 0.1f * ctx->kw_cost_r_x10; pp.cost_probe = 0.01f * ctx->kw_cost_probe_x100;
    pp.dense = snap.maps->d_dense.as<uint32_t>(); pp.lists = snap.lists.as<ListDesc>();
    KwPlanScratch sc;
    sc.n_blocks = (uint32_t*)(dp + at_nb); sc.len_a = (uint32_t*)(dp + at_la); sc.len_b = (uint32_t*)(dp + at_lb); sc.cnt = (uint32_t*)(dp + at_cnt); sc.chunk = (uint32_t*)(dp + at_ch);
    sc.key = (unsigned long long*)(dp + at_key);
    KwQueryDev* const dq = (KwQueryDev*)(dp + at_q);
    const dim3 grid((n_queries + 255) / 256), block(256);
    hipLaunchKernelGGL(kw_plan_resolve_kernel, grid, block, 0, s, pp, (const KwPlanIn*)L.d_plan_in.p, dq, sc, d_tot);
    hipLaunchKernelGGL(kw_plan_chunk_kernel, grid, block, 0, s, pp, (const KwQueryDev*)dq, sc, d_tot);
    TSGPU_HIP_TRY(hipGetLastError());
    TSGPU_HIP_TRY(hipMemcpyAsync(h_tot, d_tot, sizeof(KwPlanTotals), hipMemcpyDeviceToHost, s));
    TSGPU_HIP_TRY(hipStreamSynchronize(s));
    const KwPlanTotals t1 = *h_tot;
    if (t1.fallback) return TSGPU_OK;
    // the hit buffer must take each table in ONE group (else the host planner's grouping / the fused kernel decide)
    for (int tb = 0; tb < 2; tb++) {
        const size_t rec_bytes = (size_t)((tb == 0 ? 3 : KW_MAX_TOKENS) + 1) * 4;
        const uint64_t budget = ctx->kw_hit_buffer_records ? ctx->kw_hit_buffer_records : ((uint64_t)ctx->kw_hit_buffer_mb << 20) / rec_bytes;
        if (t1.hit_blocks[tb] * (uint64_t)BLOCK_IDS > budget) return TSGPU_OK;
    }
    const size_t n_work = (size_t)t1.n_work[0] + t1.n_work[1];
    if ((rc = L.d_plan_work.reserve(std::max<size_t>(n_work, 1) * (sizeof(KwWorkItem) + 8) + 64))) return rc;
    KwWorkItem* const dw = (KwWorkItem*)L.d_plan_work.p;
    unsigned long long* const hoff = (unsigned long long*)((uint8_t*)L.d_plan_work.p + ((std::max<size_t>(n_work, 1) * sizeof(KwWorkItem) + 63) & ~(size_t)63));
    TSGPU_HIP_TRY(hipMemsetAsync(dp + at_fw, 0, (at_hb - at_fw) + (size_t)n_queries * 8, s));
    hipLaunchKernelGGL(kw_plan_rank_kernel, dim3(grid.x, KW_PLAN_JPARTS), block, 0, s, pp, sc, (uint32_t*)(dp + at_fw), (unsigned long long*)(dp + at_hb));
    hipLaunchKernelGGL(kw_plan_emit_kernel, grid, block, 0, s, pp, dq, sc, (const uint32_t*)(dp + at_fw), (const unsigned long long*)(dp + at_hb), dw, hoff, (const KwPlanTotals*)d_tot);
    TSGPU_HIP_TRY(hipGetLastError());
    P.status.assign(n_queries, TSGPU_OK);
    P.cutoff.assign(n_queries, 0);
    P.max_k = std::max<uint32_t>(t1.max_k, 1); P.any_s2 = t1.any_s2 != 0; P.list_bytes = t1.list_bytes; P.n_numeric_sort_q = t1.n_numeric_sort_q;
    DP.on = true;
    DP.dq = dq; DP.dw = dw; DP.daux = (const uint32_t*)(dp + at_aux); DP.hoff = (const uint64_t*)hoff;
    DP.n_work[0] = t1.n_work[0]; DP.n_work[1] = t1.n_work[1]; DP.hit_blocks[0] = t1.hit_blocks[0]; DP.hit_blocks[1] = t1.hit_blocks[1];
    return TSGPU_OK;
}

// ---- lanes ----
namespace {
struct LaneLock {                                     // holds one execution lane for the duration of a batch (FIFO: LaneDispenser)
    tsgpu_ctx* ctx; KwLane* L; int index;
    explicit LaneLock(tsgpu_ctx* c, int want = -1) : ctx(c), L(nullptr), index(-1) {
        index = c->lane_dispenser.acquire(want, c->n_lanes);
        L = &c->lanes[index];
        L->mu.lock();                                 // (uncontended among LaneLock holders; tsgpu_set_stream takes it, too)
        c->last_lane.store(index);
    }
    ~LaneLock() { L->mu.unlock(); ctx->lane_dispenser.release(index, ctx->n_lanes); }
};
// Slices of a sliced host-output batch: they are ENQUEUED in slice order (turn), whichever host thread finishes planning first — the large first
// slice has to run first, so that its device-to-host copy runs under the small ones' kernels (left to the planning race, the small second slice was
// usually enqueued ahead of it: kernel timeline in profiles/r04) — and a slice's kernels start when the previous slice's have finished (last).
struct SliceChain {
    std::mutex m; std::condition_variable cv; uint32_t turn = 0; hipEvent_t last = nullptr;
    // RAII for one slice: wait_turn() before enqueueing; the destructor passes the turn on (also on every error path, after waiting for it)
    struct Turn {
        SliceChain* c; uint32_t idx; bool waited = false; std::unique_lock<std::mutex> lk;
        Turn(SliceChain* c_, uint32_t i) : c(c_), idx(i) {}
        void wait_turn() { if (!c) return; lk = std::unique_lock<std::mutex>(c->m); c->cv.wait(lk, [&] { return c->turn == idx; }); waited = true; }
        void pass() { if (!c) return; if (!waited) wait_turn(); c->turn = idx + 1; c = nullptr; lk.unlock(); }
        ~Turn() { if (c) { SliceChain* cc = c; pass(); cc->cv.notify_all(); } }
    };
};
struct BatchOpts {
    bool wildcard = false;
    bool keep_ids = false;                            // emit matched ids into the lane's id arena
    std::vector<int32_t>* status_host = nullptr;      // receives the per-query status codes
    std::vector<int32_t>* cutoff_host = nullptr;      // ... and the per-query search_cutoff flags
    tsgpu_id_lists* id_lists = nullptr;               // when set: the matched ids of every query, gathered + downloaded (implies keep_ids)
    bool record_last = true;                          // remember the id segments for the legacy tsgpu_result_ids API
    uint32_t chain_index = 0;                         // ... this slice's position in it
    SliceChain* chain = nullptr;                      // sliced host-output batch: this slice's kernels start when the previous slice's have finished (a
                                                      // device-side wait; the previous slice's device-to-host copies run meanwhile: two slices never compute at once)
    bool alias_out = false;                           // host output through the lane's pinned image: point `out`'s arrays INTO the image instead of copying
                                                      // them out (the coalesced round hands every caller its slice straight from there)
    bool timing = true;                               // record the phase events (tsgpu_timings); a coalesced round has no single caller to report to
    const KwVFlat* vflat = nullptr;                   // wildcard form ranking a distance matrix: the flat branch of the vector search (tsgpu_vector_search_batch)
    DevBuf* ids_dev = nullptr;                        // with id_lists: gather the matched ids into THIS device buffer (the caller's) and leave id_lists->ids empty — for a
    bool* ids_dev_done = nullptr;                     // consumer on the device (group_by, tsgpu_groupby.inc.h); not done (false) when a query's ids need the host's sort
    const uint16_t* present_elsewhere = nullptr;      // doc-range shard of a group whose members' dictionaries differ: per query, the tokens that exist on ANOTHER shard
                                                      // (missing here they are empty lists, not dropped tokens: kw_search_batch_masked, tsgpu_host.h); host planner
};
}
static int kw_batch_on_lane(tsgpu_ctx* ctx, KwLane& L, const tsgpu_kw_query* queries, uint32_t n_queries, tsgpu_hits* out, const BatchOpts& bo);
static int kw_dispatch(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_queries, tsgpu_hits* out, bool wildcard, tsgpu_id_lists** ids_out, DevBuf* ids_dev = nullptr, bool* ids_dev_done = nullptr,
                       const uint16_t* present_elsewhere = nullptr);
static int kw_split_host(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_queries, tsgpu_hits* out);

int tsgpu_keyword_search_batch(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_queries, tsgpu_hits* out) {
    return kw_dispatch(ctx, queries, n_queries, out, false, nullptr);
}

int tsgpu_wildcard_search_batch(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_queries, tsgpu_hits* out) {
    return kw_dispatch(ctx, queries, n_queries, out, true, nullptr);
}

int tsgpu_keyword_search_batch_ids(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_queries, tsgpu_hits* out, tsgpu_id_lists** ids_out) {
    if (!ids_out) return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_search_batch_ids: ids_out is NULL");
    *ids_out = nullptr;
    return kw_dispatch(ctx, queries, n_queries, out, false, ids_out);
}

uint64_t tsgpu_id_lists_count(const tsgpu_id_lists* l, uint32_t q) { return (l && (size_t)q + 1 < l->begin.size()) ? l->begin[q + 1] - l->begin[q] : 0; }
const uint32_t* tsgpu_id_lists_ids(const tsgpu_id_lists* l, uint32_t q) { return (l && (size_t)q + 1 < l->begin.size()) ? l->ids.data() + l->begin[q] : nullptr; }
void tsgpu_id_lists_free(tsgpu_id_lists* l) { delete l; }

}  // extern "C"

namespace tsgpu {
struct KwRequest : ParkedRequest {
    const tsgpu_kw_query* q = nullptr;
    tsgpu_hits* out = nullptr;
    tsgpu_id_lists* ids = nullptr;                   // non-null: the caller wants its matched ids
    uint64_t t_arrive = 0, t_done = 0;               // diagnostics: when it parked / when its round's results were ready
};
}

// A small keyword call from one of several concurrent request threads: parked in the combiner, executed as part of one round.
static int kw_coalesced(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_queries, tsgpu_hits* out, tsgpu_id_lists** ids_out) {
    std::unique_ptr<tsgpu_id_lists> lists;
    if (ids_out) { lists.reset(new (std::nothrow) tsgpu_id_lists); if (!lists) return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_keyword_search_batch_ids: host allocation failed"); }
    KwRequest me;
    me.units = n_queries; me.q = queries; me.out = out; me.ids = lists.get();
    me.t_arrive = now_us();
    auto amax = [](std::atomic<uint64_t>& a, uint64_t v) { uint64_t c = a.load(); while (v > c && !a.compare_exchange_weak(c, v)) {} };
    auto acquire = [&]() { return std::unique_ptr<LaneLock>(new LaneLock(ctx)); };
    const uint32_t round_cap = std::max<uint32_t>(ctx->batch_round_queries, n_queries);
    auto pick = [&](std::vector<KwRequest*>& pending, std::vector<KwRequest*>& round) {
        uint32_t units = 0;
        size_t take = 0;
        while (take < pending.size() && (take == 0 || units + pending[take]->units <= round_cap)) units += pending[take++]->units;
        round.assign(pending.begin(), pending.begin() + take);
        pending.erase(pending.begin(), pending.begin() + take);
    };
    auto exec = [&](std::vector<KwRequest*>& round, std::unique_ptr<LaneLock>& guard) {
        KwLane& L = *guard->L;
        int rc = TSGPU_OK;
        std::string err;
        try {
            uint32_t total = 0, KS = 1;
            bool want_ids = false;
            // staging stride = the widest Topster of the round (not only the widest caller buffer): a caller whose own k_stride is too
            // small for its topster_size must fail ALONE (the per-caller `n > ks` check below), not take the round's other callers with it
            for (KwRequest* r : round) {
                total += r->units; KS = std::max(KS, r->out->k_stride); want_ids = want_ids || r->ids != nullptr;
                for (uint32_t i = 0; i < r->units; i++) KS = std::max(KS, std::min<uint32_t>(resolve_topster_size(ctx, r->q[i]), TSGPU_MAX_TOPK));
            }
            L.c_q.resize(total);
            uint32_t at = 0;
            for (KwRequest* r : round) { memcpy(L.c_q.data() + at, r->q, (size_t)r->units * sizeof(tsgpu_kw_query)); at += r->units; }
            const size_t slots = (size_t)total * KS;
            auto grow = [](auto& v, size_t n) { if (v.size() < n) v.resize(n); };     // (grow-only: a shrink + regrow zero-fills the difference every round)
            grow(L.c_keys, slots); grow(L.c_scores, slots * 3); grow(L.c_tm, slots); grow(L.c_vd, slots); grow(L.c_msi, slots);
            grow(L.c_nh, total); grow(L.c_nm, total); grow(L.c_st, total); grow(L.c_co, total);
            tsgpu_hits h;
            h.mem = TSGPU_MEM_HOST; h.k_stride = KS;
            h.keys = L.c_keys.data(); h.scores = L.c_scores.data(); h.text_match = L.c_tm.data(); h.vector_distance = L.c_vd.data();
            h.match_score_index = L.c_msi.data(); h.n_hits = L.c_nh.data(); h.num_matched = L.c_nm.data(); h.status = L.c_st.data(); h.search_cutoff = L.c_co.data();
            tsgpu_id_lists all_ids;
            BatchOpts bo;
            bo.keep_ids = want_ids;
            bo.id_lists = want_ids ? &all_ids : nullptr;
            bo.record_last = false;
            bo.timing = false;
            bo.alias_out = true;                     // h's arrays may come back pointing into the lane's pinned result image
            const uint64_t te0 = now_us();
            uint64_t qsum = 0;
            for (KwRequest* r : round) { amax(ctx->kw_max_queue_us, te0 - r->t_arrive); qsum += te0 - r->t_arrive; }
            ctx->kw_queue_us.fetch_add(qsum);
            rc = kw_batch_on_lane(ctx, L, L.c_q.data(), total, &h, bo);
            const uint64_t te1 = now_us();
            ctx->batch_exec_us.fetch_add(te1 - te0);
            if (rc != TSGPU_OK) err = tls_error();
            else {
                at = 0;
                for (KwRequest* r : round) {
                    tsgpu_hits& o = *r->out;
                    const uint32_t ks = o.k_stride;
                    for (uint32_t i = 0; i < r->units; i++) {
                        const uint32_t g = at + i;
                        int32_t st = h.status[g];
                        uint32_t n = h.n_hits[g];
                        if (st == TSGPU_OK && n > ks) { st = TSGPU_ERR_INVALID; n = 0; }       // (this caller's k_stride is smaller than its topster_size)
                        o.status[i] = st;
                        o.n_hits[i] = n;
                        if (o.num_matched) o.num_matched[i] = st == TSGPU_OK ? h.num_matched[g] : 0;
                        if (o.search_cutoff) o.search_cutoff[i] = h.search_cutoff[g];
                        const size_t src = (size_t)g * KS, dst = (size_t)i * ks;
                        memcpy(o.keys + dst, h.keys + src, (size_t)n * 8);
                        memcpy(o.scores + dst * 3, h.scores + src * 3, (size_t)n * 24);
                        if (o.text_match) memcpy(o.text_match + dst, h.text_match + src, (size_t)n * 8);
                        if (o.vector_distance) memcpy(o.vector_distance + dst, h.vector_distance + src, (size_t)n * 4);
                        if (o.match_score_index) memcpy(o.match_score_index + dst, h.match_score_index + src, (size_t)n);
                    }
                    if (r->ids) {
                        r->ids->begin.resize((size_t)r->units + 1);
                        const uint64_t b0 = all_ids.begin[at];
                        for (uint32_t i = 0; i <= r->units; i++) r->ids->begin[i] = all_ids.begin[at + i] - b0;
                        r->ids->ids.assign(all_ids.ids.begin() + b0, all_ids.ids.begin() + all_ids.begin[at + r->units]);
                    }
                    at += r->units;
                }
                ctx->batch_scatter_us.fetch_add(now_us() - te1);
            }
        } catch (const std::bad_alloc&) { rc = TSGPU_ERR_NO_MEMORY; err = "tsgpu_keyword_search_batch: host allocation failed"; }
        const uint64_t td = now_us();
        for (KwRequest* r : round) { r->rc = rc; r->err = err; r->t_done = td; }
    };
    ctx->kw_comb.run(me, ctx->kw_callers, ctx->batch_window_us, acquire, pick, exec);
    if (me.t_done) { const uint64_t w = now_us() - me.t_done; amax(ctx->kw_max_wake_us, w); ctx->kw_wake_us.fetch_add(w); }
    if (me.rc != TSGPU_OK) return fail(me.rc, me.err);
    if (ids_out) *ids_out = lists.release();
    return ok();
}

extern "C" {

// Entry of every keyword / wildcard search call. Small host-output calls from concurrent request threads (the reference calls
// the seam once per query from its thread pool, src/index.cpp:3488, src/http_server.cpp:827-832) are coalesced by the
// micro-batcher into one launch; everything else takes a lane directly.
// A LARGE batch whose results go to host memory: 82-112 MB of hit arrays per 10 000 queries cross PCIe, 1.5-2 ms behind a 6.3 ms step when
// the copies start after the last kernel. Served in slices instead, each on a lane and host thread of its own: a slice computes while the
// previous one's copies run (the slices are ENQUEUED in slice order and their kernels chained by events: sharing the chip, each would take
// twice as long and they would reach their copies together). Results are those of separate calls per slice = those of one call (a query
// never influences another).
static int kw_split_host(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_queries, tsgpu_hits* out) {
    // slice sizes: the LAST slice's copies are the part nothing overlaps, and every slice costs launch tails + copy calls: a large
    // first slice (kw_host_split_first_pct of the batch), the rest in one slice (two with kw_host_split_tail_slices = 2)
    std::vector<uint32_t> start(1, 0);
    {
        uint32_t first = (uint32_t)((uint64_t)n_queries * ctx->kw_host_split_first_pct / 100);
        first = std::min(std::max(first, ctx->kw_host_split_queries), n_queries - ctx->kw_host_split_queries);
        start.push_back(first);
        // ONE tail slice (round 4, kernel timeline in profiles/r04/exp_host_delivery_r04.txt): every extra launch has its own tail of long work items —
        // three slices kept the GPU busy for 8.4 ms against 6.3 ms for one launch — and with the slices enqueued in order the large one's copy
        // hides under ONE small slice's kernels: 85 % + 15 % = 7.5 ms per 10 000 queries (70 / 15 / 15: 8.0 ms; unsliced: 8.1 ms)
        const uint32_t rest = n_queries - first, parts = ctx->kw_host_split_tail_slices >= 2 && rest >= 2 * ctx->kw_host_split_queries ? 2 : 1;
        for (uint32_t i = 1; i <= parts; i++) start.push_back(first + (uint32_t)((uint64_t)rest * i / parts));
    }
    const uint32_t n_slices = (uint32_t)start.size() - 1;
    SliceChain chain;
    std::mutex err_mu;
    std::atomic<uint32_t> next{0};
    int first_rc = TSGPU_OK;
    std::string first_err;
    const std::function<void()> job = [&]() {
        for (;;) {
            const uint32_t si = next.fetch_add(1);
            if (si >= n_slices) break;
            const uint32_t a = start[si], n = start[si + 1] - a;
            const size_t at = (size_t)a * out->k_stride;
            tsgpu_hits h = *out;
            h.keys = out->keys + at; h.scores = out->scores + at * 3; h.n_hits = out->n_hits + a; h.status = out->status + a;
            if (out->text_match) h.text_match = out->text_match + at;
            if (out->vector_distance) h.vector_distance = out->vector_distance + at;
            if (out->match_score_index) h.match_score_index = out->match_score_index + at;
            if (out->num_matched) h.num_matched = out->num_matched + a;
            if (out->search_cutoff) h.search_cutoff = out->search_cutoff + a;
            BatchOpts bo;
            bo.record_last = false;
            bo.chain = &chain;
            bo.chain_index = si;
            int rc;
            { LaneLock ll(ctx); rc = kw_batch_on_lane(ctx, *ll.L, queries + a, n, &h, bo); }
            if (rc != TSGPU_OK) { std::lock_guard<std::mutex> lk(err_mu); if (first_rc == TSGPU_OK) { first_rc = rc; first_err = tls_error(); } }
        }
    };
    // one host thread (and lane) per slice: every slice is planned at once, enqueued in order, and waits for / copies out its own results
    try { ctx->split_pool.run(job, std::min<uint32_t>(n_slices, (uint32_t)ctx->n_lanes) - 1); } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_keyword_search_batch: host allocation failed"); }
    if (first_rc != TSGPU_OK) return fail(first_rc, first_err);
    return ok();
}

static int kw_dispatch(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_queries, tsgpu_hits* out, bool wildcard, tsgpu_id_lists** ids_out, DevBuf* ids_dev, bool* ids_dev_done,
                       const uint16_t* present_elsewhere) {
    if (ids_dev_done) *ids_dev_done = false;
    if (!ctx || !out) return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_search_batch: NULL argument");
    if (n_queries == 0) { if (ids_out) { *ids_out = new (std::nothrow) tsgpu_id_lists; if (*ids_out) (*ids_out)->begin.assign(1, 0); } return ok(); }
    if (!queries) return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_search_batch: queries is NULL");
    if (!out->keys || !out->scores || !out->n_hits || !out->status) return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_search_batch: missing output arrays");
    struct CallerCount { std::atomic<int>& c; explicit CallerCount(std::atomic<int>& x) : c(x) { c.fetch_add(1); } ~CallerCount() { c.fetch_sub(1); } } cc(ctx->kw_callers);
    const bool legacy_keep = ctx->keep_ids;
    if (!wildcard && !legacy_keep && !ids_dev && !present_elsewhere && !tsgpu::tls_no_coalesce() && out->mem == TSGPU_MEM_HOST && n_queries <= ctx->batch_max_queries && ctx->kw_callers.load() > 1)
        return kw_coalesced(ctx, queries, n_queries, out, ids_out);
    if (!wildcard && !legacy_keep && !ids_out && !present_elsewhere && out->mem == TSGPU_MEM_HOST && ctx->kw_host_split_queries && ctx->n_lanes >= 2 &&
        (uint64_t)n_queries >= 4ull * ctx->kw_host_split_queries)
        return kw_split_host(ctx, queries, n_queries, out);
    std::unique_ptr<tsgpu_id_lists> lists;
    if (ids_out) { lists.reset(new (std::nothrow) tsgpu_id_lists); if (!lists) return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_keyword_search_batch_ids: host allocation failed"); }
    BatchOpts bo;
    bo.wildcard = wildcard;
    bo.keep_ids = legacy_keep || ids_out != nullptr;
    bo.id_lists = lists.get();
    bo.ids_dev = lists ? ids_dev : nullptr; bo.ids_dev_done = ids_dev_done;
    bo.record_last = legacy_keep;
    bo.present_elsewhere = present_elsewhere;
    LaneLock ll(ctx, legacy_keep ? 0 : (tsgpu::tls_avoid_lane0() ? -2 : -1));          // the legacy "last batch" id API is single-caller: always lane 0
    const int rc = kw_batch_on_lane(ctx, *ll.L, queries, n_queries, out, bo);
    if (rc == TSGPU_OK && ids_out) *ids_out = lists.release();
    return rc;
}

// Waits for everything enqueued on the lane's stream WITHOUT burning a CPU: hipEventSynchronize on a hipEventBlockingSync event still
// spins inside the HSA runtime for most of a small round (19 % of the process's CPU time under 256 callers, profiles/r03/
// sigprof_256threads_before.txt), and with many request threads on a CPU quota that time is the throughput. Sleep for most of the
// lane's usual wait, then poll the event between short sleeps (timer slack lowered for the sleeps: the default 50 us would triple them).
static hipError_t sleeping_wait(KwLane& L, hipStream_t s) {
    hipError_t e = hipEventRecord(L.ev_block, s);
    if (e != hipSuccess) return e;
    const uint64_t t0 = now_us();
#if defined(__linux__)
    const int slack = prctl(PR_GET_TIMERSLACK, 0, 0, 0, 0);
    (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
    auto nap = [](uint64_t us) { timespec ts; ts.tv_sec = 0; ts.tv_nsec = (long)(us * 1000); (void)nanosleep(&ts, nullptr); };
    if (L.wait_ema_us > 40) nap(std::min<uint64_t>(L.wait_ema_us * 6 / 10, 2000));
    while ((e = hipEventQuery(L.ev_block)) == hipErrorNotReady) nap(15);
    if (slack > 0) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)slack, 0, 0, 0);
#else
    e = hipEventSynchronize(L.ev_block);
#endif
    const uint64_t w = now_us() - t0;
    L.wait_ema_us = (L.wait_ema_us * 7 + w) / 8;
    return e;
}

// the lane's mutex is held by the caller
static int kw_batch_on_lane(tsgpu_ctx* ctx, KwLane& L, const tsgpu_kw_query* queries, uint32_t n_queries, tsgpu_hits* out, const BatchOpts& bo) {
    (void)hipSetDevice(ctx->device);
    const std::shared_ptr<const Snapshot> snap_ref = ctx->snapshot();     // this batch runs on this snapshot, whatever is committed meanwhile
    const Snapshot& snap = *snap_ref;
    const bool wildcard = bo.wildcard;
    const bool keep_ids = bo.keep_ids || bo.id_lists != nullptr;
    std::vector<int32_t>* status_host = bo.status_host;
    hipStream_t s = L.stream;
    SliceChain::Turn chain_turn(bo.chain, bo.chain_index);     // (passes the turn on when this slice leaves, whatever happens to it)
    try {
        static const bool host_timing = getenv("TSGPU_HOST_TIMING") != nullptr;      // diagnostics: host phases of the call on stderr
        const uint64_t t_enter = now_us();
        Plan P;
        DevPlan DP;
        int rc;
        // big batches of plain single-field queries: the plan is made on the device (three small kernels, one read-back) instead of ~0.5 ms of
        // host threads; any other shape — and anything the device planner hands back — goes through plan_batch()
        // (not for the chained slices of a sliced host delivery: there the host plans slice i + 1 WHILE slice i runs, and a planning kernel on the
        //  second lane would wait behind the running find kernel for a place on the chip — measured: 10.05 -> 10.18 ms per 10 000 queries)
        // (batches that keep the matched ids: only when nobody reads the segments back per query — the candidate call marks its id sets from the device tables)
        if (!wildcard && !bo.present_elsewhere && (!keep_ids || (!bo.record_last && !bo.id_lists)) && !bo.vflat && (!bo.chain || (bo.chain_index == 0 && ctx->kw_host_split_device_plan)) && ctx->kw_two_kernels && ctx->kw_device_plan_min_queries && n_queries >= ctx->kw_device_plan_min_queries) {
            if ((rc = plan_batch_device(ctx, L, snap, queries, n_queries, P, DP, s))) return rc;
            if (DP.on) ctx->kw_device_plans.fetch_add(1); else ctx->kw_device_plan_fallbacks.fetch_add(1);
            if (DP.on && keep_ids) P.ids_total = (DP.hit_blocks[0] + DP.hit_blocks[1]) * (uint64_t)BLOCK_IDS;
        }
        if (!DP.on && (rc = plan_batch(ctx, snap, queries, n_queries, P, keep_ids, wildcard, bo.vflat, bo.present_elsewhere))) return rc;
        const uint64_t t_planned = now_us();
        if (out->k_stride < P.max_k) return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_search_batch: k_stride smaller than the largest topster_size");
        const uint32_t n_work = DP.on ? DP.n_work[0] + DP.n_work[1]
                                      : (uint32_t)(P.work_small.size() + P.work_big.size() + P.work_mf_small.size() + P.work_mf_big.size() + P.work_wild.size());
        const uint32_t KS = out->k_stride;
        const int cap = P.max_k + KW_THREADS <= 512 ? 512 : (P.max_k + KW_THREADS <= 1024 ? 1024 : 2048);

        // ---- the plan travels in ONE pinned staging buffer and ONE host-to-device copy (queries, work items, aux ids, multi-field
        //      descriptors, hit-record offsets): a pageable source costs a staged synchronous copy per call, five calls per batch ----
        std::vector<KwWorkItem> work(P.work_small);
        work.insert(work.end(), P.work_big.begin(), P.work_big.end());
        work.insert(work.end(), P.work_mf_small.begin(), P.work_mf_small.end());
        work.insert(work.end(), P.work_mf_big.begin(), P.work_mf_big.end());
        work.insert(work.end(), P.work_wild.begin(), P.work_wild.end());
        P.aux.push_back(0);
        // single-field tables (<= 3 tokens / up to 10 tokens) and multi-field tables: find + score kernels when the hit buffer fits, else the
        // fused kernel. A work item can yield at most one hit per driver id, so its segment of the hit buffer holds (blk_end - blk_begin) * 256
        // records of 1 + TMAX words; the items run in groups whose segments fit the budget.
        struct TablePlan { bool two = false; std::vector<uint64_t> hoff; std::vector<size_t> group_start; uint64_t need = 0; size_t rec_bytes = 0; size_t hoff_at = 0; };
        uint64_t hit_records = 0;
        auto prep_table = [&](const std::vector<KwWorkItem>& tab, int TM, bool MFT) {
            TablePlan tp;
            tp.group_start.assign(1, 0);
            if (tab.empty()) return tp;
            const size_t nws = tab.size();
            tp.rec_bytes = (size_t)((MFT ? TM * KW_MAX_FIELDS : TM) + 1) * 4;
            tp.two = ctx->kw_two_kernels;
            if (tp.two) {
                uint64_t largest = 0, all = 0;
                for (size_t i = 0; i < nws; i++) { const uint64_t c = (uint64_t)(tab[i].blk_end - tab[i].blk_begin) * BLOCK_IDS; largest = std::max(largest, c); all += c; }
                hit_records += all;
                const uint64_t budget = std::max<uint64_t>(ctx->kw_hit_buffer_records ? ctx->kw_hit_buffer_records : ((uint64_t)ctx->kw_hit_buffer_mb << 20) / tp.rec_bytes, largest);
                tp.hoff.resize(nws);
                uint64_t used = 0;
                for (size_t i = 0; i < nws; i++) {
                    const uint64_t c = (uint64_t)(tab[i].blk_end - tab[i].blk_begin) * BLOCK_IDS;
                    if (used + c > budget) { tp.group_start.push_back(i); used = 0; }
                    tp.hoff[i] = used; used += c; tp.need = std::max(tp.need, used);
                }
                tp.group_start.push_back(nws);
                tp.two = tp.group_start.size() <= 3;          // each group drains the chip between its two kernels: beyond two groups the fused kernel wins
            }
            if (tp.two && L.d_hits.reserve(std::max<uint64_t>(tp.need, 1) * tp.rec_bytes)) {
                (void)hipGetLastError();                // no room for the hit buffer: the fused kernel needs none
                tp.two = false;
            }
            if (!tp.two) tp.hoff.clear();
            return tp;
        };
        TablePlan tps[4] = {prep_table(P.work_small, 3, false), prep_table(P.work_big, KW_MAX_TOKENS, false), prep_table(P.work_mf_small, 3, true),
                            prep_table(P.work_mf_big, KW_MAX_TOKENS, true)};
        size_t tab_n[5] = {P.work_small.size(), P.work_big.size(), P.work_mf_small.size(), P.work_mf_big.size(), P.work_wild.size()};
        if (DP.on) {                                    // the device planner's two tables: one group each (it checked the budget), offsets already on the device
            for (int tb = 0; tb < 2; tb++) {
                TablePlan& tp = tps[tb];
                tab_n[tb] = DP.n_work[tb];
                if (!DP.n_work[tb]) continue;
                tp.two = true;
                tp.rec_bytes = (size_t)((tb == 0 ? 3 : KW_MAX_TOKENS) + 1) * 4;
                tp.need = DP.hit_blocks[tb] * (uint64_t)BLOCK_IDS;
                tp.group_start = {0, (size_t)DP.n_work[tb]};
                hit_records += tp.need;
                if ((rc = L.d_hits.reserve(std::max<uint64_t>(tp.need, 1) * tp.rec_bytes))) return rc;
            }
        }
        // multi-field queries with filter + excluded ids: counted from the hit records of ONE find launch of their table; a table that was
        // cut into groups or runs fused cannot serve them -> 501 for those queries (their hits are not reported)
        std::vector<uint32_t> oc_jobs[2];
        for (uint32_t qi : P.ordered_count_q) {
            const int tb = P.q[qi].n_lists <= 3 ? 0 : 1;
            const TablePlan& tp = tps[2 + tb];
            if (tp.two && tp.group_start.size() == 2) oc_jobs[tb].push_back(qi);
            else if (P.status[qi] == TSGPU_OK) P.status[qi] = TSGPU_ERR_UNSUPPORTED;
        }
        size_t plan_bytes = 0;
        auto place = [&](size_t bytes) { const size_t at = (plan_bytes + 63) & ~(size_t)63; plan_bytes = at + bytes; return at; };
        const size_t at_q = place(P.q.size() * sizeof(KwQueryDev)), at_w = place(work.size() * sizeof(KwWorkItem)), at_aux = place(P.aux.size() * 4),
                     at_mf = place(P.mf.size() * sizeof(KwQueryMF));
        for (auto& tp : tps) tp.hoff_at = place(tp.hoff.size() * 8);
        const size_t at_grp = place(P.groups.size() * sizeof(KwMergeGroup));
        const size_t at_oc[2] = {place(oc_jobs[0].size() * 4), place(oc_jobs[1].size() * 4)};
        if (!DP.on && ((rc = L.h_plan.reserve(plan_bytes + 64)) || (rc = L.d_plan.reserve(plan_bytes + 64)))) return rc;
        if (!DP.on) {
            uint8_t* hp = (uint8_t*)L.h_plan.p;
            memcpy(hp + at_q, P.q.data(), P.q.size() * sizeof(KwQueryDev));
            memcpy(hp + at_w, work.data(), work.size() * sizeof(KwWorkItem));
            memcpy(hp + at_aux, P.aux.data(), P.aux.size() * 4);
            if (!P.mf.empty()) memcpy(hp + at_mf, P.mf.data(), P.mf.size() * sizeof(KwQueryMF));
            for (auto& tp : tps) if (!tp.hoff.empty()) memcpy(hp + tp.hoff_at, tp.hoff.data(), tp.hoff.size() * 8);
            if (!P.groups.empty()) memcpy(hp + at_grp, P.groups.data(), P.groups.size() * sizeof(KwMergeGroup));
            for (int tb = 0; tb < 2; tb++) if (!oc_jobs[tb].empty()) memcpy(hp + at_oc[tb], oc_jobs[tb].data(), oc_jobs[tb].size() * 4);
            TSGPU_HIP_TRY(hipMemcpyAsync(L.d_plan.p, hp, plan_bytes, hipMemcpyHostToDevice, s));
        }
        uint8_t* const dplan = (uint8_t*)L.d_plan.p;

        // ---- scratch ----
        const size_t pw = (size_t)std::max<uint32_t>(n_work, 1) + P.groups.size();      // partial lists: one per work item + one per merge group
        if ((rc = L.d_part_s0.reserve(pw * KS * 8))) return rc;
        if ((rc = L.d_part_s1.reserve(pw * KS * 8))) return rc;
        if ((rc = L.d_part_s2.reserve(pw * KS * 8))) return rc;
        if ((rc = L.d_part_key.reserve(pw * KS * 8))) return rc;
        if ((rc = L.d_part_cnt.reserve(pw * 4))) return rc;
        if ((rc = L.d_part_nm.reserve(pw * 4))) return rc;
        if ((rc = L.d_part_ne.reserve(pw * 4))) return rc;
        if ((rc = L.d_part_ow.reserve(pw * 8))) return rc;
        if ((rc = L.d_part_f.reserve(pw * 16))) return rc;
        if ((rc = L.d_out_ow.reserve((size_t)n_queries * 8))) return rc;
        uint32_t* ids_out = nullptr;
        if (keep_ids) {
            if ((rc = L.d_ids_out.reserve(std::max<uint64_t>(P.ids_total, 1) * 4))) return rc;
            ids_out = L.d_ids_out.as<uint32_t>();
        }
        KwPartials part;
        part.s0 = L.d_part_s0.as<int64_t>(); part.s1 = L.d_part_s1.as<int64_t>(); part.s2 = L.d_part_s2.as<int64_t>();
        part.key = L.d_part_key.as<int64_t>(); part.cnt = L.d_part_cnt.as<uint32_t>(); part.n_match = L.d_part_nm.as<uint32_t>();
        part.n_emit = L.d_part_ne.as<uint32_t>(); part.off_words = L.d_part_ow.as<uint64_t>(); part.k_stride = KS;
        part.n_match1 = L.d_part_f.as<uint32_t>(); part.first_rank = part.n_match1 + pw; part.last_rank = part.first_rank + pw; part.fflags = part.last_rank + pw;

        const size_t slots = (size_t)n_queries * KS;
        KwOut o;
        o.k_stride = KS;
        o.off_words = L.d_out_ow.as<uint64_t>();
        size_t out_at[8] = {0, 0, 0, 0, 0, 0, 0, 0}, out_bytes = 0;
        const bool dev_out = out->mem == TSGPU_MEM_DEVICE;
        bool zero_copy = false;
        if (dev_out) {
            if (!out->text_match || !out->vector_distance || !out->match_score_index || !out->num_matched)
                return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_search_batch: device output needs every tsgpu_hits array");
            o.keys = out->keys; o.scores = out->scores; o.text_match = out->text_match; o.vector_distance = out->vector_distance;
            o.match_score_index = out->match_score_index; o.n_hits = out->n_hits; o.num_matched = out->num_matched;
        } else {
            // host outputs: ONE device image [n_hits | num_matched | off_words | keys | scores | text_match | vector_distance | msi]
            // -> one device-to-host copy per batch (seven separate copies cost a small batch ~60 us of launch overhead)
            const size_t nq8 = ((size_t)n_queries * 4 + 7) & ~(size_t)7;
            out_at[0] = 0; out_at[1] = nq8; out_at[2] = out_at[1] + (size_t)n_queries * 8; out_at[3] = out_at[2] + (size_t)n_queries * 8;
            out_at[4] = out_at[3] + slots * 8; out_at[5] = out_at[4] + slots * 24; out_at[6] = out_at[5] + slots * 8; out_at[7] = out_at[6] + ((slots * 4 + 7) & ~(size_t)7);
            out_bytes = out_at[7] + ((slots + 7) & ~(size_t)7);
            // a SMALL round's merge kernel writes the image straight into the lane's pinned host buffer (device-visible, coherent): no
            // device-to-host copy kernel (4-25 us) and no launch gap (~6 us) behind the merge; only the hits themselves cross the link
            zero_copy = out_bytes <= (8u << 20) && n_queries <= ctx->kw_zero_copy_max_queries;
            if (zero_copy) { if ((rc = L.h_out.reserve(out_bytes + 64))) return rc; }
            else if ((rc = L.d_out_keys.reserve(out_bytes))) return rc;
            uint8_t* ob = zero_copy ? (uint8_t*)L.h_out.p : (uint8_t*)L.d_out_keys.p;
            o.n_hits = (uint32_t*)(ob + out_at[0]); o.num_matched = (uint64_t*)(ob + out_at[1]); o.off_words = (uint64_t*)(ob + out_at[2]);
            o.keys = (uint64_t*)(ob + out_at[3]); o.scores = (int64_t*)(ob + out_at[4]); o.text_match = (int64_t*)(ob + out_at[5]);
            o.vector_distance = (float*)(ob + out_at[6]); o.match_score_index = (int8_t*)(ob + out_at[7]);
        }

        // ---- launch ----
        if (bo.chain) {
            chain_turn.wait_turn();
            if (bo.chain->last) TSGPU_HIP_TRY(hipStreamWaitEvent(s, bo.chain->last, 0));
        }
        const uint64_t t_uploaded = now_us();
        IndexView v = make_view(ctx, snap);
        v.mf = (const KwQueryMF*)(dplan + at_mf);
        if (P.fbits_words) {
            if ((rc = L.d_fbits.reserve(P.fbits_words * 4))) return rc;
            TSGPU_HIP_TRY(hipMemsetAsync(L.d_fbits.p, 0, P.fbits_words * 4, s));
        }
        v.fbits = L.d_fbits.as<uint32_t>();
        if ((rc = L.d_t0.reserve(64))) return rc;
        v.t0 = L.d_t0.as<long long>();
        v.ticks_per_us = ctx->ticks_per_us;
        if (P.any_deadline) {
            if ((rc = L.d_cut.reserve((size_t)n_queries * 4))) return rc;
            TSGPU_HIP_TRY(hipMemsetAsync(L.d_cut.p, 0, (size_t)n_queries * 4, s));
            hipLaunchKernelGGL(kw_stamp_kernel, dim3(1), dim3(1), 0, s, L.d_t0.as<long long>());       // the queries' budgets count from here
        }
        v.cutoff = L.d_cut.as<uint32_t>();
        const bool count_touched = ctx->kw_count_touched;
        if (count_touched) {                         // measurement option: the find kernel's COUNT instantiation adds the bytes it requests here
            if ((rc = L.d_touched.reserve(8 * 8))) return rc;
            TSGPU_HIP_TRY(hipMemsetAsync(L.d_touched.p, 0, 8 * 8, s));
            v.touched = L.d_touched.as<unsigned long long>();
        }
        const KwQueryDev* dq = DP.on ? DP.dq : (const KwQueryDev*)(dplan + at_q);
        const KwWorkItem* dw = DP.on ? DP.dw : (const KwWorkItem*)(dplan + at_w);
        L.last_tab_q = dq; L.last_tab_w = dw; L.last_tab_n_work = n_work;
        const uint32_t* daux = DP.on ? DP.daux : (const uint32_t*)(dplan + at_aux);
        const bool timing = bo.timing && n_queries >= ctx->kw_timing_min_queries;      // (each record is a marker packet in the stream: ~2 us of a small round)
        if (timing) TSGPU_HIP_TRY(hipEventRecord(L.ev[0], s));
        auto shifted = [&](size_t sh) {                 // every kernel indexes the partials by its own blockIdx: shift the bases
            KwPartials pb = part;
            pb.s0 += sh * KS; pb.s1 += sh * KS; pb.s2 += sh * KS; pb.key += sh * KS;
            pb.cnt += sh; pb.n_match += sh; pb.n_emit += sh; pb.off_words += sh;
            pb.n_match1 += sh; pb.first_rank += sh; pb.last_rank += sh; pb.fflags += sh;
            return pb;
        };
        uint32_t hit_groups = 0;
        bool find_marked = false;
        auto run_table = [&](size_t nws, const TablePlan& tp, size_t first, auto tmax_tag, auto mf_tag) {
            constexpr int TM = decltype(tmax_tag)::value;
            constexpr bool MFT = decltype(mf_tag)::value;
            if (nws == 0) return;
            if (tp.two) {
                hit_groups += (uint32_t)tp.group_start.size() - 1;
                const uint64_t* hoff_dev = DP.on ? DP.hoff + first : (const uint64_t*)(dplan + tp.hoff_at);
                for (size_t gi = 0; gi + 1 < tp.group_start.size(); gi++) {
                    const size_t a = tp.group_start[gi], b = tp.group_start[gi + 1];
                    if (b <= a) continue;
                    if constexpr (MFT) {
                        MfOrderedCount oc;
                        const int tb = TM == 3 ? 0 : 1;
                        if (!oc_jobs[tb].empty() && tp.group_start.size() == 2) {
                            oc.jobs = (const uint32_t*)(dplan + at_oc[tb]); oc.n_jobs = (uint32_t)oc_jobs[tb].size(); oc.table_first = (uint32_t)first;
                            oc.work_all = dw; oc.part_all = part;
                        }
                        const int mf_pipe = !ctx->kw_mf_pipelined ? 0 : (P.mf_max_fields <= 2 ? 2 : (P.mf_max_fields <= 4 ? 4 : 0));
                        if (mf_pipe) ctx->kw_mf_pipelined_launches++;
                        launch_find_score_mf<TM>(cap, s, (uint32_t)(b - a), v, dq, dw + first + a, shifted(first + a), daux, ids_out, L.d_hits.as<uint32_t>(), hoff_dev + a, oc, !P.any_aux && !P.any_array, mf_pipe);
                    }
                    else {
                        const bool mark = !find_marked;
                        find_marked = true;
                        launch_find_score<TM>(cap, s, (uint32_t)(b - a), v, dq, dw + first + a, shifted(first + a), daux, ids_out, P.any_s2, L.d_hits.as<uint32_t>(), hoff_dev + a, mark && timing ? L.ev[3] : nullptr, ctx->kw_pair_blocks, !P.any_aux);
                    }
                }
            } else if constexpr (MFT) launch_search_mf_cap<TM>(cap, s, (uint32_t)nws, v, dq, dw + first, shifted(first), daux, ids_out);
            else launch_search_cap<TM>(cap, s, (uint32_t)nws, v, dq, dw + first, shifted(first), daux, ids_out, P.any_s2);
        };
        run_table(tab_n[0], tps[0], 0, std::integral_constant<int, 3>(), std::false_type());
        size_t sh = tab_n[0];
        run_table(tab_n[1], tps[1], sh, std::integral_constant<int, KW_MAX_TOKENS>(), std::false_type());
        sh += tab_n[1];
        run_table(tab_n[2], tps[2], sh, std::integral_constant<int, 3>(), std::true_type());
        sh += tab_n[2];
        run_table(tab_n[3], tps[3], sh, std::integral_constant<int, KW_MAX_TOKENS>(), std::true_type());
        sh += tab_n[3];
        if (!P.work_wild.empty()) {
            const uint32_t nw = (uint32_t)P.work_wild.size();
            if (cap == 512) hipLaunchKernelGGL((kw_wildcard_kernel<512>), dim3(nw), dim3(KW_THREADS), 0, s, v, dq, dw + sh, shifted(sh), daux, ids_out);
            else if (cap == 1024) hipLaunchKernelGGL((kw_wildcard_kernel<1024>), dim3(nw), dim3(KW_THREADS), 0, s, v, dq, dw + sh, shifted(sh), daux, ids_out);
            else hipLaunchKernelGGL((kw_wildcard_kernel<2048>), dim3(nw), dim3(KW_THREADS), 0, s, v, dq, dw + sh, shifted(sh), daux, ids_out);
        }
        if (timing) TSGPU_HIP_TRY(hipEventRecord(L.ev[1], s));
        if (!P.groups.empty()) {
            const KwMergeGroup* dg = (const KwMergeGroup*)(dplan + at_grp);
            const uint32_t ng = (uint32_t)P.groups.size();
            if (cap == 512) hipLaunchKernelGGL((kw_merge_groups_kernel<512>), dim3(ng), dim3(KW_THREADS), 0, s, dq, part, dg);
            else if (cap == 1024) hipLaunchKernelGGL((kw_merge_groups_kernel<1024>), dim3(ng), dim3(KW_THREADS), 0, s, dq, part, dg);
            else hipLaunchKernelGGL((kw_merge_groups_kernel<2048>), dim3(ng), dim3(KW_THREADS), 0, s, dq, part, dg);
        }
        launch_merge(cap, s, n_queries, dq, part, o, ids_out, dw, ctx->kw_merge_select_min);
        if (bo.vflat) hipLaunchKernelGGL(kw_vflat_distance_kernel, dim3(n_queries), dim3(KW_THREADS), 0, s, dq, daux, o);     // KV::vector_distance of the hits
        if (timing) TSGPU_HIP_TRY(hipEventRecord(L.ev[2], s));
        TSGPU_HIP_TRY(hipGetLastError());
        if (bo.chain) { TSGPU_HIP_TRY(hipEventRecord(L.ev_chain, s)); bo.chain->last = L.ev_chain; SliceChain* cc = bo.chain; chain_turn.pass(); cc->cv.notify_all(); }
        const uint64_t t_launched = now_us();

        // ---- results ----
        std::vector<uint64_t> off_words(n_queries);
        if (!dev_out) {
            // small results: the whole image through the pinned staging buffer (a pageable destination costs ~150 us per copy call);
            // large ones: straight to the caller's arrays (the runtime pipelines them)
            const bool stage = out_bytes <= (8u << 20);
            if (stage) {
                if (!zero_copy) {
                    if ((rc = L.h_out.reserve(out_bytes + 64))) return rc;
                    TSGPU_HIP_TRY(hipMemcpyAsync(L.h_out.p, L.d_out_keys.p, out_bytes, hipMemcpyDeviceToHost, s));
                }
                // (a spinning wait costs one CPU per lane for the whole round: with many request threads — and a CPU quota — the waiting
                //  thread sleeps on a blocking event instead: +20-40 us of latency, four CPUs back)
                if (ctx->kw_callers.load() >= ctx->blocking_sync_min_callers) TSGPU_HIP_TRY(sleeping_wait(L, s));
                else TSGPU_HIP_TRY(hipStreamSynchronize(s));
                uint8_t* hb = (uint8_t*)L.h_out.p;
                const uint32_t* nh = (const uint32_t*)(hb + out_at[0]);
                memcpy(off_words.data(), hb + out_at[2], (size_t)n_queries * 8);
                if (bo.alias_out) {
                    out->n_hits = (uint32_t*)(hb + out_at[0]); out->num_matched = (uint64_t*)(hb + out_at[1]);
                    out->keys = (uint64_t*)(hb + out_at[3]); out->scores = (int64_t*)(hb + out_at[4]); out->text_match = (int64_t*)(hb + out_at[5]);
                    out->vector_distance = (float*)(hb + out_at[6]); out->match_score_index = (int8_t*)(hb + out_at[7]);
                } else {
                    memcpy(out->n_hits, nh, (size_t)n_queries * 4);
                    if (out->num_matched) memcpy(out->num_matched, hb + out_at[1], (size_t)n_queries * 8);
                    for (uint32_t i = 0; i < n_queries; i++) {           // only the hits: slots behind n_hits[i] are undefined in the image too
                        const size_t at = (size_t)i * KS, n = std::min<uint32_t>(nh[i], KS);
                        memcpy(out->keys + at, hb + out_at[3] + at * 8, n * 8);
                        memcpy(out->scores + at * 3, hb + out_at[4] + at * 24, n * 24);
                        if (out->text_match) memcpy(out->text_match + at, hb + out_at[5] + at * 8, n * 8);
                        if (out->vector_distance) memcpy(out->vector_distance + at, hb + out_at[6] + at * 4, n * 4);
                        if (out->match_score_index) memcpy(out->match_score_index + at, hb + out_at[7] + at, n);
                    }
                }
            } else {
                TSGPU_HIP_TRY(hipMemcpyAsync(out->n_hits, o.n_hits, (size_t)n_queries * 4, hipMemcpyDeviceToHost, s));
                if (out->num_matched) TSGPU_HIP_TRY(hipMemcpyAsync(out->num_matched, o.num_matched, (size_t)n_queries * 8, hipMemcpyDeviceToHost, s));
                TSGPU_HIP_TRY(hipMemcpyAsync(out->keys, o.keys, slots * 8, hipMemcpyDeviceToHost, s));
                TSGPU_HIP_TRY(hipMemcpyAsync(out->scores, o.scores, slots * 24, hipMemcpyDeviceToHost, s));
                if (out->text_match) TSGPU_HIP_TRY(hipMemcpyAsync(out->text_match, o.text_match, slots * 8, hipMemcpyDeviceToHost, s));
                if (out->vector_distance) TSGPU_HIP_TRY(hipMemcpyAsync(out->vector_distance, o.vector_distance, slots * 4, hipMemcpyDeviceToHost, s));
                if (out->match_score_index) TSGPU_HIP_TRY(hipMemcpyAsync(out->match_score_index, o.match_score_index, slots, hipMemcpyDeviceToHost, s));
                TSGPU_HIP_TRY(hipMemcpyAsync(off_words.data(), o.off_words, (size_t)n_queries * 8, hipMemcpyDeviceToHost, s));
                TSGPU_HIP_TRY(hipStreamSynchronize(s));
            }
            for (uint32_t i = 0; i < n_queries; i++) out->status[i] = P.status[i];
            if (out->search_cutoff) for (uint32_t i = 0; i < n_queries; i++) out->search_cutoff[i] = P.cutoff[i];
        } else {
            TSGPU_HIP_TRY(hipMemcpyAsync(out->status, P.status.data(), (size_t)n_queries * 4, hipMemcpyHostToDevice, s));
            if (out->search_cutoff) TSGPU_HIP_TRY(hipMemcpyAsync(out->search_cutoff, P.cutoff.data(), (size_t)n_queries * 4, hipMemcpyHostToDevice, s));
            TSGPU_HIP_TRY(hipMemcpyAsync(off_words.data(), o.off_words, (size_t)n_queries * 8, hipMemcpyDeviceToHost, s));
            TSGPU_HIP_TRY(hipStreamSynchronize(s));
        }
        if (P.any_deadline) {                        // work items that ran out of time raised their query's flag: partial hits + search_cutoff
            std::vector<uint32_t> cut(n_queries);
            TSGPU_HIP_TRY(hipMemcpy(cut.data(), L.d_cut.p, (size_t)n_queries * 4, hipMemcpyDeviceToHost));
            bool any = false;
            for (uint32_t i = 0; i < n_queries; i++) if (cut[i]) { P.cutoff[i] = 1; any = true; }
            if (any && out->search_cutoff) {
                if (dev_out) TSGPU_HIP_TRY(hipMemcpy(out->search_cutoff, P.cutoff.data(), (size_t)n_queries * 4, hipMemcpyHostToDevice));
                else for (uint32_t i = 0; i < n_queries; i++) out->search_cutoff[i] = P.cutoff[i];
            }
        }
        if (status_host) status_host->assign(P.status.begin(), P.status.end());
        if (bo.cutoff_host) bo.cutoff_host->assign(P.cutoff.begin(), P.cutoff.end());
        const uint64_t t_synced = now_us();

        // ---- bookkeeping: timings + algorithmic bytes ----
        float ms_a = 0, ms_b = 0;
        if (timing) (void)hipEventElapsedTime(&ms_a, L.ev[0], L.ev[1]);
        if (timing) (void)hipEventElapsedTime(&ms_b, L.ev[1], L.ev[2]);
        float ms_find = 0;
        if (timing && find_marked && hit_groups == 1) (void)hipEventElapsedTime(&ms_find, L.ev[0], L.ev[3]);       // one find launch, then one score launch
        uint64_t bytes = P.list_bytes;
        if (!dev_out && out->num_matched) {
            for (uint32_t i = 0; i < n_queries; i++) {
                uint32_t n_num = 0;
                if (DP.on) { for (uint32_t k = 0; k < queries[i].n_sort; k++) if (queries[i].sort[k].kind == TSGPU_SORT_INT64_COLUMN) n_num++; }
                else for (uint32_t k = 0; k < P.q[i].n_sort; k++) if (P.q[i].sort_kind[k] == TSGPU_SORT_INT64_COLUMN) n_num++;
                bytes += 4ull * off_words[i] + 8ull * out->num_matched[i] * n_num;
            }
        } else {
            for (uint32_t i = 0; i < n_queries; i++) bytes += 4ull * off_words[i];
        }
        if (timing || bo.timing) {
            std::lock_guard<std::mutex> tl(ctx->tm_mu);
            ctx->timings.kw_search_ms = ms_a;
            ctx->timings.kw_merge_ms = ms_b;
            ctx->timings.kw_find_ms = ms_find;
            ctx->timings.total_ms = ms_a + ms_b;
            ctx->timings.kw_algorithmic_bytes = bytes;
            ctx->kw_last_hit_groups = hit_groups;
            ctx->kw_last_hit_records = hit_records;
        }
        if (count_touched) {
            uint64_t c[8];
            TSGPU_HIP_TRY(hipMemcpy(c, L.d_touched.p, sizeof(c), hipMemcpyDeviceToHost));
            std::lock_guard<std::mutex> tl(ctx->tm_mu);
            tsgpu_kw_touched& tt = ctx->kw_touched;
            tt.find_driver_ids = c[0]; tt.find_metadata = c[1]; tt.find_tile_dma = c[2]; tt.find_probes = c[3]; tt.find_records = c[4];
            tt.find_work_items = c[5]; tt.find_hit_records = c[6];
            tt.find_requested_bytes = c[0] + c[1] + c[2] + c[3] + c[4];
            // the score kernel's requests per hit record are fixed sizes (kw_score_kernel / load_runs_staged): the record itself, per token one
            // BlockMeta (32 B), the offset_index pair (two 8-byte fetches) and the first offsets (8 B); per numeric sort key one column value; the
            // rare third.. occurrence of a token in a document (one more 8-byte fetch each) is not counted
            uint64_t sb = 0;
            std::vector<uint64_t> nm_host;
            const uint64_t* nm = !dev_out ? out->num_matched : nullptr;
            if (dev_out && o.num_matched) {                          // (measurement only: one small read-back)
                nm_host.resize(n_queries);
                if (hipMemcpy(nm_host.data(), o.num_matched, (size_t)n_queries * 8, hipMemcpyDeviceToHost) == hipSuccess) nm = nm_host.data();
            }
            for (uint32_t i = 0; nm && !DP.on && i < n_queries; i++) {
                uint32_t n_num = 0;
                for (uint32_t k = 0; k < P.q[i].n_sort; k++) if (P.q[i].sort_kind[k] == TSGPU_SORT_INT64_COLUMN) n_num++;
                sb += nm[i] * (4ull * ((P.q[i].n_lists <= 3 ? 3 : KW_MAX_TOKENS) + 1) + 56ull * P.q[i].n_lists + 8ull * n_num);
            }
            tt.score_requested_bytes = sb;
        }
        // ---- matched ids (id_buff / all_result_ids, src/index.cpp:5549, 5565): every work item left an ascending segment ----
        std::vector<uint32_t> ne;
        if (keep_ids && n_work && (bo.record_last || bo.id_lists)) {
            ne.resize(n_work);
            TSGPU_HIP_TRY(hipMemcpy(ne.data(), part.n_emit, (size_t)n_work * 4, hipMemcpyDeviceToHost));
        }
        if (bo.record_last) {                        // legacy single-caller API (tsgpu_result_ids): remember where the segments live
            L.last_ids_off.assign(n_queries, 0);
            L.last_chunk_emit.assign(n_queries, {});
            L.last_chunk_off.assign(n_queries, {});
            L.last_ids_unsorted.assign(n_queries, 0);
            if (keep_ids && n_work) {
                for (uint32_t i = 0; i < n_queries; i++) {
                    if (P.status[i] != TSGPU_OK || P.q[i].n_work == 0) continue;
                    L.last_ids_off[i] = P.q[i].ids_out_off;
                    L.last_chunk_emit[i].assign(ne.begin() + P.q[i].first_work, ne.begin() + P.q[i].first_work + P.q[i].n_work);
                    L.last_chunk_off[i].resize(P.q[i].n_work);
                    for (uint32_t c = 0; c < P.q[i].n_work; c++) L.last_chunk_off[i][c] = work[P.q[i].first_work + c].ids_out_off;
                    L.last_ids_unsorted[i] = P.q[i].mf_index != KW_NONE;      // several driver lists: segments are sorted, their union is not
                }
            }
        }
        if (bo.id_lists) {
            // per-call id lists: the segments are gathered into one dense array on the device (one launch, one download) and belong
            // to THIS call — concurrent callers never see each other's ids
            tsgpu_id_lists& il = *bo.id_lists;
            il.begin.assign((size_t)n_queries + 1, 0);
            std::vector<KwIdCopy> segs;
            uint64_t at = 0;
            for (uint32_t i = 0; i < n_queries; i++) {
                il.begin[i] = at;
                if (P.status[i] != TSGPU_OK || P.q[i].n_work == 0) continue;
                for (uint32_t c = 0; c < P.q[i].n_work; c++) {
                    const uint32_t cnt = ne[P.q[i].first_work + c];
                    if (!cnt) continue;
                    segs.push_back({P.q[i].ids_out_off + work[P.q[i].first_work + c].ids_out_off, at, cnt, 0u});
                    at += cnt;
                }
            }
            il.begin[n_queries] = at;
            bool to_dev = bo.ids_dev != nullptr;              // the ids stay on the device, in the caller's buffer (several driver lists: the union is sorted on the host below)
            for (uint32_t i = 0; to_dev && i < n_queries; i++) if (P.status[i] == TSGPU_OK && P.q[i].mf_index != KW_NONE) to_dev = false;
            if (bo.ids_dev_done) *bo.ids_dev_done = to_dev;
            if (to_dev) {
                if (at) {
                    if ((rc = bo.ids_dev->reserve(at * 4)) || (rc = upload(L.d_idseg, segs.data(), segs.size() * sizeof(KwIdCopy), s))) return rc;
                    hipLaunchKernelGGL(kw_ids_gather_kernel, dim3((uint32_t)segs.size()), dim3(KW_THREADS), 0, s, (const uint32_t*)ids_out, L.d_idseg.as<KwIdCopy>(), bo.ids_dev->as<uint32_t>());
                    TSGPU_HIP_TRY(hipGetLastError());
                    TSGPU_HIP_TRY(hipStreamSynchronize(s));    // (the consumer runs on another stream)
                }
                at = 0;                                       // nothing to download
            }
            il.ids.resize(at);
            if (at) {
                if ((rc = L.d_idflat.reserve(at * 4)) || (rc = upload(L.d_idseg, segs.data(), segs.size() * sizeof(KwIdCopy), s))) return rc;
                hipLaunchKernelGGL(kw_ids_gather_kernel, dim3((uint32_t)segs.size()), dim3(KW_THREADS), 0, s, (const uint32_t*)ids_out, L.d_idseg.as<KwIdCopy>(), L.d_idflat.as<uint32_t>());
                TSGPU_HIP_TRY(hipGetLastError());
                TSGPU_HIP_TRY(hipMemcpyAsync(il.ids.data(), L.d_idflat.p, at * 4, hipMemcpyDeviceToHost, s));
                TSGPU_HIP_TRY(hipStreamSynchronize(s));
                for (uint32_t i = 0; i < n_queries; i++)                      // several driver lists (query_by over several fields): ascending union
                    if (P.q[i].mf_index != KW_NONE && P.status[i] == TSGPU_OK) std::sort(il.ids.begin() + il.begin[i], il.ids.begin() + il.begin[i + 1]);
            }
        }
        {
            const uint64_t t_end = now_us();
            ctx->kw_batches.fetch_add(1); ctx->kw_plan_us.fetch_add(t_planned - t_enter); ctx->kw_upload_us.fetch_add(t_uploaded - t_planned);
            ctx->kw_launch_us.fetch_add(t_launched - t_uploaded); ctx->kw_wait_us.fetch_add(t_synced - t_launched); ctx->kw_book_us.fetch_add(t_end - t_synced);
            auto amax = [](std::atomic<uint64_t>& a, uint64_t v) { uint64_t c = a.load(); while (v > c && !a.compare_exchange_weak(c, v)) {} };
            amax(ctx->kw_max_plan_us, t_planned - t_enter); amax(ctx->kw_max_upload_us, t_uploaded - t_planned);
            amax(ctx->kw_max_launch_us, t_launched - t_uploaded); amax(ctx->kw_max_wait_us, t_synced - t_launched);
        }
        if (host_timing)
            fprintf(stderr, "[tsgpu] kw batch %u queries: plan %llu us, upload+reserve %llu us, launch %llu us, wait+copy %llu us, bookkeeping %llu us\n", n_queries,
                    (unsigned long long)(t_planned - t_enter), (unsigned long long)(t_uploaded - t_planned), (unsigned long long)(t_launched - t_uploaded),
                    (unsigned long long)(t_synced - t_launched), (unsigned long long)(now_us() - t_synced));
        // queries that were not run must not expose stale slots
        if (!dev_out) for (uint32_t i = 0; i < n_queries; i++) if (P.status[i] != TSGPU_OK) { out->n_hits[i] = 0; if (out->num_matched) out->num_matched[i] = 0; }
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_keyword_search_batch: host allocation failed"); }
    return ok();
}

// device-side shard merge (the host version, tsgpu_merge_shard_hits, lives in tsgpu_vec.hip)
// Index::compute_aux_scores' text half (src/index.cpp:8800-8846): text_match score of given documents for given queries' tokens
int tsgpu_keyword_aux_scores(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_queries, const uint32_t* item_query, const uint32_t* item_seq_id,
                             uint32_t n_items, int64_t* scores_out) {
    if (!ctx || !queries || (n_items && (!item_query || !item_seq_id || !scores_out))) return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_aux_scores: NULL argument");
    if (n_items == 0) return ok();
    (void)hipSetDevice(ctx->device);
    const std::shared_ptr<const Snapshot> snap_ref = ctx->snapshot();
    const Snapshot& snap = *snap_ref;
    try {
        std::vector<KwQueryDev> qd(n_queries);
        std::vector<KwQueryMF> mf(n_queries);
        for (uint32_t i = 0; i < n_queries; i++) {
            const tsgpu_kw_query& in = queries[i];
            KwQueryDev& q = qd[i];
            KwQueryMF& m = mf[i];
            memset(&q, 0, sizeof q);
            memset(&m, 0xFF, sizeof m);
            if (in.n_tokens == 0 || in.n_tokens > TSGPU_MAX_QUERY_TOKENS || in.n_fields == 0 || in.n_fields > (uint32_t)KW_MAX_FIELDS || in.match_type > TSGPU_SUM_SCORE)
                return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_keyword_aux_scores: 1..10 tokens, 1..4 query_by fields");
            m.n_fields = in.n_fields;
            m.driver_token = 0;
            for (uint32_t f = 0; f < in.n_fields; f++) {
                const auto fa = snap.field_is_array.find(in.field_ids[f]);
                if (fa == snap.field_is_array.end()) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_keyword_aux_scores: unknown field");
                m.is_array[f] = fa->second ? 1 : 0;
                m.weight[f] = in.field_weights[f];
            }
            // one or_iterator per token that exists in some field (get_field_token_its, src/index.cpp:5598-5660), query order
            uint32_t nl = 0;
            for (uint32_t t = 0; t < in.n_tokens; t++) {
                bool found = false;
                for (uint32_t f = 0; f < in.n_fields; f++) {
                    const uint32_t h = snap.find_handle(in.field_ids[f], in.term_ids[t]);
                    if (h == 0xFFFFFFFFu) continue;
                    m.list[nl][f] = h;
                    found = true;
                }
                if (found) nl++;
            }
            q.n_lists = nl;
            q.n_query_tokens = in.n_tokens;
            q.match_type = in.match_type;
            q.prio_exact = in.prioritize_exact_match; q.prio_pos = in.prioritize_token_position; q.prio_nfields = in.prioritize_num_matching_fields;
            q.total_cost = 0;                                    // compute_aux_scores passes total_cost = 0, syn_orig_num_tokens = -1, no synonym flags (src/index.cpp:8826-8835)
            q.syn_orig_num_tokens = -1;
            q.weight = in.field_weights[0];
            q.mf_index = i;
        }
        for (uint32_t i = 0; i < n_items; i++) if (item_query[i] >= n_queries) return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_aux_scores: item_query out of range");
        LaneLock lane(ctx);
        KwLane& L = *lane.L;
        hipStream_t s = L.stream;
        const size_t o_q = 0, o_m = (sizeof(KwQueryDev) * n_queries + 15) & ~(size_t)15, o_i = (o_m + sizeof(KwQueryMF) * n_queries + 15) & ~(size_t)15;
        const size_t o_out = (o_i + sizeof(KwAuxItem) * n_items + 15) & ~(size_t)15, total = o_out + 8 * (size_t)n_items;
        std::vector<unsigned char> h(total);
        memcpy(h.data() + o_q, qd.data(), sizeof(KwQueryDev) * n_queries);
        memcpy(h.data() + o_m, mf.data(), sizeof(KwQueryMF) * n_queries);
        KwAuxItem* items = (KwAuxItem*)(h.data() + o_i);
        for (uint32_t i = 0; i < n_items; i++) { items[i].query = item_query[i]; items[i].seq_id = item_seq_id[i]; }
        int rc;
        if ((rc = L.d_plan.reserve(total))) return rc;
        TSGPU_HIP_TRY(hipMemcpyAsync(L.d_plan.p, h.data(), o_out, hipMemcpyHostToDevice, s));
        IndexView v = make_view(ctx, snap);
        char* base = (char*)L.d_plan.p;
        hipLaunchKernelGGL(kw_aux_score_kernel, dim3((n_items + 63) / 64), dim3(64), 0, s, v, (const KwQueryDev*)(base + o_q), (const KwQueryMF*)(base + o_m),
                           (const KwAuxItem*)(base + o_i), n_items, (int64_t*)(base + o_out));
        TSGPU_HIP_TRY(hipGetLastError());
        TSGPU_HIP_TRY(hipMemcpyAsync(scores_out, base + o_out, 8 * (size_t)n_items, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipStreamSynchronize(s));
        return ok();
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_keyword_aux_scores: host allocation failed"); }
}

int tsgpu_merge_shard_hits_device(tsgpu_ctx* ctx, const tsgpu_hits* gathered, uint32_t n_shards, uint32_t n_queries, uint32_t k, tsgpu_hits* out) {
    if (!ctx || !gathered || !out) return fail(TSGPU_ERR_INVALID, "tsgpu_merge_shard_hits_device: NULL argument");
    if (gathered->mem != TSGPU_MEM_DEVICE || out->mem != TSGPU_MEM_DEVICE) return fail(TSGPU_ERR_INVALID, "tsgpu_merge_shard_hits_device: device arrays only");
    if (!gathered->keys || !gathered->scores || !gathered->n_hits || !out->keys || !out->scores || !out->n_hits)
        return fail(TSGPU_ERR_INVALID, "tsgpu_merge_shard_hits_device: missing arrays");
    if (n_queries == 0) return ok();
    if (n_shards == 0 || k == 0 || out->k_stride < k) return fail(TSGPU_ERR_INVALID, "tsgpu_merge_shard_hits_device: bad sizes");
    const uint64_t cap_need = (uint64_t)n_shards * gathered->k_stride;
    if (cap_need > 4096 || cap_need > 65535) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_merge_shard_hits_device: n_shards * k_stride > 4096");
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    KwShardIn in;
    in.keys = gathered->keys; in.scores = gathered->scores; in.text_match = gathered->text_match; in.vector_distance = gathered->vector_distance;
    in.match_score_index = gathered->match_score_index; in.n_hits = gathered->n_hits; in.num_matched = gathered->num_matched;
    in.n_shards = n_shards; in.n_queries = n_queries; in.k_in = gathered->k_stride;
    in.packed = nullptr; in.shard_stride = 0; in.words = 0; in.q_out_offset = 0; in.status_out = nullptr; in.cap_per_query = nullptr;
    KwOut o;
    o.keys = out->keys; o.scores = out->scores; o.text_match = out->text_match; o.vector_distance = out->vector_distance;
    o.match_score_index = out->match_score_index; o.n_hits = out->n_hits; o.num_matched = out->num_matched; o.off_words = nullptr; o.k_stride = out->k_stride;
    hipStream_t s = ctx->stream;
    if (cap_need <= 512) hipLaunchKernelGGL((kw_shard_merge_kernel<512>), dim3(n_queries), dim3(KW_THREADS), 0, s, in, o, k);
    else if (cap_need <= 1024) hipLaunchKernelGGL((kw_shard_merge_kernel<1024>), dim3(n_queries), dim3(KW_THREADS), 0, s, in, o, k);
    else if (cap_need <= 2048) hipLaunchKernelGGL((kw_shard_merge_kernel<2048>), dim3(n_queries), dim3(KW_THREADS), 0, s, in, o, k);
    else hipLaunchKernelGGL((kw_shard_merge_kernel<4096>), dim3(n_queries), dim3(KW_THREADS), 0, s, in, o, k);
    TSGPU_HIP_TRY(hipGetLastError());
    TSGPU_HIP_TRY(hipStreamSynchronize(s));
    return ok();
}

// SURVEY §8f rank 2 — Index::search_all_candidates (src/index.cpp:1794-1894) for a batch of user queries: the candidate-token
// combinations of group g are combos[group_begin[g] .. group_begin[g+1]) in the reference's pass order; all of them run as ONE
// keyword batch, kw_candidates_merge_kernel folds each group like the shared Topster, the id-set kernels like id_buff.
int tsgpu_keyword_search_candidates_batch(tsgpu_ctx* ctx, const tsgpu_kw_query* combos, const uint32_t* group_begin, uint32_t n_groups,
                                          tsgpu_hits* out, uint32_t* query_index, uint64_t* found) {
    return kw_candidates_batch_ex(ctx, combos, group_begin, n_groups, out, query_index, found, false, nullptr, nullptr);
}

}  // extern "C"

namespace tsgpu {
int kw_candidates_batch_ex(tsgpu_ctx* ctx, const tsgpu_kw_query* combos, const uint32_t* group_begin, uint32_t n_groups,
                           tsgpu_hits* out, uint32_t* query_index, uint64_t* found, bool raw_pass, uint32_t* pass_mask_dev, const uint16_t* present_elsewhere) {
    if (!ctx || !out || !group_begin) return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_search_candidates_batch: NULL argument");
    if (n_groups == 0) return ok();
    if (!out->keys || !out->scores || !out->n_hits || !out->status) return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_search_candidates_batch: missing output arrays");
    const uint32_t n_combos = group_begin[n_groups];
    if (n_combos && !combos) return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_search_candidates_batch: combos is NULL");
    const uint32_t KS = out->k_stride;
    uint32_t max_passes = 0;
    for (uint32_t g = 0; g < n_groups; g++) {
        if (group_begin[g + 1] < group_begin[g]) return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_search_candidates_batch: group_begin must be non-decreasing");
        max_passes = std::max(max_passes, group_begin[g + 1] - group_begin[g]);
    }
    if (max_passes > (uint32_t)KW_MAX_CANDIDATE_PASSES || (uint64_t)max_passes * KS > 4096)
        return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_keyword_search_candidates_batch: more than 16 combinations per query, or combinations * k_stride > 4096");
    const bool dev_out = out->mem == TSGPU_MEM_DEVICE;
    if (dev_out && (!out->text_match || !out->vector_distance || !out->match_score_index || !out->num_matched))
        return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_search_candidates_batch: device output needs every tsgpu_hits array");
    LaneLock ll(ctx, 0);                              // (tsgpu_candidates_result_ids reads this lane's bitmaps afterwards)
    KwLane& L = *ll.L;
    (void)hipSetDevice(ctx->device);
    hipStream_t s = L.stream;
    try {
        int rc;
        const size_t pslots = (size_t)std::max<uint32_t>(n_combos, 1) * KS, pn = std::max<uint32_t>(n_combos, 1);
        if ((rc = L.d_cand_keys.reserve(pslots * 8)) || (rc = L.d_cand_scores.reserve(pslots * 24)) || (rc = L.d_cand_tm.reserve(pslots * 8)) ||
            (rc = L.d_cand_vd.reserve(pslots * 4)) || (rc = L.d_cand_msi.reserve(pslots)) || (rc = L.d_cand_nh.reserve(pn * 4)) ||
            (rc = L.d_cand_nm.reserve(pn * 8)) || (rc = L.d_cand_st.reserve(pn * 4)))
            return rc;
        tsgpu_hits pass;
        memset(&pass, 0, sizeof(pass));
        pass.mem = TSGPU_MEM_DEVICE; pass.k_stride = KS;
        pass.keys = L.d_cand_keys.as<uint64_t>(); pass.scores = L.d_cand_scores.as<int64_t>(); pass.text_match = L.d_cand_tm.as<int64_t>();
        pass.vector_distance = L.d_cand_vd.as<float>(); pass.match_score_index = L.d_cand_msi.as<int8_t>();
        pass.n_hits = L.d_cand_nh.as<uint32_t>(); pass.num_matched = L.d_cand_nm.as<uint64_t>(); pass.status = L.d_cand_st.as<int32_t>();
        std::vector<int32_t> st, co;
        // all_result_ids: one bitmap per group (below). They are cleared on a second stream while the passes run — 1.25 MB per group at 10M documents is
        // HBM-rate work the scalar-bound find kernel does not notice (as a memset in front of the marks it was 0.18 ms of the 7.3 ms step of 1 000 groups)
        const uint64_t id_words = std::max<uint64_t>(((uint64_t)ctx->num_docs + 31) / 32, 1);
        L.last_cand_groups = 0;
        if (found) {
            if ((uint64_t)n_groups * id_words * 4 > (8ull << 30))
                return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_keyword_search_candidates_batch: id-set bitmaps (n_groups * num_docs / 8 bytes) exceed 8 GiB; split the batch");
            if ((rc = L.d_cand_bits.reserve((size_t)n_groups * id_words * 4)) || (rc = L.d_cand_found.reserve((size_t)n_groups * 8))) return rc;
            if (!L.aux_stream) TSGPU_HIP_TRY(hipStreamCreateWithFlags(&L.aux_stream, hipStreamNonBlocking));
            if (!L.ev_aux) TSGPU_HIP_TRY(hipEventCreateWithFlags(&L.ev_aux, hipEventDisableTiming));
            TSGPU_HIP_TRY(hipEventRecord(L.ev_aux, s));                       // (whatever the caller's stream still does with the last call's bitmaps comes first)
            TSGPU_HIP_TRY(hipStreamWaitEvent(L.aux_stream, L.ev_aux, 0));
            TSGPU_HIP_TRY(hipMemsetAsync(L.d_cand_bits.p, 0, (size_t)n_groups * id_words * 4, L.aux_stream));
            TSGPU_HIP_TRY(hipMemsetAsync(L.d_cand_found.p, 0, (size_t)n_groups * 8, L.aux_stream));
            TSGPU_HIP_TRY(hipEventRecord(L.ev_aux, L.aux_stream));
        }
        L.last_tab_n_work = 0;
        if (n_combos) {
            BatchOpts bo;
            bo.keep_ids = found != nullptr;                    // the union needs every pass's emitted ids
            bo.status_host = &st;
            bo.cutoff_host = &co;
            bo.record_last = false;                            // (the marks below read the batch's tables on the device; tsgpu_candidates_result_ids reads the bitmaps)
            bo.present_elsewhere = present_elsewhere;
            rc = kw_batch_on_lane(ctx, L, combos, n_combos, &pass, bo);
            if (rc) { if (found) (void)hipStreamSynchronize(L.aux_stream); return rc; }
        }
        // a group runs only if every combination of it ran; otherwise it reports the first failing status and no hits
        std::vector<uint32_t> range((size_t)n_groups * 3 + 3, 0);        // per group: first entry, one past the last, Topster capacity
        std::vector<int32_t> gstatus(n_groups, TSGPU_OK), gcut(n_groups, 0);       // a group is cut off when any of its passes was
        for (uint32_t g = 0; g < n_groups; g++) {
            for (uint32_t e = group_begin[g]; e < group_begin[g + 1]; e++) if (e < co.size() && co[e]) gcut[g] = 1;
            for (uint32_t e = group_begin[g]; e < group_begin[g + 1]; e++) if (st[e] != TSGPU_OK) { gstatus[g] = st[e]; break; }
            if (gstatus[g] == TSGPU_OK && group_begin[g + 1] > group_begin[g]) {
                range[3 * g] = group_begin[g]; range[3 * g + 1] = group_begin[g + 1];
                range[3 * g + 2] = std::min(resolve_topster_size(ctx, combos[group_begin[g]]), KS);      // the shared Topster is sized once, by the first pass
            }
        }
        if ((rc = upload(L.d_cand_gb, range.data(), range.size() * 4, s))) return rc;

        const size_t slots = (size_t)n_groups * KS;
        KwOut o;
        o.k_stride = KS; o.off_words = nullptr;
        uint32_t* qi_dev = nullptr;
        if (dev_out) {
            o.keys = out->keys; o.scores = out->scores; o.text_match = out->text_match; o.vector_distance = out->vector_distance;
            o.match_score_index = out->match_score_index; o.n_hits = out->n_hits; o.num_matched = out->num_matched;
            qi_dev = query_index;
        } else {
            if ((rc = L.d_out_keys.reserve(slots * 8)) || (rc = L.d_out_scores.reserve(slots * 24)) || (rc = L.d_out_tm.reserve(slots * 8)) ||
                (rc = L.d_out_vd.reserve(slots * 4)) || (rc = L.d_out_msi.reserve(slots)) || (rc = L.d_out_nh.reserve((size_t)n_groups * 4)) ||
                (rc = L.d_out_nm.reserve((size_t)n_groups * 8)) || (rc = L.d_cand_qi.reserve(slots * 4)))
                return rc;
            o.keys = L.d_out_keys.as<uint64_t>(); o.scores = L.d_out_scores.as<int64_t>(); o.text_match = L.d_out_tm.as<int64_t>();
            o.vector_distance = L.d_out_vd.as<float>(); o.match_score_index = L.d_out_msi.as<int8_t>();
            o.n_hits = L.d_out_nh.as<uint32_t>(); o.num_matched = L.d_out_nm.as<uint64_t>();
            qi_dev = query_index ? L.d_cand_qi.as<uint32_t>() : nullptr;
        }
        KwCandIn in;
        in.keys = pass.keys; in.scores = pass.scores; in.text_match = pass.text_match; in.vector_distance = pass.vector_distance;
        in.match_score_index = pass.match_score_index; in.n_hits = pass.n_hits; in.num_matched = pass.num_matched;
        in.group_range = L.d_cand_gb.as<uint32_t>(); in.k_in = KS;
        in.raw_pass = raw_pass ? 1u : 0u; in.pass_mask = pass_mask_dev;
        const uint64_t cap_need = (uint64_t)std::max<uint32_t>(max_passes, 1) * KS;
        bool any_s2 = false;
        for (uint32_t e = 0; e < n_combos; e++) any_s2 = any_s2 || combos[e].n_sort > 2;
        // the sort-free fold (kw_candidates_rank_kernel); three sort keys AND more than 2 048 slots do not fit its LDS next to the hash table: the sorting kernel
        if (ctx->kw_candidates_rank_fold && !(cap_need > 2048 && any_s2)) {
            ctx->kw_candidates_rank_launches++;
#define TSGPU_CAND_RANK(C) do { if (any_s2) hipLaunchKernelGGL((kw_candidates_rank_kernel<C, true>), dim3(n_groups), dim3(KW_THREADS), 0, s, in, o, qi_dev); \
                                else hipLaunchKernelGGL((kw_candidates_rank_kernel<C, false>), dim3(n_groups), dim3(KW_THREADS), 0, s, in, o, qi_dev); } while (0)
            if (cap_need <= 512) TSGPU_CAND_RANK(512);
            else if (cap_need <= 1024) TSGPU_CAND_RANK(1024);
            else if (cap_need <= 2048) TSGPU_CAND_RANK(2048);
            else hipLaunchKernelGGL((kw_candidates_rank_kernel<4096, false>), dim3(n_groups), dim3(KW_THREADS), 0, s, in, o, qi_dev);
#undef TSGPU_CAND_RANK
        } else
        if (cap_need <= 512) hipLaunchKernelGGL((kw_candidates_merge_kernel<512>), dim3(n_groups), dim3(KW_THREADS), 0, s, in, o, qi_dev);
        else if (cap_need <= 1024) hipLaunchKernelGGL((kw_candidates_merge_kernel<1024>), dim3(n_groups), dim3(KW_THREADS), 0, s, in, o, qi_dev);
        else if (cap_need <= 2048) hipLaunchKernelGGL((kw_candidates_merge_kernel<2048>), dim3(n_groups), dim3(KW_THREADS), 0, s, in, o, qi_dev);
        else hipLaunchKernelGGL((kw_candidates_merge_kernel<4096>), dim3(n_groups), dim3(KW_THREADS), 0, s, in, o, qi_dev);
        TSGPU_HIP_TRY(hipGetLastError());

        // ---- all_result_ids: one bitmap per group, marked from the passes' id segments; the marks count the union (kw_idset_mark_items_kernel) ----
        L.last_cand_found.assign(n_groups, 0);
        if (found) {
            TSGPU_HIP_TRY(hipStreamWaitEvent(s, L.ev_aux, 0));               // the cleared bitmaps
            if (n_combos && L.last_tab_n_work) {
                std::vector<uint32_t> group_of(n_combos, KW_NONE);
                for (uint32_t g = 0; g < n_groups; g++)
                    if (gstatus[g] == TSGPU_OK) for (uint32_t e = group_begin[g]; e < group_begin[g + 1]; e++) group_of[e] = g;
                if ((rc = upload(L.d_cand_segs, group_of.data(), group_of.size() * 4, s))) return rc;
                hipLaunchKernelGGL(kw_idset_mark_items_kernel, dim3(L.last_tab_n_work, 8), dim3(KW_THREADS), 0, s, L.d_ids_out.as<uint32_t>(),
                                   (const KwQueryDev*)L.last_tab_q, (const KwWorkItem*)L.last_tab_w, L.d_part_ne.as<uint32_t>(), L.d_cand_segs.as<uint32_t>(),
                                   L.d_cand_bits.as<uint32_t>(), id_words, L.d_cand_found.as<unsigned long long>());
                TSGPU_HIP_TRY(hipGetLastError());
            }
            TSGPU_HIP_TRY(hipMemcpyAsync(L.last_cand_found.data(), L.d_cand_found.p, (size_t)n_groups * 8, hipMemcpyDeviceToHost, s));
            if (dev_out) TSGPU_HIP_TRY(hipMemcpyAsync(found, L.d_cand_found.p, (size_t)n_groups * 8, hipMemcpyDeviceToDevice, s));
            L.last_cand_groups = n_groups;
            L.last_cand_words = id_words;
        }

        // ---- results ----
        if (!dev_out) {
            TSGPU_HIP_TRY(hipMemcpyAsync(out->n_hits, o.n_hits, (size_t)n_groups * 4, hipMemcpyDeviceToHost, s));
            if (out->num_matched) TSGPU_HIP_TRY(hipMemcpyAsync(out->num_matched, o.num_matched, (size_t)n_groups * 8, hipMemcpyDeviceToHost, s));
            TSGPU_HIP_TRY(hipMemcpyAsync(out->keys, o.keys, slots * 8, hipMemcpyDeviceToHost, s));
            TSGPU_HIP_TRY(hipMemcpyAsync(out->scores, o.scores, slots * 24, hipMemcpyDeviceToHost, s));
            if (out->text_match) TSGPU_HIP_TRY(hipMemcpyAsync(out->text_match, o.text_match, slots * 8, hipMemcpyDeviceToHost, s));
            if (out->vector_distance) TSGPU_HIP_TRY(hipMemcpyAsync(out->vector_distance, o.vector_distance, slots * 4, hipMemcpyDeviceToHost, s));
            if (out->match_score_index) TSGPU_HIP_TRY(hipMemcpyAsync(out->match_score_index, o.match_score_index, slots, hipMemcpyDeviceToHost, s));
            if (query_index) TSGPU_HIP_TRY(hipMemcpyAsync(query_index, qi_dev, slots * 4, hipMemcpyDeviceToHost, s));
            for (uint32_t g = 0; g < n_groups; g++) out->status[g] = gstatus[g];
            if (out->search_cutoff) for (uint32_t g = 0; g < n_groups; g++) out->search_cutoff[g] = gcut[g];
        } else {
            TSGPU_HIP_TRY(hipMemcpyAsync(out->status, gstatus.data(), (size_t)n_groups * 4, hipMemcpyHostToDevice, s));
            if (out->search_cutoff) TSGPU_HIP_TRY(hipMemcpyAsync(out->search_cutoff, gcut.data(), (size_t)n_groups * 4, hipMemcpyHostToDevice, s));
        }
        TSGPU_HIP_TRY(hipStreamSynchronize(s));
        if (found && !dev_out) for (uint32_t g = 0; g < n_groups; g++) found[g] = L.last_cand_found[g];
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_keyword_search_candidates_batch: host allocation failed"); }
    return ok();
}

uint64_t kw_dictionary_fingerprint(tsgpu_ctx* ctx) {
    const std::shared_ptr<const Snapshot> sn = ctx->snapshot();
    return (sn && sn->maps) ? sn->maps->dict_fp : 0ull;
}
int kw_terms_present(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_queries, uint16_t* masks) {
    const std::shared_ptr<const Snapshot> sn = ctx->snapshot();
    for (uint32_t i = 0; i < n_queries; i++) {
        const tsgpu_kw_query& in = queries[i];
        uint16_t m = 0;
        for (uint32_t t = 0; t < in.n_tokens && t < TSGPU_MAX_QUERY_TOKENS; t++)
            for (uint32_t f = 0; f < in.n_fields; f++)
                if (sn && sn->find_handle(in.field_ids[f], in.term_ids[t]) != 0xFFFFFFFFu) { m |= (uint16_t)(1u << t); break; }
        masks[i] = m;
    }
    return TSGPU_OK;
}
int kw_search_batch_masked(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_queries, tsgpu_hits* out, const uint16_t* present_elsewhere) {
    return kw_dispatch(ctx, queries, n_queries, out, false, nullptr, nullptr, nullptr, present_elsewhere);
}
int group_cand_tag(tsgpu_ctx* ctx, const tsgpu_hits* loc, const uint32_t* pass_of_hit, uint32_t n_groups, const uint32_t* pass_mask, const uint64_t* found, uint64_t* meta, hipStream_t s) {
    if (n_groups == 0) return TSGPU_OK;
    (void)hipSetDevice(ctx->device);
    const uint64_t n = (uint64_t)n_groups * loc->k_stride;
    hipLaunchKernelGGL(kw_group_cand_tag_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, loc->keys, pass_of_hit, (const uint32_t*)loc->n_hits, (const int32_t*)loc->status, loc->k_stride, n_groups,
                       pass_mask, (const unsigned long long*)found, (unsigned long long*)meta);
    TSGPU_HIP_TRY(hipGetLastError());
    return TSGPU_OK;
}
int group_cand_fix(tsgpu_ctx* ctx, uint64_t* keys, uint32_t* query_index, const uint32_t* n_hits, uint32_t k_stride, uint32_t q0, uint32_t q1, const uint64_t* meta_all, uint32_t n_shards,
                   uint32_t n_groups, uint64_t* found, hipStream_t s) {
    if (q1 <= q0) return TSGPU_OK;
    (void)hipSetDevice(ctx->device);
    const uint64_t n = (uint64_t)(q1 - q0) * k_stride;
    hipLaunchKernelGGL(kw_group_cand_fix_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, keys, query_index, n_hits, k_stride, q0, q1, (const unsigned long long*)meta_all, n_shards, n_groups,
                       (unsigned long long*)found);
    TSGPU_HIP_TRY(hipGetLastError());
    return TSGPU_OK;
}
}  // namespace tsgpu

extern "C" {

uint64_t tsgpu_candidates_result_ids(tsgpu_ctx* ctx, uint32_t group, uint32_t* out_host, uint64_t cap) {
    if (!ctx) return 0;
    LaneLock ll(ctx, 0);
    KwLane& L = *ll.L;
    (void)hipSetDevice(ctx->device);
    if (group >= L.last_cand_groups) return 0;
    const uint64_t total = L.last_cand_found[group];
    if (!out_host || cap == 0 || total == 0) return total;
    const uint64_t m = std::min(total, cap);
    if (L.d_cand_ids.reserve(m * 4)) return 0;
    hipLaunchKernelGGL(kw_idset_expand_kernel, dim3(1), dim3(KW_THREADS), 0, L.stream, L.d_cand_bits.as<uint32_t>() + (uint64_t)group * L.last_cand_words,
                       L.last_cand_words, L.d_cand_ids.as<uint32_t>(), m);
    if (hipMemcpyAsync(out_host, L.d_cand_ids.p, m * 4, hipMemcpyDeviceToHost, L.stream) != hipSuccess) return 0;
    if (hipStreamSynchronize(L.stream) != hipSuccess) return 0;
    return total;
}

// legacy single-caller form (the ids of the LAST batch run with tsgpu_keep_result_ids(ctx, 1)); concurrent callers use
// tsgpu_keyword_search_batch_ids, whose id lists belong to the call
uint64_t tsgpu_result_ids(tsgpu_ctx* ctx, uint32_t q, uint32_t* out_host, uint64_t cap) {
    if (!ctx) return 0;
    LaneLock ll(ctx, 0);
    KwLane& L = *ll.L;
    (void)hipSetDevice(ctx->device);
    if (q >= L.last_chunk_emit.size()) return 0;
    uint64_t total = 0;
    const auto& ce = L.last_chunk_emit[q];
    for (size_t c = 0; c < ce.size(); c++) total += ce[c];
    if (!out_host || cap == 0) return total;
    if (L.last_ids_unsorted[q]) {
        // multi-field query: one ascending segment per (driver field, chunk); id_buff is ascending in the reference -> merge on the host
        std::vector<uint32_t> all(total);
        uint64_t at = 0;
        for (size_t c = 0; c < ce.size(); c++) {
            if (!ce[c]) continue;
            if (hipMemcpy(all.data() + at, L.d_ids_out.as<uint32_t>() + L.last_ids_off[q] + L.last_chunk_off[q][c], (size_t)ce[c] * 4, hipMemcpyDeviceToHost) != hipSuccess) return 0;
            at += ce[c];
        }
        std::sort(all.begin(), all.end());
        memcpy(out_host, all.data(), (size_t)std::min<uint64_t>(total, cap) * 4);
        return total;
    }
    uint64_t done = 0;
    for (size_t c = 0; c < ce.size() && done < cap; c++) {
        const uint64_t seg = L.last_ids_off[q] + L.last_chunk_off[q][c];
        const uint64_t m = std::min<uint64_t>(ce[c], cap - done);
        if (m && hipMemcpy(out_host + done, L.d_ids_out.as<uint32_t>() + seg, m * 4, hipMemcpyDeviceToHost) != hipSuccess) return 0;
        done += m;
    }
    return total;
}

// TSGPU_PROF builds (tools/ only; not declared in include/tsgpu.h): per-phase cycle counters of kw_search_kernel
int tsgpu_debug_prof(tsgpu_ctx* ctx, int reset, uint64_t* out16) {
    if (!ctx) return TSGPU_ERR_INVALID;
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    if (ctx->d_prof.reserve(16 * 8)) return TSGPU_ERR_NO_MEMORY;
    if (out16) TSGPU_HIP_TRY(hipMemcpy(out16, ctx->d_prof.p, 16 * 8, hipMemcpyDeviceToHost));
    if (reset) TSGPU_HIP_TRY(hipMemset(ctx->d_prof.p, 0, 16 * 8));
    return TSGPU_OK;
}

int tsgpu_kw_last_touched(tsgpu_ctx* ctx, tsgpu_kw_touched* out) {
    if (!ctx || !out) return fail(TSGPU_ERR_INVALID, "tsgpu_kw_last_touched: NULL argument");
    std::lock_guard<std::mutex> lk(ctx->tm_mu);
    *out = ctx->kw_touched;
    return ok();
}

// measurement hook: what the DISTINCT posting lists of a set of (field, term) pairs occupy in the mirror — the working set a batch's requested
// bytes are compared with (bench.py `roofline.l2_refetch`). Exact: the lists' block records are read back from the device.
int tsgpu_kw_lists_footprint(tsgpu_ctx* ctx, const uint32_t* field_ids, const uint32_t* term_ids, uint32_t n, tsgpu_kw_footprint* out) {
    if (!ctx || !out || (n && (!field_ids || !term_ids))) return fail(TSGPU_ERR_INVALID, "tsgpu_kw_lists_footprint: NULL argument");
    (void)hipSetDevice(ctx->device);
    std::shared_ptr<const Snapshot> snap = std::atomic_load(&ctx->snap);
    memset(out, 0, sizeof(*out));
    if (!snap || !snap->ar) return ok();
    std::vector<uint32_t> handles;
    for (uint32_t i = 0; i < n; i++) { const uint32_t h = snap->find_handle(field_ids[i], term_ids[i]); if (h != 0xFFFFFFFFu) handles.push_back(h); }
    std::sort(handles.begin(), handles.end());
    handles.erase(std::unique(handles.begin(), handles.end()), handles.end());
    std::vector<BlockMeta> bm;
    for (uint32_t h : handles) {
        const ListDesc& d = snap->h_lists[h];
        bm.resize(d.n_blocks);
        if (d.n_blocks) TSGPU_HIP_TRY(hipMemcpy(bm.data(), snap->ar->blk_meta.as<BlockMeta>() + d.blk_base, (size_t)d.n_blocks * sizeof(BlockMeta), hipMemcpyDeviceToHost));
        for (const BlockMeta& m : bm) {
            out->ids_bytes += 4ull * packed_words(m.n_ids, m.ids_bits);
            out->payload_bytes += 4ull * (packed_words(m.n_ids, m.oi_bits) + packed_words(m.n_off, m.off_bits));
        }
        out->block_metadata_bytes += (uint64_t)d.n_blocks * (sizeof(BlockIds) + 4 + sizeof(BlockMeta));
        if (d.dir_slot && snap->dir_pool) out->directory_bytes += 8ull * snap->dir_pool->slot_entries;
        out->n_ids += d.n_ids;
    }
    out->n_lists = handles.size();
    return ok();
}

int tsgpu_last_aux_timings(tsgpu_ctx* ctx, tsgpu_aux_timings* out) {
    if (!ctx || !out) return fail(TSGPU_ERR_INVALID, "tsgpu_last_aux_timings: NULL argument");
    std::lock_guard<std::mutex> lk(ctx->tm_mu);
    *out = ctx->aux_timings;
    return ok();
}

int tsgpu_last_timings(tsgpu_ctx* ctx, tsgpu_timings* out) {
    if (!ctx || !out) return fail(TSGPU_ERR_INVALID, "tsgpu_last_timings: NULL argument");
    std::lock_guard<std::mutex> lk(ctx->tm_mu);
    *out = ctx->timings;
    return ok();
}

}  // extern "C"

// ---- tsgpu_group (tsgpu_group.hip): the device-side halves of the keyword exchange ----
namespace tsgpu {
// The flat branch of the vector search (tsgpu_vec.hip computes the distance matrix): every query ranks its filter ids by its sort keys with
// the id's distance as KV::vector_distance / the vector_distance sort key — the wildcard machinery (work items of 256-id blocks, LDS
// Topster, kw_merge_kernel) with one more column. num_matched = the ids the threshold kept = `found`; ids_out (optional) = those ids.
int kw_vector_flat_search(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_queries, tsgpu_hits* out, const KwVFlat* vf, tsgpu_id_lists** ids_out) {
    if (!ctx || !out || !queries || !vf || n_queries == 0) return fail(TSGPU_ERR_INVALID, "tsgpu_vector_search_batch: bad arguments");
    struct CallerCount { std::atomic<int>& c; explicit CallerCount(std::atomic<int>& x) : c(x) { c.fetch_add(1); } ~CallerCount() { c.fetch_sub(1); } } cc(ctx->kw_callers);
    std::unique_ptr<tsgpu_id_lists> lists;
    if (ids_out) { *ids_out = nullptr; lists.reset(new (std::nothrow) tsgpu_id_lists); if (!lists) return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_vector_search_batch: host allocation failed"); }
    BatchOpts bo;
    bo.wildcard = true;
    bo.vflat = vf;
    bo.keep_ids = ids_out != nullptr;
    bo.id_lists = lists.get();
    bo.record_last = false;
    LaneLock ll(ctx, tsgpu::tls_avoid_lane0() ? -2 : -1);
    const int rc = kw_batch_on_lane(ctx, *ll.L, queries, n_queries, out, bo);
    if (rc == TSGPU_OK && ids_out) *ids_out = lists.release();
    return rc;
}

int group_pack_keyword(tsgpu_ctx* ctx, const tsgpu_hits* loc, uint32_t n_q, uint32_t k, uint32_t words, uint64_t* block, hipStream_t s) {
    if (!loc || loc->mem != TSGPU_MEM_DEVICE || !loc->keys || !loc->scores || !loc->n_hits || k == 0 || k > loc->k_stride || (words != 4 && words != 5))
        return fail(TSGPU_ERR_INVALID, "tsgpu_group: bad local result");
    (void)hipSetDevice(ctx->device);
    KwOut o;
    o.keys = loc->keys; o.scores = loc->scores; o.text_match = loc->text_match; o.vector_distance = nullptr; o.match_score_index = nullptr;
    o.n_hits = loc->n_hits; o.num_matched = loc->num_matched; o.off_words = nullptr; o.k_stride = loc->k_stride;
    const uint64_t n = (uint64_t)n_q * k;
    hipLaunchKernelGGL(kw_group_pack_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, o, (const int32_t*)loc->status, n_q, k, words, block);
    TSGPU_HIP_TRY(hipGetLastError());
    return TSGPU_OK;
}
// bound-pruned exchange (kw_kernels.hip.h, "bound-pruned exchange"): this shard's kq-th best entry per query -> kth[n_q][4]
static KwOut kw_out_of(const tsgpu_hits* loc) {
    KwOut o;
    o.keys = loc->keys; o.scores = loc->scores; o.text_match = loc->text_match; o.vector_distance = nullptr; o.match_score_index = nullptr;
    o.n_hits = loc->n_hits; o.num_matched = loc->num_matched; o.off_words = nullptr; o.k_stride = loc->k_stride;
    return o;
}
int group_kw_kth(tsgpu_ctx* ctx, const tsgpu_hits* loc, uint32_t n_q, uint32_t k, uint32_t n_shards, const uint32_t* caps_dev, int64_t* kth, hipStream_t s) {
    (void)hipSetDevice(ctx->device);
    hipLaunchKernelGGL(kw_group_kth_kernel, dim3((n_q + 255) / 256), dim3(256), 0, s, kw_out_of(loc), (const int32_t*)loc->status, caps_dev, n_q, k, n_shards, kth);
    TSGPU_HIP_TRY(hipGetLastError());
    return TSGPU_OK;
}
// ... the bounds from the gathered reports, this shard's entries at or above them packed into n_dst slices of slice_words u64 (per header pairs + room for
// per * k entries); cursor[n_dst] (zeroed here) ends as the slices' entry totals
int group_kw_prune_pack(tsgpu_ctx* ctx, const tsgpu_hits* loc, uint32_t n_q, uint32_t n_pad, uint32_t k, uint32_t words, const uint32_t* caps_dev, const int64_t* kth_all,
                        uint32_t n_shards, uint32_t per, uint32_t n_dst, uint64_t slice_words, uint64_t* block, uint32_t* cursor, hipStream_t s) {
    (void)hipSetDevice(ctx->device);
    TSGPU_HIP_TRY(hipMemsetAsync(cursor, 0, (size_t)n_dst * 4, s));
    hipLaunchKernelGGL(kw_group_prune_pack_kernel, dim3((n_pad + 3) / 4), dim3(256), 0, s, kw_out_of(loc), (const int32_t*)loc->status, caps_dev, n_q, n_pad, k, words, kth_all, n_shards,
                       per, slice_words, block, cursor);
    TSGPU_HIP_TRY(hipGetLastError());
    return TSGPU_OK;
}
int group_store_keyword_slice(tsgpu_ctx* ctx, const tsgpu_hits* loc, uint32_t n_q, uint32_t q_out_offset, uint32_t k, const tsgpu_hits* out, hipStream_t s) {
    if (n_q == 0) return TSGPU_OK;
    (void)hipSetDevice(ctx->device);
    KwOut l, o;
    l.keys = loc->keys; l.scores = loc->scores; l.text_match = loc->text_match; l.vector_distance = nullptr; l.match_score_index = nullptr;
    l.n_hits = loc->n_hits; l.num_matched = loc->num_matched; l.off_words = nullptr; l.k_stride = loc->k_stride;
    o.keys = out->keys; o.scores = out->scores; o.text_match = out->text_match; o.vector_distance = nullptr; o.match_score_index = nullptr;
    o.n_hits = out->n_hits; o.num_matched = out->num_matched; o.off_words = nullptr; o.k_stride = out->k_stride;
    const uint64_t n = (uint64_t)n_q * k;
    hipLaunchKernelGGL(kw_group_store_slice_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, l, (const int32_t*)loc->status, n_q, q_out_offset, k, o, out->status);
    TSGPU_HIP_TRY(hipGetLastError());
    return TSGPU_OK;
}
void group_resolve_topster_sizes(const tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_q, uint32_t* caps) {
    for (uint32_t i = 0; i < n_q; i++) caps[i] = resolve_topster_size(ctx, queries[i]);
}
int group_merge_keyword(tsgpu_ctx* ctx, const uint64_t* gathered, uint64_t shard_stride_words, uint32_t n_shards, uint32_t n_q, uint32_t q_out_offset, uint32_t k, uint32_t words,
                        const uint32_t* caps_dev, const tsgpu_hits* out, hipStream_t s, uint32_t pruned_per) {
    if (n_q == 0) return TSGPU_OK;
    if (!out || out->mem != TSGPU_MEM_DEVICE || !out->keys || !out->scores || !out->n_hits || out->k_stride < k) return fail(TSGPU_ERR_INVALID, "tsgpu_group: bad output arrays");
    const uint64_t cap_need = (uint64_t)n_shards * k;
    if (cap_need > 4096) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_group: members * k > 4096");
    (void)hipSetDevice(ctx->device);
    KwShardIn in;
    memset(&in, 0, sizeof in);
    in.n_shards = n_shards; in.n_queries = n_q; in.k_in = k;
    in.packed = gathered; in.shard_stride = shard_stride_words; in.words = words; in.q_out_offset = q_out_offset; in.status_out = out->status; in.cap_per_query = caps_dev;
    in.pruned_per = pruned_per;
    KwOut o;
    o.keys = out->keys; o.scores = out->scores; o.text_match = out->text_match; o.vector_distance = out->vector_distance;
    o.match_score_index = nullptr; o.n_hits = out->n_hits; o.num_matched = out->num_matched; o.off_words = nullptr; o.k_stride = out->k_stride;
    if (cap_need <= 512) hipLaunchKernelGGL((kw_shard_merge_kernel<512>), dim3(n_q), dim3(KW_THREADS), 0, s, in, o, k);
    else if (cap_need <= 1024) hipLaunchKernelGGL((kw_shard_merge_kernel<1024>), dim3(n_q), dim3(KW_THREADS), 0, s, in, o, k);
    else if (cap_need <= 2048) hipLaunchKernelGGL((kw_shard_merge_kernel<2048>), dim3(n_q), dim3(KW_THREADS), 0, s, in, o, k);
    else hipLaunchKernelGGL((kw_shard_merge_kernel<4096>), dim3(n_q), dim3(KW_THREADS), 0, s, in, o, k);
    TSGPU_HIP_TRY(hipGetLastError());
    return TSGPU_OK;
}
}  // namespace tsgpu

#include "tsgpu_groupby.inc.h"
