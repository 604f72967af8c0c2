// tsgpu_groupby.inc.h — host side of tsgpu_keyword_search_grouped_batch (include/tsgpu.h; kernels: kw_groupby.hip.h). Included at the end of
// tsgpu.hip: it translates queries like plan_batch does and runs the id pass through kw_dispatch.
//
// A grouped batch runs in two steps on the device:
//   1. the ordinary keyword pass with a Topster of ONE entry per query, keeping the matched ids (tsgpu_keyword_search_batch_ids' path:
//      find kernels, exclusion / filter ids, num_keyword_matches, the deadline) — a wildcard query's ids are its filter ids (every seq_id)
//      minus the excluded ids, laid out by the host;
//   2. kw_groupby.hip.h over the matched ids: score every id, fold the groups, select, deliver.
// The matched ids cross PCIe twice on the way (down with the id lists, up again as the kernels' input): 8 bytes per matched document next to
// ~100 bytes of table traffic on the device; the ungrouped keyword path is not touched.
#pragma once

namespace tsgpu {
struct GroupByScratch {
    DevBuf ids, gq, qd, mf, s0, s1, s2, dkey, rslot, hkey, hcount, hbest, hrank, members, glist, gcount;
    DevBuf n_groups, g_total, g_dkey, g_found, g_size, g_mofs, g_mcur, loglog, loglog_hist;
    DevBuf o_keys, o_scores, o_tm, o_vd, o_msi, o_nhits;
    void release() {
        DevBuf* b[] = {&ids, &gq, &qd, &mf, &s0, &s1, &s2, &dkey, &rslot, &hkey, &hcount, &hbest, &hrank, &members, &glist, &gcount, &loglog_hist, &n_groups, &g_total, &g_dkey, &g_found,
                       &g_size, &g_mofs, &g_mcur, &loglog, &o_keys, &o_scores, &o_tm, &o_vd, &o_msi, &o_nhits};
        for (auto* x : b) x->release();
    }
};

// LogLogBeta::cardinality() (include/loglogbeta.h:30-44 betaApprox, :62-75 regSumAndZeros, :107-121) from the sketch's registers, or from how many
// registers hold each value (gb_select_kernel's histogram): the reference adds 2^-register over the registers IN ORDER; with no register above 38 every
// partial sum is a multiple of 2^-38 below 2^15 — 53 bits, exact in a double in any order — so the histogram gives the same double. A register above 38
// (a hash with 38 leading zeros behind the bucket bits: once in 2^38 keys) sends the caller to the ordered sum over the registers.
static double gb_loglog_estimate(double sum, double ez) {
    const double m = 16384.0;
    const double alpha = 0.7213 / (1.0 + 1.079 / m);
    const double zl = std::log(ez + 1.0);
    const double beta = -0.370393911 * ez + 0.070471823 * zl + 0.17393686 * std::pow(zl, 2) + 0.16339839 * std::pow(zl, 3) - 0.09237745 * std::pow(zl, 4)
                        + 0.03738027 * std::pow(zl, 5) - 0.005384159 * std::pow(zl, 6) + 0.00042419 * std::pow(zl, 7);
    double estimate = alpha * m * (m - ez) / (beta + sum);
    if (estimate < 0.0) estimate = 0.0;
    return estimate;
}
static uint64_t gb_loglog_cardinality(const uint8_t* regs) {
    double sum = 0.0, ez = 0.0;
    for (uint32_t i = 0; i < 16384; i++) {
        if (regs[i] == 0) ez += 1.0;
        sum += std::ldexp(1.0, -(int)regs[i]);
    }
    return (uint64_t)gb_loglog_estimate(sum, ez);
}
static bool gb_loglog_cardinality_hist(const uint32_t* hist, uint64_t* out) {      // false: a register above 38, use the registers
    uint64_t scaled = 0;                                                          // sum * 2^38
    for (uint32_t v = 0; v < GB_LOGLOG_HIST; v++) {
        if (!hist[v]) continue;
        if (v > 38) return false;
        scaled += (uint64_t)hist[v] << (38 - v);
    }
    *out = (uint64_t)gb_loglog_estimate(std::ldexp((double)scaled, -38), (double)hist[0]);
    return true;
}

// tsgpu_kw_query -> the device description the scoring functions read, in the multi-field form (a single query_by field is the one-field case of
// compute_aggregated_score); the checks are plan_batch's
static int gb_translate(const tsgpu_ctx* ctx, const Snapshot& snap, const tsgpu_kw_query& in, bool wildcard, KwQueryDev& q, KwQueryMF& m) {
    memset(&q, 0, sizeof q);
    memset(&m, 0xFF, sizeof m);
    q.mf_index = KW_NONE;
    q.syn_orig_num_tokens = -1;
    if (in.n_sort > TSGPU_MAX_SORT_KEYS) return TSGPU_ERR_INVALID;
    for (uint32_t s = 0; s < in.n_sort; s++) {
        if (in.sort[s].kind > TSGPU_SORT_INT64_COLUMN) return TSGPU_ERR_UNSUPPORTED;
        if (in.sort[s].kind == TSGPU_SORT_INT64_COLUMN && in.sort[s].column >= ctx->columns.size()) return TSGPU_ERR_UNSUPPORTED;
        if (in.sort[s].order != 1 && in.sort[s].order != -1) return TSGPU_ERR_UNSUPPORTED;
        q.sort_kind[s] = in.sort[s].kind; q.sort_order[s] = in.sort[s].order; q.sort_col[s] = in.sort[s].column;
    }
    q.n_sort = (uint8_t)in.n_sort;
    if (wildcard) { m.n_fields = 0; return TSGPU_OK; }
    if (in.n_tokens == 0 || in.n_tokens > TSGPU_MAX_QUERY_TOKENS || in.n_fields == 0 || in.n_fields > (uint32_t)KW_MAX_FIELDS) return TSGPU_ERR_UNSUPPORTED;
    if (in.n_dropped > TSGPU_MAX_DROPPED_TOKENS || in.n_tokens + in.n_dropped > TSGPU_MAX_QUERY_TOKENS) return TSGPU_ERR_UNSUPPORTED;
    if (in.match_type > TSGPU_SUM_SCORE) return TSGPU_ERR_INVALID;
    m.n_fields = in.n_fields;
    m.driver_token = 0;
    for (uint32_t f = 0; f < in.n_fields; f++) {
        const auto fa = snap.field_is_array.find(in.field_ids[f]);
        if (fa == snap.field_is_array.end()) return TSGPU_ERR_NOT_FOUND;
        m.is_array[f] = fa->second ? 1 : 0;
        m.weight[f] = in.field_weights[f];
    }
    uint32_t nl = 0;
    auto add_token = [&](uint32_t term) {                // one or_iterator per token that exists in some field, query order (get_field_token_its, src/index.cpp:5598-5660)
        bool found = false;
        for (uint32_t f = 0; f < in.n_fields; f++) {
            const uint32_t h = snap.find_handle(in.field_ids[f], term);
            if (h == 0xFFFFFFFFu) continue;
            m.list[nl][f] = h;
            found = true;
        }
        if (found) nl++;
    };
    for (uint32_t t = 0; t < in.n_tokens; t++) add_token(in.term_ids[t]);
    q.n_required = nl;
    for (uint32_t t = 0; t < in.n_dropped; t++) add_token(in.dropped_term_ids[t]);        // after the query's own tokens (:5271-5290)
    q.n_lists = nl;
    q.n_query_tokens = in.n_tokens;
    q.match_type = in.match_type;
    q.prio_exact = in.prioritize_exact_match ? 1 : 0; q.prio_pos = in.prioritize_token_position ? 1 : 0; q.prio_nfields = in.prioritize_num_matching_fields ? 1 : 0;
    q.total_cost = in.total_cost;
    q.weight = in.field_weights[0];
    q.syn_orig_num_tokens = (int8_t)((int)in.syn_orig_num_tokens_p1 - 1);
    q.orig_num_tokens = in.orig_num_tokens; q.is_synonym = in.is_synonym_query ? 1 : 0; q.demote_synonym = in.demote_synonym_match ? 1 : 0;
    return TSGPU_OK;
}
}  // namespace tsgpu

extern "C" {

void tsgpu_groupby_destroy(tsgpu_ctx* ctx) {
    if (ctx->groupby) { ctx->groupby->release(); delete ctx->groupby; ctx->groupby = nullptr; }
}

int tsgpu_keyword_search_grouped_batch(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, const tsgpu_group_by* groups, uint32_t n_queries,
                                       tsgpu_hits* out, tsgpu_grouped_hits* gout, tsgpu_id_lists** ids_out) {
    if (ids_out) *ids_out = nullptr;
    if (!ctx || !out || !gout || (n_queries && (!queries || !groups))) return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_search_grouped_batch: NULL argument");
    if (n_queries == 0) return ok();
    if (out->mem != TSGPU_MEM_HOST) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_keyword_search_grouped_batch: host output arrays only");
    if (!out->keys || !out->scores || !out->n_hits || !out->status) return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_search_grouped_batch: keys / scores / n_hits / status are required");
    if (!gout->n_groups || !gout->distinct_key || !gout->group_size || !gout->group_found || gout->g_stride == 0 || out->k_stride == 0)
        return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_search_grouped_batch: n_groups / distinct_key / group_size / group_found and the strides are required");
    std::lock_guard<std::mutex> lk(ctx->mu);                 // like a search holds Index::mutex: no commit between the id pass and the scoring
    (void)hipSetDevice(ctx->device);
    const std::shared_ptr<const Snapshot> snap_ref = ctx->snapshot();
    const Snapshot& snap = *snap_ref;
    hipStream_t s = ctx->stream;
    static const bool host_timing = getenv("TSGPU_HOST_TIMING") != nullptr;
    const uint64_t t_enter = now_us();
    try {
        std::vector<int32_t> status(n_queries, TSGPU_OK), cutoff(n_queries, 0);
        std::vector<uint64_t> num_matched(n_queries, 0);
        std::vector<KwQueryDev> qd(n_queries);
        std::vector<KwQueryMF> mf(n_queries);
        std::vector<GbQuery> gq(n_queries);
        std::vector<uint32_t> kcap(n_queries, 1);
        bool any_first = false, any_second = false;
        uint32_t max_k = 1;
        for (uint32_t i = 0; i < n_queries; i++) {
            const tsgpu_kw_query& in = queries[i];
            const tsgpu_group_by& gb = groups[i];
            memset(&gq[i], 0, sizeof(GbQuery));
            int st = gb_translate(ctx, snap, in, gb.wildcard != 0, qd[i], mf[i]);
            if (st == TSGPU_OK && (gb.group_limit == 0 || gb.group_limit > TSGPU_MAX_GROUP_LIMIT)) st = TSGPU_ERR_INVALID;
            if (st == TSGPU_OK && gb.column >= ctx->columns.size()) st = TSGPU_ERR_NOT_FOUND;
            if (st == TSGPU_OK && (in.n_filter != 0 && !in.filter_ids)) st = TSGPU_ERR_INVALID;
            if (st == TSGPU_OK && (in.n_excluded != 0 && !in.excluded_ids)) st = TSGPU_ERR_INVALID;
            uint32_t k = 1;
            if (st == TSGPU_OK) {
                k = resolve_topster_size(ctx, in);
                if (k > TSGPU_MAX_TOPK) st = TSGPU_ERR_UNSUPPORTED;
                else if (k > gout->g_stride) st = TSGPU_ERR_INVALID;
                else if ((uint64_t)k * (gb.first_pass ? 1u : gb.group_limit) > out->k_stride) st = TSGPU_ERR_INVALID;   // second pass: slot r * group_limit + j
            }
            status[i] = st;
            if (st != TSGPU_OK) continue;
            kcap[i] = k;
            max_k = std::max(max_k, k);
            qd[i].k = k;
            gq[i].k = k; gq[i].group_limit = gb.group_limit; gq[i].column = gb.column;
            gq[i].first_pass = gb.first_pass ? 1 : 0; gq[i].group_missing_values = gb.group_missing_values ? 1 : 0; gq[i].wildcard = gb.wildcard ? 1 : 0;
            (gb.first_pass ? any_first : any_second) = true;
        }
        // ---- step 1: the matched ids ----
        std::vector<std::vector<uint32_t>> wild_ids(n_queries);
        std::unique_ptr<tsgpu_id_lists> idl;
        std::vector<uint32_t> kw_index(n_queries, 0xFFFFFFFFu);
        {
            std::vector<tsgpu_kw_query> kq;
            std::vector<uint32_t> kq_of;
            for (uint32_t i = 0; i < n_queries; i++) {
                if (status[i] != TSGPU_OK) continue;
                const tsgpu_kw_query& in = queries[i];
                if (groups[i].wildcard) {
                    // Index::search_wildcard ranks the filter ids (every seq_id without a filter) minus the excluded ids (src/index.cpp:6674-6676)
                    std::vector<uint32_t>& w = wild_ids[i];
                    const uint32_t n = in.n_filter ? in.n_filter : ctx->num_docs;
                    w.reserve(n);
                    uint32_t e = 0;
                    for (uint32_t j = 0; j < n; j++) {
                        const uint32_t id = in.n_filter ? in.filter_ids[j] : j;
                        while (e < in.n_excluded && in.excluded_ids[e] < id) e++;
                        if (e < in.n_excluded && in.excluded_ids[e] == id) continue;
                        w.push_back(id);
                    }
                    num_matched[i] = w.size();
                    continue;
                }
                kw_index[i] = (uint32_t)kq.size();
                kq.push_back(in);
                kq.back().topster_size = 1;              // the pass is run for its ids and counters; the Topster it fills is not read
                kq_of.push_back(i);
            }
            if (!kq.empty()) {
                const uint32_t nk = (uint32_t)kq.size();
                std::vector<uint64_t> t_keys(nk), t_nm(nk);
                std::vector<int64_t> t_scores((size_t)nk * 3);
                std::vector<int8_t> t_msi(nk);
                std::vector<uint32_t> t_nh(nk);
                std::vector<int32_t> t_st(nk), t_co(nk);
                tsgpu_hits th;
                memset(&th, 0, sizeof th);
                th.mem = TSGPU_MEM_HOST; th.k_stride = 1;
                th.keys = t_keys.data(); th.scores = t_scores.data(); th.match_score_index = t_msi.data(); th.n_hits = t_nh.data(); th.num_matched = t_nm.data();
                th.status = t_st.data(); th.search_cutoff = t_co.data();
                tsgpu_id_lists* raw = nullptr;
                const int rc = kw_dispatch(ctx, kq.data(), nk, &th, false, &raw);
                idl.reset(raw);
                if (rc != TSGPU_OK) return rc;
                for (uint32_t j = 0; j < nk; j++) {
                    const uint32_t i = kq_of[j];
                    status[i] = t_st[j]; cutoff[i] = t_co[j]; num_matched[i] = t_nm[j];
                }
            }
        }
        const uint64_t t_ids = now_us();
        // ---- layout ----
        uint64_t n_items = 0, n_slots = 0, n_blocks = 0;
        for (uint32_t i = 0; i < n_queries; i++) {
            GbQuery& g = gq[i];
            g.item_begin = n_items; g.tab_off = n_slots;
            g.run = status[i] == TSGPU_OK ? 1 : 0;
            uint64_t n = 0;
            if (g.run) n = groups[i].wildcard ? wild_ids[i].size() : tsgpu_id_lists_count(idl.get(), kw_index[i]);
            if (n > 0x7FFFFFFFull) { status[i] = TSGPU_ERR_UNSUPPORTED; g.run = 0; n = 0; }
            g.n_items = (uint32_t)n;
            uint64_t size = 64;
            while (size < 2 * n) size <<= 1;
            g.tab_mask = (uint32_t)(size - 1);
            g.first_block = n_blocks;
            n_blocks += (n + GB_THREADS - 1) / GB_THREADS;
            n_items += n;
            n_slots += size + 1;                          // + the slot of the key ~0
        }
        if (n_blocks > 0x1FFFFFFFull) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_keyword_search_grouped_batch: too many matched ids in one batch");
        if (n_slots * 20 + n_items * 48 > (64ull << 30)) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_keyword_search_grouped_batch: the group tables of this batch exceed 64 GiB; split it");
        if (!ctx->groupby) ctx->groupby = new GroupByScratch;
        GroupByScratch& S = *ctx->groupby;
        const uint32_t gs = gout->g_stride, ks = out->k_stride;
        const size_t n_out = (size_t)n_queries * ks, n_g = (size_t)n_queries * gs;
        const uint64_t ni = std::max<uint64_t>(n_items, 1);
        int rc;
        if ((rc = S.ids.reserve(ni * 4)) || (rc = S.gq.reserve(sizeof(GbQuery) * n_queries)) || (rc = S.qd.reserve(sizeof(KwQueryDev) * n_queries)) ||
            (rc = S.mf.reserve(sizeof(KwQueryMF) * n_queries)) || (rc = S.s0.reserve(ni * 8)) || (rc = S.s1.reserve(ni * 8)) || (rc = S.s2.reserve(ni * 8)) ||
            (rc = S.dkey.reserve(ni * 8)) || (rc = S.rslot.reserve(ni * 4)) || (rc = S.members.reserve(ni * 4)) || (rc = S.glist.reserve(ni * 4)) ||
            (rc = S.gcount.reserve((size_t)n_queries * 4)) || (rc = S.loglog_hist.reserve((size_t)n_queries * GB_LOGLOG_HIST * 4)) || (rc = S.hkey.reserve(n_slots * 8)) ||
            (rc = S.hcount.reserve(n_slots * 4)) || (rc = S.hbest.reserve(n_slots * 4)) || (rc = S.hrank.reserve(n_slots * 4)) ||
            (rc = S.n_groups.reserve((size_t)n_queries * 4)) || (rc = S.g_total.reserve((size_t)n_queries * 8)) || (rc = S.g_dkey.reserve(n_g * 8)) ||
            (rc = S.g_found.reserve(n_g * 4)) || (rc = S.g_size.reserve(n_g * 4)) || (rc = S.g_mofs.reserve(n_g * 4)) || (rc = S.g_mcur.reserve(n_g * 4)) ||
            (rc = S.o_keys.reserve(n_out * 8)) || (rc = S.o_scores.reserve(n_out * 24)) || (rc = S.o_tm.reserve(n_out * 8)) || (rc = S.o_vd.reserve(n_out * 4)) ||
            (rc = S.o_msi.reserve(n_out)) || (rc = S.o_nhits.reserve((size_t)n_queries * 4)))
            return rc;
        const bool want_loglog = any_first;
        if (want_loglog && (rc = S.loglog.reserve((size_t)n_queries * GB_LOGLOG_M))) return rc;
        std::vector<uint32_t> flat_ids(n_items);                 // ONE upload (a copy per query cost 7 us each: 7 ms of a 1 000-query batch)
        for (uint32_t i = 0; i < n_queries; i++) {
            if (!gq[i].run || gq[i].n_items == 0) continue;
            const uint32_t* src = groups[i].wildcard ? wild_ids[i].data() : tsgpu_id_lists_ids(idl.get(), kw_index[i]);
            memcpy(flat_ids.data() + gq[i].item_begin, src, (size_t)gq[i].n_items * 4);
        }
        if (n_items) TSGPU_HIP_TRY(hipMemcpyAsync(S.ids.p, flat_ids.data(), (size_t)n_items * 4, hipMemcpyHostToDevice, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(S.gq.p, gq.data(), sizeof(GbQuery) * n_queries, hipMemcpyHostToDevice, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(S.qd.p, qd.data(), sizeof(KwQueryDev) * n_queries, hipMemcpyHostToDevice, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(S.mf.p, mf.data(), sizeof(KwQueryMF) * n_queries, hipMemcpyHostToDevice, s));
        TSGPU_HIP_TRY(hipMemsetAsync(S.hkey.p, 0xFF, n_slots * 8, s));
        TSGPU_HIP_TRY(hipMemsetAsync(S.hcount.p, 0, n_slots * 4, s));
        TSGPU_HIP_TRY(hipMemsetAsync(S.hbest.p, 0xFF, n_slots * 4, s));
        TSGPU_HIP_TRY(hipMemsetAsync(S.hrank.p, 0xFF, n_slots * 4, s));
        TSGPU_HIP_TRY(hipMemsetAsync(S.gcount.p, 0, (size_t)n_queries * 4, s));
        if (want_loglog) TSGPU_HIP_TRY(hipMemsetAsync(S.loglog.p, 0, (size_t)n_queries * GB_LOGLOG_M, s));     // (the kernel writes the non-zero register words)
        // (the output staging arrays are not cleared: a caller reads the slots n_groups / group_size / n_hits describe, and those are written)
        GbArgs a;
        a.gq = S.gq.as<GbQuery>(); a.n_queries = n_queries; a.queries = S.qd.as<KwQueryDev>(); a.mfs = S.mf.as<KwQueryMF>();
        a.n_items = n_items; a.ids = S.ids.as<uint32_t>();
        a.s0 = S.s0.as<int64_t>(); a.s1 = S.s1.as<int64_t>(); a.s2 = S.s2.as<int64_t>(); a.dkey = S.dkey.as<unsigned long long>(); a.rslot = S.rslot.as<uint32_t>();
        a.hkey = S.hkey.as<unsigned long long>(); a.hcount = S.hcount.as<uint32_t>(); a.hbest = S.hbest.as<uint32_t>(); a.hrank = S.hrank.as<uint32_t>();
        a.members = S.members.as<uint32_t>(); a.glist = S.glist.as<uint32_t>(); a.gcount = S.gcount.as<uint32_t>(); a.g_stride = gs;
        a.n_groups = S.n_groups.as<uint32_t>(); a.groups_total = S.g_total.as<unsigned long long>();
        a.g_dkey = S.g_dkey.as<unsigned long long>(); a.g_found = S.g_found.as<uint32_t>(); a.g_size = S.g_size.as<uint32_t>();
        a.g_mofs = S.g_mofs.as<uint32_t>(); a.g_mcur = S.g_mcur.as<uint32_t>();
        a.loglog = want_loglog ? S.loglog.as<uint8_t>() : nullptr; a.loglog_hist = S.loglog_hist.as<uint32_t>();
        a.out.keys = S.o_keys.as<uint64_t>(); a.out.scores = S.o_scores.as<int64_t>(); a.out.text_match = S.o_tm.as<int64_t>();
        a.out.vector_distance = S.o_vd.as<float>(); a.out.match_score_index = S.o_msi.as<int8_t>(); a.out.n_hits = S.o_nhits.as<uint32_t>();
        a.out.num_matched = nullptr; a.out.off_words = nullptr; a.out.k_stride = ks;
        IndexView v = make_view(ctx, snap);
        if (n_items) {
            uint32_t max_lists = 0;
            for (uint32_t i = 0; i < n_queries; i++) if (gq[i].run) max_lists = std::max(max_lists, qd[i].n_lists);
            if (max_lists <= 3) hipLaunchKernelGGL((gb_score_kernel<3>), dim3((uint32_t)n_blocks), dim3(GB_THREADS), 0, s, v, a);
            else hipLaunchKernelGGL((gb_score_kernel<KW_MAX_TOKENS>), dim3((uint32_t)(n_blocks * (GB_THREADS / 64))), dim3(64), 0, s, v, a);
            hipLaunchKernelGGL(gb_insert_kernel, dim3((uint32_t)n_blocks), dim3(GB_THREADS), 0, s, a);
        }
        if (max_k + GB_THREADS <= 512) hipLaunchKernelGGL((gb_select_kernel<512>), dim3(n_queries), dim3(GB_THREADS), 0, s, a);
        else if (max_k + GB_THREADS <= 1024) hipLaunchKernelGGL((gb_select_kernel<1024>), dim3(n_queries), dim3(GB_THREADS), 0, s, a);
        else hipLaunchKernelGGL((gb_select_kernel<2048>), dim3(n_queries), dim3(GB_THREADS), 0, s, a);
        if (any_second && n_items) {
            hipLaunchKernelGGL(gb_scatter_kernel, dim3((uint32_t)n_blocks), dim3(GB_THREADS), 0, s, a);
            const uint64_t pairs = (uint64_t)n_queries * gs;
            hipLaunchKernelGGL(gb_members_kernel, dim3((uint32_t)((pairs + GB_THREADS / 64 - 1) / (GB_THREADS / 64))), dim3(GB_THREADS), 0, s, a);
        }
        TSGPU_HIP_TRY(hipGetLastError());
        uint64_t t_launched = now_us(), t_kernels = t_launched;
        if (host_timing) { TSGPU_HIP_TRY(hipStreamSynchronize(s)); t_kernels = now_us(); }
        // ---- delivery ----
        TSGPU_HIP_TRY(hipMemcpyAsync(out->keys, S.o_keys.p, n_out * 8, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(out->scores, S.o_scores.p, n_out * 24, hipMemcpyDeviceToHost, s));
        if (out->text_match) TSGPU_HIP_TRY(hipMemcpyAsync(out->text_match, S.o_tm.p, n_out * 8, hipMemcpyDeviceToHost, s));
        if (out->vector_distance) TSGPU_HIP_TRY(hipMemcpyAsync(out->vector_distance, S.o_vd.p, n_out * 4, hipMemcpyDeviceToHost, s));
        if (out->match_score_index) TSGPU_HIP_TRY(hipMemcpyAsync(out->match_score_index, S.o_msi.p, n_out, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(out->n_hits, S.o_nhits.p, (size_t)n_queries * 4, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(gout->n_groups, S.n_groups.p, (size_t)n_queries * 4, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(gout->distinct_key, S.g_dkey.p, n_g * 8, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(gout->group_size, S.g_size.p, n_g * 4, hipMemcpyDeviceToHost, s));
        TSGPU_HIP_TRY(hipMemcpyAsync(gout->group_found, S.g_found.p, n_g * 4, hipMemcpyDeviceToHost, s));
        if (gout->groups_total) TSGPU_HIP_TRY(hipMemcpyAsync(gout->groups_total, S.g_total.p, (size_t)n_queries * 8, hipMemcpyDeviceToHost, s));
        std::vector<uint32_t> hist_host;
        if (want_loglog && gout->groups_count) {
            hist_host.resize((size_t)n_queries * GB_LOGLOG_HIST);
            TSGPU_HIP_TRY(hipMemcpyAsync(hist_host.data(), S.loglog_hist.p, hist_host.size() * 4, hipMemcpyDeviceToHost, s));
        }
        if (gout->loglog_registers) {
            if (want_loglog) TSGPU_HIP_TRY(hipMemcpyAsync(gout->loglog_registers, S.loglog.p, (size_t)n_queries * GB_LOGLOG_M, hipMemcpyDeviceToHost, s));
            else memset(gout->loglog_registers, 0, (size_t)n_queries * GB_LOGLOG_M);
        }
        TSGPU_HIP_TRY(hipStreamSynchronize(s));
        const uint64_t t_delivered = now_us();
        if (host_timing)
            fprintf(stderr, "[tsgpu] grouped batch %u queries, %llu matched ids, %llu slots: id pass %llu us, layout+upload+launch %llu us, kernels %llu us, delivery %llu us\n", n_queries,
                    (unsigned long long)n_items, (unsigned long long)n_slots, (unsigned long long)(t_ids - t_enter), (unsigned long long)(t_launched - t_ids),
                    (unsigned long long)(t_kernels - t_launched), (unsigned long long)(t_delivered - t_kernels));
        for (uint32_t i = 0; i < n_queries; i++) {
            out->status[i] = status[i];
            if (out->num_matched) out->num_matched[i] = status[i] == TSGPU_OK ? num_matched[i] : 0;
            if (out->search_cutoff) out->search_cutoff[i] = cutoff[i];
            if (status[i] != TSGPU_OK) { out->n_hits[i] = 0; gout->n_groups[i] = 0; if (gout->groups_total) gout->groups_total[i] = 0; }
            if (gout->groups_count) {
                uint64_t card = 0;
                if (status[i] == TSGPU_OK && groups[i].first_pass && !gb_loglog_cardinality_hist(hist_host.data() + (size_t)i * GB_LOGLOG_HIST, &card)) {
                    std::vector<uint8_t> regs(GB_LOGLOG_M);      // (a register above 38: the ordered sum over the registers themselves)
                    TSGPU_HIP_TRY(hipMemcpy(regs.data(), S.loglog.as<uint8_t>() + (size_t)i * GB_LOGLOG_M, GB_LOGLOG_M, hipMemcpyDeviceToHost));
                    card = gb_loglog_cardinality(regs.data());
                }
                gout->groups_count[i] = card;
            }
        }
        if (ids_out) {
            std::unique_ptr<tsgpu_id_lists> il(new tsgpu_id_lists);
            il->begin.assign((size_t)n_queries + 1, 0);
            for (uint32_t i = 0; i < n_queries; i++) il->begin[i + 1] = il->begin[i] + gq[i].n_items;
            il->ids.resize(il->begin[n_queries]);
            for (uint32_t i = 0; i < n_queries; i++) {
                if (!gq[i].n_items) continue;
                const uint32_t* src = groups[i].wildcard ? wild_ids[i].data() : tsgpu_id_lists_ids(idl.get(), kw_index[i]);
                memcpy(il->ids.data() + il->begin[i], src, (size_t)gq[i].n_items * 4);
            }
            *ids_out = il.release();
        }
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_keyword_search_grouped_batch: host allocation failed"); }
    return ok();
}

}  // extern "C"
