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
    // One block per direction and kind, sub-arrays at 16-byte aligned offsets: a call costs ONE upload, TWO memsets and ONE download whatever its size (a 1-query call
    // from a request thread spent 150 us in a dozen small pageable copies before). in = [GbQuery | KwQueryDev | KwQueryMF | ids]; ff = what starts as 0xFF
    // [hkey | hbest | hrank]; zero = [hcount | gcount]; out = [n_hits | n_groups | groups_total | loglog_hist | g_dkey | g_size | g_found | keys | scores |
    // match_score_index | text_match | vector_distance] (the optional arrays last: only the requested prefix crosses PCIe).
    DevBuf in, ff, zero, out, work, loglog;          // work = per-item records + member lists + per-group offsets (never initialised, never delivered)
    DevBuf ids_dev;                                  // the id pass' matched ids when they never leave the device (a batch of single-field keyword queries)
    PinBuf h_in, h_out;
    // staging of a coalesced round (gb_coalesced, under ctx->mu): grow-only — value-initialising 4 MB of vectors per round cost a millisecond
    std::vector<uint64_t> c_keys, c_nm, c_dk, c_gtot, c_gcnt;
    std::vector<int64_t> c_scores, c_tm;
    std::vector<float> c_vd;
    std::vector<int8_t> c_msi;
    std::vector<uint32_t> c_nh, c_ng, c_gsz, c_gfound;
    std::vector<int32_t> c_st, c_co;
    std::vector<uint8_t> c_regs;
    void release() {
        DevBuf* b[] = {&in, &ff, &zero, &out, &work, &loglog, &ids_dev};
        for (auto* x : b) x->release();
        h_in.release(); h_out.release();
    }
};
struct GbLayout {                                    // byte offsets inside a block
    size_t at = 0;
    size_t take(size_t bytes) { const size_t o = at; at = (at + bytes + 15) & ~(size_t)15; return o; }
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

}  // extern "C"

// one grouped batch; the caller holds ctx->mu (like a search holds Index::mutex: no commit lands between the id pass and the scoring; the scratch is the context's).
// User query u owns the candidate combinations combos[cfirst[u] .. cfirst[u + 1]) — one search_across_fields pass each over ONE collector (Index::search_all_candidates,
// src/index.cpp:1794-1894); the plain entry point passes one combination per query. query_index (nullable): KV::query_index per hit slot.
// `shard` (nullable; tsgpu_group_keyword_search_grouped_batch, one combination per query): this context answers as a doc-range shard of a group —
//   forced_keys / forced_begin: the groups of query i are GIVEN (keys forced_keys[forced_begin[i] .. forced_begin[i + 1]), best first, as the whole collection selected
//     them): returned group r is the r-th key whether or not the shard holds documents of it (then group_found = group_size = 0), groups_count is not computed;
//   pass_mask (out, per user query, zeroed by the caller): bit p = combination p matched something here, and query_index holds every hit's PASS instead of
//     the count of earlier matching passes (which is a property of all shards together);
//   present_elsewhere: per COMBINATION, the tokens that exist on ANOTHER shard (a token missing here is then an empty list, not a dropped token: kw_dispatch).
// A q = * query of a context with a doc range matches the seq_ids the context OWNS.
static int gb_batch_locked(tsgpu_ctx* ctx, const tsgpu_kw_query* combos, const uint32_t* cfirst, const tsgpu_group_by* groups, uint32_t n_queries,
                           tsgpu_hits* out, tsgpu_grouped_hits* gout, uint32_t* query_index, tsgpu_id_lists** ids_out, const tsgpu::GbShard* shard = nullptr) {
    (void)hipSetDevice(ctx->device);
    const std::shared_ptr<const Snapshot> snap_ref = ctx->snapshot();
    const Snapshot& snap = *snap_ref;
    hipStream_t s = ctx->stream;
    static const bool host_timing = getenv("TSGPU_HOST_TIMING") != nullptr;
    const uint64_t t_enter = now_us();
    const uint32_t n_combos = cfirst[n_queries];
    try {
        std::vector<int32_t> status(n_queries, TSGPU_OK), cutoff(n_queries, 0);
        std::vector<uint64_t> num_matched(n_queries, 0);
        std::vector<KwQueryDev> qd(std::max<uint32_t>(n_combos, 1));
        std::vector<KwQueryMF> mf(std::max<uint32_t>(n_combos, 1));
        std::vector<GbQuery> gq(n_queries);
        bool any_first = false, any_second = false, any_dedupe = false;
        uint32_t max_k = 1;
        for (uint32_t i = 0; i < n_queries; i++) {
            const tsgpu_group_by& gb = groups[i];
            memset(&gq[i], 0, sizeof(GbQuery));
            const uint32_t c0 = cfirst[i], nc = cfirst[i + 1] - c0;
            gq[i].first_combo = c0; gq[i].n_combos = nc;
            int st = TSGPU_OK;
            if (nc == 0 || nc > (uint32_t)KW_MAX_CANDIDATE_PASSES || (gb.wildcard && nc != 1)) st = TSGPU_ERR_INVALID;
            for (uint32_t c = c0; st == TSGPU_OK && c < c0 + nc; c++) {
                const tsgpu_kw_query& in = combos[c];
                st = gb_translate(ctx, snap, in, gb.wildcard != 0, qd[c], mf[c]);
                if (st == TSGPU_OK && (in.n_filter != 0 && !in.filter_ids)) st = TSGPU_ERR_INVALID;
                if (st == TSGPU_OK && (in.n_excluded != 0 && !in.excluded_ids)) st = TSGPU_ERR_INVALID;
            }
            if (st == TSGPU_OK && (gb.group_limit == 0 || gb.group_limit > TSGPU_MAX_GROUP_LIMIT)) st = TSGPU_ERR_INVALID;
            if (st == TSGPU_OK && gb.column >= ctx->columns.size()) st = TSGPU_ERR_NOT_FOUND;
            uint32_t k = 1;
            const bool forced = shard && shard->forced_begin;
            if (st == TSGPU_OK) {
                k = resolve_topster_size(ctx, combos[c0]);                 // ONE collector for all combinations (src/index.cpp:3506-3514)
                if (forced) k = std::max<uint32_t>(1, shard->forced_begin[i + 1] - shard->forced_begin[i]);     // (the given groups: all of them are returned)
                if (k > TSGPU_MAX_TOPK) st = TSGPU_ERR_UNSUPPORTED;
                else if (k > gout->g_stride) st = TSGPU_ERR_INVALID;
                else if ((uint64_t)k * (gb.first_pass ? 1u : gb.group_limit) > out->k_stride) st = TSGPU_ERR_INVALID;   // second pass: slot r * group_limit + j
            }
            status[i] = st;
            if (st != TSGPU_OK) continue;
            max_k = std::max(max_k, k);
            gq[i].k = k; gq[i].group_limit = gb.group_limit; gq[i].column = gb.column;
            gq[i].first_pass = gb.first_pass ? 1 : 0; gq[i].group_missing_values = gb.group_missing_values ? 1 : 0; gq[i].wildcard = gb.wildcard ? 1 : 0;
            if (forced) { gq[i].forced = 1; gq[i].forced_begin = shard->forced_begin[i]; gq[i].n_forced = shard->forced_begin[i + 1] - shard->forced_begin[i]; }
            gq[i].dedupe = (!gb.first_pass && nc > 1) ? 1 : 0;             // a second pass counts a document once, with its greatest KV (group_doc_seq_ids + replace-unless-smaller)
            any_dedupe = any_dedupe || gq[i].dedupe;
            (gb.first_pass ? any_first : any_second) = true;
        }
        // ---- step 1: the matched ids ----
        std::vector<std::vector<uint32_t>> wild_ids(n_queries);
        std::vector<uint8_t> iota(n_queries, 0);             // q = * without filter / excluded ids: the matched ids are 0 .. num_docs - 1, written on the device
        bool any_iota = false;
        std::unique_ptr<tsgpu_id_lists> idl;
        std::vector<uint32_t> kw_index(std::max<uint32_t>(n_combos, 1), 0xFFFFFFFFu);      // per combination: its query in the id pass
        std::vector<int32_t> c_status(std::max<uint32_t>(n_combos, 1), TSGPU_OK);
        std::vector<uint64_t> c_matched(std::max<uint32_t>(n_combos, 1), 0);
        bool ids_on_dev = false, any_wild = false;
        for (uint32_t i = 0; i < n_queries; i++) any_wild = any_wild || (status[i] == TSGPU_OK && groups[i].wildcard);
        {
            std::vector<tsgpu_kw_query> kq;
            std::vector<uint32_t> kq_of;
            for (uint32_t i = 0; i < n_queries; i++) {
                if (status[i] != TSGPU_OK) continue;
                const tsgpu_kw_query& in = combos[cfirst[i]];
                if (groups[i].wildcard) {
                    // Index::search_wildcard ranks the filter ids (every seq_id without a filter) minus the excluded ids (src/index.cpp:6674-6676)
                    // (a doc-range shard of a group ranks the ids it OWNS, like tsgpu_wildcard_search_batch: context options doc_range_lo / _hi)
                    const bool ranged = ctx->doc_range_set;
                    const uint32_t own_lo = ranged ? ctx->doc_range_lo : 0u, own_hi = ranged ? std::min(ctx->doc_range_hi, ctx->num_docs) : ctx->num_docs;
                    if (in.n_filter == 0 && in.n_excluded == 0 && !ranged) { iota[i] = 1; any_iota = true; num_matched[i] = ctx->num_docs; continue; }
                    std::vector<uint32_t>& w = wild_ids[i];
                    uint32_t j0 = 0, n = in.n_filter ? in.n_filter : own_hi;
                    if (in.n_filter && ranged) {
                        j0 = (uint32_t)(std::lower_bound(in.filter_ids, in.filter_ids + in.n_filter, own_lo) - in.filter_ids);
                        n = (uint32_t)(std::lower_bound(in.filter_ids, in.filter_ids + in.n_filter, own_hi) - in.filter_ids);
                    } else if (!in.n_filter) j0 = std::min(own_lo, own_hi);
                    w.reserve(n - j0);
                    uint32_t e = 0;
                    for (uint32_t j = j0; j < n; j++) {
                        const uint32_t id = in.n_filter ? in.filter_ids[j] : j;
                        while (e < in.n_excluded && in.excluded_ids[e] < id) e++;
                        if (e < in.n_excluded && in.excluded_ids[e] == id) continue;
                        w.push_back(id);
                    }
                    num_matched[i] = w.size();
                    continue;
                }
                for (uint32_t c = cfirst[i]; c < cfirst[i + 1]; c++) {
                    kw_index[c] = (uint32_t)kq.size();
                    kq.push_back(combos[c]);
                    kq.back().topster_size = 1;          // the pass is run for its ids and counters; the Topster it fills is not read
                    kq_of.push_back(c);
                }
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
                // a batch without wildcard queries: the ids go straight into this call's device buffer (no download + upload) unless a query's ids need the host's sort
                if (!ctx->groupby) ctx->groupby = new GroupByScratch;
                struct NoCoalesce { bool old; NoCoalesce() : old(tsgpu::tls_no_coalesce()) { tsgpu::tls_no_coalesce() = true; } ~NoCoalesce() { tsgpu::tls_no_coalesce() = old; } } nc;
                std::vector<uint16_t> pe;
                if (shard && shard->present_elsewhere) { pe.resize(nk); for (uint32_t j = 0; j < nk; j++) pe[j] = shard->present_elsewhere[kq_of[j]]; }
                const int rc = kw_dispatch(ctx, kq.data(), nk, &th, false, &raw, any_wild ? nullptr : &ctx->groupby->ids_dev, &ids_on_dev, pe.empty() ? nullptr : pe.data());
                idl.reset(raw);
                if (rc != TSGPU_OK) return rc;
                for (uint32_t j = 0; j < nk; j++) { c_status[kq_of[j]] = t_st[j]; c_matched[kq_of[j]] = t_nm[j]; }
                for (uint32_t i = 0; i < n_queries; i++) {
                    if (status[i] != TSGPU_OK || groups[i].wildcard) continue;
                    for (uint32_t c = cfirst[i]; c < cfirst[i + 1]; c++) {
                        if (status[i] == TSGPU_OK && c_status[c] != TSGPU_OK) status[i] = c_status[c];       // a failing combination fails its user query
                        cutoff[i] = cutoff[i] || t_co[kw_index[c]];
                    }
                    num_matched[i] = c_matched[cfirst[i + 1] - 1];                                         // the reference assigns num_keyword_matches per pass: the last one stays
                }
            }
        }
        const uint64_t t_ids = now_us();
        // ---- layout ----
        uint64_t n_items = 0, n_slots = 0, n_blocks = 0, n_sblocks = 0;
        std::vector<unsigned long long> combo_begin((size_t)n_combos + 1, 0);
        std::vector<uint32_t> qidx_of_combo(std::max<uint32_t>(n_combos, 1), 0);
        for (uint32_t i = 0; i < n_queries; i++) {
            GbQuery& g = gq[i];
            g.item_begin = n_items; g.tab_off = n_slots;
            g.run = status[i] == TSGPU_OK ? 1 : 0;
            g.iota = iota[i];
            uint64_t n = 0, gap = 0;
            uint32_t matched_before = 0;                  // searched_queries.size() at the time of a pass: the earlier combinations that matched anything (:5580-5585)
            for (uint32_t c = cfirst[i]; c < cfirst[i + 1]; c++) {
                combo_begin[c] = n_items + n + gap;
                qidx_of_combo[c] = (shard && shard->pass_mask) ? c - cfirst[i] : matched_before;      // (shard form: the hit's PASS travels; "matched anything" is decided over all shards)
                uint64_t nc = 0;
                if (g.run) nc = iota[i] ? ctx->num_docs : (groups[i].wildcard ? wild_ids[i].size() : tsgpu_id_lists_count(idl.get(), kw_index[c]));
                // a user query that FAILED after the id pass (one of its combinations ran out of time, ...) while another of its combinations succeeded: with the
                // ids on the device, kw_dispatch packed the successful combination's ids into ids_dev like everybody else's. They are a GAP of the item space
                // (nobody reads them: the query has no items), or every later query's combo_begin / item_begin would be short by that many ids
                else if (ids_on_dev && kw_index[c] != 0xFFFFFFFFu) gap += tsgpu_id_lists_count(idl.get(), kw_index[c]);
                n += nc;
                if (nc) { matched_before++; if (shard && shard->pass_mask) shard->pass_mask[i] |= 1u << (c - cfirst[i]); }
            }
            if (n > 0x7FFFFFFFull) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_keyword_search_grouped_batch: more than 2^31 matched ids in one query");
            g.n_items = (uint32_t)n;
            uint64_t size = 64;
            while (size < 2 * n) size <<= 1;
            g.tab_mask = (uint32_t)(size - 1);
            g.first_block = n_blocks;
            n_blocks += (n + GB_THREADS - 1) / GB_THREADS;
            g.first_sblock = (uint32_t)n_sblocks;
            n_sblocks += (n + GB_SCATTER_ITEMS - 1) / GB_SCATTER_ITEMS;
            n_items += n + gap;
            n_slots += size + 1;                          // + the slot of the key ~0
        }
        combo_begin[n_combos] = n_items;
        // second passes: the chunk work lists of the big groups and their partial top-L buffers
        uint64_t n_pw = 0, n_pb = 0;
        for (uint32_t i = 0; i < n_queries; i++) {
            GbQuery& g = gq[i];
            g.pw_begin = (uint32_t)n_pw; g.pbuf_off = n_pb;
            g.pw_cap = (g.run && !g.first_pass) ? g.n_items / GB_CHUNK + g.n_items / GB_BIG + 1 : 0;
            n_pw += g.pw_cap;
            n_pb += (uint64_t)g.pw_cap * g.group_limit;
        }
        if (n_pw > 0x7FFFFFFFull) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_keyword_search_grouped_batch: too many matched ids in one batch");
        if (n_blocks > 0x1FFFFFFFull) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_keyword_search_grouped_batch: too many matched ids in one batch");
        if (n_slots * 20 + n_items * 48 > (64ull << 30)) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_keyword_search_grouped_batch: the group tables of this batch exceed 64 GiB; split it");
        if (!ctx->groupby) ctx->groupby = new GroupByScratch;
        GroupByScratch& S = *ctx->groupby;
        const uint32_t gs = gout->g_stride, ks = out->k_stride;
        const size_t n_out = (size_t)n_queries * ks, n_g = (size_t)n_queries * gs;
        const uint64_t ni = std::max<uint64_t>(n_items, 1);
        GbLayout Lin, Lff, Lzero, Lout, Lwork;
        const uint32_t ncz = std::max<uint32_t>(n_combos, 1);
        const size_t i_gq = Lin.take(sizeof(GbQuery) * n_queries), i_qd = Lin.take(sizeof(KwQueryDev) * ncz), i_mf = Lin.take(sizeof(KwQueryMF) * ncz),
                     i_cb = Lin.take(((size_t)n_combos + 1) * 8), i_qx = Lin.take((size_t)ncz * 4),
                     i_fk = Lin.take(((shard && shard->forced_begin) ? (size_t)shard->forced_begin[n_queries] : 0) * 8 + 8), i_ids = Lin.take(ids_on_dev ? 16 : ni * 4);
        const size_t f_hkey = Lff.take(n_slots * 8), f_hbest = Lff.take(n_slots * 4), f_hrank = Lff.take(n_slots * 4),
                     f_dkey = Lff.take(any_dedupe ? n_slots * 4 : 16), f_dbest = Lff.take(any_dedupe ? n_slots * 4 : 16);      // the document tables of the second passes over several combinations
        const size_t z_hcount = Lzero.take(n_slots * 4), z_gcount = Lzero.take((size_t)n_queries * 4), z_ticket = Lzero.take(n_g * 4);
        const size_t o_nhits = Lout.take((size_t)n_queries * 4), o_ng = Lout.take((size_t)n_queries * 4), o_gtot = Lout.take((size_t)n_queries * 8),
                     o_hist = Lout.take((size_t)n_queries * GB_LOGLOG_HIST * 4), o_gdkey = Lout.take(n_g * 8), o_gsize = Lout.take(n_g * 4), o_gfound = Lout.take(n_g * 4),
                     o_keys = Lout.take(n_out * 8), o_scores = Lout.take(n_out * 24), o_msi = Lout.take(n_out);
        const size_t end_required = Lout.at;
        const size_t o_tm = Lout.take(n_out * 8);
        const size_t end_tm = Lout.at;
        const size_t o_vd = Lout.take(n_out * 4);
        const size_t end_vd = Lout.at;
        const size_t o_qx = Lout.take(n_out * 4);                 // KV::query_index per hit slot (only with several combinations per query)
        const size_t w_s0 = Lwork.take(ni * 8), w_s1 = Lwork.take(ni * 8), w_s2 = Lwork.take(ni * 8), w_dkey = Lwork.take(ni * 8), w_rslot = Lwork.take(ni * 4),
                     w_members = Lwork.take(ni * 4), w_glist = Lwork.take(ni * 4), w_mofs = Lwork.take(n_g * 4), w_mcur = Lwork.take(n_g * 4), w_pass = Lwork.take(ni),
                     w_pfirst = Lwork.take(n_g * 4), w_nchunk = Lwork.take(n_g * 4), w_pwlist = Lwork.take(std::max<uint64_t>(n_pw, 1) * 4), w_pwn = Lwork.take(std::max<uint64_t>(n_pw, 1) * 4),
                     w_pwcount = Lwork.take((size_t)n_queries * 4), w_pb0 = Lwork.take(std::max<uint64_t>(n_pb, 1) * 8), w_pb1 = Lwork.take(std::max<uint64_t>(n_pb, 1) * 8),
                     w_pb2 = Lwork.take(std::max<uint64_t>(n_pb, 1) * 8), w_pbk = Lwork.take(std::max<uint64_t>(n_pb, 1) * 8);
        const size_t out_bytes = query_index ? Lout.at : (out->vector_distance ? end_vd : (out->text_match ? end_tm : end_required));      // what the caller asked for, as a prefix
        int rc;
        if ((rc = S.in.reserve(Lin.at)) || (rc = S.ff.reserve(Lff.at)) || (rc = S.zero.reserve(Lzero.at)) || (rc = S.out.reserve(Lout.at)) || (rc = S.work.reserve(Lwork.at)) ||
            (rc = S.h_out.reserve(out_bytes <= (4u << 20) ? out_bytes : o_gdkey)))
            return rc;
        // the pinned staging holds the descriptions and the id arrays that come FROM the host: not the ids the id pass left on the device, not the ids
        // gb_iota_kernel writes for q = * (40 MB of pinned memory per 10M documents otherwise)
        uint64_t host_items_end = 0;
        for (uint32_t i = 0; !ids_on_dev && i < n_queries; i++)
            if (gq[i].run && gq[i].n_items && !iota[i]) host_items_end = std::max<uint64_t>(host_items_end, (uint64_t)gq[i].item_begin + gq[i].n_items);
        if ((rc = S.h_in.reserve(i_ids + 16 + host_items_end * 4))) return rc;
        const bool want_loglog = any_first;
        if (want_loglog && (rc = S.loglog.reserve((size_t)n_queries * GB_LOGLOG_M))) return rc;
        {
            char* hin = (char*)S.h_in.p;
            memcpy(hin + i_gq, gq.data(), sizeof(GbQuery) * n_queries);
            memcpy(hin + i_qd, qd.data(), sizeof(KwQueryDev) * n_combos);
            memcpy(hin + i_mf, mf.data(), sizeof(KwQueryMF) * n_combos);
            memcpy(hin + i_cb, combo_begin.data(), ((size_t)n_combos + 1) * 8);
            memcpy(hin + i_qx, qidx_of_combo.data(), (size_t)n_combos * 4);
            if (shard && shard->forced_begin && shard->forced_begin[n_queries]) memcpy(hin + i_fk, shard->forced_keys, (size_t)shard->forced_begin[n_queries] * 8);
            for (uint32_t i = 0; !ids_on_dev && i < n_queries; i++) {
                if (!gq[i].run || gq[i].n_items == 0 || iota[i]) continue;
                if (groups[i].wildcard) { memcpy(hin + i_ids + (size_t)gq[i].item_begin * 4, wild_ids[i].data(), (size_t)gq[i].n_items * 4); continue; }
                for (uint32_t c = cfirst[i]; c < cfirst[i + 1]; c++) {       // the combinations' ascending id lists, one after the other
                    const uint64_t nc = combo_begin[c + 1] - combo_begin[c];
                    if (nc) memcpy(hin + i_ids + (size_t)combo_begin[c] * 4, tsgpu_id_lists_ids(idl.get(), kw_index[c]), (size_t)nc * 4);
                }
            }
        }
        if (ids_on_dev) TSGPU_HIP_TRY(hipMemcpyAsync(S.in.p, S.h_in.p, i_ids, hipMemcpyHostToDevice, s));      // the descriptions only: the ids are where the id pass gathered them
        else if (!any_iota) TSGPU_HIP_TRY(hipMemcpyAsync(S.in.p, S.h_in.p, i_ids + (size_t)n_items * 4, hipMemcpyHostToDevice, s));      // (pinned: the ids' one extra host copy buys a DMA at link speed)
        else {
            // the descriptions, then the id arrays of the queries that have one; the others' ids (0 .. num_docs - 1) are written by gb_iota_kernel
            TSGPU_HIP_TRY(hipMemcpyAsync(S.in.p, S.h_in.p, i_ids, hipMemcpyHostToDevice, s));
            for (uint32_t i = 0; i < n_queries; i++)
                if (gq[i].run && gq[i].n_items && !iota[i])
                    TSGPU_HIP_TRY(hipMemcpyAsync((char*)S.in.p + i_ids + (size_t)gq[i].item_begin * 4, (char*)S.h_in.p + i_ids + (size_t)gq[i].item_begin * 4, (size_t)gq[i].n_items * 4, hipMemcpyHostToDevice, s));
        }
        TSGPU_HIP_TRY(hipMemsetAsync(S.ff.p, 0xFF, Lff.at, s));
        TSGPU_HIP_TRY(hipMemsetAsync(S.zero.p, 0, Lzero.at, s));
        if (want_loglog) TSGPU_HIP_TRY(hipMemsetAsync(S.loglog.p, 0, (size_t)n_queries * GB_LOGLOG_M, s));     // (the kernel writes the non-zero register words)
        // (the output block is not cleared: a caller reads the slots n_groups / group_size / n_hits describe, and those are written)
        GbArgs a;
        char* din = (char*)S.in.p; char* dff = (char*)S.ff.p; char* dz = (char*)S.zero.p; char* dout = (char*)S.out.p; char* dw = (char*)S.work.p;
        a.gq = (const GbQuery*)(din + i_gq); a.n_queries = n_queries; a.queries = (const KwQueryDev*)(din + i_qd); a.mfs = (const KwQueryMF*)(din + i_mf);
        a.forced_keys = (const unsigned long long*)(din + i_fk);
        a.combo_begin = (const unsigned long long*)(din + i_cb); a.qidx_of_combo = (const uint32_t*)(din + i_qx); a.pass = (uint8_t*)(dw + w_pass);
        a.dkey32 = (uint32_t*)(dff + f_dkey); a.dbest = (uint32_t*)(dff + f_dbest); a.out_qidx = (uint32_t*)(dout + o_qx);
        a.n_pw = (uint32_t)n_pw; a.pw_list = (uint32_t*)(dw + w_pwlist); a.pw_count = (uint32_t*)(dw + w_pwcount); a.pw_n = (uint32_t*)(dw + w_pwn);
        a.g_pfirst = (uint32_t*)(dw + w_pfirst); a.g_nchunk = (uint32_t*)(dw + w_nchunk); a.g_ticket = (uint32_t*)(dz + z_ticket);
        a.pb_s0 = (int64_t*)(dw + w_pb0); a.pb_s1 = (int64_t*)(dw + w_pb1); a.pb_s2 = (int64_t*)(dw + w_pb2); a.pb_key = (int64_t*)(dw + w_pbk);
        a.n_items = n_items; a.ids = ids_on_dev ? S.ids_dev.as<uint32_t>() : (const uint32_t*)(din + i_ids);
        a.s0 = (int64_t*)(dw + w_s0); a.s1 = (int64_t*)(dw + w_s1); a.s2 = (int64_t*)(dw + w_s2); a.dkey = (unsigned long long*)(dw + w_dkey); a.rslot = (uint32_t*)(dw + w_rslot);
        a.hkey = (unsigned long long*)(dff + f_hkey); a.hbest = (uint32_t*)(dff + f_hbest); a.hrank = (uint32_t*)(dff + f_hrank);
        a.hcount = (uint32_t*)(dz + z_hcount); a.gcount = (uint32_t*)(dz + z_gcount);
        a.members = (uint32_t*)(dw + w_members); a.glist = (uint32_t*)(dw + w_glist); a.g_stride = gs;
        a.n_groups = (uint32_t*)(dout + o_ng); a.groups_total = (unsigned long long*)(dout + o_gtot);
        a.g_dkey = (unsigned long long*)(dout + o_gdkey); a.g_found = (uint32_t*)(dout + o_gfound); a.g_size = (uint32_t*)(dout + o_gsize);
        a.g_mofs = (uint32_t*)(dw + w_mofs); a.g_mcur = (uint32_t*)(dw + w_mcur);
        a.loglog = want_loglog ? S.loglog.as<uint8_t>() : nullptr; a.loglog_hist = (uint32_t*)(dout + o_hist);
        a.out.keys = (uint64_t*)(dout + o_keys); a.out.scores = (int64_t*)(dout + o_scores); a.out.text_match = (int64_t*)(dout + o_tm);
        a.out.vector_distance = (float*)(dout + o_vd); a.out.match_score_index = (int8_t*)(dout + o_msi); a.out.n_hits = (uint32_t*)(dout + o_nhits);
        a.out.num_matched = nullptr; a.out.off_words = nullptr; a.out.k_stride = ks;
        IndexView v = make_view(ctx, snap);
        for (int e = 0; e < 4; e++) if (!ctx->aux_ev[e]) TSGPU_HIP_TRY(hipEventCreate(&ctx->aux_ev[e]));      // (tsgpu_last_aux_timings: four marks per batch, ~2 us each)
        TSGPU_HIP_TRY(hipEventRecord(ctx->aux_ev[0], s));
        if (n_items) {
            if (any_iota) hipLaunchKernelGGL(gb_iota_kernel, dim3((uint32_t)n_blocks), dim3(GB_THREADS), 0, s, a);
            uint32_t max_lists = 0;
            for (uint32_t i = 0; i < n_queries; i++) if (gq[i].run) for (uint32_t c = cfirst[i]; c < cfirst[i + 1]; c++) max_lists = std::max(max_lists, qd[c].n_lists);
            if (max_lists <= 3) hipLaunchKernelGGL((gb_score_kernel<3>), dim3((uint32_t)n_blocks), dim3(GB_THREADS), 0, s, v, a);
            else hipLaunchKernelGGL((gb_score_kernel<KW_MAX_TOKENS>), dim3((uint32_t)(n_blocks * (GB_THREADS / 64))), dim3(64), 0, s, v, a);
            if (any_dedupe) hipLaunchKernelGGL(gb_dedupe_kernel, dim3((uint32_t)n_blocks), dim3(GB_THREADS), 0, s, a);
            hipLaunchKernelGGL(gb_insert_kernel, dim3((uint32_t)n_blocks), dim3(GB_THREADS), 0, s, a);
        }
        TSGPU_HIP_TRY(hipEventRecord(ctx->aux_ev[1], s));
        if (max_k + GB_THREADS <= 512) hipLaunchKernelGGL((gb_select_kernel<512>), dim3(n_queries), dim3(GB_THREADS), 0, s, a);
        else if (max_k + GB_THREADS <= 1024) hipLaunchKernelGGL((gb_select_kernel<1024>), dim3(n_queries), dim3(GB_THREADS), 0, s, a);
        else hipLaunchKernelGGL((gb_select_kernel<2048>), dim3(n_queries), dim3(GB_THREADS), 0, s, a);
        TSGPU_HIP_TRY(hipEventRecord(ctx->aux_ev[2], s));
        if (any_second && n_items) {
            hipLaunchKernelGGL(gb_scatter_kernel, dim3((uint32_t)n_sblocks), dim3(GB_THREADS), 0, s, a);
            const uint64_t pairs = (uint64_t)n_queries * gs;
            hipLaunchKernelGGL(gb_members_kernel, dim3((uint32_t)((pairs + GB_THREADS / 64 - 1) / (GB_THREADS / 64))), dim3(GB_THREADS), 0, s, a);
            if (n_pw) hipLaunchKernelGGL(gb_chunk_kernel, dim3((uint32_t)n_pw), dim3(GB_THREADS), 0, s, a);      // (workgroups beyond a query's chunk count leave at once)
        }
        TSGPU_HIP_TRY(hipEventRecord(ctx->aux_ev[3], s));
        TSGPU_HIP_TRY(hipGetLastError());
        uint64_t t_launched = now_us(), t_kernels = t_launched;
        if (host_timing) { TSGPU_HIP_TRY(hipStreamSynchronize(s)); t_kernels = now_us(); }
        // ---- delivery: one download + plain copies of the used extents into the caller's arrays (calls of a few queries: a dozen small copies cost 150 us);
        // a big batch's arrays go straight to the caller (two passes over 30+ MB on the host cost more than the copy calls) ----
        const bool packed_delivery = out_bytes <= (4u << 20);
        const size_t small_bytes = o_gdkey;                      // [n_hits | n_groups | groups_total | loglog_hist]
        TSGPU_HIP_TRY(hipMemcpyAsync(S.h_out.p, S.out.p, packed_delivery ? out_bytes : small_bytes, hipMemcpyDeviceToHost, s));
        if (!packed_delivery) {
            TSGPU_HIP_TRY(hipMemcpyAsync(gout->distinct_key, dout + o_gdkey, n_g * 8, hipMemcpyDeviceToHost, s));
            TSGPU_HIP_TRY(hipMemcpyAsync(gout->group_size, dout + o_gsize, n_g * 4, hipMemcpyDeviceToHost, s));
            TSGPU_HIP_TRY(hipMemcpyAsync(gout->group_found, dout + o_gfound, n_g * 4, hipMemcpyDeviceToHost, s));
            TSGPU_HIP_TRY(hipMemcpyAsync(out->keys, dout + o_keys, n_out * 8, hipMemcpyDeviceToHost, s));
            TSGPU_HIP_TRY(hipMemcpyAsync(out->scores, dout + o_scores, n_out * 24, hipMemcpyDeviceToHost, s));
            if (out->match_score_index) TSGPU_HIP_TRY(hipMemcpyAsync(out->match_score_index, dout + o_msi, n_out, hipMemcpyDeviceToHost, s));
            if (out->text_match) TSGPU_HIP_TRY(hipMemcpyAsync(out->text_match, dout + o_tm, n_out * 8, hipMemcpyDeviceToHost, s));
            if (out->vector_distance) TSGPU_HIP_TRY(hipMemcpyAsync(out->vector_distance, dout + o_vd, n_out * 4, hipMemcpyDeviceToHost, s));
            if (query_index) TSGPU_HIP_TRY(hipMemcpyAsync(query_index, dout + o_qx, n_out * 4, hipMemcpyDeviceToHost, s));
        }
        if (gout->loglog_registers) {
            if (want_loglog) TSGPU_HIP_TRY(hipMemcpyAsync(gout->loglog_registers, S.loglog.p, (size_t)n_queries * GB_LOGLOG_M, hipMemcpyDeviceToHost, s));
            else memset(gout->loglog_registers, 0, (size_t)n_queries * GB_LOGLOG_M);
        }
        TSGPU_HIP_TRY(hipStreamSynchronize(s));
        const uint64_t t_delivered = now_us();
        {
            float fold = 0, sel = 0, all = 0;
            (void)hipEventElapsedTime(&fold, ctx->aux_ev[0], ctx->aux_ev[1]); (void)hipEventElapsedTime(&sel, ctx->aux_ev[1], ctx->aux_ev[2]); (void)hipEventElapsedTime(&all, ctx->aux_ev[0], ctx->aux_ev[3]);
            std::lock_guard<std::mutex> tl(ctx->tm_mu);
            tsgpu_aux_timings& A = ctx->aux_timings;
            A.gb_id_pass_ms = (float)((t_ids - t_enter) * 1e-3); A.gb_kernels_ms = all; A.gb_fold_ms = fold; A.gb_select_ms = sel;
            A.gb_matched_ids = n_items; A.gb_table_slots = n_slots; A.gb_algorithmic_bytes = n_items * 36ull + n_slots * 20ull;
        }
        if (host_timing)
            fprintf(stderr, "[tsgpu] grouped batch %u queries, %llu matched ids, %llu slots: id pass %llu us, layout+upload+launch %llu us, kernels %llu us, delivery %llu us\n", n_queries,
                    (unsigned long long)n_items, (unsigned long long)n_slots, (unsigned long long)(t_ids - t_enter), (unsigned long long)(t_launched - t_ids),
                    (unsigned long long)(t_kernels - t_launched), (unsigned long long)(t_delivered - t_kernels));
        {
            const char* ho = (const char*)S.h_out.p;
            memcpy(out->n_hits, ho + o_nhits, (size_t)n_queries * 4);
            memcpy(gout->n_groups, ho + o_ng, (size_t)n_queries * 4);
            if (gout->groups_total) memcpy(gout->groups_total, ho + o_gtot, (size_t)n_queries * 8);
            // a caller reads the slots its counts describe: copy the rows' used extents, not the strides
            for (uint32_t i = 0; packed_delivery && i < n_queries; i++) {
                if (status[i] != TSGPU_OK) continue;
                const size_t ng = gout->n_groups[i], ext = groups[i].first_pass ? out->n_hits[i] : ng * groups[i].group_limit;
                const size_t go = (size_t)i * gs, ro = (size_t)i * ks;
                memcpy(gout->distinct_key + go, ho + o_gdkey + go * 8, ng * 8);
                memcpy(gout->group_size + go, ho + o_gsize + go * 4, ng * 4);
                memcpy(gout->group_found + go, ho + o_gfound + go * 4, ng * 4);
                memcpy(out->keys + ro, ho + o_keys + ro * 8, ext * 8);
                memcpy(out->scores + ro * 3, ho + o_scores + ro * 24, ext * 24);
                if (out->match_score_index) memcpy(out->match_score_index + ro, ho + o_msi + ro, ext);
                if (out->text_match) memcpy(out->text_match + ro, ho + o_tm + ro * 8, ext * 8);
                if (out->vector_distance) memcpy(out->vector_distance + ro, ho + o_vd + ro * 4, ext * 4);
                if (query_index) memcpy(query_index + ro, ho + o_qx + ro * 4, ext * 4);
            }
        }
        const uint32_t* hist_host = (const uint32_t*)((const char*)S.h_out.p + o_hist);
        for (uint32_t i = 0; i < n_queries; i++) {
            out->status[i] = status[i];
            if (out->num_matched) out->num_matched[i] = status[i] == TSGPU_OK ? num_matched[i] : 0;
            if (out->search_cutoff) out->search_cutoff[i] = cutoff[i];
            if (status[i] != TSGPU_OK) { out->n_hits[i] = 0; gout->n_groups[i] = 0; if (gout->groups_total) gout->groups_total[i] = 0; }
            if (gout->groups_count) {
                uint64_t card = 0;
                if (status[i] == TSGPU_OK && groups[i].first_pass && !gq[i].forced && !gb_loglog_cardinality_hist(hist_host + (size_t)i * GB_LOGLOG_HIST, &card)) {
                    std::vector<uint8_t> regs(GB_LOGLOG_M);      // (a register above 38: the ordered sum over the registers themselves)
                    TSGPU_HIP_TRY(hipMemcpy(regs.data(), S.loglog.as<uint8_t>() + (size_t)i * GB_LOGLOG_M, GB_LOGLOG_M, hipMemcpyDeviceToHost));
                    card = gb_loglog_cardinality(regs.data());
                }
                gout->groups_count[i] = card;
            }
        }
        if (ids_out) {
            // all_result_ids per user query: the union of its combinations' matched ids, ascending (id_buff -> sort + unique + or_scalar, src/index.cpp:5565-5578)
            std::unique_ptr<tsgpu_id_lists> il(new tsgpu_id_lists);
            std::vector<uint32_t> flat(n_items);
            if (ids_on_dev && n_items) TSGPU_HIP_TRY(hipMemcpy(flat.data(), S.ids_dev.p, (size_t)n_items * 4, hipMemcpyDeviceToHost));
            else if (n_items) {
                for (uint32_t i = 0; i < n_queries; i++) {
                    if (!gq[i].n_items) continue;
                    if (iota[i]) { for (uint32_t j = 0; j < gq[i].n_items; j++) flat[gq[i].item_begin + j] = j; continue; }
                    memcpy(flat.data() + gq[i].item_begin, (const char*)S.h_in.p + i_ids + (size_t)gq[i].item_begin * 4, (size_t)gq[i].n_items * 4);
                }
            }
            il->begin.assign((size_t)n_queries + 1, 0);
            for (uint32_t i = 0; i < n_queries; i++) {
                uint32_t* b = flat.data() + gq[i].item_begin;
                uint32_t n = gq[i].n_items;
                if (gq[i].n_combos > 1 && n) { std::sort(b, b + n); n = (uint32_t)(std::unique(b, b + n) - b); }
                il->ids.insert(il->ids.end(), b, b + n);
                il->begin[i + 1] = il->ids.size();
            }
            *ids_out = il.release();
        }
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_keyword_search_grouped_batch: host allocation failed"); }
    return ok();
}

namespace tsgpu {
struct GbRequest : ParkedRequest {
    const tsgpu_kw_query* q = nullptr;
    const tsgpu_group_by* g = nullptr;
    tsgpu_hits* out = nullptr;
    tsgpu_grouped_hits* gout = nullptr;
    tsgpu_id_lists* ids = nullptr;                   // non-null: the caller wants its matched ids
};
}

// The reference reaches the grouped seam like the plain one: once per query, from many request threads (src/index.cpp:3488). Small concurrent calls are
// parked in a combiner (tsgpu_batcher.h) and run as ONE grouped batch; every caller gets its own slice, id list and status (a caller whose strides are too
// small for its own request fails alone).
static int gb_coalesced(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, const tsgpu_group_by* groups, uint32_t n_queries, tsgpu_hits* out, tsgpu_grouped_hits* gout,
                        tsgpu_id_lists** ids_out) {
    std::unique_ptr<tsgpu_id_lists> lists;
    if (ids_out) { lists.reset(new (std::nothrow) tsgpu_id_lists); if (!lists) return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_keyword_search_grouped_batch: host allocation failed"); }
    GbRequest me;
    me.units = n_queries; me.q = queries; me.g = groups; me.out = out; me.gout = gout; me.ids = lists.get();
    typedef std::unique_lock<std::mutex> Lock;
    auto acquire = [&]() { return std::unique_ptr<Lock>(new Lock(ctx->mu)); };
    const uint32_t round_cap = std::max<uint32_t>(ctx->batch_round_queries, n_queries);
    auto pick = [&](std::vector<GbRequest*>& pending, std::vector<GbRequest*>& round) {
        uint32_t units = 0;
        size_t take = 0;
        while (take < pending.size() && (take == 0 || units + pending[take]->units <= round_cap)) units += pending[take++]->units;
        round.assign(pending.begin(), pending.begin() + take);
        pending.erase(pending.begin(), pending.begin() + take);
    };
    auto exec = [&](std::vector<GbRequest*>& round, std::unique_ptr<Lock>&) {
        int rc = TSGPU_OK;
        std::string err;
        try {
            uint32_t total = 0, KS = 1, GS = 1;
            bool want_ids = false, want_regs = false;
            for (GbRequest* r : round) {
                total += r->units; KS = std::max(KS, r->out->k_stride); GS = std::max(GS, r->gout->g_stride);
                want_ids = want_ids || r->ids != nullptr; want_regs = want_regs || r->gout->loglog_registers != nullptr;
            }
            std::vector<tsgpu_kw_query> q(total);
            std::vector<tsgpu_group_by> g(total);
            uint32_t at = 0;
            for (GbRequest* r : round) {
                memcpy(q.data() + at, r->q, (size_t)r->units * sizeof(tsgpu_kw_query)); memcpy(g.data() + at, r->g, (size_t)r->units * sizeof(tsgpu_group_by));
                at += r->units;
            }
            const size_t slots = (size_t)total * KS, gslots = (size_t)total * GS;
            if (!ctx->groupby) ctx->groupby = new GroupByScratch;
            GroupByScratch& S = *ctx->groupby;
            auto grow = [](auto& v, size_t n) { if (v.size() < n) v.resize(n); };
            grow(S.c_keys, slots); grow(S.c_nm, total); grow(S.c_dk, gslots); grow(S.c_gtot, total); grow(S.c_gcnt, total); grow(S.c_scores, slots * 3); grow(S.c_tm, slots);
            grow(S.c_vd, slots); grow(S.c_msi, slots); grow(S.c_nh, total); grow(S.c_ng, total); grow(S.c_gsz, gslots); grow(S.c_gfound, gslots); grow(S.c_st, total); grow(S.c_co, total);
            if (want_regs) grow(S.c_regs, (size_t)total * GB_LOGLOG_M);
            auto &keys = S.c_keys, &nm = S.c_nm, &dk = S.c_dk, &gtot = S.c_gtot, &gcnt = S.c_gcnt;
            auto &scores = S.c_scores, &tm = S.c_tm; auto& vd = S.c_vd; auto& msi = S.c_msi;
            auto &nh = S.c_nh, &ng = S.c_ng, &gsz = S.c_gsz, &gfound = S.c_gfound; auto &st = S.c_st, &co = S.c_co; auto& regs = S.c_regs;
            tsgpu_hits h;
            memset(&h, 0, sizeof h);
            h.mem = TSGPU_MEM_HOST; h.k_stride = KS;
            h.keys = keys.data(); h.scores = scores.data(); h.text_match = tm.data(); h.vector_distance = vd.data(); h.match_score_index = msi.data();
            h.n_hits = nh.data(); h.num_matched = nm.data(); h.status = st.data(); h.search_cutoff = co.data();
            tsgpu_grouped_hits gh;
            memset(&gh, 0, sizeof gh);
            gh.g_stride = GS; gh.n_groups = ng.data(); gh.distinct_key = dk.data(); gh.group_size = gsz.data(); gh.group_found = gfound.data();
            gh.groups_total = gtot.data(); gh.groups_count = gcnt.data(); gh.loglog_registers = want_regs ? regs.data() : nullptr;
            tsgpu_id_lists* all_raw = nullptr;
            std::vector<uint32_t> cf((size_t)total + 1);
            for (uint32_t i = 0; i <= total; i++) cf[i] = i;     // one combination per query
            rc = gb_batch_locked(ctx, q.data(), cf.data(), g.data(), total, &h, &gh, nullptr, want_ids ? &all_raw : nullptr);
            std::unique_ptr<tsgpu_id_lists> all_ids(all_raw);
            if (rc != TSGPU_OK) err = tls_error();
            else {
                at = 0;
                for (GbRequest* r : round) {
                    tsgpu_hits& o = *r->out;
                    tsgpu_grouped_hits& og = *r->gout;
                    for (uint32_t i = 0; i < r->units; i++) {
                        const uint32_t gi = at + i;
                        int32_t s_ = st[gi];
                        uint32_t groups_n = ng[gi], hits_n = nh[gi];
                        size_t extent = r->g[i].first_pass ? hits_n : (size_t)groups_n * r->g[i].group_limit;       // second pass: group r at slot r * group_limit
                        if (s_ == TSGPU_OK && (extent > o.k_stride || groups_n > og.g_stride)) { s_ = TSGPU_ERR_INVALID; groups_n = 0; hits_n = 0; extent = 0; }   // this caller's strides are too small
                        if (s_ != TSGPU_OK) { groups_n = 0; hits_n = 0; extent = 0; }
                        o.status[i] = s_; o.n_hits[i] = hits_n;
                        if (o.num_matched) o.num_matched[i] = s_ == TSGPU_OK ? nm[gi] : 0;
                        if (o.search_cutoff) o.search_cutoff[i] = co[gi];
                        const size_t src = (size_t)gi * KS, dst = (size_t)i * o.k_stride;
                        memcpy(o.keys + dst, keys.data() + src, extent * 8);
                        memcpy(o.scores + dst * 3, scores.data() + src * 3, extent * 24);
                        if (o.text_match) memcpy(o.text_match + dst, tm.data() + src, extent * 8);
                        if (o.vector_distance) memcpy(o.vector_distance + dst, vd.data() + src, extent * 4);
                        if (o.match_score_index) memcpy(o.match_score_index + dst, msi.data() + src, extent);
                        const size_t gsrc = (size_t)gi * GS, gdst = (size_t)i * og.g_stride;
                        og.n_groups[i] = groups_n;
                        memcpy(og.distinct_key + gdst, dk.data() + gsrc, (size_t)groups_n * 8);
                        memcpy(og.group_size + gdst, gsz.data() + gsrc, (size_t)groups_n * 4);
                        memcpy(og.group_found + gdst, gfound.data() + gsrc, (size_t)groups_n * 4);
                        if (og.groups_total) og.groups_total[i] = s_ == TSGPU_OK ? gtot[gi] : 0;
                        if (og.groups_count) og.groups_count[i] = s_ == TSGPU_OK ? gcnt[gi] : 0;
                        if (og.loglog_registers) memcpy(og.loglog_registers + (size_t)i * GB_LOGLOG_M, regs.data() + (size_t)gi * GB_LOGLOG_M, GB_LOGLOG_M);
                    }
                    if (r->ids) {
                        r->ids->begin.resize((size_t)r->units + 1);
                        const uint64_t b0 = all_ids->begin[at];
                        for (uint32_t i = 0; i <= r->units; i++) r->ids->begin[i] = all_ids->begin[at + i] - b0;
                        r->ids->ids.assign(all_ids->ids.begin() + b0, all_ids->ids.begin() + all_ids->begin[at + r->units]);
                    }
                    at += r->units;
                }
            }
        } catch (const std::bad_alloc&) { rc = TSGPU_ERR_NO_MEMORY; err = "tsgpu_keyword_search_grouped_batch: host allocation failed"; }
        for (GbRequest* r : round) { r->rc = rc; r->err = err; }
    };
    ctx->gb_comb.run(me, ctx->gb_callers, ctx->batch_window_us, acquire, pick, exec);
    if (me.rc != TSGPU_OK) return fail(me.rc, me.err);
    if (ids_out) *ids_out = lists.release();
    return ok();
}

namespace tsgpu {
uint64_t gb_registers_cardinality(const uint8_t* regs) { return gb_loglog_cardinality(regs); }
int gb_shard_batch(tsgpu_ctx* ctx, const tsgpu_kw_query* combos, const uint32_t* cfirst, const tsgpu_group_by* groups, uint32_t n_queries, tsgpu_hits* out, tsgpu_grouped_hits* gout,
                   uint32_t* query_index, const GbShard* shard) {
    if (!ctx || !out || !gout || !cfirst || (n_queries && (!combos || !groups))) return fail(TSGPU_ERR_INVALID, "tsgpu_group_keyword_search_grouped_batch: NULL argument");
    if (n_queries == 0) return ok();
    std::lock_guard<std::mutex> lk(ctx->mu);
    return gb_batch_locked(ctx, combos, cfirst, groups, n_queries, out, gout, query_index, nullptr, shard);
}
}  // namespace tsgpu

extern "C" {

int tsgpu_keyword_search_grouped_batch(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, const tsgpu_group_by* groups, uint32_t n_queries,
                                       tsgpu_hits* out, tsgpu_grouped_hits* gout, tsgpu_id_lists** ids_out) {
    if (ids_out) *ids_out = nullptr;
    if (!ctx || !out || !gout || (n_queries && (!queries || !groups))) return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_search_grouped_batch: NULL argument");
    if (n_queries == 0) return ok();
    if (out->mem != TSGPU_MEM_HOST) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_keyword_search_grouped_batch: host output arrays only");
    if (!out->keys || !out->scores || !out->n_hits || !out->status) return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_search_grouped_batch: keys / scores / n_hits / status are required");
    if (!gout->n_groups || !gout->distinct_key || !gout->group_size || !gout->group_found || gout->g_stride == 0 || out->k_stride == 0)
        return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_search_grouped_batch: n_groups / distinct_key / group_size / group_found and the strides are required");
    struct CallerCount { std::atomic<int>& c; explicit CallerCount(std::atomic<int>& x) : c(x) { c.fetch_add(1); } ~CallerCount() { c.fetch_sub(1); } } cc(ctx->gb_callers);
    if (n_queries <= ctx->batch_max_queries && ctx->gb_callers.load() > 1) return gb_coalesced(ctx, queries, groups, n_queries, out, gout, ids_out);
    std::vector<uint32_t> cf((size_t)n_queries + 1);
    for (uint32_t i = 0; i <= n_queries; i++) cf[i] = i;         // one combination per query
    std::lock_guard<std::mutex> lk(ctx->mu);
    return gb_batch_locked(ctx, queries, cf.data(), groups, n_queries, out, gout, nullptr, ids_out);
}

int tsgpu_keyword_search_grouped_candidates_batch(tsgpu_ctx* ctx, const tsgpu_kw_query* combos, const uint32_t* group_begin, const tsgpu_group_by* groups, uint32_t n_user,
                                                  tsgpu_hits* out, tsgpu_grouped_hits* gout, uint32_t* query_index, tsgpu_id_lists** ids_out) {
    if (ids_out) *ids_out = nullptr;
    if (!ctx || !out || !gout || (n_user && (!combos || !groups || !group_begin))) return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_search_grouped_candidates_batch: NULL argument");
    if (n_user == 0) return ok();
    if (out->mem != TSGPU_MEM_HOST) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_keyword_search_grouped_candidates_batch: host output arrays only");
    if (!out->keys || !out->scores || !out->n_hits || !out->status) return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_search_grouped_candidates_batch: keys / scores / n_hits / status are required");
    if (!gout->n_groups || !gout->distinct_key || !gout->group_size || !gout->group_found || gout->g_stride == 0 || out->k_stride == 0)
        return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_search_grouped_candidates_batch: n_groups / distinct_key / group_size / group_found and the strides are required");
    if (group_begin[0] != 0) return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_search_grouped_candidates_batch: group_begin[0] must be 0");
    for (uint32_t u = 0; u < n_user; u++) if (group_begin[u + 1] < group_begin[u]) return fail(TSGPU_ERR_INVALID, "tsgpu_keyword_search_grouped_candidates_batch: group_begin must be non-decreasing");
    std::lock_guard<std::mutex> lk(ctx->mu);
    return gb_batch_locked(ctx, combos, group_begin, groups, n_user, out, gout, query_index, ids_out);
}

}  // extern "C"
