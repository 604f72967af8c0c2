// kw_find2.hip.h — the "find" half of the two-kernel keyword form with TWO driver blocks per iteration (included by kw_kernels.hip.h).
//
// kw_search_kernel<.., DEFER = true> spends ~43 % of a work item's cycles on per-BLOCK costs (the plan for the next block, the wait for
// the driver block's metadata, stage 0, tile store, barrier, compaction, queue bookkeeping; TSGPU_PROF profile in DESIGN.md §5) and its
// busiest issue port is the CU's one scalar unit. Here one iteration serves 512 driver ids: ONE plan (the run of second-list blocks under
// both driver blocks), ONE tile, ONE barrier and ONE compaction barrier per pair; every thread carries two candidates — slot t of either
// block — whose block / slot searches are two independent dependency chains. Everything else (window registers, tile pipeline, hit
// records, third-list probes, deadline check) is the find kernel's; both kernels leave identical hit records (ascending ids per work item).
#pragma once

#ifndef TSGPU_F2_SPAN
#define TSGPU_F2_SPAN 12
#endif
#ifndef TSGPU_F2_FAST
#define TSGPU_F2_FAST 1
#endif
static const int KW_F2_SPAN = TSGPU_F2_SPAN;                 // runs of up to this many second-list blocks: block search by v_readlane over the window registers
#ifndef TSGPU_F2_WAVES
#define TSGPU_F2_WAVES 7
#endif
#ifdef TSGPU_HIP_EMU
#define KW_F2_WAVES
#else
#define KW_F2_WAVES __attribute__((amdgpu_waves_per_eu(TSGPU_F2_WAVES)))
#endif
#ifndef TSGPU_F2_QUAD
#define TSGPU_F2_QUAD 0
#endif
static const bool KW_F2_QUAD = TSGPU_F2_QUAD != 0;           // 4-ary slot search (three samples per round) instead of binary
static const bool KW_F2_FAST = TSGPU_F2_FAST != 0;
#ifndef TSGPU_F2_MIDSLABS
#define TSGPU_F2_MIDSLABS 0
#endif
#ifndef TSGPU_F2_ROUNDS
#define TSGPU_F2_ROUNDS 0
#endif
// Runs wider than the tile (the second list holds > 8 ids per driver id there): 1 = searched in several tile rounds, 0 (default since round 4) = probed
// per candidate like broken runs. With id directories a probe of a long list is one load, and the pair loop without the multi-round code is
// a smaller kernel: find 5.78 -> 5.03 ms on the 10 000-query batch (profiles/r04/exp_kw_find_split_probe.txt).
static const bool KW_F2_ROUNDS = TSGPU_F2_ROUNDS != 0;
#ifdef TSGPU_F2_NOBAR                                       // tools/ ablation only (results are WRONG): what the barrier at the top of the pair loop costs (the compaction barrier stays: it keeps the queue counts uniform)
#define KW_F2_LOOP_BARRIER()
#else
#define KW_F2_LOOP_BARRIER() __syncthreads()
#endif           // interleaved slot searches when a wavefront's blocks are all full 16-bit blocks

// (kw_glds_slabs / kw_glds_wait — the LDS-DMA tile fill — live in kw_kernels.hip.h: the multi-field find kernel uses them, too)

// COUNT = true: the kernel also COUNTS THE BYTES IT REQUESTS (every lane adds the width of each of its own loads / DMA words / stores to one of five
// per-thread counters: driver ids, block metadata, tile DMA, third.. list probes, hit records; one wave reduction + five atomics per wave at the
// end into IndexView::touched). A second instantiation launched only under option kw_count_touched — the timed kernel carries none of it.
// Same results either way (tests/test_emu_keyword.py runs both and compares).
template <int TMAX, bool COUNT = false>
__device__ __forceinline__ void kw_find2_body(const IndexView& ix, const KwQueryDev* __restrict__ queries, const KwWorkItem* __restrict__ work, const KwPartials& part,
                                              uint32_t* __restrict__ hits_all, const uint64_t* __restrict__ hit_off, const uint32_t bid) {
    __shared__ KwSmem<TMAX, 512, false, true, true> sm;
    __shared__ uint32_t wave_cnt_b[2][KW_THREADS / 64];        // second half's wave counts (the first half uses sm.wave_cnt2)
    uint32_t* __restrict__ hits = hits_all + hit_off[bid] * (uint64_t)(TMAX + 1);
    __shared__ KwQueryDev sq;
    const uint32_t t = threadIdx.x;
    const KwWorkItem wi = work[bid];
    uint32_t cb_ids = 0, cb_meta = 0, cb_tile = 0, cb_probe = 0, cb_rec = 0;      // COUNT only: bytes THIS lane requested
    {
        const uint32_t* src = (const uint32_t*)(queries + wi.query);
        uint32_t* dst = (uint32_t*)&sq;
        for (uint32_t i = t; i < sizeof(KwQueryDev) / 4; i += KW_THREADS) { dst[i] = src[i]; if constexpr (COUNT) cb_meta += 4; }
    }
    if (t == 0) { sm.q1_cnt = 0; sm.qf_cnt = 0; sm.tk_cnt = 0; sm.have_thr = 0; sm.n_match = 0; sm.n_emit = 0; sm.off_words = 0; }
    __syncthreads();
    const KwQueryDev& q = sq;
    const uint32_t T = q.n_lists;
    const ListDesc dA = ix.lists[q.list[q.probe_order[0]]];
    const ListDesc dB = ix.lists[q.list[q.probe_order[T >= 2 ? 1 : 0]]];
    const uint32_t* __restrict__ blB = ix.blk_last + dB.blk_base;
    const BlockIds* __restrict__ biA = ix.blk_ids + dA.blk_base;
    const BlockIds* __restrict__ biB = ix.blk_ids + dB.blk_base;
    const uint32_t* __restrict__ idwA = ix.ids_payload + dA.ids_base;
    const uint32_t* __restrict__ idwB = ix.ids_payload + dB.ids_base;
    const uint32_t lane = t & 63, wave = t >> 6;
    const BlockIds PAD = {0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u};
    if constexpr (COUNT) { if (lane == 0) cb_meta += 2 * (uint32_t)sizeof(ListDesc) + 16; }      // (uniform loads: once per wave) two descriptors + the work item

    auto load_id_raw = [&](const BlockIds& m, uint32_t slot) -> uint32_t {
        const uint32_t n = m.n_ids_bits & 0xFFFF;
        const uint32_t s2 = slot < n ? slot : 0;
        const uint32_t* __restrict__ w = idwA + m.ids_woff;
        if constexpr (COUNT) { if (slot < n) cb_ids += (m.n_ids_bits >> 16) == 16 ? 2u : 4u; }
        return (m.n_ids_bits >> 16) == 16 ? (uint32_t)((const uint16_t*)w)[s2] : w[s2];
    };
    const uint32_t nB = T >= 2 ? dB.n_blocks : 0u;       // (no second list: every window slot is padding)
    auto load_window = [&](uint32_t base) -> BlockIds {
        if constexpr (COUNT) { if (base + lane < nB) cb_meta += 16; }
        return base + lane < nB ? biB[base + lane] : PAD;
    };
    // Driver-list metadata: lane j of every wave holds BlockIds[abase + j] — ONE vector load serves 32 pairs (most work items need only the
    // prologue's), a block's record is four v_readlane. (As per-pair loads they were uniform, so hipcc wanted them in SGPRs at once: a
    // global_load + s_waitcnt vmcnt(0) at the END of every iteration — a full memory round trip exposed per pair, which also drained the next
    // pair's tile DMA and driver ids before the iteration could end. TSGPU_PROF: 14 % of a work item's time.)
    uint32_t abase = wi.blk_begin;
    auto load_awin = [&](uint32_t base) -> BlockIds { const uint32_t bb = base + lane; if constexpr (COUNT) { if (bb < wi.blk_end) cb_meta += 16; } return biA[bb < wi.blk_end ? bb : wi.blk_end - 1]; };
    BlockIds awin = load_awin(abase);
    auto meta = [&](uint32_t bb) -> BlockIds {          // bb: uniform, abase <= min(bb, blk_end - 1) < abase + 64
        const int j = (int)((bb < wi.blk_end ? bb : wi.blk_end - 1) - abase);
        BlockIds m;
        m.first_id = (uint32_t)__builtin_amdgcn_readlane((int)awin.first_id, j);
        m.last_id = (uint32_t)__builtin_amdgcn_readlane((int)awin.last_id, j);
        m.ids_woff = (uint32_t)__builtin_amdgcn_readlane((int)awin.ids_woff, j);
        m.n_ids_bits = (uint32_t)__builtin_amdgcn_readlane((int)awin.n_ids_bits, j);
        return m;
    };

    uint32_t wbase = 0, wver = 0;
    BlockIds win = load_window(0), nxt = load_window(32);
    uint32_t win_dirty = 1;                               // (an integer, not a bool: a uniform bool lives in an SGPR PAIR as a lane mask)
    struct Plan { uint32_t mode, rlo, rhi, w_begin, W, ver, base, buf; };   // mode: 0 tile, 1 tile in several rounds, 2 wide / broken run (probe), 3 exhausted, 4 no second list
    constexpr int PIPE_WORDS = KW_FIND_PIPE_WORDS;
    constexpr int TILE_WORDS = KW_FIND_TILE_WORDS;
    static_assert(TILE_WORDS == 2 * PIPE_WORDS * KW_THREADS, "two tile buffers: one searched, one being filled");
    constexpr int HALF = PIPE_WORDS * KW_THREADS;
    uint32_t tbuf = 0;                                   // the buffer the last DMA went to
    // how the driver ids in [lo_id, hi_id] meet the second list; mode 0 also requests the tile
    auto make_plan = [&](uint32_t lo_id, uint32_t hi_id) -> Plan {
        Plan P; P.mode = 4; P.rlo = P.rhi = P.w_begin = P.W = 0; P.ver = wver; P.base = wbase; P.buf = 0;
        if (T < 2) return P;
        unsigned long long mk = __ballot(win.last_id >= lo_id ? 1 : 0);
        if (mk != 0 && (uint32_t)__builtin_ctzll(mk) >= 32) {          // cursor entered the upper half: slide by 32 blocks
            wbase += 32; win = nxt; nxt = load_window(wbase + 32); win_dirty = 1;
            mk = __ballot(win.last_id >= lo_id ? 1 : 0);
        }
        if (mk == 0) {                                                   // all 64 blocks end before lo_id: uniform search, re-centre
            uint32_t lo = wbase + 64, hi = dB.n_blocks;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if constexpr (COUNT) { if (lane == 0) cb_meta += 4; } if (blB[mid] >= lo_id) hi = mid; else lo = mid + 1; }
            wbase = lo; win = load_window(wbase); nxt = load_window(wbase + 32); win_dirty = 1;
            mk = __ballot(win.last_id >= lo_id ? 1 : 0);
        }
        P.rlo = (uint32_t)__builtin_ctzll(mk);
        P.base = wbase;
        if (wbase + P.rlo >= dB.n_blocks) { P.mode = 3; return P; }
        const unsigned long long mh = __ballot(win.last_id >= hi_id ? 1 : 0);
        if (mh == 0) { P.mode = 2; return P; }
        P.rhi = (uint32_t)__builtin_ctzll(mh);
        if (wbase + P.rhi >= dB.n_blocks) P.rhi = dB.n_blocks - 1 - wbase;
        if (win_dirty) {
            wver ^= 1;
            if (t < 64) { sm.bw_last[wver][t] = win.last_id; sm.bw_first[wver][t] = win.first_id; sm.bw_woff[wver][t] = win.ids_woff; sm.bw_nb[wver][t] = win.n_ids_bits; }
            win_dirty = 0;
        }
        P.ver = wver;
        uint32_t has_breaks = dB.flags & LIST_HAS_BREAKS;
        KW_UNIFORM_OPAQUE(has_breaks);
        if (has_breaks) {
            const uint32_t w_endw = win.ids_woff + packed_words(win.n_ids_bits & 0xFFFF, win.n_ids_bits >> 16);
            const uint32_t nxt_woff = (uint32_t)__shfl(win.ids_woff, (int)((lane + 1) & 63));
            const bool brk = lane >= P.rlo && lane < P.rhi && w_endw != nxt_woff;
            if (__ballot(brk ? 1 : 0) != 0) { P.mode = 2; return P; }
        }
        // (rlo / rhi are uniform: v_readlane + scalar arithmetic, not two ds_bpermute round trips and a per-lane multiply)
        P.w_begin = (uint32_t)__builtin_amdgcn_readlane((int)win.ids_woff, (int)P.rlo);
        const uint32_t nb_hi = (uint32_t)__builtin_amdgcn_readlane((int)win.n_ids_bits, (int)P.rhi);
        P.W = (uint32_t)__builtin_amdgcn_readlane((int)win.ids_woff, (int)P.rhi) + packed_words(nb_hi & 0xFFFF, nb_hi >> 16) - P.w_begin;
        if (P.W <= (uint32_t)(PIPE_WORDS * KW_THREADS)) {
            P.mode = 0;
            tbuf ^= 1; P.buf = tbuf;
            const uint32_t* lane_src = idwB + P.w_begin + t;
            uint32_t* lds_wave_base = sm.btile + tbuf * HALF + wave * 64;
#if TSGPU_F2_MIDSLABS
            // (three fill sizes: the COUNT instantiation showed 10.5 of the find kernel's 17 GB of requests per 10 000-query batch to be tile DMA, two thirds of
            //  the pairs taking the 7-slab form for runs of 513..1024 words)
            if (P.W <= 2u * KW_THREADS) kw_glds_slabs<2>(lane_src, lds_wave_base);
            else if (PIPE_WORDS > 4 && P.W <= 4u * KW_THREADS) kw_glds_slabs<4>(lane_src, lds_wave_base);
            else kw_glds_slabs<PIPE_WORDS>(lane_src, lds_wave_base);
            if constexpr (COUNT) cb_tile += 4u * (P.W <= 2u * KW_THREADS ? 2u : ((PIPE_WORDS > 4 && P.W <= 4u * KW_THREADS) ? 4u : (uint32_t)PIPE_WORDS));
#else
            if (P.W <= 2u * KW_THREADS) kw_glds_slabs<2>(lane_src, lds_wave_base);
            else kw_glds_slabs<PIPE_WORDS>(lane_src, lds_wave_base);
            if constexpr (COUNT) cb_tile += 4u * (P.W <= 2u * KW_THREADS ? 2u : (uint32_t)PIPE_WORDS);
#endif
        } else P.mode = KW_F2_ROUNDS ? 1 : 2;
        return P;
    };

    // the pair (b, b + 1): metadata, raw ids and the plan one pair ahead. Of a pair's metadata only (first id, id count) of either block live on into the
    // iteration that searches it; the next pair's records are read out of the window registers where they are used (as four whole records carried across
    // the loop's back edge they were 16 of the kernel's 67 spilled SGPRs, each reload a v_readlane on the vector ALU: profiles/r06/exp_find2_valu.txt)
    uint32_t a_first, a_nb, b_first, b_nb;
    uint32_t araw0, araw1;
    Plan P;
    {
        const BlockIds mA = meta(wi.blk_begin), mB = meta(wi.blk_begin + 1);
        araw0 = load_id_raw(mA, t); araw1 = load_id_raw(mB, t);
        P = make_plan(mA.first_id, wi.blk_begin + 1 < wi.blk_end ? mB.last_id : mA.last_id);
        a_first = mA.first_id; a_nb = mA.n_ids_bits; b_first = mB.first_id; b_nb = mB.n_ids_bits;
    }
    const uint32_t has_deadline = q.deadline_rem_us;   // (read once: inside the loop the compiler re-reads the LDS copy of the query — and waits for it — every iteration)
    uint32_t q1n = 0, qfn = 0, par = 0, qh = 0;         // survivor queue: a RING of KW_QCAP entries, head qh, q1n queued; qfn = complete hits written (all uniform, in registers)
    KW_PROF_DECL
    static_assert((KW_QCAP & (KW_QCAP - 1)) == 0 && KW_QCAP >= 2 * KW_THREADS, "ring of at least 255 left-over + 256 new entries");

    // third.. lists for the first n_take queued survivors (one per thread, queue order = ascending id), complete hits to the work item's segment.
    // Two barriers per batch — the appends become visible / the ordered compaction — and no queue shuffle: the head just advances. (The shared
    // kw_probe_rest_stage keeps its counters in LDS and moves the left-over entries down: six barriers per batch.)
    auto probe_batch = [&](uint32_t n_take) {
        __syncthreads();                                 // (appends of every wave are visible; nobody is still reading the entries a previous batch freed)
        bool ok = t < n_take;
        uint32_t id = 0, v[TMAX];
#pragma unroll
        for (int k = 0; k < TMAX; k++) v[k] = 0;
        const IndexView ixp = KW_RELOAD_VIEW(ix);       // (the view's fields are read from the kernarg segment here, not kept in SGPRs across the pair loop)
        if (ok) {
            const uint32_t e = (qh + t) & (uint32_t)(KW_QCAP - 1);
            id = sm.q1_id[e];
            const uint32_t p0 = sm.q1_p0[e], p1 = sm.q1_p1[e];
#pragma unroll
            for (int k = 0; k < TMAX; k++) { if (k == q.probe_order[0]) v[k] = p0; if (k == q.probe_order[1]) v[k] = p1; }
            for (uint32_t s = 2; s < T && ok; s++) {
                const uint32_t tok = q.probe_order[s];
                uint32_t p;
                if constexpr (COUNT) ok = probe_list<true>(ixp, ixp.lists[q.list[tok]], id, p, &cb_probe);
                else ok = probe_list(ixp, ixp.lists[q.list[tok]], id, p);
#pragma unroll
                for (int k = 0; k < TMAX; k++) if ((uint32_t)k == tok) v[k] = p;
            }
        }
        uint32_t total;
        const uint32_t my = block_compact1(ok, sm.wave_cnt2[par], total);
        par ^= 1;
        if (ok) kw_hit_store<TMAX>(hits, qfn + my, id, v);
        if constexpr (COUNT) { if (ok) cb_rec += 4u * (TMAX + 1); if (lane == 0 && T > 2) cb_meta += (T - 2) * (uint32_t)sizeof(ListDesc); }
        qfn += total;
        qh = (qh + n_take) & (uint32_t)(KW_QCAP - 1);
        q1n -= n_take;
    };
    auto drain_q1 = [&]() { while (q1n >= (uint32_t)KW_THREADS) probe_batch(KW_THREADS); };   // (uniform) full batches only

    for (uint32_t b = wi.blk_begin, it = 0; b < wi.blk_end; b += 2, it++) {
        if (P.mode == 3) break;
        uint32_t Tl = T;                                 // (the loop's own copy of T: its T >= 2 / T >= 3 tests are compares, not hoisted lane masks)
        KW_UNIFORM_OPAQUE(Tl);
        KW_PROF(8)
        kw_glds_wait();                                  // this pair's tile (and driver ids) have landed
        KW_PROF(1)
        if (has_deadline) { if ((it & 7) == 0 && kw_out_of_time<true>(KW_RELOAD_VIEW(ix), q, wi.query, &sm.stop)) break; }
        const bool two = b + 1 < wi.blk_end;
        const uint32_t n0 = a_nb & 0xFFFF, n1 = two ? (b_nb & 0xFFFF) : 0u;
        bool ok0 = t < n0, ok1 = t < n1;
        const uint32_t id0 = ok0 ? a_first + araw0 : 0xFFFFFFFFu, id1 = ok1 ? b_first + araw1 : 0xFFFFFFFFu;
        const Plan C = P;
        KW_PROF(0)
        const uint32_t* __restrict__ tile = sm.btile + C.buf * HALF;
        KW_F2_LOOP_BARRIER();
        KW_PROF(2)
        // ---- request the next pair first (driver ids, plan, tile DMA: in flight during both searches of this pair). The plan may slide the
        //      window registers: this pair's block search reads the last ids it was planned on from a copy ----
        const uint32_t cur_last = win.last_id;
        uint32_t araw0n = 0, araw1n = 0;
        if (b + 2 < wi.blk_end) {
            const BlockIds mC = meta(b + 2), mD = meta(b + 3);
            araw0n = load_id_raw(mC, t);
            araw1n = load_id_raw(mD, t);
            KW_PROF(10)
            P = make_plan(mC.first_id, b + 3 < wi.blk_end ? mD.last_id : mC.last_id);
            a_first = mC.first_id; a_nb = mC.n_ids_bits; b_first = mD.first_id; b_nb = mD.n_ids_bits;      // (this pair's ids were formed above)
        }
        KW_PROF(4)
        // ---- (a) which block of the run, for both candidates ----
        const uint32_t span = C.rhi - C.rlo;
        uint32_t pos0 = C.rlo, pos1 = C.rlo;
        if (C.mode <= 1 && span <= (uint32_t)KW_F2_SPAN) {
            for (uint32_t j = C.rlo; j < C.rhi; j++) {
                const uint32_t last_j = (uint32_t)__builtin_amdgcn_readlane((int)cur_last, (int)j);
                pos0 += last_j < id0 ? 1u : 0u;
                pos1 += last_j < id1 ? 1u : 0u;
            }
        }
        bool done0 = !ok0 || C.mode >= 2, done1 = !ok1 || C.mode >= 2, found0 = false, found1 = false;
        uint32_t p10 = 0, p11 = 0;                      // posting positions in the second list
        uint32_t first0 = 0, nb0 = 0, rel0 = 0, first1 = 0, nb1 = 0, rel1 = 0;
        if (C.mode <= 1) {
            const uint32_t* __restrict__ bl = sm.bw_last[C.ver];
            if (span > (uint32_t)KW_F2_SPAN) {
                for (uint32_t step = 1u << (31 - __builtin_clz(span)); step > 0; step >>= 1) {
                    const uint32_t j0 = pos0 + step, j1 = pos1 + step;
                    const uint32_t v0 = bl[(j0 <= C.rhi ? j0 : C.rhi) - 1], v1 = bl[(j1 <= C.rhi ? j1 : C.rhi) - 1];
                    pos0 = (j0 <= C.rhi && v0 < id0) ? j0 : pos0;
                    pos1 = (j1 <= C.rhi && v1 < id1) ? j1 : pos1;
                }
            }
            // (inactive lanes carry id = 0xFFFFFFFF: pos = rhi, harmless reads)
            first0 = sm.bw_first[C.ver][pos0]; first1 = sm.bw_first[C.ver][pos1];
            const uint32_t l0 = bl[pos0], l1 = bl[pos1];
            nb0 = sm.bw_nb[C.ver][pos0]; nb1 = sm.bw_nb[C.ver][pos1];
            rel0 = sm.bw_woff[C.ver][pos0] - C.w_begin; rel1 = sm.bw_woff[C.ver][pos1] - C.w_begin;
            if (l0 < id0 || id0 < first0) done0 = true;                  // beyond B's end / in the gap between two blocks
            if (l1 < id1 || id1 < first1) done1 = true;
        }
        KW_PROF(3)
        // ---- (b) which slot: branch-free lower bound over the block's ids in the LDS tile ----
        auto slot_search = [&](const uint32_t* __restrict__ tile_r, uint32_t id, uint32_t b_first, uint32_t b_nb, uint32_t tile_rel, uint32_t kb, bool& found, uint32_t& p1) {
            const uint32_t n = b_nb & 0xFFFF, target = id - b_first;
            uint32_t pos = 0, hit;
            if ((b_nb >> 16) == 16) {
                const uint16_t* __restrict__ a16 = (const uint16_t*)(tile_r + tile_rel);
                if (n == (uint32_t)BLOCK_IDS) {
#pragma unroll
                    for (uint32_t step = 128; step > 0; step >>= 1) { const uint32_t v = a16[pos + step - 1]; pos = v < target ? pos + step : pos; }
                } else {
#pragma unroll
                    for (uint32_t step = 128; step > 0; step >>= 1) {
                        const uint32_t j = pos + step;
                        const uint32_t v = a16[(j <= n ? j : n) - 1];
                        pos = (j <= n && v < target) ? j : pos;
                    }
                }
                hit = a16[pos];
            } else {
                const uint32_t* __restrict__ a32 = tile_r + tile_rel;
#pragma unroll
                for (uint32_t step = 128; step > 0; step >>= 1) {
                    const uint32_t j = pos + step;
                    const uint32_t v = a32[(j <= n ? j : n) - 1];
                    pos = (j <= n && v < target) ? j : pos;
                }
                hit = a32[pos];
            }
            if (hit == target) { found = true; p1 = (C.base + kb) * BLOCK_IDS + pos; }
        };
        if (C.mode == 0) {
            // every block but a list's last is full, and 16-bit wherever the list is dense: when that holds for all of a wavefront's
            // candidates the two searches run as ONE straight-line sequence of two independent chains (candidates that dropped out
            // search the tile's first block: harmless reads)
            constexpr uint32_t FULL16 = (16u << 16) | (uint32_t)BLOCK_IDS;
            const bool fast = (done0 || nb0 == FULL16) && (done1 || nb1 == FULL16);
            if (KW_F2_FAST && __ballot(fast ? 0 : 1) == 0) {
                const uint16_t* __restrict__ a0 = (const uint16_t*)(tile + (done0 ? 0u : rel0));
                const uint16_t* __restrict__ a1 = (const uint16_t*)(tile + (done1 ? 0u : rel1));
                const uint32_t t0 = id0 - first0, t1 = id1 - first1;
                uint32_t s0 = 0, s1 = 0;
                if constexpr (KW_F2_QUAD) {
                    // 4-ary lower bound: three independent samples per round, four rounds instead of eight dependent LDS reads; the last
                    // round's four ids are two 32-bit words (s is a multiple of 4: 8-byte offset from the word-aligned block start)
#pragma unroll
                    for (uint32_t step = 64; step >= 4; step >>= 2) {
                        const uint32_t x0 = a0[s0 + step - 1], y0 = a0[s0 + 2 * step - 1], z0 = a0[s0 + 3 * step - 1];
                        const uint32_t x1 = a1[s1 + step - 1], y1 = a1[s1 + 2 * step - 1], z1 = a1[s1 + 3 * step - 1];
                        s0 += step * ((x0 < t0 ? 1u : 0u) + (y0 < t0 ? 1u : 0u) + (z0 < t0 ? 1u : 0u));
                        s1 += step * ((x1 < t1 ? 1u : 0u) + (y1 < t1 ? 1u : 0u) + (z1 < t1 ? 1u : 0u));
                    }
                    const uint32_t* __restrict__ w0 = (const uint32_t*)(a0 + s0);
                    const uint32_t* __restrict__ w1 = (const uint32_t*)(a1 + s1);
                    const uint32_t wa0 = w0[0], wb0 = w0[1], wa1 = w1[0], wb1 = w1[1];
                    const uint32_t e00 = wa0 & 0xFFFF, e01 = wa0 >> 16, e02 = wb0 & 0xFFFF, e03 = wb0 >> 16;
                    const uint32_t e10 = wa1 & 0xFFFF, e11 = wa1 >> 16, e12 = wb1 & 0xFFFF, e13 = wb1 >> 16;
                    found0 = !done0 && (e00 == t0 || e01 == t0 || e02 == t0 || e03 == t0);
                    found1 = !done1 && (e10 == t1 || e11 == t1 || e12 == t1 || e13 == t1);
                    s0 += (e00 < t0 ? 1u : 0u) + (e01 < t0 ? 1u : 0u) + (e02 < t0 ? 1u : 0u);
                    s1 += (e10 < t1 ? 1u : 0u) + (e11 < t1 ? 1u : 0u) + (e12 < t1 ? 1u : 0u);
                } else {
                    // (the cursor IS the LDS address: per step one add, one compare, one select — the index form costs a fourth VALU instruction per step and
                    //  chain for the address, and the vector ALU is the kernel's busiest port: profiles/r06/exp_find2_valu.txt)
                    const uint16_t* __restrict__ c0 = a0;
                    const uint16_t* __restrict__ c1 = a1;
#pragma unroll
                    for (uint32_t step = 128; step > 0; step >>= 1) {
                        const uint32_t v0 = c0[step - 1], v1 = c1[step - 1];
                        c0 = v0 < t0 ? c0 + step : c0;
                        c1 = v1 < t1 ? c1 + step : c1;
                    }
                    const uint32_t h0 = c0[0], h1 = c1[0];
                    s0 = (uint32_t)(c0 - a0); s1 = (uint32_t)(c1 - a1);
                    found0 = !done0 && h0 == t0; found1 = !done1 && h1 == t1;
                }
                p10 = (C.base + pos0) * BLOCK_IDS + s0; p11 = (C.base + pos1) * BLOCK_IDS + s1;
            } else {
                if (!done0) slot_search(tile, id0, first0, nb0, rel0, pos0, found0, p10);
                if (!done1) slot_search(tile, id1, first1, nb1, rel1, pos1, found1, p11);
            }
        } else if (KW_F2_ROUNDS && C.mode == 1) {
            // the run does not fit the pipelined tile: rounds over [rlo, rhi], each a coalesced copy of as many whole blocks as fit
            const uint32_t* __restrict__ woff = sm.bw_woff[C.ver];
            const uint32_t* __restrict__ wnb = sm.bw_nb[C.ver];
            uint32_t* __restrict__ rt = sm.btile + (tbuf ^ 1) * HALF;        // the buffer no DMA is writing
            for (uint32_t r_lo = C.rlo; r_lo <= C.rhi;) {
                const uint32_t w_begin = woff[r_lo];
                uint32_t r_hi = r_lo;
                while (r_hi < C.rhi && woff[r_hi + 1] + packed_words(wnb[r_hi + 1] & 0xFFFF, wnb[r_hi + 1] >> 16) - w_begin <= (uint32_t)HALF) r_hi++;
                const uint32_t W = woff[r_hi] + packed_words(wnb[r_hi] & 0xFFFF, wnb[r_hi] >> 16) - w_begin;
                const uint32_t* __restrict__ src = idwB + w_begin;
                __syncthreads();
                for (uint32_t i0 = t; i0 < W + t; i0 += 2 * KW_THREADS) {
                    const uint32_t i1 = i0 + KW_THREADS;
                    const uint32_t c0 = src[i0 < W ? i0 : 0], c1 = src[i1 < W ? i1 : 0];
                    if (i0 < W) rt[i0] = c0;
                    if (i1 < W) rt[i1] = c1;
                }
                __syncthreads();
                if (!done0 && pos0 >= r_lo && pos0 <= r_hi) { slot_search(rt, id0, first0, nb0, woff[pos0] - w_begin, pos0, found0, p10); done0 = true; }
                if (!done1 && pos1 >= r_lo && pos1 <= r_hi) { slot_search(rt, id1, first1, nb1, woff[pos1] - w_begin, pos1, found1, p11); done1 = true; }
                r_lo = r_hi + 1;
            }
        } else if (C.mode == 2) {
            const IndexView ixp = KW_RELOAD_VIEW(ix);   // (wide / broken runs: the view and the second list's descriptor re-read here)
            const ListDesc dBp = ixp.lists[q.list[q.probe_order[1]]];
            if constexpr (COUNT) {
                if (ok0) found0 = probe_list<true>(ixp, dBp, id0, p10, &cb_probe);
                if (ok1) found1 = probe_list<true>(ixp, dBp, id1, p11, &cb_probe);
            } else {
                ProbeReq r0, r1;                             // (both candidates' directory entries requested before either is looked at)
                r0.e = make_uint2(0u, 0u); r0.state = 0; r1.e = make_uint2(0u, 0u); r1.state = 0;
                if (ok0) probe_issue(ixp, dBp, id0, r0);
                if (ok1) probe_issue(ixp, dBp, id1, r1);
                if (ok0) found0 = probe_finish(ixp, dBp, id0, r0, p10);
                if (ok1) found1 = probe_finish(ixp, dBp, id1, r1, p11);
            }
        }
        if (Tl >= 2) { ok0 = ok0 && found0; ok1 = ok1 && found1; }
        KW_PROF(5)
        // ---- ordered compaction of both halves behind ONE barrier ----
        const unsigned long long m0 = __ballot(ok0 ? 1 : 0), m1 = __ballot(ok1 ? 1 : 0);
        const uint32_t lo0 = (uint32_t)__popcll(m0 & ((1ull << lane) - 1ull)), lo1 = (uint32_t)__popcll(m1 & ((1ull << lane) - 1ull));
        if (lane == 0) { sm.wave_cnt2[par][wave] = (uint32_t)__popcll(m0); wave_cnt_b[par][wave] = (uint32_t)__popcll(m1); }
        __syncthreads();
        uint32_t base0 = 0, tot0 = 0, base1 = 0, tot1 = 0;
#pragma unroll
        for (int w = 0; w < KW_THREADS / 64; w++) {
            const uint32_t c0 = sm.wave_cnt2[par][w], c1 = wave_cnt_b[par][w];
            if ((uint32_t)w < wave) { base0 += c0; base1 += c1; }
            tot0 += c0; tot1 += c1;
        }
        par ^= 1;
        KW_PROF(6)
        const uint32_t pa0 = b * BLOCK_IDS + t, pa1 = (b + 1) * BLOCK_IDS + t;        // posting positions in the driver list
#ifdef TSGPU_F2_NOSTAGE2                                     // tools/ ablation only (results are WRONG): the pair loop without the third.. lists
        if (Tl >= 3) { q1n = 0; } else
#endif
        if (Tl >= 3) {
            if (ok0) { const uint32_t slot = (qh + q1n + base0 + lo0) & (uint32_t)(KW_QCAP - 1); sm.q1_id[slot] = id0; sm.q1_p0[slot] = pa0; sm.q1_p1[slot] = p10; }
            q1n += tot0;
            KW_PROF(7)
            drain_q1();                                  // (the queue holds 512 entries: 255 left over + one block's survivors)
            KW_PROF(11)
            if (ok1) { const uint32_t slot = (qh + q1n + base1 + lo1) & (uint32_t)(KW_QCAP - 1); sm.q1_id[slot] = id1; sm.q1_p0[slot] = pa1; sm.q1_p1[slot] = p11; }
            q1n += tot1;
            KW_PROF(7)
            drain_q1();
            KW_PROF(11)
        } else {
            // one or two lists: the stage-1 survivors ARE the complete hits (ascending: block b's, then block b + 1's)
            if (ok0) {
                uint32_t v[TMAX];
#pragma unroll
                for (int k = 0; k < TMAX; k++) { v[k] = 0; if (k == q.probe_order[0]) v[k] = pa0; if (Tl >= 2 && k == q.probe_order[1]) v[k] = p10; }
                kw_hit_store<TMAX>(hits, qfn + base0 + lo0, id0, v);
                if constexpr (COUNT) cb_rec += 4u * (TMAX + 1);
            }
            if (ok1) {
                uint32_t v[TMAX];
#pragma unroll
                for (int k = 0; k < TMAX; k++) { v[k] = 0; if (k == q.probe_order[0]) v[k] = pa1; if (Tl >= 2 && k == q.probe_order[1]) v[k] = p11; }
                kw_hit_store<TMAX>(hits, qfn + tot0 + base1 + lo1, id1, v);
                if constexpr (COUNT) cb_rec += 4u * (TMAX + 1);
            }
            qfn += tot0 + tot1;
        }
        KW_PROF(7)
        if (b + 5 >= abase + 64 && b + 4 < wi.blk_end) { abase = b + 4; awin = load_awin(abase); }   // (every 30 pairs; waited for by the next iteration's kw_glds_wait, read by its meta(b + 2), meta(b + 3))
        araw0 = araw0n; araw1 = araw1n;
    }
    if (T >= 3) while (q1n > 0) probe_batch(q1n < (uint32_t)KW_THREADS ? q1n : (uint32_t)KW_THREADS);
    if (t == 0) part.cnt[bid] = qfn;                           // hits handed to kw_score_kernel
    if constexpr (COUNT) {
        uint32_t c[5] = {cb_ids, cb_meta, cb_tile, cb_probe, cb_rec};
#pragma unroll
        for (int k = 0; k < 5; k++) {
            for (int m = 32; m > 0; m >>= 1) c[k] += __shfl_xor(c[k], m);
            if (lane == 0 && ix.touched) atomicAdd(ix.touched + k, (unsigned long long)c[k]);
        }
        if (t == 0 && ix.touched) { atomicAdd(ix.touched + 5, 1ull); atomicAdd(ix.touched + 6, (unsigned long long)qfn); }      // work items, hit records
    }
    KW_PROF(9)
    KW_PROF_FLUSH(ix.prof)
}

template <int TMAX, bool COUNT = false>
__global__ __launch_bounds__(KW_THREADS) KW_F2_WAVES void kw_find2_kernel(IndexView ix, const KwQueryDev* __restrict__ queries,
                                                                                      const KwWorkItem* __restrict__ work, KwPartials part,
                                                                                      uint32_t* __restrict__ hits_all, const uint64_t* __restrict__ hit_off) {
    kw_find2_body<TMAX, COUNT>(ix, queries, work, part, hits_all, hit_off, blockIdx.x);
}
