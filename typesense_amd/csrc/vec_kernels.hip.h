// vec_kernels.hip.h — gfx950 kernels of the vector hot path (seam B2, include/tsgpu.h).
//
// Replaces, for a whole batch of queries, the reference's per-query distance loop (process_results_bruteforce,
// src/index.cpp:3345-3374: dist = space->get_dist_func()(q, x, &dim) = 1 - <q,x>, hnswlib InnerProductSpace) and
// the top-k it feeds (searchKnnCloserFirst result + Topster, src/index.cpp:3384-3389, 3682-3725) with an EXACT
// scan: S = X · Qᵀ on the fp32-input matrix cores; the N x B score matrix never exists in memory.
//
// Three kernels, used twice each per batch:
//   vec_scan_kernel    the GEMM. Workgroup = 4 waves = 128 base rows x QT queries (QT = 128, or 64 for small
//                      batches); each wave owns a 64 x QT/2 sub-tile = 2 x CB accumulators of
//                      v_mfma_f32_32x32x2_f32 (exact fp32). The K dimension streams through a double-buffered,
//                      conflict-free LDS ring in 32-float chunks (one barrier per chunk, next chunk's global
//                      loads in flight under the MFMAs); operands are fetched with ds_read_b128, 4 k-pairs per
//                      read. Epilogue = a FILTER, not a heap: a score is kept iff its key (ord(dist)<<32 | row)
//                      is <= the query's threshold key tau; survivors are appended to the query's candidate list
//                      in HBM with one atomic. In steady state nothing survives (one max + compare per
//                      accumulator block), so the epilogue costs ~1% of the MFMA time.
//   vec_select_kernel  one workgroup per query: exact k-th smallest key of a candidate list by MSB-first radix
//                      select (8-bit digits, LDS histograms), then a bitonic sort of the <= k winners.
//   Pass 1 ("sample"): vec_scan over every (n_tiles/512)-th row tile in DENSE mode (all scores written), then
//   vec_select gives tau[q] = k-th best of the sample — a valid upper bound of the final k-th best because the
//   sample is a subset of the rows. Pass 2: vec_scan over ALL rows filtered by tau (expected survivors per query
//   = k * N / sample_rows), then vec_select produces the final k. A candidate list that overflows tightens its
//   own tau from what it holds and the pass is repeated (exactness never depends on the data distribution).
//   * blockIdx -> (slab, query tile) is XCD-aware: the query tiles of one slab run on the same XCD back to
//     back, so a slab is fetched from HBM once and re-served from that XCD's L2.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tsgpu {

static const int VEC_THREADS = 256;
static const int VEC_ROWS = 128;        // base rows per tile (2 wave rows x 64)
static const int VEC_KC = 32;           // K chunk staged in LDS per step
#ifndef VEC_STORE_AT_G
#define VEC_STORE_AT_G 3   // k-group whose MFMAs cover the LDS stores of the next chunk (4 = after the MFMAs)
#endif
static const int VEC_LDW = VEC_KC + 4;  // padded row stride (words): 36*r mod 64 hits 16 distinct 4-bank groups -> ds_read_b128 conflict-free
static const uint64_t VEC_KEY_INF = 0xFFFFFFFFFFFFFFFFull;
static const int VEC_SELECT_MAXK = 1024;   // TSGPU_MAX_TOPK

typedef float vec_f32x16 __attribute__((ext_vector_type(16)));

// order-preserving float -> uint32 (smaller distance = smaller key)
__device__ inline uint32_t f32_ord(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ inline float ord_f32(uint32_t o) {
    const uint32_t b = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(b);
}

struct VecScanArgs {
    const float* X;            // [n_rows][dim] row-major
    const uint8_t* row_ok;     // nullable: 0 = skip row (deleted / filtered out)
    const float* Q;            // [n_q][dim]
    uint32_t n_rows, dim, n_q;
    uint32_t n_ord;            // tile ordinals covered by this launch; tile id = ordinal * tile_stride
    uint32_t tile_stride;
    uint32_t ord_per_slab;     // ordinals per workgroup slab
    uint32_t n_slabs;          // multiple of 8 (XCD-aware mapping)
    uint32_t n_qtiles;
    const uint64_t* tau;       // [n_q] threshold keys (sparse mode); null = keep everything
    uint64_t* cand;            // sparse mode: [n_q][cand_cap]
    uint32_t* cand_cnt;        // [n_q] (keeps counting past cand_cap: overflow is detected by the select kernel)
    uint32_t cand_cap;
    uint64_t* dense;           // dense mode (non-null): dense[q * dense_stride + ordinal*128 + r] = key or INF
    uint32_t dense_stride;
};

template <int QT>
struct VecScanSmem {
    alignas(16) float xs[2][VEC_ROWS * VEC_LDW];
    alignas(16) float qs[2][QT * VEC_LDW];
};

// CB = 32-query column blocks per wave (QT = 64 * CB); ALIGNED = dim % 4 == 0 and 16-byte aligned bases
template <int CB, bool ALIGNED>
__global__ __launch_bounds__(VEC_THREADS, 2) void vec_scan_kernel(VecScanArgs a) {
    constexpr int QT = 64 * CB;
    constexpr int XV = VEC_ROWS * VEC_KC / 4 / VEC_THREADS;     // float4 per thread per X chunk (4)
    constexpr int QV = QT * VEC_KC / 4 / VEC_THREADS;           // 4 (QT=128) or 2 (QT=64)
    __shared__ VecScanSmem<QT> sm;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t wrow = (wave >> 1) * 64, wcol = (wave & 1) * (32 * CB);
    // XCD-aware: blocks b, b+8, b+16.. share an XCD; give them the query tiles of the same slab
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7, j = b >> 3;
    const uint32_t qtile = j % a.n_qtiles;
    const uint32_t slab = (j / a.n_qtiles) * 8 + xcd;
    const uint32_t q0 = qtile * QT;
    const uint32_t ord_begin = slab * a.ord_per_slab;
    uint32_t ord_end = ord_begin + a.ord_per_slab;
    if (ord_end > a.n_ord) ord_end = a.n_ord;
    if (ord_begin >= ord_end) return;
    const uint32_t n_chunks = (a.dim + VEC_KC - 1) / VEC_KC;
    const uint32_t total_steps = (ord_end - ord_begin) * n_chunks;

    float4 xr[XV], qr[QV];
    // Guarded load without touching the loaded registers (a select on them would force the wave to wait for the
    // global load BEFORE the chunk's MFMAs instead of after them): out-of-range rows / queries read a clamped
    // in-range row — their scores are computed and then dropped by the epilogue (row < n_rows, gq < n_q), and every
    // accumulator element depends on one row and one query only. Only the K tail (dim % 32 != 0, last chunk) needs
    // zeros; they are applied in store_step, i.e. after the MFMAs of the previous chunk.
    auto load4 = [&](const float* base, uint32_t row, uint32_t row_lim, uint32_t gk) -> float4 {
        const uint32_t r = row < row_lim ? row : row_lim - 1;
        float4 val;
        if (ALIGNED) {
            const uint32_t kk = gk < a.dim ? gk : a.dim - 4;
            val = *(const float4*)(base + (size_t)r * a.dim + kk);
        } else {
            const float* p = base + (size_t)r * a.dim;
            const uint32_t d1 = a.dim - 1;
            val.x = p[gk < d1 ? gk : d1];
            val.y = p[gk + 1 < d1 ? gk + 1 : d1];
            val.z = p[gk + 2 < d1 ? gk + 2 : d1];
            val.w = p[gk + 3 < d1 ? gk + 3 : d1];
        }
        return val;
    };
    // global -> registers for pipeline step s (tile ordinal = ord_begin + s / n_chunks, chunk = s % n_chunks)
    auto load_step = [&](uint32_t s) {
        const uint32_t o = ord_begin + s / n_chunks, c = s % n_chunks;
        const uint32_t r0 = o * a.tile_stride * VEC_ROWS, k0 = c * VEC_KC;
#pragma unroll
        for (int v = 0; v < XV; v++) {
            const uint32_t idx = t + v * VEC_THREADS;
            const uint32_t row = r0 + idx / (VEC_KC / 4), gk = k0 + (idx % (VEC_KC / 4)) * 4;
            xr[v] = load4(a.X, row, a.n_rows, gk);
        }
#pragma unroll
        for (int v = 0; v < QV; v++) {
            const uint32_t idx = t + v * VEC_THREADS;
            const uint32_t gq = q0 + idx / (VEC_KC / 4), gk = k0 + (idx % (VEC_KC / 4)) * 4;
            qr[v] = load4(a.Q, gq, a.n_q, gk);
        }
    };
    // registers -> LDS for pipeline step s (the K tail of the last chunk is zeroed here, on BOTH operands)
    auto store_step = [&](uint32_t buf, uint32_t s) {
        const uint32_t k0 = (s % n_chunks) * VEC_KC;
        if (k0 + VEC_KC > a.dim) {                              // uniform: only the last chunk of a dim % 32 != 0 index
            const uint32_t gk = k0 + (t % (VEC_KC / 4)) * 4;    // idx % 8 == t % 8 for every v
#pragma unroll
            for (int v = 0; v < XV; v++) {
                if (gk >= a.dim) xr[v].x = 0.f;
                if (gk + 1 >= a.dim) xr[v].y = 0.f;
                if (gk + 2 >= a.dim) xr[v].z = 0.f;
                if (gk + 3 >= a.dim) xr[v].w = 0.f;
            }
#pragma unroll
            for (int v = 0; v < QV; v++) {
                if (gk >= a.dim) qr[v].x = 0.f;
                if (gk + 1 >= a.dim) qr[v].y = 0.f;
                if (gk + 2 >= a.dim) qr[v].z = 0.f;
                if (gk + 3 >= a.dim) qr[v].w = 0.f;
            }
        }
#pragma unroll
        for (int v = 0; v < XV; v++) {
            const uint32_t idx = t + v * VEC_THREADS;
            *(float4*)&sm.xs[buf][(idx / (VEC_KC / 4)) * VEC_LDW + (idx % (VEC_KC / 4)) * 4] = xr[v];
        }
#pragma unroll
        for (int v = 0; v < QV; v++) {
            const uint32_t idx = t + v * VEC_THREADS;
            *(float4*)&sm.qs[buf][(idx / (VEC_KC / 4)) * VEC_LDW + (idx % (VEC_KC / 4)) * 4] = qr[v];
        }
    };

    // per-lane thresholds: this lane's query column in each of its CB blocks
    uint64_t tau_key[CB];
    float tau_dist[CB];
#pragma unroll
    for (int cb = 0; cb < CB; cb++) {
        const uint32_t gq = q0 + wcol + cb * 32 + (lane & 31);
        tau_key[cb] = (a.tau && gq < a.n_q) ? a.tau[gq] : VEC_KEY_INF;
        tau_dist[cb] = ord_f32((uint32_t)(tau_key[cb] >> 32));
    }

    vec_f32x16 acc[2][CB];
    load_step(0);
    store_step(0, 0);
    __syncthreads();
    for (uint32_t s = 0; s < total_steps; s++) {
        const uint32_t c = s % n_chunks, buf = s & 1;
        if (c == 0) {
#pragma unroll
            for (int rb = 0; rb < 2; rb++)
#pragma unroll
                for (int cb = 0; cb < CB; cb++)
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[rb][cb][e] = 0;
        }
        // UNCONDITIONAL (the last step re-requests its own block): a load under a runtime branch makes hipcc merge old/new
        // registers with copies right behind the loads, i.e. wait for them BEFORE the MFMAs instead of after
        load_step(s + 1 < total_steps ? s + 1 : s);          // in flight while the MFMAs run
        // lane (r = lane&31, h = lane>>5) supplies k = 8*g + 4*h + i to MFMA i of group g: every k exactly once
        const float* xa = &sm.xs[buf][(wrow + (lane & 31)) * VEC_LDW + 4 * (lane >> 5)];
        const float* qb = &sm.qs[buf][(wcol + (lane & 31)) * VEC_LDW + 4 * (lane >> 5)];
        // operands of k-group g+1 are requested from LDS before the 16 MFMAs of group g are issued (register double
        // buffer), and the 4 accumulators are interleaved so that consecutive MFMAs never depend on each other
        float4 av[2][2], bv[2][CB];
#pragma unroll
        for (int rb = 0; rb < 2; rb++) av[0][rb] = *(const float4*)(xa + rb * 32 * VEC_LDW);
#pragma unroll
        for (int cb = 0; cb < CB; cb++) bv[0][cb] = *(const float4*)(qb + cb * 32 * VEC_LDW);
#pragma unroll
        for (int g = 0; g < VEC_KC / 8; g++) {
            const int cur = g & 1, nx = cur ^ 1;
            if (g + 1 < VEC_KC / 8) {
#pragma unroll
                for (int rb = 0; rb < 2; rb++) av[nx][rb] = *(const float4*)(xa + rb * 32 * VEC_LDW + 8 * (g + 1));
#pragma unroll
                for (int cb = 0; cb < CB; cb++) bv[nx][cb] = *(const float4*)(qb + cb * 32 * VEC_LDW + 8 * (g + 1));
            }
            // the next chunk's registers -> LDS (other ring slot) go out under the last MFMA group instead of after it
            if (g == VEC_STORE_AT_G && s + 1 < total_steps) store_step(buf ^ 1, s + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rb = 0; rb < 2; rb++)
#pragma unroll
                for (int cb = 0; cb < CB; cb++) acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][rb].x, bv[cur][cb].x, acc[rb][cb], 0, 0, 0);
#pragma unroll
            for (int rb = 0; rb < 2; rb++)
#pragma unroll
                for (int cb = 0; cb < CB; cb++) acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][rb].y, bv[cur][cb].y, acc[rb][cb], 0, 0, 0);
#pragma unroll
            for (int rb = 0; rb < 2; rb++)
#pragma unroll
                for (int cb = 0; cb < CB; cb++) acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][rb].z, bv[cur][cb].z, acc[rb][cb], 0, 0, 0);
#pragma unroll
            for (int rb = 0; rb < 2; rb++)
#pragma unroll
                for (int cb = 0; cb < CB; cb++) acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][rb].w, bv[cur][cb].w, acc[rb][cb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (VEC_STORE_AT_G >= VEC_KC / 8 && s + 1 < total_steps) store_step(buf ^ 1, s + 1);
        if (c == n_chunks - 1) {
            // ---- tile epilogue: distance = 1 - dot; key = (ord(distance) << 32) | row ----
            const uint32_t o = ord_begin + s / n_chunks;
            const uint32_t r0 = o * a.tile_stride * VEC_ROWS;
#pragma unroll
            for (int cb = 0; cb < CB; cb++) {
                const uint32_t gq = q0 + wcol + cb * 32 + (lane & 31);
                if (a.dense) {
                    if (gq < a.n_q) {
#pragma unroll
                        for (int rb = 0; rb < 2; rb++)
#pragma unroll
                            for (int e = 0; e < 16; e++) {
                                const uint32_t lr = wrow + rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                                const uint32_t row = r0 + lr;
                                bool okr = row < a.n_rows;
                                if (okr && a.row_ok) okr = a.row_ok[row] != 0;
                                const float dist = 1.0f - acc[rb][cb][e];
                                a.dense[(size_t)gq * a.dense_stride + (size_t)(o * VEC_ROWS + lr)] =
                                    okr ? (((uint64_t)f32_ord(dist) << 32) | row) : VEC_KEY_INF;
                            }
                    }
                } else {
                    // cheap reject: the best (largest) dot of this lane's 32 scores for the column
                    float amax = acc[0][cb][0];
#pragma unroll
                    for (int rb = 0; rb < 2; rb++)
#pragma unroll
                        for (int e = 0; e < 16; e++) amax = fmaxf(amax, acc[rb][cb][e]);
                    if (gq < a.n_q && !((1.0f - amax) > tau_dist[cb])) {     // NaN-safe: NaN never rejects here
#pragma unroll
                        for (int rb = 0; rb < 2; rb++)
#pragma unroll
                            for (int e = 0; e < 16; e++) {
                                const uint32_t row = r0 + wrow + rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                                const float dist = 1.0f - acc[rb][cb][e];
                                const uint64_t key = ((uint64_t)f32_ord(dist) << 32) | row;
                                if (key <= tau_key[cb] && row < a.n_rows && (!a.row_ok || a.row_ok[row] != 0)) {
                                    const uint32_t slot = atomicAdd(&a.cand_cnt[gq], 1u);
                                    if (slot < a.cand_cap) a.cand[(size_t)gq * a.cand_cap + slot] = key;
                                }
                            }
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// One workgroup per query: exact selection of the k smallest keys of a candidate list.
//   keys = base + q * stride, n = cnt ? min(cnt[q], cap) : cap (dense lists are fully populated; INF = padding).
//   mode 0 (final): writes the k smallest keys ascending as (distance, label) + n_out.
//   mode 1 (threshold): writes tau[q] = k-th smallest key (INF when fewer than k real keys).
//   Overflow (cnt[q] > cap): tau[q] is tightened to the k-th smallest of the cap keys held (still a valid upper
//   bound of the true k-th best) and *overflow is raised, so the host re-runs the scan for the batch.
__global__ __launch_bounds__(VEC_THREADS) void vec_select_kernel(const uint64_t* __restrict__ base, size_t stride, const uint32_t* __restrict__ cnt,
                                                                  uint32_t cap, uint32_t k, int mode, const uint64_t* __restrict__ labels,
                                                                  float* __restrict__ dist_out, uint64_t* __restrict__ label_out,
                                                                  uint32_t* __restrict__ n_out, uint64_t* __restrict__ tau, uint32_t* __restrict__ overflow) {
    __shared__ uint32_t hist[256];
    __shared__ uint64_t win[VEC_SELECT_MAXK];
    __shared__ uint64_t s_prefix, s_diff;
    __shared__ uint32_t s_krem, s_win, s_valid;
    const uint32_t t = threadIdx.x, q = blockIdx.x;
    const uint64_t* __restrict__ keys = base + (size_t)q * stride;
    const uint32_t total = cnt ? cnt[q] : cap;
    const bool over = total > cap;
    const uint32_t n = over ? cap : total;
    if (t == 0) { s_prefix = 0; s_diff = 0; s_krem = k; s_win = 0; s_valid = 0; }
    __syncthreads();
    // number of real keys + the bits in which they differ (skips radix passes over a common prefix)
    {
        const uint64_t first = n ? keys[0] : 0;
        uint64_t d = 0;
        uint32_t valid = 0;
        for (uint32_t i = t; i < n; i += VEC_THREADS) { const uint64_t kv = keys[i]; d |= kv ^ first; valid += kv != VEC_KEY_INF; }
        if (d) atomicOr((unsigned long long*)&s_diff, (unsigned long long)d);
        if (valid) atomicAdd(&s_valid, valid);
    }
    __syncthreads();
    const uint32_t n_valid = s_valid;
    uint64_t kstar = VEC_KEY_INF - 1;                       // "take every real key"
    if (n_valid > k) {
        const uint64_t diff = s_diff;
        int top = 7;
        while (top > 0 && ((diff >> (8 * top)) & 0xFF) == 0) top--;
        uint64_t prefix = 0, pmask = 0;
        if (top < 7) { pmask = ~0ull << (8 * (top + 1)); prefix = keys[0] & pmask; }
        for (int p = top; p >= 0; p--) {
            hist[t] = 0;
            __syncthreads();
            const int shift = 8 * p;
            for (uint32_t i = t; i < n; i += VEC_THREADS) {
                const uint64_t kv = keys[i];
                if ((kv & pmask) == prefix) atomicAdd(&hist[(uint32_t)(kv >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (t == 0) {
                uint32_t krem = s_krem, cum = 0, d = 0;
                for (; d < 256; d++) { if (cum + hist[d] >= krem) break; cum += hist[d]; }
                s_krem = krem - cum;
                s_prefix = prefix | ((uint64_t)d << shift);
            }
            __syncthreads();
            prefix = s_prefix;
            pmask |= 0xFFull << shift;
            __syncthreads();
        }
        kstar = prefix;                                     // exact k-th smallest key (keys are unique per row)
    }
    if (mode == 1 || over) {
        if (t == 0) {
            if (mode == 1) tau[q] = n_valid >= k ? kstar : VEC_KEY_INF;
            else { tau[q] = kstar; atomicAdd(overflow, 1u); }
        }
        if (mode == 1) return;
    }
    // winners -> LDS, bitonic sort ascending, emit
    for (uint32_t i = t; i < (uint32_t)VEC_SELECT_MAXK; i += VEC_THREADS) win[i] = VEC_KEY_INF;
    __syncthreads();
    for (uint32_t i = t; i < n; i += VEC_THREADS) {
        const uint64_t kv = keys[i];
        if (kv <= kstar && kv != VEC_KEY_INF) { const uint32_t slot = atomicAdd(&s_win, 1u); if (slot < (uint32_t)VEC_SELECT_MAXK) win[slot] = kv; }
    }
    __syncthreads();
    uint32_t m = s_win < k ? s_win : k;
    uint32_t sz = 2;
    while (sz < s_win && sz < (uint32_t)VEC_SELECT_MAXK) sz <<= 1;
    for (uint32_t size = 2; size <= sz; size <<= 1) {
        for (uint32_t strd = size >> 1; strd > 0; strd >>= 1) {
            __syncthreads();
            for (uint32_t p = t; p < sz / 2; p += VEC_THREADS) {
                const uint32_t i = 2 * p - (p & (strd - 1));
                const uint32_t jx = i + strd;
                const bool asc = ((i & size) == 0);
                const uint64_t x = win[i], y = win[jx];
                if (asc ? (x > y) : (x < y)) { win[i] = y; win[jx] = x; }
            }
        }
    }
    __syncthreads();
    for (uint32_t i = t; i < m; i += VEC_THREADS) {
        const uint64_t kv = win[i];
        const uint32_t row = (uint32_t)(kv & 0xFFFFFFFFull);
        dist_out[(size_t)q * k + i] = ord_f32((uint32_t)(kv >> 32));
        label_out[(size_t)q * k + i] = labels[row];
    }
    if (t == 0) n_out[q] = m;
}

// ------------------------------------------------------------------------------------------------
// L2 normalisation exactly as hnsw_index_t::normalize_vector (include/index.h:379-388): sequential fp32 sum of
// squares with the multiply and the add rounded separately (the reference's generic x86-64 build has no FMA),
// then x * 1/(sqrt(sum)+1e-30). sqrtf and '/' are correctly rounded under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt; contraction is switched off for this function only.
__global__ void vec_normalize_rows_kernel(float* __restrict__ X, uint32_t n_rows, uint32_t dim) {
#pragma clang fp contract(off)
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    float* x = X + (size_t)r * dim;
    float norm = 0.0f;
    for (uint32_t i = 0; i < dim; i++) { const float sq = x[i] * x[i]; norm = norm + sq; }
    norm = 1.0f / (sqrtf(norm) + 1e-30f);
    for (uint32_t i = 0; i < dim; i++) x[i] = x[i] * norm;
}

// ================================================================================================
// bf16 PREFILTER path (default for the batched k-NN). Same exact results as the fp32 scan above, ~16x less matrix
// time: the full N x B sweep runs on v_mfma_f32_32x32x16_bf16 over a bf16 copy of X (half the HBM bytes of the fp32
// rows), but only to BRACKET every score — never to rank it:
//     |s~ - s^| <= e(q,r) = c * ||q|| * ||x_r||        s~ = bf16 MFMA score, s^ = the fp32 score the reference computes
// with c = 2^-7 + 2^-14 (round-to-nearest bf16 of both operands: (2u + u^2), u = 2^-8, times Cauchy-Schwarz) + dim*2^-21
// (fp32 accumulation of either sum), inflated by 1 %. Let L = the k-th largest LOWER bound s~ - e over any subset of
// the rows: at least k rows score >= L, so a row with UPPER bound s~ + e < L is not among the k nearest. Rows that
// survive that test twice (L1 from a strided sample, then L2 from the survivors themselves; ~1.1-3 k rows per query
// remain) are re-scored EXACTLY in fp32 with the reference's own summation order (hnswlib InnerProductSpace: 16
// accumulator lanes, multiply and add rounded separately, sequential horizontal add — bit-identical distances), and
// the final k are selected from those exact keys. Exactness therefore never depends on bf16 rounding, on the MFMA's
// internal accumulation order, or on the data distribution (overflowing candidate lists tighten L1 and re-scan).
// Non-finite scores / norms get lb = -inf, ub = +inf: always re-scored, never used as a bound.
typedef __bf16 vec_bf16x8 __attribute__((ext_vector_type(8)));
static const int VEC_HKC = 64;                  // bf16 K chunk per LDS step = 128 bytes per row, same LDS geometry as the fp32 scan
static const uint64_t VEC_ENT_INVALID = 0xFFFFFFFFFFFFFFFFull;

// descending-order key of a float: larger value -> smaller key; NaN never produced by callers
__device__ inline uint32_t f32_desc_key(float f) { return ~f32_ord(f); }
__device__ inline float desc_key_f32(uint32_t k) { return ord_f32(~k); }
__device__ inline bool f32_finite(float f) { return (__float_as_uint(f) & 0x7F800000u) != 0x7F800000u; }

// fp32 -> bf16, round to nearest even; finite values that would round to infinity saturate to the largest finite
// bf16 (relative error still <= 2^-8), NaN stays NaN, +-inf stays +-inf
__device__ inline uint32_t f32_to_bf16_bits(float f) {
    uint32_t u = __float_as_uint(f);
    const uint32_t a = u & 0x7FFFFFFFu;
    if (a > 0x7F800000u) return (u >> 16) | 0x0040u;
    if (a == 0x7F800000u) return u >> 16;
    u += 0x7FFFu + ((u >> 16) & 1u);
    uint32_t h = u >> 16;
    if ((h & 0x7FFFu) >= 0x7F80u) h = (h & 0x8000u) | 0x7F7Fu;
    return h;
}

// HBM layouts of the bf16 mirrors (private to this library, chosen for the scan kernel's access pattern: every
// pipeline step of a workgroup reads ONE contiguous 16 KB block of rows and one contiguous block of queries — a
// row-major mirror would make each step gather 128-byte pieces at a 2*dim-byte stride, which starves HBM):
//   rows    Xh[tile = row/128][chunk = k/64][row%128][k%64]      (tile = 128 rows, whole tiles are allocated)
//   queries Qh[chunk = k/64][q][k%64]                             (q padded to n_pad, a multiple of 128)
__device__ inline size_t vec_xh_index(uint32_t row, uint32_t k, uint32_t n_chunks) {
    return ((size_t)(row >> 7) * n_chunks + (k >> 6)) * (size_t)(VEC_ROWS * VEC_HKC) + (size_t)(row & 127) * VEC_HKC + (k & 63);
}
__device__ inline size_t vec_qh_index(uint32_t q, uint32_t k, uint32_t n_pad) { return ((size_t)(k >> 6) * n_pad + q) * VEC_HKC + (k & 63); }

// one wave per row: bf16 copy (zero padded to dimp, a multiple of 64) in the tiled layout + norm_out[row] = scale * ||x||_2
// (fp32 sum of squares, `scale` carries the safety inflation / the query-side constant c). q_pad == 0: row layout, else
// query layout with n_pad = q_pad.
__global__ __launch_bounds__(256) void vec_to_bf16_kernel(const float* __restrict__ X, uint16_t* __restrict__ Xh, float* __restrict__ norm_out,
                                                           uint32_t row0, uint32_t n_rows, uint32_t dim, uint32_t dimp, float scale, uint32_t q_pad) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n_rows) return;                       // whole wave exits together
    const uint32_t r = row0 + i;
    const float* __restrict__ x = X + (size_t)r * dim;
    const uint32_t n_chunks = dimp / VEC_HKC;
    float ss = 0.0f;
    for (uint32_t k = 2 * lane; k < dimp; k += 128) {
        const float a = k < dim ? x[k] : 0.0f, b = k + 1 < dim ? x[k + 1] : 0.0f;
        ss = fmaf(a, a, ss);
        ss = fmaf(b, b, ss);
        const size_t at = q_pad ? vec_qh_index(r, k, q_pad) : vec_xh_index(r, k, n_chunks);
        *(uint32_t*)(Xh + at) = f32_to_bf16_bits(a) | (f32_to_bf16_bits(b) << 16);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    if (lane == 0) norm_out[r] = sqrtf(ss) * scale;
}

// per 128-row tile: max of the row norms (+inf if any is NaN / inf) — the scan epilogue's cheap bound
__global__ void vec_tile_nmax_kernel(const float* __restrict__ xnorm, float* __restrict__ tile_nmax, uint32_t tile0, uint32_t n_tiles, uint32_t n_rows) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tiles) return;
    const uint32_t tile = tile0 + i;
    float m = 0.0f;
    for (uint32_t r = tile * VEC_ROWS; r < (tile + 1) * VEC_ROWS && r < n_rows; r++) {
        const float v = xnorm[r];
        if (!f32_finite(v)) m = __uint_as_float(0x7F800000u);
        else if (v > m) m = v;
    }
    tile_nmax[tile] = m;
}

struct VecHScanArgs {
    const uint16_t* Xh;        // bf16 rows, tiled layout (vec_xh_index)
    const uint8_t* row_ok;     // nullable
    const uint16_t* Qh;        // bf16 queries, chunk-major layout (vec_qh_index), n_q_pad rows per chunk
    uint32_t n_q_pad;
    const float* tile_nmax;    // [n_tiles]
    const float* cq;           // [n_q] c * ||q||
    uint32_t n_rows, dimp, n_q;
    uint32_t n_ord, tile_stride, ord_per_slab, n_slabs, n_qtiles;
    int mode;                  // 0 = filtered scan (candidate segments), 1 = sample (group maxima of the lower bounds)
    const float* L1;           // mode 0: [n_q] thresholds
    // mode 0: candidates go to a segment PRIVATE to (slab, query) — slots come from an LDS counter, no global atomics:
    //   seg[(slab * n_q + q) * seg_cap + slot] = (s~ bits << 32 | row),  seg_cnt[slab * n_q + q] = candidates seen (may exceed seg_cap)
    uint64_t* seg;
    uint32_t* seg_cnt;
    uint32_t seg_cap;
    // mode 1: gmax[q * gstride + ordinal * 4 + g] = descending key of the largest lower bound among the 32 rows of group g
    // (g = 64-row strip * 2 + half) of the ordinal-th sampled tile; 0xFFFFFFFF = no valid row in the group
    uint32_t* gmax;
    uint32_t gstride;
};

// 16 bytes per lane straight from global memory into LDS (LDS-DMA, global_load_lds_dwordx4): the destination is
// wave-uniform base (M0) + lane * 16, the source address is per lane — so the LDS image is linear and any swizzle is applied
// on the SOURCE side. Issued through inline asm ON PURPOSE: hipcc cannot prove that the DMA into ring slot buf^1 does not
// alias the ds_reads of slot buf and would drain it (s_waitcnt vmcnt(0)) before the first operand fetch of every step; an asm
// statement is invisible to its counters, and the one wait this pipeline needs is vec_glds_wait() before the step's barrier
// (cdna_hip_programming.md §5.7: M0 is written in the same statement that reads it; s_nop 0 covers the M0 hazard).
__device__ inline void vec_glds16(const void* gsrc, void* lds_wave_base) {
#ifdef TSGPU_HIP_EMU
    hipemu_global_load_lds16(gsrc, lds_wave_base);
#else
    const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_wave_base);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
#endif
}
template <int N>
__device__ inline void vec_glds_wait() {        // all but the youngest N vector-memory operations of this wave are complete
#ifndef TSGPU_HIP_EMU
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}

// LDS image of one pipeline step: rows of 128 bytes (64 bf16), NO padding; the eight 16-byte pieces of row R are stored
// at piece position p ^ ((R >> 1) & 7). With that XOR the 16 lanes of every ds_read_b128 service group (MI355X_MICROARCH.md
// §LDS) hit 16 distinct 4-bank groups: conflict-free operand fetch from an unpadded, DMA-filled tile.
//
// Workgroup = 8 waves = TWO consecutive tile ordinals (256 rows) x QT queries; wave (wr = wave>>1, wc = wave&1) owns rows
// [64 wr, 64 wr + 64) x queries [32 CB wc, +32 CB). One workgroup per CU (128 KB of LDS), 2 waves per SIMD.
// Pipeline (memory latency under load is thousands of cycles, a step's MFMAs only ~0.5K): both rings have 3 slots and run
// TWO steps ahead. Per step every wave issues its share of Q(s+2) and X(s+2), multiplies slot s, then s_waitcnt
// vmcnt(XV+QV) — loads complete in issue order, so everything but this step's own DMAs has landed, i.e. X(s+1) and Q(s+1) —
// and the workgroup barrier. (Dedicated loader waves were measured slower: two waves cannot issue a step's 48 DMAs as fast
// as eight do, and the limiter of the combined loop is LDS bandwidth — DMA writes + operand reads — not MFMA issue.)
// Round-2 measurements at B = 256 on 10M x 768 (tools/exp_vec2.py), none of which beat this form (scan 4.6 ms):
//   * ablations: row blocks only 2.99 ms (5.1 TB/s), query blocks only (L2-resident) 2.4-2.5 ms (6.1-6.5 TB/s chip-wide = ~25 GB/s
//     per CU: the LDS-DMA landing rate of a CU, MI355X_MICROARCH.md "ldsdma-fill"), both 3.5 ms (8.7 TB/s of LDS-DMA), MFMA + operand
//     reads without DMA 3.1 ms; at B = 64 the same kernel streams the rows at 5.9-6.2 TB/s (of ~6.3 achievable);
//   * separate rings — waves 0-3 issue only row blocks (5-7 slots), waves 4-7 only query blocks (2-4 slots), so more ROW bytes are
//     in flight per CU (80-96 KB instead of 48): 5.0-5.5 ms. Depth is not the limiter;
//   * workgroups of an XCD walking the k chunks in rotated order (no two ask L2 for the same query block at once): no change;
//   * query blocks through registers (global_load -> ds_write_b128 by waves 4-7) to leave the DMA path to the rows: 6.3-8 ms (the
//     loading waves stall their own MFMA stream on L2 latency);
//   * (round 3) the non-temporal hint on the row-block DMAs (`global_load_lds_dwordx4 ... nt`; each row block is read by ONE workgroup at B = 256):
//     4.9 -> 5.9 ms; `s_setprio 1` for the second-dispatched half of the workgroup: within run-to-run noise (+-3 %) in an A/B/A/B run;
//   * row operand straight from global memory into the MFMA's registers (a 16-byte load per lane = its 8 k of a row; a re-tiled mirror makes
//     a wave-wide load 1 KB contiguous; 4-deep register ring, asm-issued loads with counted waits), only the query block in LDS: half the
//     DMA landings, two thirds of the operand fetches, identical results, 4.74-4.81 ms. Its ablations: ds_read_b128 + barriers alone 2.07 ms,
//     + MFMAs 3.18 ms, all data movement without MFMAs 3.57 ms (tools/experiments/, profiles/r02/exp_vec_rows_direct.txt).
static const int VEC_HTHREADS = 512;
static const int VEC_HROWS = 2 * VEC_ROWS;          // rows per workgroup step (two 128-row tiles)
static const int VEC_HMAX_PER = 2048;               // tile ordinals per slab whose norm maxima are staged in LDS
// BK = k elements per pipeline step: 64 (rows of 128 B, 8 pieces, swizzle (R>>1)&7) for the 64 / 128-query tiles; 32 (rows of
// 64 B, 4 pieces, swizzle (R>>2)&3 — the same 16-distinct-bank-groups argument) for the 256-query tile, whose wave tile is
// 64 x 128 (CB = 4): 6 operand fetches feed 8 MFMAs instead of 4 feeding 4, and the row block is DMA'd once per 256 queries.
template <int QT, int BK, int NS>
struct VecHScanSmem {
    alignas(16) uint32_t xs[NS][VEC_HROWS * BK / 2];
    alignas(16) uint32_t qs[NS][QT * BK / 2];
    float nmax[VEC_HMAX_PER];  // tile_nmax of the slab's ordinals
    uint32_t cnt[QT];          // mode 0: candidates of this workgroup per query column
};

// (round 4's int8 mirror — v_mfma_i32_32x32x32_i8 over per-tile quantised rows: the scan itself 3.1 ms instead of 4.4, but its wider bracket sent 3-4x the rows
//  to the exact re-score, a net loss end to end; profiles/r04/exp_vec_int8.txt — was removed in round 6)
template <int CB>
__global__ __launch_bounds__(VEC_HTHREADS) void vec_hscan_kernel(VecHScanArgs a) {
    constexpr int QT = 64 * CB;
    constexpr int BK = CB >= 4 ? 32 : 64;              // k per step
    constexpr int PR = BK / 8;                          // 16-byte pieces per LDS row
    constexpr int RW = BK / 2;                          // words per LDS row
    constexpr int KS = BK / 16;                         // MFMA k-steps per pipeline step
    constexpr int SWS = PR == 8 ? 1 : 2;                // swizzle = (R >> SWS) & (PR - 1)
    constexpr int NS = CB >= 4 ? 4 : 3;                 // ring slots: the rings run NS - 1 steps ahead (32 KB slots at CB = 4 leave room for four)
    __shared__ VecHScanSmem<QT, BK, NS> sm;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t wr = wave >> 1;
    const uint32_t wrow = wr * 64, wcol = (wave & 1) * (32 * CB);
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7, j = b >> 3;
    const uint32_t qtile = j % a.n_qtiles;
    const uint32_t slab = (j / a.n_qtiles) * 8 + xcd;
    const uint32_t q0 = qtile * QT;
    const uint32_t ord_begin = slab * a.ord_per_slab;
    uint32_t ord_end = ord_begin + a.ord_per_slab;
    if (ord_end > a.n_ord) ord_end = a.n_ord;
    if (ord_begin >= ord_end) return;
    const uint32_t n_ord = ord_end - ord_begin;          // <= VEC_HMAX_PER (host)
    const uint32_t n_chunks = a.dimp / BK;               // pipeline steps per tile pair
    const uint32_t n_c64 = a.dimp / VEC_HKC;             // 64-wide chunks of the mirrors' layout
    const uint32_t total_steps = ((n_ord + 1) / 2) * n_chunks;

    for (uint32_t i = t; i < (uint32_t)QT; i += VEC_HTHREADS) sm.cnt[i] = 0;
    for (uint32_t i = t; i < n_ord; i += VEC_HTHREADS) sm.nmax[i] = a.tile_nmax[(ord_begin + i) * a.tile_stride];
    __syncthreads();                                     // plain loads are done before the first DMA is issued

    // Thread t moves pieces idx = t + v*512 of a step: LDS position idx (linear image), LDS row R = idx / PR (rows 0..127 = first
    // ordinal of the pair, 128..255 = second), source piece (idx % PR) ^ swizzle(R) of that row's BK-wide slice. Whole tiles /
    // padded query rows exist in memory: no clamping of rows (scores of rows >= n_rows / queries >= n_q are dropped by the epilogue).
    constexpr int XV = VEC_HROWS * PR / VEC_HTHREADS;    // 16-byte pieces per thread per X step (4 or 2)
    constexpr int QV = QT * PR / VEC_HTHREADS;           // 1, 2 or 2
    uint32_t src_off[XV > QV ? XV : QV];                 // 16-byte units inside a 128-rows x 128-byte block (row * 8 + piece)
#pragma unroll
    for (int v = 0; v < (XV > QV ? XV : QV); v++) {
        const uint32_t idx = t + v * VEC_HTHREADS, R = idx / PR;
        src_off[v] = (R & 127) * 8 + ((idx % PR) ^ ((R >> SWS) & (PR - 1)));
    }
    auto load_x = [&](uint32_t s, uint32_t slot) {
        const uint32_t p = s / n_chunks, ss = s % n_chunks;
        const uint32_t c = ss * BK / VEC_HKC, sub = (ss * BK % VEC_HKC) / 8;           // 64-chunk and first piece of this step's slice
#pragma unroll
        for (int v = 0; v < XV; v++) {
            const uint32_t half = (t + v * VEC_HTHREADS) / (128 * PR);                  // which of the two ordinals (uniform per v)
            uint32_t o = ord_begin + 2 * p + half;
            o = o < ord_end ? o : ord_end - 1;                                         // odd tail: re-read the last ordinal (dropped by the epilogue)
            const uint4* __restrict__ xsrc = (const uint4*)(a.Xh + ((size_t)(o * a.tile_stride) * n_c64 + c) * (size_t)(VEC_ROWS * VEC_HKC));
            vec_glds16(xsrc + src_off[v] + sub, &sm.xs[slot][(v * VEC_HTHREADS + wave * 64) * 4]);
        }
    };
    auto load_q = [&](uint32_t s, uint32_t slot) {
        const uint32_t ss = s % n_chunks;
        const uint32_t c = ss * BK / VEC_HKC, sub = (ss * BK % VEC_HKC) / 8;
        const uint4* __restrict__ qsrc = (const uint4*)(a.Qh + ((size_t)c * a.n_q_pad + q0) * VEC_HKC);
#pragma unroll
        for (int v = 0; v < QV; v++) {
            const uint32_t idx = t + v * VEC_HTHREADS, R = idx / PR;                    // query rows are not folded to 128
            vec_glds16(qsrc + R * 8 + ((idx % PR) ^ ((R >> SWS) & (PR - 1))) + sub, &sm.qs[slot][(v * VEC_HTHREADS + wave * 64) * 4]);
        }
    };

    // per-lane query constants: this lane's query column in each of its CB blocks
    float L1v[CB], cqv[CB];
#pragma unroll
    for (int cb = 0; cb < CB; cb++) {
        const uint32_t gq = q0 + wcol + cb * 32 + (lane & 31);
        const uint32_t gc = gq < a.n_q ? gq : a.n_q - 1;
        cqv[cb] = a.cq[gc];
        L1v[cb] = (a.mode == 0) ? a.L1[gc] : __uint_as_float(0xFF800000u);
    }
    __syncthreads();                                     // (their waits drain nothing of ours: no DMA issued yet)

    const uint32_t last = total_steps - 1;
    vec_f32x16 acc[2][CB];
    // The per-lane constants above come from plain global loads, and hipcc waits for a load where its value is FIRST USED — here the tile
    // epilogue, INSIDE the pipelined loop: `s_waitcnt vmcnt(3) .. vmcnt(0)` before the four `cq * nmax` products, in every epilogue. Its counter
    // model knows nothing of the asm-issued LDS-DMAs, the hardware counter does: each of those waits drained the DMA ring, once per tile and
    // wave (the "epilogue" that cost 0.5 of the scan's 4.6 ms in round 2's ablation was mostly this). Using the values here makes the
    // compiler wait NOW, before the first DMA is in flight, and leaves nothing of its own pending inside the loop.
#ifndef TSGPU_HIP_EMU
#pragma unroll
    for (int cb = 0; cb < CB; cb++) asm volatile("" ::"v"(cqv[cb]), "v"(L1v[cb]));
#endif
#pragma unroll
    for (int i = 0; i < NS - 1; i++) {
        load_q((uint32_t)i < total_steps ? i : last, i);
        load_x((uint32_t)i < total_steps ? i : last, i);
    }
    vec_glds_wait<(NS - 2) * (XV + QV)>();                 // step 0 has landed, the later ones may stay in flight
    __syncthreads();
    // operand addresses: lane (r = lane&31, h = lane>>5) reads piece 2g+h of its rows = k 16g+8h .. +7: one ds_read_b128 = one
    // MFMA operand; piece position = (2g+h) ^ swizzle(r) (rows of a lane differ by multiples of 32 -> same swizzle)
    const uint32_t swz = ((lane & 31) >> SWS) & (PR - 1), hh = lane >> 5;
    uint32_t poff[KS];
#pragma unroll
    for (int g = 0; g < KS; g++) poff[g] = (((uint32_t)(2 * g) + hh) ^ swz) * 4;
    uint32_t xslot = 0;                                  // s % NS
    uint4 av[2][2], bv[2][CB];                           // operand double buffer, carried across steps
    {
        const uint32_t* xa0 = &sm.xs[0][(wrow + (lane & 31)) * RW];
        const uint32_t* qb0 = &sm.qs[0][(wcol + (lane & 31)) * RW];
#pragma unroll
        for (int rb = 0; rb < 2; rb++) av[0][rb] = *(const uint4*)(xa0 + rb * 32 * RW + poff[0]);
#pragma unroll
        for (int cb = 0; cb < CB; cb++) bv[0][cb] = *(const uint4*)(qb0 + cb * 32 * RW + poff[0]);
    }
    for (uint32_t s = 0; s < total_steps; s++) {
        const uint32_t c = s % n_chunks;
        if (c == 0) {
#pragma unroll
            for (int rb = 0; rb < 2; rb++)
#pragma unroll
                for (int cb = 0; cb < CB; cb++)
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[rb][cb][e] = 0.0f;
        }
        // no branch around the DMAs: steps past the end re-request the last blocks into slots nobody reads any more
#if !defined(VEC_ABL) || !(VEC_ABL & 4)   // VEC_ABL: tools/ ablation builds only (bit 0 no epilogue, bit 1 no MFMA, bit 2 no DMA, bit 3 no operand fetches, bit 4 no step barrier, bit 5 one query DMA in seven skipped)
#if defined(VEC_ABL) && (VEC_ABL & 32)     // (results WRONG) the query DMA of one k chunk in seven is skipped: what an LDS-RESIDENT seventh of the query block (all that fits beside the rings) would save
        if ((s + NS - 1) % n_chunks % 7 != 0)
#endif
        load_q(s + NS - 1 < total_steps ? s + NS - 1 : last, xslot >= 1 ? xslot - 1 : NS - 1);       // (s + NS - 1) % NS
        load_x(s + NS - 1 < total_steps ? s + NS - 1 : last, xslot >= 1 ? xslot - 1 : NS - 1);
#endif
        const uint32_t* xa = &sm.xs[xslot][(wrow + (lane & 31)) * RW];
        const uint32_t* qb = &sm.qs[xslot][(wcol + (lane & 31)) * RW];
#pragma unroll
        for (int g = 0; g < KS; g++) {
            const int cur = g & 1, nx = cur ^ 1;                 // KS is even: every step starts on buffer 0
#if defined(VEC_ABL) && (VEC_ABL & 8)
            if (g + 1 == KS) {
                vec_glds_wait<(NS - 2) * (XV + QV)>();
#if !(VEC_ABL & 16)
                __syncthreads();
#endif
            }
            av[nx][0] = av[cur][0]; av[nx][1] = av[cur][1];
#pragma unroll
            for (int cb = 0; cb < CB; cb++) bv[nx][cb] = bv[cur][cb];
            if (true) {} else
#endif
            if (g + 1 < KS) {
#pragma unroll
                for (int rb = 0; rb < 2; rb++) av[nx][rb] = *(const uint4*)(xa + rb * 32 * RW + poff[g + 1 < KS ? g + 1 : g]);
#pragma unroll
                for (int cb = 0; cb < CB; cb++) bv[nx][cb] = *(const uint4*)(qb + cb * 32 * RW + poff[g + 1 < KS ? g + 1 : g]);
            } else {
                // The step's barrier sits HERE, before the last MFMA group, not after it: every operand of this step is in registers,
                // so slot s is free; the next step's blocks have landed (counted wait), and its first operands are requested right
                // behind the barrier — their LDS latency and the waves' barrier skew hide under the last group's MFMAs instead of
                // stalling every wave at the top of the next step.
                vec_glds_wait<(NS - 2) * (XV + QV)>();             // all but the youngest NS - 2 steps' DMAs have landed: X(s+1), Q(s+1)
                __syncthreads();
                const uint32_t nslot = xslot == NS - 1 ? 0 : xslot + 1;
                const uint32_t* xa1 = &sm.xs[nslot][(wrow + (lane & 31)) * RW];
                const uint32_t* qb1 = &sm.qs[nslot][(wcol + (lane & 31)) * RW];
#pragma unroll
                for (int rb = 0; rb < 2; rb++) av[nx][rb] = *(const uint4*)(xa1 + rb * 32 * RW + poff[0]);
#pragma unroll
                for (int cb = 0; cb < CB; cb++) bv[nx][cb] = *(const uint4*)(qb1 + cb * 32 * RW + poff[0]);
            }
#if defined(VEC_ABL) && (VEC_ABL & 2)
#pragma unroll
            for (int rb = 0; rb < 2; rb++)
#pragma unroll
                for (int cb = 0; cb < CB; cb++) acc[rb][cb][0] += (decltype(acc[rb][cb][0] + 0))__uint_as_float(av[cur][rb].x ^ bv[cur][cb].y);
#else
#pragma unroll
            for (int rb = 0; rb < 2; rb++)
#pragma unroll
                for (int cb = 0; cb < CB; cb++) {
                    acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(vec_bf16x8, av[cur][rb]), __builtin_bit_cast(vec_bf16x8, bv[cur][cb]),
                                                                               acc[rb][cb], 0, 0, 0);
                }
#endif
        }
#if defined(VEC_ABL) && (VEC_ABL & 1)
        if (c == n_chunks - 1) {
            float keep = 0.0f;
#pragma unroll
            for (int rb = 0; rb < 2; rb++)
#pragma unroll
                for (int cb = 0; cb < CB; cb++)
#pragma unroll
                    for (int el = 0; el < 16; el++) keep = fmaxf(keep, (float)acc[rb][cb][el]);
            if (keep == 1.2345e30f) a.seg_cnt[0] = 1;
        }
        if (false) {
#else
        if (c == n_chunks - 1) {
#endif
            // ---- tile epilogue: this wave's 64 rows belong to ordinal 2p + (wr >> 1) ----
            const uint32_t oi = 2 * (s / n_chunks) + (wr >> 1);           // ordinal index inside the slab
            if (oi < n_ord) {
                const uint32_t o = ord_begin + oi;
                const uint32_t r0 = o * a.tile_stride * VEC_ROWS;
                const float nmax = sm.nmax[oi];
                const uint32_t strip = wr & 1;                               // 64-row strip inside the tile
                const uint32_t rbase = r0 + strip * 64 + 4 * (lane >> 5);
#pragma unroll
                for (int cb = 0; cb < CB; cb++) {
                    const uint32_t col = wcol + cb * 32 + (lane & 31);
                    const uint32_t gq = q0 + col;
                    const float e = cqv[cb] * nmax + 1e-30f;             // >= every row's error radius in this tile
                    if (a.mode == 0) {
                        // keep a row iff its upper bound reaches L1: !(sc + e < L1), tested as !(sc < L1 - e) — the 1 % slack
                        // inside c dwarfs the rounding of that subtraction. Non-finite rows / queries have e = inf or NaN, so
                        // thr is -inf / NaN and every compare below passes: nothing non-finite is ever rejected.
                        const float thr = L1v[cb] - e;
                        float amax = acc[0][cb][0];
#pragma unroll
                        for (int rb = 0; rb < 2; rb++)
#pragma unroll
                            for (int el = 0; el < 16; el++) amax = fmaxf(amax, acc[rb][cb][el]);       // v_max3_f32 chain
                        if (gq < a.n_q && !(amax < thr)) {
                            // (nearly every wave has SOME lane in here for every column block — ~1.2 candidates per (wave, block) at 10M rows — so what
                            // counts is the instructions the wave issues inside: the 32 scores are tested in groups of 4 behind their maximum, and only
                            // a group that holds a candidate for some lane is walked)
                            uint64_t* __restrict__ seg = a.seg + ((size_t)slab * a.n_q + gq) * a.seg_cap;
#pragma unroll
                            for (int rb = 0; rb < 2; rb++)
#pragma unroll
                                for (int g4 = 0; g4 < 4; g4++) {
                                    const float gm = fmaxf(fmaxf(acc[rb][cb][4 * g4], acc[rb][cb][4 * g4 + 1]), fmaxf(acc[rb][cb][4 * g4 + 2], acc[rb][cb][4 * g4 + 3]));
                                    if (gm < thr) continue;
#pragma unroll
                                    for (int e4 = 0; e4 < 4; e4++) {
                                        const int el = 4 * g4 + e4;
                                        const uint32_t row = rbase + rb * 32 + (el & 3) + 8 * (el >> 2);
                                        const float sc = acc[rb][cb][el];
                                        if (!(sc < thr) && row < a.n_rows && (!a.row_ok || a.row_ok[row] != 0)) {
                                            const uint32_t slot = atomicAdd(&sm.cnt[col], 1u);      // LDS
                                            if (slot < a.seg_cap) seg[slot] = ((uint64_t)__float_as_uint(sc) << 32) | row;
                                        }
                                    }
                                }
                        }
                    } else if (gq < a.n_q) {
                        // sample pass: one key per (query, 32-row group) — the group's best lower bound. The k-th largest of
                        // these maxima is a valid lower bound of the k-th best score (k groups -> k distinct rows).
                        float best = __uint_as_float(0xFF800000u);
#pragma unroll
                        for (int rb = 0; rb < 2; rb++)
#pragma unroll
                            for (int el = 0; el < 16; el++) {
                                const uint32_t row = rbase + rb * 32 + (el & 3) + 8 * (el >> 2);
                                bool okr = row < a.n_rows;
                                if (okr && a.row_ok) okr = a.row_ok[row] != 0;
                                const float lb = acc[rb][cb][el] - e;
                                if (okr && f32_finite(lb) && lb > best) best = lb;
                            }
                        a.gmax[(size_t)gq * a.gstride + (size_t)o * 4 + strip * 2 + (lane >> 5)] = f32_finite(best) ? f32_desc_key(best) : 0xFFFFFFFFu;
                    }
                }
            }
        }
        xslot = xslot == NS - 1 ? 0 : xslot + 1;
    }
    __syncthreads();                                     // sm.cnt is complete
    if (a.mode == 0)
        for (uint32_t i = t; i < (uint32_t)QT; i += VEC_HTHREADS)
            if (q0 + i < a.n_q) a.seg_cnt[(size_t)slab * a.n_q + q0 + i] = sm.cnt[i];
}

// block-wide exact k-th smallest of n 32-bit keys produced by key(i) (re-evaluated every radix pass): 4 passes of
// 8 bits, LDS histogram. Every thread of the 256-thread block must call it. Returns the key (all threads).
template <class KeyFn>
__device__ inline uint32_t block_kth_smallest_u32(uint32_t n, uint32_t k, KeyFn key, uint32_t* hist /*[256]*/, uint32_t* s_state /*[2]*/) {
    const uint32_t t = threadIdx.x;
    uint32_t prefix = 0, pmask = 0, krem = k;
    for (int p = 3; p >= 0; p--) {
        hist[t] = 0;
        __syncthreads();
        const int shift = 8 * p;
        for (uint32_t i = t; i < n; i += VEC_THREADS) {
            const uint32_t kv = key(i);
            if ((kv & pmask) == prefix) atomicAdd(&hist[(kv >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (t == 0) {
            uint32_t cum = 0, d = 0;
            for (; d < 255; d++) { if (cum + hist[d] >= krem) break; cum += hist[d]; }
            s_state[0] = krem - cum;
            s_state[1] = prefix | (d << shift);
        }
        __syncthreads();
        krem = s_state[0];
        prefix = s_state[1];
        pmask |= 0xFFu << shift;
        __syncthreads();
    }
    return prefix;
}

// sample pass -> L1[q] = k-th largest group maximum of the sample's lower bounds (-inf when fewer than k groups hold a row)
__global__ __launch_bounds__(VEC_THREADS) void vec_thresh_kernel(const uint32_t* __restrict__ gmax, size_t stride, uint32_t n, uint32_t k, float* __restrict__ L1) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_state[2], s_valid;
    const uint32_t t = threadIdx.x, q = blockIdx.x;
    const uint32_t* __restrict__ keys = gmax + (size_t)q * stride;
    if (t == 0) s_valid = 0;
    __syncthreads();
    uint32_t valid = 0;
    for (uint32_t i = t; i < n; i += VEC_THREADS) valid += keys[i] != 0xFFFFFFFFu;
    if (valid) atomicAdd(&s_valid, valid);
    __syncthreads();
    const bool enough = s_valid >= k;
    __syncthreads();
    const uint32_t kk = block_kth_smallest_u32(n, k, [&](uint32_t i) { return keys[i]; }, hist, s_state);
    if (t == 0) L1[q] = enough ? desc_key_f32(kk) : __uint_as_float(0xFF800000u);
}

// one workgroup per query. Gathers the query's candidate segments (one per slab), L2 = k-th largest lower bound
// s~ - c|q||x_row| among them (exact per-row norms), survivors (upper bound >= L2) -> surv rows for the exact re-score.
// The lower-bound keys of the gathered entries live in LDS for the radix passes (VEC_REFINE_LCAP entries); a query with more
// candidates (its k-th neighbour sits in a cluster of tens of thousands of near-ties) spills the rest to its slice of a global list
// (gkeys, gcap entries per query; the radix passes then read both).
//   overflow[0] += 1 : some segment or both lists overflowed -> L1[q] was raised to the L2 of what is held (valid, tighter)
//                      and the host scans again;
//   overflow[1] += 1 : stuck — the bound cannot move (mass ties) or more than surv_cap rows sit inside the bracket of the k-th
//                      best (near-duplicates): the host runs the group on the fp32 scan.
//   overflow[2] += 1 : a (slab, query) segment overflowed; overflow[3] += 1 : ... and the bound could not move: the host grows the
//                      segments and scans again (stuck for good only when they cannot grow).
static const uint32_t VEC_REFINE_LCAP = 24576;
__global__ __launch_bounds__(VEC_THREADS) void vec_refine_kernel(const uint64_t* __restrict__ seg, const uint32_t* __restrict__ seg_cnt, uint32_t n_slabs,
                                                                  uint32_t n_q, uint32_t seg_cap, uint32_t k, const float* __restrict__ cq,
                                                                  const float* __restrict__ xnorm, float* __restrict__ L1, uint32_t* __restrict__ surv_base,
                                                                  uint32_t surv_cap, uint32_t* __restrict__ surv_cnt, uint32_t* __restrict__ overflow,
                                                                  uint32_t* __restrict__ gkeys_base, uint32_t gcap) {
    __shared__ uint32_t lkeys[VEC_REFINE_LCAP];
    uint32_t* __restrict__ gkeys = gkeys_base + (size_t)blockIdx.x * gcap;
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_state[2], s_n, s_valid, s_surv, s_over;
    const uint32_t t = threadIdx.x, q = blockIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t* __restrict__ surv = surv_base + (size_t)q * surv_cap;
    const float cqq = cq[q];
    const float NEG_INF = __uint_as_float(0xFF800000u), POS_INF = __uint_as_float(0x7F800000u);
    if (t == 0) { s_n = 0; s_valid = 0; s_surv = 0; s_over = 0; }
    __syncthreads();
    // lower / upper bound of an entry; non-finite -> (-inf, +inf): always re-scored, never used as a bound
    auto bounds = [&](uint64_t ev, float& lb, float& ub) {
        const uint32_t row = (uint32_t)ev;
        const float sc = __uint_as_float((uint32_t)(ev >> 32));
        const float e = cqq * xnorm[row] + 1e-30f;
        lb = sc - e; ub = sc + e;
        if (!f32_finite(sc) || !f32_finite(e) || !f32_finite(lb) || !f32_finite(ub)) { lb = NEG_INF; ub = POS_INF; }
    };
    // ---- gather: wave w takes slabs w, w+4, ...; lanes stride over the segment ----
    for (uint32_t sl = wave; sl < n_slabs; sl += VEC_THREADS / 64) {
        const uint32_t total = seg_cnt[(size_t)sl * n_q + q];
        const uint32_t n = total < seg_cap ? total : seg_cap;
        if (total > seg_cap && lane == 0) s_over = 1;
        const uint64_t* __restrict__ sp = seg + ((size_t)sl * n_q + q) * seg_cap;
        for (uint32_t j = lane; j < n; j += 64) {
            float lb, ub;
            bounds(sp[j], lb, ub);
            const uint32_t slot = atomicAdd(&s_n, 1u);
            const uint32_t kv = lb == NEG_INF ? 0xFFFFFFFEu : f32_desc_key(lb);
            if (slot < VEC_REFINE_LCAP) lkeys[slot] = kv;
            else if (slot - VEC_REFINE_LCAP < gcap) gkeys[slot - VEC_REFINE_LCAP] = kv;
        }
    }
    __syncthreads();
    // (the barrier above orders the spilled keys' global stores before the reads below at workgroup scope)
    const uint32_t n_all = s_n;
    const uint32_t hold = VEC_REFINE_LCAP + gcap;
    const uint32_t n_held = n_all < hold ? n_all : hold;
    const bool over = s_over != 0 || n_all > hold;
    auto key_at = [&](uint32_t i) { return i < VEC_REFINE_LCAP ? lkeys[i] : gkeys[i - VEC_REFINE_LCAP]; };
    uint32_t valid = 0;
    for (uint32_t i = t; i < n_held; i += VEC_THREADS) valid += key_at(i) < 0xFFFFFFFEu;
    if (valid) atomicAdd(&s_valid, valid);
    __syncthreads();
    const bool enough = s_valid >= k;
    __syncthreads();
    const uint32_t kk = block_kth_smallest_u32(n_held, k, key_at, hist, s_state);
    const float L2 = enough ? desc_key_f32(kk) : NEG_INF;
    if (over) {
        if (t == 0) {
            const bool raised = L2 > L1[q];
            if (raised) L1[q] = L2;
            if (s_over) atomicAdd(overflow + 2, 1u);             // a (slab, query) segment was too small: the host may grow the segments
            if (!raised) atomicAdd(overflow + (s_over ? 3 : 1), 1u);   // [1]: stuck for good; [3]: stuck unless the segments grow
            atomicAdd(overflow, 1u);
            surv_cnt[q] = 0;
        }
        return;
    }
    // ---- survivors: second walk over the segments ----
    for (uint32_t sl = wave; sl < n_slabs; sl += VEC_THREADS / 64) {
        const uint32_t n = seg_cnt[(size_t)sl * n_q + q];          // <= seg_cap here
        const uint64_t* __restrict__ sp = seg + ((size_t)sl * n_q + q) * seg_cap;
        for (uint32_t j = lane; j < n; j += 64) {
            float lb, ub;
            const uint64_t ev = sp[j];
            bounds(ev, lb, ub);
            if (!(ub < L2)) { const uint32_t slot = atomicAdd(&s_surv, 1u); if (slot < surv_cap) surv[slot] = (uint32_t)ev; }
        }
    }
    __syncthreads();
    if (t == 0) {
        if (s_surv > surv_cap) { atomicAdd(overflow, 1u); atomicAdd(overflow + 1, 1u); surv_cnt[q] = 0; }
        else surv_cnt[q] = s_surv;
    }
}

// hnswlib InnerProductSpace::get_dist_func arithmetic for one (query, row) pair, evaluated by a 16-lane group
// (space_ip.h: InnerProductSIMD16Ext / SIMD4Ext / *Residuals): products and sums rounded separately (no FMA),
// 16 (or 4) running lane sums, sequential horizontal add. sub = lane index inside the group (0..15); every lane of
// the group returns the distance. qs = the query (LDS or global), x = the row in global memory.
// Contraction is switched off lexically in every function below (the __fmul_rn/__fadd_rn header intrinsics are plain
// operators compiled under the default -ffp-contract=fast and DO fuse after inlining).
__device__ inline float ip_mul_add(float acc, float a, float b) {
#pragma clang fp contract(off)
    const float pr = a * b;
    return acc + pr;
}
__device__ inline float ip_mul(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ inline float ip_add(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
// ---- hnswlib InnerProductSpace summation orders (space_ip.h; restated in oracle/oracle_index.h: ip_simd16ext / ip_simd4ext) ----
// `lanes` = the SIMD width hnswlib was COMPILED for, which fixes the order the products are added in: 4 = SSE (the reference's stock
// flags pass no -march: CMakeLists.txt:8, BUILD:81-88 — element i accumulates in lane i % 4, final T0+T1+T2+T3; library default),
// 8 = AVX (lane i % 8, T0+..+T7; dims % 4: __m256 over the 16-multiples folded lo+hi into an __m128 for the rest), 16 = AVX-512
// (lane i % 16, T0+..+T15; dims % 4 as AVX). Option "vec_ip_lanes". Every form multiplies and adds with separate roundings.
//
// Generic form: 16 GPU lanes per row (sub = lane & 15), any dim; lanes sub < L own the L accumulator lanes (a strided scalar walk:
// the by-id / odd-dimension paths, not the batched re-score).
__device__ inline float ip_acc_strided(float acc, const float* qs, const float* __restrict__ x, uint32_t from, uint32_t to, uint32_t sub, uint32_t L) {
    if (sub < L) for (uint32_t i = from + sub; i < to; i += L) acc = ip_mul_add(acc, qs[i], x[i]);
    return acc;
}
__device__ inline float ip_hsum(float accl, uint32_t L) {                 // T0 + T1 + .. + T(L-1), left to right, over the 16-lane group's first L lanes
    const int b0 = (int)(threadIdx.x & 48u);
    float sum = 0.0f;
    if (L == 4) {
        const float l0 = __shfl(accl, b0), l1 = __shfl(accl, b0 + 1), l2 = __shfl(accl, b0 + 2), l3 = __shfl(accl, b0 + 3);
        return ip_add(ip_add(ip_add(l0, l1), l2), l3);                    // (SSE form: no leading 0 + ..)
    }
#pragma unroll
    for (int l = 0; l < 16; l++) { const float v = __shfl(accl, b0 + l); if ((uint32_t)l < L) sum = ip_add(sum, v); }
    return sum;
}
__device__ inline float ip_simd16ext(const float* qs, const float* __restrict__ x, uint32_t qty, uint32_t sub, uint32_t L) {    // qty % 16 == 0
    return ip_hsum(ip_acc_strided(0.0f, qs, x, 0, qty, sub, L), L);
}
__device__ inline float ip_simd4ext(const float* qs, const float* __restrict__ x, uint32_t qty, uint32_t sub, uint32_t L) {     // qty % 4 == 0
    if (L == 4) return ip_hsum(ip_acc_strided(0.0f, qs, x, 0, qty, sub, 4), 4);
    const uint32_t q16 = qty / 16 * 16;                                   // InnerProductSIMD4ExtAVX
    const float a8 = ip_acc_strided(0.0f, qs, x, 0, q16, sub, 8);
    const float hi = __shfl(a8, (int)((threadIdx.x & 48u) + ((sub + 4) & 15)));
    const float s4 = ip_add(a8, hi);                                      // lanes 0..3: lo + hi
    return ip_hsum(ip_acc_strided(s4, qs, x, q16, qty, sub, 4), 4);
}
__device__ inline float ip_scalar(const float* qs, const float* __restrict__ x, uint32_t off, uint32_t n) {
    float r = 0.0f;
    for (uint32_t i = 0; i < n; i++) r = ip_mul_add(r, qs[off + i], x[off + i]);
    return r;
}
__device__ inline float ip_distance_group16(const float* qs, const float* __restrict__ x, uint32_t dim, uint32_t sub, uint32_t L) {
    if (dim % 16 == 0) return ip_add(1.0f, -ip_simd16ext(qs, x, dim, sub, L));
    if (dim % 4 == 0) return ip_add(1.0f, -ip_simd4ext(qs, x, dim, sub, L));
    if (dim > 16) { const uint32_t qn = dim >> 4 << 4; const float a1 = ip_simd16ext(qs, x, qn, sub, L); return ip_add(1.0f, -ip_add(a1, ip_scalar(qs, x, qn, dim - qn))); }
    if (dim > 4) { const uint32_t qn = dim >> 2 << 2; const float a1 = ip_simd4ext(qs, x, qn, sub, L); return ip_add(1.0f, -ip_add(a1, ip_scalar(qs, x, qn, dim - qn))); }
    return ip_add(1.0f, -ip_scalar(qs, x, 0, dim));
}

// Fast form for dim % 16 == 0 (the batched re-score and the graph traversal): FOUR GPU lanes per row, 16-byte loads; lane v of the
// quad loads elements 16 j + 4 v .. + 3 of every 16-element block j and forms their four products p[v][0..3]; a wavefront covers 16
// rows per round. Who ADDS which product follows the order:
//   L = 16: element 16 j + 4 v + c belongs to accumulator lane 4 v + c -> lane v keeps four chains of its own products;
//   L = 8:  accumulator lane (4 v + c) % 8 -> quad lanes 0 / 1 own lanes 0..3 / 4..7 and add their own products, then those of lane
//           v + 2 (one quad_perm exchange per block);
//   L = 4:  accumulator lane c takes p[0][c], p[1][c], p[2][c], p[3][c] in that order -> the 4 x 4 block of products is TRANSPOSED
//           inside the quad (two butterfly stages over quad_perm), quad lane c owns accumulator lane c.
// Same products, same order inside every accumulator lane, same left-to-right final sum as the CPU forms; U = blocks requested per
// lane before the first product is consumed (memory-level parallelism: a 768-dimension row costs two round trips).
__device__ inline float quad_xor1(float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0xB1, 0xF, 0xF, true)); }   // quad_perm [1,0,3,2]
__device__ inline float quad_xor2(float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0x4E, 0xF, 0xF, true)); }   // quad_perm [2,3,0,1]
template <int U>
__device__ inline float ip_dot16_quad(const float* qs, const float* __restrict__ x, uint32_t n16, uint32_t v, uint32_t L) {
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const bool odd = (v & 1) != 0, hi = (v & 2) != 0;
    for (uint32_t i = 0; i < n16; i += 16 * U) {
        float4 xv[U];
#pragma unroll
        for (int u = 0; u < U; u++) xv[u] = *(const float4*)(x + (i + 16 * u < n16 ? i + 16 * u : 0) + 4 * v);
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (i + 16 * u < n16) {                                       // (uniform per block: n16, i, u are wave-uniform)
                const float4 qv = *(const float4*)(qs + i + 16 * u + 4 * v);
                if (L == 16) {
                    acc[0] = ip_mul_add(acc[0], qv.x, xv[u].x); acc[1] = ip_mul_add(acc[1], qv.y, xv[u].y);
                    acc[2] = ip_mul_add(acc[2], qv.z, xv[u].z); acc[3] = ip_mul_add(acc[3], qv.w, xv[u].w);
                } else {
                    float p[4] = {ip_mul(qv.x, xv[u].x), ip_mul(qv.y, xv[u].y), ip_mul(qv.z, xv[u].z), ip_mul(qv.w, xv[u].w)};
                    if (L == 8) {
#pragma unroll
                        for (int c = 0; c < 4; c++) acc[c] = ip_add(ip_add(acc[c], p[c]), quad_xor2(p[c]));      // (meaningful on quad lanes 0 and 1)
                    } else {
#pragma unroll
                        for (int k = 0; k < 2; k++) {                     // stage A: swap bit 0 of (lane, component)
                            const float recv = quad_xor1(odd ? p[2 * k] : p[2 * k + 1]);
                            if (odd) p[2 * k] = recv; else p[2 * k + 1] = recv;
                        }
#pragma unroll
                        for (int k = 0; k < 2; k++) {                     // stage B: swap bit 1
                            const float recv = quad_xor2(hi ? p[k] : p[k + 2]);
                            if (hi) p[k] = recv; else p[k + 2] = recv;
                        }
                        acc[0] = ip_add(ip_add(ip_add(ip_add(acc[0], p[0]), p[1]), p[2]), p[3]);                  // quad lane c = accumulator lane c
                    }
                }
            }
        }
    }
    const int q0 = (int)(threadIdx.x & 60u);
    if (L == 4) {
        const float t0 = __shfl(acc[0], q0), t1 = __shfl(acc[0], q0 + 1), t2 = __shfl(acc[0], q0 + 2), t3 = __shfl(acc[0], q0 + 3);
        return ip_add(ip_add(ip_add(t0, t1), t2), t3);
    }
    float sum = 0.0f;
    const int nl = L == 16 ? 4 : 2;                                       // quad lanes holding accumulator lanes, four each
#pragma unroll
    for (int vv = 0; vv < 4; vv++)
#pragma unroll
        for (int c = 0; c < 4; c++) { const float t = __shfl(acc[c], q0 + vv); if (vv < nl) sum = ip_add(sum, t); }
    return sum;
}

// distances of one query to explicit rows (flat scan over filter ids, src/index.cpp:3345-3374; getDataByLabel + get_dist_func of
// compute_aux_scores, :8856-8880): 16 lanes per row, hnswlib's own summation order (ip_distance_group16) — the same bits the k-NN
// paths return. rows[i] == 0xFFFFFFFF -> NaN (label missing).
__global__ __launch_bounds__(256) void vec_row_distances_kernel(const float* __restrict__ X, const float* __restrict__ q, uint32_t dim,
                                                                 const uint32_t* __restrict__ rows, uint32_t n, float* __restrict__ out, uint32_t ip_lanes) {
    const uint32_t sub = threadIdx.x & 15;
    const uint32_t i = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
    const uint32_t ic = i < n ? i : n - 1;               // idle groups recompute the last row (keeps the wave's shuffles uniform)
    const uint32_t row = rows[ic];
    const float d = ip_distance_group16(q, X + (size_t)(row != 0xFFFFFFFFu ? row : 0) * dim, dim, sub, ip_lanes);
    if (i < n && sub == 0) out[i] = row == 0xFFFFFFFFu ? __uint_as_float(0x7FC00000u) : d;
}

// the FLAT branch's distance matrix (process_results_bruteforce, src/index.cpp:3345-3374, for a batch of queries sharing one filter):
// out[query][i] = exact distance of row rows[i], same 16-lane summation as above; grid (rows / 16, queries). A missing label stays NaN.
__global__ __launch_bounds__(256) void vec_flat_distances_kernel(const float* __restrict__ X, const float* __restrict__ Q, uint32_t dim, const uint32_t* __restrict__ rows,
                                                                  uint32_t n, float* __restrict__ out, size_t out_stride, uint32_t ip_lanes) {
    const uint32_t sub = threadIdx.x & 15;
    const uint32_t i = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
    const uint32_t ic = i < n ? i : n - 1;
    const uint32_t row = rows[ic];
    const float d = ip_distance_group16(Q + (size_t)blockIdx.y * dim, X + (size_t)(row != 0xFFFFFFFFu ? row : 0) * dim, dim, sub, ip_lanes);
    if (i < n && sub == 0) out[(size_t)blockIdx.y * out_stride + i] = row == 0xFFFFFFFFu ? __uint_as_float(0x7FC00000u) : d;
}

// the same for (query, row) pairs of a batch: item i = (queries[qidx[i]], rows[i])
__global__ __launch_bounds__(256) void vec_pair_distances_kernel(const float* __restrict__ X, const float* __restrict__ Q, uint32_t dim, const uint32_t* __restrict__ qidx,
                                                                  const uint32_t* __restrict__ rows, uint32_t n, float* __restrict__ out, uint32_t ip_lanes) {
    const uint32_t sub = threadIdx.x & 15;
    const uint32_t i = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
    const uint32_t ic = i < n ? i : n - 1;
    const uint32_t row = rows[ic];
    const float d = ip_distance_group16(Q + (size_t)qidx[ic] * dim, X + (size_t)(row != 0xFFFFFFFFu ? row : 0) * dim, dim, sub, ip_lanes);
    if (i < n && sub == 0) out[i] = row == 0xFFFFFFFFu ? __uint_as_float(0x7FC00000u) : d;
}

// exact re-scoring of the survivors: grid (n_q, splits); 16 lanes per (query, row) pair; key = ord(dist) << 32 | row
static const uint32_t VEC_RESCORE_LDS_DIM = 4096;     // queries up to this dim are staged in LDS; longer ones are read from L1/L2
__global__ __launch_bounds__(VEC_THREADS) void vec_rescore_kernel(const float* __restrict__ X, const float* __restrict__ Q, uint32_t dim,
                                                                   const uint32_t* __restrict__ surv_base, const uint32_t* __restrict__ surv_cnt, size_t stride,
                                                                   uint64_t* __restrict__ keys_base, uint32_t ip_lanes) {
    __shared__ __attribute__((aligned(16))) float qs_lds[VEC_RESCORE_LDS_DIM];
    const uint32_t t = threadIdx.x, q = blockIdx.x;
    const uint32_t n = surv_cnt[q];
    const bool quad = dim % 16 == 0;                     // four lanes per row, 16-byte loads (ip_part16_quad): 64 rows per trip, else 16
    const uint32_t rows_per_trip = quad ? VEC_THREADS / 4 : 16;
    const uint32_t per = gridDim.y * rows_per_trip;
    if (blockIdx.y * rows_per_trip >= n) return;
    const float* qs = Q + (size_t)q * dim;
    if (dim <= VEC_RESCORE_LDS_DIM) {
        for (uint32_t i = t; i < dim; i += VEC_THREADS) qs_lds[i] = qs[i];
        __syncthreads();
        qs = qs_lds;
    }
    const uint32_t* __restrict__ surv = surv_base + (size_t)q * stride;
    uint64_t* __restrict__ keys = keys_base + (size_t)q * stride;
    // every thread of the block runs the same number of trips (i0 is block-uniform); shuffles stay inside a quad / a 16-lane group
    if (quad) {
        for (uint32_t i0 = blockIdx.y * rows_per_trip; i0 < n; i0 += per) {
            const uint32_t i = i0 + (t >> 2);
            const uint32_t row = surv[i < n ? i : n - 1];   // idle quads recompute the last pair (keeps the wave's shuffles uniform)
            const float d = ip_add(1.0f, -ip_dot16_quad<24>(qs, X + (size_t)row * dim, dim, t & 3, ip_lanes));
            if (i < n && (t & 3) == 0) keys[i] = ((uint64_t)f32_ord(d) << 32) | row;
        }
        return;
    }
    const uint32_t sub = t & 15, grp = t >> 4;
    for (uint32_t i0 = blockIdx.y * 16; i0 < n; i0 += per) {
        const uint32_t i = i0 + grp;
        const uint32_t ic = i < n ? i : n - 1;
        const uint32_t row = surv[ic];
        const float d = ip_distance_group16(qs, X + (size_t)row * dim, dim, sub, ip_lanes);
        if (i < n && sub == 0) keys[i] = ((uint64_t)f32_ord(d) << 32) | row;
    }
}

// ------------------------------------------------------------------------------------------------
// Doc-range shards (tsgpu_group): a member's k nearest of every query as ONE u64 per hit, ord(dist) << 32 | label
// (labels are seq_ids: 32 bits; a larger label raises *bad) — ascending u64 order = (distance, label) ascending = the order
// flat_knn / searchKnnCloserFirst results merge in. Unused slots = all ones.
__global__ void vec_group_pack_kernel(const float* __restrict__ dist, const uint64_t* __restrict__ label, const uint32_t* __restrict__ cnt,
                                      uint32_t n_q, uint32_t k, uint64_t* __restrict__ dst, uint32_t* bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t q = i / k, j = i - q * k;
    if (q >= n_q) return;
    uint64_t key = VEC_KEY_INF;
    if (j < cnt[q]) {
        const uint64_t l = label[(size_t)q * k + j];
        if (l > 0xFFFFFFFFull) atomicAdd(bad, 1u);
        key = ((uint64_t)f32_ord(dist[(size_t)q * k + j]) << 32) | (l & 0xFFFFFFFFull);
    }
    dst[(size_t)q * k + j] = key;
}
// exact merge of G gathered blocks ([shard][query][k] keys, shard_stride words apart): one workgroup per query, bitonic sort of the
// G * k keys in LDS (CAP >= G * k, power of two), the k smallest come out closest first, ties -> smaller label
template <int CAP>
__global__ __launch_bounds__(VEC_THREADS) void vec_group_merge_kernel(const uint64_t* __restrict__ gathered, uint64_t shard_stride, uint32_t n_shards, uint32_t n_q, uint32_t k,
                                                                       float* __restrict__ dist_out, uint64_t* __restrict__ label_out, uint32_t* __restrict__ cnt_out) {
    __shared__ uint64_t keys[CAP];
    const uint32_t t = threadIdx.x, q = blockIdx.x;
    for (uint32_t i = t; i < (uint32_t)CAP; i += VEC_THREADS) {
        const uint32_t g = i / k, j = i - g * k;
        keys[i] = g < n_shards ? gathered[g * shard_stride + (size_t)q * k + j] : VEC_KEY_INF;
    }
    for (int size = 2; size <= CAP; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int p = t; p < CAP / 2; p += VEC_THREADS) {
                const int i = 2 * p - (p & (stride - 1)), j = i + stride;
                const bool up = (i & size) == 0;
                const uint64_t a = keys[i], b = keys[j];
                if ((a > b) == up) { keys[i] = b; keys[j] = a; }
            }
        }
    }
    __syncthreads();
    uint32_t n = 0;
    for (uint32_t i = t; i < k; i += VEC_THREADS) {
        const uint64_t key = keys[i];
        const bool live = key != VEC_KEY_INF;
        dist_out[(size_t)q * k + i] = live ? ord_f32((uint32_t)(key >> 32)) : 0.0f;
        label_out[(size_t)q * k + i] = live ? (key & 0xFFFFFFFFull) : 0;
        n += live ? 1u : 0u;
    }
    // count = number of live keys among the first k (they are sorted: live ones first)
    __shared__ uint32_t s_n;
    if (t == 0) s_n = 0;
    __syncthreads();
    if (n) atomicAdd(&s_n, n);
    __syncthreads();
    if (t == 0) cnt_out[q] = s_n;
}

// ================================================================================================
// HNSW graph search: hnswlib::HierarchicalNSW<float>::searchKnnCloserFirst(q, k, ef, filter) of
// the Typesense fork (call site src/index.cpp:3376-3445; VectorFilterFunctor include/index.h:325-354) on a MIRROR of the
// server's graph — hnswlib's level-0 link lists (count + up to 2M ids per node, coalesced 4*(1+2M)-byte records) and upper-level
// lists stream from HBM; rows = hnswlib internal ids = insertion order. hnswlib itself is not under /root/reference (PARITY
// UNPINNED, oracle/hnsw_graph.h): the traversal below restates the published algorithm — greedy descent through the upper
// layers, then the ef-bounded best-first search of layer 0 with two binary heaps — and is checked against the oracle's
// restatement on the same graph. One wavefront per query: the 64 lanes fetch a node's neighbour list at once, test/mark the
// visited tags, and compute the distances of the unvisited neighbours four at a time (16-lane groups = hnswlib's own
// InnerProductSpace summation order, so every compare sees the bits the CPU would see); lane 0 then replays hnswlib's
// sequential heap logic over them (std::priority_queue = libstdc++ push_heap / pop_heap, restated so that ties fall the same way).
struct VecHnswArgs {
    const float* X; const float* Q; uint32_t dim, n_rows, n_q;
    const uint32_t* q_rows;                             // nullable: query q is ROW q_rows[q] of X (the bulk build searches for the rows it inserts) instead of Q[q]
    const uint32_t* q_sel;                              // nullable: this launch serves the queries q_sel[0 .. n_q) of the batch (the ones whose heaps outgrew a smaller tier)
    uint32_t base_level;                                // BUILD instantiations only: the beam runs on this layer's lists (the descent stops above it); the search proper is layer 0
    const uint32_t* link0; uint32_t s0;                 // [n][s0], s0 = 1 + 2M
    const uint64_t* upper_ptr; const uint32_t* upper_links; uint32_t su;      // su = 1 + M
    int32_t maxlevel; uint32_t enterpoint;
    const uint8_t* row_ok;     // nullable: 0 = deleted or filtered out (isMarkedDeleted / !isIdAllowed)
    uint32_t strict;           // a filter functor is present or the index has deletions (hnswalg.h searchBaseLayerST break rule)
    uint32_t k, ef;
    uint32_t ip_lanes;                                  // summation order of the distance (4 / 8 / 16: the SIMD level hnswlib was compiled for)
    uint16_t* visited; uint32_t epoch_base;             // tag mode: [slots][n_rows] 16-bit tags (hnswlib's VisitedList is 16-bit, too); slot = blockIdx.x
    // hash mode (default): the ids a query visits (a few thousand) live in ITS open-addressing set of vhash_slots words (a power of two,
    // 64 x the tier's heap capacity), cleared per query — memory per concurrent query no longer depends on the row count (16-bit tags:
    // 2 B x rows, 41 GB for 2 048 queries at 10M rows; sets: 32 KB - 256 KB each). A set that fills beyond half reports the query like a
    // candidate-heap overflow (re-run on the largest tier, then exactly by the caller).
    uint32_t* vhash; uint32_t vhash_slots;
    uint32_t* overflow_cnt;                             // [0] queries whose candidate heap outgrew CANDCAP; [1..2] u64 expansions, [3..4] u64 distances (batch totals)
    const uint64_t* labels;                             // nullable: internal ids come back
    float* dist_out; uint64_t* label_out; uint32_t* n_out;   // [n_q][k]; n_out = 0xFFFFFFFF: candidate heap overflow (caller re-runs exactly)
};
#ifndef TSGPU_HNSW_ROWS
#define TSGPU_HNSW_ROWS 4
#endif
#ifndef TSGPU_HNSW_CHUNK
#define TSGPU_HNSW_CHUNK 24
#endif
static const int VEC_HNSW_CHUNK = TSGPU_HNSW_CHUNK;   // distance phase: 16-byte row pieces per lane requested before the first is consumed
static const uint32_t VEC_HNSW_MAX_EF = 1024;
static const uint32_t VEC_HNSW_CAND_CAP = 4096;
static const uint32_t VEC_HNSW_QDIM = 1024;             // queries up to this dim are staged in LDS

struct HnswEntry { float d; uint32_t id; };                      // one 8-byte LDS word per heap entry: a sift step reads both children with one ds_read2_b64
struct HnswHeap {          // max-heap on .d (CompareByFirst: a.first < b.first), libstdc++ algorithms
    HnswEntry* e; uint32_t n;
    __device__ inline float top_d() const { return e[0].d; }
    __device__ inline uint32_t top_id() const { return e[0].id; }
    __device__ inline void push(float vd, uint32_t vid) {       // std::push_heap after push_back
        uint32_t hole = n++;
        while (hole > 0) {
            const uint32_t parent = (hole - 1) / 2;
            const HnswEntry p = e[parent];
            if (!(p.d < vd)) break;
            e[hole] = p;
            hole = parent;
        }
        e[hole] = HnswEntry{vd, vid};
    }
    __device__ inline void pop() {                               // std::pop_heap + pop_back
        const uint32_t len = --n;                                // heap of len elements after removing the last
        if (len == 0) return;
        const HnswEntry v = e[len];                              // value = *(last - 1); its slot receives the old top (dropped)
        uint32_t hole = 0, second = 0;
        while (second < (len - 1) / 2) {
            second = 2 * (second + 1);
            const HnswEntry r = e[second], l = e[second - 1];    // adjacent: one LDS round trip per level
            HnswEntry c = r;
            if (r.d < l.d) { second--; c = l; }
            e[hole] = c;
            hole = second;
        }
        if ((len & 1) == 0 && second == (len - 2) / 2) {
            second = 2 * (second + 1);
            e[hole] = e[second - 1];
            hole = second - 1;
        }
        while (hole > 0) {                                       // __push_heap(first, hole, 0, value)
            const uint32_t parent = (hole - 1) / 2;
            const HnswEntry p = e[parent];
            if (!(p.d < v.d)) break;
            e[hole] = p;
            hole = parent;
        }
        e[hole] = v;
    }
};

// TOPCAP >= max(ef, k) + 1, CANDCAP = candidate heap capacity: LDS tiers chosen by the host from ef, so that a CU holds as many
// concurrent queries as its wave slots allow at the usual ef (13 KB per query at ef <= 128 instead of 45 KB).
template <uint32_t TOPCAP, uint32_t CANDCAP, bool BUILD = false>
__global__ __launch_bounds__(64) void vec_hnsw_search_kernel(VecHnswArgs a) {
    const uint32_t base = BUILD ? a.base_level : 0u;
    __shared__ HnswEntry top_e[TOPCAP];
    __shared__ HnswEntry cand_e[CANDCAP];
    __shared__ __attribute__((aligned(16))) float qs_lds[VEC_HNSW_QDIM];
    __shared__ uint32_t nb_id[64];
    __shared__ float nb_d[64];
    __shared__ uint8_t nb_ok[64];
    __shared__ uint32_t s_cur, s_state, s_top_n;
    __shared__ float s_lb;
    const uint32_t lane = threadIdx.x, sub = lane & 15, grp = lane >> 4;
    volatile uint16_t* vis = a.vhash ? nullptr : a.visited + (size_t)blockIdx.x * a.n_rows;      // volatile: tags written by this wave are re-read later (no stale L1 lines)
    uint32_t* vset = a.vhash ? a.vhash + (size_t)blockIdx.x * a.vhash_slots : nullptr;
    const uint32_t vmask = a.vhash_slots - 1;
    // visited test-and-set: true = first visit. Hash mode: one CAS per probe at L2 (a wave's concurrent inserts are distinct ids; two of them
    // racing for one empty slot are ordered by the CAS)
    auto visit = [&](uint32_t id, uint16_t epoch, uint32_t& n_vis) -> bool {
        if (!vset) { const bool fresh = vis[id] != epoch; if (fresh) vis[id] = epoch; return fresh; }
        uint32_t h = (id * 2654435761u) & vmask;
        for (;;) {
            const uint32_t cur = atomicCAS(&vset[h], 0u, id + 1u);
            if (cur == 0u) { n_vis++; return true; }
            if (cur == id + 1u) return false;
            h = (h + 1) & vmask;
        }
    };
    uint32_t iter = 0;
    for (uint32_t qn = blockIdx.x; qn < a.n_q; qn += gridDim.x, iter++) {
        const uint32_t q = a.q_sel ? a.q_sel[qn] : qn;
        const uint16_t epoch = (uint16_t)(a.epoch_base + iter);
        uint32_t n_vis = 0;                                      // (per lane; summed over the wave when checked)
        if (vset) {
            for (uint32_t i = lane; i < a.vhash_slots; i += 64) vset[i] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");        // the clears are in L2 before the first CAS (atomics execute there)
        }
        const float* qs = a.q_rows ? a.X + (size_t)a.q_rows[q] * a.dim : a.Q + (size_t)q * a.dim;
        __syncthreads();
        if (a.dim <= VEC_HNSW_QDIM) {
            for (uint32_t i = lane; i < a.dim; i += 64) qs_lds[i] = qs[i];
            qs = qs_lds;
        }
        __syncthreads();
        // distances of the nb_n ids staged in nb_id[] -> nb_d[]: sixteen per round (ip_part16_quad) when the dimension is a multiple of
        // 16, else four (16-lane groups)
        auto distances = [&](uint32_t nb_n) {
            if (a.dim % 16 == 0) {
                for (uint32_t i0 = 0; i0 < nb_n; i0 += 16) {           // 16 rows per round, four lanes each
                    const uint32_t i = i0 + (lane >> 2);
                    const float dot = ip_dot16_quad<VEC_HNSW_CHUNK>(qs, a.X + (size_t)nb_id[i < nb_n ? i : nb_n - 1] * a.dim, a.dim, lane & 3, a.ip_lanes);
                    if (i < nb_n && (lane & 3) == 0) nb_d[i] = ip_add(1.0f, -dot);
                }
            } else {
                for (uint32_t i0 = 0; i0 < nb_n; i0 += 4) {
                    const uint32_t i = i0 + grp;
                    const uint32_t row = nb_id[i < nb_n ? i : nb_n - 1];
                    const float d = ip_distance_group16(qs, a.X + (size_t)row * a.dim, a.dim, sub, a.ip_lanes);
                    if (i < nb_n && sub == 0) nb_d[i] = d;
                }
            }
            __syncthreads();
        };
        // ---- upper layers: greedy descent (hnswalg.h searchKnn) ----
        if (lane == 0) { nb_id[0] = a.enterpoint; }
        __syncthreads();
        distances(1);
        uint32_t cur = a.enterpoint;
        float curdist = nb_d[0];
        __syncthreads();
        for (int level = a.maxlevel; level > (int)base; level--) {
            bool changed = true;
            while (changed) {
                changed = false;
                const uint32_t* __restrict__ lst = a.upper_links + (a.upper_ptr[cur] + (uint32_t)(level - 1)) * a.su;
                const uint32_t cnt = lst[0];
                const uint32_t cw = lst[lane < a.su - 1 ? 1 + lane : 0];
                if (lane < cnt) nb_id[lane] = cw;
                __syncthreads();
                distances(cnt);
                for (uint32_t i = 0; i < cnt; i++) {                 // in list order, like the reference's loop
                    const float d = nb_d[i];
                    if (d < curdist) { curdist = d; cur = nb_id[i]; changed = true; }
                }
                __syncthreads();
            }
        }
        // ---- layer 0: searchBaseLayerST(cur, q, max(ef, k), filter) ----
        const uint32_t ef = a.ef > a.k ? a.ef : a.k;
        HnswHeap top{top_e, 0}, cand{cand_e, 0};
        float lowerBound;
        bool overflow = false;
        uint32_t n_exp = 0, n_dist = 0;
#ifdef TSGPU_HNSW_PROF          // tools/ builds: wall-clock ticks (100 MHz) per phase of the layer-0 loop, batch totals behind the statistics
        unsigned long long pt[4] = {0, 0, 0, 0}, pl = wall_clock64();
#define HNSW_PROF(i) { const unsigned long long _n = wall_clock64(); pt[i] += _n - pl; pl = _n; }
#else
#define HNSW_PROF(i)
#endif
        if (lane == 0) {
            const bool ok = !a.row_ok || a.row_ok[cur] != 0;
            if (ok) { lowerBound = curdist; top.push(curdist, cur); cand.push(-curdist, cur); }
            else { lowerBound = 3.402823466e+38f; cand.push(-lowerBound, cur); }
            (void)visit(cur, epoch, n_vis);
        }
        for (;;) {
            if (lane == 0) {
                uint32_t st = 0;                                  // 0 = expand s_cur, 1 = finished
                if (cand.n == 0) st = 1;
                else {
                    const float cd = cand.top_d();
                    if ((-cd) > lowerBound && (top.n == ef || !a.strict)) st = 1;
                    else { s_cur = cand.top_id(); cand.pop(); }
                }
                s_state = st; s_top_n = top.n; s_lb = lowerBound;
            }
            __syncthreads();
            if (s_state) break;
            HNSW_PROF(0)
            const uint32_t node = s_cur;
            const uint32_t* __restrict__ lst = (BUILD && base) ? a.upper_links + (a.upper_ptr[node] + (base - 1)) * a.su : a.link0 + (size_t)node * a.s0;
            const uint32_t lw = (BUILD && base) ? a.su : a.s0;
            const uint32_t cnt = lst[0];
            uint32_t c = 0;
            bool fresh = false;
            // (count and ids requested together: a record has 1 + 2M words whatever its count; ids past the count are ignored)
            const uint32_t cw = lst[lane < lw - 1 ? 1 + lane : 0];
            // the set stays at most half full (n_dist counts every first visit but the entry point's; a list adds at most cnt): a query that
            // would outgrow it stops here and is reported like a candidate-heap overflow
            const bool room = !vset || n_dist + 1 + cnt <= a.vhash_slots / 2;
            if (!room && lane == 0) { overflow = true; cand.n = 0; }
            if (room && lane < cnt) { c = cw; fresh = visit(c, epoch, n_vis); }
            const unsigned long long m = __ballot(fresh ? 1 : 0);
            const uint32_t nf = (uint32_t)__popcll(m);
            if (fresh) {
                const uint32_t slot = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));      // unvisited neighbours, list order kept
                nb_id[slot] = c;
                nb_ok[slot] = (!a.row_ok || a.row_ok[c] != 0) ? 1 : 0;
            }
            __syncthreads();
            HNSW_PROF(1)
            if (nf) distances(nf);
            HNSW_PROF(2)
            n_exp++; n_dist += nf;
            // Which of them can enter the heaps at all? The reference tests `top.size() < ef || lowerBound > d` one neighbour after the
            // other; once the result heap is full its bound only falls while this list is processed, so a neighbour that fails the test
            // against the bound as it stands NOW fails it in sequence, too: all lanes test in parallel, lane 0 replays the sequential
            // heap logic over the few that pass (a heap that is still filling takes every neighbour: no filter then).
            const bool full = s_top_n == ef;
            const bool pass = lane < nf && (!full || s_lb > nb_d[lane]);
            unsigned long long pm = __ballot(pass ? 1 : 0);
            if (lane == 0) {
                while (pm) {
                    const uint32_t i = (uint32_t)__builtin_ctzll(pm);
                    pm &= pm - 1;
                    const float d = nb_d[i];
                    const uint32_t cid = nb_id[i];
                    if (top.n < ef || lowerBound > d) {
                        if (cand.n >= CANDCAP) { overflow = true; break; }
                        cand.push(-d, cid);
                        if (nb_ok[i]) top.push(d, cid);
                        if (top.n > ef) top.pop();
                        if (top.n) lowerBound = top.top_d();
                    }
                }
                if (overflow) { cand.n = 0; }
            }
            __syncthreads();
            HNSW_PROF(3)
        }
#ifdef TSGPU_HNSW_PROF
        if (lane == 0) for (int i = 0; i < 4; i++) atomicAdd((unsigned long long*)(a.overflow_cnt + 6 + 2 * i), pt[i]);
#endif
        // ---- result: keep the k closest, closest first (searchKnn + searchKnnCloserFirst) ----
        if (lane == 0) {
            atomicAdd((unsigned long long*)(a.overflow_cnt + 2), (unsigned long long)n_exp);
            atomicAdd((unsigned long long*)(a.overflow_cnt + 4), (unsigned long long)n_dist);
            if (overflow) { a.n_out[q] = 0xFFFFFFFFu; atomicAdd(a.overflow_cnt, 1u); }
            else {
                while (top.n > a.k) top.pop();
                uint32_t sz = top.n;
                a.n_out[q] = sz;
                while (top.n) { --sz; a.dist_out[(size_t)q * a.k + sz] = top.top_d(); a.label_out[(size_t)q * a.k + sz] = a.labels ? a.labels[top.top_id()] : (uint64_t)top.top_id(); top.pop(); }
            }
        }
        __syncthreads();
    }
}

}  // namespace tsgpu
