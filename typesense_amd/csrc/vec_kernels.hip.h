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
                    for (int e = 0; e < 16; e++) acc[rb][cb][e] = 0.0f;
        }
        if (s + 1 < total_steps) load_step(s + 1);           // in flight while the MFMAs run
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

// distances of one query to explicit rows: one wave per row, lane-strided partial sums (flat scan over filter
// ids, src/index.cpp:3345-3374). rows[i] == 0xFFFFFFFF -> NaN (label missing).
__global__ __launch_bounds__(256) void vec_row_distances_kernel(const float* __restrict__ X, const float* __restrict__ q, uint32_t dim,
                                                                 const uint32_t* __restrict__ rows, uint32_t n, float* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const bool live = i < n;
    const uint32_t row = live ? rows[i] : 0xFFFFFFFFu;
    float s = 0.0f;
    if (row != 0xFFFFFFFFu) {
        const float* x = X + (size_t)row * dim;
        for (uint32_t k = lane; k < dim; k += 64) s = fmaf(q[k], x[k], s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (live && lane == 0) out[i] = row == 0xFFFFFFFFu ? __uint_as_float(0x7FC00000u) : 1.0f - s;
}

}  // namespace tsgpu
