// vec_kernels.hip.h — gfx950 kernels of the vector hot path (seam B2, include/tsgpu.h).
//
// Replaces, for a whole batch of queries, the reference's per-query distance loop (process_results_bruteforce,
// src/index.cpp:3345-3374: dist = space->get_dist_func()(q, x, &dim) = 1 - <q,x>, hnswlib InnerProductSpace) and
// the top-k it feeds (searchKnnCloserFirst result + Topster, src/index.cpp:3384-3389, 3682-3725) with an EXACT
// scan: S = X · Qᵀ on the fp32-input matrix cores and a fused running top-k, so the N x B score matrix never
// exists in memory.
//
// Mapping to the machine:
//   * v_mfma_f32_32x32x2_f32 (exact fp32, bit-identical to a k-ordered fmaf chain): A = 32 base rows, B = 32
//     queries, so one lane's 16 accumulator registers all belong to ONE query (column = lane & 31) — the
//     per-query threshold lives in a register and the top-k filter is 16 compares per lane, no cross-lane traffic;
//   * workgroup = 4 waves = 128 base rows x QT queries (QT = 64, or 32 when k > 128); the K dimension streams
//     through LDS in 64-float chunks, next chunk prefetched into registers while the current one feeds the MFMAs;
//   * each workgroup walks a contiguous slab of base rows for one query tile and keeps, per query, an LDS list
//     of the KL best (distance, row) keys; candidates reach the list through per-query mini queues and a
//     re-scan loop, so nothing is ever dropped (exact);
//   * blockIdx -> (slab, query tile) is XCD-aware: the query tiles of one slab run on the same XCD back to
//     back, so the slab is fetched from HBM once and re-served from that XCD's L2;
//   * vec_merge_kernel folds the per-slab lists of a query into the final k (ties: smaller row first).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tsgpu {

static const int VEC_THREADS = 256;
static const int VEC_ROWS = 128;        // base rows per tile (4 waves x 32)
static const int VEC_KC = 64;           // K chunk staged in LDS
static const int VEC_LDW = VEC_KC + 1;  // padded row stride (words): conflict-free column reads
static const int VEC_QCAP = 8;          // per-query mini queue entries per round
static const uint64_t VEC_KEY_INF = 0xFFFFFFFFFFFFFFFFull;

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

struct VecKnnArgs {
    const float* X;            // [n_rows][dim] row-major
    const uint8_t* row_ok;     // nullable: 0 = skip row (deleted / filtered out)
    const float* Q;            // [n_q][dim]
    uint32_t n_rows, dim, n_q;
    uint32_t rows_per_slab;    // multiple of VEC_ROWS
    uint32_t n_slabs;          // multiple of 8 (XCD-aware mapping)
    uint32_t n_qtiles;
    uint64_t* part_keys;       // [n_slabs][n_qtiles*QT][KL]
    uint32_t* part_cnt;        // [n_slabs][n_qtiles*QT]
};

template <int QT, int KL>
struct VecSmem {
    float xs[VEC_ROWS * VEC_LDW];
    float qs[QT * VEC_LDW];
    uint64_t list[KL * QT];             // [slot][query] : conflict-free for one-lane-per-query scans
    uint64_t tau[QT];                   // current worst key of a full list (INF while filling)
    uint32_t cnt[QT];
    uint64_t mq[VEC_QCAP * QT];         // mini queues [slot][query]
    uint32_t mq_cnt[QT];
    uint32_t again;
};

// QT queries per workgroup (NB = QT/32 MFMA column blocks per wave), KL list slots per query (>= k)
template <int QT, int KL>
__global__ __launch_bounds__(VEC_THREADS) void vec_knn_kernel(VecKnnArgs a) {
    constexpr int NB = QT / 32;
    __shared__ VecSmem<QT, KL> sm;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // XCD-aware: blocks b, b+8, b+16.. share an XCD; give them the query tiles of the same slab
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7, j = b >> 3;
    const uint32_t qtile = j % a.n_qtiles;
    const uint32_t slab = (j / a.n_qtiles) * 8 + xcd;
    const uint32_t q0 = qtile * QT;
    const uint32_t row_begin = slab * a.rows_per_slab;
    uint32_t row_end = row_begin + a.rows_per_slab;
    if (row_end > a.n_rows) row_end = a.n_rows;

    for (uint32_t i = t; i < (uint32_t)QT; i += VEC_THREADS) { sm.tau[i] = VEC_KEY_INF; sm.cnt[i] = 0; sm.mq_cnt[i] = 0; }
    if (t == 0) sm.again = 0;
    __syncthreads();

    const uint32_t n_chunks = (a.dim + VEC_KC - 1) / VEC_KC;
    // staging assignment: X tile = 128 rows x 64 floats = 2048 float4 -> 8 per thread; Q tile = QT x 64 floats
    constexpr int XV = VEC_ROWS * VEC_KC / 4 / VEC_THREADS;     // 8
    constexpr int QV = QT * VEC_KC / 4 / VEC_THREADS;           // 4 (QT=64) or 2 (QT=32)

    for (uint32_t r0 = row_begin; r0 < row_end; r0 += VEC_ROWS) {
        vec_f32x16 acc[NB];
#pragma unroll
        for (int n = 0; n < NB; n++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[n][e] = 0.0f;

        float4 xr[XV], qr[QV];
        auto load_chunk = [&](uint32_t c) {
            const uint32_t k0 = c * VEC_KC;
#pragma unroll
            for (int v = 0; v < XV; v++) {
                const uint32_t idx = t + v * VEC_THREADS;           // float4 index in the tile
                const uint32_t row = idx / (VEC_KC / 4), c4 = idx % (VEC_KC / 4);
                const uint32_t gr = r0 + row, gk = k0 + c4 * 4;
                float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gr < row_end) {
                    const float* p = a.X + (size_t)gr * a.dim + gk;
                    if (gk + 3 < a.dim && ((a.dim & 3) == 0)) val = *(const float4*)p;
                    else {
                        if (gk < a.dim) val.x = p[0];
                        if (gk + 1 < a.dim) val.y = p[1];
                        if (gk + 2 < a.dim) val.z = p[2];
                        if (gk + 3 < a.dim) val.w = p[3];
                    }
                }
                xr[v] = val;
            }
#pragma unroll
            for (int v = 0; v < QV; v++) {
                const uint32_t idx = t + v * VEC_THREADS;
                const uint32_t qi = idx / (VEC_KC / 4), c4 = idx % (VEC_KC / 4);
                const uint32_t gq = q0 + qi, gk = k0 + c4 * 4;
                float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gq < a.n_q) {
                    const float* p = a.Q + (size_t)gq * a.dim + gk;
                    if (gk + 3 < a.dim && ((a.dim & 3) == 0)) val = *(const float4*)p;
                    else {
                        if (gk < a.dim) val.x = p[0];
                        if (gk + 1 < a.dim) val.y = p[1];
                        if (gk + 2 < a.dim) val.z = p[2];
                        if (gk + 3 < a.dim) val.w = p[3];
                    }
                }
                qr[v] = val;
            }
        };
        auto store_chunk = [&]() {
#pragma unroll
            for (int v = 0; v < XV; v++) {
                const uint32_t idx = t + v * VEC_THREADS;
                const uint32_t row = idx / (VEC_KC / 4), c4 = idx % (VEC_KC / 4);
                float* d = &sm.xs[row * VEC_LDW + c4 * 4];
                d[0] = xr[v].x; d[1] = xr[v].y; d[2] = xr[v].z; d[3] = xr[v].w;
            }
#pragma unroll
            for (int v = 0; v < QV; v++) {
                const uint32_t idx = t + v * VEC_THREADS;
                const uint32_t qi = idx / (VEC_KC / 4), c4 = idx % (VEC_KC / 4);
                float* d = &sm.qs[qi * VEC_LDW + c4 * 4];
                d[0] = qr[v].x; d[1] = qr[v].y; d[2] = qr[v].z; d[3] = qr[v].w;
            }
        };

        load_chunk(0);
        for (uint32_t c = 0; c < n_chunks; c++) {
            __syncthreads();                 // previous chunk fully consumed
            store_chunk();
            __syncthreads();
            if (c + 1 < n_chunks) load_chunk(c + 1);   // in flight while the MFMAs run
            const float* xa = &sm.xs[(wave * 32 + (lane & 31)) * VEC_LDW + (lane >> 5)];
            const float* qb = &sm.qs[(lane & 31) * VEC_LDW + (lane >> 5)];
#pragma unroll 8
            for (int kk = 0; kk < VEC_KC; kk += 2) {
                const float av = xa[kk];
#pragma unroll
                for (int n = 0; n < NB; n++) {
                    const float bv = qb[n * 32 * VEC_LDW + kk];
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[n], 0, 0, 0);
                }
            }
        }

        // ---- fused top-k: this lane owns query column (lane & 31) of each of its NB blocks ----
        // distance = 1 - dot ; key = (ord(distance) << 32) | row   (smaller = closer; ties: smaller row)
        uint64_t keys[NB][16];
#pragma unroll
        for (int n = 0; n < NB; n++) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const uint32_t row = r0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const uint32_t gq = q0 + n * 32 + (lane & 31);
                bool ok = row < row_end && gq < a.n_q;
                if (ok && a.row_ok) ok = a.row_ok[row] != 0;
                const float dist = 1.0f - acc[n][e];
                keys[n][e] = ok ? (((uint64_t)f32_ord(dist) << 32) | row) : VEC_KEY_INF;
            }
        }
        for (;;) {
            // push: every still-unqueued key better than the query's threshold
            bool left = false;
#pragma unroll
            for (int n = 0; n < NB; n++) {
                const uint32_t ql = n * 32 + (lane & 31);
                const uint64_t tau = sm.tau[ql];
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const uint64_t kv = keys[n][e];
                    if (kv < tau) {
                        const uint32_t slot = atomicAdd(&sm.mq_cnt[ql], 1u);
                        if (slot < (uint32_t)VEC_QCAP) { sm.mq[slot * QT + ql] = kv; keys[n][e] = VEC_KEY_INF; }
                        else left = true;
                    }
                }
            }
            if (left) sm.again = 1;
            __syncthreads();
            // drain: thread q owns query q's list
            if (t < (uint32_t)QT) {
                uint32_t nq = sm.mq_cnt[t];
                if (nq > (uint32_t)VEC_QCAP) nq = VEC_QCAP;
                uint32_t cnt = sm.cnt[t];
                uint64_t tau = sm.tau[t];
                for (uint32_t i = 0; i < nq; i++) {
                    const uint64_t kv = sm.mq[i * QT + t];
                    if (cnt < (uint32_t)KL) {
                        sm.list[cnt * QT + t] = kv;
                        cnt++;
                        if (cnt == (uint32_t)KL) {          // list just filled: threshold = its worst key
                            uint64_t mx = 0;
                            for (int s = 0; s < KL; s++) { const uint64_t v = sm.list[s * QT + t]; if (v > mx) mx = v; }
                            tau = mx;
                        }
                    } else if (kv < tau) {                  // replace the worst, recompute the threshold
                        uint64_t mx = 0;
                        int mpos = 0;
                        for (int s = 0; s < KL; s++) { const uint64_t v = sm.list[s * QT + t]; if (v == tau) mpos = s; }
                        sm.list[mpos * QT + t] = kv;
                        for (int s = 0; s < KL; s++) { const uint64_t v = sm.list[s * QT + t]; if (v > mx) mx = v; }
                        tau = mx;
                    }
                }
                sm.cnt[t] = cnt;
                sm.tau[t] = tau;
                sm.mq_cnt[t] = 0;
            }
            __syncthreads();
            const uint32_t again = sm.again;
            __syncthreads();
            if (t == 0) sm.again = 0;
            __syncthreads();                 // the reset must not overtake the next round's "left" flag
            if (!again) break;
        }
    }
    __syncthreads();
    // ---- slab result: the (unsorted) lists ----
    const size_t qstride = (size_t)a.n_qtiles * QT;
    for (uint32_t i = t; i < (uint32_t)(KL * QT); i += VEC_THREADS) {
        const uint32_t s = i / QT, ql = i % QT;
        if (s < sm.cnt[ql]) a.part_keys[((size_t)slab * qstride + q0 + ql) * KL + s] = sm.list[s * QT + ql];
    }
    for (uint32_t i = t; i < (uint32_t)QT; i += VEC_THREADS) a.part_cnt[(size_t)slab * qstride + q0 + i] = sm.cnt[i];
}

// ------------------------------------------------------------------------------------------------
// one workgroup per query: k smallest keys over all slabs, ascending
template <int KL, int BUF>
__global__ __launch_bounds__(VEC_THREADS) void vec_merge_kernel(const uint64_t* __restrict__ part_keys, const uint32_t* __restrict__ part_cnt,
                                                                 uint32_t n_slabs, uint32_t q_stride, uint32_t k,
                                                                 const uint64_t* __restrict__ labels, float* __restrict__ dist_out,
                                                                 uint64_t* __restrict__ label_out, uint32_t* __restrict__ n_out) {
    __shared__ uint64_t buf[BUF];
    __shared__ uint32_t s_cnt;
    const uint32_t t = threadIdx.x, q = blockIdx.x;
    if (t == 0) s_cnt = 0;
    __syncthreads();
    auto sort_buf = [&]() {   // ascending bitonic sort of BUF keys (padding = INF)
        for (uint32_t i = t; i < (uint32_t)BUF; i += VEC_THREADS) if (i >= s_cnt) buf[i] = VEC_KEY_INF;
        for (int size = 2; size <= BUF; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                __syncthreads();
                for (int p = t; p < BUF / 2; p += VEC_THREADS) {
                    const int i = 2 * p - (p & (stride - 1));
                    const int jx = i + stride;
                    const bool asc = ((i & size) == 0);
                    const uint64_t x = buf[i], y = buf[jx];
                    if (asc ? (x > y) : (x < y)) { buf[i] = y; buf[jx] = x; }
                }
            }
        }
        __syncthreads();
    };
    for (uint32_t s = 0; s < n_slabs; s++) {
        const uint32_t n = part_cnt[(size_t)s * q_stride + q];
        if (s_cnt + n > (uint32_t)BUF) {
            sort_buf();
            if (t == 0) s_cnt = s_cnt < k ? s_cnt : k;
            __syncthreads();
        }
        const uint32_t base = s_cnt;
        for (uint32_t i = t; i < n; i += VEC_THREADS) buf[base + i] = part_keys[((size_t)s * q_stride + q) * KL + i];
        __syncthreads();
        if (t == 0) s_cnt = base + n;
        __syncthreads();
    }
    sort_buf();
    const uint32_t n = s_cnt < k ? s_cnt : k;
    for (uint32_t i = t; i < n; i += VEC_THREADS) {
        const uint64_t kv = buf[i];
        const uint32_t row = (uint32_t)(kv & 0xFFFFFFFFull);
        dist_out[(size_t)q * k + i] = ord_f32((uint32_t)(kv >> 32));
        label_out[(size_t)q * k + i] = labels[row];
    }
    if (t == 0) n_out[q] = n;
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
