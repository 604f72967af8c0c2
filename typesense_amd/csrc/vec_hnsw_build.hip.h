// vec_hnsw_build.hip.h — BULK construction of the HNSW graph on the device (tsgpu_vec_hnsw_build; included by tsgpu_vec.hip after vec_kernels.hip.h).
//
// Why: the reference fills its hnswlib index one addPoint at a time from the indexing threads (Index::index_field_in_memory ->
// vecdex->addPoint(vec, seq_id, true), /root/reference/src/index.cpp:1002-1075) — 8 K rows/s on 16 host threads at 768 dimensions
// (csrc/tsgpu_hnsw_build.h restates that insertion), i.e. twenty minutes and more for BASELINE config 3's 10M rows. A collection that is LOADED
// (restart, import) has all its rows before the first query: the graph can be built in batches instead, with the device doing the distance work.
//   1. levels are drawn for every row exactly as hnswlib draws them (default_random_engine(seed), mult = 1 / ln M, label order);
//   2. the SEED SET — every row with level >= 2 (one in M^2) and the first rows up to a minimum — is inserted on the host by the sequential algorithm
//      (tsgpu_hnsw_build.h, compact ids mapped back): the layers above 1 and the entry point are final after this step (should no seed row have an
//      upper level at all, the level-1 rows join the seed set: a device row never rises above the entry point);
//   3. the remaining rows (level 0 or 1), in id order, in batches of at most a sixteenth of what is already linked: every row of a batch searches the
//      graph AS IT STOOD BEFORE THE BATCH (vec_hnsw_search_kernel: greedy descent + the ef_construction beam — searchBaseLayer and searchBaseLayerST
//      with a filter coincide) — on layer 1 if it has that level, and on layer 0 —, keeps <= M neighbours per layer by hnswlib's distance heuristic
//      (getNeighborsByHeuristic2) — kernel 1 below — and files a reverse-link request with each of them; when the whole batch has chosen, every
//      node that received requests takes them all at once — appended while its list has room, else the heuristic over (old list + requests),
//      closest first — kernel 4. Layer 1's links of the batch, then layer 0's: both from beams taken before either was applied.
// Rows of one batch do not see each other (that is what makes them independent); everything else is mutuallyConnectNewElement's rule set. The
// distances are the ones every other path computes (hnswlib InnerProductSpace order, option vec_ip_lanes), ties are ordered (distance, id), so the
// graph is a deterministic function of (rows, M, ef_construction, seed, seed_min, max_batch): oracle/hnsw_graph.h restates it (bulk_build) and the
// tests compare link for link. hnswlib itself is not under /root/reference: PARITY UNPINNED as for the search.
#pragma once

namespace tsgpu {

static const uint32_t HNSW_BUILD_TCAP = 256;      // candidates one reverse-link re-selection looks at: the node's list + the closest requests of the batch

struct HnswBuildArgs {
    const float* X; uint32_t dim, ip_lanes;
    uint32_t* link0; uint32_t s0, M;              // [n][s0], s0 = 1 + 2M
    const uint64_t* upper_ptr; uint32_t* upper_links; uint32_t su;      // the upper lists (su = 1 + M)
    uint32_t layer;                               // the layer this round links on: 0 = link0 (lists of 2M), l >= 1 = the node's l-th upper list (lists of M)
    // the batch: rows[q] = the new node, its beam (closest first): cand_id[q * k + i] (u64: the search kernel's label type), cand_d, n_cand[q] (0xFFFFFFFF = the search overflowed)
    const uint32_t* rows; uint32_t n_batch, k;
    const uint64_t* cand_id; const float* cand_d; const uint32_t* n_cand;
    // reverse-link requests: req_s[q * M + j] = the neighbour asked (0xFFFFFFFF = none), req_d = its distance to rows[q]
    uint32_t* req_s; float* req_d;
    uint32_t* cnt; uint32_t* fill; uint32_t* start;        // [n]: requests per node (zero between batches), scatter cursor, first slot in seg_*
    uint32_t* touched; uint32_t* counters;                 // nodes with requests; counters[0] = how many, [1] = request slots handed out, [2] = rows left without links
    uint32_t* seg_c; float* seg_d;                         // the requests grouped by node: (new node, distance)
};

__device__ inline uint32_t* hnsw_build_list(const HnswBuildArgs& a, uint32_t node) {
    return a.layer ? a.upper_links + (a.upper_ptr[node] + (a.layer - 1)) * a.su : a.link0 + (size_t)node * a.s0;
}
__device__ inline uint32_t hnsw_build_list_words(const HnswBuildArgs& a) { return a.layer ? a.su : a.s0; }

// distances of the nb_n ids in nb_id[] to the vector qs -> nb_d[] (the search kernel's distance phase: sixteen rows per round when dim % 16 == 0)
__device__ inline void hnsw_wave_distances(const float* qs, const float* __restrict__ X, uint32_t dim, uint32_t ip_lanes, const uint32_t* nb_id, float* nb_d, uint32_t nb_n) {
    const uint32_t lane = threadIdx.x, sub = lane & 15, grp = lane >> 4;
    if (dim % 16 == 0) {
        for (uint32_t i0 = 0; i0 < nb_n; i0 += 16) {
            const uint32_t i = i0 + (lane >> 2);
            const float dot = ip_dot16_quad<VEC_HNSW_CHUNK>(qs, X + (size_t)nb_id[i < nb_n ? i : nb_n - 1] * dim, dim, lane & 3, ip_lanes);
            if (i < nb_n && (lane & 3) == 0) nb_d[i] = ip_add(1.0f, -dot);
        }
    } else {
        for (uint32_t i0 = 0; i0 < nb_n; i0 += 4) {
            const uint32_t i = i0 + grp;
            const float d = ip_distance_group16(qs, X + (size_t)nb_id[i < nb_n ? i : nb_n - 1] * dim, dim, sub, ip_lanes);
            if (i < nb_n && sub == 0) nb_d[i] = d;
        }
    }
    __syncthreads();
}

// getNeighborsByHeuristic2 over n candidates given CLOSEST FIRST (ids / distances to the centre in LDS or global memory): a candidate is kept unless a neighbour
// kept before it is nearer to it than the centre is; at most `limit` (<= 64) are kept -> kept_id[], kept_d[] (LDS), returns how many. One wavefront.
template <typename IdT>
__device__ inline uint32_t hnsw_wave_select(const HnswBuildArgs& a, const IdT* cid, const float* cd, uint32_t n, uint32_t limit, float* qs_lds, uint32_t* kept_id, float* kept_d, float* nb_d) {
    const uint32_t lane = threadIdx.x;
    if (n < limit) {                                                  // (`if (top_candidates.size() < M) return;`)
        for (uint32_t i = lane; i < n; i += 64) { kept_id[i] = (uint32_t)cid[i]; kept_d[i] = cd[i]; }
        __syncthreads();
        return n;
    }
    uint32_t n_kept = 0;
    for (uint32_t i = 0; i < n && n_kept < limit; i++) {
        const uint32_t c = (uint32_t)cid[i];
        const float dq = cd[i];
        bool good = true;
        if (n_kept) {
            const float* qs = a.X + (size_t)c * a.dim;
            if (a.dim <= VEC_HNSW_QDIM) { for (uint32_t j = lane; j < a.dim; j += 64) qs_lds[j] = qs[j]; qs = qs_lds; }
            __syncthreads();
            hnsw_wave_distances(qs, a.X, a.dim, a.ip_lanes, kept_id, nb_d, n_kept);
            good = __ballot((lane < n_kept && nb_d[lane] < dq) ? 1 : 0) == 0ull;
        }
        if (good) { if (lane == 0) { kept_id[n_kept] = c; kept_d[n_kept] = dq; } n_kept++; }
        __syncthreads();
    }
    return n_kept;
}

// kernel 1: one wavefront per new row — its <= M neighbours, its list, its reverse-link requests
__global__ __launch_bounds__(64) void vec_hnsw_build_select_kernel(HnswBuildArgs a) {
    __shared__ __attribute__((aligned(16))) float qs_lds[VEC_HNSW_QDIM];
    __shared__ uint32_t kept_id[64];
    __shared__ float kept_d[64], nb_d[64];
    const uint32_t lane = threadIdx.x;
    for (uint32_t q = blockIdx.x; q < a.n_batch; q += gridDim.x) {
        uint32_t n = a.n_cand[q];
        if (n == 0xFFFFFFFFu) n = 0;
        if (n > a.k) n = a.k;
        const uint32_t node = a.rows[q];
        const uint32_t n_kept = hnsw_wave_select<uint64_t>(a, a.cand_id + (size_t)q * a.k, a.cand_d + (size_t)q * a.k, n, a.M, qs_lds, kept_id, kept_d, nb_d);
        if (n_kept == 0 && lane == 0) atomicAdd(a.counters + 2, 1u);
        uint32_t* lst = hnsw_build_list(a, node);
        if (lane < hnsw_build_list_words(a)) lst[lane] = lane == 0 ? n_kept : (lane - 1 < n_kept ? kept_id[lane - 1] : 0u);        // (slots behind the count stay zero: a canonical image)
        if (lane < a.M) {
            const bool on = lane < n_kept;
            const uint32_t s = on ? kept_id[lane] : 0xFFFFFFFFu;
            a.req_s[(size_t)q * a.M + lane] = s;
            a.req_d[(size_t)q * a.M + lane] = on ? kept_d[lane] : 0.0f;
            if (on && atomicAdd(a.cnt + s, 1u) == 0u) a.touched[atomicAdd(a.counters, 1u)] = s;
        }
        __syncthreads();
    }
}

// kernel 2: a segment of request slots for every node that was asked
__global__ __launch_bounds__(256) void vec_hnsw_build_offsets_kernel(HnswBuildArgs a) {
    const uint32_t n_t = a.counters[0];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_t; i += gridDim.x * blockDim.x) {
        const uint32_t s = a.touched[i];
        a.start[s] = atomicAdd(a.counters + 1, a.cnt[s]);
    }
}

// kernel 3: the requests move into their node's segment (any order: kernel 4 sorts them)
__global__ __launch_bounds__(256) void vec_hnsw_build_scatter_kernel(HnswBuildArgs a) {
    const uint64_t total = (uint64_t)a.n_batch * a.M;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t s = a.req_s[i];
        if (s == 0xFFFFFFFFu) continue;
        const uint32_t pos = a.start[s] + atomicAdd(a.fill + s, 1u);
        a.seg_c[pos] = a.rows[i / a.M];
        a.seg_d[pos] = a.req_d[i];
    }
}

__device__ inline bool hnsw_pair_less(float da, uint32_t ia, float db, uint32_t ib) { return da < db || (da == db && ia < ib); }

// kernel 4: one wavefront per node that was asked — all the batch's requests at once (mutuallyConnectNewElement's reverse links)
__global__ __launch_bounds__(64) void vec_hnsw_build_backlink_kernel(HnswBuildArgs a) {
    __shared__ __attribute__((aligned(16))) float qs_lds[VEC_HNSW_QDIM];
    __shared__ uint32_t c_id[HNSW_BUILD_TCAP], s_id[HNSW_BUILD_TCAP];
    __shared__ float c_d[HNSW_BUILD_TCAP], s_d[HNSW_BUILD_TCAP];
    __shared__ uint32_t kept_id[64];
    __shared__ float kept_d[64], nb_d[64];
    const uint32_t lane = threadIdx.x;
    const uint32_t n_t = a.counters[0];
    const uint32_t lw = hnsw_build_list_words(a), cap = lw - 1;        // 2M on layer 0, M above
    for (uint32_t t = blockIdx.x; t < n_t; t += gridDim.x) {
        const uint32_t s = a.touched[t];
        uint32_t* lst = hnsw_build_list(a, s);
        const uint32_t e = lst[0], m = a.cnt[s], seg = a.start[s];
        const uint32_t room = HNSW_BUILD_TCAP - e;                      // requests looked at: the closest `room` of them
        const uint32_t r = m < room ? m : room;
        // the requests, ordered (distance, id): rank by counting (m is a handful; a hub's hundreds still cost m^2 / 64 compares per lane)
        for (uint32_t i = lane; i < m; i += 64) {
            const float di = a.seg_d[seg + i]; const uint32_t ci = a.seg_c[seg + i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < m; j++) rank += hnsw_pair_less(a.seg_d[seg + j], a.seg_c[seg + j], di, ci) ? 1u : 0u;
            if (rank < r) { c_id[e + rank] = ci; c_d[e + rank] = di; }
        }
        __syncthreads();
        if (e + m <= cap) {                                            // room for all of them: appended, closest first
            if (lane < m) lst[1 + e + lane] = c_id[e + lane];
            if (lane == 0) lst[0] = e + m;
        } else {
            // the list's own members with their distances to s, then everything ordered (distance, id) and the heuristic over it
            const float* qs = a.X + (size_t)s * a.dim;
            if (a.dim <= VEC_HNSW_QDIM) { for (uint32_t j = lane; j < a.dim; j += 64) qs_lds[j] = qs[j]; qs = qs_lds; }
            if (lane < e) c_id[lane] = lst[1 + lane];
            __syncthreads();
            hnsw_wave_distances(qs, a.X, a.dim, a.ip_lanes, c_id, c_d, e);
            const uint32_t T = e + r;
            for (uint32_t i = lane; i < T; i += 64) {
                const float di = c_d[i]; const uint32_t ci = c_id[i];
                uint32_t rank = 0;
                for (uint32_t j = 0; j < T; j++) rank += hnsw_pair_less(c_d[j], c_id[j], di, ci) ? 1u : 0u;
                s_id[rank] = ci; s_d[rank] = di;
            }
            __syncthreads();
            const uint32_t n_kept = hnsw_wave_select<uint32_t>(a, s_id, s_d, T, cap, qs_lds, kept_id, kept_d, nb_d);
            if (lane < lw) lst[lane] = lane == 0 ? n_kept : (lane - 1 < n_kept ? kept_id[lane - 1] : 0u);
        }
        if (lane == 0) { a.cnt[s] = 0; a.fill[s] = 0; }
        __syncthreads();
    }
}

// the level-0 lists of the seed set arrive in compact ids: list i of the compact graph belongs to node ids[i], its members map through ids[] as well
__global__ __launch_bounds__(256) void vec_hnsw_build_place_seed_kernel(const uint32_t* __restrict__ compact_link0, const uint32_t* __restrict__ ids, uint32_t n_seed, uint32_t s0,
                                                                        uint32_t* __restrict__ link0) {
    const uint64_t total = (uint64_t)n_seed * s0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t node = (uint32_t)(i / s0), w = (uint32_t)(i % s0);
        const uint32_t cnt = compact_link0[(size_t)node * s0];
        const uint32_t v = compact_link0[i];
        link0[(size_t)ids[node] * s0 + w] = w == 0 ? v : (w - 1 < cnt ? ids[v] : 0u);
    }
}

// rows of the seed set gathered for the host (compact order)
__global__ __launch_bounds__(256) void vec_hnsw_build_gather_rows_kernel(const float* __restrict__ X, const uint32_t* __restrict__ ids, uint32_t n_seed, uint32_t dim, float* __restrict__ out) {
    const uint64_t total = (uint64_t)n_seed * dim;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = X[(size_t)ids[i / dim] * dim + i % dim];
}

}  // namespace tsgpu
