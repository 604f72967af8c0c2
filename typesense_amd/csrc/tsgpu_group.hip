// tsgpu_group.hip — doc-range shards behind the C-ABI (BASELINE config 5): G contexts, one per GPU, each holding the
// postings / sort columns / vectors of its seq_id range with GLOBAL seq_ids. A batch call scores the WHOLE batch on every member,
// then ONE exchange of the per-member top-k per path (keyword: {key, scores[3](, text_match)} x k + {n_hits, num_matched, status}
// per query; k-NN: one u64 {ord(dist), label} x k per query) and an exact merge in the reference's own order (Topster:
// include/topster.h:146-149; k-NN: (distance, label) ascending). Hybrid fuses AFTER the merge: reciprocal ranks are global ranks
// (src/index.cpp:4036-4221). The reference has no counterpart (it is a single-node index); what is kept is its result.
//
// Two forms:
//   * tsgpu_group_create_local  — ONE process owns G contexts (how the C++ server would link it). Members run on G host threads.
//   * tsgpu_group_create_rank   — one process per GPU (torch.distributed / MPI launchers, bench.py --gpus N): every rank passes its
//                                 context and the 128-byte id rank 0 got from tsgpu_group_unique_id() (broadcast by the host's own channel).
// Three transports (the exchange is the only thing that differs):
//   * TSGPU_XCHG_RCCL — ncclAllGather / ncclAllToAll on the members' streams (RCCL over xGMI). librccl.so.1 is resolved with dlopen when
//                       the first RCCL group is created: the library itself has no link-time dependency on it (single-GPU servers, the test tiers).
//   * TSGPU_XCHG_COPY — device-to-device copies into member 0's gather buffer (local form only; members may share one device: the
//                       single-GPU rehearsal and the emulator tier).
//   * TSGPU_XCHG_HOST — rank form over the CALLER's collectives on host memory (tsgpu_group_create_rank_host: an all-gather and an
//                       all-to-all callback — MPI, gloo, the server's own RPC): blocks are staged through pinned host buffers. Ranks need
//                       not own a GPU each (two ranks may share one device; the emulator tier has none), and it is the transport between
//                       NODES, where there is no xGMI. Same packed blocks, same merge kernels, same results as RCCL.
// Rank form, every transport: the ranks AGREE on the outcome of their local phase before any data collective (one 8-byte all-gather of
// {return code, call signature}): a rank whose shard failed (out of memory, a bad argument) makes every rank return together instead of
// leaving the others inside a collective for ever, and ranks that were handed different arguments (n_queries, k, k_stride, the set of
// optional output arrays) fail with 400 instead of exchanging blocks of different sizes.
// No kernels here: the device-side halves are group_pack_* / group_merge_* (tsgpu.hip, tsgpu_vec.hip).
#include <dlfcn.h>
#include <thread>
#include <chrono>
#include "tsgpu_host.h"

using namespace tsgpu;

namespace {

// the slice of RCCL's C ABI the exchange needs (rccl.h; values are ABI constants of NCCL 2.x)
typedef void* xComm;
struct xUniqueId { char internal[128]; };
enum { X_NCCL_UINT8 = 1, X_NCCL_UINT64 = 5 };
struct RcclApi {
    void* h = nullptr;
    int (*GetUniqueId)(xUniqueId*) = nullptr;
    int (*CommInitRank)(xComm*, int, xUniqueId, int) = nullptr;
    int (*CommInitAll)(xComm*, int, const int*) = nullptr;
    int (*CommDestroy)(xComm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, xComm, hipStream_t) = nullptr;
    int (*AllToAll)(const void*, void*, size_t, int, xComm, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, xComm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, xComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
};
RcclApi* rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {getenv("TSGPU_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) { if (n && *n && (api.h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break; }
        if (!api.h) { api.why = std::string("librccl.so.1 not found (") + (dlerror() ? dlerror() : "?") + ")"; return; }
        auto sym = [&](const char* s) { void* p = dlsym(api.h, s); if (!p && api.why.empty()) api.why = std::string("RCCL symbol missing: ") + s; return p; };
        api.GetUniqueId = (int (*)(xUniqueId*))sym("ncclGetUniqueId");
        api.CommInitRank = (int (*)(xComm*, int, xUniqueId, int))sym("ncclCommInitRank");
        api.CommInitAll = (int (*)(xComm*, int, const int*))sym("ncclCommInitAll");
        api.CommDestroy = (int (*)(xComm))sym("ncclCommDestroy");
        api.AllGather = (int (*)(const void*, void*, size_t, int, xComm, hipStream_t))sym("ncclAllGather");
        api.AllToAll = (int (*)(const void*, void*, size_t, int, xComm, hipStream_t))sym("ncclAllToAll");
        api.Send = (int (*)(const void*, size_t, int, int, xComm, hipStream_t))sym("ncclSend");
        api.Recv = (int (*)(void*, size_t, int, int, xComm, hipStream_t))sym("ncclRecv");
        api.GroupStart = (int (*)())sym("ncclGroupStart");
        api.GroupEnd = (int (*)())sym("ncclGroupEnd");
        api.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
        if (!api.why.empty()) { dlclose(api.h); api.h = nullptr; }
    });
    return &api;
}
int rccl_fail(const char* what, int rc) {
    RcclApi* r = rccl();
    return fail(TSGPU_ERR_DEVICE, std::string("tsgpu_group: ") + what + ": " + (r->GetErrorString ? r->GetErrorString(rc) : "RCCL error"));
}

double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }

struct Member {
    tsgpu_ctx* ctx = nullptr;
    xComm comm = nullptr;
    DevBuf send, recv;                                   // exchange block of this member / the G gathered blocks
    DevBuf l_keys, l_scores, l_tm, l_vd, l_msi, l_nh, l_nm, l_st, l_co;   // this member's local keyword result (device, stride KL)
    DevBuf v_dist, v_lab, v_cnt, v_bad;                  // ... local k-NN result
    DevBuf o_keys, o_scores, o_tm, o_nh, o_nm, o_st, o_vd, o_lab, o_cnt;   // merged result staged on the device (host outputs)
    DevBuf caps;                                         // per-query Topster capacity (the merged list of a query never exceeds its own Topster)
    DevBuf c_fp, c_fp_all, c_tok, c_tok_all;             // dictionary fingerprints / per-query token presence masks of the members (group_keyword_core step 0)
    std::vector<uint16_t> h_tok, h_elsewhere;
    DevBuf c_pass, c_mask, c_found, c_meta, c_meta_all, c_qi;   // candidate combinations: every local hit's pass, the shard's pass masks / union counts, their all-gathered pairs
    DevBuf kth_send, kth_recv, p_tot, p_totall;          // bound-pruned exchange: this shard's reported entries / every shard's; the slices' cursors = entry totals; every rank's totals
    std::vector<uint32_t> h_tot;
    std::vector<uint32_t> h_caps;
    PinBuf h_send, h_recv;                               // HOST transport: the staged blocks
    DevBuf agree_d;                                      // RCCL rank form: the ranks' status words
    PinBuf agree_h;
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // member 0 only: its exchange kernels bracketed (tsgpu_group_timings::exchange_kernels_ms)
    int rc = TSGPU_OK;
    std::string err;
};

}  // namespace

struct tsgpu_group {
    int transport = TSGPU_XCHG_RCCL;
    bool local = true;
    uint32_t n = 1, rank = 0;            // members of the group; this process's member (rank form)
    std::vector<Member> m;               // local form: all members; rank form: the one this process owns
    std::mutex mu;                       // one batch at a time per group
    bool replicas = false;               // every member mirrors the WHOLE collection: the batch is cut into query slices (option "replicas")
    uint32_t test_fail_pack_rank = 0;    // TESTS ONLY (option "test_fail_prune_pack_rank" = rank + 1): that rank fails between the agreement and the sized exchange
    bool own_slice_only = false;         // rank form, slice exchange (option "kw_own_slice_only"): a rank delivers only the slice of the batch it merged — queries
                                         // [rank * per, (rank + 1) * per), per = ceil(n_queries / n_ranks) — into its output arrays (at those queries' slots); no all-gather of
                                         // the merged lists. What a deployment with one request router per rank needs, and what the local form does with host outputs.
    bool kw_pruned_force = false;        // (option value 2: also with one member — exercises the bounds all-gather and the send / recv pairs on a single GPU)
    bool kw_pruned = true;               // keyword exchange: every shard sends only its entries at or above the query's bound (option "kw_exchange_pruned")
    int kw_slices = 1;                   // (2 = also with one member: exercises the collectives on a single GPU) keyword exchange: all-to-all of query slices + slice merge + all-gather of the merged lists (false: one all-gather, full merge on every rank)
    tsgpu_host_collectives coll{};       // TSGPU_XCHG_HOST
    tsgpu_group_timings tm{};
};

namespace {

// run f(member index) for every member this process owns: member 0 on the calling thread, the others on their own threads
// (each blocks in its context's batch call; every member sets its own device)
template <class F> int for_members(tsgpu_group* g, F f) {
    std::vector<std::thread> th;
    for (size_t i = 1; i < g->m.size(); i++) th.emplace_back([&, i] { g->m[i].rc = f(i); if (g->m[i].rc) g->m[i].err = tsgpu_last_error(); });
    g->m[0].rc = f(0);
    if (g->m[0].rc) g->m[0].err = tsgpu_last_error();
    for (auto& t : th) t.join();
    for (auto& mem : g->m) if (mem.rc) return fail(mem.rc, mem.err);
    return TSGPU_OK;
}

// HOST transport: `send_bytes` of this rank's device memory -> pinned host -> the caller's collective -> pinned host -> device `recv`
// (all-gather: n x per_rank_bytes arrive, ordered by rank; all-to-all: send holds n slices of per_rank_bytes, slice j goes to rank j).
// Synchronous on the member's stream: the staging buffers are reused by the next collective of the call.
int host_collective(tsgpu_group* g, bool all_to_all, const void* send_dev, size_t send_bytes, void* recv_dev, size_t per_rank_bytes) {
    Member& mem = g->m[0];
    (void)hipSetDevice(mem.ctx->device);
    int rc;
    if ((rc = mem.h_send.reserve(send_bytes)) || (rc = mem.h_recv.reserve(per_rank_bytes * g->n))) return rc;
    TSGPU_HIP_TRY(hipMemcpyAsync(mem.h_send.p, send_dev, send_bytes, hipMemcpyDeviceToHost, mem.ctx->stream));
    TSGPU_HIP_TRY(hipStreamSynchronize(mem.ctx->stream));
    const int crc = all_to_all ? g->coll.all_to_all(g->coll.user, mem.h_send.p, mem.h_recv.p, per_rank_bytes) : g->coll.all_gather(g->coll.user, mem.h_send.p, mem.h_recv.p, per_rank_bytes);
    if (crc) return fail(TSGPU_ERR_DEVICE, std::string("tsgpu_group: the caller's ") + (all_to_all ? "all_to_all" : "all_gather") + " callback failed (" + std::to_string(crc) + ")");
    TSGPU_HIP_TRY(hipMemcpyAsync(recv_dev, mem.h_recv.p, per_rank_bytes * g->n, hipMemcpyHostToDevice, mem.ctx->stream));
    TSGPU_HIP_TRY(hipStreamSynchronize(mem.ctx->stream));
    return TSGPU_OK;
}

// member 0's exchange kernels are timed with HIP events on its stream (three brackets: pack or bounds | prune + pack | merge)
void mark(tsgpu_group* g, size_t member, int i) {
    if (member != 0) return;
    Member& mem = g->m[0];
    if (!mem.ev[i]) { (void)hipSetDevice(mem.ctx->device); if (hipEventCreate(&mem.ev[i]) != hipSuccess) { mem.ev[i] = nullptr; return; } }
    (void)hipEventRecord(mem.ev[i], mem.ctx->stream);
}
float marked_ms(tsgpu_group* g, bool mid) {
    Member& mem = g->m[0];
    float tot = 0, ms = 0;
    for (int b = 0; b < 4; b++) {
        if (b == 3 || (b == 1 && !mid)) continue;          // brackets: 0 pack or bounds | 1 prune + pack | 2 merge
        if (mem.ev[2 * b] && mem.ev[2 * b + 1] && hipEventElapsedTime(&ms, mem.ev[2 * b], mem.ev[2 * b + 1]) == hipSuccess) {
            tot += ms;
            if (getenv("TSGPU_GROUP_DEBUG")) fprintf(stderr, "[tsgpu_group] exchange kernels, bracket %d: %.3f ms\n", b, ms);
        }
    }
    return tot;
}

// HOST transport, bound-pruned slices: the used prefix (used_bytes) of each of this rank's n slices (stride_bytes apart in its send buffer) -> pinned host,
// the caller's all_to_all on equal pieces of used_bytes, one copy back: the received pieces are used_bytes apart in the recv buffer
int host_all_to_all_prefix(tsgpu_group* g, size_t stride_bytes, size_t used_bytes) {
    Member& mem = g->m[0];
    (void)hipSetDevice(mem.ctx->device);
    int rc;
    if ((rc = mem.h_send.reserve(used_bytes * g->n)) || (rc = mem.h_recv.reserve(used_bytes * g->n)) || (rc = mem.recv.reserve(used_bytes * g->n))) return rc;   // (all within the local phase's reservations)
    for (uint32_t j = 0; j < g->n; j++) TSGPU_HIP_TRY(hipMemcpyAsync((char*)mem.h_send.p + (size_t)j * used_bytes, (const char*)mem.send.p + (size_t)j * stride_bytes, used_bytes, hipMemcpyDeviceToHost, mem.ctx->stream));
    TSGPU_HIP_TRY(hipStreamSynchronize(mem.ctx->stream));
    const int crc = g->coll.all_to_all(g->coll.user, mem.h_send.p, mem.h_recv.p, used_bytes);
    if (crc) return fail(TSGPU_ERR_DEVICE, "tsgpu_group: the caller's all_to_all callback failed (" + std::to_string(crc) + ")");
    TSGPU_HIP_TRY(hipMemcpyAsync(mem.recv.p, mem.h_recv.p, used_bytes * g->n, hipMemcpyHostToDevice, mem.ctx->stream));
    TSGPU_HIP_TRY(hipStreamSynchronize(mem.ctx->stream));
    return TSGPU_OK;
}

uint32_t call_signature(std::initializer_list<uint64_t> v) {
    uint64_t h = 1469598103934665603ull;
    for (uint64_t x : v) { h ^= x; h *= 1099511628211ull; }
    return (uint32_t)(h ^ (h >> 32));
}

// Rank form: every rank reports {rc of its local phase, signature of the call's arguments}; all ranks leave with the same verdict
// BEFORE any data collective. The failing rank keeps its own error message; the others learn which rank failed.
// (round 6: a second word rides along — `extra`, e.g. the rank's dictionary fingerprint; *extra_equal = every rank sent the same one)
int agree(tsgpu_group* g, int rc_local, uint32_t sig, uint64_t extra = 0, bool* extra_equal = nullptr) {
    if (extra_equal) *extra_equal = true;
    if (g->local || g->n == 1) return rc_local;
    Member& mem = g->m[0];
    const std::string own_err = rc_local ? std::string(tsgpu_last_error()) : std::string();
    const uint64_t w[2] = {((uint64_t)(uint32_t)rc_local << 32) | sig, extra};
    std::vector<uint64_t> both((size_t)g->n * 2, 0), all(g->n, 0);
    (void)hipSetDevice(mem.ctx->device);
    if (g->transport == TSGPU_XCHG_HOST) {
        const int crc = g->coll.all_gather(g->coll.user, w, both.data(), 16);
        if (crc) return fail(TSGPU_ERR_DEVICE, "tsgpu_group: the caller's all_gather callback failed (" + std::to_string(crc) + ")");
    } else {
        int rc;
        if ((rc = mem.agree_d.reserve((size_t)(g->n + 1) * 16)) || (rc = mem.agree_h.reserve((size_t)(g->n + 1) * 16))) return rc;     // (64 KB minimum each: allocated once)
        uint64_t* hw = mem.agree_h.as<uint64_t>();
        hw[2 * g->n] = w[0]; hw[2 * g->n + 1] = w[1];
        TSGPU_HIP_TRY(hipMemcpyAsync(mem.agree_d.as<uint64_t>() + 2 * g->n, hw + 2 * g->n, 16, hipMemcpyHostToDevice, mem.ctx->stream));
        if ((rc = rccl()->AllGather(mem.agree_d.as<uint64_t>() + 2 * g->n, mem.agree_d.p, 2, X_NCCL_UINT64, mem.comm, mem.ctx->stream))) return rccl_fail("ncclAllGather (status)", rc);
        TSGPU_HIP_TRY(hipMemcpyAsync(hw, mem.agree_d.p, (size_t)g->n * 16, hipMemcpyDeviceToHost, mem.ctx->stream));
        TSGPU_HIP_TRY(hipStreamSynchronize(mem.ctx->stream));
        for (uint32_t r = 0; r < 2 * g->n; r++) both[r] = hw[r];
    }
    for (uint32_t r = 0; r < g->n; r++) { all[r] = both[2 * r]; if (extra_equal && both[2 * r + 1] != both[1]) *extra_equal = false; }
    for (uint32_t r = 0; r < g->n; r++) {
        const int rr = (int)(uint32_t)(all[r] >> 32);
        if (!rr) continue;
        if (r == g->rank) return fail(rr, own_err);
        return fail(rr, "tsgpu_group: rank " + std::to_string(r) + " failed in its local phase (code " + std::to_string(rr) + "): the call is abandoned on every rank");
    }
    for (uint32_t r = 0; r < g->n; r++)
        if ((uint32_t)all[r] != sig) return fail(TSGPU_ERR_INVALID, "tsgpu_group: rank " + std::to_string(r) + " was called with different arguments (n_queries / k / k_stride / optional output arrays / options must be the same on every rank)");
    return TSGPU_OK;
}

// every owned member's block (bytes each) -> the gathered [n][bytes] in the recv buffer of every member (RCCL / HOST) / of member 0 (COPY)
int exchange(tsgpu_group* g, size_t bytes) {
    for (auto& mem : g->m) { int rc = mem.recv.reserve(bytes * g->n); if (rc) return rc; }
    if (g->transport == TSGPU_XCHG_RCCL) {
        RcclApi* r = rccl();
        int rc;
        if (g->m.size() > 1 && (rc = r->GroupStart())) return rccl_fail("ncclGroupStart", rc);
        for (auto& mem : g->m) {
            (void)hipSetDevice(mem.ctx->device);
            if ((rc = r->AllGather(mem.send.p, mem.recv.p, bytes / 8, X_NCCL_UINT64, mem.comm, mem.ctx->stream))) { if (g->m.size() > 1) (void)r->GroupEnd(); return rccl_fail("ncclAllGather", rc); }
        }
        if (g->m.size() > 1 && (rc = r->GroupEnd())) return rccl_fail("ncclGroupEnd", rc);
        return TSGPU_OK;
    }
    if (g->transport == TSGPU_XCHG_HOST) return host_collective(g, false, g->m[0].send.p, bytes, g->m[0].recv.p, bytes);
    // COPY: the packs must have finished on their own streams, then member 0's stream pulls every block
    for (auto& mem : g->m) { (void)hipSetDevice(mem.ctx->device); TSGPU_HIP_TRY(hipStreamSynchronize(mem.ctx->stream)); }
    Member& root = g->m[0];
    (void)hipSetDevice(root.ctx->device);
    for (size_t j = 0; j < g->m.size(); j++) {
        Member& src = g->m[j];
        if (src.ctx->device == root.ctx->device) TSGPU_HIP_TRY(hipMemcpyAsync((char*)root.recv.p + j * bytes, src.send.p, bytes, hipMemcpyDeviceToDevice, root.ctx->stream));
        else TSGPU_HIP_TRY(hipMemcpyPeerAsync((char*)root.recv.p + j * bytes, root.ctx->device, src.send.p, src.ctx->device, bytes, root.ctx->stream));
    }
    return TSGPU_OK;
}

int copy_between(Member& dst, void* d, Member& src, const void* s_, size_t bytes);
// `bytes` of every member's `send` buffer -> [n][bytes] in EVERY owned member's `recv` buffer (both reserved by the caller in the local phase)
int all_gather_everywhere(tsgpu_group* g, DevBuf Member::*send, DevBuf Member::*recv, size_t bytes) {
    if (g->transport == TSGPU_XCHG_RCCL) {
        RcclApi* r = rccl();
        int rc;
        if (g->m.size() > 1 && (rc = r->GroupStart())) return rccl_fail("ncclGroupStart", rc);
        for (auto& mem : g->m) {
            (void)hipSetDevice(mem.ctx->device);
            if ((rc = r->AllGather((mem.*send).p, (mem.*recv).p, bytes / 8, X_NCCL_UINT64, mem.comm, mem.ctx->stream))) { if (g->m.size() > 1) (void)r->GroupEnd(); return rccl_fail("ncclAllGather (bounds)", rc); }
        }
        if (g->m.size() > 1 && (rc = r->GroupEnd())) return rccl_fail("ncclGroupEnd", rc);
        return TSGPU_OK;
    }
    if (g->transport == TSGPU_XCHG_HOST) return host_collective(g, false, (g->m[0].*send).p, bytes, (g->m[0].*recv).p, bytes);
    for (auto& mem : g->m) { (void)hipSetDevice(mem.ctx->device); TSGPU_HIP_TRY(hipStreamSynchronize(mem.ctx->stream)); }
    for (size_t d = 0; d < g->m.size(); d++) {
        (void)hipSetDevice(g->m[d].ctx->device);
        for (size_t j = 0; j < g->m.size(); j++) { int rc = copy_between(g->m[d], (char*)(g->m[d].*recv).p + j * bytes, g->m[j], (g->m[j].*send).p, bytes); if (rc) return rc; }
    }
    return TSGPU_OK;
}
// bound-pruned exchange: every member's per-destination entry totals (p_tot: n_dst u32 on its device) -> tot_all[src * n_dst + dst] on the host
// rc_local (rank form): this rank failed between the agreement and here. It still takes part — with totals of 0xFFFFFFFF, which no slice can have — so that
// every rank sees the failure in the same collective and all of them leave together (nobody waits in the sized exchange for a rank that returned)
int gather_totals(tsgpu_group* g, uint32_t n_dst, std::vector<uint32_t>& tot_all, int rc_local = TSGPU_OK) {
    tot_all.assign((size_t)g->n * n_dst, 0u);
    if (g->local) {
        if (rc_local) return rc_local;
        for (size_t i = 0; i < g->m.size(); i++) {
            Member& mem = g->m[i];
            (void)hipSetDevice(mem.ctx->device);
            TSGPU_HIP_TRY(hipMemcpyAsync(tot_all.data() + i * n_dst, mem.p_tot.p, (size_t)n_dst * 4, hipMemcpyDeviceToHost, mem.ctx->stream));
        }
        for (auto& mem : g->m) { (void)hipSetDevice(mem.ctx->device); TSGPU_HIP_TRY(hipStreamSynchronize(mem.ctx->stream)); }
        return TSGPU_OK;
    }
    Member& mem = g->m[0];
    (void)hipSetDevice(mem.ctx->device);
    if (g->transport == TSGPU_XCHG_HOST) {
        std::vector<uint32_t> mine(n_dst, rc_local ? 0xFFFFFFFFu : 0u);
        if (!rc_local) {
            TSGPU_HIP_TRY(hipMemcpyAsync(mine.data(), mem.p_tot.p, (size_t)n_dst * 4, hipMemcpyDeviceToHost, mem.ctx->stream));
            TSGPU_HIP_TRY(hipStreamSynchronize(mem.ctx->stream));
        }
        const int crc = g->coll.all_gather(g->coll.user, mine.data(), tot_all.data(), (size_t)n_dst * 4);
        if (crc) return fail(TSGPU_ERR_DEVICE, "tsgpu_group: the caller's all_gather callback failed (" + std::to_string(crc) + ")");
    } else {
        int rc;
        if (rc_local) (void)hipMemsetAsync(mem.p_tot.p, 0xFF, (size_t)n_dst * 4, mem.ctx->stream);
        if ((rc = rccl()->AllGather(mem.p_tot.p, mem.p_totall.p, (size_t)n_dst * 4, X_NCCL_UINT8, mem.comm, mem.ctx->stream))) return rccl_fail("ncclAllGather (slice totals)", rc);
        TSGPU_HIP_TRY(hipMemcpyAsync(tot_all.data(), mem.p_totall.p, tot_all.size() * 4, hipMemcpyDeviceToHost, mem.ctx->stream));
        TSGPU_HIP_TRY(hipStreamSynchronize(mem.ctx->stream));
    }
    if (rc_local) return rc_local;
    for (uint32_t v : tot_all) if (v == 0xFFFFFFFFu) return fail(TSGPU_ERR_DEVICE, "tsgpu_group: another rank failed while it packed its slices of the bound-pruned exchange");
    return TSGPU_OK;
}

int copy_between(Member& dst, void* d, Member& src, const void* s_, size_t bytes) {
    if (!bytes) return TSGPU_OK;
    if (src.ctx->device == dst.ctx->device) TSGPU_HIP_TRY(hipMemcpyAsync(d, s_, bytes, hipMemcpyDeviceToDevice, dst.ctx->stream));
    else TSGPU_HIP_TRY(hipMemcpyPeerAsync(d, dst.ctx->device, s_, src.ctx->device, bytes, dst.ctx->stream));
    return TSGPU_OK;
}

// all-to-all: slice j (slice_bytes) of every member's send buffer -> member j's recv buffer, ordered by source member
int exchange_slices(tsgpu_group* g, size_t slice_bytes) {
    for (auto& mem : g->m) { int rc = mem.recv.reserve(slice_bytes * g->n); if (rc) return rc; }
    if (g->transport == TSGPU_XCHG_RCCL) {
        RcclApi* r = rccl();
        int rc;
        if (g->m.size() > 1 && (rc = r->GroupStart())) return rccl_fail("ncclGroupStart", rc);
        for (auto& mem : g->m) {
            (void)hipSetDevice(mem.ctx->device);
            if ((rc = r->AllToAll(mem.send.p, mem.recv.p, slice_bytes / 8, X_NCCL_UINT64, mem.comm, mem.ctx->stream))) { if (g->m.size() > 1) (void)r->GroupEnd(); return rccl_fail("ncclAllToAll", rc); }
        }
        if (g->m.size() > 1 && (rc = r->GroupEnd())) return rccl_fail("ncclGroupEnd", rc);
        return TSGPU_OK;
    }
    if (g->transport == TSGPU_XCHG_HOST) return host_collective(g, true, g->m[0].send.p, slice_bytes * g->n, g->m[0].recv.p, slice_bytes);
    for (auto& mem : g->m) { (void)hipSetDevice(mem.ctx->device); TSGPU_HIP_TRY(hipStreamSynchronize(mem.ctx->stream)); }
    for (size_t d = 0; d < g->m.size(); d++) {
        (void)hipSetDevice(g->m[d].ctx->device);
        for (size_t j = 0; j < g->m.size(); j++) { int rc = copy_between(g->m[d], (char*)g->m[d].recv.p + j * slice_bytes, g->m[j], (const char*)g->m[j].send.p + d * slice_bytes, slice_bytes); if (rc) return rc; }
    }
    return TSGPU_OK;
}

// bound-pruned all-to-all at exact sizes: slice j of a member's send buffer (slice_words apart) = `per` header pairs + tot[src][j] entries -> slot src
// of member j's recv buffer (slots slice_words apart: the merge kernel's stride). RCCL: grouped ncclSend / ncclRecv pairs; COPY: device copies.
int exchange_slices_exact(tsgpu_group* g, size_t slice_words, const std::vector<uint32_t>& tot, uint32_t per, uint32_t words) {
    for (auto& mem : g->m) { int rc = mem.recv.reserve(slice_words * 8 * g->n); if (rc) return rc; }      // (reserved in the local phase at the unpruned size: never grows here)
    auto used = [&](uint32_t src, uint32_t dst) { return ((size_t)per * 2 + (size_t)tot[(size_t)src * g->n + dst] * words) * 8; };
    if (g->transport == TSGPU_XCHG_RCCL) {
        RcclApi* r = rccl();
        int rc;
        if ((rc = r->GroupStart())) return rccl_fail("ncclGroupStart", rc);
        for (size_t i = 0; i < g->m.size(); i++) {
            Member& mem = g->m[i];
            (void)hipSetDevice(mem.ctx->device);
            const uint32_t me = g->local ? (uint32_t)i : g->rank;
            for (uint32_t j = 0; j < g->n; j++) {
                if ((rc = r->Send((const char*)mem.send.p + (size_t)j * slice_words * 8, used(me, j), X_NCCL_UINT8, (int)j, mem.comm, mem.ctx->stream)) ||
                    (rc = r->Recv((char*)mem.recv.p + (size_t)j * slice_words * 8, used(j, me), X_NCCL_UINT8, (int)j, mem.comm, mem.ctx->stream))) { (void)r->GroupEnd(); return rccl_fail("ncclSend / ncclRecv (pruned slices)", rc); }
            }
        }
        if ((rc = r->GroupEnd())) return rccl_fail("ncclGroupEnd", rc);
        return TSGPU_OK;
    }
    for (auto& mem : g->m) { (void)hipSetDevice(mem.ctx->device); TSGPU_HIP_TRY(hipStreamSynchronize(mem.ctx->stream)); }
    for (size_t d = 0; d < g->m.size(); d++) {
        (void)hipSetDevice(g->m[d].ctx->device);
        for (size_t j = 0; j < g->m.size(); j++) { int rc = copy_between(g->m[d], (char*)g->m[d].recv.p + j * slice_words * 8, g->m[j], (const char*)g->m[j].send.p + d * slice_words * 8, used((uint32_t)j, (uint32_t)d)); if (rc) return rc; }
    }
    return TSGPU_OK;
}

// every member holds its slice [i * per, (i + 1) * per) of `elem`-byte records in its own copy of an array: afterwards every member
// (RCCL: in-place ncclAllGather) / member 0 (COPY) holds all slices
int replicate_slices(tsgpu_group* g, const std::vector<void*>& arr /* per member */, size_t slice_bytes) {
    if (g->transport == TSGPU_XCHG_RCCL) {
        RcclApi* r = rccl();
        for (size_t i = 0; i < g->m.size(); i++) {
            Member& mem = g->m[i];
            (void)hipSetDevice(mem.ctx->device);
            const uint32_t rank = g->local ? (uint32_t)i : g->rank;
            int rc = r->AllGather((const char*)arr[i] + (size_t)rank * slice_bytes, arr[i], slice_bytes, X_NCCL_UINT8, mem.comm, mem.ctx->stream);
            if (rc) return rccl_fail("ncclAllGather (merged lists)", rc);
        }
        return TSGPU_OK;
    }
    if (g->transport == TSGPU_XCHG_HOST) return host_collective(g, false, (const char*)arr[0] + (size_t)g->rank * slice_bytes, slice_bytes, arr[0], slice_bytes);
    // COPY: the slices were written on their owners' streams (merge / store kernels): they must have finished before root's stream reads them
    for (size_t j = 1; j < g->m.size(); j++) { (void)hipSetDevice(g->m[j].ctx->device); TSGPU_HIP_TRY(hipStreamSynchronize(g->m[j].ctx->stream)); }
    Member& root = g->m[0];
    (void)hipSetDevice(root.ctx->device);
    for (size_t j = 1; j < g->m.size(); j++) { int rc = copy_between(root, (char*)arr[0] + j * slice_bytes, g->m[j], (const char*)arr[j] + j * slice_bytes, slice_bytes); if (rc) return rc; }
    return TSGPU_OK;
}

uint32_t local_topster_stride(const tsgpu_kw_query* q, uint32_t n, uint32_t k) {
    uint32_t ks = k;
    for (uint32_t i = 0; i < n; i++) ks = std::max<uint32_t>(ks, q[i].topster_size ? std::min<uint32_t>(q[i].topster_size, TSGPU_MAX_TOPK) : TSGPU_DEFAULT_TOPSTER_SIZE);
    return ks;
}

int copy_out(void* dst, const void* src, size_t bytes, int mem_out, hipStream_t s) {
    if (!dst || !bytes) return TSGPU_OK;
    TSGPU_HIP_TRY(hipMemcpyAsync(dst, src, bytes, mem_out == TSGPU_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, s));
    return TSGPU_OK;
}


struct KwArr { DevBuf Member::*buf; void* dst; size_t elem; };

// merged / own slices are staged per member in full-batch arrays (n_pad queries, stride = the caller's k_stride); deliver them:
// local form + host outputs: every member copies its own slice over its own PCIe link; otherwise the slices are replicated
// (RCCL: in-place all-gathers, every rank ends with everything; COPY: into member 0) and copied out once
int deliver_slices(tsgpu_group* g, const tsgpu_hits* out, uint32_t n_queries, uint32_t per, bool force = false) {
    const size_t KS = out->k_stride;
    const bool to_host = out->mem == TSGPU_MEM_HOST;
    const KwArr arrs[] = {{&Member::o_keys, out->keys, KS * 8}, {&Member::o_scores, out->scores, KS * 24}, {&Member::o_tm, out->text_match, KS * 8},
                          {&Member::o_nh, out->n_hits, 4}, {&Member::o_nm, out->num_matched, 8}, {&Member::o_st, out->status, 4}};
    int rc;
    if (!g->local && g->own_slice_only) {                 // rank form: this rank's merged slice, nothing else (the other slots of `out` stay untouched)
        Member& mem = g->m[0];
        (void)hipSetDevice(mem.ctx->device);
        const uint32_t q0 = g->rank * per;
        if (q0 < n_queries) {
            const uint32_t nq = std::min<uint32_t>(per, n_queries - q0);
            for (const KwArr& a : arrs) if (a.dst)
                if ((rc = copy_out((char*)a.dst + (size_t)q0 * a.elem, (const char*)(mem.*(a.buf)).p + (size_t)q0 * a.elem, (size_t)nq * a.elem, out->mem, mem.ctx->stream))) return rc;
        }
        return TSGPU_OK;
    }
    if ((g->n > 1 || force) && !(g->local && to_host)) {
        const bool grouped = g->transport == TSGPU_XCHG_RCCL && g->m.size() > 1;
        if (grouped) { int r2 = rccl()->GroupStart(); if (r2) return rccl_fail("ncclGroupStart", r2); }
        for (const KwArr& a : arrs) {
            if (!a.dst) continue;
            std::vector<void*> per_member;
            for (auto& mem : g->m) per_member.push_back((mem.*(a.buf)).p);
            if ((rc = replicate_slices(g, per_member, (size_t)per * a.elem))) { if (grouped) (void)rccl()->GroupEnd(); return rc; }
        }
        if (grouped) { int r2 = rccl()->GroupEnd(); if (r2) return rccl_fail("ncclGroupEnd", r2); }
    }
    if (g->n > 1 && g->local && to_host) {
        for (size_t i = 0; i < g->m.size(); i++) {
            Member& mem = g->m[i];
            (void)hipSetDevice(mem.ctx->device);
            const uint32_t q0 = (uint32_t)i * per;
            if (q0 >= n_queries) break;
            const uint32_t nq = std::min<uint32_t>(per, n_queries - q0);
            for (const KwArr& a : arrs) if (a.dst)
                if ((rc = copy_out((char*)a.dst + (size_t)q0 * a.elem, (const char*)(mem.*(a.buf)).p + (size_t)q0 * a.elem, (size_t)nq * a.elem, TSGPU_MEM_HOST, mem.ctx->stream))) return rc;
        }
    } else {
        Member& root = g->m[0];
        (void)hipSetDevice(root.ctx->device);
        for (const KwArr& a : arrs) if (a.dst)
            if ((rc = copy_out(a.dst, (root.*(a.buf)).p, (size_t)n_queries * a.elem, out->mem, root.ctx->stream))) return rc;
    }
    return TSGPU_OK;
}

void fill_keyword_constants(const tsgpu_kw_query* queries, uint32_t n_queries, tsgpu_hits* out) {
    // per-hit constants of a keyword pass: vector_distance = -1 (include/topster.h:29); match_score_index = position of _text_match among the sort keys
    if (out->mem != TSGPU_MEM_HOST || !(out->match_score_index || out->vector_distance)) return;
    for (uint32_t q = 0; q < n_queries; q++) {
        int8_t msi = -1;
        for (uint32_t j = 0; j < queries[q].n_sort && j < TSGPU_MAX_SORT_KEYS; j++) if (queries[q].sort[j].kind == TSGPU_SORT_TEXT_MATCH) { msi = (int8_t)j; break; }
        for (uint32_t i = 0; i < out->k_stride; i++) {
            if (out->match_score_index) out->match_score_index[(size_t)q * out->k_stride + i] = msi;
            if (out->vector_distance) out->vector_distance[(size_t)q * out->k_stride + i] = -1.0f;
        }
    }
}

tsgpu_hits staged_hits(Member& mem, const tsgpu_hits* out) {
    tsgpu_hits d;
    memset(&d, 0, sizeof d);
    d.mem = TSGPU_MEM_DEVICE; d.k_stride = out->k_stride;
    d.keys = mem.o_keys.as<uint64_t>(); d.scores = mem.o_scores.as<int64_t>(); d.text_match = out->text_match ? mem.o_tm.as<int64_t>() : nullptr;
    d.n_hits = mem.o_nh.as<uint32_t>(); d.num_matched = mem.o_nm.as<uint64_t>(); d.status = mem.o_st.as<int32_t>();
    return d;
}

int reserve_staging(Member& mem, const tsgpu_hits* out, uint32_t n_pad) {
    int rc;
    const size_t slots = (size_t)n_pad * out->k_stride;
    (void)hipSetDevice(mem.ctx->device);
    if ((rc = mem.o_keys.reserve(slots * 8)) || (rc = mem.o_scores.reserve(slots * 24)) || (out->text_match && (rc = mem.o_tm.reserve(slots * 8))) ||
        (rc = mem.o_nh.reserve((size_t)n_pad * 4)) || (rc = mem.o_nm.reserve((size_t)n_pad * 8)) || (rc = mem.o_st.reserve((size_t)n_pad * 4)) || (rc = mem.caps.reserve((size_t)n_pad * 4))) return rc;
    return TSGPU_OK;
}

uint64_t hits_mask(const tsgpu_hits* o) {
    return (o->keys ? 1u : 0) | (o->scores ? 2u : 0) | (o->text_match ? 4u : 0) | (o->n_hits ? 8u : 0) | (o->num_matched ? 16u : 0) | (o->status ? 32u : 0) | ((uint64_t)o->mem << 8);
}
// HOST transport: the pinned staging buffers are sized in the LOCAL phase, so that nothing can fail on one rank alone after the agreement
int reserve_host_staging(tsgpu_group* g, Member& mem, size_t bytes) {
    if (g->transport != TSGPU_XCHG_HOST) return TSGPU_OK;
    int rc;
    if ((rc = mem.h_send.reserve(bytes)) || (rc = mem.h_recv.reserve(bytes))) return rc;
    return TSGPU_OK;
}

// replicas form: member i answers queries [i * per, (i + 1) * per) of the batch on its own full mirror; the slices are then delivered /
// replicated like merged slices. No merge: a member's Topster for a query IS the global one.
int keyword_replicas(tsgpu_group* g, const tsgpu_kw_query* queries, uint32_t n_queries, uint32_t k, tsgpu_hits* out) {
    const uint32_t per = (n_queries + g->n - 1) / g->n, n_pad = per * g->n;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = for_members(g, [&](size_t i) -> int {
        Member& mem = g->m[i];
        const uint32_t rank = g->local ? (uint32_t)i : g->rank;
        const uint32_t q0 = rank * per, nq = q0 < n_queries ? std::min<uint32_t>(per, n_queries - q0) : 0;
        int r;
        if ((r = reserve_staging(mem, out, n_pad)) || (r = reserve_host_staging(g, mem, (size_t)n_pad * out->k_stride * 24))) return r;
        tsgpu_hits st = staged_hits(mem, out);
        TSGPU_HIP_TRY(hipMemsetAsync(mem.o_nh.p, 0, (size_t)n_pad * 4, mem.ctx->stream));
        TSGPU_HIP_TRY(hipMemsetAsync(mem.o_st.p, 0, (size_t)n_pad * 4, mem.ctx->stream));
        if (nq == 0) return TSGPU_OK;
        const uint32_t KL = local_topster_stride(queries + q0, nq, k);
        const size_t slots = (size_t)nq * KL;
        if ((r = mem.l_keys.reserve(slots * 8)) || (r = mem.l_scores.reserve(slots * 24)) || (r = mem.l_tm.reserve(slots * 8)) || (r = mem.l_vd.reserve(slots * 4)) || (r = mem.l_msi.reserve(slots)) ||
            (r = mem.l_nh.reserve((size_t)nq * 4)) || (r = mem.l_nm.reserve((size_t)nq * 8)) || (r = mem.l_st.reserve((size_t)nq * 4)) || (r = mem.l_co.reserve((size_t)nq * 4))) return r;
        tsgpu_hits loc;
        memset(&loc, 0, sizeof loc);
        loc.mem = TSGPU_MEM_DEVICE; loc.k_stride = KL;
        loc.keys = mem.l_keys.as<uint64_t>(); loc.scores = mem.l_scores.as<int64_t>(); loc.text_match = mem.l_tm.as<int64_t>();
        loc.vector_distance = mem.l_vd.as<float>(); loc.match_score_index = mem.l_msi.as<int8_t>();
        loc.n_hits = mem.l_nh.as<uint32_t>(); loc.num_matched = mem.l_nm.as<uint64_t>(); loc.status = mem.l_st.as<int32_t>(); loc.search_cutoff = mem.l_co.as<int32_t>();
        if ((r = tsgpu_keyword_search_batch(mem.ctx, queries + q0, nq, &loc))) return r;
        return group_store_keyword_slice(mem.ctx, &loc, nq, q0, k, &st, mem.ctx->stream);
    });
    if ((rc = agree(g, rc, call_signature({1, n_queries, k, out->k_stride, hits_mask(out)})))) return rc;
    const double t_local = ms_since(t0);
    const auto t1 = std::chrono::steady_clock::now();
    if ((rc = deliver_slices(g, out, n_queries, per))) return rc;
    if (out->search_cutoff) { if (out->mem == TSGPU_MEM_HOST) memset(out->search_cutoff, 0, (size_t)n_queries * 4); else TSGPU_HIP_TRY(hipMemsetAsync(out->search_cutoff, 0, (size_t)n_queries * 4, g->m[0].ctx->stream)); }
    fill_keyword_constants(queries, n_queries, out);
    for (auto& mem : g->m) { (void)hipSetDevice(mem.ctx->device); TSGPU_HIP_TRY(hipStreamSynchronize(mem.ctx->stream)); }
    g->tm.local_ms = (float)t_local; g->tm.exchange_merge_ms = (float)ms_since(t1);
    g->tm.exchange_bytes_per_member = (uint64_t)per * ((size_t)out->k_stride * (32 + (out->text_match ? 8 : 0)) + 16) * (g->n - 1);
    return ok();
}

}  // namespace

extern "C" {

int tsgpu_group_unique_id(uint8_t id[128]) {
    if (!id) return fail(TSGPU_ERR_INVALID, "tsgpu_group_unique_id: NULL argument");
    RcclApi* r = rccl();
    if (!r->h) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_group: RCCL transport unavailable: " + r->why);
    xUniqueId u;
    int rc = r->GetUniqueId(&u);
    if (rc) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(id, u.internal, 128);
    return ok();
}

int tsgpu_group_create_local(tsgpu_ctx* const* members, uint32_t n_members, int transport, tsgpu_group** out) {
    if (!members || !out || n_members == 0 || n_members > 64) return fail(TSGPU_ERR_INVALID, "tsgpu_group_create_local: bad arguments");
    *out = nullptr;
    for (uint32_t i = 0; i < n_members; i++) if (!members[i]) return fail(TSGPU_ERR_INVALID, "tsgpu_group_create_local: NULL member");
    if (transport != TSGPU_XCHG_RCCL && transport != TSGPU_XCHG_COPY) return fail(TSGPU_ERR_INVALID, "tsgpu_group_create_local: unknown transport");
    std::unique_ptr<tsgpu_group> g(new (std::nothrow) tsgpu_group);
    if (!g) return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_group_create_local: host allocation failed");
    g->transport = transport; g->local = true; g->n = n_members; g->rank = 0;
    g->m.resize(n_members);
    for (uint32_t i = 0; i < n_members; i++) g->m[i].ctx = members[i];
    if (transport == TSGPU_XCHG_RCCL) {
        RcclApi* r = rccl();
        if (!r->h) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_group: RCCL transport unavailable: " + r->why);
        std::vector<int> devs(n_members);
        for (uint32_t i = 0; i < n_members; i++) {
            devs[i] = members[i]->device;
            for (uint32_t j = 0; j < i; j++) if (devs[j] == devs[i]) return fail(TSGPU_ERR_INVALID, "tsgpu_group_create_local: RCCL needs one GPU per member (use TSGPU_XCHG_COPY for members that share a device)");
        }
        std::vector<xComm> comms(n_members, nullptr);
        int rc = r->CommInitAll(comms.data(), (int)n_members, devs.data());
        if (rc) return rccl_fail("ncclCommInitAll", rc);
        for (uint32_t i = 0; i < n_members; i++) g->m[i].comm = comms[i];
    }
    *out = g.release();
    return ok();
}

int tsgpu_group_create_rank(tsgpu_ctx* ctx, const uint8_t id[128], uint32_t rank, uint32_t n_ranks, tsgpu_group** out) {
    if (!ctx || !id || !out || n_ranks == 0 || rank >= n_ranks) return fail(TSGPU_ERR_INVALID, "tsgpu_group_create_rank: bad arguments");
    *out = nullptr;
    RcclApi* r = rccl();
    if (!r->h) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_group: RCCL transport unavailable: " + r->why);
    std::unique_ptr<tsgpu_group> g(new (std::nothrow) tsgpu_group);
    if (!g) return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_group_create_rank: host allocation failed");
    g->transport = TSGPU_XCHG_RCCL; g->local = false; g->n = n_ranks; g->rank = rank;
    g->m.resize(1);
    g->m[0].ctx = ctx;
    (void)hipSetDevice(ctx->device);
    xUniqueId u;
    memcpy(u.internal, id, 128);
    int rc = r->CommInitRank(&g->m[0].comm, (int)n_ranks, u, (int)rank);
    if (rc) return rccl_fail("ncclCommInitRank", rc);
    // the agreement step's buffers exist from here on: a reservation that fails inside agree() would leave the other ranks in its collective
    if ((rc = g->m[0].agree_d.reserve((size_t)(n_ranks + 1) * 8)) || (rc = g->m[0].agree_h.reserve((size_t)(n_ranks + 1) * 8))) { (void)r->CommDestroy(g->m[0].comm); return rc; }
    *out = g.release();
    return ok();
}

// Rank form over the caller's own collectives (TSGPU_XCHG_HOST): no RCCL, no GPU-per-rank requirement. `coll` is copied.
int tsgpu_group_create_rank_host(tsgpu_ctx* ctx, const tsgpu_host_collectives* coll, uint32_t rank, uint32_t n_ranks, tsgpu_group** out) {
    if (!ctx || !coll || !out || n_ranks == 0 || rank >= n_ranks) return fail(TSGPU_ERR_INVALID, "tsgpu_group_create_rank_host: bad arguments");
    *out = nullptr;
    if (!coll->all_gather || !coll->all_to_all) return fail(TSGPU_ERR_INVALID, "tsgpu_group_create_rank_host: both callbacks are required");
    std::unique_ptr<tsgpu_group> g(new (std::nothrow) tsgpu_group);
    if (!g) return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_group_create_rank_host: host allocation failed");
    g->transport = TSGPU_XCHG_HOST; g->local = false; g->n = n_ranks; g->rank = rank;
    g->coll = *coll;
    g->m.resize(1);
    g->m[0].ctx = ctx;
    *out = g.release();
    return ok();
}

void tsgpu_group_destroy(tsgpu_group* g) {
    if (g) for (auto& mem : g->m) for (hipEvent_t& e : mem.ev) if (e) { (void)hipSetDevice(mem.ctx->device); (void)hipEventDestroy(e); e = nullptr; }
    if (!g) return;
    for (auto& mem : g->m) {
        (void)hipSetDevice(mem.ctx->device);
        if (mem.comm && rccl()->CommDestroy) (void)rccl()->CommDestroy(mem.comm);
        DevBuf* cb[] = {&mem.c_pass, &mem.c_mask, &mem.c_found, &mem.c_meta, &mem.c_meta_all, &mem.c_qi, &mem.c_fp, &mem.c_fp_all, &mem.c_tok, &mem.c_tok_all};
        for (auto* x : cb) x->release();
        DevBuf* b[] = {&mem.send, &mem.recv, &mem.l_keys, &mem.l_scores, &mem.l_tm, &mem.l_vd, &mem.l_msi, &mem.l_nh, &mem.l_nm, &mem.l_st, &mem.l_co, &mem.v_dist, &mem.v_lab, &mem.v_cnt, &mem.v_bad,
                       &mem.o_keys, &mem.o_scores, &mem.o_tm, &mem.o_nh, &mem.o_nm, &mem.o_st, &mem.o_vd, &mem.o_lab, &mem.o_cnt, &mem.caps};
        for (auto* x : b) x->release();
        mem.agree_d.release(); mem.h_send.release(); mem.h_recv.release(); mem.agree_h.release();
    }
    delete g;
}

uint32_t tsgpu_group_size(const tsgpu_group* g) { return g ? g->n : 0; }

int tsgpu_group_last_timings(tsgpu_group* g, tsgpu_group_timings* out) {
    if (!g || !out) return fail(TSGPU_ERR_INVALID, "tsgpu_group_last_timings: NULL argument");
    std::lock_guard<std::mutex> lk(g->mu);
    *out = g->tm;
    return ok();
}

// Keyword search of one batch over every shard: out (host, or device memory of member 0 / of this rank) receives the GLOBAL top-k per
// query in Topster order: keys, scores, n_hits and — when the arrays are given — text_match, num_matched (= the sum over the shards),
// status, search_cutoff (0 here: a shard's in-flight cutoff is reported through status 0 + its partial hits, as on one GPU).
// k <= out->k_stride; every shard's Topster holds max(k, the queries' topster_size) entries, its first k travel.
// Exchange (option "kw_exchange_slices", default 1): ncclAllToAll — member j receives, from every member, only the records of the
// queries it merges (a 1/G slice of the batch: (G-1)/G x one block per GPU on the wire instead of (G-1) blocks, and 1/G of the merge
// per GPU) — then the merged slices are replicated with in-place ncclAllGathers (every rank ends with the whole result; the local form
// delivers the slices straight to the caller). 0: ONE ncclAllGather of the blocks, every rank merges everything.
}  // extern "C"

namespace {
// The keyword exchange behind tsgpu_group_keyword_search_batch AND its candidate-combination form: `queries` describes the n_queries result lists (Topster
// capacities, per-hit constants); `local_search(member, loc)` fills the member's local result (device arrays, stride KL). Called with g->mu held.
// `lq` / `n_lq` = the queries the local search runs (the batch itself, or every combination of a candidates call): when the members' dictionaries differ
// (kw_dictionary_fingerprint) the shards first exchange which of those queries' tokens each of them holds, and local_search receives per query the tokens that
// exist on ANOTHER shard (present_elsewhere; nullptr when every shard holds the same terms — the usual case, and then nothing is exchanged).
typedef std::function<int(Member&, tsgpu_hits*, const uint16_t*)> LocalSearch;

// do all members hold the same (field, term) dictionary? Local form: a comparison. (Rank form: the fingerprints ride in the call's FIRST agreement step.)
bool local_dictionaries_equal(tsgpu_group* g) {
    const uint64_t f0 = kw_dictionary_fingerprint(g->m[0].ctx);
    for (auto& mem : g->m) if (kw_dictionary_fingerprint(mem.ctx) != f0) return false;
    return true;
}
// per member and local query: the tokens that some OTHER member holds (Member::h_elsewhere)
int exchange_token_masks(tsgpu_group* g, const tsgpu_kw_query* lq, uint32_t n_lq) {
    const size_t bytes = ((size_t)n_lq * 2 + 15) & ~(size_t)15;
    for (auto& mem : g->m) {
        int rc;
        (void)hipSetDevice(mem.ctx->device);
        if ((rc = mem.c_tok.reserve(bytes)) || (rc = mem.c_tok_all.reserve(bytes * g->n)) || (rc = reserve_host_staging(g, mem, bytes * g->n))) return rc;
        mem.h_tok.assign(bytes / 2, 0);
        if ((rc = kw_terms_present(mem.ctx, lq, n_lq, mem.h_tok.data()))) return rc;
        TSGPU_HIP_TRY(hipMemcpyAsync(mem.c_tok.p, mem.h_tok.data(), bytes, hipMemcpyHostToDevice, mem.ctx->stream));
        TSGPU_HIP_TRY(hipStreamSynchronize(mem.ctx->stream));
    }
    int rc;
    if ((rc = all_gather_everywhere(g, &Member::c_tok, &Member::c_tok_all, bytes))) return rc;
    for (size_t i = 0; i < g->m.size(); i++) {
        Member& mem = g->m[i];
        Member& src = mem;
        (void)hipSetDevice(src.ctx->device);
        std::vector<uint16_t> all(bytes / 2 * g->n);
        TSGPU_HIP_TRY(hipMemcpyAsync(all.data(), src.c_tok_all.p, bytes * g->n, hipMemcpyDeviceToHost, src.ctx->stream));
        TSGPU_HIP_TRY(hipStreamSynchronize(src.ctx->stream));
        const uint32_t me = g->local ? (uint32_t)i : g->rank;
        mem.h_elsewhere.assign(n_lq, 0);
        for (uint32_t r = 0; r < g->n; r++) if (r != me) for (uint32_t q = 0; q < n_lq; q++) mem.h_elsewhere[q] |= all[(size_t)r * (bytes / 2) + q];
    }
    return TSGPU_OK;
}

int group_keyword_core(tsgpu_group* g, const tsgpu_kw_query* queries, uint32_t n_queries, uint32_t k, tsgpu_hits* out, const LocalSearch& local_search, uint64_t sig_tag,
                       const tsgpu_kw_query* lq, uint32_t n_lq) {
    try {
        // 0) the reference drops a token that matches no field (get_field_token_its, src/index.cpp:5651-5655): "no field" means no field of the WHOLE collection.
        //    Shards whose dictionaries differ exchange per query which tokens each of them holds; a token missing here and present elsewhere is an empty list here.
        //    Rank form: a first agreement step carries the arguments' signature AND the rank's dictionary fingerprint (every rank's first collective of a call is an
        //    agreement step, whatever happened to it locally); the second one, after the local phase, carries the local phase's return code.
        bool same_dict = true;
        if (g->local) same_dict = g->n == 1 || n_lq == 0 || local_dictionaries_equal(g);
        else { int rc0 = agree(g, TSGPU_OK, call_signature({sig_tag, n_queries, k, out->k_stride, hits_mask(out), 0xD1C7u}), n_lq ? kw_dictionary_fingerprint(g->m[0].ctx) : 0ull, &same_dict); if (rc0) return rc0; }
        if (!same_dict) { int rc0 = exchange_token_masks(g, lq, n_lq); if (rc0) return rc0; }
        const uint32_t words = out->text_match ? 5 : 4;
        const uint32_t KL = local_topster_stride(queries, n_queries, k);
        const size_t qw = group_kw_record_words(k, words);
        const bool slices = g->kw_slices == 2 || (g->kw_slices && g->n > 1);
        const uint32_t per = slices ? (n_queries + g->n - 1) / g->n : n_queries;      // queries a member merges
        const uint32_t n_pad = slices ? per * g->n : n_queries;
        const size_t KS = out->k_stride;
        const bool pruned = g->kw_pruned && (g->n > 1 || g->kw_pruned_force);
        const uint32_t n_dst = slices ? g->n : 1;                                      // destination slices of a member's exchange block
        const auto t0 = std::chrono::steady_clock::now();
        g->m[0].h_caps.assign(n_pad, 0u);
        group_resolve_topster_sizes(g->m[0].ctx, queries, n_queries, g->m[0].h_caps.data());
        // 1) every member: its shard's Topster (device), packed into its exchange block (one contiguous record per query) — or, bound-pruned
        //    exchange, its kq-th best entry per query (the block is packed after the bounds are known)
        int rc = for_members(g, [&](size_t i) -> int {
            Member& mem = g->m[i];
            (void)hipSetDevice(mem.ctx->device);
            int r;
            const size_t slots = (size_t)n_queries * KL;
            if ((r = mem.l_keys.reserve(slots * 8)) || (r = mem.l_scores.reserve(slots * 24)) || (r = mem.l_tm.reserve(slots * 8)) || (r = mem.l_vd.reserve(slots * 4)) || (r = mem.l_msi.reserve(slots)) || (r = mem.l_nh.reserve((size_t)n_queries * 4)) ||
                (r = mem.l_nm.reserve((size_t)n_queries * 8)) || (r = mem.l_st.reserve((size_t)n_queries * 4)) || (r = mem.l_co.reserve((size_t)n_queries * 4)) ||
                (r = mem.send.reserve((size_t)n_pad * qw * 8)) || (r = mem.recv.reserve((slices ? (size_t)n_pad : (size_t)n_queries * g->n) * qw * 8)) ||
                ((slices || i == 0) && (r = reserve_staging(mem, out, n_pad))) ||
                (r = reserve_host_staging(g, mem, std::max((slices ? (size_t)n_pad : (size_t)n_queries * g->n) * qw * 8, (size_t)n_pad * KS * 24)))) return r;
            if (pruned && ((r = mem.kth_send.reserve((size_t)n_queries * 64)) || (r = mem.kth_recv.reserve((size_t)n_queries * 64 * g->n)) || (r = mem.p_tot.reserve((size_t)n_dst * 4)) || (r = mem.p_totall.reserve((size_t)n_dst * 4 * g->n)) || (r = mem.caps.reserve((size_t)n_pad * 4)) ||
                           (r = reserve_host_staging(g, mem, (size_t)n_queries * 64 * g->n)))) return r;
            tsgpu_hits loc;
            memset(&loc, 0, sizeof loc);
            loc.mem = TSGPU_MEM_DEVICE; loc.k_stride = KL;
            loc.keys = mem.l_keys.as<uint64_t>(); loc.scores = mem.l_scores.as<int64_t>(); loc.text_match = mem.l_tm.as<int64_t>();
            loc.vector_distance = mem.l_vd.as<float>(); loc.match_score_index = mem.l_msi.as<int8_t>();
            loc.n_hits = mem.l_nh.as<uint32_t>(); loc.num_matched = mem.l_nm.as<uint64_t>(); loc.status = mem.l_st.as<int32_t>(); loc.search_cutoff = mem.l_co.as<int32_t>();
            if ((r = local_search(mem, &loc, same_dict ? nullptr : mem.h_elsewhere.data()))) return r;
            mark(g, i, 0);
            if (pruned) {
                TSGPU_HIP_TRY(hipMemcpyAsync(mem.caps.p, g->m[0].h_caps.data(), (size_t)n_pad * 4, hipMemcpyHostToDevice, mem.ctx->stream));      // (h_caps is a member field: it outlives the copy)
                r = group_kw_kth(mem.ctx, &loc, n_queries, k, g->n, mem.caps.as<uint32_t>(), mem.kth_send.as<int64_t>(), mem.ctx->stream);
                mark(g, i, 1);
                return r;
            }
            if (n_pad > n_queries) TSGPU_HIP_TRY(hipMemsetAsync(mem.send.as<uint64_t>() + (size_t)n_queries * qw, 0, (size_t)(n_pad - n_queries) * qw * 8, mem.ctx->stream));   // padding records: no hits
            r = group_pack_keyword(mem.ctx, &loc, n_queries, k, words, mem.send.as<uint64_t>(), mem.ctx->stream);
            mark(g, i, 1);
            return r;
        });
        if ((rc = agree(g, rc, call_signature({sig_tag, n_queries, k, out->k_stride, hits_mask(out), (uint64_t)slices, (uint64_t)g->own_slice_only, (uint64_t)pruned})))) return rc;
        const double t_local = ms_since(t0);
        const auto t1 = std::chrono::steady_clock::now();
        // 2) the exchange, 3) the exact merge, staged per member in arrays of n_pad queries (stride = the caller's k_stride)
        const size_t mergers = slices ? g->m.size() : 1;                                 // (staging arrays: reserved in the local phase)
        size_t slice_words = (size_t)per * qw;                                           // one destination slice of a member's block, in u64 words
        std::vector<uint32_t> tot_all;                                                   // bound-pruned: entries member src sends to destination slice dst
        if (pruned) {
            // 2a) the bounds: every shard's two reported entries per query, everywhere (64 B per query and shard); 2b) each member packs its entries at or
            //     above the bound into n_dst slices (capacity stride: per header pairs + per * k entries), the slices' cursors end as their entry totals;
            //     2c) the totals of every (source, destination) pair reach every rank: the exact transfer sizes
            if ((rc = all_gather_everywhere(g, &Member::kth_send, &Member::kth_recv, (size_t)n_queries * 64))) return rc;
            slice_words = (size_t)per * 2 + (size_t)per * k * words;                     // (<= per * qw: two header words replace three)
            mark(g, 0, 2);
            for (auto& mem : g->m) {
                tsgpu_hits loc;
                memset(&loc, 0, sizeof loc);
                loc.mem = TSGPU_MEM_DEVICE; loc.k_stride = KL; loc.keys = mem.l_keys.as<uint64_t>(); loc.scores = mem.l_scores.as<int64_t>(); loc.text_match = mem.l_tm.as<int64_t>();
                loc.n_hits = mem.l_nh.as<uint32_t>(); loc.num_matched = mem.l_nm.as<uint64_t>(); loc.status = mem.l_st.as<int32_t>();
                if (!g->local && g->test_fail_pack_rank == g->rank + 1) { rc = fail(TSGPU_ERR_DEVICE, "tsgpu_group: injected failure before the sized exchange (test_fail_prune_pack_rank)"); break; }
                if ((rc = group_kw_prune_pack(mem.ctx, &loc, n_queries, n_pad, k, words, mem.caps.as<uint32_t>(), mem.kth_recv.as<int64_t>(), g->n, per, n_dst, slice_words,
                                              mem.send.as<uint64_t>(), mem.p_tot.as<uint32_t>(), mem.ctx->stream))) break;      // (a rank that fails here still joins the totals' collective)
                if (&mem == &g->m[0]) mark(g, 0, 3);
            }
            if ((rc = gather_totals(g, n_dst, tot_all, rc))) return rc;
        }
        // Pruned slices travel at their EXACT sizes where the transport can (RCCL send / recv pairs, device copies: when one shard owns a query's winners it
        // alone sends entries for it); the HOST callbacks and the literal all-gather form move equal-sized pieces: the used prefix of the largest slice.
        const bool exact = pruned && slices && g->transport != TSGPU_XCHG_HOST;
        if (pruned && !exact) {
            uint64_t most = 0;
            for (uint32_t v : tot_all) most = std::max<uint64_t>(most, v);
            const size_t used = (size_t)per * 2 + (size_t)most * words;
            if (slices) { if ((rc = host_all_to_all_prefix(g, slice_words * 8, used * 8))) return rc; }       // HOST: strided prefixes out, contiguous in
            else if ((rc = exchange(g, used * 8))) return rc;                                                // one slice per member: its prefix is contiguous
            slice_words = used;                                                                              // the gathered pieces are `used` words apart
        } else if ((rc = exact ? exchange_slices_exact(g, slice_words, tot_all, per, words) : slices ? exchange_slices(g, slice_words * 8) : exchange(g, slice_words * 8))) return rc;
        for (size_t i = 0; i < mergers; i++) {
            Member& mem = g->m[i];
            (void)hipSetDevice(mem.ctx->device);
            const uint32_t rank = g->local ? (uint32_t)i : g->rank;
            const uint32_t q0 = slices ? rank * per : 0;
            const uint32_t nq = slices ? (q0 < n_queries ? std::min<uint32_t>(per, n_queries - q0) : 0) : n_queries;
            if (!pruned) TSGPU_HIP_TRY(hipMemcpyAsync(mem.caps.p, g->m[0].h_caps.data(), (size_t)n_pad * 4, hipMemcpyHostToDevice, mem.ctx->stream));      // (h_caps is a member field: it outlives the copy)
            tsgpu_hits d = staged_hits(mem, out);
            mark(g, i, 4);
            if ((rc = group_merge_keyword(mem.ctx, mem.recv.as<uint64_t>(), slice_words, g->n, nq, q0, k, words, mem.caps.as<uint32_t>(), &d, mem.ctx->stream, pruned ? per : 0u))) return rc;
            mark(g, i, 5);
        }
        // 4) delivery
        if (slices) { if ((rc = deliver_slices(g, out, n_queries, per, g->kw_slices == 2))) return rc; }
        else {
            Member& root = g->m[0];
            (void)hipSetDevice(root.ctx->device);
            const KwArr arrs[] = {{&Member::o_keys, out->keys, KS * 8}, {&Member::o_scores, out->scores, KS * 24}, {&Member::o_tm, out->text_match, KS * 8},
                                  {&Member::o_nh, out->n_hits, 4}, {&Member::o_nm, out->num_matched, 8}, {&Member::o_st, out->status, 4}};
            for (const KwArr& a : arrs) if (a.dst)
                if ((rc = copy_out(a.dst, (root.*(a.buf)).p, (size_t)n_queries * a.elem, out->mem, root.ctx->stream))) return rc;
        }
        if (out->search_cutoff) { if (out->mem == TSGPU_MEM_HOST) memset(out->search_cutoff, 0, (size_t)n_queries * 4); else TSGPU_HIP_TRY(hipMemsetAsync(out->search_cutoff, 0, (size_t)n_queries * 4, g->m[0].ctx->stream)); }
        fill_keyword_constants(queries, n_queries, out);
        // everything this call enqueued on the members' streams is awaited here
        for (auto& mem : g->m) { (void)hipSetDevice(mem.ctx->device); TSGPU_HIP_TRY(hipStreamSynchronize(mem.ctx->stream)); }
        g->tm.local_ms = (float)t_local; g->tm.exchange_merge_ms = (float)ms_since(t1);
        g->tm.exchange_kernels_ms = marked_ms(g, pruned);
        g->tm.hit_exchange_bytes_per_member = (pruned ? (uint64_t)n_queries * 64 * (g->n - 1) : 0ull) + (uint64_t)slice_words * 8 * (g->n - 1);
        if (exact) {                                     // what the busiest receiver of this process's members took in
            uint64_t worst = 0;
            for (size_t i = 0; i < g->m.size(); i++) {
                const uint32_t d = g->local ? (uint32_t)i : g->rank;
                uint64_t in = 0;
                for (uint32_t j = 0; j < g->n; j++) if (j != d) in += ((uint64_t)per * 2 + (uint64_t)tot_all[(size_t)j * n_dst + d] * words) * 8;
                worst = std::max(worst, in);
            }
            g->tm.hit_exchange_bytes_per_member = (uint64_t)n_queries * 64 * (g->n - 1) + worst;
        }
        // everything one member receives: the hit exchange + (slice form, unless every rank keeps only its own slice) the replication of the merged lists
        g->tm.exchange_bytes_per_member = g->tm.hit_exchange_bytes_per_member +
            ((slices && !(!g->local && g->own_slice_only)) ? (uint64_t)per * (KS * (32 + (out->text_match ? 8 : 0)) + 16) * (g->n - 1) : 0ull);
        return ok();
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_group_keyword_search_batch: host allocation failed"); }
      catch (const std::system_error&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_group_keyword_search_batch: could not start a member thread"); }
}
}  // namespace

extern "C" {

int tsgpu_group_keyword_search_batch(tsgpu_group* g, const tsgpu_kw_query* queries, uint32_t n_queries, uint32_t k, tsgpu_hits* out) {
    if (!g || !out || (n_queries && !queries)) return fail(TSGPU_ERR_INVALID, "tsgpu_group_keyword_search_batch: NULL argument");
    if (n_queries == 0) { std::lock_guard<std::mutex> lk0(g->mu); return agree(g, TSGPU_OK, call_signature({2, 0})); }      // (rank form: a rank called with an empty batch still meets
                                                                                                                             //  the others in the agreement step — they learn of the mismatch instead of waiting for it forever)
    int pre = TSGPU_OK;
    if (k == 0 || k > out->k_stride || k > TSGPU_MAX_TOPK) pre = fail(TSGPU_ERR_INVALID, "tsgpu_group_keyword_search_batch: k must be in 1..min(k_stride, 1024)");
    else if (!out->keys || !out->scores || !out->n_hits) pre = fail(TSGPU_ERR_INVALID, "tsgpu_group_keyword_search_batch: missing output arrays");
    else if ((uint64_t)g->n * k > 4096) pre = fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_group_keyword_search_batch: members * k > 4096");
    std::lock_guard<std::mutex> lk(g->mu);
    if (pre) return agree(g, pre, 0);          // (rank form: the other ranks are told instead of being left in their first collective)
    if (g->replicas) {
        try { return keyword_replicas(g, queries, n_queries, k, out); }
        catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_group_keyword_search_batch: host allocation failed"); }
        catch (const std::system_error&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_group_keyword_search_batch: could not start a member thread"); }
    }
    return group_keyword_core(g, queries, n_queries, k, out, [&](Member& mem, tsgpu_hits* loc, const uint16_t* elsewhere) { return kw_search_batch_masked(mem.ctx, queries, n_queries, loc, elsewhere); }, 2,
                              queries, n_queries);
}

// Wildcard search (q = *) over doc-range shards (Index::search_wildcard, /root/reference/src/index.cpp:6616-6818; the single-GPU form: tsgpu_wildcard_search_batch):
// every member ranks the ids it OWNS — context options doc_range_lo / doc_range_hi, set when the shard is loaded — by the sort keys, the per-shard Topsters take
// the keyword exchange (exact merge; num_matched = ids ranked, added up).
int tsgpu_group_wildcard_search_batch(tsgpu_group* g, const tsgpu_kw_query* queries, uint32_t n_queries, uint32_t k, tsgpu_hits* out) {
    if (!g || !out || (n_queries && !queries)) return fail(TSGPU_ERR_INVALID, "tsgpu_group_wildcard_search_batch: NULL argument");
    if (n_queries == 0) { std::lock_guard<std::mutex> lk0(g->mu); return agree(g, TSGPU_OK, call_signature({7, 0})); }
    int pre = TSGPU_OK;
    if (k == 0 || k > out->k_stride || k > TSGPU_MAX_TOPK) pre = fail(TSGPU_ERR_INVALID, "tsgpu_group_wildcard_search_batch: k must be in 1..min(k_stride, 1024)");
    else if (!out->keys || !out->scores || !out->n_hits) pre = fail(TSGPU_ERR_INVALID, "tsgpu_group_wildcard_search_batch: missing output arrays");
    else if ((uint64_t)g->n * k > 4096) pre = fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_group_wildcard_search_batch: members * k > 4096");
    else if (!g->local && g->own_slice_only) pre = fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_group_wildcard_search_batch: not with kw_own_slice_only");
    else if (g->n > 1 && !g->replicas) for (auto& mem : g->m) if (!mem.ctx->doc_range_set) { pre = fail(TSGPU_ERR_INVALID, "tsgpu_group_wildcard_search_batch: every member needs its doc range (tsgpu_set_option doc_range_lo / doc_range_hi): a shard ranks the ids it owns"); break; }
    std::lock_guard<std::mutex> lk(g->mu);
    if (pre) return agree(g, pre, 0);
    if (g->replicas) {                       // every member mirrors the whole collection: one of them answers
        int rc = agree(g, TSGPU_OK, call_signature({8, n_queries, k, out->k_stride, hits_mask(out)}));
        if (rc) return rc;
        return tsgpu_wildcard_search_batch(g->m[0].ctx, queries, n_queries, out);
    }
    return group_keyword_core(g, queries, n_queries, k, out, [&](Member& mem, tsgpu_hits* loc, const uint16_t*) { return tsgpu_wildcard_search_batch(mem.ctx, queries, n_queries, loc); }, 7, nullptr, 0);
}

// Facet counts (the hash-index branch of Index::do_facets, /root/reference/src/index.cpp:1659-1771; single GPU: tsgpu_facet_count_batch) over doc-range shards:
// every member walks the matched ids through ITS facet mirror (a document it does not hold has no hashes there: it contributes nothing), the per-shard
// (hash, count, doc_id, array_pos) lists — ascending hashes — are gathered and merged per query: counts add up, doc_id / array_pos are those of the greatest
// document that carried the value (the walk's "last one wins", the ids ascend). A hash among the first `cap` of the union is among the first `cap` of every shard
// that has it, so the truncated lists merge exactly; n_values is exact while no shard's list was truncated, else a lower bound (still > cap: "truncated").
// sample_mod > 1 (estimate_facets: the ids at positions i % sample_mod == 0 of the WHOLE list) is a property of the global list: every member gets the whole list.
int tsgpu_group_facet_count_batch(tsgpu_group* g, uint32_t facet_field_id, const uint32_t* const* result_ids, const uint64_t* n_result_ids, uint32_t n_queries,
                                  uint32_t sample_mod, const uint32_t* allowed_hashes, uint32_t n_allowed, tsgpu_facet_counts* out) {
    if (!g || !out || (n_queries && (!result_ids || !n_result_ids))) return fail(TSGPU_ERR_INVALID, "tsgpu_group_facet_count_batch: NULL argument");
    std::lock_guard<std::mutex> lk(g->mu);
    if (n_queries == 0) return agree(g, TSGPU_OK, call_signature({10, 0}));
    int pre = TSGPU_OK;
    if (!out->cap || !out->hash || !out->count || !out->doc_id || !out->array_pos || !out->n_values) pre = fail(TSGPU_ERR_INVALID, "tsgpu_group_facet_count_batch: missing output arrays");
    if (pre) return agree(g, pre, 0);
    try {
        const uint32_t cap = out->cap;
        const size_t nm = g->m.size();
        struct Loc { std::vector<uint32_t> h, c, d, p, nv; };
        std::vector<Loc> loc(nm);
        int rc = for_members(g, [&](size_t i) -> int {
            Member& mem = g->m[i];
            (void)hipSetDevice(mem.ctx->device);
            Loc& L = loc[i];
            L.h.resize((size_t)n_queries * cap); L.c.resize(L.h.size()); L.d.resize(L.h.size()); L.p.resize(L.h.size()); L.nv.assign(n_queries, 0);
            tsgpu_facet_counts fc;
            fc.cap = cap; fc.hash = L.h.data(); fc.count = L.c.data(); fc.doc_id = L.d.data(); fc.array_pos = L.p.data(); fc.n_values = L.nv.data();
            // the slice of every id list inside the member's doc range (they ascend) — unless positions in the whole list matter (sample_mod) or the range is unknown
            std::vector<const uint32_t*> ptr(n_queries);
            std::vector<uint64_t> cnt(n_queries);
            const bool slice = sample_mod <= 1 && mem.ctx->doc_range_set && !g->replicas;
            for (uint32_t q = 0; q < n_queries; q++) {
                const uint32_t* b = result_ids[q];
                const uint32_t* e = b + n_result_ids[q];
                if (slice && b) { b = std::lower_bound(b, e, mem.ctx->doc_range_lo); e = std::lower_bound(b, e, mem.ctx->doc_range_hi); }
                ptr[q] = b; cnt[q] = (uint64_t)(e - b);
            }
            if (!g->local) {                                     // (the first gather's buffers: a failure here is still covered by the agreement step)
                const size_t b1 = ((size_t)n_queries * 4 + 7) & ~(size_t)7;
                int r;
                if ((r = mem.c_meta.reserve(b1)) || (r = mem.c_meta_all.reserve(b1 * g->n))) return r;
            }
            if (g->replicas && i > 0) return TSGPU_OK;           // every member mirrors everything: member 0 answers, the others contribute nothing
            return tsgpu_facet_count_batch(mem.ctx, facet_field_id, ptr.data(), cnt.data(), n_queries, sample_mod, allowed_hashes, n_allowed, &fc);
        });
        if ((rc = agree(g, rc, call_signature({10, n_queries, cap, sample_mod, n_allowed})))) return rc;
        // the lists of every shard, per query: [n][n_queries] counts, then [n][n_queries][stride] entries of 4 words
        std::vector<uint32_t> nv_all((size_t)g->n * n_queries, 0), ent_all;
        size_t stride = 0;
        auto used = [&](uint32_t v) { return std::min<uint32_t>(v, cap); };
        if (g->local) {
            for (size_t i = 0; i < nm; i++) for (uint32_t q = 0; q < n_queries; q++) { nv_all[i * n_queries + q] = loc[i].nv[q]; stride = std::max<size_t>(stride, used(loc[i].nv[q])); }
        } else {
            Member& mem = g->m[0];
            (void)hipSetDevice(mem.ctx->device);
            const size_t b1 = ((size_t)n_queries * 4 + 7) & ~(size_t)7;
            TSGPU_HIP_TRY(hipMemcpyAsync(mem.c_meta.p, loc[0].nv.data(), (size_t)n_queries * 4, hipMemcpyHostToDevice, mem.ctx->stream));
            if ((rc = all_gather_everywhere(g, &Member::c_meta, &Member::c_meta_all, b1))) return rc;
            std::vector<uint32_t> tmp(b1 / 4 * g->n);
            TSGPU_HIP_TRY(hipMemcpyAsync(tmp.data(), mem.c_meta_all.p, b1 * g->n, hipMemcpyDeviceToHost, mem.ctx->stream));
            TSGPU_HIP_TRY(hipStreamSynchronize(mem.ctx->stream));
            for (uint32_t r = 0; r < g->n; r++) for (uint32_t q = 0; q < n_queries; q++) { nv_all[(size_t)r * n_queries + q] = tmp[r * (b1 / 4) + q]; stride = std::max<size_t>(stride, used(tmp[r * (b1 / 4) + q])); }
        }
        stride = std::max<size_t>(stride, 1);
        const size_t per = (size_t)n_queries * stride * 4;      // words of one shard's block
        ent_all.assign(per * g->n, 0);
        auto pack = [&](const Loc& L, uint32_t* dst) {
            for (uint32_t q = 0; q < n_queries; q++)
                for (uint32_t j = 0; j < used(L.nv[q]); j++) {
                    uint32_t* e = dst + ((size_t)q * stride + j) * 4;
                    const size_t at = (size_t)q * cap + j;
                    e[0] = L.h[at]; e[1] = L.c[at]; e[2] = L.d[at]; e[3] = L.p[at];
                }
        };
        if (g->local) { for (size_t i = 0; i < nm; i++) pack(loc[i], ent_all.data() + i * per); }
        else {
            Member& mem = g->m[0];
            std::vector<uint32_t> mine(per, 0);
            pack(loc[0], mine.data());
            if ((rc = mem.c_meta.reserve(per * 4)) || (rc = mem.c_meta_all.reserve(per * 4 * g->n))) return rc;   // (sized by the gathered counts, the same on every rank; a rank that cannot allocate here leaves the others in the gather: out of memory is not survivable anyway)
            TSGPU_HIP_TRY(hipMemcpyAsync(mem.c_meta.p, mine.data(), per * 4, hipMemcpyHostToDevice, mem.ctx->stream));
            if ((rc = all_gather_everywhere(g, &Member::c_meta, &Member::c_meta_all, per * 4))) return rc;
            TSGPU_HIP_TRY(hipMemcpyAsync(ent_all.data(), mem.c_meta_all.p, per * 4 * g->n, hipMemcpyDeviceToHost, mem.ctx->stream));
            TSGPU_HIP_TRY(hipStreamSynchronize(mem.ctx->stream));
        }
        // the merge, query by query: ascending hashes over the shards' cursors
        std::vector<uint32_t> cur(g->n);
        for (uint32_t q = 0; q < n_queries; q++) {
            std::fill(cur.begin(), cur.end(), 0u);
            uint32_t n_out = 0, distinct = 0, most = 0;
            bool truncated = false;
            for (uint32_t r = 0; r < g->n; r++) { const uint32_t v = nv_all[(size_t)r * n_queries + q]; most = std::max(most, v); truncated = truncated || v > cap; }
            for (;;) {
                uint32_t h = 0; bool any = false;
                for (uint32_t r = 0; r < g->n; r++) {
                    if (cur[r] >= used(nv_all[(size_t)r * n_queries + q])) continue;
                    const uint32_t hr = ent_all[r * per + ((size_t)q * stride + cur[r]) * 4];
                    if (!any || hr < h) { h = hr; any = true; }
                }
                if (!any) break;
                uint64_t count = 0; uint32_t doc = 0, pos = 0; bool first = true;
                for (uint32_t r = 0; r < g->n; r++) {
                    if (cur[r] >= used(nv_all[(size_t)r * n_queries + q])) continue;
                    const uint32_t* e = &ent_all[r * per + ((size_t)q * stride + cur[r]) * 4];
                    if (e[0] != h) continue;
                    count += e[1];
                    if (first || e[2] > doc) { doc = e[2]; pos = e[3]; first = false; }
                    cur[r]++;
                }
                distinct++;
                if (n_out < cap) {
                    const size_t at = (size_t)q * cap + n_out;
                    out->hash[at] = h; out->count[at] = (uint32_t)std::min<uint64_t>(count, 0xFFFFFFFFull); out->doc_id[at] = doc; out->array_pos[at] = pos;
                    n_out++;
                }
            }
            out->n_values[q] = truncated ? std::max(std::max(distinct, most), cap + 1) : distinct;
        }
        return ok();
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_group_facet_count_batch: host allocation failed"); }
      catch (const std::system_error&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_group_facet_count_batch: could not start a member thread"); }
}

namespace {
// the part of every matched-id list a member walks: the ids inside its doc range (they ascend), or the whole lists when positions in the whole list matter
// (sample_mod > 1), the member has no doc range, or the members are replicas (then member 0 answers alone: *skip = true for the others)
void member_id_slices(tsgpu_group* g, size_t i, const uint32_t* const* result_ids, const uint64_t* n_result_ids, uint32_t n_queries, uint32_t sample_mod,
                      std::vector<const uint32_t*>& ptr, std::vector<uint64_t>& cnt, bool* skip) {
    Member& mem = g->m[i];
    const bool slice = sample_mod <= 1 && mem.ctx->doc_range_set && !g->replicas;
    ptr.resize(n_queries); cnt.resize(n_queries);
    for (uint32_t q = 0; q < n_queries; q++) {
        const uint32_t* b = result_ids[q];
        const uint32_t* e = b + n_result_ids[q];
        if (slice && b) { b = std::lower_bound(b, e, mem.ctx->doc_range_lo); e = std::lower_bound(b, e, mem.ctx->doc_range_hi); }
        ptr[q] = b; cnt[q] = (uint64_t)(e - b);
    }
    *skip = g->replicas && i > 0;
}
// rank form: `bytes` (a multiple of 8, the same on every rank) of host data from every rank -> all[n][bytes]; local form: the members' blocks side by side
int gather_host_blocks(tsgpu_group* g, const std::vector<std::vector<uint8_t>>& mine, size_t bytes, std::vector<uint8_t>& all) {
    all.assign(bytes * g->n, 0);
    if (g->local) { for (size_t i = 0; i < g->m.size(); i++) memcpy(all.data() + i * bytes, mine[i].data(), bytes); return TSGPU_OK; }
    Member& mem = g->m[0];
    (void)hipSetDevice(mem.ctx->device);
    int rc;
    if ((rc = mem.c_meta.reserve(bytes)) || (rc = mem.c_meta_all.reserve(bytes * g->n))) return rc;
    TSGPU_HIP_TRY(hipMemcpyAsync(mem.c_meta.p, mine[0].data(), bytes, hipMemcpyHostToDevice, mem.ctx->stream));
    if ((rc = all_gather_everywhere(g, &Member::c_meta, &Member::c_meta_all, bytes))) return rc;
    TSGPU_HIP_TRY(hipMemcpyAsync(all.data(), mem.c_meta_all.p, bytes * g->n, hipMemcpyDeviceToHost, mem.ctx->stream));
    TSGPU_HIP_TRY(hipStreamSynchronize(mem.ctx->stream));
    return TSGPU_OK;
}
}  // namespace

// Range facets (tsgpu_facet_range_count_batch; src/index.cpp:1738-1750) over doc-range shards: every member counts the documents of its range, the counts add up
// (0 = the range is in no shard's result_map). The grouped form (group_column) is not sharded: a group's documents live on several shards (501).
int tsgpu_group_facet_range_count_batch(tsgpu_group* g, uint32_t facet_field_id, uint32_t value_column, const int64_t* range_upper, const int64_t* range_lower, uint32_t n_ranges,
                                        const uint32_t* const* result_ids, const uint64_t* n_result_ids, uint32_t n_queries, uint32_t sample_mod, uint32_t* counts) {
    if (!g || (n_queries && (!result_ids || !n_result_ids || !counts)) || (n_ranges && (!range_upper || !range_lower))) return fail(TSGPU_ERR_INVALID, "tsgpu_group_facet_range_count_batch: NULL argument");
    std::lock_guard<std::mutex> lk(g->mu);
    if (n_queries == 0 || n_ranges == 0) return agree(g, TSGPU_OK, call_signature({11, 0}));
    try {
        const size_t words = (size_t)n_queries * n_ranges, bytes = (words * 4 + 7) & ~(size_t)7;
        std::vector<std::vector<uint8_t>> mine(g->m.size(), std::vector<uint8_t>(bytes, 0));
        int rc = for_members(g, [&](size_t i) -> int {
            Member& mem = g->m[i];
            (void)hipSetDevice(mem.ctx->device);
            std::vector<const uint32_t*> ptr; std::vector<uint64_t> cnt; bool skip;
            member_id_slices(g, i, result_ids, n_result_ids, n_queries, sample_mod, ptr, cnt, &skip);
            if (skip) return TSGPU_OK;
            return tsgpu_facet_range_count_batch(mem.ctx, facet_field_id, value_column, range_upper, range_lower, n_ranges, ptr.data(), cnt.data(), n_queries, sample_mod,
                                                 TSGPU_NO_COLUMN, 0, (uint32_t*)mine[i].data());
        });
        if ((rc = agree(g, rc, call_signature({11, n_queries, n_ranges, sample_mod})))) return rc;
        std::vector<uint8_t> all;
        if ((rc = gather_host_blocks(g, mine, bytes, all))) return rc;
        for (size_t w = 0; w < words; w++) {
            uint64_t c = 0;
            for (uint32_t r = 0; r < g->n; r++) c += ((const uint32_t*)(all.data() + r * bytes))[w];
            counts[w] = (uint32_t)std::min<uint64_t>(c, 0xFFFFFFFFull);
        }
        return ok();
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_group_facet_range_count_batch: host allocation failed"); }
      catch (const std::system_error&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_group_facet_range_count_batch: could not start a member thread"); }
}

// Facet stats (tsgpu_facet_stats_batch; compute_facet_stats, src/index.cpp:1430-1460) over doc-range shards: min of the minima, max of the maxima, the counts and the
// sums add up (integer fields: every partial sum is an exact integer in a double while the TOTAL count * max|value| < 2^53 — sum_exact is recomputed from the merged
// values; float fields: the reference's in-order sum up to double rounding, as on one GPU). Shards that saw nothing carry the reference's initial values, which merge away.
int tsgpu_group_facet_stats_batch(tsgpu_group* g, uint32_t facet_field_id, int value_type, const uint32_t* const* result_ids, const uint64_t* n_result_ids, uint32_t n_queries,
                                  uint32_t sample_mod, const uint32_t* int64_map_hashes, const int64_t* int64_map_values, uint32_t n_map, tsgpu_facet_stats* out) {
    if (!g || (n_queries && (!result_ids || !n_result_ids || !out))) return fail(TSGPU_ERR_INVALID, "tsgpu_group_facet_stats_batch: NULL argument");
    std::lock_guard<std::mutex> lk(g->mu);
    if (n_queries == 0) return agree(g, TSGPU_OK, call_signature({12, 0}));
    try {
        const size_t bytes = (size_t)n_queries * sizeof(tsgpu_facet_stats);
        static_assert(sizeof(tsgpu_facet_stats) % 8 == 0, "gathered in 8-byte words");
        std::vector<std::vector<uint8_t>> mine(g->m.size(), std::vector<uint8_t>(bytes, 0));
        // (a member that is skipped — replicas — must still carry the initial values: it runs the call on EMPTY lists)
        int rc = for_members(g, [&](size_t i) -> int {
            Member& mem = g->m[i];
            (void)hipSetDevice(mem.ctx->device);
            std::vector<const uint32_t*> ptr; std::vector<uint64_t> cnt; bool skip;
            member_id_slices(g, i, result_ids, n_result_ids, n_queries, sample_mod, ptr, cnt, &skip);
            if (skip) std::fill(cnt.begin(), cnt.end(), 0ull);
            return tsgpu_facet_stats_batch(mem.ctx, facet_field_id, value_type, ptr.data(), cnt.data(), n_queries, sample_mod, int64_map_hashes, int64_map_values, n_map,
                                           (tsgpu_facet_stats*)mine[i].data());
        });
        if ((rc = agree(g, rc, call_signature({12, n_queries, (uint64_t)value_type, sample_mod, n_map})))) return rc;
        std::vector<uint8_t> all;
        if ((rc = gather_host_blocks(g, mine, bytes, all))) return rc;
        for (uint32_t q = 0; q < n_queries; q++) {
            tsgpu_facet_stats m = ((const tsgpu_facet_stats*)all.data())[q];
            for (uint32_t r = 1; r < g->n; r++) {
                const tsgpu_facet_stats& x = ((const tsgpu_facet_stats*)(all.data() + r * bytes))[q];
                m.fvmin = std::min(m.fvmin, x.fvmin); m.fvmax = std::max(m.fvmax, x.fvmax); m.fvsum += x.fvsum; m.fvcount += x.fvcount;
            }
            if (m.fvcount == 0) m.sum_exact = 1;
            else if (value_type == TSGPU_FACET_FLOAT) m.sum_exact = 0;
            else m.sum_exact = ( (long double)m.fvcount * (long double)std::max(std::fabs(m.fvmin), std::fabs(m.fvmax)) < 9007199254740992.0L) ? 1 : 0;
            m.pad = 0;
            out[q] = m;
        }
        return ok();
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_group_facet_stats_batch: host allocation failed"); }
      catch (const std::system_error&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_group_facet_stats_batch: could not start a member thread"); }
}

// Candidate-token combinations over doc-range shards (Index::search_all_candidates, /root/reference/src/index.cpp:1794-1894; the single-GPU form:
// tsgpu_keyword_search_candidates_batch). The fold of a user query's passes is per KEY (a key met by several passes keeps its greatest KV, the later pass on ties,
// include/topster.h:392-406) and a document lives in ONE shard, so every shard folds its own passes and the shards' folded Topsters are merged like any keyword
// result: same exchange (bound-pruned slices, exact merge), num_matched = the shards' last-pass counts added up, found = their union counts added up. What is NOT
// shard-local is KV::query_index = "the earlier passes that matched ANYTHING" (:5511, :5580-5585): the shards exchange per user query the mask of passes that
// matched on them (one more all-gather of 16 bytes per user query), every hit travels with its pass in the low 4 bits of its key (order-preserving: keys are
// unique), and after the merge query_index = popcount(OR of the masks below the hit's pass).
int tsgpu_group_keyword_search_candidates_batch(tsgpu_group* g, const tsgpu_kw_query* combos, const uint32_t* group_begin, uint32_t n_groups, uint32_t k, tsgpu_hits* out,
                                                uint32_t* query_index, uint64_t* found) {
    if (!g || !out || !group_begin || (n_groups && group_begin[n_groups] && !combos)) return fail(TSGPU_ERR_INVALID, "tsgpu_group_keyword_search_candidates_batch: NULL argument");
    if (n_groups == 0) { std::lock_guard<std::mutex> lk0(g->mu); return agree(g, TSGPU_OK, call_signature({5, 0})); }
    int pre = TSGPU_OK;
    uint32_t max_passes = 0;
    for (uint32_t u = 0; u < n_groups; u++) {
        if (group_begin[u + 1] <= group_begin[u]) { pre = fail(TSGPU_ERR_INVALID, "tsgpu_group_keyword_search_candidates_batch: every user query needs at least one combination"); break; }
        max_passes = std::max(max_passes, group_begin[u + 1] - group_begin[u]);
    }
    if (pre) {}
    else if (k == 0 || k > out->k_stride || k > TSGPU_MAX_TOPK) pre = fail(TSGPU_ERR_INVALID, "tsgpu_group_keyword_search_candidates_batch: k must be in 1..min(k_stride, 1024)");
    else if (!out->keys || !out->scores || !out->n_hits) pre = fail(TSGPU_ERR_INVALID, "tsgpu_group_keyword_search_candidates_batch: missing output arrays");
    else if ((uint64_t)g->n * k > 4096) pre = fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_group_keyword_search_candidates_batch: members * k > 4096");
    else if (max_passes > 16) pre = fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_group_keyword_search_candidates_batch: more than 16 combinations per user query");
    else if (!g->local && g->own_slice_only) pre = fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_group_keyword_search_candidates_batch: not with kw_own_slice_only");
    std::lock_guard<std::mutex> lk(g->mu);
    if (pre) return agree(g, pre, 0);
    if (g->replicas) {
        // every member mirrors the WHOLE collection: any of them answers the call alone (member 0 / this rank; no exchange, the agreement step only)
        int rc = agree(g, TSGPU_OK, call_signature({6, n_groups, k, out->k_stride, hits_mask(out)}));
        if (rc) return rc;
        if ((rc = tsgpu_keyword_search_candidates_batch(g->m[0].ctx, combos, group_begin, n_groups, out, query_index, found))) return rc;
        // (the member returns its whole Topster: the call's k cuts the lists)
        if (out->mem == TSGPU_MEM_HOST) { for (uint32_t u = 0; u < n_groups; u++) out->n_hits[u] = std::min(out->n_hits[u], k); }
        else {
            std::vector<uint32_t> nh(n_groups);
            (void)hipSetDevice(g->m[0].ctx->device);
            TSGPU_HIP_TRY(hipMemcpy(nh.data(), out->n_hits, (size_t)n_groups * 4, hipMemcpyDeviceToHost));
            for (auto& v : nh) v = std::min(v, k);
            TSGPU_HIP_TRY(hipMemcpy(out->n_hits, nh.data(), (size_t)n_groups * 4, hipMemcpyHostToDevice));
        }
        return ok();
    }
    try {
        // the result lists are described by every user query's FIRST combination (the shared Topster is sized once, by the first pass: tsgpu.hip)
        std::vector<tsgpu_kw_query> firsts(n_groups);
        for (uint32_t u = 0; u < n_groups; u++) firsts[u] = combos[group_begin[u]];
        const uint32_t KL = local_topster_stride(firsts.data(), n_groups, k);
        const size_t meta_bytes = (size_t)n_groups * 16;
        int rc = group_keyword_core(g, firsts.data(), n_groups, k, out, [&](Member& mem, tsgpu_hits* loc, const uint16_t* elsewhere) -> int {
            int r;
            if ((r = mem.c_pass.reserve((size_t)n_groups * KL * 4)) || (r = mem.c_mask.reserve((size_t)n_groups * 4)) || (r = mem.c_found.reserve((size_t)n_groups * 8)) ||
                (r = mem.c_meta.reserve(meta_bytes)) || (r = mem.c_meta_all.reserve(meta_bytes * g->n)) || (r = mem.c_qi.reserve((size_t)n_groups * out->k_stride * 4)) ||
                (r = reserve_host_staging(g, mem, meta_bytes * g->n))) return r;
            if ((r = kw_candidates_batch_ex(mem.ctx, combos, group_begin, n_groups, loc, mem.c_pass.as<uint32_t>(), found ? mem.c_found.as<uint64_t>() : nullptr, true, mem.c_mask.as<uint32_t>(), elsewhere))) return r;
            return group_cand_tag(mem.ctx, loc, mem.c_pass.as<uint32_t>(), n_groups, mem.c_mask.as<uint32_t>(), found ? mem.c_found.as<uint64_t>() : nullptr, mem.c_meta.as<uint64_t>(), mem.ctx->stream);
        }, 5, combos, group_begin[n_groups]);
        if (rc) return rc;
        // every shard's (pass mask, union count) per user query, everywhere; then the tags come off the merged keys
        if ((rc = all_gather_everywhere(g, &Member::c_meta, &Member::c_meta_all, meta_bytes))) return rc;
        Member& root = g->m[0];
        (void)hipSetDevice(root.ctx->device);
        hipStream_t s = root.ctx->stream;
        const size_t KS = out->k_stride, slots = (size_t)n_groups * KS;
        if (out->mem == TSGPU_MEM_DEVICE) {
            if ((rc = group_cand_fix(root.ctx, out->keys, query_index, out->n_hits, (uint32_t)KS, 0, n_groups, root.c_meta_all.as<uint64_t>(), g->n, n_groups, found, s))) return rc;
            TSGPU_HIP_TRY(hipStreamSynchronize(s));
        } else {
            // host outputs: the merged keys go up once more, come back untagged with their query_index (n_groups x k_stride words: small next to the exchange)
            if ((rc = root.o_keys.reserve(slots * 8)) || (rc = root.o_nh.reserve((size_t)n_groups * 4)) || (rc = root.c_found.reserve((size_t)n_groups * 8))) return rc;
            TSGPU_HIP_TRY(hipMemcpyAsync(root.o_keys.p, out->keys, slots * 8, hipMemcpyHostToDevice, s));
            TSGPU_HIP_TRY(hipMemcpyAsync(root.o_nh.p, out->n_hits, (size_t)n_groups * 4, hipMemcpyHostToDevice, s));
            if ((rc = group_cand_fix(root.ctx, root.o_keys.as<uint64_t>(), root.c_qi.as<uint32_t>(), root.o_nh.as<uint32_t>(), (uint32_t)KS, 0, n_groups, root.c_meta_all.as<uint64_t>(), g->n, n_groups,
                                     root.c_found.as<uint64_t>(), s))) return rc;
            TSGPU_HIP_TRY(hipMemcpyAsync(out->keys, root.o_keys.p, slots * 8, hipMemcpyDeviceToHost, s));
            if (query_index) TSGPU_HIP_TRY(hipMemcpyAsync(query_index, root.c_qi.p, slots * 4, hipMemcpyDeviceToHost, s));
            if (found) TSGPU_HIP_TRY(hipMemcpyAsync(found, root.c_found.p, (size_t)n_groups * 8, hipMemcpyDeviceToHost, s));
            TSGPU_HIP_TRY(hipStreamSynchronize(s));
        }
        return ok();
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_group_keyword_search_candidates_batch: host allocation failed"); }
}

// group_by over doc-range shards (tsgpu_keyword_search_grouped_batch on one GPU; Topster(capacity, distinct, first_pass), /root/reference/include/topster.h:266-466,
// populate_result_kvs' grouped branch, src/index.cpp:8962-9011). A group's documents live on SEVERAL shards, so a top-k exchange is not enough — the exchange is keyed:
//   round 1  every shard runs the pass as a FIRST pass: its `capacity` best groups, each with its greatest KV (+ the sketch registers of a first pass). A group the whole
//            collection selects is among the best `capacity` of the shard that holds its greatest KV (fewer than `capacity` groups beat it anywhere), so the union of the
//            shards' lists — per distinct key the greatest head — holds the collection's selection: its `capacity` greatest heads, best first, on every rank alike.
//   round 2  every shard runs the pass again with those groups GIVEN (GbShard::forced_keys): slot r = group r on every shard — its member count here and, second pass,
//            its group_limit greatest KVs here. The counts add up (groups_processed), a group's KV lists merge to its group_limit greatest (a document lives in one shard).
// groups_count: the shards' LogLogBeta registers merge by their maxima (the sketch of the union); num_matched adds up. Each shard searches twice (the second pass of the
// reference's own two-pass protocol costs as much); gout->groups_total — the exact number of distinct keys, not a reference quantity — is not computed across shards (501
// unless NULL), ids_out is not offered. out / gout: HOST arrays, as on one GPU.
// `queries` = the candidate combinations; user query i owns queries[cfirst[i] .. cfirst[i + 1]) (the plain call: one each, query_index == nullptr)
static int group_grouped_core(tsgpu_group* g, const tsgpu_kw_query* queries, const uint32_t* cfirst, const tsgpu_group_by* groups, uint32_t n_queries, tsgpu_hits* out, tsgpu_grouped_hits* gout,
                              uint32_t* query_index, bool candidates) {
    if (!g || !out || !gout || !cfirst || (n_queries && (!queries || !groups))) return fail(TSGPU_ERR_INVALID, "tsgpu_group_keyword_search_grouped_batch: NULL argument");
    if (n_queries == 0) { std::lock_guard<std::mutex> lk0(g->mu); return agree(g, TSGPU_OK, call_signature({13, 0})); }
    const uint32_t n_combos = cfirst[n_queries];
    int pre = TSGPU_OK;
    if (cfirst[0] != 0) pre = fail(TSGPU_ERR_INVALID, "tsgpu_group_keyword_search_grouped_candidates_batch: group_begin[0] must be 0");
    for (uint32_t i = 0; i < n_queries && !pre; i++)
        if (cfirst[i + 1] <= cfirst[i] || cfirst[i + 1] - cfirst[i] > 16) pre = fail(TSGPU_ERR_INVALID, "tsgpu_group_keyword_search_grouped_candidates_batch: every user query needs 1..16 combinations");
    if (pre) {} else
    if (out->mem != TSGPU_MEM_HOST) pre = fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_group_keyword_search_grouped_batch: host output arrays only");
    else if (!out->keys || !out->scores || !out->n_hits || !out->status) pre = fail(TSGPU_ERR_INVALID, "tsgpu_group_keyword_search_grouped_batch: keys / scores / n_hits / status are required");
    else if (!gout->n_groups || !gout->distinct_key || !gout->group_size || !gout->group_found || gout->g_stride == 0 || out->k_stride == 0)
        pre = fail(TSGPU_ERR_INVALID, "tsgpu_group_keyword_search_grouped_batch: n_groups / distinct_key / group_size / group_found and the strides are required");
    else if (gout->groups_total && !g->replicas) pre = fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_group_keyword_search_grouped_batch: groups_total (the exact distinct-key count) is not computed across shards: pass NULL");
    else if (g->n > 1 && !g->replicas) {
        bool any_wild = false;
        for (uint32_t i = 0; i < n_queries; i++) any_wild = any_wild || groups[i].wildcard;
        for (auto& mem : g->m) if (any_wild && !mem.ctx->doc_range_set) { pre = fail(TSGPU_ERR_INVALID, "tsgpu_group_keyword_search_grouped_batch: a q = * query needs every member's doc range (tsgpu_set_option doc_range_lo / doc_range_hi): a shard groups the ids it owns"); break; }
    }
    std::lock_guard<std::mutex> lk(g->mu);
    if (pre) return agree(g, pre, 0);
    const uint64_t omask = hits_mask(out) | (gout->groups_count ? 1ull << 20 : 0) | (gout->loglog_registers ? 1ull << 21 : 0) | (gout->groups_total ? 1ull << 22 : 0) | (query_index ? 1ull << 23 : 0) |
                           (candidates ? 1ull << 24 : 0) | ((uint64_t)n_combos << 32);
    if (g->replicas) {
        // every member mirrors the WHOLE collection: member 0 / this rank answers alone
        int rc = agree(g, TSGPU_OK, call_signature({14, n_queries, out->k_stride, gout->g_stride, omask}));
        if (rc) return rc;
        return candidates ? tsgpu_keyword_search_grouped_candidates_batch(g->m[0].ctx, queries, cfirst, groups, n_queries, out, gout, query_index, nullptr)
                          : tsgpu_keyword_search_grouped_batch(g->m[0].ctx, queries, groups, n_queries, out, gout, nullptr);
    }
    try {
        // 0) token existence is a property of the whole collection (group_keyword_core's step 0)
        bool same_dict = true, any_kw = false;
        for (uint32_t i = 0; i < n_queries; i++) any_kw = any_kw || !groups[i].wildcard;
        if (g->local) same_dict = g->n == 1 || !any_kw || local_dictionaries_equal(g);
        else { int rc0 = agree(g, TSGPU_OK, call_signature({13, n_queries, out->k_stride, gout->g_stride, omask}), any_kw ? kw_dictionary_fingerprint(g->m[0].ctx) : 0ull, &same_dict); if (rc0) return rc0; }
        if (!same_dict) { int rc0 = exchange_token_masks(g, queries, n_combos); if (rc0) return rc0; }
        const size_t nm = g->m.size();
        std::vector<uint32_t> caps(n_queries, 0u);
        {
            // ONE collector per user query, sized by its first combination (src/index.cpp:3506-3514)
            std::vector<tsgpu_kw_query> firsts(n_queries);
            for (uint32_t i = 0; i < n_queries; i++) firsts[i] = queries[cfirst[i]];
            group_resolve_topster_sizes(g->m[0].ctx, firsts.data(), n_queries, caps.data());
        }
        uint32_t K1 = 1;
        bool want_regs = false;
        for (uint32_t i = 0; i < n_queries; i++) { K1 = std::max(K1, std::min<uint32_t>(caps[i], TSGPU_MAX_TOPK)); want_regs = want_regs || (groups[i].first_pass && (gout->groups_count || gout->loglog_registers)); }
        std::vector<tsgpu_group_by> g1(groups, groups + n_queries);
        for (auto& x : g1) x.first_pass = 1;
        // ---- round 1: every shard's best groups and their heads ----
        struct Hdr1 { int32_t status, cutoff; uint32_t n_groups, pad; uint64_t num_matched; };
        struct Ent1 { uint64_t dkey, key; int64_t s0, s1, s2; };
        const size_t o_hdr = 0, o_ent = o_hdr + sizeof(Hdr1) * n_queries, o_regs = o_ent + sizeof(Ent1) * n_queries * K1, bytes1 = (o_regs + (want_regs ? (size_t)n_queries * 16384 : 0) + 7) & ~(size_t)7;
        std::vector<std::vector<uint8_t>> mine(nm, std::vector<uint8_t>(bytes1, 0));
        int rc = for_members(g, [&](size_t m) -> int {
            Member& mem = g->m[m];
            (void)hipSetDevice(mem.ctx->device);
            const size_t slots = (size_t)n_queries * K1;
            std::vector<uint64_t> keys(slots), nmv(n_queries), dk(slots);
            std::vector<int64_t> sc(slots * 3);
            std::vector<uint32_t> nh(n_queries), ng(n_queries), gsz(slots), gf(slots);
            std::vector<int32_t> st(n_queries), co(n_queries);
            tsgpu_hits h; memset(&h, 0, sizeof h);
            h.mem = TSGPU_MEM_HOST; h.k_stride = K1; h.keys = keys.data(); h.scores = sc.data(); h.n_hits = nh.data(); h.num_matched = nmv.data(); h.status = st.data(); h.search_cutoff = co.data();
            tsgpu_grouped_hits gh; memset(&gh, 0, sizeof gh);
            gh.g_stride = K1; gh.n_groups = ng.data(); gh.distinct_key = dk.data(); gh.group_size = gsz.data(); gh.group_found = gf.data();
            gh.loglog_registers = want_regs ? mine[m].data() + o_regs : nullptr;
            const GbShard sh = {nullptr, nullptr, same_dict ? nullptr : mem.h_elsewhere.data(), nullptr};
            const int r = gb_shard_batch(mem.ctx, queries, cfirst, g1.data(), n_queries, &h, &gh, nullptr, &sh);
            if (r) return r;
            Hdr1* hd = (Hdr1*)(mine[m].data() + o_hdr);
            Ent1* en = (Ent1*)(mine[m].data() + o_ent);
            for (uint32_t i = 0; i < n_queries; i++) {
                hd[i].status = st[i]; hd[i].cutoff = co[i]; hd[i].n_groups = st[i] == TSGPU_OK ? ng[i] : 0; hd[i].pad = 0; hd[i].num_matched = nmv[i];
                for (uint32_t r2 = 0; r2 < hd[i].n_groups; r2++) {
                    const size_t o = (size_t)i * K1 + r2;
                    en[o].dkey = dk[o]; en[o].key = keys[o]; en[o].s0 = sc[o * 3]; en[o].s1 = sc[o * 3 + 1]; en[o].s2 = sc[o * 3 + 2];
                }
            }
            return TSGPU_OK;
        });
        if ((rc = agree(g, rc, call_signature({13, 1, n_queries, K1, (uint64_t)want_regs})))) return rc;
        std::vector<uint8_t> all1;
        if ((rc = gather_host_blocks(g, mine, bytes1, all1))) return rc;
        const size_t stride1 = bytes1;
        // ---- the collection's selection per query: per distinct key its greatest head, the `capacity` greatest of those, best first (the same on every rank) ----
        auto kv_greater = [](const Ent1& a, const Ent1& b) { if (a.s0 != b.s0) return a.s0 > b.s0; if (a.s1 != b.s1) return a.s1 > b.s1; if (a.s2 != b.s2) return a.s2 > b.s2; return (int64_t)a.key > (int64_t)b.key; };
        std::vector<int32_t> status(n_queries, TSGPU_OK), cutoff(n_queries, 0);
        std::vector<uint64_t> num_matched(n_queries, 0);
        std::vector<uint32_t> fbegin((size_t)n_queries + 1, 0);
        std::vector<uint64_t> fkeys;
        std::vector<Ent1> pool;
        uint32_t K2 = 1, G2 = 1;
        for (uint32_t i = 0; i < n_queries; i++) {
            pool.clear();
            for (uint32_t r = 0; r < g->n; r++) {
                const Hdr1& hd = ((const Hdr1*)(all1.data() + r * stride1 + o_hdr))[i];
                if (hd.status != TSGPU_OK && status[i] == TSGPU_OK) status[i] = hd.status;
                cutoff[i] = cutoff[i] || hd.cutoff; num_matched[i] += hd.num_matched;
                const Ent1* en = (const Ent1*)(all1.data() + r * stride1 + o_ent) + (size_t)i * K1;
                pool.insert(pool.end(), en, en + std::min<uint32_t>(hd.n_groups, K1));
            }
            if (status[i] == TSGPU_OK) {
                std::sort(pool.begin(), pool.end(), [&](const Ent1& a, const Ent1& b) { return a.dkey != b.dkey ? a.dkey < b.dkey : kv_greater(a, b); });
                pool.erase(std::unique(pool.begin(), pool.end(), [](const Ent1& a, const Ent1& b) { return a.dkey == b.dkey; }), pool.end());
                std::sort(pool.begin(), pool.end(), kv_greater);
                // the caller's strides against the CAPACITY, as on one GPU (tsgpu_groupby.inc.h): slot r * group_limit + j of a second pass
                const uint32_t cap = std::min<uint32_t>(caps[i], TSGPU_MAX_TOPK), L = groups[i].first_pass ? 1u : groups[i].group_limit;
                if (cap > gout->g_stride || (uint64_t)cap * L > out->k_stride) status[i] = TSGPU_ERR_INVALID;
                else {
                    const uint32_t n = (uint32_t)std::min<size_t>(pool.size(), cap);
                    for (uint32_t r = 0; r < n; r++) fkeys.push_back(pool[r].dkey);
                    K2 = std::max<uint32_t>(K2, std::max(n, 1u) * L); G2 = std::max(G2, n);        // (a query without groups still passes the shard's capacity x group_limit check: capacity 1)
                }
            }
            fbegin[i + 1] = (uint32_t)fkeys.size();
        }
        // ---- round 2: the given groups on every shard ----
        const size_t slots2 = (size_t)n_queries * K2, gsl2 = (size_t)n_queries * G2;
        const size_t p_st = 0, p_found = p_st + (size_t)n_queries * 16, p_size = p_found + ((gsl2 * 4 + 7) & ~(size_t)7), p_keys = p_size + ((gsl2 * 4 + 7) & ~(size_t)7),
                     p_sc = p_keys + slots2 * 8, p_tm = p_sc + slots2 * 24, p_vd = p_tm + slots2 * 8, p_msi = p_vd + ((slots2 * 4 + 7) & ~(size_t)7), p_qx = p_msi + ((slots2 + 7) & ~(size_t)7), bytes2 = p_qx + ((slots2 * 4 + 7) & ~(size_t)7);
        // (every rank holds every shard's round-2 block: capacity x group_limit x 45 bytes per query and shard — a bound instead of an allocation failure half-way;
        //  every rank derives the same sizes from the same gathered round-1 data: they all leave here together)
        if ((uint64_t)bytes2 * g->n > (4ull << 30)) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_group_keyword_search_grouped_batch: the gathered group lists of this batch exceed 4 GiB; split it");
        for (auto& b : mine) { b.assign(bytes2, 0); }
        rc = for_members(g, [&](size_t m) -> int {
            Member& mem = g->m[m];
            (void)hipSetDevice(mem.ctx->device);
            uint8_t* blk = mine[m].data();
            std::vector<uint64_t> nmv(n_queries), dk(gsl2);
            std::vector<uint32_t> nh(n_queries), ng(n_queries);
            std::vector<int32_t> st(n_queries), co(n_queries);
            tsgpu_hits h; memset(&h, 0, sizeof h);
            h.mem = TSGPU_MEM_HOST; h.k_stride = K2; h.keys = (uint64_t*)(blk + p_keys); h.scores = (int64_t*)(blk + p_sc); h.text_match = (int64_t*)(blk + p_tm); h.vector_distance = (float*)(blk + p_vd);
            h.match_score_index = (int8_t*)(blk + p_msi); h.n_hits = nh.data(); h.num_matched = nmv.data(); h.status = st.data(); h.search_cutoff = co.data();
            tsgpu_grouped_hits gh; memset(&gh, 0, sizeof gh);
            gh.g_stride = G2; gh.n_groups = ng.data(); gh.distinct_key = dk.data(); gh.group_size = (uint32_t*)(blk + p_size); gh.group_found = (uint32_t*)(blk + p_found);
            std::vector<uint32_t> pmask(n_queries, 0u);
            const GbShard sh = {fkeys.data(), fbegin.data(), same_dict ? nullptr : mem.h_elsewhere.data(), pmask.data()};
            const int r = gb_shard_batch(mem.ctx, queries, cfirst, groups, n_queries, &h, &gh, (uint32_t*)(blk + p_qx), &sh);      // (query_index = the hits' PASSES)
            if (r) return r;
            int32_t* pst = (int32_t*)(blk + p_st);
            for (uint32_t i = 0; i < n_queries; i++) {
                pst[4 * i] = st[i]; pst[4 * i + 1] = co[i]; pst[4 * i + 2] = (int32_t)pmask[i]; pst[4 * i + 3] = 0;
                if (st[i] != TSGPU_OK) for (uint32_t r2 = 0; r2 < G2; r2++) ((uint32_t*)(blk + p_size))[(size_t)i * G2 + r2] = ((uint32_t*)(blk + p_found))[(size_t)i * G2 + r2] = 0;
            }
            return TSGPU_OK;
        });
        if ((rc = agree(g, rc, call_signature({13, 2, n_queries, K2, G2, fkeys.size()})))) return rc;
        std::vector<uint8_t> all2;
        if ((rc = gather_host_blocks(g, mine, bytes2, all2))) return rc;
        // ---- merge: counts add up, a group's KV lists merge to its group_limit greatest ----
        struct KV { int64_t s0, s1, s2; uint64_t key; int64_t tm; float vd; int8_t msi; uint32_t pass; };
        auto kv2_greater = [](const KV& a, const KV& b) { if (a.s0 != b.s0) return a.s0 > b.s0; if (a.s1 != b.s1) return a.s1 > b.s1; if (a.s2 != b.s2) return a.s2 > b.s2; return (int64_t)a.key > (int64_t)b.key; };
        std::vector<KV> kvs;
        std::vector<uint8_t> regs(16384);
        for (uint32_t i = 0; i < n_queries; i++) {
            for (uint32_t r = 0; r < g->n && status[i] == TSGPU_OK; r++) {
                const int32_t* pst = (const int32_t*)(all2.data() + r * bytes2 + p_st);
                if (pst[4 * i] != TSGPU_OK) status[i] = pst[4 * i];
                cutoff[i] = cutoff[i] || pst[4 * i + 1];
            }
            // KV::query_index = the earlier passes of the user query that matched ANYTHING (src/index.cpp:5511, :5580-5585) — anywhere: the OR of the shards' pass masks
            uint32_t gmask = 0;
            for (uint32_t r = 0; r < g->n; r++) gmask |= (uint32_t)((const int32_t*)(all2.data() + r * bytes2 + p_st))[4 * i + 2];
            out->status[i] = status[i];
            if (out->search_cutoff) out->search_cutoff[i] = cutoff[i];
            if (out->num_matched) out->num_matched[i] = status[i] == TSGPU_OK ? num_matched[i] : 0;
            if (gout->groups_count) gout->groups_count[i] = 0;
            if (gout->loglog_registers) memset(gout->loglog_registers + (size_t)i * 16384, 0, 16384);
            if (status[i] != TSGPU_OK) { out->n_hits[i] = 0; gout->n_groups[i] = 0; continue; }
            const uint32_t n = fbegin[i + 1] - fbegin[i], L = groups[i].first_pass ? 1u : groups[i].group_limit;
            uint32_t hits = 0;
            for (uint32_t gr = 0; gr < n; gr++) {
                uint64_t found = 0;
                kvs.clear();
                for (uint32_t r = 0; r < g->n; r++) {
                    const uint8_t* blk = all2.data() + r * bytes2;
                    found += ((const uint32_t*)(blk + p_found))[(size_t)i * G2 + gr];
                    const uint32_t sz = std::min(((const uint32_t*)(blk + p_size))[(size_t)i * G2 + gr], L);
                    for (uint32_t j = 0; j < sz; j++) {
                        const size_t o = (size_t)i * K2 + (size_t)gr * L + j;
                        KV kv;
                        kv.key = ((const uint64_t*)(blk + p_keys))[o];
                        const int64_t* sc = (const int64_t*)(blk + p_sc) + o * 3;
                        kv.s0 = sc[0]; kv.s1 = sc[1]; kv.s2 = sc[2];
                        kv.tm = ((const int64_t*)(blk + p_tm))[o]; kv.vd = ((const float*)(blk + p_vd))[o]; kv.msi = ((const int8_t*)(blk + p_msi))[o]; kv.pass = ((const uint32_t*)(blk + p_qx))[o];
                        kvs.push_back(kv);
                    }
                }
                std::sort(kvs.begin(), kvs.end(), kv2_greater);
                const uint32_t take = (uint32_t)std::min<size_t>(kvs.size(), L);
                const size_t go = (size_t)i * gout->g_stride + gr;
                gout->distinct_key[go] = fkeys[fbegin[i] + gr];
                gout->group_found[go] = (uint32_t)std::min<uint64_t>(found, 0xFFFFFFFFull);
                gout->group_size[go] = take;
                for (uint32_t j = 0; j < take; j++) {
                    const size_t o = (size_t)i * out->k_stride + (size_t)gr * L + j;
                    out->keys[o] = kvs[j].key; out->scores[o * 3] = kvs[j].s0; out->scores[o * 3 + 1] = kvs[j].s1; out->scores[o * 3 + 2] = kvs[j].s2;
                    if (out->text_match) out->text_match[o] = kvs[j].tm;
                    if (out->vector_distance) out->vector_distance[o] = kvs[j].vd;
                    if (out->match_score_index) out->match_score_index[o] = kvs[j].msi;
                    if (query_index) query_index[o] = (uint32_t)__builtin_popcount(gmask & ((1u << (kvs[j].pass & 31u)) - 1u));
                }
                hits += take;
            }
            gout->n_groups[i] = n;
            out->n_hits[i] = groups[i].first_pass ? n : hits;
            if (groups[i].first_pass && want_regs) {
                // the sketch of the union: the registers' maxima (LogLogBeta::merge), then cardinality() over them
                std::fill(regs.begin(), regs.end(), 0);
                for (uint32_t r = 0; r < g->n; r++) {
                    const uint8_t* rr = all1.data() + r * stride1 + o_regs + (size_t)i * 16384;
                    for (uint32_t x = 0; x < 16384; x++) regs[x] = std::max(regs[x], rr[x]);
                }
                if (gout->groups_count) gout->groups_count[i] = gb_registers_cardinality(regs.data());
                if (gout->loglog_registers) memcpy(gout->loglog_registers + (size_t)i * 16384, regs.data(), 16384);
            }
        }
        return ok();
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_group_keyword_search_grouped_batch: host allocation failed"); }
      catch (const std::system_error&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_group_keyword_search_grouped_batch: could not start a member thread"); }
}

int tsgpu_group_keyword_search_grouped_batch(tsgpu_group* g, const tsgpu_kw_query* queries, const tsgpu_group_by* groups, uint32_t n_queries, tsgpu_hits* out, tsgpu_grouped_hits* gout) {
    try {
        std::vector<uint32_t> cf((size_t)n_queries + 1);
        for (uint32_t i = 0; i <= n_queries; i++) cf[i] = i;         // one combination per query
        return group_grouped_core(g, queries, cf.data(), groups, n_queries, out, gout, nullptr, false);
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_group_keyword_search_grouped_batch: host allocation failed"); }
}

// The same over candidate-token combinations (tsgpu_keyword_search_grouped_candidates_batch; Index::search_all_candidates with group_limit != 0, src/index.cpp:1794-1894): the
// fold of a user query's passes is per document (a second pass counts a document once, with its greatest KV) and per group (a first pass keeps a group's greatest KV over all
// passes), and a document lives in ONE shard — so both rounds run the shards' own folds. What is not shard-local is KV::query_index (the earlier passes that matched
// ANYTHING): every hit travels with its pass, the shards' pass masks are OR'ed, query_index = the matching passes below the hit's. num_matched = the LAST pass' counts added up.
int tsgpu_group_keyword_search_grouped_candidates_batch(tsgpu_group* g, const tsgpu_kw_query* combos, const uint32_t* group_begin, const tsgpu_group_by* groups, uint32_t n_user,
                                                        tsgpu_hits* out, tsgpu_grouped_hits* gout, uint32_t* query_index) {
    return group_grouped_core(g, combos, group_begin, groups, n_user, out, gout, query_index, true);
}

int tsgpu_group_set_option(tsgpu_group* g, const char* name, int64_t value) {
    if (!g || !name) return fail(TSGPU_ERR_INVALID, "tsgpu_group_set_option: NULL argument");
    std::lock_guard<std::mutex> lk(g->mu);
    if (!strcmp(name, "kw_exchange_pruned")) { g->kw_pruned = value != 0; g->kw_pruned_force = value == 2; return ok(); }
    if (!strcmp(name, "kw_exchange_slices")) { g->kw_slices = value == 2 ? 2 : (value != 0); return ok(); }
    if (!strcmp(name, "replicas")) { g->replicas = value != 0; return ok(); }
    if (!strcmp(name, "kw_own_slice_only")) { g->own_slice_only = value != 0; return ok(); }
    if (!strcmp(name, "test_fail_prune_pack_rank")) { g->test_fail_pack_rank = (uint32_t)std::max<int64_t>(value, 0); return ok(); }
    return fail(TSGPU_ERR_NOT_FOUND, std::string("tsgpu_group_set_option: unknown option ") + name);
}

// Exact k-NN of one batch over every shard (tsgpu_vec_knn_batch per member, then the exchange): closest first, ties -> smaller label.
// Q: host memory, or device memory readable by every owned member (rank form / members sharing a device). allow_ids / excluded_ids
// (sorted, GLOBAL seq_ids, host) restrict the whole batch like the VectorFilterFunctor. Labels must fit 32 bits (seq_ids do).
int tsgpu_group_vec_knn_batch(tsgpu_group* g, uint32_t vec_field_id, const float* Q, int mem_q, uint32_t n_queries, uint32_t k,
                              const uint32_t* allow_ids, uint32_t n_allow, const uint32_t* excluded_ids, uint32_t n_excluded,
                              float* dist_out, uint64_t* label_out, uint32_t* n_out, int mem_out) {
    if (!g || !Q || !dist_out || !label_out || !n_out) return fail(TSGPU_ERR_INVALID, "tsgpu_group_vec_knn_batch: NULL argument");
    if (n_queries == 0) { std::lock_guard<std::mutex> lk0(g->mu); return agree(g, TSGPU_OK, call_signature({4, 0})); }
    int pre = TSGPU_OK;
    if (k == 0 || k > TSGPU_MAX_TOPK) pre = fail(TSGPU_ERR_INVALID, "tsgpu_group_vec_knn_batch: k must be in 1..1024");
    else if ((uint64_t)g->n * k > 8192) pre = fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_group_vec_knn_batch: members * k > 8192");
    std::lock_guard<std::mutex> lk(g->mu);
    if (pre) return agree(g, pre, 0);
    try {
        if (g->replicas) {
            // every member mirrors the whole matrix: member i scans for queries [i * per, (i + 1) * per), the slices are replicated / delivered
            uint32_t dim = 0;
            int rc = group_vec_dim(g->m[0].ctx, vec_field_id, &dim);
            if (rc) return agree(g, rc, 0);
            const uint32_t per = (n_queries + g->n - 1) / g->n, n_pad = per * g->n;
            const auto t0 = std::chrono::steady_clock::now();
            rc = for_members(g, [&](size_t i) -> int {
                Member& mem = g->m[i];
                (void)hipSetDevice(mem.ctx->device);
                const uint32_t rank = g->local ? (uint32_t)i : g->rank;
                const uint32_t q0 = rank * per, nq = q0 < n_queries ? std::min<uint32_t>(per, n_queries - q0) : 0;
                int r;
                if ((r = mem.o_vd.reserve((size_t)n_pad * k * 4)) || (r = mem.o_lab.reserve((size_t)n_pad * k * 8)) || (r = mem.o_cnt.reserve((size_t)n_pad * 4)) ||
                    (r = reserve_host_staging(g, mem, (size_t)n_pad * k * 8))) return r;
                TSGPU_HIP_TRY(hipMemsetAsync(mem.o_cnt.p, 0, (size_t)n_pad * 4, mem.ctx->stream));
                TSGPU_HIP_TRY(hipStreamSynchronize(mem.ctx->stream));
                if (nq == 0) return TSGPU_OK;
                return tsgpu_vec_knn_batch(mem.ctx, vec_field_id, Q + (size_t)q0 * dim, mem_q, nq, k, allow_ids, n_allow, excluded_ids, n_excluded,
                                           mem.o_vd.as<float>() + (size_t)q0 * k, mem.o_lab.as<uint64_t>() + (size_t)q0 * k, mem.o_cnt.as<uint32_t>() + q0, TSGPU_MEM_DEVICE);
            });
            if ((rc = agree(g, rc, call_signature({3, n_queries, k, (uint64_t)mem_out, n_allow, n_excluded})))) return rc;
            const double t_local = ms_since(t0);
            const auto t1 = std::chrono::steady_clock::now();
            const KwArr arrs[] = {{&Member::o_vd, dist_out, (size_t)k * 4}, {&Member::o_lab, label_out, (size_t)k * 8}, {&Member::o_cnt, n_out, 4}};
            const bool to_host = mem_out == TSGPU_MEM_HOST;
            if (g->n > 1 && !(g->local && to_host)) {
                const bool grouped = g->transport == TSGPU_XCHG_RCCL && g->m.size() > 1;
                if (grouped) { int r2 = rccl()->GroupStart(); if (r2) return rccl_fail("ncclGroupStart", r2); }
                for (const KwArr& a : arrs) {
                    std::vector<void*> pm;
                    for (auto& mem : g->m) pm.push_back((mem.*(a.buf)).p);
                    if ((rc = replicate_slices(g, pm, (size_t)per * a.elem))) { if (grouped) (void)rccl()->GroupEnd(); return rc; }
                }
                if (grouped) { int r2 = rccl()->GroupEnd(); if (r2) return rccl_fail("ncclGroupEnd", r2); }
            }
            if (g->n > 1 && g->local && to_host) {
                for (size_t i = 0; i < g->m.size(); i++) {
                    Member& mem = g->m[i];
                    (void)hipSetDevice(mem.ctx->device);
                    const uint32_t q0 = (uint32_t)i * per;
                    if (q0 >= n_queries) break;
                    const uint32_t nq = std::min<uint32_t>(per, n_queries - q0);
                    for (const KwArr& a : arrs) if ((rc = copy_out((char*)a.dst + (size_t)q0 * a.elem, (const char*)(mem.*(a.buf)).p + (size_t)q0 * a.elem, (size_t)nq * a.elem, TSGPU_MEM_HOST, mem.ctx->stream))) return rc;
                }
            } else {
                Member& root = g->m[0];
                (void)hipSetDevice(root.ctx->device);
                for (const KwArr& a : arrs) if ((rc = copy_out(a.dst, (root.*(a.buf)).p, (size_t)n_queries * a.elem, mem_out, root.ctx->stream))) return rc;
            }
            for (auto& mem : g->m) { (void)hipSetDevice(mem.ctx->device); TSGPU_HIP_TRY(hipStreamSynchronize(mem.ctx->stream)); }
            g->tm.local_ms = (float)t_local; g->tm.exchange_merge_ms = (float)ms_since(t1); g->tm.exchange_bytes_per_member = (uint64_t)per * ((size_t)k * 12 + 4) * (g->n - 1); g->tm.hit_exchange_bytes_per_member = 0;
            return ok();
        }
        const size_t block_words = (size_t)n_queries * k;
        const auto t0 = std::chrono::steady_clock::now();
        int rc = for_members(g, [&](size_t i) -> int {
            Member& mem = g->m[i];
            (void)hipSetDevice(mem.ctx->device);
            int r;
            if ((r = mem.v_dist.reserve(block_words * 4)) || (r = mem.v_lab.reserve(block_words * 8)) || (r = mem.v_cnt.reserve((size_t)n_queries * 4)) || (r = mem.v_bad.reserve(64)) ||
                (r = mem.send.reserve(block_words * 8)) || (r = mem.recv.reserve(block_words * 8 * g->n)) || (r = reserve_host_staging(g, mem, block_words * 8 * g->n)) ||
                (i == 0 && mem_out == TSGPU_MEM_HOST && ((r = mem.o_vd.reserve(block_words * 4)) || (r = mem.o_lab.reserve(block_words * 8)) || (r = mem.o_cnt.reserve((size_t)n_queries * 4))))) return r;
            if ((r = tsgpu_vec_knn_batch(mem.ctx, vec_field_id, Q, mem_q, n_queries, k, allow_ids, n_allow, excluded_ids, n_excluded,
                                         mem.v_dist.as<float>(), mem.v_lab.as<uint64_t>(), mem.v_cnt.as<uint32_t>(), TSGPU_MEM_DEVICE))) return r;
            TSGPU_HIP_TRY(hipMemsetAsync(mem.v_bad.p, 0, 4, mem.ctx->stream));
            return group_pack_knn(mem.ctx, mem.v_dist.as<float>(), mem.v_lab.as<uint64_t>(), mem.v_cnt.as<uint32_t>(), n_queries, k, mem.send.as<uint64_t>(), mem.v_bad.as<uint32_t>(), mem.ctx->stream);
        });
        if ((rc = agree(g, rc, call_signature({4, n_queries, k, (uint64_t)mem_out, n_allow, n_excluded})))) return rc;
        const double t_local = ms_since(t0);
        const auto t1 = std::chrono::steady_clock::now();
        if ((rc = exchange(g, block_words * 8))) return rc;
        Member& root = g->m[0];
        (void)hipSetDevice(root.ctx->device);
        hipStream_t s = root.ctx->stream;
        float* d_dist = dist_out; uint64_t* d_lab = label_out; uint32_t* d_cnt = n_out;
        if (mem_out == TSGPU_MEM_HOST) {
            if ((rc = root.o_vd.reserve(block_words * 4)) || (rc = root.o_lab.reserve(block_words * 8)) || (rc = root.o_cnt.reserve((size_t)n_queries * 4))) return rc;
            d_dist = root.o_vd.as<float>(); d_lab = root.o_lab.as<uint64_t>(); d_cnt = root.o_cnt.as<uint32_t>();
        }
        if ((rc = group_merge_knn(root.ctx, root.recv.as<uint64_t>(), block_words, g->n, n_queries, k, d_dist, d_lab, d_cnt, s))) return rc;
        if (mem_out == TSGPU_MEM_HOST) {
            if ((rc = copy_out(dist_out, d_dist, block_words * 4, TSGPU_MEM_HOST, s)) || (rc = copy_out(label_out, d_lab, block_words * 8, TSGPU_MEM_HOST, s)) ||
                (rc = copy_out(n_out, d_cnt, (size_t)n_queries * 4, TSGPU_MEM_HOST, s))) return rc;
        }
        uint32_t bad = 0;
        for (auto& mem : g->m) {
            (void)hipSetDevice(mem.ctx->device);
            uint32_t b = 0;
            TSGPU_HIP_TRY(hipMemcpyAsync(&b, mem.v_bad.p, 4, hipMemcpyDeviceToHost, mem.ctx->stream));
            TSGPU_HIP_TRY(hipStreamSynchronize(mem.ctx->stream));
            bad += b;
        }
        if (bad) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_group_vec_knn_batch: a label beyond 32 bits (the exchange carries seq_ids)");
        g->tm.local_ms = (float)t_local; g->tm.exchange_merge_ms = (float)ms_since(t1); g->tm.exchange_bytes_per_member = block_words * 8 * (g->n - 1); g->tm.hit_exchange_bytes_per_member = g->tm.exchange_bytes_per_member;
        return ok();
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_group_vec_knn_batch: host allocation failed"); }
      catch (const std::system_error&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_group_vec_knn_batch: could not start a member thread"); }
}

// Hybrid search over the shards: the keyword Topsters (capacity = out->k_stride, with text_match) and the k nearest vectors are each
// gathered and merged, THEN fused exactly as src/index.cpp:4036-4221 (tsgpu_hybrid_fuse_batch on member 0 / this rank): reciprocal
// ranks are ranks in the GLOBAL lists. Host outputs, host or (see above) device queries; rerank_hybrid_matches: the shard that owns a hit supplies its missing score.
int tsgpu_group_hybrid_search_batch(tsgpu_group* g, const tsgpu_kw_query* queries, uint32_t vec_field_id, int metric, const tsgpu_hybrid_params* p,
                                    const float* Q, int mem_q, uint32_t dim, uint32_t n_queries, tsgpu_hits* out) {
    if (!g || !queries || !p || !Q || !out) return fail(TSGPU_ERR_INVALID, "tsgpu_group_hybrid_search_batch: NULL argument");
    if (out->mem != TSGPU_MEM_HOST) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_group_hybrid_search_batch: host outputs only");
    if (p->rerank_hybrid_matches && (!out->text_match || !out->vector_distance || !out->match_score_index))
        return fail(TSGPU_ERR_INVALID, "tsgpu_group_hybrid_search_batch: rerank_hybrid_matches needs the text_match, vector_distance and match_score_index outputs");
    if (n_queries == 0) return tsgpu_group_keyword_search_batch(g, queries, 0, 1, out);      // (the agreement step of the keyword half: a rank with an empty batch still meets the others)
    const uint32_t k = p->k == 0 ? std::max<uint32_t>(p->fetch_size, 100) : p->k;                 // src/index.cpp:4060-4063
    try {
        const uint32_t KS = out->k_stride;
        std::vector<uint64_t> keys((size_t)n_queries * KS), nm(n_queries);
        std::vector<int64_t> scores((size_t)n_queries * KS * 3), tm((size_t)n_queries * KS);
        std::vector<float> vd((size_t)n_queries * KS);
        std::vector<int8_t> msi((size_t)n_queries * KS);
        std::vector<uint32_t> nh(n_queries);
        std::vector<int32_t> st(n_queries), co(n_queries);
        tsgpu_hits kw;
        kw.mem = TSGPU_MEM_HOST; kw.k_stride = KS;
        kw.keys = keys.data(); kw.scores = scores.data(); kw.text_match = tm.data(); kw.vector_distance = vd.data();
        kw.match_score_index = msi.data(); kw.n_hits = nh.data(); kw.num_matched = nm.data(); kw.status = st.data(); kw.search_cutoff = co.data();
        int rc = tsgpu_group_keyword_search_batch(g, queries, n_queries, std::min<uint32_t>(KS, TSGPU_MAX_TOPK), &kw);
        if (rc) return rc;
        std::vector<float> kd((size_t)n_queries * k);
        std::vector<uint64_t> kl((size_t)n_queries * k);
        std::vector<uint32_t> kc(n_queries);
        if ((rc = tsgpu_group_vec_knn_batch(g, vec_field_id, Q, mem_q, n_queries, k, nullptr, 0, nullptr, 0, kd.data(), kl.data(), kc.data(), TSGPU_MEM_HOST))) return rc;
        for (uint32_t q = 0; q < n_queries; q++) {          // filter_by / hidden hits: that query's own exact k-NN over its allowed ids (VectorFilterFunctor)
            if (st[q] != TSGPU_OK || (queries[q].n_excluded == 0 && queries[q].n_filter == 0)) continue;
            if (mem_q != TSGPU_MEM_HOST) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_group_hybrid_search_batch: filtered queries need host-resident query vectors");
            if ((rc = tsgpu_group_vec_knn_batch(g, vec_field_id, Q + (size_t)q * dim, mem_q, 1, k, queries[q].n_filter ? queries[q].filter_ids : nullptr, queries[q].n_filter,
                                                queries[q].n_excluded ? queries[q].excluded_ids : nullptr, queries[q].n_excluded, kd.data() + (size_t)q * k, kl.data() + (size_t)q * k, kc.data() + q, TSGPU_MEM_HOST))) return rc;
        }
        tsgpu_hybrid_params p0 = *p;
        p0.rerank_hybrid_matches = 0;
        if ((rc = tsgpu_hybrid_fuse_batch(g->m[0].ctx, queries, &p0, metric, &kw, kd.data(), kl.data(), kc.data(), k, n_queries, out)) || !p->rerank_hybrid_matches) return rc;
        // rerank_hybrid_matches (Index::compute_aux_scores, src/index.cpp:8793-8923) over the shards: the fused lists are the same on every rank, so every rank
        // derives the same items — hits found by one side only. The shard that OWNS a document knows the missing score: every member scores every item on its
        // shard (text_match of a document it does not hold: 0; distance of a label it does not hold: NaN), the answers are gathered, the owner's is taken
        // (max of the scores — they are not negative —, the one distance that is a number), and every rank re-fuses.
        std::vector<uint32_t> item_q, item_id, pair_q;
        std::vector<uint64_t> pair_label;
        std::vector<size_t> item_slot, pair_slot;
        hybrid_missing_items(n_queries, out, item_q, item_id, item_slot, pair_q, pair_label, pair_slot);
        const size_t ni = item_q.size(), np_ = pair_q.size(), words = ni + (np_ + 1) / 2;      // one u64 per score, two distances per u64
        std::lock_guard<std::mutex> lk(g->mu);
        std::vector<std::vector<uint64_t>> mine(g->m.size(), std::vector<uint64_t>(std::max<size_t>(words, 1), 0));
        rc = TSGPU_OK;
        for (size_t i = 0; i < g->m.size() && !rc; i++) {
            Member& mem = g->m[i];
            (void)hipSetDevice(mem.ctx->device);
            std::vector<int64_t> sc(ni);
            std::vector<float> d(np_);
            if (ni) rc = tsgpu_keyword_aux_scores(mem.ctx, queries, n_queries, item_q.data(), item_id.data(), (uint32_t)ni, sc.data());
            if (!rc && np_) rc = hybrid_missing_distances(mem.ctx, vec_field_id, Q, mem_q, n_queries, pair_q.data(), pair_label.data(), (uint32_t)np_, d.data());
            if (rc) break;
            memcpy(mine[i].data(), sc.data(), ni * 8);
            memcpy(mine[i].data() + ni, d.data(), np_ * 4);
            if ((rc = mem.c_meta.reserve(std::max<size_t>(words, 1) * 8)) || (rc = mem.c_meta_all.reserve(std::max<size_t>(words, 1) * 8 * g->n))) break;
        }
        if ((rc = agree(g, rc, call_signature({9, n_queries, (uint64_t)ni, (uint64_t)np_})))) return rc;      // (a rank whose shard failed takes the others with it BEFORE the gather)
        std::vector<uint64_t> all((size_t)g->n * std::max<size_t>(words, 1), 0);
        if (g->local) { for (size_t i = 0; i < g->m.size(); i++) memcpy(all.data() + i * std::max<size_t>(words, 1), mine[i].data(), std::max<size_t>(words, 1) * 8); }
        else if (words) {
            Member& mem = g->m[0];
            (void)hipSetDevice(mem.ctx->device);
            TSGPU_HIP_TRY(hipMemcpyAsync(mem.c_meta.p, mine[0].data(), words * 8, hipMemcpyHostToDevice, mem.ctx->stream));
            if ((rc = all_gather_everywhere(g, &Member::c_meta, &Member::c_meta_all, words * 8))) return rc;
            TSGPU_HIP_TRY(hipMemcpyAsync(all.data(), mem.c_meta_all.p, (size_t)g->n * words * 8, hipMemcpyDeviceToHost, mem.ctx->stream));
            TSGPU_HIP_TRY(hipStreamSynchronize(mem.ctx->stream));
        }
        const size_t stride = std::max<size_t>(words, 1);
        for (size_t i = 0; i < ni; i++) {
            int64_t best = 0;
            for (uint32_t r = 0; r < g->n; r++) best = std::max(best, (int64_t)all[r * stride + i]);
            out->text_match[item_slot[i]] = best;
        }
        for (size_t i = 0; i < np_; i++)
            for (uint32_t r = 0; r < g->n; r++) {
                const float d = ((const float*)(all.data() + r * stride + ni))[i];
                if (d == d) { out->vector_distance[pair_slot[i]] = d; break; }       // (no shard holds a vector for the label: left as it is, like the single-GPU call)
            }
        hybrid_refuse(p, n_queries, out);
        return ok();
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_group_hybrid_search_batch: host allocation failed"); }
}

}  // extern "C"
