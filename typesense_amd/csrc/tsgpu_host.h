// tsgpu_host.h — host-side state behind the C-ABI (include/tsgpu.h): the context, the committed HBM snapshot
// of the keyword index, grow-only device scratch, error plumbing. HIP runtime only; no torch, no CPU scoring.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <deque>
#include <functional>
#include <vector>
#include <mutex>
#include <condition_variable>
#include <memory>
#include <atomic>
#include <thread>
#include <utility>
#include <unordered_map>
#include <algorithm>
#include <chrono>
#include <new>
#include <exception>
#include "../../include/tsgpu.h"
#include "tsgpu_format.h"
#include "tsgpu_pack.h"
#include "tsgpu_batcher.h"

namespace tsgpu {

// Option<T>-style error plumbing (include/option.h of the reference): code + message, never throw.
inline std::string& tls_error() { static thread_local std::string e; return e; }
inline bool& tls_no_coalesce() { static thread_local bool b = false; return b; }     // this thread's keyword batches go to a lane directly, never into the combiner of the 1-query
                                                                                      // callers (a grouped call's nested id pass holds ctx->mu: it must not park there)
inline bool& tls_avoid_lane0() { static thread_local bool b = false; return b; }      // this thread's keyword batches keep off lane 0 (it shares the vector path's stream)
inline int fail(int code, const std::string& msg) { tls_error() = msg; return code; }
inline int ok() { return TSGPU_OK; }

#define TSGPU_HIP_TRY(expr)                                                                          \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            return ::tsgpu::fail(_e == hipErrorOutOfMemory ? TSGPU_ERR_NO_MEMORY : TSGPU_ERR_DEVICE, \
                                 std::string(#expr) + ": " + hipGetErrorString(_e));                \
        }                                                                                            \
    } while (0)

// hipFree / hipHostFree synchronise the WHOLE device: a scratch buffer that outgrows itself in the middle of serving queries must not
// stall every other lane for milliseconds (the latency tail of the 1-query-caller regime). The outgrown allocation is parked here
// instead (growth is geometric: what is parked is at most what is in use) and freed where a device-wide wait hurts nobody:
// tsgpu_commit, tsgpu_destroy, or when more than 1 GiB is parked.
struct DeferredFrees {
    std::mutex m;
    std::vector<std::pair<void*, bool>> ptrs;        // (pointer, pinned host memory?)
    size_t bytes = 0;
    void park(void* p, size_t n, bool pinned) {
        if (!p) return;
        bool now = false;
        { std::lock_guard<std::mutex> lk(m); if (n > (256u << 20) || bytes + n > (1ull << 30)) now = true; else { ptrs.emplace_back(p, pinned); bytes += n; } }
        if (now) { if (pinned) (void)hipHostFree(p); else (void)hipFree(p); }
    }
    void drain() {
        std::vector<std::pair<void*, bool>> v;
        { std::lock_guard<std::mutex> lk(m); v.swap(ptrs); bytes = 0; }
        for (auto& e : v) { if (e.second) (void)hipHostFree(e.first); else (void)hipFree(e.first); }
    }
};
inline DeferredFrees& deferred_frees() { static DeferredFrees* d = new DeferredFrees; return *d; }      // (never destroyed: no static-destruction order issues at exit)

typedef void (*OomHook)();
inline OomHook& oom_hook() { static OomHook h = nullptr; return h; }      // set once RetireBin exists (below): what else can be freed when hipMalloc fails

// grow-only device / pinned-host buffers
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return TSGPU_OK;
        // geometric growth (and never tiny): with exact-fit growth the lanes of the 1-query-caller regime, whose rounds differ in
        // size, kept re-allocating (latency tail of 10-50 ms)
        size_t want = bytes + bytes / 4 + 256;
        if (want < 2 * cap) want = 2 * cap;
        if (cap < (16u << 20) && want < 4 * cap) want = 4 * cap;       // (small buffers quadruple: half as many allocation stalls on the way up)
        if (want < (64u << 10)) want = 64u << 10;
        if (p) { deferred_frees().park(p, cap, false); p = nullptr; cap = 0; }
        if (hipMalloc(&p, want) != hipSuccess) {         // out of memory with buffers parked: free them and try once more
            (void)hipGetLastError();
            p = nullptr;
            deferred_frees().drain();
            if (oom_hook()) oom_hook()();                // ... and the retired snapshots' buffers (RetireBin::drain_all)
            TSGPU_HIP_TRY(hipMalloc(&p, want));
        }
        cap = want;
        return TSGPU_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return TSGPU_OK;
        size_t want = bytes + bytes / 4 + 256;
        if (want < 2 * cap) want = 2 * cap;
        if (want < (64u << 10)) want = 64u << 10;
        if (p) { deferred_frees().park(p, cap, true); p = nullptr; cap = 0; }
        TSGPU_HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
        return TSGPU_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

// One (field, term) posting list on the host — the source of every device upload. `pl` holds the packed blocks (blk_ids / blk_meta
// word offsets index pl.ids_payload / pl.payload; a re-written block's new words are appended, the old ones are garbage until the
// list is re-packed). `dev[b]` = where block b's words live in the device arenas (NOPOS = not uploaded yet). Single-document
// mutations (tsgpu_posting_upsert / _erase) work on ONE decoded "open" block per term, packed again when another block is touched
// or at commit.
struct TermHost {
    static const uint64_t NOPOS = ~0ull;
    struct BlockPos { uint64_t idw = NOPOS, pw = NOPOS; };
    PackedList pl;
    std::vector<BlockPos> dev;
    uint32_t handle = 0xFFFFFFFFu;                   // its slot in the snapshot's list table (stable across commits)
    bool dirty = true;                               // differs from the published snapshot (a NEW term starts dirty and NOT queued)
    bool queued = false;                             // already has an entry in ctx->dirty_terms for the next commit (mark_dirty sets, a successful commit clears)
    // the list's descriptor arrays on the device: blk_last / blk_ids / blk_meta [d_blk_base, + d_blk_cap), the first d_blk_n entries
    // published. Blocks appended behind the published ones are written INTO the spare entries (no published entry changes: a search on
    // an older snapshot never looks beyond its own n_blocks); any other change (a published block re-written, split, removed) needs
    // the arrays re-written at the arena tail.
    uint64_t d_blk_base = NOPOS;
    uint32_t d_blk_cap = 0, d_blk_n = 0;
    bool desc_rewrite = true;
    bool has_breaks = false;                         // some block does not follow its predecessor in the ids arena (LIST_HAS_BREAKS)
    int64_t open_b = -1;                             // decoded block being mutated (-1 = none)
    std::vector<uint32_t> o_ids, o_oi, o_offs;       // o_oi has one extra end entry (= o_offs.size()) while open
    uint64_t garbage_idw = 0, garbage_pw = 0;        // host words no block refers to any more
    uint64_t dev_idw = 0, dev_pw = 0;                // arena words the NEWEST SNAPSHOT's version of this list occupies (0 before its first commit)
};

struct FieldHost {
    bool is_array = false;
    std::unordered_map<uint32_t, TermHost> terms;
};

// Device buffers of RETIRED snapshots / arena sets. The last reference to a snapshot is usually dropped by a SEARCH thread (RCU), and
// hipFree synchronises the whole device: the destructors hand their buffers over instead, the commit thread (tsgpu_commit, on entry and
// after publishing) and tsgpu_destroy free them — a search never pays for a commit's garbage.
struct RetireBin {
    std::mutex m;
    std::vector<void*> dev;
    size_t bytes = 0;
    RetireBin() { oom_hook() = &RetireBin::drain_all; std::lock_guard<std::mutex> lk(registry_mu()); registry().push_back(this); }
    // More than 1 GiB parked — a whole retired index copy after a compaction while writes go idle — is not kept until the next commit, but it is not
    // freed HERE either: put() runs on whichever thread drops the last reference, usually a search thread finishing a batch on an old snapshot, and
    // hipFree synchronises the device (a compaction under load stalled that search and serialised every lane). The buffer goes to a
    // reaper thread, started on first use, whose hipFree calls block nobody's request.
    struct Reaper {
        std::mutex m; std::condition_variable cv; std::vector<void*> q; bool started = false;
        void give(void* p) {
            std::lock_guard<std::mutex> lk(m);
            q.push_back(p);
            if (!started) { started = true; std::thread([this] { run(); }).detach(); }
            cv.notify_one();
        }
        void run() {
            for (;;) {
                std::vector<void*> v;
                { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return !q.empty(); }); v.swap(q); }
                for (void* p : v) (void)hipFree(p);
            }
        }
    };
    static Reaper& reaper() { static Reaper* r = new Reaper; return *r; }      // (never destroyed: outlives every bin)
    void put(DevBuf& b) {
        if (!b.p) return;
        bool over = false;
        { std::lock_guard<std::mutex> lk(m); if (bytes + b.cap > (1ull << 30)) over = true; else { dev.push_back(b.p); bytes += b.cap; } }
        if (over) reaper().give(b.p);
        b.p = nullptr; b.cap = 0;
    }
    void drain() { std::vector<void*> v; { std::lock_guard<std::mutex> lk(m); v.swap(dev); bytes = 0; } for (void* p : v) (void)hipFree(p); }
    ~RetireBin() {
        { std::lock_guard<std::mutex> lk(registry_mu()); auto& r = registry(); for (size_t i = 0; i < r.size(); i++) if (r[i] == this) { r[i] = r.back(); r.pop_back(); break; } }
        drain();
    }
    // every live bin: an allocation that fails drains them all before it gives up (DevBuf::reserve)
    static std::mutex& registry_mu() { static std::mutex* mu = new std::mutex; return *mu; }
    static std::vector<RetireBin*>& registry() { static std::vector<RetireBin*>* r = new std::vector<RetireBin*>; return *r; }
    static void drain_all() { std::lock_guard<std::mutex> lk(registry_mu()); for (RetireBin* b : registry()) b->drain(); }
};

// the device arenas of the posting lists; shared by consecutive snapshots: an incremental commit appends at the tails (regions no
// published snapshot refers to) and publishes a new descriptor table
struct ArenaSet {
    DevBuf blk_last, blk_ids, blk_meta, ids_payload, payload;
    uint64_t cap_blocks = 0, cap_idw = 0, cap_pw = 0;        // capacities (elements)
    uint64_t used_blocks = 0, used_idw = 0, used_pw = 0;     // tails
    uint64_t live_blocks = 0, live_idw = 0, live_pw = 0;     // referenced by the newest snapshot (used - live = garbage)
    ArenaSet() = default;
    ArenaSet(const ArenaSet&) = delete;
    ArenaSet& operator=(const ArenaSet&) = delete;
    std::shared_ptr<RetireBin> bin;
    ~ArenaSet() {
        if (bin) { bin->put(blk_last); bin->put(blk_ids); bin->put(blk_meta); bin->put(ids_payload); bin->put(payload); }
        else { blk_last.release(); blk_ids.release(); blk_meta.release(); ids_payload.release(); payload.release(); }
    }
    uint64_t bytes() const { return blk_last.cap + blk_ids.cap + blk_meta.cap + ids_payload.cap + payload.cap; }
};

// (field << 32 | term) -> list handle; shared by the snapshots between which no term appeared or disappeared
struct HandleMaps {
    std::unordered_map<uint64_t, uint32_t> handle_of;
    // device mirror of the flat tables for the device-side planner (kw_plan.hip.h): 64 x {offset, n} words, then the fields' tables back to
    // back; built by upload_dense() on the commit thread right after rebuild_dense(). Absent (p == nullptr) -> the host planner is used.
    DevBuf d_dense;
    std::shared_ptr<RetireBin> bin;
    HandleMaps() = default;
    HandleMaps(const HandleMaps& o) : handle_of(o.handle_of), bin(o.bin), dense_handle(o.dense_handle), dict_fp(o.dict_fp) {}     // (the copy gets its own mirror at its upload_dense())
    HandleMaps& operator=(const HandleMaps&) = delete;
    ~HandleMaps() { if (bin) bin->put(d_dense); else d_dense.release(); }
    void upload_dense() {
        if (dense_handle.size() > 64) return;
        std::vector<uint32_t> img(128, 0u);
        for (size_t f = 0; f < dense_handle.size(); f++) {
            img[2 * f] = (uint32_t)(img.size() - 128); img[2 * f + 1] = (uint32_t)dense_handle[f].size();
            img.insert(img.end(), dense_handle[f].begin(), dense_handle[f].end());
        }
        if (d_dense.reserve(img.size() * 4) != TSGPU_OK) { (void)hipGetLastError(); d_dense.p = nullptr; d_dense.cap = 0; return; }
        if (hipMemcpy(d_dense.p, img.data(), img.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); if (bin) bin->put(d_dense); else d_dense.release(); }
    }
    // the same map as flat tables for small field / term ids (the planner resolves three tokens per query, 10 000 queries per
    // batch: an indexed load instead of a hash probe); 0xFFFFFFFF = absent; ids beyond the tables go through handle_of
    std::vector<std::vector<uint32_t>> dense_handle;                // [field][term]
    // order-free fingerprint of the (field, term) pairs that have a posting list: doc-range shards compare theirs (tsgpu_group) — only when they differ can a
    // token be missing from one shard and present on another, and only then do the shards exchange per-query token masks
    uint64_t dict_fp = 0;
    void rebuild_dense() {
        std::vector<uint32_t> max_term;
        dict_fp = 0;
        for (const auto& e : handle_of) { uint64_t z = e.first + 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; dict_fp += z ^ (z >> 31); }
        for (const auto& e : handle_of) {
            const uint32_t f = (uint32_t)(e.first >> 32), term = (uint32_t)e.first;
            if (f >= 64 || term >= (4u << 20)) continue;
            if (f >= max_term.size()) max_term.resize(f + 1, 0);
            max_term[f] = std::max(max_term[f], term + 1);
        }
        dense_handle.assign(max_term.size(), {});
        for (size_t f = 0; f < dense_handle.size(); f++) dense_handle[f].assign(max_term[f], 0xFFFFFFFFu);
        for (const auto& e : handle_of) {
            const uint32_t f = (uint32_t)(e.first >> 32), term = (uint32_t)e.first;
            if (f < dense_handle.size() && term < dense_handle[f].size()) dense_handle[f][term] = e.second;
        }
    }
};

// ID DIRECTORIES of the long lists (tsgpu_format.h): fixed-size slots of one device buffer. A snapshot holds a reference per list; lists that
// did not change between two commits share theirs, a changed list gets a fresh slot (its directory is rebuilt on the device), and a slot
// returns to the pool's free list when the last snapshot that shows it is dropped — by whichever thread drops it: no device call.
struct IdDirPool {
    DevBuf buf;
    std::shared_ptr<RetireBin> bin;
    uint32_t slot_entries = 0, cap_ids = 0, n_slots = 0;
    std::mutex mu;
    std::vector<uint32_t> free_slots;
    IdDirPool() = default;
    IdDirPool(const IdDirPool&) = delete;
    IdDirPool& operator=(const IdDirPool&) = delete;
    ~IdDirPool() { if (bin) bin->put(buf); else buf.release(); }
    bool take(uint32_t& slot) { std::lock_guard<std::mutex> lk(mu); if (free_slots.empty()) return false; slot = free_slots.back(); free_slots.pop_back(); return true; }
    void give(uint32_t slot) { std::lock_guard<std::mutex> lk(mu); free_slots.push_back(slot); }
};
struct IdDirRef {
    std::shared_ptr<IdDirPool> pool;
    uint32_t slot = 0;
    IdDirRef(std::shared_ptr<IdDirPool> p, uint32_t s) : pool(std::move(p)), slot(s) {}
    IdDirRef(const IdDirRef&) = delete;
    IdDirRef& operator=(const IdDirRef&) = delete;
    ~IdDirRef() { if (pool) pool->give(slot); }
};

struct Snapshot {            // immutable view of all posting lists; published by tsgpu_commit, shared (RCU) by the searches that started on it
    std::shared_ptr<ArenaSet> ar;
    std::shared_ptr<IdDirPool> dir_pool;                             // id directories of the long lists (null: none)
    std::vector<std::shared_ptr<IdDirRef>> dir_of;                   // by list handle (shorter than h_lists / null entries: no directory)
    DevBuf lists;                                                   // ListDesc table (one per snapshot)
    std::vector<ListDesc> h_lists;                                  // host copy of the descriptors
    std::shared_ptr<const HandleMaps> maps;
    std::unordered_map<uint32_t, bool> field_is_array;              // the query_by fields of this snapshot (field id -> string[])
    uint64_t bytes = 0;
    uint32_t num_docs = 0;                                          // num_seq_ids() when the snapshot was published
    Snapshot() = default;
    Snapshot(const Snapshot&) = delete;
    Snapshot& operator=(const Snapshot&) = delete;
    std::shared_ptr<RetireBin> bin;
    ~Snapshot() { if (bin) bin->put(lists); else lists.release(); }
    uint32_t find_handle(uint32_t field, uint32_t term) const {
        if (!maps) return 0xFFFFFFFFu;
        if (field < maps->dense_handle.size() && term < maps->dense_handle[field].size()) return maps->dense_handle[field][term];
        auto it = maps->handle_of.find(((uint64_t)field << 32) | term);
        return it == maps->handle_of.end() ? 0xFFFFFFFFu : it->second;
    }
};

struct ColumnDev { DevBuf data; uint32_t n = 0; std::vector<int64_t> host; /* mirror for the <=k-hit host steps (vector / hybrid) */ };

struct VecField;             // tsgpu_vec.hip
struct FacetField;           // tsgpu_facet.hip

// One in-flight keyword batch: its own stream, events and every piece of per-batch scratch (plan upload, partial top-K lists,
// hit records, outputs). The context owns two lanes: while one batch runs on the GPU the next is planned, uploaded and launched
// on the other stream — searches never serialise on one global mutex, only on the lane they run on.
struct KwLane {
    std::mutex mu;                                   // one batch at a time per lane
    std::atomic<int> waiters{0};
    hipStream_t stream = nullptr;
    bool own_stream = true;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_chain = nullptr;                   // "this slice's kernels are done": the next slice of a sliced host-output batch waits for it ON THE DEVICE
    uint64_t wait_ema_us = 100;                      // how long this lane's recent rounds waited for the GPU (sleeping_wait)
    hipEvent_t ev_block = nullptr;                   // hipEventBlockingSync: the waiting thread sleeps instead of spinning (many concurrent callers)
    DevBuf d_touched;                                // option kw_count_touched: the find kernel's 8 byte counters
    DevBuf d_plan, d_ids_out;                        // the batch plan (queries, work items, aux ids, multi-field descriptors, hit offsets): one upload
    DevBuf d_plan_in, d_plan_work;                   // device-side planner (kw_plan.hip.h): per-query input records; work items + hit offsets it writes
    PinBuf h_plan_tot;                               // ... and its totals, read back twice per batch
    DevBuf d_part_s0, d_part_s1, d_part_s2, d_part_key, d_part_cnt, d_part_nm, d_part_ne, d_part_ow, d_part_f;
    DevBuf d_out_keys, d_out_scores, d_out_tm, d_out_vd, d_out_msi, d_out_nh, d_out_nm, d_out_ow, d_out_cut;
    // candidate-combination batches (tsgpu_keyword_search_candidates_batch): per-pass hits, group table, id-set bitmaps
    DevBuf d_cand_keys, d_cand_scores, d_cand_tm, d_cand_vd, d_cand_msi, d_cand_nh, d_cand_nm, d_cand_st, d_cand_gb, d_cand_qi, d_cand_found,
           d_cand_segs, d_cand_bits, d_cand_ids;
    DevBuf d_hits;                                   // hit records of the two-kernel form
    DevBuf d_t0, d_cut;                              // batch start stamp (device wall clock) + per-query cutoff flags (in-flight deadline)
    DevBuf d_fbits;                                  // rank bitmaps of filtered multi-field queries
    DevBuf d_idseg, d_idflat;                        // per-call id lists: segment table + the gathered ids
    PinBuf h_out, h_plan;
    // host side of a coalesced round (micro-batcher): the round's queries and its results before they are handed to the callers
    std::vector<tsgpu_kw_query> c_q;
    std::vector<uint64_t> c_keys, c_nm;
    std::vector<int64_t> c_scores, c_tm;
    std::vector<float> c_vd;
    std::vector<int8_t> c_msi;
    std::vector<uint32_t> c_nh;
    std::vector<int32_t> c_st, c_co;
    // state of the lane's last batch (legacy single-caller API: tsgpu_result_ids / tsgpu_candidates_result_ids)
    uint32_t last_cand_groups = 0;
    uint64_t last_cand_words = 0;
    std::vector<uint64_t> last_cand_found;
    const void* last_tab_q = nullptr;                // the last batch's tables ON THE DEVICE (queries, work items in partial-list order; valid until the lane's next batch):
    const void* last_tab_w = nullptr;                //   the candidate call's id-set marks read them instead of a host-built segment list
    uint32_t last_tab_n_work = 0;
    hipStream_t aux_stream = nullptr;                // candidate calls: the id-set bitmaps are cleared here WHILE the passes' find / score kernels run (created on first use)
    hipEvent_t ev_aux = nullptr;
    std::vector<uint64_t> last_ids_off;              // per query offset into d_ids_out of the last batch
    std::vector<std::vector<uint32_t>> last_chunk_emit;  // ids emitted per work item of the query
    std::vector<std::vector<uint32_t>> last_chunk_off;   // where each work item's id segment starts (relative to last_ids_off)
    std::vector<uint8_t> last_ids_unsorted;
    void release() {
        DevBuf* bufs[] = {&d_plan, &d_plan_in, &d_plan_work, &d_ids_out, &d_part_s0, &d_part_s1, &d_part_s2, &d_part_key, &d_part_cnt, &d_part_nm, &d_part_ne,
                          &d_part_ow, &d_part_f, &d_out_keys, &d_out_scores, &d_out_tm, &d_out_vd, &d_out_msi, &d_out_nh, &d_out_nm, &d_out_ow, &d_out_cut,
                          &d_cand_keys, &d_cand_scores, &d_cand_tm, &d_cand_vd, &d_cand_msi, &d_cand_nh, &d_cand_nm, &d_cand_st, &d_cand_gb, &d_cand_qi,
                          &d_cand_found, &d_cand_segs, &d_cand_bits, &d_cand_ids, &d_hits, &d_idseg, &d_idflat, &d_fbits, &d_t0, &d_cut};
        for (auto* b : bufs) b->release();
        h_out.release(); h_plan.release(); h_plan_tot.release();
        for (auto& e : ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        if (ev_block) { (void)hipEventDestroy(ev_block); ev_block = nullptr; }
        if (ev_chain) { (void)hipEventDestroy(ev_chain); ev_chain = nullptr; }
        if (ev_aux) { (void)hipEventDestroy(ev_aux); ev_aux = nullptr; }
        if (aux_stream) { (void)hipStreamDestroy(aux_stream); aux_stream = nullptr; }
        if (own_stream && stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
    }
};

struct KwRequest;                                    // tsgpu.hip
struct GroupByScratch;                               // tsgpu_groupby.inc.h
struct GbRequest;                                    // tsgpu_groupby.inc.h
struct VecRequest;                                   // tsgpu_vec.hip

}  // namespace tsgpu

// per-call matched-id lists (tsgpu_keyword_search_batch_ids): owned by the caller after the call
struct tsgpu_id_lists {
    std::vector<uint64_t> begin;                     // [n_queries + 1]
    std::vector<uint32_t> ids;                       // ascending per query
};

// a few parked host threads for per-query host work of a batch (hybrid rank fusion): starting 16 std::threads per call cost more than
// the fusion of 256 queries itself
struct HostPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, done_cv;
    const std::function<void()>* job = nullptr;
    uint64_t gen = 0;
    int want = 0, pending = 0;
    bool stop = false;
    void worker(int idx) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void()>* j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || (gen != seen && idx < want); });
                if (stop) return;
                seen = gen;
                j = job;
            }
            std::exception_ptr ex;
            try { (*j)(); } catch (...) { ex = std::current_exception(); }
            { std::lock_guard<std::mutex> lk(mu); if (ex && !failed) failed = ex; if (--pending == 0) done_cv.notify_all(); }
        }
    }
    // runs f on `helpers` pool threads and on the caller; returns when all are done. One call at a time (callers serialise on run_mu).
    // An exception thrown by f on any thread (std::bad_alloc of a plan slice) is rethrown here AFTER every thread has left f.
    std::mutex run_mu;
    std::exception_ptr failed;
    void run(const std::function<void()>& f, int helpers) {
        std::lock_guard<std::mutex> rl(run_mu);
        try { while ((int)th.size() < helpers) { const int idx = (int)th.size(); th.emplace_back([this, idx] { worker(idx); }); } } catch (...) { helpers = (int)th.size(); }
        { std::lock_guard<std::mutex> lk(mu); job = &f; want = helpers; pending = helpers; gen++; }
        cv.notify_all();
        std::exception_ptr mine;
        try { f(); } catch (...) { mine = std::current_exception(); }
        std::unique_lock<std::mutex> lk(mu);
        done_cv.wait(lk, [&] { return pending == 0; });
        job = nullptr; want = 0;
        std::exception_ptr ex = mine ? mine : failed;
        failed = nullptr;
        lk.unlock();
        if (ex) std::rethrow_exception(ex);
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
};

// Execution lanes are handed out in ARRIVAL ORDER. (A bare try_lock / lock on the lanes' mutexes let newly arriving round leaders barge
// past one that was already waiting: under 192+ request threads a round could starve for 40-60 ms — 0.2 % of the calls, and the whole
// tail of the latency distribution.) want < 0: any lane below n_lanes; else that lane.
struct LaneDispenser {
    struct Waiter { int want; int granted = -1; };
    std::mutex m;
    std::condition_variable cv;
    std::deque<Waiter*> q;
    bool busy[16] = {};
    static bool can(int lane, int want, int n) { return want == -2 ? (lane < n && (lane >= 1 || n == 1)) : (want < 0 ? lane < n : want == lane); }    // -2: any lane but 0 (lane 0 shares the vector path's stream)
    int acquire(int want, int n_lanes) {
        std::unique_lock<std::mutex> lk(m);
        for (int lane = 0; lane < 16; lane++) {
            if (busy[lane] || !can(lane, want, n_lanes)) continue;
            bool earlier = false;
            for (const Waiter* w : q) if (can(lane, w->want, n_lanes)) { earlier = true; break; }
            if (!earlier) { busy[lane] = true; return lane; }
        }
        Waiter me{want};
        q.push_back(&me);
        cv.wait(lk, [&] { return me.granted >= 0; });
        return me.granted;
    }
    void release(int lane, int n_lanes) {
        std::lock_guard<std::mutex> lk(m);
        for (auto it = q.begin(); it != q.end(); ++it) {
            if (can(lane, (*it)->want, n_lanes)) { (*it)->granted = lane; q.erase(it); cv.notify_all(); return; }      // the lane changes hands, still busy
        }
        busy[lane] = false;
    }
};

struct tsgpu_ctx {
    std::shared_ptr<tsgpu::RetireBin> retire_bin = std::make_shared<tsgpu::RetireBin>();
    HostPool host_pool;
    HostPool split_pool;                             // the second thread of a host-output batch served in slices (kw_split_host)
    LaneDispenser lane_dispenser;
    static const int N_LANES = 8;                    // lanes that exist; `n_lanes` of them are used (option "kw_lanes")
    int n_lanes = 4;
    int device = 0;
    hipStream_t stream = nullptr;                    // the vector path's stream (= the stream tsgpu_set_stream installs; lane 0 shares it)
    bool own_stream = true;
    std::mutex mu;                                   // index mutation + the vector path's scratch (one vector batch at a time)
    std::mutex tm_mu;                                // timings / counters of the last batch

    std::unordered_map<uint32_t, tsgpu::FieldHost> fields;
    std::shared_ptr<const tsgpu::Snapshot> snap;     // published snapshot: std::atomic_load / atomic_store
    std::shared_ptr<const tsgpu::Snapshot> snapshot() const { return std::atomic_load(&snap); }
    bool dirty = false;
    std::vector<uint64_t> dirty_terms;               // (field << 32 | term) touched since the last commit (may repeat)
    bool dirty_fields = false;                       // a query_by field was declared since the last commit
    uint64_t index_min_slack_words = 1u << 20;       // room (words per arena) a full commit leaves behind the data at least (option "index_min_slack_words")
    bool commit_force_full = false;                  // option "commit_full": the next commit re-packs everything (compaction)
    uint64_t erased_dev_idw = 0, erased_dev_pw = 0;  // arena words of lists erased since the last commit (they become garbage when it publishes)
    uint64_t index_compact_min_words = 1u << 20;     // garbage below this many words per arena never triggers a compaction (option "index_compact_min_words")
    uint64_t commit_compactions = 0;                 // full commits taken because the arenas' garbage outweighed their live words
    uint64_t commit_last_us = 0, commit_last_uploaded_bytes = 0, commit_full_count = 0, commit_incremental_count = 0;
    std::vector<tsgpu::ColumnDev> columns;
    tsgpu::DevBuf d_col_ptrs, d_col_len;
    uint32_t num_docs = 0;
    bool num_docs_set = false;

    tsgpu::KwLane lanes[N_LANES];
    std::atomic<int> last_lane{0};                   // lane of the most recent batch (legacy tsgpu_result_ids)
    tsgpu::Combiner<tsgpu::KwRequest> kw_comb;
    tsgpu::Combiner<tsgpu::VecRequest> vec_comb;
    tsgpu::Combiner<tsgpu::GbRequest> gb_comb;       // grouped keyword calls (tsgpu_keyword_search_grouped_batch)
    std::atomic<int> kw_callers{0}, vec_callers{0}, gb_callers{0};  // threads currently inside the search entry points
    uint32_t ticks_per_us = 100;                     // device wall clock (hipDeviceAttributeWallClockRate)
    bool hybrid_overlap = true;                      // tsgpu_hybrid_search_batch: keyword pass and vector pass at the same time (option "hybrid_overlap")
    uint32_t facet_ids_per_block = 0;      // 0 = chosen per batch (tsgpu_facet_count_batch)
    uint32_t vec_batch_post_window_us = 300;         // vector rounds: after the executor is free the leader waits this long for the callers of the round that just
                                                     // finished to come back (a round's cost hardly grows with its size: tsgpu_batcher.h)
    uint32_t batch_window_us = 10;                   // micro-batcher: how long a round's leader waits for more callers
    uint32_t batch_max_queries = 64;                 // calls with more queries than this are not coalesced (they are batches already)
    uint32_t batch_round_queries = 1024;             // queries per coalesced round at most

    // host-side phase totals of every keyword batch (us; introspection for the latency budget of small batches)
    std::atomic<uint64_t> kw_max_queue_us{0}, kw_max_wake_us{0};
    std::atomic<uint64_t> kw_queue_us{0}, kw_wake_us{0};   // sums over the coalesced calls: parked -> round starts / results ready -> caller resumes
    std::atomic<uint64_t> kw_max_plan_us{0}, kw_max_upload_us{0}, kw_max_launch_us{0}, kw_max_wait_us{0};     // slowest phase of any batch since the last read (diagnostics)
    std::atomic<uint64_t> kw_batches{0}, kw_plan_us{0}, kw_upload_us{0}, kw_launch_us{0}, kw_wait_us{0}, kw_book_us{0}, batch_exec_us{0}, batch_scatter_us{0};
    tsgpu::DevBuf d_prof;                            // TSGPU_PROF builds only (null otherwise)
    bool keep_ids = false;
    uint32_t kw_max_partials = 16;                   // work items (= partial top-K lists) per query at most; longer driver lists get longer items
    uint32_t kw_cost_r_x10 = 10, kw_cost_probe_x100 = 20;  // ... + 0.1 x kw_cost_r_x10 x |B|/|A| + 0.01 x kw_cost_probe_x100 x (stage-1 survivors per block)
    uint32_t kw_cost_fixed = 16;                     // launch-order cost model: item cost = driver blocks x (kw_cost_fixed + |B|/|A|)
    bool kw_sort_work = true;                        // lay the work table out heaviest query first
    uint32_t kw_host_split_queries = 1000;           // a host-output keyword batch of at least FOUR times this many queries is served in slices on two lanes: a slice's
                                                     // results cross PCIe while the next one computes (0 = never); no slice is smaller than this
    bool kw_host_split_device_plan = true;           // ... whose first (large) slice is planned on the device (kw_plan.hip.h) when it qualifies
    uint32_t kw_host_split_tail_slices = 1;          // slices behind the first one (1 or 2)
    uint32_t kw_host_split_first_pct = 85;           // ... the first slice's share of the batch (the rest: ONE tail slice; two with kw_host_split_tail_slices = 2)
    uint32_t kw_zero_copy_max_queries = 256;         // host-output keyword batches up to this many queries: the merge kernel writes into pinned host memory (0 = always copy)
    uint32_t kw_timing_min_queries = 64;             // keyword batches below this many queries skip the phase events (tsgpu_timings reports 0 ms for them)
    uint32_t kw_merge_select_min = 2;                // queries with at least this many partial lists are merged by selection (kw_select_partials: tree merge); 0 = always fold
    bool kw_count_touched = false;                   // measurement option: keyword batches launch the byte-counting instantiation of the find kernel
    tsgpu_kw_touched kw_touched{};                   // ... and leave its counters here (tsgpu_kw_last_touched; under tm_mu)
    std::atomic<uint64_t> kw_mf_pipelined_launches{0}, kw_candidates_rank_launches{0};
    bool kw_candidates_rank_fold = true;             // candidate combinations: the sort-free fold (kw_candidates_rank_kernel) instead of two bitonic sorts
    bool kw_mf_pipelined = true;                     // multi-field find kernel: the pipelined form for launches of <= 2 query_by fields (kw_find_mf2.hip.h)
    bool kw_pair_blocks = true;                      // find kernel variant: two driver blocks per iteration (kw_find2.hip.h)
    bool doc_range_set = false; uint32_t doc_range_lo = 0, doc_range_hi = 0;     // a doc-range shard of a group (options doc_range_lo / doc_range_hi; hi = 0: the whole collection): which seq_ids this context
                                                     // OWNS — a wildcard search (q = *) ranks only those (tsgpu_group_wildcard_search_batch); postings need no range: they are what was fed
    long long kw_iddir_min_ids = 256;                // id directories (tsgpu_format.h): lists of at least max(this, num_docs / kw_iddir_density_div) ids get one; 0 = none
    long long kw_iddir_density_div = 64;
    long long kw_iddir_budget_mb = 4096;             // device memory for the directory pool (longest lists first)
    uint64_t kw_iddir_built = 0;                     // counter: directories (re)built by commits
    bool kw_two_kernels = true;                      // queries of <= 3 tokens: find kernel + score kernel instead of the fused kernel
    uint32_t kw_device_plan_min_queries = 512;       // batches of plain single-field queries from this size on are planned ON THE DEVICE (kw_plan.hip.h); 0 = always on the host
    std::atomic<uint64_t> kw_device_plans{0}, kw_device_plan_fallbacks{0};
    uint32_t kw_hit_buffer_mb = 20480;               // budget of the hit-record buffer between the two (work items run in groups that fit)
    uint32_t kw_last_hit_groups = 0;
    uint64_t kw_hit_buffer_records = 0, kw_last_hit_records = 0;
    uint32_t kw_chunk_blocks = 0;                    // driver blocks per work item (0 = sized per batch, see plan_batch)
    uint32_t vec_rows_per_slab = 0;                  // 0 = automatic
    uint32_t vec_sample_tiles = 0;                   // 0 = automatic (8192 tiles on the bf16 prefilter path, 512 on the fp32 scan): 128-row tiles of the threshold sample
    uint32_t vec_cand_cap = 0;                       // candidate slots per query in pass 2 (0 = automatic)
    uint64_t commit_failed_count = 0;                // commits that returned an error (the next one re-packs from the host lists)
    uint32_t vec_ip_lanes = 4;                       // order of the exact distances' sums: 4 = hnswlib built for SSE (the reference's stock flags), 8 = AVX, 16 = AVX-512
    uint32_t vec_prefilter = 1;                      // 1 = bf16 bracket scan + exact fp32 re-score (default); 0 = fp32 MFMA scan
    uint64_t vec_prefilter_groups = 0;               // query groups answered by the bf16 bracket path
    int hnsw_test_tiny_cand = 0;                      // TESTS ONLY (option "hnsw_test_tiny_cand"): searches with max(ef, k) <= 128 start on a 24-entry candidate heap, so that
                                                     // the re-run of the overflowed queries on the largest tier is exercised on small graphs
    uint64_t hnsw_tier_reruns = 0;                   // queries that were run again on the largest tier since the context was created (counter "hnsw_tier_reruns")
    int hnsw_visited_hash = 1;                        // 1 = per-query hash sets of visited ids (default), 0 = 16-bit tags per row and concurrent query
    int hnsw_visited_max_gib = 64;                    // cap of one field's HNSW visited-tag array (option): bounds the queries traversing concurrently
    int blocking_sync_min_callers = 48;               // from this many threads inside the keyword entry point a lane waits for its round by sleeping (option)
    int plan_threads = 8;                             // host threads that plan a big keyword batch in slices ("plan_threads")
    uint32_t plan_parallel_min_queries = 2048;        // ... from this many queries on (0 = never; "plan_parallel_min_queries")
    int fuse_threads = 32;                            // host threads of the hybrid rank fusion (option "fuse_threads")
    uint64_t hnsw_last_expansions = 0, hnsw_last_distances = 0;   // last HNSW batch: candidates expanded / distances computed at layer 0 (all queries)
    uint64_t vec_rescored_rows = 0;                  // survivors re-scored in fp32 by the last prefilter group (sum over its queries; 0 unless vec_count_rescored)
    uint32_t vec_count_rescored = 0;
    uint64_t vec_prefilter_fallbacks = 0;            // query groups the bf16 bracket could not separate (ran on the fp32 scan)
    uint64_t vec_overflow_rounds = 0;                // pass-2 repeats caused by candidate overflow (introspection)

    std::unordered_map<uint32_t, tsgpu::VecField*> vec_fields;
    std::unordered_map<uint32_t, tsgpu::FacetField*> facet_fields;
    tsgpu::GroupByScratch* groupby = nullptr;        // device scratch of tsgpu_keyword_search_grouped_batch (under mu)

    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // [3..7]: the vector path
    tsgpu_timings timings{};
    tsgpu_aux_timings aux_timings{};                 // group_by / facet batches (tsgpu_last_aux_timings; under tm_mu)
    hipEvent_t aux_ev[4] = {nullptr, nullptr, nullptr, nullptr};   // created on first use, on the context's device
    bool scan_events_valid = false;                  // ev[6]/ev[7] bracket the main k-NN scan of the last batch's first query group
};

// ---- device-side halves of tsgpu_group's exchange (tsgpu_group.hip orchestrates; kernels: kw_kernels.hip.h / vec_kernels.hip.h).
// All of them ENQUEUE on `s` and do not synchronise. A keyword exchange block = n_q records of k * words + 3 u64 (KwShardIn::packed),
// a k-NN block = n_q * k u64 (vec_group_pack_kernel).
namespace tsgpu {
// flat vector branch (tsgpu_vec.hip -> tsgpu.hip): query i ranks row i of a device distance matrix aligned with its filter ids
struct KwVFlat { const float* dist_dev; size_t stride; float threshold; bool abs; };
int kw_vector_flat_search(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_queries, tsgpu_hits* out, const KwVFlat* vf, tsgpu_id_lists** ids_out);
inline size_t group_kw_record_words(uint32_t k, uint32_t words) { return (size_t)k * words + 3; }
int group_pack_keyword(tsgpu_ctx* ctx, const tsgpu_hits* local_dev, uint32_t n_q, uint32_t k, uint32_t words, uint64_t* block, hipStream_t s);
// merges records [0, n_q) of every gathered block into queries [q_out_offset, q_out_offset + n_q) of out_dev (caps_dev is indexed by OUTPUT query)
int group_merge_keyword(tsgpu_ctx* ctx, const uint64_t* gathered, uint64_t shard_stride_words, uint32_t n_shards, uint32_t n_q, uint32_t q_out_offset, uint32_t k, uint32_t words,
                        const uint32_t* caps_dev, const tsgpu_hits* out_dev, hipStream_t s, uint32_t pruned_per = 0);
// bound-pruned exchange (kw_kernels.hip.h): the shard's kq-th entries; counts against the gathered bounds; the pruned exchange block
// shard form of the candidate fold (tsgpu_group_keyword_search_candidates_batch): the candidates call with query_index = the hit's PASS and the shard's pass masks
// (device arrays), the tagging of the keys before the exchange and the fix-up after the merge (kw_group_cand_*_kernel)
int kw_candidates_batch_ex(tsgpu_ctx* ctx, const tsgpu_kw_query* combos, const uint32_t* group_begin, uint32_t n_groups, tsgpu_hits* out, uint32_t* query_index, uint64_t* found,
                           bool raw_pass, uint32_t* pass_mask_dev, const uint16_t* present_elsewhere = nullptr);
// rerank_hybrid_matches in pieces (tsgpu_vec.hip), shared with the group call: the missing distances BY LABEL (NaN: not in this context), which hits miss what, the re-fusion
int hybrid_missing_distances(tsgpu_ctx* ctx, uint32_t vec_field_id, const float* Q, int mem_q, uint32_t n_queries, const uint32_t* pair_q, const uint64_t* pair_label, uint32_t n, float* d_out);   // tsgpu_vec.hip
void hybrid_missing_items(uint32_t n_queries, const tsgpu_hits* out, std::vector<uint32_t>& item_q, std::vector<uint32_t>& item_id, std::vector<size_t>& item_slot,
                          std::vector<uint32_t>& pair_q, std::vector<uint64_t>& pair_label, std::vector<size_t>& pair_slot);
void hybrid_refuse(const tsgpu_hybrid_params* p, uint32_t n_queries, tsgpu_hits* out);
// Doc-range shards and the reference's "a token that matches no field is dropped" (get_field_token_its, src/index.cpp:5651-5655): whether a token EXISTS is a property of
// the whole collection. kw_dictionary_fingerprint: order-free hash of the context's (field, term) pairs with postings; kw_terms_present: bit t of masks[i] = token t of
// query i has a list in one of the query's fields HERE; kw_search_batch_masked: tsgpu_keyword_search_batch where bit t of present_elsewhere[i] says the token exists on
// ANOTHER shard — missing here, it is then an EMPTY list (the AND finds nothing on this shard) instead of a dropped token.
uint64_t kw_dictionary_fingerprint(tsgpu_ctx* ctx);
int kw_terms_present(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_queries, uint16_t* masks);
int kw_search_batch_masked(tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_queries, tsgpu_hits* out, const uint16_t* present_elsewhere);
int group_cand_tag(tsgpu_ctx* ctx, const tsgpu_hits* local_dev, const uint32_t* pass_of_hit, uint32_t n_groups, const uint32_t* pass_mask, const uint64_t* found, uint64_t* meta, hipStream_t s);
int group_cand_fix(tsgpu_ctx* ctx, uint64_t* keys, uint32_t* query_index, const uint32_t* n_hits, uint32_t k_stride, uint32_t q0, uint32_t q1, const uint64_t* meta_all, uint32_t n_shards,
                   uint32_t n_groups, uint64_t* found, hipStream_t s);
// shard form of group_by (tsgpu_group_keyword_search_grouped_batch): one grouped pass of a doc-range shard (tsgpu_groupby.inc.h) — tsgpu_keyword_search_grouped_batch with
// the shard options below, never coalesced with other callers. forced_begin == nullptr: the shard's own selection (round 1); else the groups of query i are the given keys
// forced_keys[forced_begin[i] .. forced_begin[i + 1]) (round 2: slot r = key r on every shard). present_elsewhere: as kw_search_batch_masked.
// cfirst: user query i owns the candidate combinations combos[cfirst[i] .. cfirst[i + 1]) (one each: the plain call). pass_mask (out, zeroed by the caller; nullable): bit p of
// pass_mask[i] = combination p of user query i matched something on this shard — and then query_index holds every hit's PASS (KV::query_index counts the earlier passes that
// matched ANYWHERE: the group call derives it from the shards' masks).
struct GbShard { const uint64_t* forced_keys; const uint32_t* forced_begin; const uint16_t* present_elsewhere; uint32_t* pass_mask; };
uint64_t gb_registers_cardinality(const uint8_t* regs);      // LogLogBeta::cardinality() of 16384 registers (the shards' sketches merge by the registers' maxima)
int gb_shard_batch(tsgpu_ctx* ctx, const tsgpu_kw_query* combos, const uint32_t* cfirst, const tsgpu_group_by* groups, uint32_t n_queries, tsgpu_hits* out, tsgpu_grouped_hits* gout,
                   uint32_t* query_index, const GbShard* shard);
int group_kw_kth(tsgpu_ctx* ctx, const tsgpu_hits* local_dev, uint32_t n_q, uint32_t k, uint32_t n_shards, const uint32_t* caps_dev, int64_t* kth, hipStream_t s);
int group_kw_prune_pack(tsgpu_ctx* ctx, const tsgpu_hits* local_dev, uint32_t n_q, uint32_t n_pad, uint32_t k, uint32_t words, const uint32_t* caps_dev, const int64_t* kth_all,
                        uint32_t n_shards, uint32_t per, uint32_t n_dst, uint64_t slice_words, uint64_t* block, uint32_t* cursor, hipStream_t s);
int group_store_keyword_slice(tsgpu_ctx* ctx, const tsgpu_hits* local_dev, uint32_t n_q, uint32_t q_out_offset, uint32_t k, const tsgpu_hits* out_dev, hipStream_t s);   // replicas form
int group_vec_dim(tsgpu_ctx* ctx, uint32_t vec_field_id, uint32_t* dim);
void group_resolve_topster_sizes(const tsgpu_ctx* ctx, const tsgpu_kw_query* queries, uint32_t n_q, uint32_t* caps_host);   // Topster capacity per query (src/index.cpp:3506-3512)
int group_pack_knn(tsgpu_ctx* ctx, const float* dist_dev, const uint64_t* label_dev, const uint32_t* cnt_dev, uint32_t n_q, uint32_t k, uint64_t* block, uint32_t* bad_dev, hipStream_t s);
int group_merge_knn(tsgpu_ctx* ctx, const uint64_t* gathered, uint64_t shard_stride_words, uint32_t n_shards, uint32_t n_q, uint32_t k,
                    float* dist_dev, uint64_t* label_dev, uint32_t* cnt_dev, hipStream_t s);
}
