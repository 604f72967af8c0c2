// tsgpu_index.hip — the posting-list mirror behind include/tsgpu.h: host-side lists, single-document mutation, and the commit that
// publishes them to HBM. What it mirrors in the reference: posting_t::upsert / erase on the ART leaf's posting object
// (src/posting.cpp:247-330 -> posting_list_t::upsert / erase, src/posting_list.cpp) under Index::mutex's unique_lock
// (src/index.cpp:575-700), i.e. ONE block of ONE list changes per (token, document). The commit therefore is block-granular:
//   * a mutated block is re-packed and its words are APPENDED at the tails of the device arenas (regions no published snapshot
//     refers to), the list's small descriptor arrays (blk_last / BlockIds / BlockMeta, 52 B per block) are re-written there too,
//     and a new descriptor TABLE is published (RCU): O(touched blocks), not O(index) — searches never wait and never see a mix;
//   * relocated blocks break the "a run of blocks is one coalesced range" property the intersection kernel exploits; such lists
//     carry LIST_HAS_BREAKS and runs across a break are probed per candidate until the next compaction (= a full re-pack, taken
//     when the tails run out of room, the arenas' garbage (used - live words, tracked per commit) outweighs their live words, or on
//     option "commit_full").
#include "tsgpu_host.h"

using namespace tsgpu;

namespace {

uint64_t wall_us() { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// one validator for every term entry point: ids strictly ascending, offset_index strictly ascending (every document owns at least one
// offset: Match reads runs[t].n - 1) and inside [0, n_off)
const char* validate_list(const uint32_t* ids, const uint64_t* oi, uint32_t n_ids, uint64_t n_off) {
    if (n_ids == 0) return nullptr;
    if (oi[n_ids - 1] >= n_off) return "offset_index beyond offsets (every document needs at least one offset)";
    for (uint32_t i = 1; i < n_ids; i++) {
        if (ids[i] <= ids[i - 1]) return "ids must be strictly ascending";
        if (oi[i] <= oi[i - 1]) return "offset_index must be strictly ascending (every document needs at least one offset)";
    }
    return nullptr;
}

inline uint32_t ids_words(const BlockMeta& m) { return packed_words(m.n_ids, m.ids_bits); }
inline uint32_t pay_words(const BlockMeta& m) { return packed_words(m.n_ids, m.oi_bits) + packed_words(m.n_off, m.off_bits); }

void mark_dirty(tsgpu_ctx* ctx, uint32_t field, uint32_t term, TermHost* t) {
    ctx->dirty = true;
    if (t && t->queued) return;                              // already queued for the next commit (one entry per term, not per posting operation)
    if (t) { t->dirty = true; t->queued = true; }            // (`dirty` starts true on a new term: it cannot double as the queue marker)
    ctx->dirty_terms.push_back(((uint64_t)field << 32) | term);
}

// a term leaves the index: what its committed version occupies in the arenas becomes garbage with the next commit
bool erase_term(tsgpu_ctx* ctx, FieldHost& f, uint32_t term_id) {
    auto it = f.terms.find(term_id);
    if (it == f.terms.end()) return false;
    ctx->erased_dev_idw += it->second.dev_idw; ctx->erased_dev_pw += it->second.dev_pw;
    f.terms.erase(it);
    return true;
}

void set_list(TermHost& t, PackedList&& pl) {
    t.pl = std::move(pl);
    t.dev.assign(t.pl.blk_last.size(), TermHost::BlockPos());
    t.open_b = -1;
    t.garbage_idw = t.garbage_pw = 0;
    t.desc_rewrite = true;
}

void refresh_desc(TermHost& t) {
    ListDesc& d = t.pl.desc;
    d.n_blocks = (uint32_t)t.pl.blk_last.size();
    uint64_t n_ids = 0, n_off = 0;
    for (const BlockMeta& m : t.pl.blk_meta) { n_ids += m.n_ids; n_off += m.n_off; }
    d.n_ids = (uint32_t)n_ids;
    d.n_off = (uint32_t)std::min<uint64_t>(n_off, 0xFFFFFFFFull);
    d.first_id = d.n_blocks ? t.pl.blk_ids[0].first_id : 0;
    d.last_id = d.n_blocks ? t.pl.blk_last.back() : 0;
}

// decode block b into the term's open-block arrays
void open_block(TermHost& t, int64_t b) {
    const BlockMeta& m = t.pl.blk_meta[(size_t)b];
    unpack_block(m, t.pl.ids_payload.data() + m.ids_woff, t.pl.payload.data() + m.oi_woff, t.pl.payload.data() + m.off_woff, t.o_ids, t.o_oi, t.o_offs);
    t.o_oi.push_back((uint32_t)t.o_offs.size());
    t.open_b = b;
}

// pack the open block back (appending its words; the previous version becomes garbage). An emptied block disappears.
void flush_open(TermHost& t) {
    if (t.open_b < 0) return;
    const size_t b = (size_t)t.open_b;
    t.open_b = -1;
    const BlockMeta old = t.pl.blk_meta[b];
    t.garbage_idw += ids_words(old);
    t.garbage_pw += pay_words(old);
    if (t.dev[b].idw != TermHost::NOPOS || t.o_ids.empty()) t.desc_rewrite = true;      // a published block changes / a block disappears
    if (t.o_ids.empty()) {
        t.pl.blk_last.erase(t.pl.blk_last.begin() + b); t.pl.blk_ids.erase(t.pl.blk_ids.begin() + b); t.pl.blk_meta.erase(t.pl.blk_meta.begin() + b);
        t.dev.erase(t.dev.begin() + b);
    } else {
        pack_block(t.pl, t.o_ids.data(), t.o_oi.data(), t.o_offs.data(), (uint32_t)t.o_ids.size(), (uint32_t)t.o_offs.size(), t.pl.blk_ids[b], t.pl.blk_meta[b]);
        t.pl.blk_last[b] = t.o_ids.back();
        t.dev[b] = TermHost::BlockPos();
    }
    refresh_desc(t);
}

// host arrays carry more garbage than live words: decode + pack afresh (every block is uploaded again by the next commit)
void repack_if_wasteful(TermHost& t) {
    if (t.garbage_idw + t.garbage_pw < 4096 || t.garbage_idw + t.garbage_pw < t.pl.ids_payload.size() / 2 + t.pl.payload.size() / 2) return;
    std::vector<uint32_t> ids, oi, offs, a, b, c;
    std::vector<uint64_t> oi64;
    for (const BlockMeta& m : t.pl.blk_meta) {
        unpack_block(m, t.pl.ids_payload.data() + m.ids_woff, t.pl.payload.data() + m.oi_woff, t.pl.payload.data() + m.off_woff, a, b, c);
        const uint64_t base = offs.size();
        ids.insert(ids.end(), a.begin(), a.end());
        for (uint32_t x : b) oi64.push_back(base + x);
        offs.insert(offs.end(), c.begin(), c.end());
    }
    const uint32_t handle = t.handle;
    set_list(t, pack_list(ids.data(), oi64.data(), offs.data(), (uint32_t)ids.size(), offs.size()));
    t.handle = handle;
}

// split the OPEN block b (256 ids) in two halves; the open block becomes the half that will hold `id`
void split_open(TermHost& t, uint32_t id) {
    const size_t b = (size_t)t.open_b;
    const size_t n = t.o_ids.size(), h = n / 2;
    std::vector<uint32_t> l_ids(t.o_ids.begin(), t.o_ids.begin() + h), r_ids(t.o_ids.begin() + h, t.o_ids.end());
    const uint32_t cut = t.o_oi[h];
    std::vector<uint32_t> l_oi(t.o_oi.begin(), t.o_oi.begin() + h), r_oi;
    for (size_t i = h; i < n; i++) r_oi.push_back(t.o_oi[i] - cut);
    std::vector<uint32_t> l_off(t.o_offs.begin(), t.o_offs.begin() + cut), r_off(t.o_offs.begin() + cut, t.o_offs.end());
    // left half stays block b (packed now), right half becomes block b + 1
    t.o_ids = l_ids; t.o_oi = l_oi; t.o_offs = l_off;
    flush_open(t);                                        // (accounts the old words as garbage once)
    BlockIds bi; BlockMeta bm;
    pack_block(t.pl, r_ids.data(), r_oi.data(), r_off.data(), (uint32_t)r_ids.size(), (uint32_t)r_off.size(), bi, bm);
    t.pl.blk_ids.insert(t.pl.blk_ids.begin() + b + 1, bi);
    t.pl.blk_meta.insert(t.pl.blk_meta.begin() + b + 1, bm);
    t.pl.blk_last.insert(t.pl.blk_last.begin() + b + 1, r_ids.back());
    t.dev.insert(t.dev.begin() + b + 1, TermHost::BlockPos());
    t.desc_rewrite = true;
    refresh_desc(t);
    open_block(t, id <= t.pl.blk_last[b] ? (int64_t)b : (int64_t)b + 1);
}

}  // namespace

extern "C" {

int tsgpu_field_create(tsgpu_ctx* ctx, uint32_t field_id, int is_array) {
    if (!ctx) return fail(TSGPU_ERR_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->fields[field_id].is_array = is_array != 0;
    ctx->dirty_fields = true;
    ctx->dirty = true;
    return ok();
}

int tsgpu_term_upsert(tsgpu_ctx* ctx, uint32_t field_id, uint32_t term_id, const uint32_t* ids, const uint32_t* offset_index,
                      const uint32_t* offsets, uint32_t n_ids, uint32_t n_offsets) {
    if (!ctx) return fail(TSGPU_ERR_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto fit = ctx->fields.find(field_id);
    if (fit == ctx->fields.end()) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_term_upsert: unknown field (call tsgpu_field_create)");
    if (n_ids == 0) { if (erase_term(ctx, fit->second, term_id)) mark_dirty(ctx, field_id, term_id, nullptr); return ok(); }
    if (!ids || !offset_index || !offsets) return fail(TSGPU_ERR_INVALID, "tsgpu_term_upsert: NULL array");
    try {
        std::vector<uint64_t> oi(offset_index, offset_index + n_ids);
        if (const char* why = validate_list(ids, oi.data(), n_ids, n_offsets)) return fail(TSGPU_ERR_INVALID, std::string("tsgpu_term_upsert: ") + why);
        TermHost& t = fit->second.terms[term_id];
        const uint32_t handle = t.handle;
        set_list(t, pack_list(ids, oi.data(), offsets, n_ids, n_offsets));
        t.handle = handle;
        mark_dirty(ctx, field_id, term_id, &t);
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_term_upsert: host allocation failed"); }
    return ok();
}

int tsgpu_terms_load_csr(tsgpu_ctx* ctx, uint32_t field_id, uint32_t n_terms, const uint32_t* term_ids, const uint64_t* ids_ptr,
                         const uint32_t* ids, const uint64_t* offset_index, const uint64_t* off_ptr, const uint32_t* offsets) {
    if (!ctx) return fail(TSGPU_ERR_INVALID, "ctx is NULL");
    if (!term_ids || !ids_ptr || !ids || !offset_index || !off_ptr || !offsets) return fail(TSGPU_ERR_INVALID, "tsgpu_terms_load_csr: NULL array");
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto fit = ctx->fields.find(field_id);
    if (fit == ctx->fields.end()) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_terms_load_csr: unknown field");
    try {
        // validate the whole load before touching the field: a rejected call leaves the pending state as it was
        for (uint32_t t = 0; t < n_terms; t++) {
            const uint64_t a = ids_ptr[t], b = ids_ptr[t + 1];
            if (b < a || off_ptr[t + 1] < off_ptr[t]) return fail(TSGPU_ERR_INVALID, "tsgpu_terms_load_csr: ids_ptr / off_ptr must be non-decreasing");
            if (b - a > 0xFFFFFFFFull) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_terms_load_csr: a list of more than 2^32 ids");
        }
        std::vector<uint64_t> oi;
        std::vector<std::pair<uint32_t, PackedList>> packed;
        packed.reserve(n_terms);
        for (uint32_t t = 0; t < n_terms; t++) {
            const uint64_t a = ids_ptr[t], b = ids_ptr[t + 1];
            if (b == a) { packed.emplace_back(term_ids[t], PackedList()); continue; }
            const uint64_t o0 = off_ptr[t], o1 = off_ptr[t + 1];
            oi.resize(b - a);
            for (uint64_t i = a; i < b; i++) {
                if (offset_index[i] < o0) return fail(TSGPU_ERR_INVALID, "tsgpu_terms_load_csr: offset_index below the list's off_ptr");
                oi[i - a] = offset_index[i] - o0;
            }
            if (const char* why = validate_list(ids + a, oi.data(), (uint32_t)(b - a), o1 - o0)) return fail(TSGPU_ERR_INVALID, std::string("tsgpu_terms_load_csr: ") + why);
            packed.emplace_back(term_ids[t], pack_list(ids + a, oi.data(), offsets + o0, (uint32_t)(b - a), o1 - o0));
        }
        for (auto& e : packed) {
            if (e.second.desc.n_ids == 0) { if (erase_term(ctx, fit->second, e.first)) mark_dirty(ctx, field_id, e.first, nullptr); continue; }
            TermHost& t = fit->second.terms[e.first];
            const uint32_t handle = t.handle;
            set_list(t, std::move(e.second));
            t.handle = handle;
            mark_dirty(ctx, field_id, e.first, &t);
        }
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_terms_load_csr: host allocation failed"); }
    return ok();
}

// posting_t::upsert(obj, id, offsets) (src/posting.cpp:247-288): document `id` gets `offsets` in the list of (field, term) — inserted in id
// order, or its run replaced when the document is already there. One block of the list changes.
int tsgpu_posting_upsert(tsgpu_ctx* ctx, uint32_t field_id, uint32_t term_id, uint32_t id, const uint32_t* offsets, uint32_t n_offsets) {
    if (!ctx) return fail(TSGPU_ERR_INVALID, "ctx is NULL");
    if (!offsets || n_offsets == 0) return fail(TSGPU_ERR_INVALID, "tsgpu_posting_upsert: a posting needs at least one offset");
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto fit = ctx->fields.find(field_id);
    if (fit == ctx->fields.end()) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_posting_upsert: unknown field (call tsgpu_field_create)");
    try {
        TermHost& t = fit->second.terms[term_id];
        if (t.pl.blk_last.empty()) {                             // a new term: one block with one document
            const uint64_t oi0 = 0;
            const uint32_t handle = t.handle;
            set_list(t, pack_list(&id, &oi0, offsets, 1, n_offsets));
            t.handle = handle;
            mark_dirty(ctx, field_id, term_id, &t);
            return ok();
        }
        // the block that holds / should hold the id: the first one whose last id is >= id, else the last block (blk_last / first_id of
        // the open block are kept current below, so the search sees its pending content)
        size_t b = std::lower_bound(t.pl.blk_last.begin(), t.pl.blk_last.end(), id) - t.pl.blk_last.begin();
        if (b == t.pl.blk_last.size()) {
            // a new document behind the whole list (the common case: new documents carry the largest ids). A PUBLISHED last block is
            // left alone — the document opens a fresh block behind it, whose descriptor goes into the list's spare entries: nothing a
            // running search can see changes, and the commit uploads this block only. (Small tail blocks are merged by the re-pack of
            // fragmented lists at commit.)
            const size_t lb = t.pl.blk_last.size() - 1;
            const bool last_open = t.open_b == (int64_t)lb;
            const size_t last_n = last_open ? t.o_ids.size() : t.pl.blk_meta[lb].n_ids;
            if (t.dev[lb].idw != TermHost::NOPOS || last_n >= BLOCK_IDS) {
                flush_open(t);
                BlockIds bi; BlockMeta bm;
                const uint32_t zero = 0;
                pack_block(t.pl, &id, &zero, offsets, 1, n_offsets, bi, bm);
                t.pl.blk_ids.push_back(bi); t.pl.blk_meta.push_back(bm); t.pl.blk_last.push_back(id); t.dev.push_back(TermHost::BlockPos());
                refresh_desc(t);
                mark_dirty(ctx, field_id, term_id, &t);
                return ok();
            }
            b = lb;
        }
        if (t.open_b != (int64_t)b) { flush_open(t); open_block(t, (int64_t)b); }
        size_t p = std::lower_bound(t.o_ids.begin(), t.o_ids.end(), id) - t.o_ids.begin();
        const bool exists = p < t.o_ids.size() && t.o_ids[p] == id;
        if (!exists && t.o_ids.size() >= BLOCK_IDS) {
            split_open(t, id);
            p = std::lower_bound(t.o_ids.begin(), t.o_ids.end(), id) - t.o_ids.begin();
        }
        if (exists) {                                            // replace the document's run
            const uint32_t s = t.o_oi[p], e = t.o_oi[p + 1];
            t.o_offs.erase(t.o_offs.begin() + s, t.o_offs.begin() + e);
            t.o_offs.insert(t.o_offs.begin() + s, offsets, offsets + n_offsets);
            const int64_t delta = (int64_t)n_offsets - (int64_t)(e - s);
            for (size_t i = p + 1; i < t.o_oi.size(); i++) t.o_oi[i] = (uint32_t)((int64_t)t.o_oi[i] + delta);
        } else {
            const uint32_t s = t.o_oi[p];
            t.o_ids.insert(t.o_ids.begin() + p, id);
            t.o_offs.insert(t.o_offs.begin() + s, offsets, offsets + n_offsets);
            t.o_oi.insert(t.o_oi.begin() + p, s);
            for (size_t i = p + 1; i < t.o_oi.size(); i++) t.o_oi[i] += n_offsets;
        }
        t.pl.blk_last[(size_t)t.open_b] = t.o_ids.back();
        t.pl.blk_ids[(size_t)t.open_b].first_id = t.o_ids.front();
        t.pl.blk_ids[(size_t)t.open_b].last_id = t.o_ids.back();
        mark_dirty(ctx, field_id, term_id, &t);
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_posting_upsert: host allocation failed"); }
    return ok();
}

// posting_t::erase(obj, id) (src/posting.cpp:290-330): the document leaves the list (no-op when it is not there); an emptied list disappears
int tsgpu_posting_erase(tsgpu_ctx* ctx, uint32_t field_id, uint32_t term_id, uint32_t id) {
    if (!ctx) return fail(TSGPU_ERR_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto fit = ctx->fields.find(field_id);
    if (fit == ctx->fields.end()) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_posting_erase: unknown field");
    auto tit = fit->second.terms.find(term_id);
    if (tit == fit->second.terms.end()) return ok();
    try {
        TermHost& t = tit->second;
        flush_open(t);
        const size_t b = std::lower_bound(t.pl.blk_last.begin(), t.pl.blk_last.end(), id) - t.pl.blk_last.begin();
        if (b == t.pl.blk_last.size() || id < t.pl.blk_ids[b].first_id) return ok();
        open_block(t, (int64_t)b);
        const size_t p = std::lower_bound(t.o_ids.begin(), t.o_ids.end(), id) - t.o_ids.begin();
        if (p == t.o_ids.size() || t.o_ids[p] != id) { t.open_b = -1; return ok(); }
        const uint32_t s = t.o_oi[p], e = t.o_oi[p + 1];
        t.o_offs.erase(t.o_offs.begin() + s, t.o_offs.begin() + e);
        t.o_ids.erase(t.o_ids.begin() + p);
        t.o_oi.erase(t.o_oi.begin() + p);
        for (size_t i = p; i < t.o_oi.size(); i++) t.o_oi[i] -= (e - s);
        flush_open(t);
        if (t.pl.blk_last.empty()) { erase_term(ctx, fit->second, term_id); mark_dirty(ctx, field_id, term_id, nullptr); }
        else mark_dirty(ctx, field_id, term_id, &t);
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_posting_erase: host allocation failed"); }
    return ok();
}

}  // extern "C"

namespace {

inline uint32_t desc_capacity(size_t nb) { return (uint32_t)(nb + std::max<size_t>(8, nb / 8)); }

// device descriptors of blocks [from, n_blocks) of a list whose blocks all have device positions
void make_descriptors(TermHost& t, size_t from, std::vector<uint32_t>& last, std::vector<BlockIds>& bids, std::vector<BlockMeta>& bmeta,
                      uint64_t ids_base, uint64_t pay_base) {
    const size_t nb = t.pl.blk_last.size();
    for (size_t b = from; b < nb; b++) {
        const BlockMeta& hm = t.pl.blk_meta[b];
        BlockIds bi = t.pl.blk_ids[b];
        BlockMeta bm = hm;
        bi.ids_woff = (uint32_t)(t.dev[b].idw - ids_base);
        bm.ids_woff = bi.ids_woff;
        bm.oi_woff = (uint32_t)(t.dev[b].pw - pay_base);
        bm.off_woff = bm.oi_woff + (hm.off_woff - hm.oi_woff);
        if (b > 0 && t.dev[b].idw != t.dev[b - 1].idw + ids_words(t.pl.blk_meta[b - 1])) t.has_breaks = true;
        last.push_back(t.pl.blk_last[b]); bids.push_back(bi); bmeta.push_back(bm);
    }
}

ListDesc list_desc(const TermHost& t, uint64_t ids_base, uint64_t pay_base, uint64_t blk_base) {
    ListDesc d = t.pl.desc;
    d.ids_base = ids_base; d.payload_base = pay_base; d.blk_base = (uint32_t)blk_base;
    d.flags = t.has_breaks ? LIST_HAS_BREAKS : 0u;
    d.dir_slot = 0;
    return d;
}

// many small tail blocks (one per write batch): decode + pack afresh into full blocks (the whole list is uploaded again)
void repack_if_fragmented(TermHost& t) {
    const size_t nb = t.pl.blk_last.size(), ideal = ((size_t)t.pl.desc.n_ids + BLOCK_IDS - 1) / BLOCK_IDS;
    if (nb <= ideal + ideal / 2 + 8) return;
    t.garbage_idw = t.pl.ids_payload.size() + 4096;          // (force the re-pack below)
    t.garbage_pw = t.pl.payload.size() + 4096;
}

// entries written into the spare slots of published descriptor arrays
__global__ void index_desc_scatter_kernel(const uint64_t* __restrict__ dst, const uint32_t* __restrict__ last, const BlockIds* __restrict__ bids,
                                          const BlockMeta* __restrict__ bmeta, uint32_t n, uint32_t* a_last, BlockIds* a_bids, BlockMeta* a_bmeta) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t d = dst[i];
    a_last[d] = last[i]; a_bids[d] = bids[i]; a_bmeta[d] = bmeta[i];
}

// ---- id directories of the long lists (tsgpu_format.h) -----------------------------------------------------------------------------------
struct IdDirJob { uint64_t ids_base; uint64_t first_block; uint32_t blk_base, n_blocks, slot, pad; };      // first_block: prefix of n_blocks over the jobs

// one workgroup per posting block of a list that gets a (new) directory; thread t = slot t: sets its id's bit, and the entry's position if
// it is the lowest id of its entry; an entry that also holds ids of the previous block is marked IDDIR_SPLIT. (atomicOr on zeroed memory:
// every writer of an entry agrees with every other.)
__global__ __launch_bounds__(256) void index_iddir_build_kernel(const IdDirJob* __restrict__ jobs, uint32_t n_jobs, const BlockIds* __restrict__ blk_ids,
                                                                const uint32_t* __restrict__ ids_payload, uint2* __restrict__ dir, uint32_t slot_entries, uint32_t cap_ids) {
    const uint64_t gb = blockIdx.x;
    uint32_t lo = 0, hi = n_jobs - 1;
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (jobs[mid].first_block <= gb) lo = mid; else hi = mid - 1; }
    const IdDirJob j = jobs[lo];
    const uint32_t b = (uint32_t)(gb - j.first_block);
    const BlockIds m = blk_ids[j.blk_base + b];
    const uint32_t n = m.n_ids_bits & 0xFFFF, t = threadIdx.x;
    if (t >= n) return;
    const uint32_t* __restrict__ w = ids_payload + j.ids_base + m.ids_woff;
    const bool w16 = (m.n_ids_bits >> 16) == 16;
    const uint32_t id = m.first_id + (w16 ? (uint32_t)((const uint16_t*)w)[t] : w[t]);
    if (id >= cap_ids) return;
    unsigned int* e = (unsigned int*)(dir + (size_t)j.slot * slot_entries + (id >> 5));
    bool lowest;
    if (t == 0) {
        const bool shared = b > 0 && (blk_ids[j.blk_base + b - 1].last_id >> 5) == (id >> 5);
        if (shared) atomicOr(e, IDDIR_SPLIT);
        lowest = !shared;
    } else lowest = ((m.first_id + (w16 ? (uint32_t)((const uint16_t*)w)[t - 1] : w[t - 1])) >> 5) != (id >> 5);
    if (lowest) atomicOr(e, b * (uint32_t)BLOCK_IDS + t);
    atomicOr(e + 1, 1u << (id & 31u));
}

// Decides which lists of the snapshot being built carry a directory, shares the unchanged ones with the previous snapshot, builds the others
// on the device and writes ListDesc::dir_slot of every list (so: call it BEFORE s.h_lists goes to the device, AFTER the arenas hold the
// snapshot's blocks). touched = handles whose list changed since `cur` (null: every list is new). Running out of slots or of memory is not
// an error: such lists simply keep the two-level probe.
int build_id_directories(tsgpu_ctx* ctx, Snapshot& s, const Snapshot* cur, const std::vector<uint32_t>* touched, uint32_t num_docs) {
    for (ListDesc& d : s.h_lists) d.dir_slot = 0;
    s.dir_pool.reset(); s.dir_of.clear();
    if (ctx->kw_iddir_min_ids <= 0 || ctx->kw_iddir_budget_mb <= 0 || !s.ar || s.h_lists.empty()) return TSGPU_OK;
    // density is judged over the id range this context's lists cover: a doc-range shard keeps GLOBAL seq_ids, so its lists are as dense
    // inside [lo, hi) as the unsharded ones are inside [0, num_docs) — with num_docs as the yardstick an eighth-shard would give directories to
    // an eighth of the lists that deserve one
    uint64_t id_lo = ~0ull, id_hi = 0;
    for (const ListDesc& d : s.h_lists) if (d.n_ids) { id_lo = std::min<uint64_t>(id_lo, d.first_id); id_hi = std::max<uint64_t>(id_hi, d.last_id); }
    const uint64_t id_span = id_hi >= id_lo ? std::min<uint64_t>(id_hi - id_lo + 1, num_docs ? num_docs : ~0ull) : num_docs;
    const uint64_t thr = std::max<uint64_t>((uint64_t)ctx->kw_iddir_min_ids, id_span / (uint64_t)ctx->kw_iddir_density_div);
    std::vector<uint32_t> want;
    for (size_t h = 0; h < s.h_lists.size(); h++) if (s.h_lists[h].n_ids >= thr && s.h_lists[h].n_blocks) want.push_back((uint32_t)h);
    if (want.empty()) return TSGPU_OK;
    std::sort(want.begin(), want.end(), [&](uint32_t a, uint32_t b) { return s.h_lists[a].n_ids != s.h_lists[b].n_ids ? s.h_lists[a].n_ids > s.h_lists[b].n_ids : a < b; });
    std::shared_ptr<IdDirPool> pool = (cur && touched && cur->dir_pool && num_docs <= cur->dir_pool->cap_ids) ? cur->dir_pool : nullptr;
    const bool reuse = (bool)pool;
    if (!pool) {
        const uint64_t cap = ((uint64_t)num_docs + num_docs / 8 + 1024 + 2047) / 2048 * 2048;
        if (cap > 0xFFFFF000ull) return TSGPU_OK;
        const uint64_t slot_bytes = cap / 32 * sizeof(uint2);
        const uint64_t n_slots = std::min<uint64_t>((uint64_t)ctx->kw_iddir_budget_mb * (1ull << 20) / slot_bytes, 2 * want.size() + 8);
        if (n_slots == 0) return TSGPU_OK;
        pool = std::make_shared<IdDirPool>();
        pool->bin = ctx->retire_bin;
        if (pool->buf.reserve(n_slots * slot_bytes) != TSGPU_OK) { tls_error().clear(); return TSGPU_OK; }
        pool->cap_ids = (uint32_t)cap; pool->slot_entries = (uint32_t)(cap / 32); pool->n_slots = (uint32_t)n_slots;
        for (uint32_t i = (uint32_t)n_slots; i-- > 0;) pool->free_slots.push_back(i);
    }
    if (want.size() > (size_t)pool->n_slots / 2 + 4) want.resize((size_t)pool->n_slots / 2 + 4);     // (a tight budget: the longest lists, with room to double-buffer them)
    std::vector<uint8_t> is_touched;
    if (touched) { is_touched.assign(s.h_lists.size(), 0); for (uint32_t h : *touched) if (h < is_touched.size()) is_touched[h] = 1; }
    s.dir_of.assign(s.h_lists.size(), nullptr);
    std::vector<IdDirJob> jobs;
    uint64_t total_blocks = 0;
    for (uint32_t h : want) {
        if (reuse && h < cur->dir_of.size() && cur->dir_of[h] && !is_touched[h]) { s.dir_of[h] = cur->dir_of[h]; continue; }
        uint32_t slot;
        if (!pool->take(slot)) continue;                 // every slot is held by snapshots that searches still use: this list goes without
        s.dir_of[h] = std::make_shared<IdDirRef>(pool, slot);
        const ListDesc& d = s.h_lists[h];
        jobs.push_back({d.ids_base, total_blocks, d.blk_base, d.n_blocks, slot, 0u});
        total_blocks += d.n_blocks;
    }
    if (!jobs.empty()) {
        if (total_blocks > 0x7FFFFFFFull) { s.dir_of.clear(); return TSGPU_OK; }
        DevBuf d_jobs;
        int rc = d_jobs.reserve(jobs.size() * sizeof(IdDirJob));
        if (rc != TSGPU_OK) { s.dir_of.clear(); tls_error().clear(); return TSGPU_OK; }
        hipError_t e = hipMemcpy(d_jobs.p, jobs.data(), jobs.size() * sizeof(IdDirJob), hipMemcpyHostToDevice);
        for (size_t i = 0; i < jobs.size() && e == hipSuccess; i++)
            e = hipMemsetAsync(pool->buf.as<uint2>() + (size_t)jobs[i].slot * pool->slot_entries, 0, (size_t)pool->slot_entries * sizeof(uint2), ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(index_iddir_build_kernel, dim3((uint32_t)total_blocks), dim3(256), 0, ctx->stream, d_jobs.as<IdDirJob>(), (uint32_t)jobs.size(), s.ar->blk_ids.as<BlockIds>(),
                               s.ar->ids_payload.as<uint32_t>(), pool->buf.as<uint2>(), pool->slot_entries, pool->cap_ids);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        }
        d_jobs.release();
        if (e != hipSuccess) { s.dir_of.clear(); return fail(TSGPU_ERR_DEVICE, std::string("tsgpu_commit: id directories: ") + hipGetErrorString(e)); }
        ctx->kw_iddir_built += jobs.size();
    }
    s.dir_pool = pool;
    for (size_t h = 0; h < s.dir_of.size(); h++) if (s.dir_of[h]) s.h_lists[h].dir_slot = s.dir_of[h]->slot + 1;
    return TSGPU_OK;
}

// everything re-packed into fresh arenas (first commit, compaction, or the tails ran out of room)
int commit_full(tsgpu_ctx* ctx) {
    std::shared_ptr<Snapshot> sp = std::make_shared<Snapshot>();
    sp->bin = ctx->retire_bin;
    Snapshot& s = *sp;
    std::shared_ptr<ArenaSet> ar = std::make_shared<ArenaSet>();
    ar->bin = ctx->retire_bin;
    std::shared_ptr<HandleMaps> maps = std::make_shared<HandleMaps>();
    std::vector<std::pair<uint64_t, TermHost*>> order;
    for (auto& f : ctx->fields) {
        s.field_is_array[f.first] = f.second.is_array;
        for (auto& t : f.second.terms) { flush_open(t.second); repack_if_fragmented(t.second); repack_if_wasteful(t.second); order.emplace_back(((uint64_t)f.first << 32) | t.first, &t.second); }
    }
    std::sort(order.begin(), order.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    uint64_t n_blocks = 0, n_slots = 0, n_pw = 0, n_idw = 0;
    uint32_t max_id = 0;
    for (auto& e : order) {
        n_slots += desc_capacity(e.second->pl.blk_last.size());
        for (const BlockMeta& m : e.second->pl.blk_meta) { n_blocks++; n_idw += ids_words(m); n_pw += pay_words(m); }
        max_id = std::max(max_id, e.second->pl.desc.last_id);
    }
    // capacity = what is there + room for incremental commits (half as much again, at least index_min_slack_words); the descriptor
    // arenas are small (52 B per block): they get room for several re-writes of every list
    const uint64_t min_slack = ctx->index_min_slack_words;
    ar->cap_blocks = n_slots + std::max<uint64_t>(2 * n_slots, std::max<uint64_t>(min_slack / 16, 16));
    if (ar->cap_blocks >= 0xFFFFFFF0ull) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_commit: more than 2^32 posting blocks");
    ar->cap_idw = n_idw + std::max<uint64_t>(n_idw / 2, min_slack) + 4;
    ar->cap_pw = n_pw + std::max<uint64_t>(n_pw / 2, min_slack) + 4;
    int rc;
    if ((rc = ar->blk_last.reserve(ar->cap_blocks * 4)) || (rc = ar->blk_ids.reserve(ar->cap_blocks * sizeof(BlockIds))) || (rc = ar->blk_meta.reserve(ar->cap_blocks * sizeof(BlockMeta))) ||
        (rc = ar->ids_payload.reserve((ar->cap_idw + KW_TILE_OVERREAD_WORDS) * 4)) || (rc = ar->payload.reserve(ar->cap_pw * 4)))
        return rc;
    std::vector<uint32_t> h_last; std::vector<BlockIds> h_bids; std::vector<BlockMeta> h_bmeta;
    h_last.reserve(n_slots); h_bids.reserve(n_slots); h_bmeta.reserve(n_slots);
    std::vector<uint32_t> h_idw(n_idw + 4, 0u), h_pw(n_pw + 4, 0u);
    uint64_t ipos = 0, ppos = 0;
    s.h_lists.reserve(order.size());
    for (auto& e : order) {
        TermHost& t = *e.second;
        const uint64_t ids_base = ipos, pay_base = ppos;
        for (size_t b = 0; b < t.pl.blk_last.size(); b++) {
            const BlockMeta& m = t.pl.blk_meta[b];
            const uint32_t iw = ids_words(m), pw = pay_words(m);
            memcpy(h_idw.data() + ipos, t.pl.ids_payload.data() + m.ids_woff, (size_t)iw * 4);
            memcpy(h_pw.data() + ppos, t.pl.payload.data() + m.oi_woff, (size_t)pw * 4);
            t.dev[b].idw = ipos; t.dev[b].pw = ppos;
            ipos += iw; ppos += pw;
        }
        if (ipos - ids_base > 0xFFFFFFF0ull || ppos - pay_base > 0xFFFFFFF0ull) return fail(TSGPU_ERR_UNSUPPORTED, "tsgpu_commit: a posting list beyond 16 GiB");
        const size_t nb = t.pl.blk_last.size();
        t.has_breaks = false;
        t.d_blk_base = h_last.size(); t.d_blk_cap = desc_capacity(nb); t.d_blk_n = (uint32_t)nb;
        make_descriptors(t, 0, h_last, h_bids, h_bmeta, ids_base, pay_base);
        h_last.resize(t.d_blk_base + t.d_blk_cap, 0u); h_bids.resize(t.d_blk_base + t.d_blk_cap); h_bmeta.resize(t.d_blk_base + t.d_blk_cap);     // spare entries
        t.handle = (uint32_t)s.h_lists.size();
        t.dirty = false; t.queued = false; t.desc_rewrite = false;
        t.dev_idw = t.dev_pw = 0;
        for (const BlockMeta& m : t.pl.blk_meta) { t.dev_idw += ids_words(m); t.dev_pw += pay_words(m); }
        maps->handle_of[e.first] = t.handle;
        s.h_lists.push_back(list_desc(t, ids_base, pay_base, t.d_blk_base));
    }
    if ((rc = s.lists.reserve(std::max<size_t>(s.h_lists.size() + s.h_lists.size() / 4 + 64, 64) * sizeof(ListDesc)))) return rc;
    if (!h_bids.empty()) TSGPU_HIP_TRY(hipMemcpy(ar->blk_ids.p, h_bids.data(), h_bids.size() * sizeof(BlockIds), hipMemcpyHostToDevice));
    TSGPU_HIP_TRY(hipMemcpy(ar->ids_payload.p, h_idw.data(), h_idw.size() * 4, hipMemcpyHostToDevice));
    if (!h_last.empty()) TSGPU_HIP_TRY(hipMemcpy(ar->blk_last.p, h_last.data(), h_last.size() * 4, hipMemcpyHostToDevice));
    if (!h_bmeta.empty()) TSGPU_HIP_TRY(hipMemcpy(ar->blk_meta.p, h_bmeta.data(), h_bmeta.size() * sizeof(BlockMeta), hipMemcpyHostToDevice));
    TSGPU_HIP_TRY(hipMemcpy(ar->payload.p, h_pw.data(), h_pw.size() * 4, hipMemcpyHostToDevice));
    s.ar = ar;
    if ((rc = build_id_directories(ctx, s, nullptr, nullptr, ctx->num_docs_set ? ctx->num_docs : std::max(ctx->num_docs, order.empty() ? 0u : max_id + 1)))) return rc;
    if (!s.h_lists.empty()) TSGPU_HIP_TRY(hipMemcpy(s.lists.p, s.h_lists.data(), s.h_lists.size() * sizeof(ListDesc), hipMemcpyHostToDevice));
    ar->used_blocks = ar->live_blocks = n_slots; ar->used_idw = ar->live_idw = n_idw; ar->used_pw = ar->live_pw = n_pw;
    ctx->erased_dev_idw = ctx->erased_dev_pw = 0;
    maps->rebuild_dense();
    maps->bin = ctx->retire_bin;
    maps->upload_dense();
    s.ar = ar; s.maps = maps;
    s.bytes = ar->bytes() + s.lists.cap;
    if (!ctx->num_docs_set) ctx->num_docs = std::max(ctx->num_docs, order.empty() ? 0u : max_id + 1);
    s.num_docs = ctx->num_docs;
    ctx->commit_last_uploaded_bytes = (n_idw + n_pw) * 4 + n_slots * (4 + sizeof(BlockIds) + sizeof(BlockMeta)) + s.h_lists.size() * sizeof(ListDesc);
    ctx->commit_full_count++;
    std::atomic_store(&ctx->snap, std::shared_ptr<const Snapshot>(sp));      // publish
    return TSGPU_OK;
}

// only what changed: the new / re-written blocks' words go to the arena tails; descriptors of blocks appended behind a list's published
// blocks go into the list's spare descriptor entries (scatter), lists that changed otherwise get their descriptor arrays re-written at
// the tail; then a new descriptor table is published. Returns TSGPU_OK, an error, or -1 = "does not fit: take the full path".
int commit_incremental(tsgpu_ctx* ctx, const std::shared_ptr<const Snapshot>& cur) {
    std::shared_ptr<ArenaSet> ar = cur->ar;
    const std::vector<uint64_t>& keys = ctx->dirty_terms;       // sorted, unique (tsgpu_commit)
    std::shared_ptr<Snapshot> sp = std::make_shared<Snapshot>();
    sp->bin = ctx->retire_bin;
    Snapshot& s = *sp;
    s.h_lists = cur->h_lists;
    std::shared_ptr<HandleMaps> new_maps;                        // copy-on-write: only when a term appears or disappears
    auto maps_rw = [&]() -> HandleMaps& { if (!new_maps) new_maps = std::make_shared<HandleMaps>(*cur->maps); return *new_maps; };
    std::vector<uint32_t> st_idw, st_pw, st_last, sc_last;      // st_*: contiguous at the tails; sc_*: scattered into spare entries
    std::vector<BlockIds> st_bids, sc_bids;
    std::vector<BlockMeta> st_bmeta, sc_bmeta;
    std::vector<uint64_t> sc_dst;
    uint64_t dead_slots = 0, new_live_idw = 0, new_live_pw = 0, dead_idw = ctx->erased_dev_idw, dead_pw = ctx->erased_dev_pw;
    uint32_t max_id = 0;
    struct Touched { TermHost* t; uint64_t key; };
    std::vector<Touched> touched;
    for (uint64_t key : keys) {
        const uint32_t field = (uint32_t)(key >> 32), term = (uint32_t)key;
        auto fit = ctx->fields.find(field);
        TermHost* t = nullptr;
        if (fit != ctx->fields.end()) { auto tit = fit->second.terms.find(term); if (tit != fit->second.terms.end()) t = &tit->second; }
        if (!t) {                                                // the term is gone: its handle leaves the maps, its blocks become garbage
            auto hit = cur->maps->handle_of.find(key);
            if (hit != cur->maps->handle_of.end()) {
                dead_slots += s.h_lists[hit->second].n_blocks;
                maps_rw().handle_of.erase(key);
                s.h_lists[hit->second].n_ids = 0; s.h_lists[hit->second].n_blocks = 0;
            }
            continue;
        }
        flush_open(*t);
        repack_if_fragmented(*t);
        repack_if_wasteful(*t);
        touched.push_back({t, key});
    }
    // place the not-yet-uploaded blocks at the tails
    uint64_t ipos = ar->used_idw, ppos = ar->used_pw;
    for (auto& tc : touched) {
        TermHost& t = *tc.t;
        for (int attempt = 0; attempt < 2; attempt++) {
            uint64_t lo_i = ~0ull, hi_i = 0, lo_p = ~0ull, hi_p = 0, ip = ipos, pp = ppos;
            for (size_t b = 0; b < t.pl.blk_last.size(); b++) {
                const BlockMeta& m = t.pl.blk_meta[b];
                uint64_t pi = t.dev[b].idw, pq = t.dev[b].pw;
                if (pi == TermHost::NOPOS) { pi = ip; pq = pp; ip += ids_words(m); pp += pay_words(m); }
                lo_i = std::min(lo_i, pi); hi_i = std::max(hi_i, pi + ids_words(m));
                lo_p = std::min(lo_p, pq); hi_p = std::max(hi_p, pq + pay_words(m));
            }
            if (hi_i - lo_i <= 0xFFFFFFF0ull && hi_p - lo_p <= 0xFFFFFFF0ull) break;
            for (auto& d : t.dev) d = TermHost::BlockPos();      // the list would span more than 16 GiB of arena: move all of it
            t.desc_rewrite = true;
        }
        for (size_t b = 0; b < t.pl.blk_last.size(); b++) {
            if (t.dev[b].idw != TermHost::NOPOS) continue;
            const BlockMeta& m = t.pl.blk_meta[b];
            const uint32_t iw = ids_words(m), pw = pay_words(m);
            st_idw.insert(st_idw.end(), t.pl.ids_payload.begin() + m.ids_woff, t.pl.ids_payload.begin() + m.ids_woff + iw);
            st_pw.insert(st_pw.end(), t.pl.payload.begin() + m.oi_woff, t.pl.payload.begin() + m.oi_woff + pw);
            t.dev[b].idw = ipos; t.dev[b].pw = ppos;
            ipos += iw; ppos += pw;
        }
        // live words: this list's new version replaces its committed one (re-written and removed blocks stay behind as garbage)
        uint64_t li = 0, lp = 0;
        for (const BlockMeta& m : t.pl.blk_meta) { li += ids_words(m); lp += pay_words(m); }
        new_live_idw += li; new_live_pw += lp;
        dead_idw += t.dev_idw; dead_pw += t.dev_pw;
    }
    // COMPACTION: when the arenas would hold more garbage than live words after this commit, re-pack everything instead (the full
    // path; also restores the tail room and clears every LIST_HAS_BREAKS). Small indexes never bother (index_compact_min_words).
    {
        const uint64_t live_i = ar->live_idw + new_live_idw - std::min(ar->live_idw + new_live_idw, dead_idw), live_p = ar->live_pw + new_live_pw - std::min(ar->live_pw + new_live_pw, dead_pw);
        const uint64_t gar_i = ipos - std::min(ipos, live_i), gar_p = ppos - std::min(ppos, live_p);
        if ((gar_i > live_i && gar_i > ctx->index_compact_min_words) || (gar_p > live_p && gar_p > ctx->index_compact_min_words)) { ctx->commit_compactions++; return -1; }
    }
    // descriptors
    const uint64_t blk0 = ar->used_blocks;
    struct Placed { TermHost* t; uint64_t base; uint32_t cap; };
    std::vector<Placed> placed;
    for (auto& tc : touched) {
        TermHost& t = *tc.t;
        const size_t nb = t.pl.blk_last.size();
        const bool known = t.handle != 0xFFFFFFFFu && t.handle < s.h_lists.size() && cur->maps->handle_of.count(tc.key);
        uint64_t ids_base, pay_base;
        const bool in_place = known && !t.desc_rewrite && t.d_blk_base != TermHost::NOPOS && nb <= t.d_blk_cap && nb >= t.d_blk_n;
        if (in_place) {
            // published entries stay as they are: the bases must not move
            ids_base = s.h_lists[t.handle].ids_base; pay_base = s.h_lists[t.handle].payload_base;
            bool fits = true;
            for (size_t b = t.d_blk_n; b < nb; b++) fits = fits && t.dev[b].idw >= ids_base && t.dev[b].idw + ids_words(t.pl.blk_meta[b]) - ids_base <= 0xFFFFFFF0ull &&
                                                           t.dev[b].pw >= pay_base && t.dev[b].pw + pay_words(t.pl.blk_meta[b]) - pay_base <= 0xFFFFFFF0ull;
            if (fits) {
                const size_t at = sc_last.size();
                make_descriptors(t, t.d_blk_n, sc_last, sc_bids, sc_bmeta, ids_base, pay_base);
                for (size_t i = at; i < sc_last.size(); i++) sc_dst.push_back(t.d_blk_base + t.d_blk_n + (i - at));
                s.h_lists[t.handle] = list_desc(t, ids_base, pay_base, t.d_blk_base);
                placed.push_back({&t, t.d_blk_base, t.d_blk_cap});
                max_id = std::max(max_id, t.pl.desc.last_id);
                continue;
            }
        }
        ids_base = pay_base = ~0ull;
        for (auto& d : t.dev) { ids_base = std::min(ids_base, d.idw); pay_base = std::min(pay_base, d.pw); }
        const uint64_t base = blk0 + st_last.size();
        const uint32_t cap = desc_capacity(nb);
        t.has_breaks = false;
        make_descriptors(t, 0, st_last, st_bids, st_bmeta, ids_base, pay_base);
        st_last.resize(base - blk0 + cap, 0u); st_bids.resize(base - blk0 + cap); st_bmeta.resize(base - blk0 + cap);
        if (known) { dead_slots += t.d_blk_cap; s.h_lists[t.handle] = list_desc(t, ids_base, pay_base, base); }
        else {
            t.handle = (uint32_t)s.h_lists.size();
            s.h_lists.push_back(list_desc(t, ids_base, pay_base, base));
            maps_rw().handle_of[tc.key] = t.handle;
        }
        placed.push_back({&t, base, cap});
        max_id = std::max(max_id, t.pl.desc.last_id);
    }
    if (blk0 + st_last.size() > std::min<uint64_t>(ar->cap_blocks, 0xFFFFFFF0ull) || ipos + 4 > ar->cap_idw || ppos + 4 > ar->cap_pw) return -1;
    int rc;
    if ((rc = s.lists.reserve(std::max<size_t>(s.h_lists.size() + 64, 64) * sizeof(ListDesc)))) return rc;
    if (!st_idw.empty()) TSGPU_HIP_TRY(hipMemcpy(ar->ids_payload.as<uint32_t>() + ar->used_idw, st_idw.data(), st_idw.size() * 4, hipMemcpyHostToDevice));
    if (!st_pw.empty()) TSGPU_HIP_TRY(hipMemcpy(ar->payload.as<uint32_t>() + ar->used_pw, st_pw.data(), st_pw.size() * 4, hipMemcpyHostToDevice));
    if (!st_last.empty()) {
        TSGPU_HIP_TRY(hipMemcpy(ar->blk_last.as<uint32_t>() + blk0, st_last.data(), st_last.size() * 4, hipMemcpyHostToDevice));
        TSGPU_HIP_TRY(hipMemcpy(ar->blk_ids.as<BlockIds>() + blk0, st_bids.data(), st_bids.size() * sizeof(BlockIds), hipMemcpyHostToDevice));
        TSGPU_HIP_TRY(hipMemcpy(ar->blk_meta.as<BlockMeta>() + blk0, st_bmeta.data(), st_bmeta.size() * sizeof(BlockMeta), hipMemcpyHostToDevice));
    }
    if (!sc_dst.empty()) {
        const size_t n = sc_dst.size();
        DevBuf d_dst, d_last, d_bids, d_bmeta;
        auto drop = [&]() { d_dst.release(); d_last.release(); d_bids.release(); d_bmeta.release(); };
        if ((rc = d_dst.reserve(n * 8)) || (rc = d_last.reserve(n * 4)) || (rc = d_bids.reserve(n * sizeof(BlockIds))) || (rc = d_bmeta.reserve(n * sizeof(BlockMeta)))) { drop(); return rc; }
        hipError_t e = hipMemcpy(d_dst.p, sc_dst.data(), n * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_last.p, sc_last.data(), n * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_bids.p, sc_bids.data(), n * sizeof(BlockIds), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_bmeta.p, sc_bmeta.data(), n * sizeof(BlockMeta), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(index_desc_scatter_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_dst.as<uint64_t>(), d_last.as<uint32_t>(), d_bids.as<BlockIds>(),
                               d_bmeta.as<BlockMeta>(), (uint32_t)n, ar->blk_last.as<uint32_t>(), ar->blk_ids.as<BlockIds>(), ar->blk_meta.as<BlockMeta>());
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        }
        drop();
        if (e != hipSuccess) return fail(TSGPU_ERR_DEVICE, std::string("tsgpu_commit: descriptor scatter: ") + hipGetErrorString(e));
    }
    {
        std::vector<uint32_t> touched_handles;
        for (auto& pc : placed) touched_handles.push_back(pc.t->handle);
        s.ar = ar;
        if ((rc = build_id_directories(ctx, s, cur.get(), &touched_handles, ctx->num_docs_set ? ctx->num_docs : std::max(ctx->num_docs, max_id + 1)))) return rc;
    }
    TSGPU_HIP_TRY(hipMemcpy(s.lists.p, s.h_lists.data(), s.h_lists.size() * sizeof(ListDesc), hipMemcpyHostToDevice));
    ar->used_idw = ipos; ar->used_pw = ppos; ar->used_blocks = blk0 + st_last.size();
    ar->live_blocks = ar->live_blocks + st_last.size() - std::min<uint64_t>(dead_slots, ar->live_blocks + st_last.size());
    ar->live_idw = ar->live_idw + new_live_idw - std::min(ar->live_idw + new_live_idw, dead_idw);
    ar->live_pw = ar->live_pw + new_live_pw - std::min(ar->live_pw + new_live_pw, dead_pw);
    ctx->erased_dev_idw = ctx->erased_dev_pw = 0;
    for (auto& tc : touched) {
        TermHost& t = *tc.t;
        t.dev_idw = t.dev_pw = 0;
        for (const BlockMeta& m : t.pl.blk_meta) { t.dev_idw += ids_words(m); t.dev_pw += pay_words(m); }
    }
    for (auto& pc : placed) { TermHost& t = *pc.t; t.d_blk_base = pc.base; t.d_blk_cap = pc.cap; t.d_blk_n = (uint32_t)t.pl.blk_last.size(); t.dirty = false; t.queued = false; t.desc_rewrite = false; }
    if (new_maps) { new_maps->rebuild_dense(); new_maps->bin = ctx->retire_bin; new_maps->upload_dense(); s.maps = new_maps; } else s.maps = cur->maps;
    s.ar = ar;
    s.field_is_array = cur->field_is_array;
    s.bytes = ar->bytes() + s.lists.cap;
    if (!ctx->num_docs_set) ctx->num_docs = std::max(ctx->num_docs, max_id + 1);
    s.num_docs = ctx->num_docs;
    ctx->commit_last_uploaded_bytes = (st_idw.size() + st_pw.size()) * 4 + (st_last.size() + sc_dst.size()) * (4 + sizeof(BlockIds) + sizeof(BlockMeta)) + sc_dst.size() * 8 +
                                      s.h_lists.size() * sizeof(ListDesc);
    ctx->commit_incremental_count++;
    std::atomic_store(&ctx->snap, std::shared_ptr<const Snapshot>(sp));      // publish
    return TSGPU_OK;
}

}  // namespace

extern "C" {

// Publishes every pending posting-list change as ONE new immutable snapshot (RCU): a search that started on the previous snapshot
// keeps it alive until it returns, a failing commit leaves the previous snapshot in place, and searches never wait on a commit.
// Cost: the changed blocks' words + the descriptor table (48 B per list, copied and uploaded whole) + a copy of the term map when a
// term appeared or disappeared — see the header of this file; O(index) only for the first commit and for compactions.
int tsgpu_commit(tsgpu_ctx* ctx) {
    if (!ctx) return fail(TSGPU_ERR_INVALID, "ctx is NULL");
    std::lock_guard<std::mutex> lk(ctx->mu);
    (void)hipSetDevice(ctx->device);
    const uint64_t t0 = wall_us();
    ctx->retire_bin->drain();                        // buffers of snapshots that searches have let go of since the last commit
    deferred_frees().drain();                        // ... and scratch buffers that outgrew themselves while serving queries
    try {
        const std::shared_ptr<const Snapshot> cur = ctx->snapshot();
        int rc = -1;
        std::sort(ctx->dirty_terms.begin(), ctx->dirty_terms.end());
        ctx->dirty_terms.erase(std::unique(ctx->dirty_terms.begin(), ctx->dirty_terms.end()), ctx->dirty_terms.end());
        const bool can_inc = cur && cur->ar && cur->maps && !ctx->commit_force_full && !ctx->dirty_fields;
        if (can_inc) rc = commit_incremental(ctx, cur);
        if (rc == -1) rc = commit_full(ctx);
        if (rc != TSGPU_OK) {
            // A failed attempt (either path) has already written arena positions / handles / descriptor bases into the host-side
            // lists for blocks that never reached the device. Nothing published refers to them (the previous snapshot is immutable
            // and stays in place), but an INCREMENTAL retry would skip those blocks as "already uploaded". The next commit therefore
            // takes the full path, which derives every position from the host lists alone.
            ctx->commit_force_full = true;
            ctx->commit_failed_count++;
            return rc;
        }
        ctx->dirty_terms.clear();
        ctx->dirty_fields = false;
        ctx->commit_force_full = false;
        ctx->dirty = false;
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_commit: host allocation failed"); }
    ctx->retire_bin->drain();                        // (the snapshot this commit replaced, unless a search still holds it)
    ctx->commit_last_us = wall_us() - t0;
    return ok();
}

uint32_t tsgpu_term_num_ids(tsgpu_ctx* ctx, uint32_t field_id, uint32_t term_id) {
    if (!ctx) return 0;
    const std::shared_ptr<const Snapshot> sn = ctx->snapshot();
    const uint32_t h = sn->find_handle(field_id, term_id);
    return h == 0xFFFFFFFFu ? 0 : sn->h_lists[h].n_ids;
}

int tsgpu_term_download(tsgpu_ctx* ctx, uint32_t field_id, uint32_t term_id, uint32_t* ids, uint32_t* offset_index, uint32_t* offsets,
                        uint32_t* n_offsets) {
    if (!ctx) return fail(TSGPU_ERR_INVALID, "ctx is NULL");
    (void)hipSetDevice(ctx->device);
    const std::shared_ptr<const Snapshot> sn = ctx->snapshot();     // the committed snapshot: pending (uncommitted) changes are not visible here
    const uint32_t h = sn->find_handle(field_id, term_id);
    if (h == 0xFFFFFFFFu) return fail(TSGPU_ERR_NOT_FOUND, "tsgpu_term_download: term not in the committed snapshot");
    const ListDesc d = sn->h_lists[h];
    try {
        std::vector<BlockMeta> meta(d.n_blocks);
        TSGPU_HIP_TRY(hipMemcpy(meta.data(), sn->ar->blk_meta.as<BlockMeta>() + d.blk_base, (size_t)d.n_blocks * sizeof(BlockMeta), hipMemcpyDeviceToHost));
        std::vector<uint32_t> a, b, c, ia, ib, ic, iw, pw;
        for (uint32_t blk = 0; blk < d.n_blocks; blk++) {          // block by block: a list's blocks need not be neighbours in the arenas
            const BlockMeta& m = meta[blk];
            iw.assign(ids_words(m) + 2, 0u); pw.assign(pay_words(m) + 2, 0u);
            TSGPU_HIP_TRY(hipMemcpy(iw.data(), sn->ar->ids_payload.as<uint32_t>() + d.ids_base + m.ids_woff, (size_t)ids_words(m) * 4, hipMemcpyDeviceToHost));
            TSGPU_HIP_TRY(hipMemcpy(pw.data(), sn->ar->payload.as<uint32_t>() + d.payload_base + m.oi_woff, (size_t)pay_words(m) * 4, hipMemcpyDeviceToHost));
            unpack_block(m, iw.data(), pw.data(), pw.data() + (m.off_woff - m.oi_woff), ia, ib, ic);
            const uint32_t base = (uint32_t)c.size();
            a.insert(a.end(), ia.begin(), ia.end());
            for (uint32_t x : ib) b.push_back(base + x);
            c.insert(c.end(), ic.begin(), ic.end());
        }
        if (n_offsets) *n_offsets = (uint32_t)c.size();
        if (ids) std::copy(a.begin(), a.end(), ids);
        if (offset_index) std::copy(b.begin(), b.end(), offset_index);
        if (offsets) std::copy(c.begin(), c.end(), offsets);
    } catch (const std::bad_alloc&) { return fail(TSGPU_ERR_NO_MEMORY, "tsgpu_term_download: host allocation failed"); }
    return ok();
}

}  // extern "C"
