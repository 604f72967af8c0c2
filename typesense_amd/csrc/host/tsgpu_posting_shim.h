// tsgpu_posting_shim.h — the index-build side of seam B1: turns the reference's IN-MEMORY posting structures into
// tsgpu_term_upsert calls. Called wherever the server mutates `search_index` under its unique_lock
// (Index::index_field_in_memory -> posting_t::upsert, src/index.cpp:1323-1395, src/posting.cpp:247-288), once per touched
// (field, token), followed by ONE tsgpu_commit per write batch.
//
// Written against the reference's own types through template parameters, so that it compiles inside the server with
//     PostingList = posting_list_t          (include/posting_list.h:48-213: root_block, block_t{ids, offset_index, offsets, next})
//     Compact     = compact_posting_list_t  (include/posting.h:14-44: length, ids_length, id_offsets[])
// and in this repository's tests with mock types of the same shape (tests/host_shims/). What it relies on:
//     block.ids / block.offset_index : sorted_array — uint32_t* uncompress() (caller delete[]s), uint32_t getLength()
//                                      (include/sorted_array.h, src/sorted_array.cpp:98-108)
//     block.offsets                  : array — same two methods (include/array.h, src/array.cpp:30-51)
//     block.next                     : the chain (posting_list.h:64)
//     offset_index[i]                : start of doc i's run inside THAT BLOCK's offsets (posting_list.cpp:857-860)
//     compact form                   : [num_offsets, off_1 .. off_n, id] per document (posting.h:21, src/posting.cpp:157-176)
// The decoded content is what tsgpu_term_upsert takes: ids ascending over the whole list, offset_index[i] = start of doc i's
// run in ONE concatenated offsets array.
#pragma once
#include <cstdint>
#include <vector>
#include "../../../include/tsgpu.h"

namespace tsgpu {

struct DecodedPosting {
    std::vector<uint32_t> ids, offset_index, offsets;
};

// posting_list_t -> flat arrays: walks the block chain from root_block (an empty root block = an empty list)
template <class PostingList>
inline void decode_posting_list(const PostingList& pl, DecodedPosting& out) {
    out.ids.clear(); out.offset_index.clear(); out.offsets.clear();
    for (auto* blk = &pl.root_block; blk != nullptr; blk = blk->next) {
        const uint32_t n = blk->ids.getLength();
        if (n == 0) continue;
        const uint32_t n_off = blk->offsets.getLength();
        uint32_t* ids = blk->ids.uncompress();
        uint32_t* oi = blk->offset_index.uncompress();
        uint32_t* off = blk->offsets.uncompress();
        const uint32_t base = (uint32_t)out.offsets.size();
        out.ids.insert(out.ids.end(), ids, ids + n);
        for (uint32_t i = 0; i < n; i++) out.offset_index.push_back(base + oi[i]);
        out.offsets.insert(out.offsets.end(), off, off + n_off);
        delete[] ids; delete[] oi; delete[] off;
    }
}

// compact_posting_list_t -> flat arrays (the same walk as compact_posting_list_t::to_full_posting_list, src/posting.cpp:157-176)
template <class Compact>
inline void decode_compact_posting(const Compact& c, DecodedPosting& out) {
    out.ids.clear(); out.offset_index.clear(); out.offsets.clear();
    size_t i = 0;
    while (i < c.length) {
        const uint32_t n = c.id_offsets[i++];
        out.offset_index.push_back((uint32_t)out.offsets.size());
        for (uint32_t j = 0; j < n; j++) out.offsets.push_back(c.id_offsets[i + j]);
        out.ids.push_back(c.id_offsets[i + n]);
        i += (size_t)n + 1;
    }
}

inline int upsert_decoded(tsgpu_ctx* ctx, uint32_t field_id, uint32_t term_id, const DecodedPosting& d) {
    return tsgpu_term_upsert(ctx, field_id, term_id, d.ids.data(), d.offset_index.data(), d.offsets.data(), (uint32_t)d.ids.size(), (uint32_t)d.offsets.size());
}

template <class PostingList>
inline int upsert_posting_list(tsgpu_ctx* ctx, uint32_t field_id, uint32_t term_id, const PostingList& pl) {
    DecodedPosting d;
    decode_posting_list(pl, d);
    return upsert_decoded(ctx, field_id, term_id, d);
}

template <class Compact>
inline int upsert_compact_posting(tsgpu_ctx* ctx, uint32_t field_id, uint32_t term_id, const Compact& c) {
    DecodedPosting d;
    decode_compact_posting(c, d);
    return upsert_decoded(ctx, field_id, term_id, d);
}

// the tagged pointer the ART leaves hold (IS_COMPACT_POSTING / COMPACT_POSTING_PTR / RAW_POSTING_PTR, include/posting.h:9-12);
// obj == nullptr removes the term
template <class PostingList, class Compact>
inline int upsert_posting(tsgpu_ctx* ctx, uint32_t field_id, uint32_t term_id, const void* obj) {
    if (obj == nullptr) return tsgpu_term_upsert(ctx, field_id, term_id, nullptr, nullptr, nullptr, 0, 0);
    if (((uintptr_t)obj & 1) != 0) return upsert_compact_posting(ctx, field_id, term_id, *(const Compact*)((uintptr_t)obj & ~(uintptr_t)1));
    return upsert_posting_list(ctx, field_id, term_id, *(const PostingList*)obj);
}

}  // namespace tsgpu
