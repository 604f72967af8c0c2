// tsgpu_format.h — HBM layout of the keyword index mirror, shared by the host packer and the gfx950 kernels.
//
// The reference keeps, per token, a chain of posting_list_t::block_t (include/posting_list.h:56-77): three
// FOR-compressed arrays per block of <=256 docs — ids (sorted), offset_index (start of each doc's run),
// offsets (token positions+1, 0 = "last token" flag; src/index.cpp:1323-1348) — reached through a
// std::map<last_id, block*> (posting_list.h:130). That byte format is private to the reference (never
// persisted), so the mirror is free to choose a layout that suits 64-wide wavefronts:
//
//   * every list is re-blocked into exactly TSGPU_BLOCK_IDS (256) ids per block (last block partial), so a
//     posting position p maps to (block p>>8, slot p&255) without a per-block length lookup;
//   * the std::map skip index becomes one contiguous u32 array blk_last[] per list (binary-searchable with
//     coalesced / broadcast loads) plus a 32-byte BlockMeta record per block;
//   * the three arrays stay frame-of-reference bit-packed (same arithmetic as libfor: value-base in `bits`
//     bits, LSB-first), but in 32-bit words, each array padded to a whole word + one guard word so a lane
//     can extract any element with two dword loads and a funnel shift — no per-block decode/alloc;
//   * the doc ids live in their OWN arena (ids_payload + a 16-byte BlockIds record per block), apart from the
//     offset_index/offsets arena: the intersection streams ids only, so a cache line it fetches holds nothing
//     but ids (with one interleaved arena ~57% of every fetched line was offsets the intersection never reads);
//   * all lists of a snapshot live in five arenas (blk_last, blk_ids, blk_meta, ids_payload, payload);
//   * blocks need not be full (a posting position is block * 256 + slot whatever the fill), and a block re-written by an incremental
//     commit lives at the arena tail: only the block's words and the list's (small) descriptor arrays are uploaded, never the list.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(TSGPU_HIP_EMU)
#define TSGPU_HD __host__ __device__
#else
#define TSGPU_HD
#endif

namespace tsgpu {

static const uint32_t BLOCK_IDS = 256;

struct BlockIds {            // 16 bytes: everything the intersection needs to decode a block's ids
    uint32_t first_id;       // FOR base of ids (= ids[0])
    uint32_t last_id;        // = blk_last[]
    uint32_t ids_woff;       // word offset relative to ListDesc::ids_base (ids_payload arena)
    uint32_t n_ids_bits;     // n_ids (1..256) | ids_bits << 16
};

struct BlockMeta {           // 32 bytes: offsets side of a block (scoring only)
    uint32_t first_id;       // FOR base of ids (= ids[0])
    uint32_t ids_woff;       // relative to ListDesc::ids_base (ids_payload arena); other word offsets: ListDesc::payload_base
    uint32_t oi_woff;        // offset_index, FOR base 0 (offset_index[0] == 0 inside a block)
    uint32_t off_woff;       // offsets, FOR base off_base
    uint32_t n_off;          // total offsets stored in this block
    uint32_t off_base;       // min offset value in the block
    uint16_t n_ids;          // 1..256
    uint8_t ids_bits;
    uint8_t oi_bits;
    uint8_t off_bits;
    uint8_t pad[3];
};

// words the find kernel's LDS-DMA tile fill may read past the end of a run (kw_find2.hip.h): the ids arena is allocated this much larger
static const uint32_t KW_TILE_OVERREAD_WORDS = 2048;

struct ListDesc {            // 48 bytes
    uint64_t payload_base;   // word index into the payload arena (offset_index + offsets)
    uint64_t ids_base;       // word index into the ids_payload arena
    uint32_t blk_base;       // index of the list's first block in blk_last[] / blk_meta[]
    uint32_t n_blocks;
    uint32_t n_ids;
    uint32_t first_id;
    uint32_t last_id;
    uint32_t n_off;          // total offsets of the list (for the algorithmic-bytes accounting)
    uint32_t flags;          // LIST_HAS_BREAKS: some block's ids do not follow the previous block's in the arena (blocks re-written by an
                             // incremental commit live at the arena tail until the next compaction); a run of blocks is ONE coalesced
                             // range only where it has no break
    uint32_t dir_slot;       // 0: none; else 1 + the slot of the list's ID DIRECTORY (below) in the snapshot's directory pool
};
static const uint32_t LIST_HAS_BREAKS = 1u;

// ID DIRECTORY of a long list (n_ids >= num_docs / 64 by default): one 8-byte entry per 32 consecutive doc ids,
//   {pos, bits}: bits = which of the ids [32w, 32w + 32) the list holds; pos = posting position (block * 256 + slot) of the lowest one.
// "Is id x in the list, and where?" — what every probe of the third.. lists asks (posting_list_t::iterator_t::skip_to + the equality test of
// or_iterator_t::intersect, /root/reference/src/or_iterator.cpp:20-79) — is ONE load: bit x & 31 of entry x >> 5, position = pos + popcount of
// the lower bits (the ids of one entry are consecutive slots of one block). Entries whose ids straddle two blocks carry IDDIR_SPLIT in pos:
// those (one per block, ~0.4 % of a list's ids) and ids beyond the pool's range take the regular two-level search. 8 bytes x num_docs / 32
// per list (2.8 MB at 10M documents), built on the device by index_iddir_build_kernel at commit time.
static const uint32_t IDDIR_SPLIT = 0x80000000u;

TSGPU_HD static inline uint32_t required_bits(uint32_t v) {   // include/array_base.h:23-25
    return v == 0 ? 0u : 32u - (uint32_t)__builtin_clz(v);
}

// words needed for n values of `bits` bits, + 1 guard word (so unpack may always read word i+1)
TSGPU_HD static inline uint32_t packed_words(uint32_t n, uint32_t bits) {
    return (uint32_t)(((uint64_t)n * bits + 31) / 32) + 1;
}

// element idx of a bit-packed array (LSB-first in 32-bit words); bits in [0,32]
TSGPU_HD static inline uint32_t unpack_at(const uint32_t* __restrict__ w, uint32_t idx, uint32_t bits) {
    if (bits == 0) return 0;
    const uint64_t bitpos = (uint64_t)idx * bits;
    const uint64_t wi = bitpos >> 5;
    const uint32_t sh = (uint32_t)(bitpos & 31);
    const uint64_t two = (uint64_t)w[wi] | ((uint64_t)w[wi + 1] << 32);
    const uint64_t mask = bits >= 32 ? 0xFFFFFFFFull : ((1ull << bits) - 1ull);
    return (uint32_t)((two >> sh) & mask);
}

}  // namespace tsgpu
