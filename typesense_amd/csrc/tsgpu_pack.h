// tsgpu_pack.h — host-side builder of the HBM posting format (tsgpu_format.h) from the DECODED content of the
// reference's posting blocks (ids / offset_index / offsets, include/posting_list.h:56-77). Index-build path,
// runs on the host once per snapshot; the query path never touches it.
#pragma once
#include <vector>
#include <cstring>
#include <algorithm>
#include "tsgpu_format.h"

namespace tsgpu {

struct PackedList {
    ListDesc desc{};                    // payload_base / blk_base filled when placed into the arenas
    std::vector<uint32_t> blk_last;     // [n_blocks]
    std::vector<BlockIds> blk_ids;      // [n_blocks]
    std::vector<BlockMeta> blk_meta;    // [n_blocks]
    std::vector<uint32_t> ids_payload;  // packed doc ids
    std::vector<uint32_t> payload;      // packed offset_index + offsets
};

static inline void pack_bits(std::vector<uint32_t>& out, const uint32_t* vals, uint32_t n, uint32_t base, uint32_t bits) {
    const size_t start = out.size();
    out.resize(start + packed_words(n, bits), 0u);
    if (bits == 0) return;
    uint32_t* w = out.data() + start;
    for (uint32_t i = 0; i < n; i++) {
        const uint64_t v = (uint64_t)(vals[i] - base);
        const uint64_t bitpos = (uint64_t)i * bits;
        const uint64_t wi = bitpos >> 5;
        const uint32_t sh = (uint32_t)(bitpos & 31);
        w[wi] |= (uint32_t)(v << sh);
        if (sh + bits > 32) w[wi + 1] |= (uint32_t)(v >> (32 - sh));
    }
}

// one block: cnt (1..256) ascending ids, oi[i] = start of doc i's run inside offs[0 .. n_off); the packed words are APPENDED to the
// list's host arrays (a re-written block leaves its old words behind as garbage until the list is re-packed)
static inline void pack_block(PackedList& pl, const uint32_t* ids, const uint32_t* oi, const uint32_t* offs, uint32_t cnt, uint32_t n_off,
                              BlockIds& bi, BlockMeta& m) {
    memset(&m, 0, sizeof m);
    m.first_id = ids[0];
    m.n_ids = (uint16_t)cnt;
    m.n_off = n_off;
    m.ids_bits = required_bits(ids[cnt - 1] - ids[0]) <= 16 ? 16 : 32;   // fixed-width deltas: one aligned load per id in the kernels
    m.oi_bits = (uint8_t)required_bits(oi[cnt - 1]);
    uint32_t lo = 0xFFFFFFFFu, hi = 0;
    for (uint32_t i = 0; i < n_off; i++) { lo = std::min(lo, offs[i]); hi = std::max(hi, offs[i]); }
    if (n_off == 0) { lo = 0; hi = 0; }
    m.off_base = lo;
    m.off_bits = (uint8_t)required_bits(hi - lo);
    m.ids_woff = (uint32_t)pl.ids_payload.size();
    pack_bits(pl.ids_payload, ids, cnt, m.first_id, m.ids_bits);
    m.oi_woff = (uint32_t)pl.payload.size();
    pack_bits(pl.payload, oi, cnt, 0, m.oi_bits);
    m.off_woff = (uint32_t)pl.payload.size();
    pack_bits(pl.payload, offs, n_off, m.off_base, m.off_bits);
    bi.first_id = m.first_id; bi.last_id = ids[cnt - 1]; bi.ids_woff = m.ids_woff;
    bi.n_ids_bits = cnt | ((uint32_t)m.ids_bits << 16);
}

// ids ascending; offset_index[i] = index into offsets[] (relative to offsets[0]) of doc i's first offset
static inline PackedList pack_list(const uint32_t* ids, const uint64_t* offset_index, const uint32_t* offsets,
                                   uint32_t n_ids, uint64_t n_off) {
    PackedList pl;
    const uint32_t nb = (n_ids + BLOCK_IDS - 1) / BLOCK_IDS;
    pl.blk_last.resize(nb);
    pl.blk_ids.resize(nb);
    pl.blk_meta.resize(nb);
    std::vector<uint32_t> oi(BLOCK_IDS);
    for (uint32_t b = 0; b < nb; b++) {
        const uint32_t s = b * BLOCK_IDS;
        const uint32_t cnt = std::min(BLOCK_IDS, n_ids - s);
        const uint64_t o0 = offset_index[s];
        const uint64_t o1 = (s + cnt == n_ids) ? n_off : offset_index[s + cnt];
        for (uint32_t i = 0; i < cnt; i++) oi[i] = (uint32_t)(offset_index[s + i] - o0);
        pack_block(pl, ids + s, oi.data(), offsets + o0, cnt, (uint32_t)(o1 - o0), pl.blk_ids[b], pl.blk_meta[b]);
        pl.blk_last[b] = ids[s + cnt - 1];
    }
    pl.desc.n_blocks = nb;
    pl.desc.n_ids = n_ids;
    pl.desc.first_id = n_ids ? ids[0] : 0;
    pl.desc.last_id = n_ids ? ids[n_ids - 1] : 0;
    pl.desc.n_off = (uint32_t)std::min<uint64_t>(n_off, 0xFFFFFFFFull);
    return pl;
}

// one block back into its three arrays (oi relative to the block's offsets); ids_w / pay_w = where the block's words start
static inline void unpack_block(const BlockMeta& m, const uint32_t* ids_w, const uint32_t* oi_w, const uint32_t* off_w,
                                std::vector<uint32_t>& ids, std::vector<uint32_t>& oi, std::vector<uint32_t>& offs) {
    ids.resize(m.n_ids); oi.resize(m.n_ids); offs.resize(m.n_off);
    for (uint32_t i = 0; i < m.n_ids; i++) { ids[i] = m.first_id + unpack_at(ids_w, i, m.ids_bits); oi[i] = unpack_at(oi_w, i, m.oi_bits); }
    for (uint32_t i = 0; i < m.n_off; i++) offs[i] = m.off_base + unpack_at(off_w, i, m.off_bits);
}

// inverse (tests / tsgpu_term_download): decode a packed list back into the three flat arrays
static inline void unpack_list(const ListDesc& d, const uint32_t* blk_last, const BlockMeta* meta, const uint32_t* ids_payload, const uint32_t* payload,
                               std::vector<uint32_t>& ids, std::vector<uint32_t>& offset_index, std::vector<uint32_t>& offsets) {
    ids.clear(); offset_index.clear(); offsets.clear();
    for (uint32_t b = 0; b < d.n_blocks; b++) {
        const BlockMeta& m = meta[b];
        const uint32_t obase = (uint32_t)offsets.size();
        for (uint32_t i = 0; i < m.n_ids; i++) {
            ids.push_back(m.first_id + unpack_at(ids_payload + m.ids_woff, i, m.ids_bits));
            offset_index.push_back(obase + unpack_at(payload + m.oi_woff, i, m.oi_bits));
        }
        for (uint32_t i = 0; i < m.n_off; i++) offsets.push_back(m.off_base + unpack_at(payload + m.off_woff, i, m.off_bits));
        (void)blk_last;
    }
}

}  // namespace tsgpu
