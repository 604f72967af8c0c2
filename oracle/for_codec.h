// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into, imported by or called from the product
// path (typesense_amd/, libtsgpu.so). Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use it, and only as the checker / reported CPU baseline.
//
// Restatement of the frame-of-reference (FOR) integer codec the reference links as a third-party
// dependency: cruppstahl/libfor @ 49611808d08d4e47116aa2a3ddcabeb418f405f7
// (pinned at /root/reference/cmake/For.cmake:3 and WORKSPACE:165-170; the source is NOT vendored
// under /root/reference, so this follows libfor's published algorithm and is anchored on the
// reference's call sites: src/array_base.cpp:3-16, src/sorted_array.cpp:5-21,23-70,219-253,
// src/array.cpp:16-60 and the header bytes they read directly, sorted_array.cpp:224-225).
//
// Format (what the reference relies on): [u32 base (little endian)][u8 bits][payload], payload =
// `length` values of (v - base), `bits` bits each, packed LSB-first into a contiguous bit stream.
// Byte-level parity with libfor: UNPINNED (no reference test fixes payload bytes; the format is
// private, never persisted). Decoded values, lengths, select and lower-bound results: pinned by
// the reference's sorted_array/array/posting_list tests, ported in tests/test_oracle_arrays.py.
#pragma once
#include <cstdint>
#include <cstring>
#include <cstddef>

namespace oracle {

static const uint32_t FOR_HEADER_BYTES = 5;  // METADATA_OVERHEAD, include/array_base.h:12

static inline uint32_t required_bits(uint32_t v) {  // include/array_base.h:23-25
    return v == 0 ? 0u : 32u - (uint32_t)__builtin_clz(v);
}

// payload bytes for `length` values of `bits` bits (libfor: groups of 32/16/8 values are byte
// aligned for every bit width, the <8 remainder is rounded up to a byte)
static inline uint32_t for_compressed_size_bits(uint32_t length, uint32_t bits) {
    uint32_t full = (length / 8) * 8;
    uint32_t rem = length - full;
    return (full / 8) * bits + (rem * bits + 7) / 8;
}

// OR the `bits` low bits of v into a zero-initialised payload at bitpos (payload has >= 8 bytes of slack)
static inline void for_put_bits(uint8_t* payload, uint64_t bitpos, uint32_t bits, uint32_t v) {
    if (bits == 0) return;
    uint64_t byte = bitpos >> 3;
    uint32_t shift = (uint32_t)(bitpos & 7);
    uint64_t cur;
    memcpy(&cur, payload + byte, 8);
    cur |= ((uint64_t)v) << shift;      // bits <= 32, shift <= 7 -> fits in 64 bits
    memcpy(payload + byte, &cur, 8);
}

static inline uint32_t for_get_bits(const uint8_t* payload, uint64_t bitpos, uint32_t bits) {
    if (bits == 0) return 0;
    uint64_t byte = bitpos >> 3;
    uint32_t shift = (uint32_t)(bitpos & 7);
    uint64_t acc;
    memcpy(&acc, payload + byte, 8);    // buffers carry >= 8 bytes of slack (array_base::encode)
    acc >>= shift;
    return bits == 32 ? (uint32_t)acc : (uint32_t)(acc & ((1ull << bits) - 1));
}

static inline uint32_t for_header_base(const uint8_t* in) { uint32_t b; memcpy(&b, in, 4); return b; }
static inline uint32_t for_header_bits(const uint8_t* in) { return in[4]; }

// returns total bytes written (header + payload)
static inline uint32_t for_compress_bits(const uint32_t* in, uint8_t* out, uint32_t length,
                                         uint32_t base, uint32_t bits) {
    memcpy(out, &base, 4);
    out[4] = (uint8_t)bits;
    uint8_t* payload = out + FOR_HEADER_BYTES;
    uint32_t nbytes = for_compressed_size_bits(length, bits);
    memset(payload, 0, nbytes + 8);     // callers allocate 8 bytes of slack past the payload
    for (uint32_t i = 0; i < length; i++) for_put_bits(payload, (uint64_t)i * bits, bits, in[i] - base);
    return FOR_HEADER_BYTES + nbytes;
}

static inline uint32_t for_compress_sorted(const uint32_t* in, uint8_t* out, uint32_t length) {
    if (length == 0) { return for_compress_bits(in, out, 0, 0, 0); }
    uint32_t base = in[0], max = in[length - 1];
    return for_compress_bits(in, out, length, base, required_bits(max - base));
}

static inline uint32_t for_compress_unsorted(const uint32_t* in, uint8_t* out, uint32_t length) {
    if (length == 0) { return for_compress_bits(in, out, 0, 0, 0); }
    uint32_t m = in[0], M = in[0];
    for (uint32_t i = 1; i < length; i++) { if (in[i] < m) m = in[i]; if (in[i] > M) M = in[i]; }
    return for_compress_bits(in, out, length, m, required_bits(M - m));
}

static inline uint32_t for_uncompress(const uint8_t* in, uint32_t* out, uint32_t length) {
    uint32_t base = for_header_base(in), bits = for_header_bits(in);
    const uint8_t* payload = in + FOR_HEADER_BYTES;
    if (bits == 0) { for (uint32_t i = 0; i < length; i++) out[i] = base; }
    else {
        // streaming decode through a 64-bit window (libfor uses unrolled per-width kernels; same values)
        const uint64_t mask = bits == 32 ? 0xFFFFFFFFull : ((1ull << bits) - 1);
        uint64_t bitpos = 0;
        for (uint32_t i = 0; i < length; i++, bitpos += bits) {
            uint64_t acc;
            memcpy(&acc, payload + (bitpos >> 3), 8);
            out[i] = base + (uint32_t)((acc >> (bitpos & 7)) & mask);
        }
    }
    return FOR_HEADER_BYTES + for_compressed_size_bits(length, bits);
}

static inline uint32_t for_select_bits(const uint8_t* payload, uint32_t base, uint32_t bits, uint32_t index) {
    return base + for_get_bits(payload, (uint64_t)index * bits, bits);
}

static inline uint32_t for_select(const uint8_t* in, uint32_t index) {
    return for_select_bits(in + FOR_HEADER_BYTES, for_header_base(in), for_header_bits(in), index);
}

static inline uint32_t for_linear_search(const uint8_t* in, uint32_t length, uint32_t value) {
    for (uint32_t i = 0; i < length; i++) if (for_select(in, i) == value) return i;
    return length;
}

// first element >= value (libfor semantics, mirrored by the reference's own copy at
// src/sorted_array.cpp:116-143): when every element is smaller, returns length-1 with *actual < value
static inline uint32_t for_lower_bound_search(const uint8_t* in, uint32_t length, uint32_t value, uint32_t* actual) {
    uint32_t base = for_header_base(in), bits = for_header_bits(in);
    const uint8_t* payload = in + FOR_HEADER_BYTES;
    uint32_t imin = 0, imax = length - 1, imid, v;
    while (imin + 1 < imax) {
        imid = imin + ((imax - imin) / 2);
        v = for_select_bits(payload, base, bits, imid);
        if (v >= value) imax = imid; else imin = imid;
    }
    v = for_select_bits(payload, base, bits, imin);
    if (v >= value) { *actual = v; return imin; }
    v = for_select_bits(payload, base, bits, imax);
    *actual = v;
    return imax;
}

}  // namespace oracle
