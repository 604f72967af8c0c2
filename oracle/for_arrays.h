// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/for_codec.h header).
// Restates the reference's FOR-compressed arrays:
//   array_base  : /root/reference/include/array_base.h:14-51, src/array_base.cpp:3-32
//   sorted_array: /root/reference/include/sorted_array.h, src/sorted_array.cpp:5-115
//   array       : /root/reference/include/array.h, src/array.cpp:3-108
// Same behaviour (values, lengths, min/max bookkeeping, indexOf/lower-bound conventions) and the
// same cost shape on the query path: uncompress() heap-allocates a fresh uint32_t[] per call.
#pragma once
#include <vector>
#include <limits>
#include <algorithm>
#include "for_codec.h"

namespace oracle {

class array_base {
protected:
    std::vector<uint8_t> in;      // [u32 base][u8 bits][payload]
    uint32_t length = 0;
    uint32_t min = std::numeric_limits<uint32_t>::max();
    uint32_t max = std::numeric_limits<uint32_t>::min();

    void encode(const uint32_t* vals, uint32_t n, uint32_t base, uint32_t bits) {
        in.assign(FOR_HEADER_BYTES + for_compressed_size_bits(n, bits) + 16, 0);
        for_compress_bits(vals, in.data(), n, base, bits);
        length = n;
    }

public:
    array_base() { in.assign(FOR_HEADER_BYTES + 16, 0); }

    // array_base::uncompress, src/array_base.cpp:3-16 (caller owns the buffer, delete[])
    uint32_t* uncompress(uint32_t len = 0) const {
        const uint32_t actual_len = std::max(len, length);
        uint32_t* out = new uint32_t[actual_len ? actual_len : 1];
        if (length == 0) return out;
        for_uncompress(in.data(), out, length);
        return out;
    }

    uint32_t getLength() const { return length; }
    uint32_t getMin() const { return min; }
    uint32_t getMax() const { return max; }
    uint32_t getSizeInBytes() const { return (uint32_t)in.size(); }
    const uint8_t* raw() const { return in.data(); }
};

class sorted_array : public array_base {
public:
    // src/sorted_array.cpp:5-21
    void load(const uint32_t* sorted, uint32_t n) {
        min = n != 0 ? sorted[0] : 0;
        max = n > 1 ? sorted[n - 1] : min;
        encode(sorted, n, n ? min : 0, n ? required_bits(max - min) : 0);
    }

    uint32_t at(uint32_t index) const { return for_select(in.data(), index); }   // :87-89
    uint32_t last() const { return length == 0 ? UINT32_MAX : max; }  // :401-403

    bool contains(uint32_t value) const {  // :91-99
        if (length == 0) return false;
        uint32_t actual;
        for_lower_bound_search(in.data(), length, value, &actual);
        return actual == value;
    }

    uint32_t indexOf(uint32_t value) const {  // :101-114
        if (length == 0) return length;
        uint32_t actual;
        uint32_t index = for_lower_bound_search(in.data(), length, value, &actual);
        return actual == value ? index : length;
    }

    // src/sorted_array.cpp:23-70 — returns the index the value landed on
    size_t append(uint32_t value) {
        if (length != 0 && value < max) {
            uint32_t* arr = uncompress(length + 1);
            uint32_t found;
            uint32_t gte = for_lower_bound_search(in.data(), length, value, &found);
            for (size_t j = length; j > gte; j--) arr[j] = arr[j - 1];
            arr[gte] = value;
            load(arr, length + 1);
            delete[] arr;
            return gte;
        }
        uint32_t* arr = uncompress(length + 1);
        arr[length] = value;
        uint32_t n = length + 1;
        uint32_t m = std::min(min, value), M = std::max(max, value);
        if (length == 0) { m = value; M = value; }
        min = m; max = M;
        encode(arr, n, arr[0], required_bits(M - arr[0]));
        delete[] arr;
        return length - 1;
    }

    bool insert(size_t index, uint32_t value) {  // :72-85
        if (index >= length) return false;
        uint32_t* arr = uncompress(length + 1);
        memmove(&arr[index + 1], &arr[index], sizeof(uint32_t) * (length - index));
        arr[index] = value;
        load(arr, length + 1);
        delete[] arr;
        return true;
    }

    void remove_value(uint32_t value) {  // :255-283
        if (length == 0) return;
        uint32_t actual;
        uint32_t idx = for_lower_bound_search(in.data(), length, value, &actual);
        if (actual != value) return;
        uint32_t* arr = uncompress();
        for (uint32_t i = idx; i + 1 < length; i++) arr[i] = arr[i + 1];
        load(arr, length - 1);
        delete[] arr;
    }

    // src/sorted_array.cpp:304-356 — count of `values` present (values sorted)
    size_t numFoundOf(const uint32_t* values, size_t n) const {
        size_t found = 0;
        for (size_t i = 0; i < n; i++) if (contains(values[i])) found++;
        return found;
    }
};

class array : public array_base {
public:
    void load(const uint32_t* vals, uint32_t n, uint32_t m, uint32_t M) {  // src/array.cpp:44-60
        min = m; max = M;
        uint32_t lo = 0, hi = 0;
        if (n) { lo = hi = vals[0]; for (uint32_t i = 1; i < n; i++) { lo = std::min(lo, vals[i]); hi = std::max(hi, vals[i]); } }
        encode(vals, n, lo, required_bits(hi - lo));
    }

    uint32_t at(uint32_t index) const { return for_select(in.data(), index); }
    bool contains(uint32_t v) const { return for_linear_search(in.data(), length, v) != length; }
    uint32_t indexOf(uint32_t v) const { return for_linear_search(in.data(), length, v); }

    bool append(uint32_t value) {  // src/array.cpp:16-42
        uint32_t* arr = uncompress(length + 1);
        arr[length] = value;
        if (value < min) min = value;
        if (value > max) max = value;
        uint32_t n = length + 1;
        uint32_t lo = arr[0], hi = arr[0];
        for (uint32_t i = 1; i < n; i++) { lo = std::min(lo, arr[i]); hi = std::max(hi, arr[i]); }
        encode(arr, n, lo, required_bits(hi - lo));
        delete[] arr;
        return true;
    }

    bool insert(size_t index, const uint32_t* values, size_t num_values) {  // :62-84
        if (index >= length) return false;
        uint32_t* arr = uncompress(length + (uint32_t)num_values);
        memmove(&arr[index + num_values], &arr[index], sizeof(uint32_t) * (length - index));
        uint32_t m = min, M = max;
        for (size_t i = 0; i < num_values; i++) {
            if (values[i] < m) m = values[i];
            if (values[i] > M) M = values[i];
            arr[index + i] = values[i];
        }
        load(arr, length + (uint32_t)num_values, m, M);
        delete[] arr;
        return true;
    }

    void remove_index(uint32_t start_index, uint32_t end_index) {  // :86-118
        uint32_t* cur = uncompress();
        std::vector<uint32_t> out;
        uint32_t m = std::numeric_limits<uint32_t>::max(), M = 0;
        for (uint32_t i = 0; i < length; i++) {
            if (i < start_index || i >= end_index) {
                out.push_back(cur[i]);
                m = std::min(m, cur[i]); M = std::max(M, cur[i]);
            }
        }
        delete[] cur;
        load(out.data(), (uint32_t)out.size(), m, M);
    }
};

}  // namespace oracle
