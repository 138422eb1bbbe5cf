// xorwow_host.h — host-side construction of the XORWOW subsequence jump tables the
// rng_init kernel consumes: jump[k] = A^(2^67 * 2^k), k = 0..31, where A is the one-step
// transition of the 160-bit xorshift part of cuRAND's XORWOW (curand_init's subsequence skip,
// used by the reference at deviceCode.cu:60 with subsequence = pixel index).
//
// Row-image layout [k][bit b][5 words]: row b is A^n applied to the state with only bit b set
// (bit b%32 of word b/32), so M*v is the XOR of the rows selected by v's set bits.
#pragma once
#include <array>
#include <cstdint>
#include <vector>

namespace xorwow_host {

using Row = std::array<uint32_t, 5>;
using Mat = std::vector<Row>;      // 160 rows

inline Row apply(const Mat& m, const Row& v) {
    Row r{0, 0, 0, 0, 0};
    for (int b = 0; b < 160; b++)
        if ((v[b >> 5] >> (b & 31)) & 1u)
            for (int q = 0; q < 5; q++) r[q] ^= m[b][q];
    return r;
}

inline Mat square(const Mat& m) {
    Mat out(160);
    for (int b = 0; b < 160; b++) out[b] = apply(m, m[b]);
    return out;
}

inline Mat step_matrix() {
    Mat a(160);
    for (int b = 0; b < 160; b++) {
        uint32_t v[5] = {0, 0, 0, 0, 0};
        v[b >> 5] = 1u << (b & 31);
        uint32_t t = v[0] ^ (v[0] >> 2);
        Row n{v[1], v[2], v[3], v[4], (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1))};
        a[b] = n;
    }
    return a;
}

// 32 x 160 x 5 words, built once per process.
inline const std::vector<uint32_t>& jump_table() {
    static const std::vector<uint32_t> table = [] {
        Mat m = step_matrix();
        for (int i = 0; i < 67; i++) m = square(m);
        std::vector<uint32_t> t;
        t.reserve(32 * 800);
        for (int k = 0; k < 32; k++) {
            for (int b = 0; b < 160; b++) for (int q = 0; q < 5; q++) t.push_back(m[b][q]);
            if (k < 31) m = square(m);
        }
        return t;
    }();
    return table;
}

}  // namespace xorwow_host
