// oracle_rng.h — TEST INFRASTRUCTURE (see oracle/README.md). PARITY UNPINNED.
//
// Restatement of cuRAND's XORWOW generator as the reference uses it:
//   curand_init(seed = 103033, subsequence = y*w + x, offset = 0, &state)   deviceCode.cu:53-61
//   curand_uniform(&state)                                                  37 call sites, SURVEY §8(a) a2
// cuRAND is a closed CUDA-Toolkit component absent from /root/reference and from this image;
// the algorithm below is the one published in curand_kernel.h (Marsaglia xorwow, 5x32-bit
// xorshift state + 32-bit Weyl counter d):
//   init:   s0 = lo32(seed)^0xaad26b49, s1 = hi32(seed)^0xf7dcefdd, t0 = 1099087573*s0,
//           t1 = 2591861531*s1; d = 6615241+t1+t0;
//           v = {123456789+t0, 362436069^t0, 521288629+t1, 88675123^t1, 5783321+t0}
//           then skip ahead subsequence * 2^67 steps of the xorshift part (d is unchanged:
//           2^67 * 362437 = 0 mod 2^32), then `offset` steps.
//   step:   t = v0^(v0>>2); v0..v3 = v1..v4; v4 = (v4^(v4<<4))^(t^(t<<1)); d += 362437;
//           return v4 + d
//   uniform: x * 2^-32 + 2^-33 in binary32, range (0, 1]
// The scramble constants are recorded from public knowledge of curand_kernel.h and are
// UNVERIFIED offline; the recurrence and the 2^67 jump are cross-checked against rocRAND's
// table (same recurrence, rocrand_xorwow.h:167-174; tests/test_oracle_rng.py).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace oracle {

struct XorwowState { uint32_t v[5]; uint32_t d; };

static inline uint32_t xorwow_next(XorwowState& s) {
    uint32_t t = s.v[0] ^ (s.v[0] >> 2);
    s.v[0] = s.v[1]; s.v[1] = s.v[2]; s.v[2] = s.v[3]; s.v[3] = s.v[4];
    s.v[4] = (s.v[4] ^ (s.v[4] << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    return s.v[4] + s.d;
}

// curand_uniform: (float)x * 2^-32 + 2^-33. 2.3283064e-10f is exactly 2^-32, so the product
// is exact and fused / unfused evaluation agree.
static inline float xorwow_uniform(XorwowState& s) {
    uint32_t x = xorwow_next(s);
    return (float)x * 2.3283064e-10f + 1.1641532e-10f;
}

// GF(2) linear algebra on the 160-bit xorshift state. A matrix is 160 rows of 5 words:
// row b is the image of basis vector e_b (bit b%32 of word b/32) — the layout
// curand_kernel.h's __curand_matvec and rocRAND's tables use.
struct XorwowMatrix { uint32_t row[160][5]; };

static inline void xorwow_matvec(const uint32_t v[5], const XorwowMatrix& m, uint32_t out[5]) {
    uint32_t r[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 32; j++)
            if (v[i] & (1u << j))
                for (int k = 0; k < 5; k++) r[k] ^= m.row[i * 32 + j][k];
    std::memcpy(out, r, sizeof(r));
}

static inline void xorwow_matsquare(const XorwowMatrix& m, XorwowMatrix& out) {
    XorwowMatrix tmp;
    for (int b = 0; b < 160; b++) xorwow_matvec(m.row[b], m, tmp.row[b]);
    out = tmp;
}

// One-step transition matrix A of the xorshift part.
static inline void xorwow_step_matrix(XorwowMatrix& a) {
    for (int b = 0; b < 160; b++) {
        XorwowState s; std::memset(&s, 0, sizeof(s));
        s.v[b / 32] = 1u << (b % 32);
        xorwow_next(s);
        for (int k = 0; k < 5; k++) a.row[b][k] = s.v[k];
    }
}

// jump[k] = A^(2^67 * 2^k), k = 0..31  (subsequence indices below 2^32).
struct XorwowJumpTable {
    std::vector<XorwowMatrix> jump;
    XorwowJumpTable() : jump(32) {
        XorwowMatrix m; xorwow_step_matrix(m);
        for (int i = 0; i < 67; i++) xorwow_matsquare(m, m);
        jump[0] = m;
        for (int k = 1; k < 32; k++) xorwow_matsquare(jump[k - 1], jump[k]);
    }
};

static inline const XorwowJumpTable& xorwow_jump_table() { static XorwowJumpTable t; return t; }

// curand_init(seed, subsequence, 0, &state)
static inline void xorwow_init(XorwowState& s, uint64_t seed, uint32_t subsequence) {
    uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u;
    uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0;
    uint32_t t1 = 2591861531u * s1;
    s.d = 6615241u + t1 + t0;
    s.v[0] = 123456789u + t0;
    s.v[1] = 362436069u ^ t0;
    s.v[2] = 521288629u + t1;
    s.v[3] = 88675123u ^ t1;
    s.v[4] = 5783321u + t0;
    const XorwowJumpTable& jt = xorwow_jump_table();
    for (int k = 0; k < 32; k++)
        if (subsequence & (1u << k)) xorwow_matvec(s.v, jt.jump[k], s.v);
}

}  // namespace oracle
