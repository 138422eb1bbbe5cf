// oracle_math.h — TEST INFRASTRUCTURE (see oracle/README.md). PARITY UNPINNED.
//
// The reference calls CUDA libdevice rsqrtf / sinf / cosf / expf / powf (util.cuh:129,
// reflectors.cuh:28-33,187, objects.cuh:294-296, deviceCode.cu:364-366). Their results are
// ulp-level implementation details of a closed library, so this restatement fixes ONE fully
// specified binary32 algorithm per function, built only from IEEE +,-,*,/,sqrt,fma and
// integer operations. The HIP kernels implement the same operation sequences and must agree
// bit-for-bit (tests/test_gpu_math.py).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace oracle {

// rsqrtf(x) := 1 / sqrt(x), both correctly rounded (util.cuh:129).
static inline float ref_rsqrtf(float x) { return 1.0f / sqrtf(x); }

// sinf/cosf for |x| < ~1e4 (the path only passes [0, 2*PI]): Cody-Waite reduction by pi/2
// in three fused steps, then the Cephes single-precision minimax polynomials on
// [-pi/4, pi/4], all Horner steps fused.
static inline void ref_sincosf(float x, float* s_out, float* c_out) {
    float k = rintf(x * 0.636619772f);                 // nearest multiple of pi/2 (ties-to-even)
    float r = fmaf(k, -1.5703125f, x);
    r = fmaf(k, -4.837512969970703125e-4f, r);
    r = fmaf(k, -7.54978995489188216e-8f, r);
    int q = (int)k;
    float r2 = r * r;
    float ps = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(ps, r2, -1.6666654611e-1f);
    float sn = fmaf(r * r2, ps, r);
    float pc = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(pc, r2, 4.166664568298827e-2f);
    float cs = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));
    float s, c;
    switch (q & 3) {
        case 0: s = sn; c = cs; break;
        case 1: s = cs; c = -sn; break;
        case 2: s = -sn; c = -cs; break;
        default: s = -cs; c = sn; break;
    }
    *s_out = s; *c_out = c;
}
static inline float ref_sinf(float x) { float s, c; ref_sincosf(x, &s, &c); return s; }
static inline float ref_cosf(float x) { float s, c; ref_sincosf(x, &s, &c); return c; }

// expf: n = rint(x*log2e); r = x - n*ln2 (two fused steps); degree-5 Cephes polynomial;
// scale by 2^n through the exponent field. Results below 2^-126 flush to 0, above
// overflow to +inf (the path only evaluates exp(-absorption*distance)).
static inline float ref_expf(float x) {
    if (x != x) return x;
    if (x > 88.5f) return INFINITY;
    if (x < -87.0f) return 0.0f;
    float n = rintf(x * 1.44269504f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = fmaf(r, 1.9875691500e-4f, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float y = fmaf(p, r * r, r) + 1.0f;
    int e = (int)n;                                     // in [-126, 128]
    // two-step scaling keeps both factors normal
    int e1 = e / 2, e2 = e - e1;
    uint32_t b1 = (uint32_t)(e1 + 127) << 23, b2 = (uint32_t)(e2 + 127) << 23;
    float f1, f2; std::memcpy(&f1, &b1, 4); std::memcpy(&f2, &b2, 4);
    return (y * f1) * f2;
}

// powf(x, 5.0f) := ((x*x)*(x*x))*x  (reflectors.cuh:187)
static inline float ref_pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }

}  // namespace oracle
