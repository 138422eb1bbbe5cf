// pt_device.h — gfx950 device primitives of the path tracer: 3-vectors in registers, the
// fixed arithmetic contract, the XORWOW generator and the packed scene records.
//
// Arithmetic contract: everything is IEEE binary32 evaluated as written (-ffp-contract=off,
// correctly rounded / and sqrt); the ONLY fused operations are the explicit __builtin_fmaf
// calls below (dot, cross, ray.at, and the Horner steps of the sin/cos/exp polynomials).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_DEV __device__ __forceinline__

namespace pt {

// util.cuh:27-29 of the reference
constexpr float kEps = 0.00001f;
constexpr float kRayEps = 0.001f;
constexpr float kPi = 3.141592f;

struct V3 { float x, y, z; };

PT_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
PT_DEV V3 v3(float a) { return V3{a, a, a}; }
PT_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_DEV V3 operator*(V3 a, float t) { return V3{a.x * t, a.y * t, a.z * t}; }
PT_DEV V3 operator*(float t, V3 a) { return V3{a.x * t, a.y * t, a.z * t}; }
PT_DEV V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
PT_DEV V3 operator/(V3 a, float t) { return V3{a.x / t, a.y / t, a.z / t}; }
PT_DEV V3 operator/(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
PT_DEV V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }

// util.cuh:116-118 — contract: x*x' rounded, then two fused accumulations (y, then z).
PT_DEV float dot(V3 a, V3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
// util.cuh:133-140 — contract: p*q - r*s = fma(p, q, -(r*s)).
PT_DEV V3 cross(V3 a, V3 b) {
    return V3{__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)), __builtin_fmaf(a.x, b.y, -(a.y * b.x))};
}
PT_DEV float fminf_(float a, float b) { return __builtin_fminf(a, b); }   // v_min_f32: NaN-ignoring
PT_DEV float fmaxf_(float a, float b) { return __builtin_fmaxf(a, b); }
// 1.0f / a, correctly rounded, in 4 instructions instead of the 11 of the IEEE division sequence: v_rcp_f32 + one Newton
// step is bit-identical to the IEEE quotient for EVERY binary32 a with 1e-12 <= |a| <= 1e30 on gfx950 (exhaustive run over
// all 2^32 inputs on the hardware, tools/exhaustive/rcp_check.hip); anything else (zero, denormal-range, huge, inf, NaN)
// takes the IEEE sequence. PT_FAST_RCP=0 builds the plain division everywhere (A/B, and the reference point of that proof).
#ifndef PT_FAST_RCP
#define PT_FAST_RCP 1
#endif
#ifndef PT_RCP_UNIFORM
#define PT_RCP_UNIFORM 1
#endif
PT_DEV float rcp_exact(float a) {
#if PT_FAST_RCP && PT_RCP_UNIFORM
    // The range test is wave-uniform, not per lane: every lane computes the fast form, and only when some lane of the wave is
    // outside its range (practically never) do the lanes run the full division and those lanes take it. A per-lane `if` costs an
    // exec-mask save / restore and two skip branches at every call site, and the LDS-resident kernels are bound by instruction
    // issue (pt_mk_hbm.hip keeps the per-lane form below: it measures 0.7-1.0 % faster there).
    const float m = __builtin_fabsf(a);
    const float r0 = __builtin_amdgcn_rcpf(a);
    float r = __builtin_fmaf(r0, __builtin_fmaf(-a, r0, 1.0f), r0);
    const bool inRange = m >= 1e-12f && m <= 1.0e30f;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(!inRange) != 0ull, 0)) r = inRange ? r : 1.0f / a;
    return r;
#else
#if PT_FAST_RCP
    const float m = __builtin_fabsf(a);
    if (m >= 1e-12f && m <= 1.0e30f) {
        const float r0 = __builtin_amdgcn_rcpf(a);
        return __builtin_fmaf(r0, __builtin_fmaf(-a, r0, 1.0f), r0);
    }
#endif
    return 1.0f / a;
#endif
}
PT_DEV float rsqrt_(float x) { return rcp_exact(__builtin_sqrtf(x)); }    // rsqrtf := 1/sqrt, both correctly rounded
PT_DEV V3 normalize(V3 v) { float il = rsqrt_(dot(v, v)); return V3{v.x * il, v.y * il, v.z * il}; }   // util.cuh:128-131
PT_DEV float length(V3 v) { return __builtin_sqrtf(dot(v, v)); }
PT_DEV float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }             // util.cuh:142-146

// sin/cos for |x| < ~1e4: 3-step Cody-Waite reduction by pi/2, Cephes minimax polynomials.
PT_DEV void sincos_(float x, float& s, float& c) {
    float k = __builtin_rintf(x * 0.636619772f);
    float r = __builtin_fmaf(k, -1.5703125f, x);
    r = __builtin_fmaf(k, -4.837512969970703125e-4f, r);
    r = __builtin_fmaf(k, -7.54978995489188216e-8f, r);
    int q = (int)k;
    float r2 = r * r;
    float ps = __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __builtin_fmaf(ps, r2, -1.6666654611e-1f);
    float sn = __builtin_fmaf(r * r2, ps, r);
    float pc = __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __builtin_fmaf(pc, r2, 4.166664568298827e-2f);
    float cs = __builtin_fmaf(r2 * r2, pc, __builtin_fmaf(r2, -0.5f, 1.0f));
    bool swap = q & 1;
    float a = swap ? cs : sn, b = swap ? sn : cs;
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}

PT_DEV float exp_(float x) {
    if (x != x) return x;
    if (x > 88.5f) return __builtin_inff();
    if (x < -87.0f) return 0.0f;
    float n = __builtin_rintf(x * 1.44269504f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = __builtin_fmaf(r, 1.9875691500e-4f, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float y = __builtin_fmaf(p, r * r, r) + 1.0f;
    int e = (int)n;
    int e1 = e / 2, e2 = e - e1;
    float f1 = __builtin_bit_cast(float, (uint32_t)(e1 + 127) << 23);
    float f2 = __builtin_bit_cast(float, (uint32_t)(e2 + 127) << 23);
    return (y * f1) * f2;
}
PT_DEV float pow5_(float x) { float x2 = x * x; return (x2 * x2) * x; }

// ---- XORWOW (cuRAND's generator as the reference uses it, deviceCode.cu:53-61) ------------
struct Rng { uint32_t v0, v1, v2, v3, v4, d; };

PT_DEV uint32_t rng_next(Rng& s) {
    uint32_t t = s.v0 ^ (s.v0 >> 2);
    s.v0 = s.v1; s.v1 = s.v2; s.v2 = s.v3; s.v3 = s.v4;
    s.v4 = (s.v4 ^ (s.v4 << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    return s.v4 + s.d;
}
// curand_uniform: x * 2^-32 + 2^-33 (the product is exact, so fused == unfused)
PT_DEV float rng_uniform(Rng& s) { return (float)rng_next(s) * 2.3283064e-10f + 1.1641532e-10f; }

// ---- packed scene records (built by pt_scene_create) ------------------------
// Internal BVH node: both children's boxes + child refs in one 64-B record (half a 128-B line).
// ref >= 0: index of an internal PNode; ref < 0: leaf whose first packed triangle is ~ref.
struct __attribute__((aligned(64))) PNode {
    float lmin[3], lmax[3], rmin[3], rmax[3];
    int32_t left, right, pad0, pad1;
};
static_assert(sizeof(PNode) == 64, "PNode");
constexpr int32_t kRefNone = (int32_t)0x80000000;

// Triangle in leaf order (position i == BVHindices[i] of the reference): v0, e1 = v1-v0,
// e2 = v2-v0 pre-subtracted (same IEEE subtraction the reference does per test).
// idx bit 31 = last triangle of its leaf; flags bit 0 = material is MAT_LEAF (shadow rays).
struct __attribute__((aligned(16))) PTri {
    float v0[3], e1[3], e2[3];
    uint32_t idx;
    int32_t material;
    uint32_t flags;
};
static_assert(sizeof(PTri) == 48, "PTri");

// Hit-only attributes, indexed by the ORIGINAL triangle index.
struct __attribute__((aligned(16))) PAttr {
    float n0[3], n1[3], n2[3];
    float uv0[2], uv1[2], uv2[2];
    float emission[3];
    int32_t material;
    int32_t lightInd;      // index into PLight or -51
};
static_assert(sizeof(PAttr) == 80, "PAttr");

struct __attribute__((aligned(16))) PLight {
    float a[3], b[3], c[3], na[3], emission[3];
    float area;            // 0.5f * length(cross(b - a, c - a)): what neePDF / nextEventEstimation recompute per call (deviceCode.cu:75, :139), once
};
static_assert(sizeof(PLight) == 64, "PLight");

struct __attribute__((aligned(16))) PMat {
    int32_t type;
    uint32_t flags;        // 1 hasTexture, 2 hasTransMap, 4 isSpecular, 8 boundary
    int32_t priority;
    int32_t texStart, texW, texH;
    float roughness, ior, transmission;
    float albedo[3], eta[3], k[3], absorption[3];
    float albedoOverPi[3]; // cosine_f(albedo) = albedo / PI (reflectors.cuh:10-13), divided once
};
static_assert(sizeof(PMat) == 96, "PMat");
constexpr uint32_t kMatHasTexture = 1, kMatHasTransMap = 2, kMatSpecular = 4, kMatBoundary = 8;

// Wide node (4 children) of the collapsed tree the SIMPLE kernel for scenes in HBM may traverse (pt_trace.h: trace_resume_w4):
// the children's boxes are the reference tree's own float boxes, axis-major; ref as in PNode (>= 0 wide node, < 0 leaf,
// kRefNone = empty slot). One 128-byte line.
struct __attribute__((aligned(128))) WNode {
    float mnx[4], mny[4], mnz[4], mxx[4], mxy[4], mxz[4];
    int32_t ref[4];
    int32_t pad[4];
};
static_assert(sizeof(WNode) == 128, "WNode");

// Compact internal node for the SIMPLE kernel for scenes in HBM (pt_trace.h: trace_resume_q): both children's boxes as 8-bit
// offsets in the node's OWN frame — origin = the node's box minimum (exact floats), one power-of-two step for all three axes,
// x = fma(q, 2^k, origin) — rounded OUTWARD so that each decoded box contains the reference's float box: the step is 1/255
// of the node's largest extent at every depth. Child words: bit 31 = leaf, bits 24-30 of `left` = k + 64, bits 0-23 = the
// node index or the leaf's first packed triangle. Half a PNode: two 16-byte loads per visit instead of four.
struct __attribute__((aligned(16))) QNode {
    float o[3];
    uint8_t lmin[3], lmax[3], rmin[3], rmax[3];
    uint32_t left, right;
};
static_assert(sizeof(QNode) == 32, "QNode");

// FLAT kernels: one record per LEAF of the tree — the leaf's own box (as stored with its parent) and its triangle range.
struct __attribute__((aligned(16))) PLeaf {
    float mn[3], mx[3];
    int32_t first, count;
};
static_assert(sizeof(PLeaf) == 32, "PLeaf");

struct DeviceScene {
    const PNode* nodes;
    const PTri* tris;
    const PAttr* attrs;
    const PLight* lights;
    const PMat* mats;
    const float4* textures;
    int32_t rootRef;
    int32_t nLights;
    int32_t nTris;
    int32_t stackSpill;     // global spill entries per lane beyond the LDS stack (0 = none needed)
};

struct CamK {      // pt_camera as the kernels read it
    V3 origin, forward, right, up;
    int w, h;
    float aperture, focalDist, fovScale, jitter;
};

PT_DEV V3 ld3(const float* p) { return V3{p[0], p[1], p[2]}; }

}  // namespace pt
