// oracle_types.h — TEST INFRASTRUCTURE (see oracle/README.md). PARITY UNPINNED.
//
// float4 algebra, constants and boundary structs of the reference, restated for a plain
// CPU build. Citations are to /root/reference.
//
// Arithmetic contract (DESIGN.md "Arithmetic contract"): every expression is evaluated
// in IEEE binary32 exactly as written (build with -ffp-contract=off); the only fused
// operations are the explicit fmaf calls in dot(), cross3() and Ray::at(). nvcc's default
// -fmad=true fuses some a*b+c of the reference in a compiler-chosen pattern that cannot be
// pinned offline; this restatement fixes the pattern below.
#pragma once
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

namespace oracle {

// util.cuh:27-31
static const float EPSILON = 0.00001f;
static const float RAY_EPSILON = 0.001f;
static const float PI = 3.141592f;

struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };

// util.cuh:35-47
static inline float4 f4(float x, float y, float z, float w = 0.0f) { return float4{x, y, z, w}; }
static inline float4 f4() { return float4{0, 0, 0, 0}; }
static inline float4 f4(float a) { return float4{a, a, a, 0}; }
static inline float2 f2(float x, float y) { return float2{x, y}; }
static inline float2 f2(float a) { return float2{a, a}; }

// util.cuh:49-114 — every binary operator forces w = 0; compound ops touch xyz only.
static inline float4 operator+(const float4& a, const float4& b) { return f4(a.x + b.x, a.y + b.y, a.z + b.z, 0.0f); }
static inline float2 operator+(const float2& a, const float2& b) { return f2(a.x + b.x, a.y + b.y); }
static inline float4 operator-(const float4& a, const float4& b) { return f4(a.x - b.x, a.y - b.y, a.z - b.z, 0.0f); }
static inline float4 operator*(const float4& a, float t) { return f4(a.x * t, a.y * t, a.z * t, 0.0f); }
static inline float4 operator*(float t, const float4& a) { return a * t; }
static inline float2 operator*(const float2& a, float t) { return f2(a.x * t, a.y * t); }
static inline float2 operator*(float t, const float2& a) { return a * t; }
static inline float4 operator/(const float4& a, float t) { return f4(a.x / t, a.y / t, a.z / t, 0.0f); }
static inline float4& operator+=(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
static inline float4& operator*=(float4& a, float t) { a.x *= t; a.y *= t; a.z *= t; return a; }
static inline float4& operator*=(float4& a, const float4& b) { a.x *= b.x; a.y *= b.y; a.z *= b.z; return a; }
static inline float4& operator/=(float4& a, float t) { a.x /= t; a.y /= t; a.z /= t; return a; }
static inline float4 operator*(const float4& a, const float4& b) { return f4(a.x * b.x, a.y * b.y, a.z * b.z, 0.0f); }
static inline float4 operator/(const float4& a, const float4& b) { return f4(a.x / b.x, a.y / b.y, a.z / b.z, 0.0f); }
static inline float4 operator-(const float4& v) { return f4(-v.x, -v.y, -v.z, -v.w); }

// util.cuh:116-118. Contract: x*x' first, then two fused accumulations.
static inline float dot(const float4& a, const float4& b) {
    return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x));
}
static inline float lengthSquared(const float4& v) { return dot(v, v); }   // util.cuh:124
static inline float length(const float4& v) { return sqrtf(dot(v, v)); }   // util.cuh:120

// util.cuh:133-140. Contract: p*q - r*s = fmaf(p, q, -(r*s)).
static inline float4 cross3(const float4& a, const float4& b) {
    return f4(fmaf(a.y, b.z, -(a.z * b.y)),
              fmaf(a.z, b.x, -(a.x * b.z)),
              fmaf(a.x, b.y, -(a.y * b.x)), 0.0f);
}

// util.cuh:142-155
static inline float clampf(float x, float lo, float hi) { if (x < lo) return lo; if (x > hi) return hi; return x; }

// Model of the GPU's v_min_f32 / v_max_f32 (IEEE mode): NaN-ignoring, and -0 < +0.
// CUDA fminf/fmaxf are NaN-ignoring too; the sign-of-zero choice never reaches a result.
static inline float ref_fminf(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a == 0.0f && b == 0.0f) return std::signbit(a) ? a : b;
    return a < b ? a : b;
}
static inline float ref_fmaxf(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    if (a == 0.0f && b == 0.0f) return std::signbit(a) ? b : a;
    return a > b ? a : b;
}
// util.cuh:188-203
static inline float4 fminf4(const float4& a, const float4& b) { return f4(ref_fminf(a.x, b.x), ref_fminf(a.y, b.y), ref_fminf(a.z, b.z)); }
static inline float4 fmaxf4(const float4& a, const float4& b) { return f4(ref_fmaxf(a.x, b.x), ref_fmaxf(a.y, b.y), ref_fmaxf(a.z, b.z)); }
static inline float getFloat4Component(const float4& v, int i) {   // util.cuh:205-213
    switch (i) { case 0: return v.x; case 1: return v.y; case 2: return v.z; case 3: return v.w; default: return 0.0f; }
}
static inline float surfaceArea(const float4& mn, const float4& mx) {   // util.cuh:225-231
    float dx = mx.x - mn.x, dy = mx.y - mn.y, dz = mx.z - mn.z;
    return 2.0f * (dx * dy + dx * dz + dy * dz);
}
static inline float4 sqrtf4(const float4& v) { return f4(sqrtf(v.x), sqrtf(v.y), sqrtf(v.z), 0.0f); }  // util.cuh:233

// ---------------------------------------------------------------------------
// Boundary structs — byte layouts of SURVEY.md Appendix A (CUDA ABI).
// ---------------------------------------------------------------------------
struct BVHnode {            // objects.cuh:12-20, 48 B
    float4 aabbMIN, aabbMAX;
    int left, right, first, primCount;
};
static_assert(sizeof(BVHnode) == 48, "BVHnode layout");

struct Triangle {           // objects.cuh:159-172, 80 B
    int aInd, bInd, cInd;
    int naInd, nbInd, ncInd;
    int uvaInd, uvbInd, uvcInd;
    int materialID;
    float4 emission;        // @48
    int lightInd;           // @64  (-51 = not a light, main.cu:1056)
    int triInd;             // @68
};
static_assert(sizeof(Triangle) == 80, "Triangle layout");

struct Vertices {           // objects.cuh:151-157 (colors is never filled: main.cu:348)
    const float4* positions; const float4* normals; const float4* colors; const float2* uvs;
};

struct Ray {                // objects.cuh:186-197
    float4 origin, direction;
    // objects.cuh:195 `origin + t*direction`. Contract: fmaf(t, d, o) per component.
    float4 at(float t) const { return f4(fmaf(t, direction.x, origin.x), fmaf(t, direction.y, origin.y), fmaf(t, direction.z, origin.z), 0.0f); }
};

struct Camera {             // objects.cuh:199-219, 112 B
    float4 cameraOrigin;
    int w, h;
    float xRot, yRot, zRot;
    float aperture, focalDist, fovScale;
    float antiAliasJitterDist;
    float4 forward, right, up;
};
static_assert(sizeof(Camera) == 112, "Camera layout");

enum IntegratorChoice { UNIDIRECTIONAL = 0, BIDIRECTIONAL = 1, NAIVE_UNIDIRECTIONAL = 2, VCM = 3, SPPM = 4 };  // objects.cuh:570-576
enum TransportMode { TRANSPORTMODE_IMPORTANCE = 0, TRANSPORTMODE_RADIANCE = 1 };                               // objects.cuh:578-581
enum MaterialType { MAT_DIFFUSE = 0, MAT_METAL = 1, MAT_SMOOTHDIELECTRIC = 2, MAT_MICROFACETDIELECTRIC = 3,
                    MAT_LEAF = 4, MAT_FLOWER = 5, MAT_DELTAMIRROR = 6 };                                       // objects.cuh:595-603

struct Material {           // objects.cuh:605-638, 176 B
    bool hasTexture; int startInd, width, height;
    bool hasTransMap; int tstartInd, twidth, theight;
    int type;
    float4 albedo;          // @48
    float roughness;        // @64
    float4 eta;             // @80
    float4 k;               // @96
    float ior;              // @112
    float metallic, specular, transmission;
    bool isSpecular, boundary, thinWalled;   // @128..130
    float4 absorption;      // @144
    int priority;           // @160
};
static_assert(sizeof(Material) == 176, "Material layout");

struct Intersection {       // objects.cuh:550-568 (`color` dropped: never read)
    float4 point, normal, emission;
    float2 uv;
    int triIDX, materialID;
    bool valid, backface;
    float dist;
    float baryU, baryV;     // kept for tests only
    Intersection() { valid = false; uv = f2(-1.0f); point = normal = emission = f4(); triIDX = 0; materialID = 0; backface = false; dist = 0; baryU = baryV = 0; }
};

struct MeshConfig { std::string path; float emissionMultiplier = 0.0f; float4 emissionColor = f4(); int materialID = 0; };  // objects.cuh:794-799

struct RenderConfig {       // objects.cuh:801-842 (only the keys the hot path reads are kept)
    int width = 0, height = 0;
    std::string name, integratorType;
    int sampleCount = 0, maxDepth = 0, bvhLeafSize = 0;
    bool postProcess = false, pinholeCamera = false;
    float4 camPos = f4(), camRot = f4();
    float camFov = 0.0f, camApeture = 0.0f, camFocalDist = 0.0f;
    std::vector<MeshConfig> meshes;
};

// Host-side scene exactly as initRender assembles it before the launcher call (main.cu:345-557).
struct Scene {
    std::vector<float4> points, normals;
    std::vector<float2> uvs;
    std::vector<Triangle> mesh, lights;
    std::vector<BVHnode> bvh;
    std::vector<int> indices;
    std::vector<Material> mats;
    std::vector<float4> textures;
    int largestLeaf = 0, backupCount = 0, maxDepthOfTree = 0;
};

// Per-pixel work counters; the HIP kernel must reproduce them exactly.
struct PixelCounters {
    uint32_t raysClosest, raysShadow, nodePops, boxTests, triTests, hits, rngDraws, iterations;
};

}  // namespace oracle
