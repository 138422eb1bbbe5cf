// oracle_render.cpp — TEST INFRASTRUCTURE (see oracle/README.md). PARITY UNPINNED.
//
// Scalar CPU restatement of the reference's unidirectional render loop, one function per
// reference function, in the reference's own data model (AoS triangles, `Vertices`
// indirection, eager attribute interpolation, explicit node stack). It deliberately shares no
// code with the product kernels: agreement between the two is the parity proof.
//
// Deviations, all from SURVEY.md Appendix D (undefined behaviour in the reference that a
// restatement has to define) are marked "App. D".
#include <algorithm>
#include <thread>

#include "oracle.h"

namespace oracle {

static inline float4 normalize(const float4& v) {          // util.cuh:128-131
    float invLen = ref_rsqrtf(dot(v, v));
    return f4(v.x * invLen, v.y * invLen, v.z * invLen, 0.0f);
}

static inline float uniform(XorwowState& st, PixelCounters* pc) {
    if (pc) pc->rngDraws++;
    return xorwow_uniform(st);
}

// util.cuh:163-185
static inline void onbTangent(const float4& n, float4& t) {
    if (fabsf(n.x) > fabsf(n.z)) t = normalize(f4(-n.y, n.x, 0.0f));
    else t = normalize(f4(0.0f, -n.z, n.y));
}
static inline void toWorld(const float4& wo_local, const float4& n, float4& wo_world) {
    float4 t, b;
    onbTangent(n, t);
    b = cross3(n, t);
    wo_world = wo_local.x * t + wo_local.y * b + wo_local.z * n;
}
static inline void toLocal(const float4& wo_world, const float4& n, float4& wo_local) {
    float4 t, b;
    onbTangent(n, t);
    b = cross3(n, t);
    wo_local = f4(dot(wo_world, t), dot(wo_world, b), dot(wo_world, n), 0.0f);
}

// ---------------------------------------------------------------------------------------
// integratorUtilities.cuh:8-42. `f = 1.0/a` is a double divide rounded to float; by the
// 2p+2 double-rounding theorem (53 >= 2*24+2) that equals the correctly rounded binary32
// quotient 1.0f/a, which is what both sides compute.
bool triangleIntersect(const Vertices& verts, const Triangle& tri, const Ray& r, float4& barycentric, float& tval) {
    float4 tria = verts.positions[tri.aInd];
    float4 trib = verts.positions[tri.bInd];
    float4 tric = verts.positions[tri.cInd];
    float4 e1 = trib - tria;
    float4 e2 = tric - tria;
    float4 h = cross3(r.direction, e2);
    float a = dot(h, e1);
    if (fabsf(a) < 1e-12f) return false;
    float f = 1.0f / a;
    float4 s = r.origin - tria;
    float u = f * dot(s, h);
    float4 q = cross3(s, e1);
    float v = f * dot(r.direction, q);
    float t = f * dot(e2, q);
    if (((u >= 0) && (v >= 0) && (u + v <= 1)) && t > 0.0f) {
        barycentric = f4(u, v, 1.0f - u - v);
        tval = t;
        return true;
    }
    barycentric = f4();
    return false;
}

// integratorUtilities.cuh:44-82
bool aabbIntersect(const Ray& r, float4 minCorner, float4 maxCorner, float& tmin, float& tmax) {
    tmin = -1e30f;
    tmax = 1e30f;
    float4 invDir = f4(1.0f / r.direction.x, 1.0f / r.direction.y, 1.0f / r.direction.z, 0.0f);
    float tx1 = (minCorner.x - r.origin.x) * invDir.x;
    float tx2 = (maxCorner.x - r.origin.x) * invDir.x;
    tmin = ref_fmaxf(tmin, ref_fminf(tx1, tx2));
    tmax = ref_fminf(tmax, ref_fmaxf(tx1, tx2));
    float ty1 = (minCorner.y - r.origin.y) * invDir.y;
    float ty2 = (maxCorner.y - r.origin.y) * invDir.y;
    tmin = ref_fmaxf(tmin, ref_fminf(ty1, ty2));
    tmax = ref_fminf(tmax, ref_fmaxf(ty1, ty2));
    float tz1 = (minCorner.z - r.origin.z) * invDir.z;
    float tz2 = (maxCorner.z - r.origin.z) * invDir.z;
    tmin = ref_fmaxf(tmin, ref_fminf(tz1, tz2));
    tmax = ref_fminf(tmax, ref_fmaxf(tz1, tz2));
    return (tmax >= tmin) && (tmax > 0.0f);
}

static inline Vertices vertsOf(const Scene& sc) { return Vertices{sc.points.data(), sc.normals.data(), nullptr, sc.uvs.data()}; }

// Shared internal-node step of both traversals (integratorUtilities.cuh:148-183 / 250-285):
// test both children, push the farther first so the nearer is popped next; ties go right.
static inline void pushChildren(const Ray& r, const std::vector<BVHnode>& BVH, const BVHnode& node, int* nodeStack, int& stackTop, PixelCounters* pc) {
    if (node.left >= 0 || node.right >= 0) {
        float tminL = 0, tmaxL = 0, tminR = 0, tmaxR = 0;
        bool hitLeft = false, hitRight = false;
        if (node.left >= 0) { hitLeft = aabbIntersect(r, BVH[node.left].aabbMIN, BVH[node.left].aabbMAX, tminL, tmaxL); if (pc) pc->boxTests++; }
        if (node.right >= 0) { hitRight = aabbIntersect(r, BVH[node.right].aabbMIN, BVH[node.right].aabbMAX, tminR, tmaxR); if (pc) pc->boxTests++; }
        if (hitLeft && hitRight) {
            if (tminL < tminR) { nodeStack[stackTop++] = node.right; nodeStack[stackTop++] = node.left; }
            else { nodeStack[stackTop++] = node.left; nodeStack[stackTop++] = node.right; }
        } else if (hitLeft) nodeStack[stackTop++] = node.left;
        else if (hitRight) nodeStack[stackTop++] = node.right;
    }
}

// integratorUtilities.cuh:84-186. nodeStack[128] without overflow check in the reference;
// App. D: the loader rejects trees deeper than 128 (Scene::maxDepthOfTree).
void BVHSceneIntersect(const Ray& r, const Scene& sc, Intersection& intersect, float max_t, int skipTri, PixelCounters* pc) {
    const std::vector<BVHnode>& BVH = sc.bvh;
    Vertices verts = vertsOf(sc);
    intersect.valid = false;
    float min_t = 3.402823466e+38f;
    int nodeStack[160];
    int stackTop = 0;
    nodeStack[stackTop++] = 0;
    if (pc) pc->raysClosest++;
    while (stackTop > 0) {
        int currentIndex = nodeStack[--stackTop];
        const BVHnode& node = BVH[currentIndex];
        if (pc) pc->nodePops++;
        if (node.primCount > 0) {
            for (int i = node.first; i < node.primCount + node.first; i++) {
                int idx = sc.indices[i];
                if (idx == skipTri) continue;
                const Triangle* tri = &sc.mesh[idx];
                float4 barycentric;
                float t;
                if (pc) pc->triTests++;
                bool hitTri = triangleIntersect(verts, *tri, r, barycentric, t);
                if (hitTri && (t < min_t) && (t < max_t)) {
                    min_t = t;
                    intersect.point = r.at(t);
                    intersect.normal = normalize(verts.normals[tri->naInd] * barycentric.z + verts.normals[tri->nbInd] * barycentric.x + verts.normals[tri->ncInd] * barycentric.y);
                    intersect.uv = verts.uvs[tri->uvaInd] * barycentric.z + verts.uvs[tri->uvbInd] * barycentric.x + verts.uvs[tri->uvcInd] * barycentric.y;
                    if (dot(intersect.normal, r.direction) > 0.0f) { intersect.normal = -intersect.normal; intersect.backface = true; }
                    else intersect.backface = false;
                    intersect.materialID = tri->materialID;
                    intersect.emission = tri->emission;
                    intersect.valid = true;
                    intersect.triIDX = idx;
                    intersect.dist = t;
                    intersect.baryU = barycentric.x; intersect.baryV = barycentric.y;
                }
            }
        } else {
            pushChildren(r, BVH, node, nodeStack, stackTop, pc);
        }
    }
    if (pc && intersect.valid) pc->hits++;
}

static inline float schlick_fresnel(float cosTheta, float etaI, float etaT) {   // reflectors.cuh:183-188
    float R0 = (etaI - etaT) / (etaI + etaT);
    R0 = R0 * R0;
    return R0 + (1.0f - R0) * ref_pow5(1.0f - fabsf(cosTheta));
}

// integratorUtilities.cuh:188-288. The triangle is tested before the skip_tri check (:210-213).
void BVHShadowRay(const Ray& r, const Scene& sc, float4& throughputScale, float max_t, int skip_tri, PixelCounters* pc) {
    const std::vector<BVHnode>& BVH = sc.bvh;
    Vertices verts = vertsOf(sc);
    int nodeStack[160];
    int stackTop = 0;
    nodeStack[stackTop++] = 0;
    throughputScale = f4(1.0f);
    if (pc) pc->raysShadow++;
    while (stackTop > 0) {
        int currentIndex = nodeStack[--stackTop];
        const BVHnode& node = BVH[currentIndex];
        if (pc) pc->nodePops++;
        if (node.primCount > 0) {
            for (int i = node.first; i < node.primCount + node.first; i++) {
                int idx = sc.indices[i];
                const Triangle* tri = &sc.mesh[idx];
                float4 barycentric;
                float t;
                if (pc) pc->triTests++;
                bool hitTri = triangleIntersect(verts, *tri, r, barycentric, t);
                if (idx == skip_tri) continue;
                if (hitTri && (t < max_t)) {
                    int matID = tri->materialID;
                    if (sc.mats[matID].type == MAT_LEAF) {
                        float4 transColor = sc.mats[matID].albedo;
                        float transmission = sc.mats[matID].transmission;
                        float4 n = verts.normals[tri->naInd] * barycentric.z + verts.normals[tri->nbInd] * barycentric.x + verts.normals[tri->ncInd] * barycentric.y;
                        float cosTheta = fabsf(dot(r.direction, normalize(n)));
                        float F = schlick_fresnel(cosTheta, 1.0f, sc.mats[matID].ior);
                        throughputScale *= transColor * transmission * (1.0f - F);
                        if (ref_fmaxf(throughputScale.x, ref_fmaxf(throughputScale.y, throughputScale.z)) < 0.01f) { throughputScale = f4(0.0f); return; }
                    } else {
                        throughputScale = f4(0.0f);
                        return;
                    }
                }
            }
        } else {
            pushChildren(r, BVH, node, nodeStack, stackTop, pc);
        }
    }
}

// integratorUtilities.cuh:290-335 — unused by the reference kernels; kept as the brute-force
// cross-check of BVHSceneIntersect.
void sceneIntersection(const Ray& r, const Scene& sc, Intersection& intersect) {
    Vertices verts = vertsOf(sc);
    intersect.valid = false;
    float min_t = 3.402823466e+38f;
    for (int i = 0; i < (int)sc.mesh.size(); i++) {
        const Triangle* tri = &sc.mesh[i];
        float4 barycentric;
        float t;
        bool hitTri = triangleIntersect(verts, *tri, r, barycentric, t);
        if (hitTri && (t < min_t)) {
            min_t = t;
            intersect.point = r.at(t);
            intersect.normal = normalize(verts.normals[tri->naInd] * barycentric.z + verts.normals[tri->nbInd] * barycentric.x + verts.normals[tri->ncInd] * barycentric.y);
            intersect.uv = verts.uvs[tri->uvaInd] * barycentric.z + verts.uvs[tri->uvbInd] * barycentric.x + verts.uvs[tri->uvcInd] * barycentric.y;
            if (dot(intersect.normal, r.direction) > 0.0f) { intersect.normal = -intersect.normal; intersect.backface = true; }
            else intersect.backface = false;
            intersect.materialID = tri->materialID;
            intersect.emission = tri->emission;
            intersect.valid = true;
            intersect.triIDX = i;
            intersect.dist = t;
            intersect.baryU = barycentric.x; intersect.baryV = barycentric.y;
        }
    }
}

// integratorUtilities.cuh:414-434
static void removeMaterialFromStack(int* stack, int* stackTop, int materialID) {
    int i_found = -1;
    for (int i = (*stackTop) - 1; i > 0; i--) if (stack[i] == materialID) { i_found = i; break; }
    if (i_found != -1) {
        for (int i = i_found; i < (*stackTop) - 1; i++) stack[i] = stack[i + 1];
        (*stackTop)--;
    }
}

static inline float4 sampleSky(const float4&) { return f4(); }     // integratorUtilities.cuh:436-438

// objects.cuh:268-307
Ray generateCameraRay(const Camera& cam, XorwowState& st, int x, int y, PixelCounters* pc) {
    Ray r;
    float aspect = (float)cam.w / (float)cam.h;
    float jitterX = (uniform(st, pc) - 0.5f) * cam.antiAliasJitterDist;
    float jitterY = (uniform(st, pc) - 0.5f) * cam.antiAliasJitterDist;
    float u = (2.0f * ((x + jitterX) / (float)cam.w) - 1.0f) * aspect * cam.fovScale;
    float v = (2.0f * ((y + jitterY) / (float)cam.h) - 1.0f) * cam.fovScale;
    float4 focalPoint = cam.cameraOrigin + (cam.right * (u * cam.focalDist)) + (cam.up * (v * cam.focalDist)) + (cam.forward * cam.focalDist);
    float4 lensOffset = f4(0.0f, 0.0f, 0.0f, 0.0f);
    if (cam.aperture > 0.0f) {
        float r_rnd = uniform(st, pc);
        float theta = 2.0f * 3.141592f * uniform(st, pc);
        float radius = cam.aperture * sqrtf(r_rnd);
        float sn, cs; ref_sincosf(theta, &sn, &cs);
        float lensU = radius * cs;
        float lensV = radius * sn;
        lensOffset = (cam.right * lensU) + (cam.up * lensV);
    }
    r.origin = cam.cameraOrigin + lensOffset;
    r.direction = normalize(focalPoint - r.origin);
    return r;
}

// ---------------------------------------------------------------------------------------
// reflectors.cuh
static inline void cosine_f(const float4& baseColor, float4& newColor) { newColor = baseColor / PI; }          // :10-13
static inline void cosine_pdf(const float4& wo_local, float& pdf) { pdf = ref_fmaxf(wo_local.z, EPSILON) / PI; }  // :15-18

static void cosine_sample_f(XorwowState& st, const float4& baseColor, float4& wo, float4& f_val, float& pdf, PixelCounters* pc) {   // :21-39
    float u1 = uniform(st, pc);
    u1 = ref_fminf(u1, 1.0f - EPSILON);
    float u2 = uniform(st, pc);
    float r = sqrtf(u1);
    float phi = 2.0f * PI * u2;
    float sn, cs; ref_sincosf(phi, &sn, &cs);
    float x = r * cs;
    float y = r * sn;
    float z = sqrtf(1.0f - u1);
    wo = f4(x, y, z);
    cosine_f(baseColor, f_val);
    cosine_pdf(wo, pdf);
}

static inline void mirror_f(float4& f_val, float4 wo) { float c = ref_fmaxf(wo.z, EPSILON); f_val = f4(1.0f / c); }   // :59-63
static inline void mirror_sample_f(float4 wi, float4& wo, float4& f_val, float& pdf) {                              // :70-76
    wo = f4(-wi.x, -wi.y, wi.z);
    float c = ref_fmaxf(wo.z, EPSILON);
    f_val = f4(1.0f / c);
    pdf = 1.0f;
}

static inline float D_GGX(const float4& h, float alpha) {       // :78-84
    float cosThetaH = h.z;
    float alpha2 = alpha * alpha;
    float denom = cosThetaH * cosThetaH * (alpha2 - 1.0f) + 1.0f;
    return alpha2 / (PI * denom * denom);
}
static inline float G1_GGX(const float4& v, float alpha) {      // :92-101
    float cosTheta = v.z;
    float tanTheta = sqrtf(1.0f - cosTheta * cosTheta) / cosTheta;
    float a = 1.0f / (alpha * tanTheta);
    if (a < 1.6f) return (3.535f * a + 2.181f * a * a) / (1.0f + 2.276f * a + 2.577f * a * a);
    return 1.0f;
}
static inline float G_Smith(const float4& wi, const float4& wo, float alpha) { return G1_GGX(wi, alpha) * G1_GGX(wo, alpha); }   // :103-106

static float4 Fresnel_Conductor(float cosTheta, const float4& eta, const float4& k) {    // :108-127 (returns Rs only)
    float4 cosTheta2 = f4(cosTheta * cosTheta);
    float4 sinTheta2 = f4(1.0f) - cosTheta2;
    float4 eta2 = eta * eta;
    float4 k2 = k * k;
    float4 t0 = eta2 - k2 - sinTheta2;
    float4 a2plusb2 = sqrtf4(t0 * t0 + 4.0f * eta2 * k2);
    float4 t1 = a2plusb2 + cosTheta2;
    float4 a = sqrtf4(0.5f * (a2plusb2 + t0));
    float4 t2 = 2.0f * cosTheta * a;
    return (t1 - t2) / (t1 + t2);
}

static void microfacet_metal_f(const float4& eta, const float4& k, float roughness, const float4& wi, const float4& wo, float4& f_val) {   // :129-150
    if (wi.z <= 0.0f || wo.z <= 0.0f) { f_val = f4(0.0f); return; }
    float nDotWi = wi.z, nDotWo = wo.z;
    float4 h = normalize(wi + wo);
    if (h.z <= 0.0f) h = f4(-h.x, -h.y, -h.z);
    float alpha = roughness * roughness;
    float D = D_GGX(h, alpha);
    float G = G_Smith(wi, wo, alpha);
    float4 f = Fresnel_Conductor(dot(wi, h), eta, k);
    f_val = (D * G * f) / ref_fmaxf(4.0f * nDotWi * nDotWo, EPSILON);
}
static void microfacet_pdf(float roughness, const float4& wi, const float4& wo, float& pdf) {   // :152-158
    float4 h = normalize(wi + wo);
    float D = D_GGX(h, roughness * roughness);
    float denom = 4.0f * dot(wo, h);
    pdf = (D * h.z) / denom;
}
static void ggx_sample_h(XorwowState& st, float roughness, float4& h_local, PixelCounters* pc) {   // :163-173 / :514-524
    float u1 = uniform(st, pc);
    float alpha = roughness * roughness;
    float phi = 2.0f * PI * uniform(st, pc);
    float cosTheta = sqrtf((1.0f - u1) / (1.0f + (alpha * alpha - 1.0f) * u1));
    float sinTheta = sqrtf(ref_fmaxf(1.0f - cosTheta * cosTheta, 0.0f));
    float sn, cs; ref_sincosf(phi, &sn, &cs);
    h_local = f4(sinTheta * cs, sinTheta * sn, cosTheta, 0.0f);
}
static void microfacet_metal_sample_f(XorwowState& st, const float4& eta, const float4& k, float roughness, const float4& wi, float4& wo, float4& f_val, float& pdf, PixelCounters* pc) {   // :160-180
    float4 h_local;
    ggx_sample_h(st, roughness, h_local, pc);
    wo = 2.0f * dot(wi, h_local) * h_local - wi;
    if (wo.z <= 0.0f) wo.z = -wo.z;
    microfacet_metal_f(eta, k, roughness, wi, wo, f_val);
    microfacet_pdf(roughness, wi, wo, pdf);
}

// :304-369. Uses mat.ior and `backface` only — the medium-stack etas the caller passes are ignored.
static void dumb_smooth_dielectric_sample_f(XorwowState& st, const float4& wi, float etaSurface, bool backface, int transportMode, float4& wo, float4& f_val, float& pdf, PixelCounters* pc) {
    float etaI, etaT;
    if (backface) { etaI = etaSurface; etaT = 1.0f; } else { etaI = 1.0f; etaT = etaSurface; }
    float cosThetaI = ref_fminf(ref_fmaxf(wi.z, EPSILON), 1.0f);
    float eta = etaI / etaT;
    float cosThetaT2 = 1.0f - eta * eta * (1.0f - cosThetaI * cosThetaI);
    float F = schlick_fresnel(cosThetaI, etaI, etaT);
    if (cosThetaT2 < 0.0f || F >= 0.99999f) {
        wo = f4(-wi.x, -wi.y, wi.z);
        f_val = f4(1.0f / ref_fmaxf(wo.z, EPSILON));
        pdf = 1.0f;
        return;
    }
    if (uniform(st, pc) < F) {
        wo = f4(-wi.x, -wi.y, wi.z);
        pdf = F;
        f_val = f4(F / ref_fmaxf(wo.z, EPSILON));
    } else {
        wo = f4(-eta * wi.x, -eta * wi.y, -(sqrtf(cosThetaT2)));
        float denom = ref_fmaxf(fabsf(wo.z), EPSILON);
        f_val = f4((1.0f - F) / denom);
        pdf = 1.0f - F;
        if (transportMode == TRANSPORTMODE_RADIANCE) f_val *= eta * eta;
    }
}

// :371-417 bilinear, wrap. App. D: a 0x0 texture (missing BMP, imageUtil.cu:146-149) would divide
// by zero in the reference; defined as "leave albedo unchanged".
static void sampleTexture(const Material& mat, const Scene& sc, const float2 uv, float4& albedo) {
    int width = mat.width, height = mat.height;
    if (width <= 0 || height <= 0) return;
    float fx = uv.x * width - 0.5f;
    float fy = uv.y * height - 0.5f;
    int x_int = (int)floorf(fx), y_int = (int)floorf(fy);
    float sx = fx - floorf(fx), sy = fy - floorf(fy);
    auto wrap = [](int val, int dim) { int r = val % dim; return r < 0 ? r + dim : r; };
    int x0 = wrap(x_int, width), y0 = wrap(y_int, height), x1 = wrap(x_int + 1, width), y1 = wrap(y_int + 1, height);
    float4 c00 = sc.textures[mat.startInd + y0 * width + x0];
    float4 c10 = sc.textures[mat.startInd + y0 * width + x1];
    float4 c01 = sc.textures[mat.startInd + y1 * width + x0];
    float4 c11 = sc.textures[mat.startInd + y1 * width + x1];
    float4 bottom = c00 * (1.0f - sx) + c10 * sx;
    float4 top = c01 * (1.0f - sx) + c11 * sx;
    albedo = bottom * (1.0f - sy) + top * sy;
}

static void leaf_f(const float4& albedo, float ior, float currIOR, float roughness, float transmission, const float4& wi, const float4& wo, float4& f_val) {   // :420-461
    bool is_reflection = wo.z * wi.z > 0.0f;
    float F = schlick_fresnel(wi.z, currIOR, ior);
    if (is_reflection) {
        float4 h = normalize(wi + wo);
        float microfacet_F = schlick_fresnel(dot(wi, h), currIOR, ior);
        float nDotWi = wi.z, nDotWo = wo.z;
        if (h.z <= 0.0f) h = -h;
        float alpha = roughness * roughness;
        float D = D_GGX(h, alpha);
        float G = G_Smith(wi, wo, alpha);
        float4 f_cuticle = f4(D * G * microfacet_F / ref_fmaxf(4.0f * nDotWi * nDotWo, EPSILON));
        float4 f_diffuse_val;
        cosine_f(albedo, f_diffuse_val);
        f_val = (1.0f - microfacet_F) * (1.0f - transmission) * f_diffuse_val + f_cuticle;
    } else {
        cosine_f(albedo, f_val);
        f_val *= transmission * (1.0f - F);
    }
}
static void leaf_pdf(float ior, float currIOR, float roughness, float transmission, const float4& wi, const float4& wo, float& pdf) {   // :463-506
    bool is_reflection = wo.z * wi.z > 0.0f;
    float F = schlick_fresnel(fabsf(wi.z), currIOR, ior);
    F = ref_fminf(F, 1.0f - 0.1f * roughness);
    float p_specular = F;
    float p_diffuse_refl = (1.0f - F) * (1.0f - transmission);
    float p_diffuse_trans = (1.0f - F) * transmission;
    if (is_reflection) {
        float4 h = normalize(wi + wo);
        if (h.z < 0.0f) h = -h;
        float alpha = roughness * roughness;
        float D = D_GGX(h, alpha);
        float denom = 4.0f * dot(wo, h);
        float pdf_cuticle_bounce = (D * h.z) / denom;
        float pdf_diffuse;
        cosine_pdf(wo, pdf_diffuse);
        pdf = (p_specular * pdf_cuticle_bounce) + (p_diffuse_refl * pdf_diffuse);
    } else {
        float pdf_trans;
        cosine_pdf(-wo, pdf_trans);
        pdf = pdf_trans * p_diffuse_trans;
    }
}
static void leaf_sample_f(XorwowState& st, const float4& wi, float ior, float currIOR, float roughness, const float4& albedo, float transmission, float4& wo, float4& f_val, float& pdf, PixelCounters* pc) {   // :508-543
    float F = schlick_fresnel(wi.z, currIOR, ior);
    if (uniform(st, pc) < F) {
        float4 h_local;
        ggx_sample_h(st, roughness, h_local, pc);
        wo = 2.0f * dot(wi, h_local) * h_local - wi;
    } else {
        if (uniform(st, pc) < transmission) { cosine_sample_f(st, albedo, wo, f_val, pdf, pc); wo.z = -wo.z; }
        else cosine_sample_f(st, albedo, wo, f_val, pdf, pc);
    }
    leaf_f(albedo, ior, currIOR, roughness, transmission, wi, wo, f_val);
    leaf_pdf(ior, currIOR, roughness, transmission, wi, wo, pdf);
}

// :547-584. Quirks kept: MAT_DIFFUSE evaluates mat.albedo, not the texture-sampled albedo (:566);
// dielectrics write nothing (:572-575). App. D: f_val starts at 0 so "nothing" is defined; the
// transmission map is sampled through the ALBEDO texture's start/size (sampleTexture takes `mat`).
void f_eval(const Scene& sc, int materialID, const float4& wi, const float4& wo, float etaI, float etaT, float4& f_val, float2 uv, int transportMode) {
    (void)etaT; (void)transportMode;
    const Material& mat = sc.mats[materialID];
    float4 albedo = mat.albedo;
    if (mat.hasTexture) sampleTexture(mat, sc, uv, albedo);
    float trans = mat.transmission;
    if (mat.hasTransMap) { float4 trans4 = f4(trans); sampleTexture(mat, sc, uv, trans4); trans = trans4.x; }
    if (mat.type == MAT_DIFFUSE) cosine_f(mat.albedo, f_val);
    else if (mat.type == MAT_METAL) microfacet_metal_f(mat.eta, mat.k, mat.roughness, -wi, wo, f_val);
    else if (mat.type == MAT_SMOOTHDIELECTRIC) {}
    else if (mat.type == MAT_LEAF) leaf_f(albedo, mat.ior, etaI, mat.roughness, trans, -wi, wo, f_val);
    else if (mat.type == MAT_DELTAMIRROR) mirror_f(f_val, wo);
}

// :588-629
void sample_f_eval(XorwowState& st, const Scene& sc, int materialID, const float4& wi, float etaI, float etaT, bool backface, float4& wo, float4& f_val, float& pdf, float2 uv, PixelCounters* pc, int transportMode) {
    (void)etaT;
    const Material& mat = sc.mats[materialID];
    float4 albedo = mat.albedo;
    if (mat.hasTexture) sampleTexture(mat, sc, uv, albedo);
    float trans = mat.transmission;
    if (mat.hasTransMap) { float4 trans4 = f4(trans); sampleTexture(mat, sc, uv, trans4); trans = trans4.x; }
    if (mat.type == MAT_DIFFUSE) cosine_sample_f(st, albedo, wo, f_val, pdf, pc);
    else if (mat.type == MAT_METAL) microfacet_metal_sample_f(st, mat.eta, mat.k, mat.roughness, -wi, wo, f_val, pdf, pc);
    else if (mat.type == MAT_SMOOTHDIELECTRIC) dumb_smooth_dielectric_sample_f(st, -wi, mat.ior, backface, transportMode, wo, f_val, pdf, pc);
    else if (mat.type == MAT_LEAF) leaf_sample_f(st, -wi, mat.ior, etaI, mat.roughness, albedo, trans, wo, f_val, pdf, pc);
    else if (mat.type == MAT_DELTAMIRROR) mirror_sample_f(-wi, wo, f_val, pdf);
}

// :633-666
void pdf_eval(const Scene& sc, int materialID, const float4& wi, const float4& wo, float etaI, float etaT, float& pdf, float2 uv) {
    (void)etaT;
    const Material& mat = sc.mats[materialID];
    float trans = mat.transmission;
    if (mat.hasTransMap) { float4 trans4 = f4(trans); sampleTexture(mat, sc, uv, trans4); trans = trans4.x; }
    if (mat.type == MAT_DIFFUSE) cosine_pdf(wo, pdf);
    else if (mat.type == MAT_METAL) microfacet_pdf(mat.roughness, -wi, wo, pdf);
    else if (mat.type == MAT_SMOOTHDIELECTRIC) pdf = 0.0f;
    else if (mat.type == MAT_LEAF) leaf_pdf(mat.ior, etaI, mat.roughness, trans, -wi, wo, pdf);
    else if (mat.type == MAT_DELTAMIRROR) pdf = 1.0f;
}

// ---------------------------------------------------------------------------------------
// deviceCode.cu:63-85
static void neePDF(const Scene& sc, int lightNum, int lightTriInd, const float4& prevPoint, float& light_pdf, const Intersection* newIntersect) {
    const Triangle& l = sc.mesh[lightTriInd];
    float4 apos = sc.points[l.aInd], bpos = sc.points[l.bInd], cpos = sc.points[l.cInd];
    float4 p = newIntersect->point;
    float4 surfaceToLight = p - prevPoint;
    float4 wi = normalize(surfaceToLight);
    float distanceSQR = lengthSquared(surfaceToLight);
    float4 lightNormal = sc.normals[l.naInd];
    float cosThetaLight = dot(lightNormal, -wi);
    float area = 0.5f * length(cross3(bpos - apos, cpos - apos));
    light_pdf = distanceSQR / (cosThetaLight * lightNum * area);
}

// deviceCode.cu:87-156
static void nextEventEstimation(XorwowState& st, const Scene& sc, int lightNum, const Intersection& intersect, const float4& wo,
                                float& light_pdf, float4& contribution, float4& surfaceToLight_local, float etaI, float etaT, PixelCounters* pc) {
    contribution = f4(0.0f, 0.0f, 0.0f);
    if (lightNum == 0) { light_pdf = -1.0f; return; }
    int index = std::min(static_cast<int>(uniform(st, pc) * lightNum), lightNum - 1);
    const Triangle l = sc.lights[index];
    float4 apos = sc.points[l.aInd], bpos = sc.points[l.bInd], cpos = sc.points[l.cInd];
    float u = sqrtf(uniform(st, pc));
    float v = uniform(st, pc);
    float4 p = (1.0f - u) * apos + u * (1.0f - v) * bpos + u * v * cpos;
    float4 n = intersect.normal;
    float4 surfaceToLight = p - intersect.point;
    float4 wi = normalize(surfaceToLight);
    Ray r; r.origin = intersect.point + wi * EPSILON; r.direction = wi;
    // App. D: `t` is uninitialised in the reference when this test fails (:121-123); defined as the
    // geometric distance to the sampled point minus EPSILON.
    float t = length(surfaceToLight) - EPSILON;
    float4 dummy;
    triangleIntersect(vertsOf(sc), l, r, dummy, t);
    float4 throughputScale = f4(1.0f);
    BVHShadowRay(r, sc, throughputScale, t * (1.0f - EPSILON), -1, pc);
    if (lengthSquared(throughputScale) > 0.0f) {
        float distanceSQR = lengthSquared(surfaceToLight);
        float4 lightNormal = sc.normals[l.naInd];
        float cosThetaLight = dot(lightNormal, -wi);
        float cosThetaSurface = fabsf(dot(n, wi));
        float area = 0.5f * length(cross3(bpos - apos, cpos - apos));
        light_pdf = distanceSQR / (cosThetaLight * lightNum * area);
        float4 Le = l.emission;
        float4 f_val = f4();
        float4 wi_local;
        toLocal(wi, intersect.normal, wi_local);
        surfaceToLight_local = wi_local;
        f_eval(sc, intersect.materialID, wo, wi_local, etaI, etaT, f_val, intersect.uv);
        contribution = f_val * Le * cosThetaSurface / light_pdf;
        contribution *= throughputScale;
    }
}

// deviceCode.cu:158-205 — one sample
float4 Li_naive_unidirectional(XorwowState& st, const Camera& cam, const Scene& sc, int maxDepth, int x, int y, PixelCounters* pc) {
    Ray r = generateCameraRay(cam, st, x, y, pc);
    float4 Li = f4();
    float4 beta = f4(1.0f);
    for (int depth = 0; depth < maxDepth; depth++) {
        if (pc) pc->iterations++;
        Intersection intersect;
        BVHSceneIntersect(r, sc, intersect, 999999.0f, -1, pc);
        if (!intersect.valid) { Li += beta * sampleSky(r.direction); break; }
        float4 toSurface_local, toNext_local = f4();
        toLocal(r.direction, intersect.normal, toSurface_local);
        float4 f_val = f4();
        float pdf = 0.0f;          // App. D: undefined for material types without a dispatch arm
        sample_f_eval(st, sc, intersect.materialID, toSurface_local, 1.0f, 1.0f, intersect.backface, toNext_local, f_val, pdf, intersect.uv, pc);
        if (pdf <= 0.0f || lengthSquared(f_val) < EPSILON) break;
        Li += intersect.emission * beta;
        beta *= f_val * fabsf(toNext_local.z) / pdf;
        float4 toNext_world;
        toWorld(toNext_local, intersect.normal, toNext_world);
        r.origin = intersect.point + ((toNext_local.z > 0.0f) ? (intersect.normal * RAY_EPSILON) : (-intersect.normal * RAY_EPSILON));
        r.direction = toNext_world;
    }
    return Li;
}

// deviceCode.cu:285-542 — one sample
float4 Li_unidirectional(XorwowState& st, const Camera& cam, const Scene& sc, int maxDepth, bool useMIS, int x, int y, PixelCounters* pc) {
    const std::vector<Material>& materials = sc.mats;
    const int lightNum = (int)sc.lights.size();
    float4 beta = f4(1.0f, 1.0f, 1.0f);
    float4 Li = f4();
    float4 wi_local = f4(), wo_local = f4();
    float4 prevRealPoint = f4();            // previousintersectREAL.point is the only field ever read (:446 -> :73)
    int mediumStack[16];
    int stackTop = 0;
    mediumStack[stackTop++] = 0;
    Ray r = generateCameraRay(cam, st, x, y, pc);
    float pdf = EPSILON, etaI = EPSILON, etaT = EPSILON;
    bool hitFirstnonSpecular = false;
    int guard = 0;                          // false hits do `depth--`; bound the total
    for (int depth = 0; depth < 100; depth++) {
        if (++guard > 4096) break;
        if (pc) pc->iterations++;
        Intersection intersect;
        BVHSceneIntersect(r, sc, intersect, 999999.0f, -1, pc);
        if (!intersect.valid) { Li += beta * sampleSky(r.direction); break; }
        int materialID = intersect.materialID;
        toLocal(r.direction, intersect.normal, wi_local);
        bool isSpecular = materials[materialID].isSpecular;
        bool trueHit = true;

        int minPrior = materials[mediumStack[0]].priority;
        int minPriorID = mediumStack[0];
        for (int i = 1; i < stackTop; i++)
            if (materials[mediumStack[i]].priority < minPrior) { minPrior = materials[mediumStack[i]].priority; minPriorID = mediumStack[i]; }

        float4 absorption_coeff = materials[minPriorID].absorption;
        float distanceTraveled = intersect.dist;
        if (distanceTraveled > EPSILON) {
            float4 attenuation = f4(ref_expf(-absorption_coeff.x * distanceTraveled), ref_expf(-absorption_coeff.y * distanceTraveled), ref_expf(-absorption_coeff.z * distanceTraveled));
            beta *= attenuation;
        }

        if (materials[materialID].boundary) {
            if (materials[materialID].priority <= minPrior) {
                if (materials[materialID].type == MAT_SMOOTHDIELECTRIC) {
                    etaI = materials[minPriorID].ior;
                    if (!intersect.backface) etaT = materials[materialID].ior;
                    else {
                        if (stackTop == 1) etaT = 1.0f;
                        else {
                            minPrior = 99;
                            int secondLowest = mediumStack[0];
                            for (int i = 0; i < stackTop; i++) {
                                if (materials[mediumStack[i]].priority) {
                                    if (minPrior > materials[mediumStack[i]].priority && mediumStack[i] != materialID) { secondLowest = mediumStack[i]; minPrior = materials[mediumStack[i]].priority; }
                                }
                            }
                            etaT = materials[secondLowest].ior;
                        }
                    }
                }
            } else {
                trueHit = false;
                if (!intersect.backface) { if (stackTop < 16) mediumStack[stackTop++] = intersect.materialID; }   // App. D: push beyond 16 ignored
                else removeMaterialFromStack(mediumStack, &stackTop, materialID);
            }
        } else etaI = materials[minPriorID].ior;

        if (trueHit) {
            if (lengthSquared(intersect.emission) > EPSILON) {
                if (depth == 0 || !hitFirstnonSpecular) Li += beta * intersect.emission;
                else if (useMIS && !isSpecular) {
                    float light_pdf = EPSILON;
                    neePDF(sc, lightNum, intersect.triIDX, prevRealPoint, light_pdf, &intersect);
                    if (light_pdf > EPSILON) {
                        float bsdfWeight = pdf * pdf / (light_pdf * light_pdf + pdf * pdf);
                        Li += beta * intersect.emission * bsdfWeight;
                    }
                }
            }
            if (useMIS && lengthSquared(intersect.emission) < EPSILON && !isSpecular) {
                float4 nee;
                float light_pdf = EPSILON;
                nextEventEstimation(st, sc, lightNum, intersect, wi_local, light_pdf, nee, wo_local, etaI, etaT, pc);
                if (light_pdf > EPSILON) {
                    pdf_eval(sc, materialID, wi_local, wo_local, etaI, etaT, pdf, intersect.uv);
                    float neeWeight = light_pdf * light_pdf / (pdf * pdf + light_pdf * light_pdf);
                    Li += beta * nee * neeWeight;
                }
            }
            float4 f_val = f4();
            sample_f_eval(st, sc, materialID, wi_local, etaI, etaT, intersect.backface, wo_local, f_val, pdf, intersect.uv, pc);
            float4 wo_world = f4();
            toWorld(wo_local, intersect.normal, wo_world);
            pdf = ref_fmaxf(pdf, 0.01f);
            if (wo_local.z < 0.0f) {
                if (!intersect.backface) { if (stackTop < 16) mediumStack[stackTop++] = intersect.materialID; }
                else removeMaterialFromStack(mediumStack, &stackTop, materialID);
            }
            beta *= (f_val * fabsf(wo_local.z) / pdf);
            if (wo_local.z > 0) r.origin = intersect.point + intersect.normal * EPSILON;
            else r.origin = intersect.point - intersect.normal * EPSILON;
            r.direction = normalize(wo_world);
            prevRealPoint = intersect.point;
        } else {
            toLocal(r.direction, intersect.normal, wo_local);
            r.origin = intersect.point + r.direction * RAY_EPSILON;
            depth--;
        }
        if (depth > maxDepth) {
            float luminance = dot(beta, f4(0.2126f, 0.7152f, 0.0722f));
            float p = clampf(luminance, 0.05f, 0.99f);
            if (uniform(st, pc) > p) break;
            beta /= p;
        }
        if (!isSpecular) hitFirstnonSpecular = true;
    }
    return Li;
}

// deviceCode.cu:544-573 / 207-236: initRNG(seed, y*w+x) then numSample launches, each adding
// one sample per pixel into `colors` with the per-pixel stream continuing across samples.
void launch(int integrator, int maxDepth, const Camera& cam, const Scene& sc, int numSample, bool useMIS,
            int w, int h, uint64_t seed, int x0, int y0, int x1, int y1,
            float4* colors, PixelCounters* counters, int nThreads) {
    (void)h;
    xorwow_jump_table();                        // build once before threads start
    auto rows = [&](int ya, int yb) {
        for (int y = ya; y < yb; y++)
            for (int x = x0; x < x1; x++) {
                int pixelIdx = y * w + x;
                XorwowState st;
                xorwow_init(st, seed, (uint32_t)pixelIdx);
                PixelCounters* pc = counters ? &counters[pixelIdx] : nullptr;
                float4 acc = colors[pixelIdx];
                for (int s = 0; s < numSample; s++) {
                    float4 Li = (integrator == NAIVE_UNIDIRECTIONAL) ? Li_naive_unidirectional(st, cam, sc, maxDepth, x, y, pc)
                                                                     : Li_unidirectional(st, cam, sc, maxDepth, useMIS, x, y, pc);
                    acc += Li;
                }
                colors[pixelIdx] = acc;
            }
    };
    nThreads = std::max(1, nThreads);
    if (nThreads == 1) { rows(y0, y1); return; }
    // interleave 8-row bands over the threads for balance
    std::vector<std::thread> pool;
    for (int t = 0; t < nThreads; t++)
        pool.emplace_back([&, t]() {
            for (int ya = y0 + 8 * t; ya < y1; ya += 8 * nThreads) rows(ya, std::min(ya + 8, y1));
        });
    for (auto& th : pool) th.join();
}

// main.cu:860-870
void finalise(float4* colors, int n, int sampleCount) {
    for (int i = 0; i < n; i++) {
        colors[i] /= (float)sampleCount;
        if (std::isnan(colors[i].x) || std::isnan(colors[i].y) || std::isnan(colors[i].z)) colors[i] = f4(1.0f, 0.0f, 1.0f);
        if (std::isinf(colors[i].x) || std::isinf(colors[i].y) || std::isinf(colors[i].z)) colors[i] = f4(0.0f, 1.0f, 0.0f);
    }
}

}  // namespace oracle
