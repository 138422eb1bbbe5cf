// pt_shade.h — camera, shading frame and BSDFs of the path (device side).
//
// Restates, on register 3-vectors and the packed PMat record, what the reference does in
// objects.cuh:268-307 (generateCameraRay), util.cuh:163-185 (toLocal/toWorld) and
// reflectors.cuh (all BSDF eval / sample / pdf arms and the dispatchers :547-666), expression
// by expression: operand order and grouping are part of the result.
#pragma once
#include "pt_trace.h"

namespace pt {

template <bool COUNT>
PT_DEV float draw(Rng& r, Ctr& c) { if (COUNT) c.draws++; return rng_uniform(r); }

// objects.cuh:268-307
template <bool COUNT>
PT_DEV void camera_ray(const CamK& cam, Rng& rng, int x, int y, V3& o, V3& d, Ctr& c) {
    float aspect = (float)cam.w / (float)cam.h;
    float jitterX = (draw<COUNT>(rng, c) - 0.5f) * cam.jitter;
    float jitterY = (draw<COUNT>(rng, c) - 0.5f) * cam.jitter;
    float u = (2.0f * (((float)x + jitterX) / (float)cam.w) - 1.0f) * aspect * cam.fovScale;
    float v = (2.0f * (((float)y + jitterY) / (float)cam.h) - 1.0f) * cam.fovScale;
    V3 focal = cam.origin + (cam.right * (u * cam.focalDist)) + (cam.up * (v * cam.focalDist)) + (cam.forward * cam.focalDist);
    V3 lens = v3(0.0f);
    if (cam.aperture > 0.0f) {
        float r_rnd = draw<COUNT>(rng, c);
        float theta = 2.0f * 3.141592f * draw<COUNT>(rng, c);
        float radius = cam.aperture * __builtin_sqrtf(r_rnd);
        float sn, cs; sincos_(theta, sn, cs);
        lens = (cam.right * (radius * cs)) + (cam.up * (radius * sn));
    }
    o = cam.origin + lens;
    d = normalize(focal - o);
}

// util.cuh:163-185
PT_DEV V3 onb_tangent(V3 n) {
    if (__builtin_fabsf(n.x) > __builtin_fabsf(n.z)) return normalize(v3(-n.y, n.x, 0.0f));
    return normalize(v3(0.0f, -n.z, n.y));
}
PT_DEV V3 to_world(V3 l, V3 n) {
    V3 t = onb_tangent(n);
    V3 b = cross(n, t);
    return l.x * t + l.y * b + l.z * n;
}
PT_DEV V3 to_local(V3 w, V3 n) {
    V3 t = onb_tangent(n);
    V3 b = cross(n, t);
    return v3(dot(w, t), dot(w, b), dot(w, n));
}
// The reference rebuilds the basis in every toLocal / toWorld call (util.cuh:163-185), three times per bounce on the
// same normal; it is a pure function of n, so one bounce builds it once (same values).
struct Onb { V3 t, b, n; };
PT_DEV Onb onb_of(V3 n) { Onb f; f.t = onb_tangent(n); f.b = cross(n, f.t); f.n = n; return f; }
PT_DEV V3 to_world(V3 l, const Onb& f) { return l.x * f.t + l.y * f.b + l.z * f.n; }
PT_DEV V3 to_local(V3 w, const Onb& f) { return v3(dot(w, f.t), dot(w, f.b), dot(w, f.n)); }

// ---- reflectors.cuh ---------------------------------------------------------------------
PT_DEV V3 cosine_f(V3 base) { return base / kPi; }                                           // :10-13
PT_DEV float cosine_pdf(V3 wo) { return fmaxf_(wo.z, kEps) / kPi; }                           // :15-18

// fPre: cosine_f(base) where the caller already has it (untextured materials: albedo / PI from scene set-up), else null.
template <bool COUNT>
PT_DEV void cosine_sample_f(Rng& rng, V3 base, V3& wo, V3& f, float& pdf, Ctr& c, const float* fPre = nullptr) {          // :21-39
    float u1 = draw<COUNT>(rng, c);
    u1 = fminf_(u1, 1.0f - kEps);
    float u2 = draw<COUNT>(rng, c);
    float r = __builtin_sqrtf(u1);
    float phi = 2.0f * kPi * u2;
    float sn, cs; sincos_(phi, sn, cs);
    wo = v3(r * cs, r * sn, __builtin_sqrtf(1.0f - u1));
    f = fPre ? ld3(fPre) : cosine_f(base);
    pdf = cosine_pdf(wo);
}

PT_DEV float D_GGX(V3 h, float alpha) {                                                        // :78-84
    float c = h.z;
    float a2 = alpha * alpha;
    float denom = c * c * (a2 - 1.0f) + 1.0f;
    return a2 / (kPi * denom * denom);
}
PT_DEV float G1_GGX(V3 v, float alpha) {                                                       // :92-101
    float c = v.z;
    float tanTheta = __builtin_sqrtf(1.0f - c * c) / c;
    float a = 1.0f / (alpha * tanTheta);
    if (a < 1.6f) return (3.535f * a + 2.181f * a * a) / (1.0f + 2.276f * a + 2.577f * a * a);
    return 1.0f;
}
PT_DEV float G_Smith(V3 wi, V3 wo, float alpha) { return G1_GGX(wi, alpha) * G1_GGX(wo, alpha); }   // :103-106
PT_DEV V3 sqrt3(V3 v) { return v3(__builtin_sqrtf(v.x), __builtin_sqrtf(v.y), __builtin_sqrtf(v.z)); }

PT_DEV V3 fresnel_conductor(float cosTheta, V3 eta, V3 k) {                                    // :108-127 (Rs only)
    V3 c2 = v3(cosTheta * cosTheta);
    V3 s2 = v3(1.0f) - c2;
    V3 eta2 = eta * eta;
    V3 k2 = k * k;
    V3 t0 = eta2 - k2 - s2;
    V3 a2b2 = sqrt3(t0 * t0 + 4.0f * eta2 * k2);
    V3 t1 = a2b2 + c2;
    V3 a = sqrt3(0.5f * (a2b2 + t0));
    V3 t2 = (2.0f * cosTheta) * a;
    return (t1 - t2) / (t1 + t2);
}

PT_DEV V3 microfacet_metal_f(V3 eta, V3 k, float roughness, V3 wi, V3 wo) {                    // :129-150
    if (wi.z <= 0.0f || wo.z <= 0.0f) return v3(0.0f);
    V3 h = normalize(wi + wo);
    if (h.z <= 0.0f) h = v3(-h.x, -h.y, -h.z);
    float alpha = roughness * roughness;
    float D = D_GGX(h, alpha);
    float G = G_Smith(wi, wo, alpha);
    V3 f = fresnel_conductor(dot(wi, h), eta, k);
    return ((D * G) * f) / fmaxf_(4.0f * wi.z * wo.z, kEps);
}
PT_DEV float microfacet_pdf(float roughness, V3 wi, V3 wo) {                                   // :152-158
    V3 h = normalize(wi + wo);
    float D = D_GGX(h, roughness * roughness);
    float denom = 4.0f * dot(wo, h);
    return (D * h.z) / denom;
}
template <bool COUNT>
PT_DEV V3 ggx_sample_h(Rng& rng, float roughness, Ctr& c) {                                     // :163-173, :514-524
    float u1 = draw<COUNT>(rng, c);
    float alpha = roughness * roughness;
    float phi = 2.0f * kPi * draw<COUNT>(rng, c);
    float cosTheta = __builtin_sqrtf((1.0f - u1) / (1.0f + (alpha * alpha - 1.0f) * u1));
    float sinTheta = __builtin_sqrtf(fmaxf_(1.0f - cosTheta * cosTheta, 0.0f));
    float sn, cs; sincos_(phi, sn, cs);
    return v3(sinTheta * cs, sinTheta * sn, cosTheta);
}

// :304-369 — ignores the caller's medium etas; uses mat.ior and `backface` only.
template <bool COUNT>
PT_DEV void dielectric_sample_f(Rng& rng, V3 wi, float etaSurface, bool backface, V3& wo, V3& f, float& pdf, Ctr& c) {
    float etaI = backface ? etaSurface : 1.0f;
    float etaT = backface ? 1.0f : etaSurface;
    float cosI = fminf_(fmaxf_(wi.z, kEps), 1.0f);
    float eta = etaI / etaT;
    float cosT2 = 1.0f - eta * eta * (1.0f - cosI * cosI);
    float F = schlick_fresnel(cosI, etaI, etaT);
    if (cosT2 < 0.0f || F >= 0.99999f) {
        wo = v3(-wi.x, -wi.y, wi.z);
        f = v3(1.0f / fmaxf_(wo.z, kEps));
        pdf = 1.0f;
        return;
    }
    if (draw<COUNT>(rng, c) < F) {
        wo = v3(-wi.x, -wi.y, wi.z);
        pdf = F;
        f = v3(F / fmaxf_(wo.z, kEps));
    } else {
        wo = v3(-eta * wi.x, -eta * wi.y, -(__builtin_sqrtf(cosT2)));
        float denom = fmaxf_(__builtin_fabsf(wo.z), kEps);
        f = v3((1.0f - F) / denom);
        pdf = 1.0f - F;
        f = f * (eta * eta);                 // TRANSPORTMODE_RADIANCE (:364-367)
    }
}

// :371-417 — bilinear, wrap; a 0x0 texture leaves the value unchanged.
PT_DEV void sample_texture(const PMat& m, const float4* tex, float uvx, float uvy, V3& albedo) {
    int width = m.texW, height = m.texH;
    if (width <= 0 || height <= 0) return;
    float fx = uvx * (float)width - 0.5f;
    float fy = uvy * (float)height - 0.5f;
    float flx = __builtin_floorf(fx), fly = __builtin_floorf(fy);
    int xi = (int)flx, yi = (int)fly;
    float sx = fx - flx, sy = fy - fly;
    auto wrap = [](int val, int dim) { int r = val % dim; return r < 0 ? r + dim : r; };
    int x0 = wrap(xi, width), y0 = wrap(yi, height), x1 = wrap(xi + 1, width), y1 = wrap(yi + 1, height);
    float4 c00 = tex[m.texStart + y0 * width + x0], c10 = tex[m.texStart + y0 * width + x1];
    float4 c01 = tex[m.texStart + y1 * width + x0], c11 = tex[m.texStart + y1 * width + x1];
    V3 bottom = v3(c00.x, c00.y, c00.z) * (1.0f - sx) + v3(c10.x, c10.y, c10.z) * sx;
    V3 top = v3(c01.x, c01.y, c01.z) * (1.0f - sx) + v3(c11.x, c11.y, c11.z) * sx;
    albedo = bottom * (1.0f - sy) + top * sy;
}

PT_DEV V3 leaf_f(V3 albedo, float ior, float currIOR, float roughness, float transmission, V3 wi, V3 wo) {   // :420-461
    bool refl = wo.z * wi.z > 0.0f;
    float F = schlick_fresnel(wi.z, currIOR, ior);
    if (refl) {
        V3 h = normalize(wi + wo);
        float mF = schlick_fresnel(dot(wi, h), currIOR, ior);
        if (h.z <= 0.0f) h = -h;
        float alpha = roughness * roughness;
        float D = D_GGX(h, alpha);
        float G = G_Smith(wi, wo, alpha);
        V3 cut = v3(D * G * mF / fmaxf_(4.0f * wi.z * wo.z, kEps));
        V3 fd = cosine_f(albedo);
        return ((1.0f - mF) * (1.0f - transmission)) * fd + cut;
    }
    V3 f = cosine_f(albedo);
    return f * (transmission * (1.0f - F));
}
PT_DEV float leaf_pdf(float ior, float currIOR, float roughness, float transmission, V3 wi, V3 wo) {   // :463-506
    bool refl = wo.z * wi.z > 0.0f;
    float F = schlick_fresnel(__builtin_fabsf(wi.z), currIOR, ior);
    F = fminf_(F, 1.0f - 0.1f * roughness);
    float pS = F;
    float pR = (1.0f - F) * (1.0f - transmission);
    float pT = (1.0f - F) * transmission;
    if (refl) {
        V3 h = normalize(wi + wo);
        if (h.z < 0.0f) h = -h;
        float alpha = roughness * roughness;
        float D = D_GGX(h, alpha);
        float denom = 4.0f * dot(wo, h);
        float pc = (D * h.z) / denom;
        return (pS * pc) + (pR * cosine_pdf(wo));
    }
    return cosine_pdf(-wo) * pT;
}
template <bool COUNT>
PT_DEV void leaf_sample_f(Rng& rng, V3 wi, float ior, float currIOR, float roughness, V3 albedo, float transmission, V3& wo, V3& f, float& pdf, Ctr& c) {   // :508-543
    float F = schlick_fresnel(wi.z, currIOR, ior);
    if (draw<COUNT>(rng, c) < F) {
        V3 h = ggx_sample_h<COUNT>(rng, roughness, c);
        wo = (2.0f * dot(wi, h)) * h - wi;
    } else {
        bool through = draw<COUNT>(rng, c) < transmission;
        cosine_sample_f<COUNT>(rng, albedo, wo, f, pdf, c);
        if (through) wo.z = -wo.z;
    }
    f = leaf_f(albedo, ior, currIOR, roughness, transmission, wi, wo);
    pdf = leaf_pdf(ior, currIOR, roughness, transmission, wi, wo);
}

// LEAN (chosen per scene by the host, pt_api.hip): no triangle's material is a MAT_LEAF and none has a texture or a
// transmission map — Cornell boxes of glass, mirrors and metals, a glass blob. The leaf arms (the largest of the five) and the
// bilinear texture fetch are then dead code, known at compile time: same values, a third less code and fewer live registers.
template <bool LEAN = false>
PT_DEV void material_inputs(const PMat& m, const float4* tex, float uvx, float uvy, bool wantAlbedo, V3& albedo, float& trans) {
    albedo = ld3(m.albedo);
    if (LEAN) { trans = m.transmission; return; }
    if (wantAlbedo && (m.flags & kMatHasTexture)) sample_texture(m, tex, uvx, uvy, albedo);
    trans = m.transmission;
    if (m.flags & kMatHasTransMap) {       // sampled through the albedo texture's start/size (reflectors.cuh:558-562)
        V3 t4 = v3(trans);
        sample_texture(m, tex, uvx, uvy, t4);
        trans = t4.x;
    }
}

// f_eval, reflectors.cuh:547-584. wi points INTO the surface (negated inside). MAT_DIFFUSE uses
// mat.albedo, not the sampled texture (:566); dielectrics and unknown types leave f at 0.
template <bool LEAN = false>
PT_DEV V3 f_eval(const PMat& m, const float4* tex, V3 wi, V3 wo, float etaI, float uvx, float uvy) {
    V3 albedo; float trans;
    material_inputs<LEAN>(m, tex, uvx, uvy, true, albedo, trans);
    if (m.type == 0) return ld3(m.albedoOverPi);                 // cosine_f(mat.albedo) = albedo / PI, divided once at scene set-up (same IEEE division)
    if (m.type == 1) return microfacet_metal_f(ld3(m.eta), ld3(m.k), m.roughness, -wi, wo);
    if (!LEAN && m.type == 4) return leaf_f(albedo, m.ior, etaI, m.roughness, trans, -wi, wo);
    if (m.type == 6) return v3(1.0f / fmaxf_(wo.z, kEps));           // mirror_f :59-63
    return v3(0.0f);
}

// pdf_eval, reflectors.cuh:633-666. Returns false when no arm writes `pdf` (it then keeps its old value).
template <bool LEAN = false>
PT_DEV bool pdf_eval(const PMat& m, const float4* tex, V3 wi, V3 wo, float etaI, float uvx, float uvy, float& pdf) {
    V3 albedo; float trans;
    material_inputs<LEAN>(m, tex, uvx, uvy, false, albedo, trans);
    if (m.type == 0) { pdf = cosine_pdf(wo); return true; }
    if (m.type == 1) { pdf = microfacet_pdf(m.roughness, -wi, wo); return true; }
    if (m.type == 2) { pdf = 0.0f; return true; }
    if (!LEAN && m.type == 4) { pdf = leaf_pdf(m.ior, etaI, m.roughness, trans, -wi, wo); return true; }
    if (m.type == 6) { pdf = 1.0f; return true; }
    return false;
}

// sample_f_eval, reflectors.cuh:588-629. Types without an arm leave wo / f / pdf untouched.
template <bool COUNT, bool LEAN = false>
PT_DEV void sample_f_eval(Rng& rng, const PMat& m, const float4* tex, V3 wi, float etaI, bool backface, V3& wo, V3& f, float& pdf, float uvx, float uvy, Ctr& c) {
    V3 albedo; float trans;
    material_inputs<LEAN>(m, tex, uvx, uvy, true, albedo, trans);
    if (m.type == 0) cosine_sample_f<COUNT>(rng, albedo, wo, f, pdf, c, (!LEAN && (m.flags & kMatHasTexture)) ? nullptr : m.albedoOverPi);
    else if (m.type == 1) {                                                                       // :160-180
        V3 w = -wi;
        V3 h = ggx_sample_h<COUNT>(rng, m.roughness, c);
        wo = (2.0f * dot(w, h)) * h - w;
        if (wo.z <= 0.0f) wo.z = -wo.z;
        f = microfacet_metal_f(ld3(m.eta), ld3(m.k), m.roughness, w, wo);
        pdf = microfacet_pdf(m.roughness, w, wo);
    } else if (m.type == 2) dielectric_sample_f<COUNT>(rng, -wi, m.ior, backface, wo, f, pdf, c);
    else if (!LEAN && m.type == 4) leaf_sample_f<COUNT>(rng, -wi, m.ior, etaI, m.roughness, albedo, trans, wo, f, pdf, c);
    else if (m.type == 6) {                                                                       // :70-76
        V3 w = -wi;
        wo = v3(-w.x, -w.y, w.z);
        f = v3(1.0f / fmaxf_(wo.z, kEps));
        pdf = 1.0f;
    }
}

}  // namespace pt
