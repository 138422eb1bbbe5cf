// pt_path.h — the per-bounce "logic step" of the unidirectional integrators, shared by the
// megakernel (state in registers) and the wavefront variant (state in HBM).
//
// Restates the loop bodies of Li_unidirectional (deviceCode.cu:318-539) and
// Li_naive_unidirectional (deviceCode.cu:172-202) on a PathState. Two shadow-ray policies:
//
//   SYNC   the shadow ray of next-event estimation is traced inside the bounce, exactly where the
//          reference calls BVHShadowRay (deviceCode.cu:127).
//   DEFER  the bounce only RECORDS the shadow ray and the unoccluded NEE term; the caller traces
//          it together with the next extension ray and the term is applied by apply_pending() at
//          the start of the path's next logic step. This is value-exact, not an approximation:
//          (1) the light pdf, the BSDF value / pdf and the MIS weight of the NEE sample do not
//          depend on visibility (deviceCode.cu:133-153, 468-469 evaluate them only `if` the ray
//          is unoccluded, from geometry alone); (2) `Li` receives the same terms in the same order
//          (emission_j, nee_j, emission_j+1, ...) with the same operand grouping
//          `(beta * (nee * thr)) * w`; (3) the RNG draws of a bounce (NEE 3, BSDF 0-4, roulette 1)
//          never depend on the shadow result. The one exception are material types WITHOUT a
//          dispatch arm in reflectors.cuh:588-629 (MAT_MICROFACETDIELECTRIC, MAT_FLOWER; no row of
//          the reference's material table uses them): there `wo` keeps whatever NEE left in it,
//          so the next direction depends on visibility. Scenes whose triangles use such a
//          material are rendered with the SYNC instantiation (pt_scene_create decides).
#pragma once
#include "pt_shade.h"

namespace pt {

struct HitInfo { V3 point, normal, emission; float uvx, uvy, dist; int tri, material, lightInd; bool backface; };

// The attribute block of BVHSceneIntersect (integratorUtilities.cuh:113-141) for the final hit.
PT_DEV void resolve_hit(const DeviceScene& S, const Hit& h, V3 o, V3 d, HitInfo& hi) {
    const PAttr& at = S.attrs[h.tri];
    float bz = 1.0f - h.u - h.v;
    hi.point = v3(__builtin_fmaf(h.t, d.x, o.x), __builtin_fmaf(h.t, d.y, o.y), __builtin_fmaf(h.t, d.z, o.z));
    V3 n = normalize(ld3(at.n0) * bz + ld3(at.n1) * h.u + ld3(at.n2) * h.v);
    hi.uvx = at.uv0[0] * bz + at.uv1[0] * h.u + at.uv2[0] * h.v;
    hi.uvy = at.uv0[1] * bz + at.uv1[1] * h.u + at.uv2[1] * h.v;
    if (dot(n, d) > 0.0f) { n = -n; hi.backface = true; } else hi.backface = false;
    hi.normal = n;
    hi.material = at.material;
    hi.emission = ld3(at.emission);
    hi.lightInd = at.lightInd;
    hi.tri = h.tri;
    hi.dist = h.t;
}

enum : uint32_t {
    kInPath = 1u,            // a path is alive; (o, d) is its next extension ray
    kHitFirstNonSpec = 2u,   // hitFirstnonSpecular, deviceCode.cu:316
    kShadowPending = 4u,     // DEFER: (so, sd, smaxt) must be traced and applied
    kNeeValid = 8u,          // DEFER: the recorded NEE sample had light_pdf > EPSILON
    kFinishPending = 16u,    // DEFER: the previous path ended with its last NEE term still pending: `Li` still holds ITS sum
};

struct PathState {
    Rng rng;
    V3 o, d;                 // extension ray
    V3 beta, Li, prevPoint, woLocal;
    float pdf, etaI, etaT;
    int depth, guard, msTop;
    uint32_t flags;
    // DEFER only. (so, sd, smaxt) are read once, right after the bounce that wrote them, to start the shadow ray: they are not
    // live across a traversal. kFinishPending: the finished path's sum stays in `Li` until apply_pending has added its last NEE
    // term — the path that follows starts at Li = 0 and receives nothing before that apply_pending, so one register triple
    // serves both (three VGPRs less across every traversal; the kernel for scenes in HBM lives at 64).
    V3 so, sd; float smaxt;  // pending shadow ray
    V3 neeRaw, neeBeta; float neeW;      // PRE (scenes without MAT_LEAF triangles): neeRaw holds the finished term (beta * nee) * w, the other two are unused
};

// mediumStack[16] (deviceCode.cu:306) — in LDS for the megakernel, packed in 4 registers for the
// wavefront variant. Entry 0 is always air (material 0).
struct LdsMedium {
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    lds_u8* p;
    PT_DEV int get(int i) const { return p[i * 64]; }
    PT_DEV void set(int i, int v) { p[i * 64] = (uint8_t)v; }
};
struct NoMedium {                    // SIMPLE kernels: the stack would only ever hold air (material 0)
    PT_DEV int get(int) const { return 0; }
    PT_DEV void set(int, int) {}
};
struct RegMedium {
    uint32_t w0, w1, w2, w3;
    PT_DEV int get(int i) const {
        uint32_t w = (i < 8) ? ((i < 4) ? w0 : w1) : ((i < 12) ? w2 : w3);
        return (int)((w >> ((i & 3) * 8)) & 0xffu);
    }
    PT_DEV void set(int i, int v) {
        uint32_t sh = (uint32_t)(i & 3) * 8u, m = ~(0xffu << sh), b = ((uint32_t)v & 0xffu) << sh;
        if (i < 4) w0 = (w0 & m) | b; else if (i < 8) w1 = (w1 & m) | b; else if (i < 12) w2 = (w2 & m) | b; else w3 = (w3 & m) | b;
    }
};

// removeMaterialFromStack, integratorUtilities.cuh:414-434 (entry 0 is never removed)
template <class MS>
PT_DEV void medium_remove(MS& ms, int& top, int materialID) {
    int found = -1;
    for (int i = top - 1; i > 0; i--) if (ms.get(i) == materialID) { found = i; break; }
    if (found != -1) {
        for (int i = found; i < top - 1; i++) ms.set(i, ms.get(i + 1));
        top--;
    }
}

// Start the next sample of a pixel (the top of the reference kernels, deviceCode.cu:294-316).
template <bool COUNT, class MS>
PT_DEV void path_begin(const CamK& cam, PathState& ps, MS& ms, int x, int y, Ctr& c) {
    camera_ray<COUNT>(cam, ps.rng, x, y, ps.o, ps.d, c);
    const bool fin = (ps.flags & kFinishPending) != 0;       // Li still holds the finished path's sum (see PathState); this path's own Li is 0 until then
    ps.beta = v3(1.0f); ps.Li = v3(fin ? ps.Li.x : 0.0f, fin ? ps.Li.y : 0.0f, fin ? ps.Li.z : 0.0f); ps.prevPoint = v3(0.0f); ps.woLocal = v3(0.0f);
    ps.pdf = kEps; ps.etaI = kEps; ps.etaT = kEps;
    ps.depth = 0; ps.guard = 0; ps.msTop = 1; ms.set(0, 0);
    ps.flags = (ps.flags & ~kHitFirstNonSpec) | kInPath;
}

// DEFER: add the NEE term recorded by the previous bounce (and close a path that ended on it): `Li += (beta * (nee * thr)) * w`
// exactly as deviceCode.cu:153 groups it, then — if that path had ended — `colors[pixelIdx] += Li` (:540).
// PRE: no triangle of the scene is a MAT_LEAF, so a shadow ray's throughput is exactly 0 or exactly 1 (BVHShadowRay,
// integratorUtilities.cuh:188-288); nee * 1.0f == nee bit for bit, and the bounce has recorded the finished term.
template <bool PRE = false>
PT_DEV void apply_pending(PathState& ps, V3 thr, V3& acc) {
    // (value selects only: a branch that picks WHICH field to update becomes a pointer select and
    // forces the whole PathState into scratch memory)
    const bool fin = (ps.flags & kFinishPending) != 0;
    const bool add = (ps.flags & kShadowPending) && (ps.flags & kNeeValid) && dot(thr, thr) > 0.0f;
    V3 term = ps.neeRaw;
    if (!PRE) term = (ps.neeBeta * (ps.neeRaw * thr)) * ps.neeW;
    const V3 sum = ps.Li + term;
    const V3 li = v3(add ? sum.x : ps.Li.x, add ? sum.y : ps.Li.y, add ? sum.z : ps.Li.z);
    const V3 done = acc + li;
    acc = v3(fin ? done.x : acc.x, fin ? done.y : acc.y, fin ? done.z : acc.z);
    ps.Li = v3(fin ? 0.0f : li.x, fin ? 0.0f : li.y, fin ? 0.0f : li.z);
    ps.flags &= ~(kShadowPending | kNeeValid | kFinishPending);
}

// The loop-top tests of the reference for a live path: `depth < 100` (MIS, deviceCode.cu:318) or
// `depth < maxDepth` (naive, :172), plus the iteration guard of DESIGN.md §4. True = path is over.
template <int INTEG>
PT_DEV bool path_exhausted(PathState& ps, int maxDepth) {
    if (ps.depth >= ((INTEG == 2) ? maxDepth : 100)) return true;
    if (INTEG != 2 && ++ps.guard > 4096) return true;
    return false;
}

// One loop body on the closest hit `h` of the extension ray (o, d). Returns true when the
// path ended (miss, zero pdf, roulette). On false, (o, d) is the next extension ray.
// shadow(ro, wi, maxt) -> throughput is only called when DEFER is false.
// (The body works on SEPARATE local scalars, not on PathState fields: stores to two fields of one
// struct in sibling branches get merged by the optimiser into one store through a computed field
// offset, which keeps the whole struct in scratch memory.)
struct NeeRecord { V3 so, sd; float smaxt; V3 neeRaw, neeBeta; float neeW; };

// SIMPLE (chosen per scene by the host, pt_api.hip: scene_simple): every triangle's material is an untextured MAT_DIFFUSE
// that is neither a boundary nor specular, and material 0 (the air every path starts in) does not absorb — the reference's
// own Cornell configuration. Then the medium stack only ever holds air, every hit is a true hit, the three dispatchers
// have one arm each, and all of that is known at compile time: same values, a third of the code.
// LEAN: see pt_shade.h (no MAT_LEAF triangle, no textures). PRE: the DEFER record holds the finished NEE term (apply_pending<PRE>)
// — valid when no triangle is a MAT_LEAF: SIMPLE and LEAN scenes, and every scene the pair form of FLAT accepts.
template <int INTEG, bool COUNT, bool DEFER, bool SIMPLE, bool LEAN, bool PRE, class MS, class ShadowFn>
PT_DEV bool bounce_core(const DeviceScene& S, Rng& rng, V3& o, V3& d, V3& beta, V3& Li, V3& prevPoint, V3& woLocal,
                        float& pdf, float& etaI, float& etaT, int& depth, int& msTop, uint32_t& flags, NeeRecord& nr,
                        MS& ms, const Hit& h, int maxDepth, int useMIS, ShadowFn shadow, Ctr& c) {
    if (COUNT) c.iters++;
    if (h.tri < 0) {
        Li = Li + beta * v3(0.0f);          // `Li += beta * sampleSky()`; the sky is black (integratorUtilities.cuh:436-438)
        return true;
    }
    HitInfo hi;
    resolve_hit(S, h, o, d, hi);
    const PMat& m = S.mats[hi.material];
    if (INTEG == 2) {
        // ---- Li_naive_unidirectional, deviceCode.cu:183-201 ----
        const Onb frame = onb_of(hi.normal);
        V3 toSurface = to_local(d, frame);
        V3 f = v3(0.0f), toNext = v3(0.0f);
        float p = 0.0f;
        if (SIMPLE) cosine_sample_f<COUNT>(rng, ld3(m.albedo), toNext, f, p, c, m.albedoOverPi);
        else sample_f_eval<COUNT, LEAN>(rng, m, S.textures, toSurface, 1.0f, hi.backface, toNext, f, p, hi.uvx, hi.uvy, c);
        if (p <= 0.0f || dot(f, f) < kEps) return true;
        Li = Li + hi.emission * beta;
        beta = beta * ((f * __builtin_fabsf(toNext.z)) / p);
        V3 nw = to_world(toNext, frame);
        o = hi.point + ((toNext.z > 0.0f) ? (hi.normal * kRayEps) : ((-hi.normal) * kRayEps));
        d = nw;
        depth++;
        return false;
    }
    // ---- Li_unidirectional, deviceCode.cu:332-537 ----
    const Onb frame = onb_of(hi.normal);                 // toLocal / toWorld of this bounce all use the hit's normal
    V3 wiLocal = to_local(d, frame);
    bool isSpecular = false, trueHit = true;
    if (!SIMPLE) {
        isSpecular = (m.flags & kMatSpecular) != 0;
        int minPriorID = ms.get(0);
        int minPrior = S.mats[minPriorID].priority;
        for (int i = 1; i < msTop; i++) {
            int id = ms.get(i);
            int pr = S.mats[id].priority;
            if (pr < minPrior) { minPrior = pr; minPriorID = id; }
        }
        const PMat& dom = S.mats[minPriorID];
        // Beer-Lambert, deviceCode.cu:362-368. A medium that does not absorb (air, plain glass: absorption exactly 0) gives
        // exp(-+0) = 1.0f exactly in this arithmetic and beta * 1.0f = beta bit for bit, so the three exps are skipped.
        if (hi.dist > kEps && !(dom.absorption[0] == 0.0f && dom.absorption[1] == 0.0f && dom.absorption[2] == 0.0f)) {
            V3 att = v3(exp_(-dom.absorption[0] * hi.dist), exp_(-dom.absorption[1] * hi.dist), exp_(-dom.absorption[2] * hi.dist));
            beta = beta * att;
        }
        if (m.flags & kMatBoundary) {
            if (m.priority <= minPrior) {
                if (m.type == 2) {
                    etaI = dom.ior;
                    if (!hi.backface) etaT = m.ior;
                    else if (msTop == 1) etaT = 1.0f;
                    else {
                        int mp = 99, second = ms.get(0);
                        for (int i = 0; i < msTop; i++) {
                            int id = ms.get(i);
                            int pr = S.mats[id].priority;
                            if (pr) { if (mp > pr && id != hi.material) { second = id; mp = pr; } }
                        }
                        etaT = S.mats[second].ior;
                    }
                }
            } else {
                trueHit = false;
                if (!hi.backface) { if (msTop < 16) { ms.set(msTop, hi.material); msTop++; } }
                else medium_remove(ms, msTop, hi.material);
            }
        } else etaI = dom.ior;
    }
    bool done = false;
    if (trueHit) {
        float le2 = dot(hi.emission, hi.emission);
        if (le2 > kEps) {
            if (depth == 0 || !(flags & kHitFirstNonSpec)) Li = Li + beta * hi.emission;
            else if (useMIS && !isSpecular) {
                // neePDF, deviceCode.cu:63-85: the hit triangle as a light
                float lightPdf = kEps;
                if (hi.lightInd >= 0) {
                    const PLight& L = S.lights[hi.lightInd];
                    V3 s2l = hi.point - prevPoint;
                    V3 wi = normalize(s2l);
                    float dist2 = dot(s2l, s2l);
                    float cosL = dot(ld3(L.na), -wi);
                    lightPdf = dist2 / (cosL * (float)S.nLights * L.area);
                }
                if (lightPdf > kEps) {
                    float wB = pdf * pdf / (lightPdf * lightPdf + pdf * pdf);
                    Li = Li + (beta * hi.emission) * wB;
                }
            }
        }
        if (useMIS && le2 < kEps && !isSpecular && S.nLights > 0) {
            // nextEventEstimation, deviceCode.cu:87-156 (with nLights == 0 it draws nothing and adds nothing)
            int index = min((int)(draw<COUNT>(rng, c) * (float)S.nLights), S.nLights - 1);
            const PLight& L = S.lights[index];
            V3 A = ld3(L.a), B = ld3(L.b), Cc = ld3(L.c);
            float u = __builtin_sqrtf(draw<COUNT>(rng, c));
            float v = draw<COUNT>(rng, c);
            V3 p = (1.0f - u) * A + (u * (1.0f - v)) * B + (u * v) * Cc;
            V3 s2l = p - hi.point;
            V3 wi = normalize(s2l);
            V3 ro = hi.point + wi * kEps;
            // t to the light triangle; if that test fails the reference leaves t uninitialised
            // (:121-123) — defined as |s2l| - EPSILON
            float t = length(s2l) - kEps;
            {
                float tt, uu, vv;
                if ((PT_RCP_UNIFORM ? moller_trumbore_sel : moller_trumbore)(A, B - A, Cc - A, ro, wi, tt, uu, vv)) t = tt;     // (branch-free in the issue-bound LDS-resident kernels)
            }
            const float maxt = t * (1.0f - kEps);
            V3 thr = v3(1.0f);
            if (!DEFER) thr = shadow(ro, wi, maxt);
            if (DEFER || dot(thr, thr) > 0.0f) {
                // everything below is independent of visibility (deviceCode.cu:133-153)
                float dist2 = dot(s2l, s2l);
                float cosL = dot(ld3(L.na), -wi);
                float cosS = __builtin_fabsf(dot(hi.normal, wi));
                float lightPdf = dist2 / (cosL * (float)S.nLights * L.area);
                V3 wiL = to_local(wi, frame);
                if (!DEFER) woLocal = wiL;
                V3 f = SIMPLE ? ld3(m.albedoOverPi) : f_eval<LEAN>(m, S.textures, wiLocal, wiL, etaI, hi.uvx, hi.uvy);
                V3 nee = ((f * ld3(L.emission)) * cosS) / lightPdf;
                if (lightPdf > kEps) {
                    float pdfB = pdf;
                    if (SIMPLE) pdfB = cosine_pdf(wiL);
                    else pdf_eval<LEAN>(m, S.textures, wiLocal, wiL, etaI, hi.uvx, hi.uvy, pdfB);
                    float wN = lightPdf * lightPdf / (pdfB * pdfB + lightPdf * lightPdf);
                    if (DEFER) {
                        if (PRE) nr.neeRaw = (beta * nee) * wN;               // apply_pending<PRE>: the finished term (thr is exactly 1 when it is added)
                        else { nr.neeRaw = nee; nr.neeBeta = beta; nr.neeW = wN; }
                        flags |= kNeeValid;
                    } else {
                        pdf = pdfB;
                        nee = nee * thr;
                        Li = Li + (beta * nee) * wN;
                    }
                }
                if (DEFER) { nr.so = ro; nr.sd = wi; nr.smaxt = maxt; flags |= kShadowPending; }
            }
        }
        V3 f = v3(0.0f);
        if (SIMPLE) cosine_sample_f<COUNT>(rng, ld3(m.albedo), woLocal, f, pdf, c, m.albedoOverPi);
        else sample_f_eval<COUNT, LEAN>(rng, m, S.textures, wiLocal, etaI, hi.backface, woLocal, f, pdf, hi.uvx, hi.uvy, c);
        V3 woWorld = to_world(woLocal, frame);
        pdf = fmaxf_(pdf, 0.01f);
        if (!SIMPLE && woLocal.z < 0.0f) {                 // (a cosine-sampled direction has z = sqrt(1 - u1) > 0: never taken when SIMPLE)
            if (!hi.backface) { if (msTop < 16) { ms.set(msTop, hi.material); msTop++; } }
            else medium_remove(ms, msTop, hi.material);
        }
        beta = beta * ((f * __builtin_fabsf(woLocal.z)) / pdf);
        if (woLocal.z > 0.0f) o = hi.point + hi.normal * kEps;
        else o = hi.point - hi.normal * kEps;
        d = normalize(woWorld);
        prevPoint = hi.point;
    } else {
        woLocal = wiLocal;                               // toLocal(ray.direction, normal) again (deviceCode.cu:516): the value computed above
        o = hi.point + d * kRayEps;
        depth--;
    }
    if (depth > maxDepth) {
        float lum = dot(beta, v3(0.2126f, 0.7152f, 0.0722f));
        float p = clampf(lum, 0.05f, 0.99f);
        if (draw<COUNT>(rng, c) > p) done = true;
        else beta = beta / p;
    }
    if (!done) {
        if (!isSpecular) flags |= kHitFirstNonSpec;
        depth++;
    }
    return done;
}


template <int INTEG, bool COUNT, bool DEFER, bool SIMPLE = false, bool LEAN = false, bool PRE = SIMPLE, class MS, class ShadowFn>
PT_DEV bool path_bounce(const DeviceScene& S, PathState& ps, MS& ms, const Hit& h, int maxDepth, int useMIS, ShadowFn shadow, Ctr& c) {
    Rng rng = ps.rng;
    V3 o = ps.o, d = ps.d, beta = ps.beta, Li = ps.Li, prevPoint = ps.prevPoint, woLocal = ps.woLocal;
    float pdf = ps.pdf, etaI = ps.etaI, etaT = ps.etaT;
    int depth = ps.depth, msTop = ps.msTop;
    uint32_t flags = ps.flags;
    // (a fresh record, not a copy of the previous bounce's: whatever the last bounce recorded was consumed — the shadow ray
    // started, the term applied — before this bounce runs, so the old values are dead here and not live across the traversal)
    NeeRecord nr;
#ifndef PT_NR_FRESH
#define PT_NR_FRESH 1
#endif
#if PT_NR_FRESH
    nr.so = v3(0.0f); nr.sd = v3(0.0f); nr.smaxt = 0.0f; nr.neeRaw = v3(0.0f); nr.neeBeta = v3(0.0f); nr.neeW = 0.0f;
#else
    nr.so = ps.so; nr.sd = ps.sd; nr.smaxt = ps.smaxt; nr.neeRaw = ps.neeRaw; nr.neeBeta = ps.neeBeta; nr.neeW = ps.neeW;
#endif
    bool done = bounce_core<INTEG, COUNT, DEFER, SIMPLE, LEAN, PRE>(S, rng, o, d, beta, Li, prevPoint, woLocal, pdf, etaI, etaT, depth, msTop, flags, nr,
                                                 ms, h, maxDepth, useMIS, shadow, c);
    ps.rng = rng;
    ps.o = o; ps.d = d; ps.beta = beta; ps.Li = Li; ps.prevPoint = prevPoint; ps.woLocal = woLocal;
    ps.pdf = pdf; ps.etaI = etaI; ps.etaT = etaT; ps.depth = depth; ps.msTop = msTop; ps.flags = flags;
    ps.so = nr.so; ps.sd = nr.sd; ps.smaxt = nr.smaxt; ps.neeRaw = nr.neeRaw; ps.neeBeta = nr.neeBeta; ps.neeW = nr.neeW;
    return done;
}

}  // namespace pt
