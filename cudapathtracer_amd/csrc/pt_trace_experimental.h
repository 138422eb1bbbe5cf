// pt_trace_experimental.h — traversals that were built, proven bit-identical and MEASURED SLOWER than the production ones
// (profiles/r01_ab_*, r02_ab_*): the DEFER pair walk of the general megakernel (option "defer_shadow"),
// speculative descent ("spec"), compact 32-byte nodes ("compact") and the 4-wide collapsed tree ("wide"). They are kept
// for the A/B only: `make EXPERIMENTAL=1` compiles them (-DPT_EXPERIMENTAL) and the kernels that use them; the default
// libptamd.so contains none of this and pt_set_option refuses the options that would select them.
#pragma once
#include "pt_trace.h"

namespace pt {

// Shadow ray (optional) then extension ray (optional) of one lane in a single loop. Both
// reciprocal directions are computed up front (convergent code); a lane that finishes its shadow
// ray switches to its extension ray with a handful of register moves and stays in the loop, so the
// wave re-converges once per PAIR of rays.
template <bool COUNT, int N>
PT_DEV void trace_pair(const DeviceScene& S, const SceneCache& C, Stack<N>& st, bool hasShadow, V3 so, V3 sd, float smaxt,
                       bool hasExt, V3 eo, V3 ed, V3& thr, Hit& h, Ctr& c) {
    h.tri = -1; h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.material = 0;
    thr = v3(1.0f);
    if (!hasShadow && !hasExt) return;
    const V3 invS = inv3(sd);
    const V3 invE = inv3(ed);
    if (COUNT) { if (hasShadow) c.raysShadow++; if (hasExt) c.raysClosest++; }
    bool isShadow = hasShadow;
    V3 o = isShadow ? so : eo, d = isShadow ? sd : ed, inv = isShadow ? invS : invE;
    float max_t = isShadow ? smaxt : 999999.0f;
    float min_t = 3.402823466e+38f;
    int32_t cur = S.rootRef;
    st.sp = 0;
    while (true) {
        while (cur >= 0) cur = descend<COUNT, N>(S, C, cur, o, inv, st, c);
        if (cur == kRefNone) {
            if (isShadow && hasExt) {                 // shadow ray done: start this lane's extension ray
                isShadow = false;
                o = eo; d = ed; inv = invE; max_t = 999999.0f;
                cur = S.rootRef; st.sp = 0;
                continue;
            }
            break;
        }
        if (COUNT) c.pops++;
        int32_t ti = ~cur;
        uint32_t idx;
        bool occluded = false;
        do {
            TriData q = load_tri(S, C, ti);
            idx = f2u(q.e.y);
            if (COUNT) c.tris++;
            float t, u, v;
            bool ok = moller_trumbore(v3(q.a.x, q.a.y, q.a.z), v3(q.a.w, q.b.x, q.b.y), v3(q.b.z, q.b.w, q.e.x), o, d, t, u, v);
            if (isShadow) {
                if (ok && (t < max_t)) {
                    uint32_t flags = f2u(q.e.w);
                    if (!(flags & 1u)) { thr = v3(0.0f); occluded = true; break; }
                    // MAT_LEAF (integratorUtilities.cuh:218-239)
                    const PMat& m = S.mats[f2i(q.e.z)];
                    const PAttr& at = S.attrs[idx & 0x7fffffffu];
                    float bz = 1.0f - u - v;
                    V3 n = ld3(at.n0) * bz + ld3(at.n1) * u + ld3(at.n2) * v;
                    float cosTheta = __builtin_fabsf(dot(d, normalize(n)));
                    float F = schlick_fresnel(cosTheta, 1.0f, m.ior);
                    V3 sc = ld3(m.albedo) * m.transmission * (1.0f - F);
                    thr = thr * sc;
                    if (fmaxf_(thr.x, fmaxf_(thr.y, thr.z)) < 0.01f) { thr = v3(0.0f); occluded = true; break; }
                }
            } else if (ok && (t < min_t) && (t < max_t)) {
                min_t = t;
                h.t = t; h.u = u; h.v = v;
                h.tri = (int32_t)(idx & 0x7fffffffu);
                h.material = f2i(q.e.z);
            }
            ti++;
        } while (!(idx & 0x80000000u));
        cur = (!occluded && st.sp > 0) ? st.pop() : kRefNone;     // an occluded shadow ray ends here (BVHShadowRay returns)
    }
    if (COUNT) { if (hasExt && h.tri >= 0) c.hits++; }
}

// ---- speculative descent (scenes in HBM) ---------------------------------------------------------
// Without culling the set of leaves a ray visits does not depend on what it hits, so a lane that reaches a leaf need not
// test it at once: it POSTPONES the leaf (one slot, FIFO) and keeps descending with the wave; the wave switches to the
// triangle loop when too few lanes can still descend, and there every lane that holds a postponed leaf tests it. Leaves are
// tested in the order they were reached, so the strict `t < min_t` tie rule sees the reference's order; per ray the visits,
// tests and counters are those of trace_resume. What changes is how many lanes a trip through the node loop carries: on
// the 263 k-triangle scene a ray takes ~6 node steps between two leaves, and a lane that waits at its leaf for the others
// is idle for those trips (26 of 64 lanes per trip with trace_resume).
// MEASURED (profiles/r02_ab_spec.log): 263 k triangles 861.3 -> 861.4 ms, 82 k triangles 254.7 -> 251.6 ms (+1.2 %) — the
// trips saved are paid for by the ballot and the postponement logic in every trip, and this kernel is not bound by the
// lanes of its node loop alone. Not the default: build with -DPT_SPEC=1 to reproduce.
// A shadow ray that the postponed leaf will occlude descends a few nodes in vain; `specShadow` = false keeps shadow rays
// strictly in the reference's order (the counting kernels need that: their node counters are part of the parity contract).
constexpr int32_t kRefHold = (int32_t)0x80000001;      // "the next ref is popped after the pending leaf" (no speculation for this ray)

template <bool COUNT, int N, bool ONCHIP, bool NOLEAF = false>
PT_DEV void trace_resume_spec(const DeviceScene& S, const SceneCache& C, Stack<N>& st, RayState& r, V3 eo, V3 ed, int minBusy,
                              V3& thr, Hit& h, Ctr& c, Keep k, bool specShadow) {
    typedef LoopExit<ONCHIP> X;
    if (!(r.flags & kRayBusy)) return;
    V3 o = r.o, d = r.d, inv = r.inv;
    float max_t = r.max_t, min_t = r.min_t;
    int32_t cur = r.cur, pend = r.pend;
    bool isShadow = (r.flags & kRayShadow) != 0, extFollows = (r.flags & kRayExtFollows) != 0, busy = true;
    uint32_t occlBits = r.flags & kRayOccluded;        // NOLEAF scenes: the caller reads the shadow ray's result from this bit (pt_trace.h: RayState)
    while (true) {
        const int active = lanes_here();
        if (active <= minBusy) break;
        // ---- node phase: descend while enough lanes can ----
        const int keepN = (active * k.node) >> 4;
        while (true) {
            if (pend == kRefNone && cur < 0 && cur != kRefNone && cur != kRefHold) {       // reached a leaf: postpone it
                pend = cur;
                const bool spec = !isShadow || (specShadow && !COUNT);
                cur = spec ? (st.sp > 0 ? st.template pop<ONCHIP>() : kRefNone) : kRefHold;
            }
            const bool can = cur >= 0;
            const int nCan = __builtin_popcountll(__builtin_amdgcn_ballot_w64(can));
            if (nCan == 0 || (X::node && nCan <= keepN)) break;
            if (can) {
                PT_UTIL_STEP(c, 0);
                cur = descend<COUNT, N, ONCHIP>(S, C, cur, o, inv, st, c);
            }
        }
        // ---- triangle phase: the postponed leaves ----
        if (pend != kRefNone) {
            int32_t ti = ~pend;
            uint32_t idx;
            bool occluded = false, more;
            int keepT = 0;
            if (X::tri) keepT = (lanes_here() * k.tri) >> 4;
            do {
                TriData q = load_tri<ONCHIP>(S, C, ti);
                idx = f2u(q.e.y);
                if (COUNT) c.tris++;
                PT_UTIL_STEP(c, 2);
                float t, u, v;
                bool ok = moller_trumbore(v3(q.a.x, q.a.y, q.a.z), v3(q.a.w, q.b.x, q.b.y), v3(q.b.z, q.b.w, q.e.x), o, d, t, u, v);
                if (isShadow) {
                    if (ok && (t < max_t)) {
                        uint32_t flags = f2u(q.e.w);
                        if (NOLEAF || !(flags & 1u)) { thr = v3(0.0f); if (NOLEAF) occlBits = kRayOccluded; occluded = true; break; }
                        // MAT_LEAF (integratorUtilities.cuh:218-239)
                        const PMat& m = S.mats[f2i(q.e.z)];
                        const PAttr& at = S.attrs[idx & 0x7fffffffu];
                        float bz = 1.0f - u - v;
                        V3 n = ld3(at.n0) * bz + ld3(at.n1) * u + ld3(at.n2) * v;
                        float cosTheta = __builtin_fabsf(dot(d, normalize(n)));
                        float F = schlick_fresnel(cosTheta, 1.0f, m.ior);
                        V3 sc = ld3(m.albedo) * m.transmission * (1.0f - F);
                        thr = thr * sc;
                        if (fmaxf_(thr.x, fmaxf_(thr.y, thr.z)) < 0.01f) { thr = v3(0.0f); occluded = true; break; }
                    }
                } else if (ok && (t < min_t) && (t < max_t)) {
                    min_t = t;
                    h.t = t; h.u = u; h.v = v;
                    h.tri = (int32_t)(idx & 0x7fffffffu);
                    h.material = f2i(q.e.z);
                }
                ti++;
                more = !(idx & 0x80000000u);
                if (X::tri && more && lanes_here() <= keepT) break;
            } while (more);
            if (X::tri && !occluded && more) pend = ~ti;                               // the rest of this leaf next time round
            else {
                if (COUNT) c.pops++;
                pend = kRefNone;
                if (occluded) { cur = kRefNone; st.sp = 0; }                             // BVHShadowRay returns at the first opaque hit
                else if (cur == kRefHold) cur = st.sp > 0 ? st.template pop<ONCHIP>() : kRefNone;
            }
        }
        if (cur == kRefNone && pend == kRefNone) {                                      // this ray is through
            if (isShadow && extFollows) {                                               // start the lane's extension ray
                isShadow = false; extFollows = false;
                o = eo; d = ed; inv = inv3(d); max_t = 999999.0f;
                cur = S.rootRef; st.sp = 0;
                continue;
            }
            busy = false;
            break;
        }
    }
    r.o = o; r.d = d; r.inv = inv; r.max_t = max_t; r.min_t = min_t; r.cur = cur; r.pend = pend;
    r.flags = (busy ? kRayBusy : 0u) | (isShadow ? kRayShadow : 0u) | (extFollows ? kRayExtFollows : 0u) | occlBits;
    if (COUNT) { if (!busy && !isShadow && h.tri >= 0) c.hits++; }
}

// ---- compact nodes (SIMPLE scenes in HBM) --------------------------------------------------------
// visited(leaf) == slab(leaf's own box) for a ray with a regular 1 / d (see inv_is_regular): the boxes ABOVE the leaves only
// have to contain them. So the inner nodes can be stored smaller than the reference stores them — QNode: 8-bit offsets in
// the node's own frame, rounded outward, 32 bytes, two 16-byte loads per visit instead of four, twice as many nodes per
// cache line and in the LDS copy of the top of the tree — as long as a leaf is entered only if its EXACT box passes (leafBox, 2 loads per
// candidate leaf) and the result is the reference's: minimum t, and where two triangles return the same t the one the
// reference visits first — decided, when it happens, by walking the reference's own nodes (exact boxes, PNode.pad0 = first
// triangle of the right subtree) down to where the two leaves part and asking which child the reference enters first
// (`tL < tR`, else the right one). Lanes whose ray has a zero direction component (`exact`) fetch the reference's PNodes
// instead, in the same loop. Shadow rays of a NOLEAF scene are occluded by any hit.
// MEASURED (profiles/r02_ab_compact.log): bit-identical frames, and SLOWER — 82 k triangles 212 -> 259 ms (32 spp), 263 k
// triangles 199 -> 239 ms (8 spp), with the 16-bit global grid of the first version as with these per-node frames. Load
// instructions -23 %, the TA's busy cycles -11 % (a lane's first 16 bytes of a line cost it ~1.7 cycles, each further 16
// bytes ~0.76: half the bytes are not half the cost), VALU instructions +58 % (the node loop is this kernel's instruction
// stream, and the decode adds ~40 to its ~65 per trip): DESIGN.md §6. Loading a leaf's first triangle together with its box
// (one latency step instead of two) is another 3 % slower. Not the default: option "compact" = 1 to reproduce.
struct Compact { const QNode* q; const f4v* leafBox; const int32_t* mids; };

PT_DEV bool tie_keeps_first(const DeviceScene& S, const int32_t* __restrict__ mids, V3 o, V3 inv, int tiBest, int tiNew) {
    // both triangles returned the same t: true if the reference visits tiBest's before tiNew's (packed indices, leaf order)
    int32_t ref = S.rootRef;
    while (ref >= 0) {
        const f4v* p = reinterpret_cast<const f4v*>(S.nodes + ref);
        NodeData n; n.a = p[0]; n.b = p[1]; n.c = p[2]; n.d = p[3];
        const int mid = mids[ref];
        const bool bestLeft = tiBest < mid, newLeft = tiNew < mid;
        if (bestLeft != newLeft) {
            float tL, tR;
            slab(n.a.x, n.a.y, n.a.z, n.a.w, n.b.x, n.b.y, o, inv, tL);
            slab(n.b.z, n.b.w, n.c.x, n.c.y, n.c.z, n.c.w, o, inv, tR);
            return (tL < tR) == bestLeft;                          // the reference enters the left child first iff tL < tR
        }
        ref = bestLeft ? f2i(n.d.x) : f2i(n.d.y);
    }
    return tiBest < tiNew;                                         // same leaf: ascending order inside it
}

template <int N>
PT_DEV void trace_resume_q(const DeviceScene& S, const SceneCache& C, const Compact& K, Stack<N>& st, RayState& r, V3 eo, V3 ed, int minBusy, V3& thr, Hit& h, Keep k) {
    if (!(r.flags & kRayBusy)) return;
    const QNode* __restrict__ Q = K.q;
    const f4v* __restrict__ leafBox = K.leafBox;
    int bestTi = r.pend;                                           // the packed index of the best hit so far (for the tie rule)
    bool inLeaf = (r.flags & kRayInLeaf) != 0;                     // `cur` resumes a leaf whose own box has been tested
    V3 o = r.o, d = r.d, inv = r.inv;
    float max_t = r.max_t, min_t = r.min_t;
    int32_t cur = r.cur;
    bool isShadow = (r.flags & kRayShadow) != 0, extFollows = (r.flags & kRayExtFollows) != 0, busy = true;
    uint32_t occlBits = r.flags & kRayOccluded;        // NOLEAF scenes: the caller reads the shadow ray's result from this bit (pt_trace.h: RayState)
    bool exact = !inv_is_regular(inv);
    const int nLdsQ = C.nNodes << 1;                               // the LDS scene cache holds the first nLdsQ compact nodes (2 per 64 bytes)
    while (true) {
        const int active = lanes_here();
        if (active <= minBusy) break;
        const int keepN = (active * k.node) >> 4;
        while (cur >= 0) {
            float b[12];                                           // left min xyz, left max xyz, right min xyz, right max xyz
            int32_t left, right;
            if (exact) {
                const f4v* p = reinterpret_cast<const f4v*>(S.nodes + cur);
                const f4v a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
                b[0] = a0.x; b[1] = a0.y; b[2] = a0.z; b[3] = a0.w; b[4] = a1.x; b[5] = a1.y;
                b[6] = a1.z; b[7] = a1.w; b[8] = a2.x; b[9] = a2.y; b[10] = a2.z; b[11] = a2.w;
                left = f2i(a3.x); right = f2i(a3.y);
            } else {
                f4v q0, q1;
                if (cur < nLdsQ) { lds_cf4* p = C.nodes + cur * 2; q0 = p[0]; q1 = p[1]; }
                else { const f4v* p = reinterpret_cast<const f4v*>(Q + cur); q0 = p[0]; q1 = p[1]; }
                const uint32_t w0 = f2u(q0.w), w1 = f2u(q1.x), w2 = f2u(q1.y), wl = f2u(q1.z), wr = f2u(q1.w);
                const float sc = __builtin_bit_cast(float, (((wl >> 24) & 0x7fu) + 63u) << 23);             // 2^k, k = field - 64
                b[0] = __builtin_fmaf((float)(w0 & 0xffu), sc, q0.x); b[1] = __builtin_fmaf((float)((w0 >> 8) & 0xffu), sc, q0.y); b[2] = __builtin_fmaf((float)((w0 >> 16) & 0xffu), sc, q0.z);
                b[3] = __builtin_fmaf((float)(w0 >> 24), sc, q0.x); b[4] = __builtin_fmaf((float)(w1 & 0xffu), sc, q0.y); b[5] = __builtin_fmaf((float)((w1 >> 8) & 0xffu), sc, q0.z);
                b[6] = __builtin_fmaf((float)((w1 >> 16) & 0xffu), sc, q0.x); b[7] = __builtin_fmaf((float)(w1 >> 24), sc, q0.y); b[8] = __builtin_fmaf((float)(w2 & 0xffu), sc, q0.z);
                b[9] = __builtin_fmaf((float)((w2 >> 8) & 0xffu), sc, q0.x); b[10] = __builtin_fmaf((float)((w2 >> 16) & 0xffu), sc, q0.y); b[11] = __builtin_fmaf((float)(w2 >> 24), sc, q0.z);
                const int32_t vl = (int32_t)(wl & 0xffffffu), vr = (int32_t)(wr & 0xffffffu);
                left = (int32_t)wl < 0 ? ~vl : vl; right = (int32_t)wr < 0 ? ~vr : vr;
            }
            float tL, tR;
            const bool hL = slab(b[0], b[1], b[2], b[3], b[4], b[5], o, inv, tL);
            const bool hR = slab(b[6], b[7], b[8], b[9], b[10], b[11], o, inv, tR);
            if (hL && hR) {
                const bool leftNear = tL < tR;
                st.push(leftNear ? right : left);
                cur = leftNear ? left : right;
            } else if (hL) cur = left;
            else if (hR) cur = right;
            else cur = st.sp > 0 ? st.pop() : kRefNone;
            if (lanes_here() <= keepN) break;
        }
        if (cur >= 0) continue;
        if (cur == kRefNone) {
            if (isShadow && extFollows) {                           // shadow ray done: start this lane's extension ray
                isShadow = false; extFollows = false;
                o = eo; d = ed; inv = inv3(d); max_t = 999999.0f;
                exact = !inv_is_regular(inv);
                cur = S.rootRef; st.sp = 0;
                continue;
            }
            busy = false;
            break;
        }
        int32_t ti = ~cur;
        bool enter = true;
        if (!exact && !inLeaf) {                                    // the leaf's own box, exactly
            const f4v l0 = leafBox[2 * ti], l1 = leafBox[2 * ti + 1];
            float t0;
            enter = slab(l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, o, inv, t0);
        }
        inLeaf = false;
        uint32_t idx;
        bool occluded = false, more = false;
        if (enter) {
            const int keepT = (lanes_here() * k.tri) >> 4;
            do {
                TriData q = load_tri<false>(S, C, ti);
                idx = f2u(q.e.y);
                float t, u, v;
                bool ok = moller_trumbore(v3(q.a.x, q.a.y, q.a.z), v3(q.a.w, q.b.x, q.b.y), v3(q.b.z, q.b.w, q.e.x), o, d, t, u, v);
                if (isShadow) {
                    if (ok && (t < max_t)) { thr = v3(0.0f); occlBits = kRayOccluded; occluded = true; break; }       // NOLEAF scene: any hit ends the ray
                } else if (ok && (t < max_t)) {
                    bool take = t < min_t;
                    if (!take && t == min_t && h.tri >= 0) take = !tie_keeps_first(S, K.mids, o, inv, bestTi, ti);
                    if (take) {
                        min_t = t; bestTi = ti;
                        h.t = t; h.u = u; h.v = v;
                        h.tri = (int32_t)(idx & 0x7fffffffu);
                        h.material = f2i(q.e.z);
                    }
                }
                ti++;
                more = !(idx & 0x80000000u);
                if (more && lanes_here() <= keepT) break;
            } while (more);
        }
        if (!occluded && more) { cur = ~ti; inLeaf = true; continue; }   // the rest of this leaf next time round
        cur = (!occluded && st.sp > 0) ? st.pop() : kRefNone;
        if (occluded) st.sp = 0;
    }
    r.o = o; r.d = d; r.inv = inv; r.max_t = max_t; r.min_t = min_t; r.cur = cur; r.pend = bestTi;
    r.flags = (inLeaf ? kRayInLeaf : 0u) | (busy ? kRayBusy : 0u) | (isShadow ? kRayShadow : 0u) | (extFollows ? kRayExtFollows : 0u) | occlBits;
}

// ---- wide traversal (SIMPLE scenes in HBM) --------------------------------------------------------
// visited(leaf) == slab(leaf's own box) (see inv_is_regular above) holds for ANY hierarchy whose inner boxes contain the
// leaf boxes: a ray that passes a leaf's box passes every enclosing box (monotone slab test), and a leaf whose box it
// misses is never tested. So the tree ABOVE the reference's leaves is ours to choose, and so is the visiting order, as
// long as the RESULT is the reference's: the closest hit is the minimum t, and where two triangles return the same t the
// reference keeps the one it visits first — that case is detected (a test that EQUALS the running minimum) and the ray is
// re-traced on the reference's binary tree in the reference's order; a shadow ray of a scene without MAT_LEAF triangles
// (NOLEAF) is occluded by any hit. Here: the reference tree collapsed to 4-wide nodes (host, pt_api.hip) — half the
// dependent node fetches per ray, one full 128-byte line per fetch, children pushed in no particular order.
// Rays with an irregular 1 / d (zero direction component) take the reference traversal as a whole.
// MEASURED (profiles/r02_ab_wide.log), parity-green: 263 k triangles 752 -> 993 ms, 82 k 215 -> 253 ms — SLOWER. 28 dwords of
// node per visit do not fit the 64 VGPRs this kernel runs best at (596 scratch instructions, many of them in the node loop),
// and a visit fetches all four children's boxes whether or not the ray needs them. Opt-in ("wide" = 1) for the A/B.
constexpr uint32_t kRayTie = 8u, kRaySlow = 16u;

template <int N>
PT_DEV void trace_resume_w4(const DeviceScene& S, const SceneCache& C, const WNode* __restrict__ W, Stack<N>& st, RayState& r, V3 eo, V3 ed, int minBusy,
                            V3& thr, Hit& h, Ctr& c, Keep k) {
    if (!(r.flags & kRayBusy)) return;
    V3 o = r.o, d = r.d, inv = r.inv;
    float max_t = r.max_t, min_t = r.min_t;
    int32_t cur = r.cur;
    bool isShadow = (r.flags & kRayShadow) != 0, extFollows = (r.flags & kRayExtFollows) != 0, tie = (r.flags & kRayTie) != 0, busy = true;
    uint32_t occlBits = r.flags & kRayOccluded;
    const int nLdsW = C.nNodes >> 1;                               // the LDS scene cache holds the first nLdsW wide nodes (2 x 64 B each)
    while (true) {
        const int active = lanes_here();
        if (active <= minBusy) break;
        const int keepN = (active * k.node) >> 4;
        while (cur >= 0) {
            f4v mnx, mny, mnz, mxx, mxy, mxz, rf;
            if (cur < nLdsW) { lds_cf4* p = C.nodes + cur * 8; mnx = p[0]; mny = p[1]; mnz = p[2]; mxx = p[3]; mxy = p[4]; mxz = p[5]; rf = p[6]; }
            else { const f4v* p = reinterpret_cast<const f4v*>(W + cur); mnx = p[0]; mny = p[1]; mnz = p[2]; mxx = p[3]; mxy = p[4]; mxz = p[5]; rf = p[6]; }
            int32_t next = kRefNone;
            float tNext = 0.0f;                                     // the nearest hit child is descended into first (a shadow ray finds its occluder sooner)
#define PT_W4_CHILD(K)                                                                                                    \
            {                                                                                                               \
                float tm_;                                                                                                  \
                const int32_t ref_ = f2i(rf.K);                                                                             \
                if (slab(mnx.K, mny.K, mnz.K, mxx.K, mxy.K, mxz.K, o, inv, tm_) && ref_ != kRefNone) {                      \
                    const bool nearer_ = next == kRefNone || tm_ < tNext;                                                   \
                    if (next != kRefNone) st.push(nearer_ ? next : ref_);                                                   \
                    if (nearer_) { next = ref_; tNext = tm_; }                                                              \
                }                                                                                                           \
            }
            PT_W4_CHILD(x) PT_W4_CHILD(y) PT_W4_CHILD(z) PT_W4_CHILD(w)
#undef PT_W4_CHILD
            cur = next != kRefNone ? next : (st.sp > 0 ? st.pop() : kRefNone);
            if (lanes_here() <= keepN) break;
        }
        if (cur >= 0) continue;
        if (cur == kRefNone) {
            if (isShadow && extFollows) {                           // shadow ray done: start this lane's extension ray
                isShadow = false; extFollows = false;
                o = eo; d = ed; inv = inv3(d); max_t = 999999.0f;
                cur = S.rootRef; st.sp = 0;
                continue;
            }
            busy = false;
            break;
        }
        int32_t ti = ~cur;
        uint32_t idx;
        bool occluded = false, more;
        const int keepT = (lanes_here() * k.tri) >> 4;
        do {
            TriData q = load_tri<false>(S, C, ti);
            idx = f2u(q.e.y);
            float t, u, v;
            bool ok = moller_trumbore(v3(q.a.x, q.a.y, q.a.z), v3(q.a.w, q.b.x, q.b.y), v3(q.b.z, q.b.w, q.e.x), o, d, t, u, v);
            if (isShadow) {
                if (ok && (t < max_t)) { thr = v3(0.0f); occlBits = kRayOccluded; occluded = true; break; }       // NOLEAF scene: any hit ends the ray
            } else if (ok && (t < max_t)) {
                if (t < min_t) {
                    min_t = t; tie = false;
                    h.t = t; h.u = u; h.v = v;
                    h.tri = (int32_t)(idx & 0x7fffffffu);
                    h.material = f2i(q.e.z);
                } else if (t == min_t) tie = true;                 // the reference's visiting order decides: re-traced below
            }
            ti++;
            more = !(idx & 0x80000000u);
            if (more && lanes_here() <= keepT) break;
        } while (more);
        if (!occluded && more) { cur = ~ti; continue; }            // the rest of this leaf next time round
        cur = (!occluded && st.sp > 0) ? st.pop() : kRefNone;
        if (occluded) st.sp = 0;
    }
    r.o = o; r.d = d; r.inv = inv; r.max_t = max_t; r.min_t = min_t; r.cur = cur;
    r.flags = (busy ? kRayBusy : 0u) | (isShadow ? kRayShadow : 0u) | (extFollows ? kRayExtFollows : 0u) | (tie ? kRayTie : 0u) | occlBits;
}

}  // namespace pt
