// pt_trace.h — BVH traversal for gfx950: one ray per lane, traversal stack in LDS.
//
// Replaces BVHSceneIntersect / BVHShadowRay / triangleIntersect / aabbIntersect of the
// reference (integratorUtilities.cuh:8-288) with the same visiting order and the same
// arithmetic, on a re-packed tree:
//   * an internal node record (PNode, 64 B) carries BOTH children's boxes, so one fetch
//     replaces the reference's parent + two child-node fetches;
//   * leaves are not nodes: a negative child ref points straight at the first packed
//     triangle (PTri, 48 B, pre-gathered in BVHindices order, edges pre-subtracted), and a
//     flag on the triangle ends the leaf — no index indirection, no Vertices double hop;
//   * 1/dir is computed once per ray (the reference recomputes it per box, :50-55; same value);
//   * the nearer child is kept in a register instead of being pushed and popped again (the
//     reference pushes far then near and immediately pops near, :161-173; same order);
//   * hit attributes are interpolated once, for the final hit (the reference does it on
//     every improving hit, :113-141; same value, it is a pure function of the winner);
//   * the top of the tree (PNodes are numbered breadth-first) and, for small scenes, all packed
//     triangles are staged once per workgroup in LDS (SceneCache): those fetches become
//     ds_read_b128 instead of L1/L2 round trips.
// Visiting order, tie rules (`tminL < tminR` else right first; strict `t < min_t`) and the
// per-ray counters are exactly the reference's.
#pragma once
#include "pt_device.h"

namespace pt {

struct Ctr {
    uint32_t raysClosest, raysShadow, pops, boxes, tris, hits, draws, iters;
    uint32_t gnodes;    // internal-node fetches a TIMED launch of the same tiles serves from global memory (index >= gnodeFrom): the L1 line-rate roofline of bench.py
    uint32_t gnodeFrom; // ... = the LDS scene-cache extent of the instantiation that launch would run (KParams::gnodeFrom; the counting kernel's own cache is smaller)
#ifdef PT_UTIL
    uint32_t u[8];      // diagnostic build (tools/lane_util.py): {wave-level, lane-level} steps of the node / triangle loops, closest then shadow
#endif
};
// Diagnostic build only (-DPT_UTIL, counting kernels): how many lanes does a trip through a traversal loop
// carry? Every active lane counts itself (slot k+1), the first active lane counts the trip (slot k).
#ifdef PT_UTIL
#define PT_UTIL_STEP(c, k) do { if (COUNT) { (c).u[(k) + 1]++; const uint64_t e_ = __builtin_amdgcn_ballot_w64(true); \
    if (__builtin_amdgcn_mbcnt_hi((uint32_t)(e_ >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)e_, 0u)) == 0u) (c).u[k]++; } } while (0)
#else
#define PT_UTIL_STEP(c, k) do {} while (0)
#endif

typedef float f4v __attribute__((ext_vector_type(4)));
// Bit views of a float BY VALUE. (__builtin_bit_cast applied directly to an ext_vector element
// lvalue, e.g. bit_cast<int>(v.y), reads element 0 with this compiler — always go through these.)
PT_DEV int32_t f2i(float f) { return __builtin_bit_cast(int32_t, f); }
PT_DEV uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
typedef __attribute__((address_space(3))) int32_t lds_i32;
typedef __attribute__((address_space(3))) const f4v lds_cf4;

// Workgroup-shared LDS copy of the first `nNodes` PNodes (4 x 16 B each) and, if the whole scene
// fits, of all `nTris` PTris (3 x 16 B each). nTris is all-or-nothing so a leaf never straddles.
struct SceneCache {
    lds_cf4* nodes; int nNodes;
    lds_cf4* tris; int nTris;
};

// Per-lane traversal stack: N entries in LDS (lane-interleaved: entry k of lane l lives at
// word k*64 + l, so every access is bank-conflict-free), overflow in a global spill area with
// the same interleave. The host sizes the spill from the tree depth; Cornell-class trees never
// reach it. The LDS pointer carries its address space so pops compile to ds_read, not flat loads.
template <int N>
struct Stack {
    lds_i32* lds;
    int32_t* spill;
    int sp;
    // ONCHIP: the host has checked that the tree never needs more than N entries (no spill path).
    template <bool ONCHIP = false>
    PT_DEV void push(int32_t v) {
        if (ONCHIP || sp < N) lds[sp * 64] = v; else spill[(sp - N) * 64] = v;
        sp++;
    }
    template <bool ONCHIP = false>
    PT_DEV int32_t pop() {
        sp--;
        int32_t v;
        if (ONCHIP || sp < N) v = lds[sp * 64]; else v = spill[(sp - N) * 64];
        return v;
    }
};

struct Hit { float t, u, v; int32_t tri; int32_t material; };

// 1 / direction per component (aabbIntersect recomputes it per box, integratorUtilities.cuh:50-55): the exact fast
// reciprocal when all three components are in its proven range — one branch per ray — else the IEEE divisions (a
// component of 0 gives the reference's +-inf, a denormal-range one its huge quotient).
PT_DEV V3 inv3(V3 d) {
#if PT_FAST_RCP
    const float lo = fminf_(fminf_(__builtin_fabsf(d.x), __builtin_fabsf(d.y)), __builtin_fabsf(d.z));
    const float hi = fmaxf_(fmaxf_(__builtin_fabsf(d.x), __builtin_fabsf(d.y)), __builtin_fabsf(d.z));
    if (lo >= 1e-12f && hi <= 1.0e30f) {                 // (NaN components fail the comparisons and take the IEEE path)
        const float rx = __builtin_amdgcn_rcpf(d.x), ry = __builtin_amdgcn_rcpf(d.y), rz = __builtin_amdgcn_rcpf(d.z);
        return v3(__builtin_fmaf(rx, __builtin_fmaf(-d.x, rx, 1.0f), rx), __builtin_fmaf(ry, __builtin_fmaf(-d.y, ry, 1.0f), ry),
                  __builtin_fmaf(rz, __builtin_fmaf(-d.z, rz, 1.0f), rz));
    }
#endif
    return v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
}

// aabbIntersect (integratorUtilities.cuh:44-82) with a hoisted reciprocal direction.
PT_DEV bool slab(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, V3 o, V3 inv, float& tmin) {
    float tx1 = (mnx - o.x) * inv.x, tx2 = (mxx - o.x) * inv.x;
    float tmn = fmaxf_(-1e30f, fminf_(tx1, tx2));
    float tmx = fminf_(1e30f, fmaxf_(tx1, tx2));
    float ty1 = (mny - o.y) * inv.y, ty2 = (mxy - o.y) * inv.y;
    tmn = fmaxf_(tmn, fminf_(ty1, ty2));
    tmx = fminf_(tmx, fmaxf_(ty1, ty2));
    float tz1 = (mnz - o.z) * inv.z, tz2 = (mxz - o.z) * inv.z;
    tmn = fmaxf_(tmn, fminf_(tz1, tz2));
    tmx = fminf_(tmx, fmaxf_(tz1, tz2));
    tmin = tmn;
    return (tmx >= tmn) && (tmx > 0.0f);
}

// The hit test alone, for the FLAT leaf loops (they never use tmin): `tmax >= tmin && tmax > 0` as ONE compare. tmin's lower clamp
// -1e30 becomes the smallest positive float: tmin' = max(tmin, FLT_TRUE_MIN), and tmax >= tmin' <=> tmax >= tmin && tmax > 0 (neither
// side is ever NaN: v_min / v_max drop NaN operands and the clamps are finite; f32 denormals are preserved, .amdhsa_float_denorm_mode_32 3).
// One v_cmp and one s_and less per test: the LDS-resident kernels run at the CU's instruction-issue ceiling.
PT_DEV bool slab_hit(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, V3 o, V3 inv) {
    float tx1 = (mnx - o.x) * inv.x, tx2 = (mxx - o.x) * inv.x;
    float tmn = fmaxf_(1.401298464e-45f, fminf_(tx1, tx2));
    float tmx = fminf_(1e30f, fmaxf_(tx1, tx2));
    float ty1 = (mny - o.y) * inv.y, ty2 = (mxy - o.y) * inv.y;
    tmn = fmaxf_(tmn, fminf_(ty1, ty2));
    tmx = fminf_(tmx, fmaxf_(ty1, ty2));
    float tz1 = (mnz - o.z) * inv.z, tz2 = (mxz - o.z) * inv.z;
    tmn = fmaxf_(tmn, fminf_(tz1, tz2));
    tmx = fminf_(tmx, fmaxf_(tz1, tz2));
    return tmx >= tmn;
}

// triangleIntersect (integratorUtilities.cuh:8-42) on a packed triangle.
PT_DEV bool moller_trumbore(V3 v0, V3 e1, V3 e2, V3 o, V3 d, float& t, float& u, float& v) {
    V3 h = cross(d, e2);
    float a = dot(h, e1);
    if (__builtin_fabsf(a) < 1e-12f) return false;
    float f = rcp_exact(a);                 // == 1.0f / a == (float)(1.0 / (double)a), DESIGN.md §4
    V3 s = o - v0;
    u = f * dot(s, h);
    V3 q = cross(s, e1);
    v = f * dot(d, q);
    t = f * dot(e2, q);
    return ((u >= 0.0f) && (v >= 0.0f) && (u + v <= 1.0f)) && t > 0.0f;
}

// The same test without a divergent branch (trace_resume<..., TRISEL>: the LEAN general bounce for scenes in HBM, +4 % on the glass
// blob; the SIMPLE kernel at 64 VGPRs loses 1-6 % to it and keeps the branches, profiles/r03_ab_scalar_lean.log): the
// degenerate-triangle exit, the range test of the exact reciprocal and the four acceptance tests are lane masks combined at the
// end; the full division only runs when some lane of the wave has |a| > 1e30 (wave-uniform, practically never). Same value of
// (ok, t, u, v) whenever ok — the callers read t, u, v only then.
PT_DEV bool moller_trumbore_sel(V3 v0, V3 e1, V3 e2, V3 o, V3 d, float& t, float& u, float& v) {
    V3 h = cross(d, e2);
    float a = dot(h, e1);
    const float m = __builtin_fabsf(a);
    const float r0 = __builtin_amdgcn_rcpf(a);
    float f = __builtin_fmaf(r0, __builtin_fmaf(-a, r0, 1.0f), r0);          // rcp_exact's fast path: exact for 1e-12 <= |a| <= 1e30
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(m > 1.0e30f) != 0ull, 0)) f = m > 1.0e30f ? 1.0f / a : f;
    V3 s = o - v0;
    u = f * dot(s, h);
    V3 q = cross(s, e1);
    v = f * dot(d, q);
    t = f * dot(e2, q);
    // (u >= 0 && v >= 0 as one compare on v_min: it drops a NaN operand, but then u + v is NaN and the next test fails anyway)
    return (m >= 1e-12f) & (fminf_(u, v) >= 0.0f) & (u + v <= 1.0f) & (t > 0.0f);
}

// ONCHIP (template flag of the traversals): every PNode and PTri is in the LDS scene cache and the stack
// fits its LDS part — Cornell-class scenes. The loops then carry no global-memory path and no branch for it.
#ifndef PT_NODE_OVERLAP
#define PT_NODE_OVERLAP 1
#endif
#ifndef PT_TRISEL_LEAN
#define PT_TRISEL_LEAN 1
#endif
#ifndef PT_SPILL_UNIFORM
#define PT_SPILL_UNIFORM 1
#endif
#ifndef PT_MT_SEL_RESUME
#define PT_MT_SEL_RESUME 1
#endif
struct NodeData { f4v a, b, c, d; };
template <bool ONCHIP = false>
PT_DEV NodeData load_node(const DeviceScene& S, const SceneCache& C, int32_t i) {
    NodeData n;
#if PT_NODE_OVERLAP
    if (!ONCHIP) {
        // Both halves of the wave in flight at once: the lanes whose node is beyond the LDS copy of the tree top issue their four
        // global loads, the others their four LDS reads INTO THE SAME REGISTERS, then one wait for both counters. Written by hand
        // because the compiler puts `s_waitcnt vmcnt(0)` between the two groups (it sees two writers of one register; the lanes
        // are disjoint, the hardware has no such hazard), which adds the LDS round trip to every global one.
        const uint32_t la = (uint32_t)(uintptr_t)C.nodes + (uint32_t)i * 64u;
        const f4v* gp = reinterpret_cast<const f4v*>(S.nodes + i);
        unsigned long long sv;
        asm volatile(
            "s_mov_b64 %[sv], exec\n\t"
            "v_cmp_le_i32 vcc, %[nn], %[i]\n\t"
            "s_and_b64 exec, %[sv], vcc\n\t"
            "s_cbranch_execz .Lng%=\n\t"
            "global_load_dwordx4 %[a], %[gp], off\n\t"
            "global_load_dwordx4 %[b], %[gp], off offset:16\n\t"
            "global_load_dwordx4 %[c], %[gp], off offset:32\n\t"
            "global_load_dwordx4 %[d], %[gp], off offset:48\n\t"
            ".Lng%=:\n\t"
            "s_andn2_b64 exec, %[sv], vcc\n\t"
            "s_cbranch_execz .Lnl%=\n\t"
            "ds_read_b128 %[a], %[la]\n\t"
            "ds_read_b128 %[b], %[la] offset:16\n\t"
            "ds_read_b128 %[c], %[la] offset:32\n\t"
            "ds_read_b128 %[d], %[la] offset:48\n\t"
            ".Lnl%=:\n\t"
            "s_mov_b64 exec, %[sv]\n\t"
            "s_waitcnt vmcnt(0) lgkmcnt(0)"
            : [a] "=&v"(n.a), [b] "=&v"(n.b), [c] "=&v"(n.c), [d] "=&v"(n.d), [sv] "=&s"(sv)
            : [i] "v"(i), [nn] "s"(C.nNodes), [la] "v"(la), [gp] "v"(gp)
            : "vcc", "memory");
        return n;
    }
#endif
    if (ONCHIP || i < C.nNodes) {
        lds_cf4* p = C.nodes + i * 4;
        n.a = p[0]; n.b = p[1]; n.c = p[2]; n.d = p[3];
    } else {
        const f4v* p = reinterpret_cast<const f4v*>(S.nodes + i);
        n.a = p[0]; n.b = p[1]; n.c = p[2]; n.d = p[3];
    }
    return n;
}
struct TriData { f4v a, b, e; };
template <bool ONCHIP = false>
PT_DEV TriData load_tri(const DeviceScene& S, const SceneCache& C, int32_t i) {
    TriData t;
    if (ONCHIP || C.nTris) {               // wave-uniform
        lds_cf4* p = C.tris + i * 3;
        t.a = p[0]; t.b = p[1]; t.e = p[2];
    } else {
        const f4v* p = reinterpret_cast<const f4v*>(S.tris + i);
        t.a = p[0]; t.b = p[1]; t.e = p[2];
    }
    return t;
}

// The dealt-out tests of the FLAT kernels need v0, e1, e2 — 36 of a PTri's 48 bytes: two 16-byte LDS reads and one word, a
// quarter less LDS return per test (C2: 109.7 -> 107.8 ms at 128 spp, profiles/r02_phase_census.md).
struct TriEdges { f4v a, b; float e2z; };
PT_DEV TriEdges load_tri_edges(const SceneCache& C, int32_t i) {
    TriEdges t;
    lds_cf4* p = C.tris + i * 3;
    t.a = p[0]; t.b = p[1];
    t.e2z = *reinterpret_cast<__attribute__((address_space(3))) const float*>(p + 2);
    return t;
}
// A leaf's box and triangle range for the FLAT kernels' wave-uniform loop over the leaves, read from the table in GLOBAL memory
// through the constant address space: a uniform address there is an s_load, the record arrives in SGPRs and the slab tests
// take its fields as their scalar operand — no LDS broadcast (1 KB of LDS return per leaf and wave), no VGPRs and no
// v_readfirstlane for it. The LDS pipeline of a CU is busy half of the headline kernel's time (SQ_LDS_IDX_ACTIVE): C2 111.4 ->
// 109.4 ms at 128 spp.
struct LeafBox { float mnx, mny, mnz, mxx, mxy, mxz; int32_t first, count; };
PT_DEV LeafBox leaf_box(const PLeaf* __restrict__ leaves, int k) {
    typedef __attribute__((address_space(4))) const float konst_f;         // read-only for the kernel's lifetime
    typedef __attribute__((address_space(4))) const int32_t konst_i;
    konst_f* p = (konst_f*)(uintptr_t)(leaves + k);
    LeafBox L;
    L.mnx = p[0]; L.mny = p[1]; L.mnz = p[2]; L.mxx = p[3]; L.mxy = p[4]; L.mxz = p[5];
    L.first = ((konst_i*)p)[6]; L.count = ((konst_i*)p)[7];
    return L;
}

// ---- leaving a loop before its last lane -------------------------------------------------------
// A wave walks `while (cur >= 0) descend` until its LAST lane has reached a leaf, then the triangle loop until
// the longest leaf is done. Without culling a ray pierces many boxes between two leaves, the counts differ
// widely between lanes, and on the 263 k-triangle scene a trip through the node loop carried 8.6 of 64 lanes
// (tools/lane_util.py). So a wave leaves the node loop once no more than `active * node / 16` lanes are still
// descending — the others go and test their leaves, the few keep `cur` and continue next time round — and,
// where enabled, the triangle loop the same way (a lane resumes its leaf at `~ti`). Per ray nothing changes:
// same nodes, same triangles, same order, same counters. +40 % on that scene, +22 % at 82 k triangles; a scene
// in LDS (Cornell) has three node steps between leaves and only pays for the test, so ONCHIP kernels keep the
// plain loops (PT_*_EXIT_ONCHIP for the A/B).
#ifndef PT_NODE_EXIT_ONCHIP
#define PT_NODE_EXIT_ONCHIP 0
#endif
#ifndef PT_TRI_EXIT_ONCHIP
#define PT_TRI_EXIT_ONCHIP 0
#endif
#ifndef PT_TRI_EXIT_HBM
#define PT_TRI_EXIT_HBM 1
#endif
template <bool ONCHIP> struct LoopExit {
    static constexpr bool node = ONCHIP ? (PT_NODE_EXIT_ONCHIP != 0) : true;
    static constexpr bool tri = ONCHIP ? (PT_TRI_EXIT_ONCHIP != 0) : (PT_TRI_EXIT_HBM != 0);
};
struct Keep { int node, tri; };         // sixteenths of the lanes that entered the loop; 0 = stay until the last lane
PT_DEV int lanes_here() { return __builtin_popcountll(__builtin_amdgcn_ballot_w64(true)); }

// One internal-node step shared by both traversals: returns the next ref to visit.
// CULL (opt-in, pt_set_culling): a child whose slab entry lies beyond `cullT` — the best hit so far, or a shadow
// ray's max_t — is not visited. The reference visits it (no such test in aabbIntersect / BVHSceneIntersect), and a
// triangle inside such a box can in principle still return a smaller t (different roundings; Moller-Trumbore at
// grazing incidence), so this mode is NOT the reference's result: 13 of 2 M pixels differ after 4.6e9 rays on the
// 263 k-triangle scene. The default instantiations do not contain the test.
template <bool COUNT, int N, bool ONCHIP = false, bool CULL = false>
PT_DEV int32_t descend_node(const NodeData& n, int32_t cur, V3 o, V3 inv, Stack<N>& st, Ctr& c, float cullT = 0.0f);

template <bool COUNT, int N, bool ONCHIP = false, bool CULL = false>
PT_DEV int32_t descend(const DeviceScene& S, const SceneCache& C, int32_t cur, V3 o, V3 inv, Stack<N>& st, Ctr& c, float cullT = 0.0f) {
    const NodeData n = load_node<ONCHIP>(S, C, cur);
    return descend_node<COUNT, N, ONCHIP, CULL>(n, cur, o, inv, st, c, cullT);
}

// ... the step itself, on a record that is already in registers.
template <bool COUNT, int N, bool ONCHIP, bool CULL>
PT_DEV int32_t descend_node(const NodeData& n, int32_t cur, V3 o, V3 inv, Stack<N>& st, Ctr& c, float cullT) {
    if (COUNT) { c.pops++; c.boxes += 2; if (!ONCHIP && (uint32_t)cur >= c.gnodeFrom) c.gnodes++; }
    float tL, tR;
    bool hL = slab(n.a.x, n.a.y, n.a.z, n.a.w, n.b.x, n.b.y, o, inv, tL);
    bool hR = slab(n.b.z, n.b.w, n.c.x, n.c.y, n.c.z, n.c.w, o, inv, tR);
    if (CULL) { hL = hL && !(tL > cullT); hR = hR && !(tR > cullT); }
    int32_t left = f2i(n.d.x), right = f2i(n.d.y);
#if PT_SPILL_UNIFORM
    if (!ONCHIP && !COUNT && !CULL && N == 16) {        // (kStackLds: the 4-wave kernels)
        // the stack's spill test once per wave and trip instead of inside push and pop (the 4-wave kernels' sixteen LDS entries are
        // rarely exceeded): no lane of the wave at the LDS part's edge -> the forms without a branch for the spill area
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(st.sp >= N - 1) == 0ull, 1)) {
            if (hL && hR) {
                bool leftNear = tL < tR;
                st.template push<true>(leftNear ? right : left);
                return leftNear ? left : right;
            }
            if (hL) return left;
            if (hR) return right;
            return st.sp > 0 ? st.template pop<true>() : kRefNone;
        }
    }
#endif
    if (hL && hR) {
        bool leftNear = tL < tR;
        st.template push<ONCHIP>(leftNear ? right : left);
        return leftNear ? left : right;
    }
    if (hL) return left;
    if (hR) return right;
    return st.sp > 0 ? st.template pop<ONCHIP>() : kRefNone;
}

// BVHSceneIntersect (integratorUtilities.cuh:84-186), max_t as the reference's 999999.
template <bool COUNT, int N, bool ONCHIP, bool CULL>
PT_DEV void trace_closest_plain(const DeviceScene& S, const SceneCache& C, V3 o, V3 d, float max_t, Stack<N>& st, Hit& hit, Ctr& c) {
    V3 inv = inv3(d);
    float min_t = 3.402823466e+38f;
    hit.tri = -1;
    st.sp = 0;
    int32_t cur = S.rootRef;
    if (COUNT) c.raysClosest++;
    while (true) {
        while (cur >= 0) { PT_UTIL_STEP(c, 0); cur = descend<COUNT, N, ONCHIP, CULL>(S, C, cur, o, inv, st, c, min_t); }
        if (cur == kRefNone) break;
        if (COUNT) c.pops++;
        int32_t ti = ~cur;
        uint32_t idx;
        do {
            TriData q = load_tri<ONCHIP>(S, C, ti);
            idx = f2u(q.e.y);
            if (COUNT) c.tris++;
            PT_UTIL_STEP(c, 2);
            float t, u, v;
            bool ok = moller_trumbore(v3(q.a.x, q.a.y, q.a.z), v3(q.a.w, q.b.x, q.b.y), v3(q.b.z, q.b.w, q.e.x), o, d, t, u, v);
            if (ok && (t < min_t) && (t < max_t)) {
                min_t = t;
                hit.t = t; hit.u = u; hit.v = v;
                hit.tri = (int32_t)(idx & 0x7fffffffu);
                hit.material = f2i(q.e.z);
            }
            ti++;
        } while (!(idx & 0x80000000u));
        cur = st.sp > 0 ? st.template pop<ONCHIP>() : kRefNone;
    }
    if (COUNT) { if (hit.tri >= 0) c.hits++; }
}

// ... and with the loop exits of LoopExit (same visits, same order, same counters).
template <bool COUNT, int N, bool ONCHIP, bool CULL>
PT_DEV void trace_closest_exits(const DeviceScene& S, const SceneCache& C, V3 o, V3 d, float max_t, Stack<N>& st, Hit& hit, Ctr& c, Keep k) {
    typedef LoopExit<ONCHIP> X;
    V3 inv = inv3(d);
    float min_t = 3.402823466e+38f;
    hit.tri = -1;
    st.sp = 0;
    int32_t cur = S.rootRef;
    if (COUNT) c.raysClosest++;
    while (true) {
        int keepN = 0;
        if (X::node) keepN = (lanes_here() * k.node) >> 4;
        while (cur >= 0) {
            PT_UTIL_STEP(c, 0);
            cur = descend<COUNT, N, ONCHIP, CULL>(S, C, cur, o, inv, st, c, min_t);
            if (X::node && lanes_here() <= keepN) break;
        }
        if (cur == kRefNone) break;
        if (X::node && cur >= 0) continue;
        int32_t ti = ~cur;
        uint32_t idx;
        int keepT = 0;
        if (X::tri) keepT = (lanes_here() * k.tri) >> 4;
        bool more;
        do {
            TriData q = load_tri<ONCHIP>(S, C, ti);
            idx = f2u(q.e.y);
            if (COUNT) c.tris++;
            PT_UTIL_STEP(c, 2);
            float t, u, v;
            bool ok = moller_trumbore(v3(q.a.x, q.a.y, q.a.z), v3(q.a.w, q.b.x, q.b.y), v3(q.b.z, q.b.w, q.e.x), o, d, t, u, v);
            if (ok && (t < min_t) && (t < max_t)) {
                min_t = t;
                hit.t = t; hit.u = u; hit.v = v;
                hit.tri = (int32_t)(idx & 0x7fffffffu);
                hit.material = f2i(q.e.z);
            }
            ti++;
            more = !(idx & 0x80000000u);
            if (X::tri && more && lanes_here() <= keepT) break;
        } while (more);
        if (X::tri && more) { cur = ~ti; continue; }          // the rest of this leaf next time round
        if (COUNT) c.pops++;
        cur = st.sp > 0 ? st.template pop<ONCHIP>() : kRefNone;
    }
    if (COUNT) { if (hit.tri >= 0) c.hits++; }
}

template <bool COUNT, int N, bool ONCHIP = false, bool CULL = false>
PT_DEV void trace_closest(const DeviceScene& S, const SceneCache& C, V3 o, V3 d, float max_t, Stack<N>& st, Hit& hit, Ctr& c, Keep k = Keep{0, 0}) {
    // (two bodies rather than one with dead branches: the plain loops of the LDS-resident kernels are instruction-bound
    // and the optimiser lays the merged form out differently, -2.7 % on Cornell)
    if (LoopExit<ONCHIP>::node || LoopExit<ONCHIP>::tri) trace_closest_exits<COUNT, N, ONCHIP, CULL>(S, C, o, d, max_t, st, hit, c, k);
    else trace_closest_plain<COUNT, N, ONCHIP, CULL>(S, C, o, d, max_t, st, hit, c);
}

PT_DEV float schlick_fresnel(float cosTheta, float etaI, float etaT) {    // reflectors.cuh:183-188
    float R0 = (etaI - etaT) / (etaI + etaT);
    R0 = R0 * R0;
    return R0 + (1.0f - R0) * pow5_(1.0f - __builtin_fabsf(cosTheta));
}

// BVHShadowRay (integratorUtilities.cuh:188-288): any hit below max_t kills the ray unless the
// triangle's material is MAT_LEAF, which attenuates and continues (cut-off 0.01).
// NOLEAF (SIMPLE scenes, pt_path.h): no triangle carries a MAT_LEAF material, so every hit below max_t ends the ray.
template <bool COUNT, int N, bool ONCHIP, bool CULL, bool NOLEAF = false>
PT_DEV V3 trace_shadow_plain(const DeviceScene& S, const SceneCache& C, V3 o, V3 d, float max_t, Stack<N>& st, Ctr& c) {
    V3 inv = inv3(d);
    V3 thr = v3(1.0f);
    st.sp = 0;
    int32_t cur = S.rootRef;
    if (COUNT) c.raysShadow++;
    while (true) {
        while (cur >= 0) { PT_UTIL_STEP(c, 4); cur = descend<COUNT, N, ONCHIP, CULL>(S, C, cur, o, inv, st, c, max_t); }
        if (cur == kRefNone) break;
        if (COUNT) c.pops++;
        int32_t ti = ~cur;
        uint32_t idx;
        do {
            TriData q = load_tri<ONCHIP>(S, C, ti);
            idx = f2u(q.e.y);
            if (COUNT) c.tris++;
            PT_UTIL_STEP(c, 6);
            float t, u, v;
            bool ok = moller_trumbore(v3(q.a.x, q.a.y, q.a.z), v3(q.a.w, q.b.x, q.b.y), v3(q.b.z, q.b.w, q.e.x), o, d, t, u, v);
            if (ok && (t < max_t)) {
                uint32_t flags = f2u(q.e.w);
                if (NOLEAF || !(flags & 1u)) return v3(0.0f);
                // MAT_LEAF (integratorUtilities.cuh:218-239)
                const PMat& m = S.mats[f2i(q.e.z)];
                const PAttr& at = S.attrs[idx & 0x7fffffffu];
                float bz = 1.0f - u - v;
                V3 n = ld3(at.n0) * bz + ld3(at.n1) * u + ld3(at.n2) * v;
                float cosTheta = __builtin_fabsf(dot(d, normalize(n)));
                float F = schlick_fresnel(cosTheta, 1.0f, m.ior);
                V3 s = ld3(m.albedo) * m.transmission * (1.0f - F);
                thr = thr * s;
                if (fmaxf_(thr.x, fmaxf_(thr.y, thr.z)) < 0.01f) return v3(0.0f);
            }
            ti++;
        } while (!(idx & 0x80000000u));
        cur = st.sp > 0 ? st.template pop<ONCHIP>() : kRefNone;
    }
    return thr;
}

// ... and with the loop exits of LoopExit.
template <bool COUNT, int N, bool ONCHIP, bool CULL, bool NOLEAF = false>
PT_DEV V3 trace_shadow_exits(const DeviceScene& S, const SceneCache& C, V3 o, V3 d, float max_t, Stack<N>& st, Ctr& c, Keep k) {
    typedef LoopExit<ONCHIP> X;
    V3 inv = inv3(d);
    V3 thr = v3(1.0f);
    st.sp = 0;
    int32_t cur = S.rootRef;
    if (COUNT) c.raysShadow++;
    while (true) {
        int keepN = 0;
        if (X::node) keepN = (lanes_here() * k.node) >> 4;
        while (cur >= 0) {
            PT_UTIL_STEP(c, 4);
            cur = descend<COUNT, N, ONCHIP, CULL>(S, C, cur, o, inv, st, c, max_t);
            if (X::node && lanes_here() <= keepN) break;
        }
        if (cur == kRefNone) break;
        if (X::node && cur >= 0) continue;
        int32_t ti = ~cur;
        uint32_t idx;
        int keepT = 0;
        if (X::tri) keepT = (lanes_here() * k.tri) >> 4;
        bool more;
        do {
            TriData q = load_tri<ONCHIP>(S, C, ti);
            idx = f2u(q.e.y);
            if (COUNT) c.tris++;
            PT_UTIL_STEP(c, 6);
            float t, u, v;
            bool ok = moller_trumbore(v3(q.a.x, q.a.y, q.a.z), v3(q.a.w, q.b.x, q.b.y), v3(q.b.z, q.b.w, q.e.x), o, d, t, u, v);
            if (ok && (t < max_t)) {
                uint32_t flags = f2u(q.e.w);
                if (NOLEAF || !(flags & 1u)) { if (COUNT) c.pops++; return v3(0.0f); }
                // MAT_LEAF (integratorUtilities.cuh:218-239)
                const PMat& m = S.mats[f2i(q.e.z)];
                const PAttr& at = S.attrs[idx & 0x7fffffffu];
                float bz = 1.0f - u - v;
                V3 n = ld3(at.n0) * bz + ld3(at.n1) * u + ld3(at.n2) * v;
                float cosTheta = __builtin_fabsf(dot(d, normalize(n)));
                float F = schlick_fresnel(cosTheta, 1.0f, m.ior);
                V3 s = ld3(m.albedo) * m.transmission * (1.0f - F);
                thr = thr * s;
                if (fmaxf_(thr.x, fmaxf_(thr.y, thr.z)) < 0.01f) { if (COUNT) c.pops++; return v3(0.0f); }
            }
            ti++;
            more = !(idx & 0x80000000u);
            if (X::tri && more && lanes_here() <= keepT) break;
        } while (more);
        if (X::tri && more) { cur = ~ti; continue; }
        if (COUNT) c.pops++;
        cur = st.sp > 0 ? st.template pop<ONCHIP>() : kRefNone;
    }
    return thr;
}

template <bool COUNT, int N, bool ONCHIP = false, bool CULL = false, bool NOLEAF = false>
PT_DEV V3 trace_shadow(const DeviceScene& S, const SceneCache& C, V3 o, V3 d, float max_t, Stack<N>& st, Ctr& c, Keep k = Keep{0, 0}) {
    if (LoopExit<ONCHIP>::node || LoopExit<ONCHIP>::tri) return trace_shadow_exits<COUNT, N, ONCHIP, CULL, NOLEAF>(S, C, o, d, max_t, st, c, k);
    return trace_shadow_plain<COUNT, N, ONCHIP, CULL, NOLEAF>(S, C, o, d, max_t, st, c);
}


// ---- FLAT closest hit: tiny LDS-resident scenes (at most 64 W internal nodes and 64 W packed triangles, W = 1 or 2) ----
// Without culling the set of leaves a ray visits does not depend on what it hits: a leaf is visited iff slab() passes
// for every box on the way down. The stack walk of a Cornell-class scene spends its time diverged (a trip through
// the node loop carries 24 of 64 lanes, through the triangle loop 16, tools/lane_util.py). Here the wave works as one:
//   1. it walks ALL internal nodes in index order (breadth-first numbering: parents first) in lockstep; each lane
//      keeps a 64-bit "visited" mask and collects the triangles of the leaves it visits in a 64-bit mask — the
//      same slab() on the same operands, gated exactly as the reference's descent;
//   2. the (ray, triangle) tests of the whole wave — the set bits of all 64 masks — are dealt out evenly: test p
//      goes to lane p mod 64, which looks up the owning ray (prefix sums, in LDS), runs the same Moller-Trumbore
//      on that ray and folds (t, triangle) into the ray's minimum with an LDS atomic;
//   3. every ray's owner re-runs the test on the winner for (t, u, v) — the same arithmetic, so the same bits.
// What this does not know is the reference's visiting ORDER, which decides between two triangles that return the
// same t (strict `t < min_t`: the first one visited wins); step 3 reconstructs that decision from the tree. The
// per-wave LDS scratch is the wave's traversal stack (16 x 64 words), unused meanwhile. Shadow rays and the counting
// kernels keep the exact traversal.
PT_DEV int select64(uint64_t m, int k) {                       // position of the k-th (0-based) set bit of m
    int pos = 0;
    for (int w = 32; w; w >>= 1) {
        const uint64_t low = m & ((1ull << w) - 1ull);
        const int c = (int)__builtin_popcountll(low);
        if (k >= c) { k -= c; m >>= w; pos += w; } else m = low;
    }
    return pos;
}
PT_DEV void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Bit sets of W x 64 bits for the FLAT traversal (W = 1: at most 64 internal nodes / triangles; W = 2: 128).
template <int W> struct BitSet {
    uint64_t w[W];
    PT_DEV static BitSet zero() { BitSet m; for (int j = 0; j < W; j++) m.w[j] = 0ull; return m; }
    PT_DEV static BitSet range(int first, int count) {             // bits [first, first + count), count >= 0
        BitSet m;
        for (int j = 0; j < W; j++) {
            const int lo = first - 64 * j, hi = lo + count;         // the range in this word's coordinates
            const int a = lo < 0 ? 0 : lo, b = hi > 64 ? 64 : hi;
            m.w[j] = (b > a) ? ((~0ull >> (64 - (b - a))) << a) : 0ull;
        }
        return m;
    }
    PT_DEV bool test(int i) const { uint64_t x = w[0]; for (int j = 1; j < W; j++) x = (i >> 6) == j ? w[j] : x; return ((x >> (i & 63)) & 1ull) != 0ull; }
    PT_DEV void or_if(bool c, const BitSet& o) { for (int j = 0; j < W; j++) w[j] |= c ? o.w[j] : 0ull; }
    PT_DEV BitSet operator&(const BitSet& o) const { BitSet m; for (int j = 0; j < W; j++) m.w[j] = w[j] & o.w[j]; return m; }
    PT_DEV bool any() const { uint64_t x = 0ull; for (int j = 0; j < W; j++) x |= w[j]; return x != 0ull; }
    PT_DEV int count() const { int n = 0; for (int j = 0; j < W; j++) n += (int)__builtin_popcountll(w[j]); return n; }
    PT_DEV int first() const {                                     // lowest set bit (the set is not empty)
        int p = 0; bool found = false;
        for (int j = 0; j < W; j++) if (!found && w[j]) { p = 64 * j + (int)__builtin_ctzll(w[j]); found = true; }
        return p;
    }
    PT_DEV void clear_first() {                                    // remove the lowest set bit
        bool done = false;
        for (int j = 0; j < W; j++) if (!done && w[j]) { w[j] &= w[j] - 1ull; done = true; }
    }
    PT_DEV void keep_from_kth(int k) {                             // drop the k lowest set bits
        for (int j = 0; j < W; j++) {
            const int c = (int)__builtin_popcountll(w[j]);
            if (k >= c) { k -= c; w[j] = 0ull; }
            else { if (k > 0) w[j] &= ~((1ull << select64(w[j], k)) - 1ull); k = 0; }
        }
    }
};

// Which leaves does a ray visit? The reference descends into a child iff its box passes aabbIntersect, so a leaf is visited
// iff EVERY box on its root path passes. Those boxes are nested exactly (a parent's box is the float-wise min / max of its
// children's), and slab() is monotone in the box: x -> RN(x - o) and y -> RN(y * inv) are monotone maps, min / max keep
// order, so per axis the parent's t-interval contains the child's, the parent's tmin is <= and its tmax >= the child's —
// if the child's box passes (`tmax >= tmin && tmax > 0`), the parent's does. Hence: visited(leaf) == slab(leaf's own box),
// with no tree walk at all — PROVIDED no NaN enters, i.e. every 1 / d component is finite and non-zero (a zero or denormal
// direction component gives inf, and 0 * inf = NaN is dropped by v_min / v_max in a way that is not monotone). A wave in
// which some ray fails that test takes the lockstep node walk for that pass (rare: a direction component must be exactly 0).
PT_DEV bool inv_is_regular(V3 inv) {
    const float a = __builtin_fabsf(inv.x), b = __builtin_fabsf(inv.y), cc = __builtin_fabsf(inv.z);
    return a > 0.0f && a < __builtin_inff() && b > 0.0f && b < __builtin_inff() && cc > 0.0f && cc < __builtin_inff();     // false for NaN too
}

// All 64 lanes call this together; `active` says who has a ray. PNode.pad0 / pad1 hold the number of triangles
// below the left / right child (patched in by the host for scenes that qualify, pt_api.hip).
template <int N, int W = 1>
PT_DEV void trace_closest_flat(const DeviceScene& S, const SceneCache& C, bool active, V3 o, V3 d, float max_t, Stack<N>& st, Hit& hit, Ctr& c, int nInternal,
                               const PLeaf* __restrict__ leaves = nullptr, int nLeaves = 0) {
    static_assert(N >= 11 + 2 * W, "the scratch layout needs (11 + 2 W) x 64 words of the wave's stack");
    typedef __attribute__((address_space(3))) unsigned long long lds_u64;
    typedef BitSet<W> Set;
    constexpr int kTm = 6, kPre = 6 + 2 * W, kKeys = 7 + 2 * W;     // scratch fields: o 0-2, d 3-5, tm, exclusive prefix, 2 x u64 keys
    constexpr int kMaxI = 64 * W - 1;
    const int lane = (int)(threadIdx.x & 63u);
    lds_i32* Wd = st.lds - lane;                                 // the wave's stack words; field f of lane l at Wd[f * 64 + l]
    const V3 inv = inv3(d);
    // 1. lockstep node walk
    Set tm = Set::zero();
    if (S.rootRef < 0) { if (active) tm = Set::range(0, S.nTris); }      // the root is the only leaf
    else if (nLeaves > 0 && __builtin_amdgcn_ballot_w64(active && !inv_is_regular(inv)) == 0ull) {
        for (int k = 0; k < nLeaves; ++k) {                       // the leaves' own boxes (see inv_is_regular): wave-uniform loop
            const LeafBox L = leaf_box(leaves, k);               // through the scalar cache: SGPR operands of the slab test
            tm.or_if(slab_hit(L.mnx, L.mny, L.mnz, L.mxx, L.mxy, L.mxz, o, inv), Set::range(L.first, L.count));
        }
        if (!active) tm = Set::zero();                           // (once, not per leaf)
    } else {
        Set vis = Set::zero();
        if (active) vis = Set::range(S.rootRef, 1);
        for (int i = 0; i < nInternal; ++i) {                     // wave-uniform loop
            const bool v = vis.test(i);
            if (__builtin_amdgcn_ballot_w64(v) == 0ull) continue;
            const NodeData n = load_node<true>(S, C, i);          // uniform address: an LDS broadcast
            float tL, tR;
            const bool hL = slab(n.a.x, n.a.y, n.a.z, n.a.w, n.b.x, n.b.y, o, inv, tL) && v;
            const bool hR = slab(n.b.z, n.b.w, n.c.x, n.c.y, n.c.z, n.c.w, o, inv, tR) && v;
            const int32_t left = __builtin_amdgcn_readfirstlane(f2i(n.d.x)), right = __builtin_amdgcn_readfirstlane(f2i(n.d.y));
            const int32_t cntL = __builtin_amdgcn_readfirstlane(f2i(n.d.z)), cntR = __builtin_amdgcn_readfirstlane(f2i(n.d.w));
            if (left >= 0) vis.or_if(hL, Set::range(left, 1)); else tm.or_if(hL, Set::range(~left, cntL));
            if (right >= 0) vis.or_if(hR, Set::range(right, 1)); else tm.or_if(hR, Set::range(~right, cntR));
        }
    }
    // 2. deal the tests out: lane j takes tests [j * per, (j + 1) * per) of the wave's sequence (rays in lane order,
    //    triangles in index order) — consecutive tests mostly belong to one ray, which the lane keeps in registers
    const int n = tm.count();
    int pre = n;                                                  // inclusive prefix sum: DPP inside rows of 16, then the row totals
    pre += __builtin_amdgcn_update_dpp(0, pre, 0x111, 0xf, 0xf, true);      // row_shr:1
    pre += __builtin_amdgcn_update_dpp(0, pre, 0x112, 0xf, 0xf, true);      // row_shr:2
    pre += __builtin_amdgcn_update_dpp(0, pre, 0x114, 0xf, 0xf, true);      // row_shr:4
    pre += __builtin_amdgcn_update_dpp(0, pre, 0x118, 0xf, 0xf, true);      // row_shr:8
    const int r0 = __builtin_amdgcn_readlane(pre, 15), r1 = __builtin_amdgcn_readlane(pre, 31), r2 = __builtin_amdgcn_readlane(pre, 47);
    pre += (lane >= 16 ? r0 : 0) + (lane >= 32 ? r1 : 0) + (lane >= 48 ? r2 : 0);
    const int total = __builtin_amdgcn_readlane(pre, 63);         // wave-uniform
    pre -= n;                                                    // exclusive prefix
    Wd[0 * 64 + lane] = __builtin_bit_cast(int32_t, o.x); Wd[1 * 64 + lane] = __builtin_bit_cast(int32_t, o.y); Wd[2 * 64 + lane] = __builtin_bit_cast(int32_t, o.z);
    Wd[3 * 64 + lane] = __builtin_bit_cast(int32_t, d.x); Wd[4 * 64 + lane] = __builtin_bit_cast(int32_t, d.y); Wd[5 * 64 + lane] = __builtin_bit_cast(int32_t, d.z);
    for (int j = 0; j < W; j++) { Wd[(kTm + 2 * j) * 64 + lane] = (int32_t)(uint32_t)tm.w[j]; Wd[(kTm + 2 * j + 1) * 64 + lane] = (int32_t)(uint32_t)(tm.w[j] >> 32); }
    Wd[kPre * 64 + lane] = pre;
    lds_u64* kLo = (lds_u64*)(Wd + kKeys * 64);                   // min over (t bits, triangle index)
    lds_u64* kHi = (lds_u64*)(Wd + (kKeys + 2) * 64);             // min over (t bits, kMaxI - triangle index)
    kLo[lane] = ~0ull; kHi[lane] = ~0ull;
    wave_lds_sync();
    auto load_set = [&](int l) {
        Set m;
        for (int j = 0; j < W; j++) m.w[j] = (uint64_t)(uint32_t)Wd[(kTm + 2 * j) * 64 + l] | ((uint64_t)(uint32_t)Wd[(kTm + 2 * j + 1) * 64 + l] << 32);
        return m;
    };
    const int per = (total + 63) >> 6;
    int l = 0;
    Set rem = Set::zero();
    V3 ro = v3(0.0f), rd = v3(0.0f);
    if (per > 0) {                                                // wave-uniform
        // every lane runs exactly `per` tests [p, p + per) inside [0, total): the last lanes repeat tests of their neighbours
        // (idempotent) and the loop below needs no per-lane guard (see trace_pair_flat)
        int p = lane * per;
        p = p < total - per ? p : total - per;
        for (int s = 32; s; s >>= 1) { const int cand = l + s; if (Wd[kPre * 64 + cand] <= p) l = cand; }      // owner of test p: the last lane whose exclusive prefix is <= p
        rem = load_set(l);
        rem.keep_from_kth(p - Wd[kPre * 64 + l]);                                                             // its triangles from the (p - prefix)-th on
        ro = v3(__builtin_bit_cast(float, Wd[0 * 64 + l]), __builtin_bit_cast(float, Wd[1 * 64 + l]), __builtin_bit_cast(float, Wd[2 * 64 + l]));
        rd = v3(__builtin_bit_cast(float, Wd[3 * 64 + l]), __builtin_bit_cast(float, Wd[4 * 64 + l]), __builtin_bit_cast(float, Wd[5 * 64 + l]));
    }
    for (int trip = 0; trip < per; ++trip) {                      // wave-uniform loop
        while (!rem.any()) {                                      // next ray that has tests (there is one: the lane's tests end below `total`)
            l++;
            rem = load_set(l);
            ro = v3(__builtin_bit_cast(float, Wd[0 * 64 + l]), __builtin_bit_cast(float, Wd[1 * 64 + l]), __builtin_bit_cast(float, Wd[2 * 64 + l]));
            rd = v3(__builtin_bit_cast(float, Wd[3 * 64 + l]), __builtin_bit_cast(float, Wd[4 * 64 + l]), __builtin_bit_cast(float, Wd[5 * 64 + l]));
        }
        const int ti = rem.first();
        rem.clear_first();
        const TriEdges q = load_tri_edges(C, ti);
        float t, u, v;
        const bool ok = moller_trumbore_sel(v3(q.a.x, q.a.y, q.a.z), v3(q.a.w, q.b.x, q.b.y), v3(q.b.z, q.b.w, q.e2z), ro, rd, t, u, v);
        if (ok & (t < max_t)) {
            const uint64_t tb = (uint64_t)f2u(t) << 32;               // t > 0: the bit pattern orders like the value
            __hip_atomic_fetch_min(kLo + l, (unsigned long long)(tb | (uint64_t)(uint32_t)ti), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_min(kHi + l, (unsigned long long)(tb | (uint64_t)(uint32_t)(kMaxI - ti)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    wave_lds_sync();
    // 3. the winner. If more than one triangle returned the minimum, the reference keeps the one it visits first:
    //    collect the tied set (re-test the ray's own triangles, rare) and walk down from the root — where the set has
    //    members on both sides both children were hit, and the reference enters the nearer one (`tL < tR`, else the
    //    right one) first; in a leaf the lowest index wins.
    if (active) {
        const uint64_t a = kLo[lane], b = kHi[lane];
        hit.tri = -1;
        if (a != ~0ull) {
            int win = (int)(uint32_t)a;
            const int last = kMaxI - (int)(uint32_t)b;
            if (win != last) {
                const uint32_t tmin = (uint32_t)(a >> 32);
                Set tied = Set::zero();
                for (Set rest = tm; rest.any(); rest.clear_first()) {
                    const int ti = rest.first();
                    const TriData q = load_tri<true>(S, C, ti);
                    float t, u, v;
                    const bool ok = moller_trumbore(v3(q.a.x, q.a.y, q.a.z), v3(q.a.w, q.b.x, q.b.y), v3(q.b.z, q.b.w, q.e.x), o, d, t, u, v);
                    tied.or_if(ok && (t < max_t) && f2u(t) == tmin, Set::range(ti, 1));
                }
                int32_t ref = S.rootRef;
                int lo = 0;                                        // first triangle of the current subtree (leaf order = left to right)
                while (ref >= 0) {
                    const NodeData nd = load_node<true>(S, C, ref);
                    const int mid = lo + f2i(nd.d.z), hi = mid + f2i(nd.d.w);
                    const Set inL = tied & Set::range(lo, mid - lo), inR = tied & Set::range(mid, hi - mid);
                    bool goLeft = !inR.any();
                    if (inL.any() && inR.any()) {
                        float tL, tR;
                        slab(nd.a.x, nd.a.y, nd.a.z, nd.a.w, nd.b.x, nd.b.y, o, inv, tL);
                        slab(nd.b.z, nd.b.w, nd.c.x, nd.c.y, nd.c.z, nd.c.w, o, inv, tR);
                        goLeft = tL < tR;
                    }
                    ref = goLeft ? f2i(nd.d.x) : f2i(nd.d.y);
                    if (!goLeft) lo = mid;
                    tied = goLeft ? inL : inR;
                }
                win = tied.first();                                // inside the leaf: the first in leaf order
            }
            const TriData q = load_tri<true>(S, C, win);
            float t, u, v;
            moller_trumbore(v3(q.a.x, q.a.y, q.a.z), v3(q.a.w, q.b.x, q.b.y), v3(q.b.z, q.b.w, q.e.x), o, d, t, u, v);
            hit.t = t; hit.u = u; hit.v = v;
            hit.tri = (int32_t)(f2u(q.e.y) & 0x7fffffffu);
            hit.material = f2i(q.e.z);
        }
    }
    wave_lds_sync();                                              // the scratch becomes the stack again
}

// ---- FLAT for a PAIR of rays per lane (DEFER logic step): the shadow ray of the bounce just shaded and the extension ray
// that continues the path share ONE lockstep node walk (node fetch, child refs, loop control paid once for both), and
// their triangle tests are dealt out over the wave together. A shadow ray needs no order at all when no triangle of the
// scene is a MAT_LEAF (NOLEAF scenes: any hit below max_t occludes), the extension ray is resolved as in
// trace_closest_flat. At most 64 internal nodes / triangles. Scratch: 24 x 64 words of the wave's stack area:
//   entry v = lane (extension ray) or 64 + lane (shadow ray): o, d, max_t, triangle mask (9 fields x 128), exclusive
//   prefix (128), the two u64 keys of the extension rays (2 x 128 words), the occlusion flags (64).
template <int N>
PT_DEV void trace_pair_flat(const DeviceScene& S, const SceneCache& C, Stack<N>& st, bool hasShadow, V3 so, V3 sd, float smaxt,
                            bool hasExt, V3 eo, V3 ed, V3& thr, Hit& hit, Ctr& c, int nInternal, const PLeaf* __restrict__ leaves = nullptr, int nLeaves = 0) {
    static_assert(N >= 24, "the scratch layout needs 24 x 64 words of the wave's stack area");
    typedef __attribute__((address_space(3))) unsigned long long lds_u64;
    const int lane = (int)(threadIdx.x & 63u);
    lds_i32* Wd = st.lds - lane;
    // (the occlusion flags of the 64 shadow rays reuse the first half of the prefix field: the prefixes are only read by the owner
    // search, which every lane of the wave has finished before the first test can set a flag — 24 x 256 B per wave instead of 25,
    // which is what lets a fifth 4-wave workgroup fit a CU's 160 KB next to its copy of the Cornell scene)
    constexpr int kPre = 9 * 128, kKeys = kPre + 128, kOcc = kPre;
    const V3 invE = inv3(ed), invS = inv3(sd);
    // 1. one lockstep node walk for both rays
    uint64_t tmE = 0ull, tmS = 0ull;
    if (S.rootRef < 0) { const uint64_t all = ~0ull >> (64 - S.nTris); tmE = hasExt ? all : 0ull; tmS = hasShadow ? all : 0ull; }
    else if (nLeaves > 0 && __builtin_amdgcn_ballot_w64((hasExt && !inv_is_regular(invE)) || (hasShadow && !inv_is_regular(invS))) == 0ull) {
        _Pragma("unroll 2")
        for (int k = 0; k < nLeaves; ++k) {                       // the leaves' own boxes, both rays (see inv_is_regular)
            const LeafBox L = leaf_box(leaves, k);                // through the scalar cache: SGPR operands of the slab tests
            const bool eH = slab_hit(L.mnx, L.mny, L.mnz, L.mxx, L.mxy, L.mxz, eo, invE);
            const bool sH = slab_hit(L.mnx, L.mny, L.mnz, L.mxx, L.mxy, L.mxz, so, invS);
            const uint64_t m = ((1ull << (uint32_t)L.count) - 1ull) << (uint32_t)L.first;      // s_bfm_b64 (a leaf of a tree with two or more leaves holds < 64 triangles)
            tmE |= eH ? m : 0ull; tmS |= sH ? m : 0ull;
        }
        tmE = hasExt ? tmE : 0ull; tmS = hasShadow ? tmS : 0ull;     // (once, not per leaf)
    } else {
        uint64_t visE = hasExt ? 1ull << (uint32_t)S.rootRef : 0ull, visS = hasShadow ? 1ull << (uint32_t)S.rootRef : 0ull;
        for (int i = 0; i < nInternal; ++i) {                     // wave-uniform loop
            const bool vE = ((visE >> i) & 1ull) != 0ull, vS = ((visS >> i) & 1ull) != 0ull;
            if (__builtin_amdgcn_ballot_w64(vE || vS) == 0ull) continue;
            const NodeData n = load_node<true>(S, C, i);
            float t0, t1;
            const bool eL = slab(n.a.x, n.a.y, n.a.z, n.a.w, n.b.x, n.b.y, eo, invE, t0) && vE;
            const bool eR = slab(n.b.z, n.b.w, n.c.x, n.c.y, n.c.z, n.c.w, eo, invE, t1) && vE;
            const bool sL = slab(n.a.x, n.a.y, n.a.z, n.a.w, n.b.x, n.b.y, so, invS, t0) && vS;
            const bool sR = slab(n.b.z, n.b.w, n.c.x, n.c.y, n.c.z, n.c.w, so, invS, t1) && vS;
            const int32_t left = __builtin_amdgcn_readfirstlane(f2i(n.d.x)), right = __builtin_amdgcn_readfirstlane(f2i(n.d.y));
            const int32_t cntL = __builtin_amdgcn_readfirstlane(f2i(n.d.z)), cntR = __builtin_amdgcn_readfirstlane(f2i(n.d.w));
            const uint64_t bL = left >= 0 ? 1ull << (uint32_t)left : (~0ull >> (64 - cntL)) << (uint32_t)(~left);
            const uint64_t bR = right >= 0 ? 1ull << (uint32_t)right : (~0ull >> (64 - cntR)) << (uint32_t)(~right);
            if (left >= 0) { visE |= eL ? bL : 0ull; visS |= sL ? bL : 0ull; } else { tmE |= eL ? bL : 0ull; tmS |= sL ? bL : 0ull; }
            if (right >= 0) { visE |= eR ? bR : 0ull; visS |= sR ? bR : 0ull; } else { tmE |= eR ? bR : 0ull; tmS |= sR ? bR : 0ull; }
        }
    }
    // 2. deal the tests of all 128 rays out: extension rays first (entries 0..63), then shadow rays (64..127)
    auto scan = [&](int v, int& total) {                          // inclusive prefix sum over the wave
        v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
        const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31), r2 = __builtin_amdgcn_readlane(v, 47);
        v += (lane >= 16 ? r0 : 0) + (lane >= 32 ? r1 : 0) + (lane >= 48 ? r2 : 0);
        total = __builtin_amdgcn_readlane(v, 63);
        return v;
    };
    const int nE = (int)__builtin_popcountll(tmE), nS = (int)__builtin_popcountll(tmS);
    int totalE, totalS;
    const int preE = scan(nE, totalE) - nE, preS = totalE + scan(nS, totalS) - nS;
    const int total = totalE + totalS;
    auto put = [&](int v, V3 ro, V3 rd, float mt, uint64_t tm, int pre) {
        Wd[0 * 128 + v] = __builtin_bit_cast(int32_t, ro.x); Wd[1 * 128 + v] = __builtin_bit_cast(int32_t, ro.y); Wd[2 * 128 + v] = __builtin_bit_cast(int32_t, ro.z);
        Wd[3 * 128 + v] = __builtin_bit_cast(int32_t, rd.x); Wd[4 * 128 + v] = __builtin_bit_cast(int32_t, rd.y); Wd[5 * 128 + v] = __builtin_bit_cast(int32_t, rd.z);
        Wd[6 * 128 + v] = __builtin_bit_cast(int32_t, mt);
        Wd[7 * 128 + v] = (int32_t)(uint32_t)tm; Wd[8 * 128 + v] = (int32_t)(uint32_t)(tm >> 32);
        Wd[kPre + v] = pre;
    };
    put(lane, eo, ed, 999999.0f, tmE, preE);
    put(64 + lane, so, sd, smaxt, tmS, preS);
    lds_u64* kLo = (lds_u64*)(Wd + kKeys);
    lds_u64* kHi = (lds_u64*)(Wd + kKeys + 128);
    kLo[lane] = ~0ull; kHi[lane] = ~0ull;
    wave_lds_sync();
    const int per = (total + 63) >> 6;
    int l = 0;
    uint64_t rem = 0ull;
    V3 ro = v3(0.0f), rd = v3(0.0f);
    float rmax = 0.0f;
    auto fetch = [&](int v) {
        rem = (uint64_t)(uint32_t)Wd[7 * 128 + v] | ((uint64_t)(uint32_t)Wd[8 * 128 + v] << 32);
        ro = v3(__builtin_bit_cast(float, Wd[0 * 128 + v]), __builtin_bit_cast(float, Wd[1 * 128 + v]), __builtin_bit_cast(float, Wd[2 * 128 + v]));
        rd = v3(__builtin_bit_cast(float, Wd[3 * 128 + v]), __builtin_bit_cast(float, Wd[4 * 128 + v]), __builtin_bit_cast(float, Wd[5 * 128 + v]));
        rmax = __builtin_bit_cast(float, Wd[6 * 128 + v]);
    };
    if (per > 0) {                                                // wave-uniform
        // Every lane runs exactly `per` tests, [p, p + per): the last lanes start early enough to stay inside [0, total) and repeat
        // tests of their neighbours — a repeated test changes nothing (same keys into the same minima, the same flag), and the loop
        // below needs no per-lane guard: this kernel runs at the CU's instruction-issue ceiling, a guard is three instructions per trip.
        int p = lane * per;
        p = p < total - per ? p : total - per;
        l = p >= totalE ? 64 : 0;                                  // owner of test p among the 128 entries: the shadow rays' tests start at totalE,
        for (int sft = 32; sft; sft >>= 1) { const int cand = l + sft; if (Wd[kPre + cand] <= p) l = cand; }     // six dependent LDS reads for the rest
        fetch(l);
        rem &= ~((1ull << select64(rem, p - Wd[kPre + l])) - 1ull);
    }
    wave_lds_sync();                                              // every lane has read its prefixes ...
    Wd[kOcc + lane] = 0;                                          // ... their first 64 words now hold the shadow rays' "occluded" flags
    wave_lds_sync();
    for (int trip = 0; trip < per; ++trip) {                      // wave-uniform loop
        while (rem == 0ull) { l++; fetch(l); }                    // next ray that has tests (there is one: the lane's tests end below `total`)
        const int ti = __builtin_ctzll(rem);
        rem &= rem - 1ull;
        const TriEdges q = load_tri_edges(C, ti);
        float t, u, v;
        const bool ok = moller_trumbore_sel(v3(q.a.x, q.a.y, q.a.z), v3(q.a.w, q.b.x, q.b.y), v3(q.b.z, q.b.w, q.e2z), ro, rd, t, u, v);
        if (ok & (t < rmax)) {
            if (l < 64) {
                const uint64_t tb = (uint64_t)f2u(t) << 32;
                __hip_atomic_fetch_min(kLo + l, (unsigned long long)(tb | (uint64_t)(uint32_t)ti), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_min(kHi + l, (unsigned long long)(tb | (uint64_t)(uint32_t)(63 - ti)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else Wd[kOcc + (l - 64)] = 1;                        // NOLEAF: any hit below max_t ends the shadow ray (BVHShadowRay returns 0)
        }
    }
    wave_lds_sync();
    // 3. results: the shadow ray's throughput, the extension ray's winner (ties as in trace_closest_flat)
    thr = (hasShadow && Wd[kOcc + lane] != 0) ? v3(0.0f) : v3(1.0f);
    hit.tri = -1; hit.t = 0.0f; hit.u = 0.0f; hit.v = 0.0f; hit.material = 0;
    if (hasExt) {
        const uint64_t a = kLo[lane], b = kHi[lane];
        if (a != ~0ull) {
            int win = (int)(uint32_t)a;
            const int last = 63 - (int)(uint32_t)b;
            if (win != last) {
                const uint32_t tmin = (uint32_t)(a >> 32);
                uint64_t tied = 0ull;
                const uint64_t mine = tmE;
                const V3 invT = invE;
                for (uint64_t rest = mine; rest; rest &= rest - 1ull) {
                    const int ti = __builtin_ctzll(rest);
                    const TriData q = load_tri<true>(S, C, ti);
                    float t, u, v;
                    const bool ok = moller_trumbore(v3(q.a.x, q.a.y, q.a.z), v3(q.a.w, q.b.x, q.b.y), v3(q.b.z, q.b.w, q.e.x), eo, ed, t, u, v);
                    if (ok && (t < 999999.0f) && f2u(t) == tmin) tied |= 1ull << ti;
                }
                int32_t ref = S.rootRef;
                int lo = 0;
                while (ref >= 0) {
                    const NodeData nd = load_node<true>(S, C, ref);
                    const int mid = lo + f2i(nd.d.z), hi = mid + f2i(nd.d.w);
                    const uint64_t inL = tied & ((~0ull >> (64 - (mid - lo))) << lo), inR = tied & ((~0ull >> (64 - (hi - mid))) << mid);
                    bool goLeft = inR == 0ull;
                    if (inL != 0ull && inR != 0ull) {
                        float tL, tR;
                        slab(nd.a.x, nd.a.y, nd.a.z, nd.a.w, nd.b.x, nd.b.y, eo, invT, tL);
                        slab(nd.b.z, nd.b.w, nd.c.x, nd.c.y, nd.c.z, nd.c.w, eo, invT, tR);
                        goLeft = tL < tR;
                    }
                    ref = goLeft ? f2i(nd.d.x) : f2i(nd.d.y);
                    if (!goLeft) lo = mid;
                    tied = goLeft ? inL : inR;
                }
                win = __builtin_ctzll(tied);
            }
            const TriData q = load_tri<true>(S, C, win);
            float t, u, v;
            moller_trumbore_sel(v3(q.a.x, q.a.y, q.a.z), v3(q.a.w, q.b.x, q.b.y), v3(q.b.z, q.b.w, q.e.x), eo, ed, t, u, v);     // (a known hit: same t, u, v)
            hit.t = t; hit.u = u; hit.v = v;
            hit.tri = (int32_t)(f2u(q.e.y) & 0x7fffffffu);
            hit.material = f2i(q.e.z);
        }
    }
    wave_lds_sync();
}

// ---- unified per-lane traverser -------------------------------------------------------------
// The same two traversals as above as ONE resumable state machine, so that a lane can run its
// rays back to back inside a single loop: the megakernel traces a bounce's shadow ray and the next
// extension ray as a pair (the wave re-converges once per pair instead of once per ray), and the
// wavefront kernel refills finished lanes from a ray queue. Visiting order, tests and counters per
// ray are identical to trace_closest / trace_shadow.
template <int N>
struct Trav {
    V3 o, d, inv;
    float max_t, min_t;
    int32_t cur;
    bool shadow;
    Hit hit;
    V3 thr;

    template <bool COUNT>
    PT_DEV void start(const DeviceScene& S, Stack<N>& st, V3 o_, V3 d_, float maxt, bool shadow_, Ctr& c) {
        o = o_; d = d_;
        inv = inv3(d);
        max_t = maxt; min_t = 3.402823466e+38f;
        cur = S.rootRef; shadow = shadow_;
        hit.tri = -1; hit.t = 0.0f; hit.u = 0.0f; hit.v = 0.0f; hit.material = 0;
        thr = v3(1.0f);
        st.sp = 0;
        if (COUNT) { if (shadow_) c.raysShadow++; else c.raysClosest++; }
    }

    // Descend to the next leaf and test its triangles. Returns true when the ray is finished.
    template <bool COUNT>
    PT_DEV bool step(const DeviceScene& S, const SceneCache& C, Stack<N>& st, Ctr& c, Keep k = Keep{0, 0}) {
        typedef LoopExit<false> X;
        int keepN = 0;
        if (X::node) keepN = (lanes_here() * k.node) >> 4;
        while (cur >= 0) {
            cur = descend<COUNT, N>(S, C, cur, o, inv, st, c);
            if (X::node && lanes_here() <= keepN) break;
        }
        if (X::node && cur >= 0) return false;          // still descending: next call
        if (cur == kRefNone) return true;
        int32_t ti = ~cur;
        uint32_t idx;
        int keepT = 0;
        if (X::tri) keepT = (lanes_here() * k.tri) >> 4;
        bool more;
        do {
            TriData q = load_tri(S, C, ti);
            idx = f2u(q.e.y);
            if (COUNT) c.tris++;
            float t, u, v;
            bool ok = moller_trumbore(v3(q.a.x, q.a.y, q.a.z), v3(q.a.w, q.b.x, q.b.y), v3(q.b.z, q.b.w, q.e.x), o, d, t, u, v);
            if (shadow) {
                if (ok && (t < max_t)) {
                    uint32_t flags = f2u(q.e.w);
                    if (!(flags & 1u)) { if (COUNT) c.pops++; thr = v3(0.0f); cur = kRefNone; return true; }
                    // MAT_LEAF (integratorUtilities.cuh:218-239)
                    const PMat& m = S.mats[f2i(q.e.z)];
                    const PAttr& at = S.attrs[idx & 0x7fffffffu];
                    float bz = 1.0f - u - v;
                    V3 n = ld3(at.n0) * bz + ld3(at.n1) * u + ld3(at.n2) * v;
                    float cosTheta = __builtin_fabsf(dot(d, normalize(n)));
                    float F = schlick_fresnel(cosTheta, 1.0f, m.ior);
                    V3 sc = ld3(m.albedo) * m.transmission * (1.0f - F);
                    thr = thr * sc;
                    if (fmaxf_(thr.x, fmaxf_(thr.y, thr.z)) < 0.01f) { if (COUNT) c.pops++; thr = v3(0.0f); cur = kRefNone; return true; }
                }
            } else if (ok && (t < min_t) && (t < max_t)) {
                min_t = t;
                hit.t = t; hit.u = u; hit.v = v;
                hit.tri = (int32_t)(idx & 0x7fffffffu);
                hit.material = f2i(q.e.z);
            }
            ti++;
            more = !(idx & 0x80000000u);
            if (X::tri && more && lanes_here() <= keepT) break;
        } while (more);
        if (X::tri && more) { cur = ~ti; return false; }       // the rest of this leaf next call
        if (COUNT) c.pops++;
        cur = st.sp > 0 ? st.pop() : kRefNone;
        return cur == kRefNone;
    }
};

// ---- resumable pair traversal (REFILL kernels) ------------------------------------------------
// trace_pair leaves a wave in its loops until its LAST lane has finished; without culling the rays of
// one wave differ widely in length, and on the 263 k-triangle scene a trip through the node loop carries 8.6 of
// 64 lanes on average (tools/lane_util.py). Here the traversal state of a lane lives in a RayState that
// survives the call: the wave leaves the loops as soon as no more than `minBusy` lanes are still tracing,
// the lanes that finished run their logic step (shade, next-event record, regenerate) and come back with new
// rays while the others simply continue where they stopped. Per ray the visiting order, the tests and the
// counters are those of trace_closest / trace_shadow.
// What a lane keeps between two calls is kept small on purpose — the kernel for scenes in HBM runs at 64 VGPRs and every
// register that lives across the loops below is one it spills around the logic step: the reciprocal direction is recomputed
// on entry (a dozen instructions per call against hundreds of node steps), the best t so far lives in the caller's Hit (h.t,
// FLT_MAX until something is hit), and the shadow ray of a scene without MAT_LEAF triangles reports "occluded" in a flag bit
// instead of a three-register throughput.
#ifndef PT_RS_KEEP_INV
#define PT_RS_KEEP_INV 0
#endif
#ifndef PT_OCCL_BOOL
#define PT_OCCL_BOOL 0              // 1: "occluded" as a lane mask (an SGPR pair live through the traversal loops: the compiler then spills SGPRs INSIDE them, -3 % on scenes in HBM)
#endif
struct RayState {
    V3 o, d;
    float max_t;
    int32_t cur;
#if PT_RS_KEEP_INV && !defined(PT_EXPERIMENTAL)
    V3 inv;
#endif
#ifdef PT_EXPERIMENTAL
    V3 inv; float min_t;
    int32_t pend;                      // trace_resume_spec / trace_resume_q: the one postponed leaf (kRefNone = none)
#endif
    uint32_t flags;                    // kRayBusy | kRayShadow | kRayExtFollows | kRayOccluded
};
constexpr uint32_t kRayBusy = 1u, kRayShadow = 2u, kRayExtFollows = 4u, kRayInLeaf = 32u, kRayOccluded = 64u;     // (8u, 16u and kRayInLeaf: pt_trace_experimental.h)

template <bool COUNT, int N>
PT_DEV void ray_start(const DeviceScene& S, Stack<N>& st, RayState& r, bool hasShadow, V3 so, V3 sd, float smaxt,
                      bool hasExt, V3 eo, V3 ed, V3& thr, Hit& h, Ctr& c) {
    h.tri = -1; h.t = 3.402823466e+38f; h.u = 0.0f; h.v = 0.0f; h.material = 0;      // h.t: the running minimum (BVHSceneIntersect's min_t) until a triangle sets it
    thr = v3(1.0f);
    if (COUNT) { if (hasShadow) c.raysShadow++; if (hasExt) c.raysClosest++; }
    r.o = hasShadow ? so : eo; r.d = hasShadow ? sd : ed;
    r.max_t = hasShadow ? smaxt : 999999.0f;
    r.cur = S.rootRef;
#if PT_RS_KEEP_INV && !defined(PT_EXPERIMENTAL)
    r.inv = inv3(r.d);
#endif
#ifdef PT_EXPERIMENTAL
    r.inv = inv3(r.d);
    r.min_t = 3.402823466e+38f;
    r.pend = kRefNone;
#endif
    r.flags = kRayBusy | (hasShadow ? kRayShadow : 0u) | ((hasShadow && hasExt) ? kRayExtFollows : 0u);
    st.sp = 0;
}

// (eo, ed): the lane's extension ray, needed when its shadow ray ends inside this call.
template <bool COUNT, int N, bool ONCHIP, bool NOLEAF = false, bool TRISEL = false>
PT_DEV void trace_resume(const DeviceScene& S, const SceneCache& C, Stack<N>& st, RayState& r, V3 eo, V3 ed, int minBusy,
                         V3& thr, Hit& h, Ctr& c, Keep k = Keep{0, 0}) {
    typedef LoopExit<ONCHIP> X;
    if (!(r.flags & kRayBusy)) return;
#if PT_RS_KEEP_INV
    V3 o = r.o, d = r.d, inv = r.inv;
#else
    V3 o = r.o, d = r.d, inv = inv3(r.d);
#endif
    float max_t = r.max_t;
    int32_t cur = r.cur;
    bool isShadow = (r.flags & kRayShadow) != 0, extFollows = (r.flags & kRayExtFollows) != 0, busy = true;
#if PT_OCCL_BOOL
    bool occl = (r.flags & kRayOccluded) != 0;
#else
    uint32_t occlBits = r.flags & kRayOccluded;                    // a VGPR bit, not a lane mask in an SGPR pair through the loops
#endif
    while (true) {
        // wave-level early exit: the lanes still here keep their state for the next call
        const int active = lanes_here();
        if (active <= minBusy) break;
        const int keepN = (active * k.node) >> 4;
        while (cur >= 0) {
            PT_UTIL_STEP(c, 0);
#ifdef PT_UTIL_DEPTH
            if (COUNT) { c.u[4] += cur < 192; c.u[5] += cur < 704; c.u[6] += cur < 1792; c.u[7] += cur < 8192; }
#endif
            cur = descend<COUNT, N, ONCHIP>(S, C, cur, o, inv, st, c);
            if (X::node && lanes_here() <= keepN) break;
        }
        if (X::node && cur >= 0) continue;
        if (cur == kRefNone) {
            if (isShadow && extFollows) {               // shadow ray done: start this lane's extension ray
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
        int keepT = 0;
        if (X::tri) keepT = (lanes_here() * k.tri) >> 4;
        do {
            TriData q = load_tri<ONCHIP>(S, C, ti);
            idx = f2u(q.e.y);
            if (COUNT) c.tris++;
            PT_UTIL_STEP(c, 2);
            float t, u, v;
            if constexpr (TRISEL && NOLEAF && !COUNT && !PT_OCCL_BOOL) {
                // select form: no divergent branch in the test or in what follows it (moller_trumbore_sel)
                const bool hitOk = moller_trumbore_sel(v3(q.a.x, q.a.y, q.a.z), v3(q.a.w, q.b.x, q.b.y), v3(q.b.z, q.b.w, q.e.x), o, d, t, u, v) & (t < max_t);
                const bool sh = isShadow & hitOk, upd = !isShadow & hitOk & (t < h.t);
                occlBits |= sh ? kRayOccluded : 0u;
                occluded = sh;
                h.t = upd ? t : h.t; h.u = upd ? u : h.u; h.v = upd ? v : h.v;
                h.tri = upd ? (int32_t)(idx & 0x7fffffffu) : h.tri;
                h.material = upd ? f2i(q.e.z) : h.material;
                ti++;
                more = !(idx & 0x80000000u) & !occluded;
                if (X::tri && more && lanes_here() <= keepT) break;
                continue;
            }
            bool ok = ((PT_MT_SEL_RESUME && !ONCHIP && !COUNT) ? moller_trumbore_sel : moller_trumbore)(v3(q.a.x, q.a.y, q.a.z), v3(q.a.w, q.b.x, q.b.y), v3(q.b.z, q.b.w, q.e.x), o, d, t, u, v);
            if (isShadow) {
                if (ok && (t < max_t)) {
                    uint32_t flags = f2u(q.e.w);
#if PT_OCCL_BOOL
                    if (NOLEAF) { occl = true; occluded = true; break; }             // the caller reads kRayOccluded; thr is not touched
#else
                    if (NOLEAF) { occlBits = kRayOccluded; occluded = true; break; }
#endif
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
            } else if (ok && (t < h.t) && (t < max_t)) {                              // h.t is min_t (ray_start)
                h.t = t; h.u = u; h.v = v;
                h.tri = (int32_t)(idx & 0x7fffffffu);
                h.material = f2i(q.e.z);
            }
            ti++;
            more = !(idx & 0x80000000u);
            if (X::tri && more && lanes_here() <= keepT) break;
        } while (more);
        if (X::tri && !occluded && more) { cur = ~ti; continue; }                 // the rest of this leaf next time round
        if (COUNT) c.pops++;
        cur = (!occluded && st.sp > 0) ? st.template pop<ONCHIP>() : kRefNone;     // an occluded shadow ray ends here (BVHShadowRay returns)
    }
    r.o = o; r.d = d; r.max_t = max_t; r.cur = cur;
#if PT_RS_KEEP_INV
    r.inv = inv;
#endif
#if PT_OCCL_BOOL
    r.flags = (busy ? kRayBusy : 0u) | (isShadow ? kRayShadow : 0u) | (extFollows ? kRayExtFollows : 0u) | (occl ? kRayOccluded : 0u);
#else
    r.flags = (busy ? kRayBusy : 0u) | (isShadow ? kRayShadow : 0u) | (extFollows ? kRayExtFollows : 0u) | occlBits;
#endif
    if (COUNT) { if (!busy && !isShadow && h.tri >= 0) c.hits++; }
}

}  // namespace pt

// A/B variants that lost their measurements: built only with -DPT_EXPERIMENTAL (make EXPERIMENTAL=1).
#ifdef PT_EXPERIMENTAL
#include "pt_trace_experimental.h"
#endif
