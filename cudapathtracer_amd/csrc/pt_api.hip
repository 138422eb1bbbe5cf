// pt_api.hip — implementation of the C ABI in include/pt_api.h: scene re-pack + upload, the
// launcher boundary (launch_unidirectional / launch_naive_unidirectional, deviceCode.cuh:8-12)
// and the probe entry points. Host code only; the kernels live in pt_kernels.hip.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pt_api.h"
#include "pt_params.h"
#include "xorwow_host.h"

using namespace pt;

static_assert(sizeof(pt_bvh_node) == 48 && sizeof(pt_triangle) == 80 && sizeof(pt_material) == 176 && sizeof(pt_camera) == 112,
              "boundary structs must keep the reference's CUDA layouts (SURVEY.md Appendix A)");
static_assert(offsetof(pt_triangle, emission) == 48 && offsetof(pt_triangle, lightInd) == 64, "Triangle layout");
static_assert(offsetof(pt_material, type) == 32 && offsetof(pt_material, albedo) == 48 && offsetof(pt_material, eta) == 80 &&
              offsetof(pt_material, ior) == 112 && offsetof(pt_material, isSpecular) == 128 && offsetof(pt_material, absorption) == 144 &&
              offsetof(pt_material, priority) == 160, "Material layout");
static_assert(offsetof(pt_camera, forward) == 64 && offsetof(pt_camera, fovScale) == 44, "Camera layout");

// ---- errors ---------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_OK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(-2, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    int ensure(size_t n) {
        if (n <= bytes) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) return fail(-2, "hipMalloc(%zu) failed: %s", n, hipGetErrorString(e));
        bytes = n;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

struct pt_scene {
    int device = 0;
    DevBuf nodes, tris, attrs, lights, mats, textures, jump, totals, leaves, wnodes, qnodes, leafBox, mids;
    bool compactTried = false, compactOk = false, compactWanted = false;
    int wfWideWg = 1;                                                                 // wavefront trace kernel: 16-wave workgroups for scenes in HBM ("wf_wide_wg")   // 32-byte quantised nodes of trace_resume_q ("compact" 1)
    int nWide = 0, wideStackNeed = 0; bool wideTried = false, wideWanted = false;   // the 4-wide collapsed tree of trace_resume_w4 ("wide" 1: opt-in, measured slower)
    int nLeaves = 0;                                  // FLAT scenes: the leaf table (pt_trace.h: visited(leaf) == slab(leaf's own box))
    DevBuf rng, spill, tilebuf, colors, pixcnt, queue, left; // work buffers, grown on demand
    DevBuf wfState, wfCtl, wfCtr, wfSpill;            // wavefront variant
    int variant = 0;                                  // 0 megakernel, 1 wavefront (pt_set_variant)
    int numCU = 256;
    DeviceScene ds{};
    int stackNeed = 0, nInternal = 0, cacheNodes = 0, cacheTris = 0;
    int nMats = 0, nLightsPacked = 0;
    bool armless = false;        // a triangle uses a material type without a dispatch arm (pt_path.h): DEFER is not exact
    // Everything below is set per scene through pt_set_option (names in quotes); the library reads no environment variable.
    int schedMask = 31;          // "sched_mask": scheduling checks every schedMask + 1 bounce iterations (tests use 3)
    bool sliceAlways = true;     // "slice_always" 0: slices only once no fresh tile is left
    bool wavesHbmOk = PT_WAVES_HBM > 0;   // "waves_hbm" 0: scenes in HBM use the 4-waves-per-SIMD kernel too (A/B)
    int nodeKeep = 10, triKeep = 8;        // "node_keep" / "tri_keep" (pt_trace.h: LoopExit)
    int spec = 2;                          // -DPT_SPEC=1 builds only (A/B): speculative descent for shadow rays too (2) or closest-hit rays only (1)
    int refill = 1, refillKeep = 6;       // "refill" / "refill_keep": REFILL instantiation of the kernel for scenes in HBM (pt_trace.h: trace_resume)
    bool refillKeepSet = false;           // (unset: the 4-wave kernel of small shares uses kRefillKeepSmall — it is chain-bound and wants its lanes back sooner)
    bool cull = false;                    // pt_set_culling / "culling": opt-in, not parity-exact by construction
    bool leafBoxes = true;                        // "leaf_boxes" 0: the FLAT kernels walk the nodes in lockstep instead of testing the leaves' own boxes (A/B)
    int flat2Wanted = 1; int lastLaunchFlat2 = 0;   // "flat2" 1 (default): SIMPLE FLAT scenes trace shadow + extension ray in one FLAT pass (DEFER logic step)
    bool noLeafTris = false;                      // no triangle carries a MAT_LEAF material: a shadow ray is occluded by any hit (order-free)
    int queueTimeoutMs = 30000;                   // "queue_timeout_ms": how long a wait on the tile queue may see no progress (tests use 0-1 to exercise the give-up path)
    int queueStalls = 0;                          // launches whose waiters gave up (q[3] != 0) although every tile was finished: not an error, counted (pt_queue_stalls)
    bool leanOk = false, leanWanted = true;       // scene qualifies for the LEAN generic bounce (no MAT_LEAF triangle, no texture / transmission map on any triangle's material) / "lean" 0 turns it off (A/B)
    int lastLaunchLean = 0;
    bool simpleOk = false, simpleWanted = true;   // scene qualifies for the SIMPLE bounce (diffuse-only, pt_path.h) / "simple" 0 turns it off (A/B)
    bool flatOk = false; int flatWanted = 1;      // "flat": 0 off, 1 (or 2) on: scenes of at most 128 nodes / triangles (64- or 128-bit masks)   // scene qualifies for the FLAT kernels (checked in repack) / "flat" 0 turns them off (A/B)
    int lastLaunchFlat = 0, lastLaunchSimple = 0, lastLaunchLeafTable = 0;
    bool lastLaunchQueued = false;        // the last megakernel launch used the tile queue (only then is its error word that launch's)
    int lastLaunchTiles = 0;              // ... and held this many tiles (pt_last_tile_handovers: pushes beyond them are hand-overs)
    int lastLaunchRefill = 0;             // ... and whether it was a REFILL instantiation
    int lastLaunchHbm = -1;               // which megakernel the last launch used (-1: none yet)
    bool wavesHbmForce = false;           // "waves_hbm" 2: ... and the 6-wave kernel whatever the tile count (tests)
    bool onchipOk = true;        // "onchip" 0: never pick the LDS-only kernel instantiation (A/B)
    int nTrisPacked = 0;
    int sliceIters = 512;        // "slice_iters": time slice of the tile queue (0 = off)
    int lptPrio = 2;             // "lpt_prio": 0 no issue-priority steering, 1 once no fresh tile is left, 2 always (A/B)
    bool persistent = true;      // "persistent" 0: one tile per wave, workgroups launched per 4 tiles (A/B)
    bool xcdBands = false;       // "xcd_bands" 1: one contiguous band of tiles per XCD (A/B; loses to interleaving)
    bool deferShadow = false;    // "defer_shadow" 1: megakernel traces shadow + extension ray as a pair (A/B; slower)
    float lastKernelMs = 0.0f;
    bool evPending = false;                            // ev0/ev1 recorded, elapsed time not read yet
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

static int queue_error(pt_scene* s);

// LDS-resident instantiation: every PNode and PTri in the scene cache, the tree no deeper than the LDS stack, and the
// records the bounce reads (PAttr, PMat, PLight) within their own LDS budget.
// Its workgroups hold ONE copy of the scene each, and a CU has 160 KB of LDS for its 16 waves: workgroups of 4, 8 or 16
// waves get 1, 2 or 4 times the budgets kCacheBytes (PNodes + PTris) and kAttrCacheBytes. Returns the smallest workgroup
// size (in waves) that holds the scene, 0 if none does.
static int scene_onchip_wg(const pt_scene* s) {
    if (!s->onchipOk || s->nTrisPacked <= 0 || s->ds.stackSpill != 0) return 0;
    const size_t geom = (size_t)s->nInternal * 64 + (size_t)s->nTrisPacked * 48, rec = attr_cache_bytes(s->nTrisPacked, s->nMats, s->nLightsPacked);
    // ... and 16 / wg such workgroups must fit a CU's 160 KB with the largest per-wave area any LDS-resident kernel uses (the pair
    // pass scratch, 24 x 256 B, plus the medium stack of the non-SIMPLE kernels): a Cornell box of glass, water and gold at four
    // waves per workgroup was 1 KB over — three workgroups per CU instead of four, 19 % of the frame (round 3)
    const size_t perWave = (size_t)kStackFlat2Rows * 256 + (s->simpleOk && s->simpleWanted ? 0 : (size_t)kMediumMax * 64);
    for (int wg = 4; wg <= 16; wg *= 2)
        if (geom <= (size_t)kCacheBytes * (wg / 4) && (kAttrCacheBytes == 0 || rec <= (size_t)kAttrCacheBytes * (wg / 4)) &&
            geom + rec + (size_t)wg * perWave <= (size_t)160 * 1024 * wg / 16) return wg;
    return 0;
}
static bool scene_onchip(const pt_scene* s) { return scene_onchip_wg(s) > 0; }

extern "C" {

int pt_fail_(int code, const char* msg) { g_err = msg; return code; }     // for the other translation units

int pt_api_version(void) { return PT_API_VERSION; }
const char* pt_last_error(void) { return g_err.c_str(); }

int pt_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(-2, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

void pt_scene_destroy(pt_scene* s) {
    if (!s) return;
    DevBuf* all[] = {&s->nodes, &s->tris, &s->attrs, &s->lights, &s->mats, &s->textures, &s->jump, &s->totals, &s->leaves, &s->wnodes, &s->qnodes, &s->leafBox, &s->mids,
                     &s->rng, &s->spill, &s->tilebuf, &s->colors, &s->pixcnt, &s->queue, &s->left, &s->wfState, &s->wfCtl, &s->wfCtr, &s->wfSpill};
    for (DevBuf* b : all) b->release();
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    delete s;
}

static int upload(DevBuf& b, const void* src, size_t bytes) {
    if (int r = b.ensure(std::max<size_t>(bytes, 16))) return r;
    if (bytes) HIP_OK(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return 0;
}

// The leaf table of the FLAT kernels (and the opt-in trees of EXPERIMENTAL builds) replace the reference's root-to-leaf box
// walk by a test of each leaf's OWN box. That equals the reference's visiting set only if every box contains the boxes
// below it and no box holds a NaN or an infinity (pt_trace.h: inv_is_regular and the monotonicity argument above it). The
// reference's builder nests exactly (a parent's box is the float-wise min / max of its children's, main.cu:20-233), but
// pt_scene_create also takes a caller's own BVH arrays — a refit, loose or corrupt tree must fall back to the walk that
// tests the boxes the caller gave. pn: packed internal nodes, children numbered after their parents.
static bool boxes_nested_and_finite(const std::vector<PNode>& pn, int nInternal) {
    for (int i = 0; i < nInternal; i++) {
        const PNode& p = pn[i];
        for (int a = 0; a < 3; a++)
            if (!(std::isfinite(p.lmin[a]) && std::isfinite(p.lmax[a]) && std::isfinite(p.rmin[a]) && std::isfinite(p.rmax[a]))) return false;
        const int32_t ref[2] = {p.left, p.right};
        for (int k = 0; k < 2; k++) {
            if (ref[k] < 0) continue;                              // a leaf child: its box IS the one the table tests
            if (ref[k] <= i || ref[k] >= nInternal) return false;
            const PNode& c = pn[ref[k]];
            const float* mn = k == 0 ? p.lmin : p.rmin; const float* mx = k == 0 ? p.lmax : p.rmax;
            for (int a = 0; a < 3; a++)
                if (!(mn[a] <= c.lmin[a] && mn[a] <= c.rmin[a] && mx[a] >= c.lmax[a] && mx[a] >= c.rmax[a])) return false;
        }
    }
    return true;
}

// Re-pack the reference's data model for gfx950.
extern "C" int pt_bvh_build_pack_(const pt_scene_desc* d, const int* mat_types, int max_leaf_size, void* d_nodes, void* d_tris, void* d_attrs,
                                  int* out5, pt_bvh_build_stats* stats);

// deviceLeaf < 0: the caller's BVH, packed on the host. deviceLeaf >= 0 (pt_scene_create_from_mesh): the tree is built AND
// packed on the device (pt_bvh_build.hip) with that leaf size; d->bvh / d->bvh_indices are not read.
static int repack(pt_scene* s, const pt_scene_desc* d, int deviceLeaf = -1, pt_bvh_build_stats* buildStats = nullptr) {
    const bool onDevice = deviceLeaf >= 0;
    const int nT = d->n_triangles, nN = onDevice ? 1 : d->n_nodes;
    if (nT <= 0 || nN <= 0 || !d->triangles || (!onDevice && (!d->bvh || !d->bvh_indices)) || !d->positions || !d->materials)
        return fail(-1, "pt_scene_create: empty scene (the reference aborts with 'No triangles loaded', main.cu:505-508)");
    if (d->n_materials <= 0 || d->n_materials > 256) return fail(-1, "pt_scene_create: %d materials (1..256 supported)", d->n_materials);

    int nInternal = 0, stackNeed = 0, rootRef = 0;
    std::vector<PNode> nodes; std::vector<PTri> tris; std::vector<PAttr> attrs;
    auto pos = [&](int i, const char* what, int tri, bool& ok) -> pt_float4 {
        if (i < 0 || i >= d->n_positions) { ok = false; fail(-1, "pt_scene_create: triangle %d %s index %d out of range", tri, what, i); return pt_float4{0, 0, 0, 0}; }
        return d->positions[i];
    };
    if (!onDevice) {
    // --- internal nodes renumbered BREADTH-FIRST from the root, so PNodes [0, K) are the top of the
    //     tree (the part every ray visits) and can be staged in LDS as one contiguous block ---
    std::vector<int> internalId(nN, -1);
    for (int i = 0; i < nN; i++) {
        const pt_bvh_node& n = d->bvh[i];
        if (n.primCount > 0) {
            if (n.first < 0 || n.first + n.primCount > nT) return fail(-1, "pt_scene_create: leaf %d range [%d,+%d) out of bounds", i, n.first, n.primCount);
        } else {
            if (n.left < 0 || n.right < 0 || n.left >= nN || n.right >= nN) return fail(-1, "pt_scene_create: internal node %d has child out of range", i);
        }
    }
    {
        std::vector<int> queue;
        if (d->bvh[0].primCount <= 0) { queue.push_back(0); internalId[0] = nInternal++; }
        for (size_t q = 0; q < queue.size(); q++) {
            const pt_bvh_node& n = d->bvh[queue[q]];
            for (int c : {n.left, n.right}) {
                if (d->bvh[c].primCount > 0) continue;
                if (internalId[c] >= 0 || (int)queue.size() >= nN) return fail(-1, "pt_scene_create: BVH is not a tree");
                internalId[c] = nInternal++;
                queue.push_back(c);
            }
        }
    }
#ifndef PT_NODE_ORDER_AREA
#define PT_NODE_ORDER_AREA 1        // 0: plain breadth-first numbering (A/B: 82 k triangles 765 -> 729 ms at 128 spp, 263 k 387 -> 379 ms at 16 spp, 7.5 % fewer node fetches from global memory; profiles/r03_ab_node_order_area.log)
#endif
#if PT_NODE_ORDER_AREA
    // Scenes in HBM: the kernels keep PNodes [0, K) in LDS, so number the internal nodes by how often rays visit them rather
    // than by level — a ray enters a box with a probability proportional to its surface area (the SAH's own estimate), and a
    // child's box lies inside its parent's, so descending area (ties: breadth-first order) still numbers parents first.
    if (nInternal > 128) {
        std::vector<float> area((size_t)nInternal, 0.0f);
        for (int i = 0; i < nN; i++) {
            if (internalId[i] < 0) continue;
            const pt_bvh_node& n = d->bvh[i];
            const float dx = n.aabbMAX.x - n.aabbMIN.x, dy = n.aabbMAX.y - n.aabbMIN.y, dz = n.aabbMAX.z - n.aabbMIN.z;
            const float a = dx * dy + dy * dz + dz * dx;
            area[internalId[i]] = std::isfinite(a) ? a : 0.0f;
        }
        std::vector<int> order((size_t)nInternal);
        for (int i = 0; i < nInternal; i++) order[i] = i;
        std::stable_sort(order.begin() + 1, order.end(), [&](int a, int b) { return area[a] > area[b]; });     // the root stays node 0
        std::vector<int> rank((size_t)nInternal);
        for (int k = 0; k < nInternal; k++) rank[order[k]] = k;
        for (int i = 0; i < nN; i++) if (internalId[i] >= 0) internalId[i] = rank[internalId[i]];
    }
#endif
    auto childRef = [&](int c) -> int32_t { return d->bvh[c].primCount > 0 ? ~d->bvh[c].first : internalId[c]; };
    nodes.assign(std::max(nInternal, 1), PNode{});
    std::vector<uint8_t> leafEnd(nT, 0);
    for (int i = 0; i < nN; i++) {
        const pt_bvh_node& n = d->bvh[i];
        if (n.primCount > 0) { leafEnd[n.first + n.primCount - 1] = 1; continue; }
        if (internalId[i] < 0) continue;                       // unreachable from the root
        PNode& p = nodes[internalId[i]];
        const pt_bvh_node& L = d->bvh[n.left];
        const pt_bvh_node& R = d->bvh[n.right];
        p.lmin[0] = L.aabbMIN.x; p.lmin[1] = L.aabbMIN.y; p.lmin[2] = L.aabbMIN.z;
        p.lmax[0] = L.aabbMAX.x; p.lmax[1] = L.aabbMAX.y; p.lmax[2] = L.aabbMAX.z;
        p.rmin[0] = R.aabbMIN.x; p.rmin[1] = R.aabbMIN.y; p.rmin[2] = R.aabbMIN.z;
        p.rmax[0] = R.aabbMAX.x; p.rmax[1] = R.aabbMAX.y; p.rmax[2] = R.aabbMAX.z;
        p.left = childRef(n.left); p.right = childRef(n.right); p.pad0 = p.pad1 = 0;
    }
    // stack need = the largest number of internal nodes on a root-to-leaf path (each can leave one
    // far child pending); also rejects cycles.
    {
        std::vector<std::pair<int, int>> st; st.push_back({0, 1});
        size_t visited = 0;
        while (!st.empty()) {
            auto [i, depth] = st.back(); st.pop_back();
            if (++visited > (size_t)nN) return fail(-1, "pt_scene_create: BVH is not a tree");
            if (d->bvh[i].primCount > 0) continue;
            stackNeed = std::max(stackNeed, depth);
            st.push_back({d->bvh[i].left, depth + 1});
            st.push_back({d->bvh[i].right, depth + 1});
        }
    }
    if (stackNeed > 128) return fail(-1, "pt_scene_create: BVH depth %d exceeds the reference's nodeStack[128] (integratorUtilities.cuh:89)", stackNeed);

    // --- triangles in leaf order ---
    tris.resize(nT);
    for (int i = 0; i < nT; i++) {
        int idx = d->bvh_indices[i];
        if (idx < 0 || idx >= nT) return fail(-1, "pt_scene_create: BVHindices[%d] = %d out of range", i, idx);
        const pt_triangle& t = d->triangles[idx];
        bool ok = true;
        pt_float4 a = pos(t.aInd, "a", idx, ok), b = pos(t.bInd, "b", idx, ok), c = pos(t.cInd, "c", idx, ok);
        if (!ok) return -1;
        if (t.materialID < 0 || t.materialID >= d->n_materials) return fail(-1, "pt_scene_create: triangle %d material %d out of range", idx, t.materialID);
        PTri& p = tris[i];
        p.v0[0] = a.x; p.v0[1] = a.y; p.v0[2] = a.z;
        p.e1[0] = b.x - a.x; p.e1[1] = b.y - a.y; p.e1[2] = b.z - a.z;      // trib - tria, integratorUtilities.cuh:13
        p.e2[0] = c.x - a.x; p.e2[1] = c.y - a.y; p.e2[2] = c.z - a.z;      // tric - tria, :14
        p.idx = (uint32_t)idx | (leafEnd[i] ? 0x80000000u : 0u);
        p.material = t.materialID;
        p.flags = d->materials[t.materialID].type == PT_MAT_LEAF ? 1u : 0u;
        {
            const int ty = d->materials[t.materialID].type;
            if (!(ty == PT_MAT_DIFFUSE || ty == PT_MAT_METAL || ty == PT_MAT_SMOOTHDIELECTRIC || ty == PT_MAT_LEAF || ty == PT_MAT_DELTAMIRROR)) s->armless = true;
        }
    }
    // --- hit attributes by original index ---
    attrs.resize(nT);
    for (int i = 0; i < nT; i++) {
        const pt_triangle& t = d->triangles[i];
        PAttr& a = attrs[i];
        const int ni[3] = {t.naInd, t.nbInd, t.ncInd}, ui[3] = {t.uvaInd, t.uvbInd, t.uvcInd};
        float* nd[3] = {a.n0, a.n1, a.n2}; float* ud[3] = {a.uv0, a.uv1, a.uv2};
        for (int k = 0; k < 3; k++) {
            if (ni[k] < 0 || ni[k] >= d->n_normals || !d->normals) return fail(-1, "pt_scene_create: triangle %d normal index %d out of range (faces without vn must be given a normal by the loader)", i, ni[k]);
            if (ui[k] < 0 || ui[k] >= d->n_uvs || !d->uvs) return fail(-1, "pt_scene_create: triangle %d uv index %d out of range", i, ui[k]);
            nd[k][0] = d->normals[ni[k]].x; nd[k][1] = d->normals[ni[k]].y; nd[k][2] = d->normals[ni[k]].z;
            ud[k][0] = d->uvs[ui[k]].x; ud[k][1] = d->uvs[ui[k]].y;
        }
        a.emission[0] = t.emission.x; a.emission[1] = t.emission.y; a.emission[2] = t.emission.z;
        a.material = t.materialID;
        a.lightInd = (t.lightInd >= 0 && t.lightInd < d->n_lights) ? t.lightInd : -51;
    }
    rootRef = childRef(0);
    }
    // --- lights ---
    std::vector<PLight> lights(std::max(d->n_lights, 1));
    std::memset(lights.data(), 0, lights.size() * sizeof(PLight));
    for (int i = 0; i < d->n_lights; i++) {
        const pt_triangle& t = d->lights[i];
        bool ok = true;
        pt_float4 a = pos(t.aInd, "a", i, ok), b = pos(t.bInd, "b", i, ok), c = pos(t.cInd, "c", i, ok);
        if (!ok) return -1;
        if (t.naInd < 0 || t.naInd >= d->n_normals) return fail(-1, "pt_scene_create: light %d normal index out of range", i);
        PLight& L = lights[i];
        L.a[0] = a.x; L.a[1] = a.y; L.a[2] = a.z; L.b[0] = b.x; L.b[1] = b.y; L.b[2] = b.z; L.c[0] = c.x; L.c[1] = c.y; L.c[2] = c.z;
        L.na[0] = d->normals[t.naInd].x; L.na[1] = d->normals[t.naInd].y; L.na[2] = d->normals[t.naInd].z;
        L.emission[0] = t.emission.x; L.emission[1] = t.emission.y; L.emission[2] = t.emission.z;
        // area = 0.5f * length(cross3(b - a, c - a)) exactly as the kernels used to evaluate it per NEE sample: IEEE
        // subtractions, cross = fma(p, q, -(r * s)), dot = fma(z, z', fma(y, y', x * x')), correctly rounded sqrt
        const float ux = b.x - a.x, uy = b.y - a.y, uz = b.z - a.z, vx = c.x - a.x, vy = c.y - a.y, vz = c.z - a.z;
        const float cx = fmaf(uy, vz, -(uz * vy)), cy = fmaf(uz, vx, -(ux * vz)), cz = fmaf(ux, vy, -(uy * vx));
        L.area = 0.5f * sqrtf(fmaf(cz, cz, fmaf(cy, cy, cx * cx)));
    }
    // --- materials ---
    std::vector<PMat> mats(d->n_materials);
    std::memset(mats.data(), 0, mats.size() * sizeof(PMat));
    for (int i = 0; i < d->n_materials; i++) {
        const pt_material& m = d->materials[i];
        PMat& p = mats[i];
        p.type = m.type;
        p.flags = (m.hasTexture ? kMatHasTexture : 0) | (m.hasTransMap ? kMatHasTransMap : 0) | (m.isSpecular ? kMatSpecular : 0) | (m.boundary ? kMatBoundary : 0);
        p.priority = m.priority;
        p.texStart = m.startInd; p.texW = m.width; p.texH = m.height;
        if ((m.hasTexture || m.hasTransMap) && m.width > 0 && m.height > 0 &&
            (m.startInd < 0 || (long long)m.startInd + (long long)m.width * m.height > d->n_texels))
            return fail(-1, "pt_scene_create: material %d texture window exceeds the texel array", i);
        p.roughness = m.roughness; p.ior = m.ior; p.transmission = m.transmission;
        p.albedo[0] = m.albedo.x; p.albedo[1] = m.albedo.y; p.albedo[2] = m.albedo.z;
        p.eta[0] = m.eta.x; p.eta[1] = m.eta.y; p.eta[2] = m.eta.z;
        p.k[0] = m.k.x; p.k[1] = m.k.y; p.k[2] = m.k.z;
        p.absorption[0] = m.absorption.x; p.absorption[1] = m.absorption.y; p.absorption[2] = m.absorption.z;
        p.albedoOverPi[0] = m.albedo.x / kPi; p.albedoOverPi[1] = m.albedo.y / kPi; p.albedoOverPi[2] = m.albedo.z / kPi;   // cosine_f, reflectors.cuh:10-13
    }

    // SIMPLE scenes (pt_path.h): every triangle's material an untextured MAT_DIFFUSE that is neither boundary nor specular,
    // in air (material 0) that does not absorb — then the medium stack never changes and the dispatchers have one arm.
    {
        bool simple = d->materials[0].absorption.x == 0.0f && d->materials[0].absorption.y == 0.0f && d->materials[0].absorption.z == 0.0f;
        for (int i = 0; i < nT && simple; i++) {
            const int id = d->triangles[i].materialID;
            if (id < 0 || id >= d->n_materials) { simple = false; break; }
            const pt_material& m = d->materials[id];
            simple = m.type == PT_MAT_DIFFUSE && !m.hasTexture && !m.hasTransMap && !m.boundary && !m.isSpecular;
        }
        s->simpleOk = simple;
        bool noLeaf = true;
        for (int i = 0; i < nT && noLeaf; i++) {
            const int id = d->triangles[i].materialID;
            noLeaf = id >= 0 && id < d->n_materials && d->materials[id].type != PT_MAT_LEAF;
        }
        s->noLeafTris = noLeaf;
        bool lean = noLeaf;                       // LEAN (pt_shade.h): additionally no triangle's material samples a texture
        for (int i = 0; i < nT && lean; i++) {
            const pt_material& m = d->materials[d->triangles[i].materialID];
            lean = !m.hasTexture && !m.hasTransMap;
        }
        s->leanOk = lean;
    }
    if (!onDevice) {
        if (int r = upload(s->nodes, nodes.data(), nodes.size() * sizeof(PNode))) return r;
        if (int r = upload(s->tris, tris.data(), tris.size() * sizeof(PTri))) return r;
        if (int r = upload(s->attrs, attrs.data(), attrs.size() * sizeof(PAttr))) return r;
    } else {
        if (int r = s->nodes.ensure((size_t)nT * sizeof(PNode))) return r;
        if (int r = s->tris.ensure((size_t)nT * sizeof(PTri))) return r;
        if (int r = s->attrs.ensure((size_t)nT * sizeof(PAttr))) return r;
        std::vector<int> types(d->n_materials);
        for (int i = 0; i < d->n_materials; i++) types[i] = d->materials[i].type;
        int out5[5] = {0, 0, 0, 0, 0};
        if (int r = pt_bvh_build_pack_(d, types.data(), deviceLeaf, s->nodes.p, s->tris.p, s->attrs.p, out5, buildStats)) return r;
        nInternal = out5[0]; stackNeed = out5[1]; rootRef = out5[2];
        if (out5[3]) s->armless = true;
        if (stackNeed > 128) return fail(-1, "pt_scene_create_from_mesh: BVH depth %d exceeds the reference's nodeStack[128] (integratorUtilities.cuh:89)", stackNeed);
#if PT_NODE_ORDER_AREA
        // the same numbering as the host re-pack gives scenes in HBM (above): descending surface area of a node's own box — here
        // read from its parent's record, the same floats — so that the LDS copy of PNodes [0, K) holds the most-visited ones
        if (nInternal > 128 && rootRef == 0) {
            std::vector<PNode> pn((size_t)nInternal);
            HIP_OK(hipMemcpy(pn.data(), s->nodes.p, (size_t)nInternal * sizeof(PNode), hipMemcpyDeviceToHost));
            std::vector<float> area((size_t)nInternal, 0.0f);
            std::vector<int> bfs; bfs.reserve((size_t)nInternal); bfs.push_back(0);
            std::vector<uint8_t> seen((size_t)nInternal, 0); seen[0] = 1;
            bool tree = true;
            for (size_t q = 0; q < bfs.size() && tree; q++) {
                const PNode& p = pn[bfs[q]];
                const int32_t ref[2] = {p.left, p.right};
                for (int k = 0; k < 2; k++) {
                    if (ref[k] < 0) continue;
                    if (ref[k] >= nInternal || seen[ref[k]]) { tree = false; break; }
                    seen[ref[k]] = 1;
                    const float* mn = k == 0 ? p.lmin : p.rmin; const float* mx = k == 0 ? p.lmax : p.rmax;
                    const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
                    const float a = dx * dy + dy * dz + dz * dx;
                    area[ref[k]] = std::isfinite(a) ? a : 0.0f;
                    bfs.push_back(ref[k]);
                }
            }
            if (tree && (int)bfs.size() == nInternal) {
                // bfs[k]: k-th node in breadth-first order; sort those positions by area (stable: ties keep breadth-first order)
                std::vector<int> order(bfs);
                std::stable_sort(order.begin() + 1, order.end(), [&](int a, int b) { return area[a] > area[b]; });
                std::vector<int> rank((size_t)nInternal);
                for (int k = 0; k < nInternal; k++) rank[order[k]] = k;
                std::vector<PNode> out((size_t)nInternal);
                for (int i = 0; i < nInternal; i++) {
                    PNode p = pn[i];
                    if (p.left >= 0) p.left = rank[p.left];
                    if (p.right >= 0) p.right = rank[p.right];
                    out[rank[i]] = p;
                }
                HIP_OK(hipMemcpy(s->nodes.p, out.data(), (size_t)nInternal * sizeof(PNode), hipMemcpyHostToDevice));
            }
        }
#endif
    }
    if (int r = upload(s->lights, lights.data(), lights.size() * sizeof(PLight))) return r;
    if (int r = upload(s->mats, mats.data(), mats.size() * sizeof(PMat))) return r;
    if (int r = upload(s->textures, d->textures, (size_t)std::max(d->n_texels, 0) * sizeof(float4))) return r;
    const std::vector<uint32_t>& jt = xorwow_host::jump_table();
    if (int r = upload(s->jump, jt.data(), jt.size() * sizeof(uint32_t))) return r;
    if (int r = s->totals.ensure(16 * sizeof(unsigned long long))) return r;      // 8 counters + 6 diagnostic stamp sums
    HIP_OK(hipMemset(s->totals.p, 0, 16 * sizeof(unsigned long long)));

    s->nInternal = nInternal;
    s->stackNeed = stackNeed;
    s->ds.nodes = (const PNode*)s->nodes.p; s->ds.tris = (const PTri*)s->tris.p; s->ds.attrs = (const PAttr*)s->attrs.p;
    s->ds.lights = (const PLight*)s->lights.p; s->ds.mats = (const PMat*)s->mats.p; s->ds.textures = (const float4*)s->textures.p;
    s->ds.rootRef = rootRef;
    s->ds.nLights = d->n_lights; s->ds.nTris = nT;
    s->nMats = d->n_materials; s->nLightsPacked = std::max(d->n_lights, 1);
    s->ds.stackSpill = std::max(0, stackNeed - kStackLds);
    // scene cache: everything if it fits the LDS budget, else only the top of the (breadth-first) tree
    s->nTrisPacked = nT;
    if ((size_t)nInternal * 64 + (size_t)nT * 48 <= (size_t)kCacheBytes) { s->cacheNodes = nInternal; s->cacheTris = nT; }
    else { s->cacheNodes = std::min(nInternal, kCacheBytes / 64); s->cacheTris = 0; }
    // FLAT kernels (pt_trace.h): one forward pass over the internal nodes with 64- or 128-bit masks — needs at most 128 of
    // each, every child numbered after its parent (the breadth-first numbering gives that; checked on the packed records)
    // and the triangle count of every leaf child, which goes into the spare words of its PNode.
    s->flatOk = false;
    if (nInternal <= 128 && nT >= 1 && nT <= 128 && scene_onchip_wg(s) > 0) {
        std::vector<PNode> pn((size_t)std::max(nInternal, 1));
        std::vector<PTri> pt((size_t)nT);
        if (nInternal > 0) HIP_OK(hipMemcpy(pn.data(), s->nodes.p, (size_t)nInternal * sizeof(PNode), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(pt.data(), s->tris.p, (size_t)nT * sizeof(PTri), hipMemcpyDeviceToHost));
        auto leafCount = [&](int first) {                      // triangles up to and including the one flagged last
            int k = first;
            while (k < nT) { uint32_t w; memcpy(&w, (const char*)&pt[k] + 36, 4); k++; if (w & 0x80000000u) return k - first; }
            return 0;                                          // runs off the end: not a well-formed leaf
        };
        bool ok = rootRef < 0 ? (~rootRef == 0 && leafCount(0) == nT) : (rootRef == 0);
        std::vector<int> first((size_t)std::max(nInternal, 1), 0);   // first triangle below node i
        for (int i = nInternal - 1; i >= 0 && ok; i--) {          // children come after their parent: bottom-up
            int32_t* ref = &pn[i].left;                         // left, right, pad0, pad1
            int firstOf[2] = {0, 0};
            for (int k = 0; k < 2 && ok; k++) {
                if (ref[k] >= 0) {
                    ok = ref[k] > i && ref[k] < nInternal;
                    if (ok) { ref[2 + k] = pn[ref[k]].pad0 + pn[ref[k]].pad1; firstOf[k] = first[ref[k]]; }
                } else {
                    const int f = ~ref[k]; const int cnt = (ref[k] != kRefNone && f < nT) ? leafCount(f) : 0;
                    ok = cnt >= 1; ref[2 + k] = cnt; firstOf[k] = f;
                }
            }
            // leaf order is left to right: the right subtree's triangles follow the left subtree's
            ok = ok && firstOf[1] == firstOf[0] + ref[2];
            first[i] = firstOf[0];
        }
        if (ok && nInternal > 0) ok = first[0] == 0 && pn[0].pad0 + pn[0].pad1 == nT;
        if (ok && nInternal > 0) HIP_OK(hipMemcpy(s->nodes.p, pn.data(), (size_t)nInternal * sizeof(PNode), hipMemcpyHostToDevice));
        s->flatOk = ok;
        s->nLeaves = 0;
        if (ok && nInternal > 0 && boxes_nested_and_finite(pn, nInternal)) {     // the leaf table: every leaf child's box and triangle range
            std::vector<PLeaf> lf;
            for (int i = 0; i < nInternal; i++) {
                const int32_t* ref = &pn[i].left;
                for (int k = 0; k < 2; k++) {
                    if (ref[k] >= 0) continue;
                    PLeaf L;
                    const float* mn = k == 0 ? pn[i].lmin : pn[i].rmin; const float* mx = k == 0 ? pn[i].lmax : pn[i].rmax;
                    for (int a = 0; a < 3; a++) { L.mn[a] = mn[a]; L.mx[a] = mx[a]; }
                    L.first = ~ref[k]; L.count = ref[2 + k];
                    lf.push_back(L);
                }
            }
            if (int r = upload(s->leaves, lf.data(), lf.size() * sizeof(PLeaf))) return r;
            s->nLeaves = (int)lf.size();
        }
    }
    return 0;
}

static pt_scene* create_scene(const pt_scene_desc* desc, int deviceLeaf, pt_bvh_build_stats* stats) {
    if (!desc) { fail(-1, "pt_scene_create: null desc"); return nullptr; }
    pt_scene* s = new pt_scene();
    if (hipError_t e = hipGetDevice(&s->device); e != hipSuccess) {
        fail(-2, "pt_scene_create: no usable HIP device (%s); the path has no CPU fallback", hipGetErrorString(e));
        delete s;
        return nullptr;
    }
    if (repack(s, desc, deviceLeaf, stats) != 0) { pt_scene_destroy(s); return nullptr; }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, s->device) == hipSuccess && prop.multiProcessorCount > 0) s->numCU = prop.multiProcessorCount;
    }
    if (hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess) { fail(-2, "hipEventCreate failed"); pt_scene_destroy(s); return nullptr; }
    return s;
}

pt_scene* pt_scene_create(const pt_scene_desc* desc) { return create_scene(desc, -1, nullptr); }

pt_scene* pt_scene_create_from_mesh(const pt_scene_desc* desc, int max_leaf_size, pt_bvh_build_stats* stats) {
    if (max_leaf_size < 0) { fail(-1, "pt_scene_create_from_mesh: negative leaf size"); return nullptr; }
    return create_scene(desc, max_leaf_size, stats);
}

// what: 0 PNodes (64 B each), 1 PTris (48 B), 2 PAttrs (80 B). Returns the record count (or < 0); copies
// min(count * size, capacity) bytes. For tests: the device re-layout must equal the host one.
int pt_debug_packed(pt_scene* s, int what, void* dst, size_t capacity) {
    if (!s) return fail(-1, "null scene");
    const void* src = what == 0 ? s->nodes.p : what == 1 ? s->tris.p : what == 2 ? s->attrs.p : nullptr;
    const size_t rec = what == 0 ? sizeof(PNode) : what == 1 ? sizeof(PTri) : sizeof(PAttr);
    const int count = what == 0 ? s->nInternal : s->nTrisPacked;
    if (!src) return fail(-1, "pt_debug_packed: unknown array %d", what);
    const size_t bytes = std::min((size_t)count * rec, capacity);
    if (dst && bytes) HIP_OK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return count;
}

// ---- helpers ----------------------------------------------------------------------------------
static int resolve_tiles(int w, int h, const pt_tile_range* tiles, TileSpan& t) {
    if (w <= 0 || h <= 0) return fail(-1, "bad image size %dx%d", w, h);
    t.tilesX = (w + 7) / 8;
    int total = t.tilesX * ((h + 7) / 8);
    if (!tiles) { t.first = 0; t.stride = 1; t.count = total; return 0; }
    t.first = tiles->first; t.stride = tiles->stride; t.count = tiles->count;
    if (t.count < 0 || t.stride < 1 || t.first < 0) return fail(-1, "bad tile range {first %d, stride %d, count %d}", t.first, t.stride, t.count);
    if (t.count > 0 && (long long)t.first + (long long)(t.count - 1) * t.stride >= total)
        return fail(-1, "tile range {first %d, stride %d, count %d} exceeds the %d tiles of a %dx%d image", t.first, t.stride, t.count, total, w, h);
    return 0;
}

static CamK cam_to_kernel(const pt_camera& c) {
    CamK k;
    k.origin = V3{c.cameraOrigin.x, c.cameraOrigin.y, c.cameraOrigin.z};
    k.forward = V3{c.forward.x, c.forward.y, c.forward.z};
    k.right = V3{c.right.x, c.right.y, c.right.z};
    k.up = V3{c.up.x, c.up.y, c.up.z};
    k.w = c.w; k.h = c.h; k.aperture = c.aperture; k.focalDist = c.focalDist; k.fovScale = c.fovScale; k.jitter = c.antiAliasJitterDist;
    return k;
}

static int check_render_args(pt_scene* s, const pt_camera* cam, int spp, int integrator) {
    if (!s || !cam) return fail(-1, "null scene or camera");
    if (spp < 0) return fail(-1, "negative sample count");
    if (integrator != PT_UNIDIRECTIONAL && integrator != PT_NAIVE_UNIDIRECTIONAL)
        return fail(-3, "integrator %d is out of scope: only UNIDIRECTIONAL (0) and NAIVE_UNIDIRECTIONAL (2) are on this path", integrator);
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != s->device) return fail(-1, "scene lives on HIP device %d but the current device is %d", s->device, dev);
    return 0;
}

// The wavefront variant of render_tiles: logic / trace kernel pairs until no path is alive.
static int render_tiles_wavefront(pt_scene* s, const pt_camera* cam, int w, int h, int spp, int maxDepth, int integrator, int useMIS,
                                  const TileSpan& t, void* d_tiles, uint32_t* d_pixcnt, bool count, hipStream_t stream) {
    if (s->armless) return fail(-3, "the wavefront variant needs every material to have a dispatch arm (pt_path.h); use the megakernel");
    WfParams W;
    W.n = t.count * 64; W.w = w; W.h = h; W.tileFirst = t.first; W.tileStride = t.stride; W.tilesX = t.tilesX;
    W.nodeKeep = s->nodeKeep; W.triKeep = s->triKeep;
    if (int r = s->wfState.ensure(wf_state_bytes(W.n))) return r;
    if (int r = s->wfCtl.ensure(64)) return r;
    wf_carve(W, s->wfState.p);
    W.qctl = (uint32_t*)s->wfCtl.p;
    W.rng = (uint32_t*)s->rng.p; W.out = (float4*)d_tiles;
    W.pathCtr = nullptr;
    if (count) {
        if (int r = s->wfCtr.ensure((size_t)W.n * 8 * sizeof(uint32_t))) return r;
        W.pathCtr = (uint32_t*)s->wfCtr.p;
    }
    // the trace kernel has its own LDS budget (no medium stacks, smaller traversal stack): the whole scene if it fits 12 KB, else
    // the top of the tree — in 16-wave workgroups, two per CU, that share 48 KB of it (`wf_wide_wg` 0: the 4-wave shape)
    int wfNodes, wfTris, wgWaves = 4;
    if ((size_t)s->nInternal * 64 + (size_t)s->ds.nTris * 48 <= (size_t)kWfCacheBytes) { wfNodes = s->nInternal; wfTris = s->ds.nTris; }
    else if (s->wfWideWg == 2 || (s->wfWideWg == 1 && (W.n + 1023) / 1024 >= s->numCU * 2)) { wgWaves = 16; wfNodes = std::min(s->nInternal, (80 * 1024 - 16 * kWfStackLds * 256) / 64); wfTris = 0; }
    else { wfNodes = std::min(s->nInternal, kWfCacheBytes / 64); wfTris = 0; }
    const int blocks = std::max(1, std::min(s->numCU * (32 / wgWaves), (W.n + 64 * wgWaves - 1) / (64 * wgWaves)));
    const int spillPerLane = std::max(0, s->stackNeed - kWfStackLds);
    int32_t* spill = nullptr;
    if (spillPerLane > 0) {
        if (int r = s->wfSpill.ensure((size_t)blocks * wgWaves * spillPerLane * 64 * sizeof(int32_t))) return r;
        spill = (int32_t*)s->wfSpill.p;
    }
    const CamK ck = cam_to_kernel(*cam);
    const bool wfSimple = s->simpleOk && s->simpleWanted && !count;        // the SIMPLE bounce (pt_path.h) in the logic kernel: diffuse-only scenes, timed launches
    s->lastLaunchSimple = wfSimple ? 1 : 0;
    HIP_OK(hipMemsetAsync(W.qctl, 0, 16, stream));
    HIP_OK(hipEventRecord(s->ev0, stream));
    HIP_OK(launch_wf_init(W, spp, stream));
    // Paths end at different iterations; the host polls the queue length every 16 iterations.
    const long long cap = (long long)std::max(spp, 1) * 4200 + 64;
    bool finished = false;
    for (long long it = 0; it < cap; it++) {
        HIP_OK(launch_wf_logic(integrator, count, wfSimple, W, s->ds, ck, maxDepth, useMIS, (int)(it & 1), stream));
        if ((it & 15) == 15) {
            uint32_t queued = 0;
            HIP_OK(hipMemcpyAsync(&queued, W.qctl + (it & 1) * 2, 4, hipMemcpyDeviceToHost, stream));
            HIP_OK(hipStreamSynchronize(stream));
            if (queued == 0) { finished = true; break; }
        }
        HIP_OK(launch_wf_trace(count, wgWaves, blocks, W, s->ds, wfNodes, wfTris, spill, spillPerLane, (int)(it & 1), stream));
    }
    if (!finished) return fail(-4, "wavefront render did not terminate within %lld iterations", cap);
    HIP_OK(launch_wf_finish(W, stream));
    if (count) HIP_OK(launch_wf_counters(W, d_pixcnt, (unsigned long long*)s->totals.p, stream));
    HIP_OK(hipEventRecord(s->ev1, stream));
    s->evPending = true;
    return 0;
}

#ifdef PT_EXPERIMENTAL
// The reference tree collapsed to 4-wide nodes for trace_resume_w4 (pt_trace_experimental.h): same leaves, same float boxes, half the
// levels. Built once per scene from the packed binary records (whoever packed them, host or device), numbered breadth-first
// so that the first K wide nodes are the top of the tree (the LDS scene cache). A slot takes the place of an internal child
// by that child's two children — the child with the largest box first — until the node has four slots or only leaves.
static int ensure_wide(pt_scene* s) {
    if (s->wideTried) return 0;
    s->wideTried = true;
    const int nI = s->nInternal;
    if (nI <= 0 || s->ds.rootRef != 0) return 0;
    std::vector<PNode> pn((size_t)nI);
    HIP_OK(hipMemcpy(pn.data(), s->nodes.p, (size_t)nI * sizeof(PNode), hipMemcpyDeviceToHost));
    if (!boxes_nested_and_finite(pn, nI)) return 0;                    // a caller's loose tree: the reference's walk (nWide stays 0)
    struct Slot { float mn[3], mx[3]; int32_t ref; };
    auto area = [](const Slot& b) { const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2]; return dx * dy + dy * dz + dz * dx; };
    auto children = [&](int32_t node, Slot out[2]) {
        const PNode& p = pn[node];
        for (int a = 0; a < 3; a++) { out[0].mn[a] = p.lmin[a]; out[0].mx[a] = p.lmax[a]; out[1].mn[a] = p.rmin[a]; out[1].mx[a] = p.rmax[a]; }
        out[0].ref = p.left; out[1].ref = p.right;
    };
    std::vector<WNode> wn;
    std::vector<int32_t> binaryOf;                    // wide node -> the binary node it was collapsed from
    std::vector<int> need;                            // stack entries a traversal can hold below this node (filled bottom-up)
    binaryOf.push_back(0);
    for (size_t w = 0; w < binaryOf.size(); w++) {    // breadth-first: a wide node's internal slots are appended as they are met
        Slot sl[4]; int n = 2;
        children(binaryOf[w], sl);
        while (n < 4) {
            int best = -1; float bestA = -1.0f;
            for (int i = 0; i < n; i++) if (sl[i].ref >= 0) { const float a = area(sl[i]); if (a > bestA) { bestA = a; best = i; } }
            if (best < 0) break;
            Slot two[2];
            children(sl[best].ref, two);
            sl[best] = two[0]; sl[n++] = two[1];
        }
        WNode W;
        std::memset(&W, 0, sizeof(W));
        for (int i = 0; i < 4; i++) {
            if (i < n) {
                W.mnx[i] = sl[i].mn[0]; W.mny[i] = sl[i].mn[1]; W.mnz[i] = sl[i].mn[2];
                W.mxx[i] = sl[i].mx[0]; W.mxy[i] = sl[i].mx[1]; W.mxz[i] = sl[i].mx[2];
                if (sl[i].ref >= 0) { W.ref[i] = (int32_t)binaryOf.size(); binaryOf.push_back(sl[i].ref); }
                else W.ref[i] = sl[i].ref;
            } else W.ref[i] = kRefNone;
        }
        W.pad[0] = n;
        wn.push_back(W);
        if (binaryOf.size() > (size_t)nI + 1) return fail(-1, "wide collapse: the packed tree is not a tree");
    }
    need.assign(wn.size(), 0);
    int worst = 0;
    for (int w = (int)wn.size() - 1; w >= 0; w--) {   // children come after their parent: bottom-up
        int below = 0;
        for (int i = 0; i < 4; i++) if (wn[w].ref[i] >= 0) below = std::max(below, need[wn[w].ref[i]]);
        need[w] = (wn[w].pad[0] - 1) + below;         // the other slots wait on the stack while one is descended into
        worst = std::max(worst, need[w]);
    }
#if PT_NODE_ORDER_AREA
    // as for the binary nodes: the LDS copy holds wide nodes [0, K) — number them by the area of their own box (the slot box in
    // their parent), root first, so that K most-visited ones are cached
    if (wn.size() > 128) {
        const int nW = (int)wn.size();
        std::vector<float> area((size_t)nW, 0.0f);
        for (int w = 0; w < nW; w++)
            for (int i = 0; i < 4; i++) {
                const int32_t c = wn[w].ref[i];
                if (c < 0 || c >= nW) continue;
                const float dx = wn[w].mxx[i] - wn[w].mnx[i], dy = wn[w].mxy[i] - wn[w].mny[i], dz = wn[w].mxz[i] - wn[w].mnz[i];
                const float a = dx * dy + dy * dz + dz * dx;
                area[c] = std::isfinite(a) ? a : 0.0f;
            }
        std::vector<int> order((size_t)nW);
        for (int w = 0; w < nW; w++) order[w] = w;
        std::stable_sort(order.begin() + 1, order.end(), [&](int a, int b) { return area[a] > area[b]; });
        std::vector<int> rank((size_t)nW);
        for (int k = 0; k < nW; k++) rank[order[k]] = k;
        std::vector<WNode> out((size_t)nW);
        for (int w = 0; w < nW; w++) {
            WNode x = wn[w];
            for (int i = 0; i < 4; i++) if (x.ref[i] >= 0 && x.ref[i] < nW) x.ref[i] = rank[x.ref[i]];
            out[rank[w]] = x;
        }
        wn.swap(out);
    }
#endif
    if (int r = upload(s->wnodes, wn.data(), wn.size() * sizeof(WNode))) return r;
    s->nWide = (int)wn.size(); s->wideStackNeed = worst;
    return 0;
}

// The compact form of the tree for trace_resume_q (pt_trace.h), built once per scene from the packed binary records: per
// internal node a QNode (both child boxes as 8-bit offsets in the node's own frame, rounded OUTWARD with the very fmaf the
// kernel decodes with), per leaf its exact float box (at the index of its first packed triangle) and per internal node the
// first triangle of its right subtree (the tie rule's walk).
static int ensure_compact(pt_scene* s) {
    if (s->compactTried) return 0;
    s->compactTried = true;
    const int nI = s->nInternal, nT = s->nTrisPacked;
    if (nI <= 0 || s->ds.rootRef != 0 || nI >= (1 << 24) || nT >= (1 << 24)) return 0;
    std::vector<PNode> pn((size_t)nI);
    HIP_OK(hipMemcpy(pn.data(), s->nodes.p, (size_t)nI * sizeof(PNode), hipMemcpyDeviceToHost));
    if (!boxes_nested_and_finite(pn, nI)) return 0;                    // a caller's loose tree: the reference's walk (compactOk stays false)
    std::vector<QNode> qn((size_t)nI);
    std::vector<float> lb((size_t)nT * 8, 0.0f);                   // two 16-byte halves per packed triangle index; filled at the first of each leaf
    std::vector<int32_t> mids((size_t)nI, 0), minFirst((size_t)nI, 0), maxFirst((size_t)nI, 0);
    bool ok = true;
    for (int i = nI - 1; i >= 0 && ok; i--) {                      // children come after their parent (breadth-first numbering): bottom-up
        const PNode& p = pn[i];
        QNode& q = qn[i];
        float org[3], ext = 0.0f;
        for (int a = 0; a < 3 && ok; a++) {
            ok = std::isfinite(p.lmin[a]) && std::isfinite(p.lmax[a]) && std::isfinite(p.rmin[a]) && std::isfinite(p.rmax[a]);
            org[a] = std::min(p.lmin[a], p.rmin[a]);
            ext = std::max(ext, std::max(p.lmax[a], p.rmax[a]) - org[a]);
        }
        if (!ok) break;
        int k = -64;                                               // step 2^k: the smallest with fma(255, 2^k, origin) >= the node's maximum on every axis
        if (ext > 0.0f) { int e; std::frexp(ext / 255.0f, &e); k = std::max(e - 1, -64); }
        for (; k <= 63; k++) {
            const float sc = std::ldexp(1.0f, k);
            bool fits = true;
            for (int a = 0; a < 3; a++) fits = fits && std::fmaf(255.0f, sc, org[a]) >= std::max(p.lmax[a], p.rmax[a]);
            if (fits) break;
        }
        if (k > 63) { ok = false; break; }
        const float sc = std::ldexp(1.0f, k);
        auto down = [&](float x, int a) {                          // largest q with decode(q) <= x
            int v = (int)std::floor(((double)x - org[a]) / sc);
            v = std::min(std::max(v, 0), 255);
            while (v > 0 && std::fmaf((float)v, sc, org[a]) > x) v--;
            return (uint8_t)v;
        };
        auto up = [&](float x, int a) {                            // smallest q with decode(q) >= x
            int v = (int)std::ceil(((double)x - org[a]) / sc);
            v = std::min(std::max(v, 0), 255);
            while (v < 255 && std::fmaf((float)v, sc, org[a]) < x) v++;
            return (uint8_t)v;
        };
        for (int a = 0; a < 3; a++) {
            q.o[a] = org[a];
            q.lmin[a] = down(p.lmin[a], a); q.lmax[a] = up(p.lmax[a], a); q.rmin[a] = down(p.rmin[a], a); q.rmax[a] = up(p.rmax[a], a);
            auto dec = [&](uint8_t v) { return std::fmaf((float)v, sc, org[a]); };
            ok = ok && dec(q.lmin[a]) <= p.lmin[a] && dec(q.lmax[a]) >= p.lmax[a] && dec(q.rmin[a]) <= p.rmin[a] && dec(q.rmax[a]) >= p.rmax[a];
        }
        int mn[2], mx[2];
        uint32_t word[2];
        const int32_t ref[2] = {p.left, p.right};
        for (int c = 0; c < 2 && ok; c++) {
            if (ref[c] >= 0) {
                ok = ref[c] > i && ref[c] < nI;
                if (ok) { mn[c] = minFirst[ref[c]]; mx[c] = maxFirst[ref[c]]; word[c] = (uint32_t)ref[c]; }
            } else {
                const int f = ~ref[c];
                ok = ref[c] != kRefNone && f >= 0 && f < nT;
                if (!ok) break;
                mn[c] = mx[c] = f; word[c] = 0x80000000u | (uint32_t)f;
                const float* bmn = c == 0 ? p.lmin : p.rmin; const float* bmx = c == 0 ? p.lmax : p.rmax;
                float* o = &lb[(size_t)f * 8];
                o[0] = bmn[0]; o[1] = bmn[1]; o[2] = bmn[2]; o[3] = bmx[0]; o[4] = bmx[1]; o[5] = bmx[2];
            }
        }
        if (!ok) break;
        q.left = word[0] | ((uint32_t)(k + 64) << 24); q.right = word[1];
        ok = mx[0] < mn[1];                                        // leaf order is left to right
        minFirst[i] = mn[0]; maxFirst[i] = mx[1]; mids[i] = mn[1];
    }
    if (!ok) return 0;
    if (int r = upload(s->qnodes, qn.data(), qn.size() * sizeof(QNode))) return r;
    if (int r = upload(s->leafBox, lb.data(), lb.size() * sizeof(float))) return r;
    if (int r = upload(s->mids, mids.data(), mids.size() * sizeof(int32_t))) return r;
    s->compactOk = true;
    return 0;
}
#endif  // PT_EXPERIMENTAL

// rng init + megakernel on `stream`; d_tiles holds t.count*64 float4.
static int render_tiles(pt_scene* s, const pt_camera* cam, int w, int h, int spp, int maxDepth, int integrator, int useMIS,
                        uint64_t seed, const TileSpan& t, void* d_tiles, uint32_t* d_pixcnt, bool count, hipStream_t stream, bool continueStreams) {
    if (t.count == 0) return 0;
    // the reference keys the stream by the camera's image size (y*w+x with the launch's w, deviceCode.cu:59)
    if (int r = s->rng.ensure((size_t)t.count * 384 * sizeof(uint32_t))) return r;
    // Which kernel (pt_kernels.hip): the LDS-resident instantiation, or — for a scene in HBM — the 6-waves-per-SIMD
    // one with its shorter LDS stack (so the spill area is laid out for THAT stack length).
    const bool deferred = s->deferShadow && !s->armless;
    const bool onchip = scene_onchip(s) && !deferred;       // (the DEFER A/B instantiation exists for the general 4-wave kernel only)
    // ... and only with enough tiles to fill its 6 waves per SIMD: with fewer (a 1/8 shard of a 1080p frame is 4050
    // tiles for 6144 slots) the extra slots stay empty and the 4-wave kernel's faster waves win (measured: 1/8 shard
    // 176 vs 185 ms, 1/4 shard equal, 1/2 shard 602 vs 518 ms on the 263 k-triangle scene).
    // ... of which the SIMPLE instantiation (diffuse-only scenes; timed launches with the resumable traversal) runs at 8 waves per SIMD
    const bool simpleHbm = !onchip && !deferred && s->wavesHbmOk && s->simpleOk && s->simpleWanted && s->refill && !s->cull && !s->armless && !count;
    const int wavesHbm = simpleHbm ? kWavesHbmSimple : kWavesHbm;
    // (the SIMPLE kernel's eight waves per SIMD pay from ~3/4 of its 8192 slots on — a 1/4 share of a 1080p frame: 201 vs 269 ms on
    //  the 263 k-triangle scene — the generic kernel's six from 5/4 of its 6144, profiles/r02_sched_shards.log)
    const long long slotsHbm = (long long)s->numCU * 4 * wavesHbm;
    const bool hbm = !onchip && !deferred && s->wavesHbmOk &&
                     (s->wavesHbmForce || (simpleHbm ? (long long)t.count * 4 >= slotsHbm * 3 : (long long)t.count * 4 >= slotsHbm * 5));
    bool wide = false, compact = false;
#ifdef PT_EXPERIMENTAL
    if (hbm && simpleHbm && s->wideWanted) {
        if (int r = ensure_wide(s)) return r;
        wide = s->nWide > 0;
    }
    if (hbm && simpleHbm && s->compactWanted && !wide) {
        if (int r = ensure_compact(s)) return r;
        compact = s->compactOk;
    }
#endif
    const int spillEntries = hbm ? std::max(0, (wide ? std::max(s->wideStackNeed, s->stackNeed) : s->stackNeed) - (simpleHbm ? kStackLdsHbm : kStackLdsHbmGen)) : s->ds.stackSpill;
    const int wgWaves = hbm ? (simpleHbm ? kWgWavesHbmSimple : kWgWavesHbm) : (onchip ? scene_onchip_wg(s) : 4);
    int blocks = megakernel_blocks(t.count, wgWaves);
    if (spillEntries > 0)
        if (int r = s->spill.ensure((size_t)blocks * wgWaves * spillEntries * 64 * sizeof(int32_t))) return r;
    // continueStreams: a later chunk of a progressive render keeps the per-pixel XORWOW states the
    // previous chunk stored (the reference reloads / stores them around every sample, deviceCode.cu:294, 541)
    if (!continueStreams) HIP_OK(launch_rng_init((const uint32_t*)s->jump.p, seed, w, h, t, (uint32_t*)s->rng.p, stream));
    if (s->variant == 1) return render_tiles_wavefront(s, cam, w, h, spp, maxDepth, integrator, useMIS, t, d_tiles, d_pixcnt, count, stream);
    KParams P;
    P.S = s->ds;
    P.cam = cam_to_kernel(*cam);
    P.w = w; P.h = h; P.spp = spp; P.maxDepth = maxDepth; P.useMIS = useMIS;
    P.tileFirst = t.first; P.tileStride = t.stride; P.tileCount = t.count; P.tilesX = t.tilesX;
    P.cacheNodes = s->cacheNodes; P.cacheTris = s->cacheTris;
    if (onchip) { P.cacheNodes = s->nInternal; P.cacheTris = s->nTrisPacked; }      // the whole scene, in workgroups large enough to hold it
    P.cacheAttrs = P.cacheMats = P.cacheLights = 0;
    P.nLeaves = 0; P.leaves = nullptr;
    if (onchip && kAttrCacheBytes > 0) { P.cacheAttrs = s->nTrisPacked; P.cacheMats = s->nMats; P.cacheLights = s->nLightsPacked; }   // the bounce's records in LDS as well
    if (onchip && kAttrCacheBytes > 0 && s->flatOk && s->leafBoxes && s->nLeaves > 0) { P.nLeaves = s->nLeaves; P.leaves = (const PLeaf*)s->leaves.p; }
    P.wgWaves = wgWaves;
    if (hbm && s->cacheTris == 0) P.cacheNodes = std::min(s->nInternal, (simpleHbm ? kCacheBytesHbmSimple : kCacheBytesHbm) / 64);     // its workgroups share a larger copy of the top of the tree
    P.gnodeFrom = P.cacheNodes;
    if (count && !onchip && !deferred && s->wavesHbmOk) {
        // a counting launch stands in for the timed launch of the same tiles (bench.py): count a node fetch as "global" against THAT
        // instantiation's LDS copy of the tree top — the SIMPLE production kernel holds 768 nodes, this counting kernel 704 (or 192)
        const bool simpleTimed = s->simpleOk && s->simpleWanted && s->refill && !s->cull && !s->armless;
        const long long slotsTimed = (long long)s->numCU * 4 * (simpleTimed ? kWavesHbmSimple : kWavesHbm);
        const bool hbmTimed = s->wavesHbmForce || (simpleTimed ? (long long)t.count * 4 >= slotsTimed * 3 : (long long)t.count * 4 >= slotsTimed * 5);
        if (s->cacheTris == 0) P.gnodeFrom = hbmTimed ? std::min(s->nInternal, (simpleTimed ? kCacheBytesHbmSimple : kCacheBytesHbm) / 64) : s->cacheNodes;
    }
    P.wide = wide ? 1 : 0; P.wnodes = wide ? (const WNode*)s->wnodes.p : nullptr;
    if (wide) P.cacheNodes = 2 * std::min(s->nWide, kCacheBytesHbmSimple / 128);            // wide nodes, counted in 64-byte halves
    P.compact = compact ? 1 : 0;
    P.qnodes = compact ? (const QNode*)s->qnodes.p : nullptr; P.leafBox = compact ? s->leafBox.p : nullptr; P.mids = compact ? (const int32_t*)s->mids.p : nullptr;
    if (compact) P.cacheNodes = std::min(s->nInternal & ~1, kCacheBytesHbmSimple / 32) / 2;     // compact nodes, two per 64-byte unit
    P.xcdBands = s->xcdBands ? 1 : 0;
    P.S.stackSpill = spillEntries;
    P.cull = (s->cull && hbm) ? 1 : 0;
    P.refill = (s->refill && !deferred && !P.cull && !s->armless && (!onchip || s->refill == 2)) ? 1 : 0;   // 2: also the LDS-resident kernel (A/B)
    P.refillKeep = (hbm || s->refillKeepSet) ? s->refillKeep : kRefillKeepSmall;      // re-swept on the round's kernels: 6 for the 8-wave kernel, 10 for the 4-wave one (1/8 shares +5.5 % / +6 %, profiles/r03_shards_hbm_final.log)
    P.spec = s->spec;
    P.nodeKeep = s->nodeKeep; P.triKeep = s->triKeep;
    P.flat = 0;                                             // 1: FLAT with 64-bit masks, 2: with 128-bit masks
    if (onchip && s->flatOk && s->flatWanted && !deferred && !P.refill) {
        P.flat = (s->nInternal <= 64 && s->nTrisPacked <= 64) ? 1 : 2;     // 65..128: +17-20 % over the stack walk with the leaf-box form (profiles/r02_flat_crossover.jsonl)
    }
    // the instantiations that have the SIMPLE bounce: FLAT, the production kernel for scenes in HBM and its 4-wave form (small shares)
    P.simple = ((P.flat && s->simpleOk && s->simpleWanted) || simpleHbm) ? 1 : 0;
    if (P.flat == 1 && s->noLeafTris && !s->armless && s->flat2Wanted && integrator == PT_UNIDIRECTIONAL && !count && useMIS) P.flat = 3;   // ... and the pair form of FLAT
    // the LEAN generic bounce exists for the three timed kernels generic scenes run: the pair form of FLAT, the kernel for scenes in HBM
    // and its 4-wave form (both REFILL)
    P.lean = (s->leanOk && s->leanWanted && !s->armless && !P.simple && !count && !P.cull && (P.flat == 3 || (!onchip && P.refill))) ? 1 : 0;
    s->lastLaunchLean = P.lean;
    s->lastLaunchRefill = P.refill; s->lastLaunchFlat = (P.flat && !count) ? 1 : 0;
    s->lastLaunchSimple = (P.simple && !count) ? 1 : 0;
    s->lastLaunchFlat2 = (P.flat == 3) ? 1 : 0;
    s->lastLaunchLeafTable = (P.flat && !count && P.nLeaves > 0) ? 1 : 0;
    s->lastLaunchHbm = hbm ? 1 : 0;
    P.onchip = onchip ? 1 : 0;
    P.wavesPerSimd = hbm ? wavesHbm : (PT_MIN_WAVES > 0 ? PT_MIN_WAVES : 4);
    P.hbm = hbm ? 1 : 0;
    P.queue = nullptr; P.queueMask = 0; P.left = nullptr; P.gridBlocks = 0;
    P.lptPrio = s->lptPrio; P.sliceIters = s->sliceIters; P.schedMask = s->schedMask; P.sliceAlways = s->sliceAlways ? 1 : 0;
    P.queueTimeout = (unsigned long long)s->queueTimeoutMs * 100000ull;        // ms -> ticks of the 100 MHz steady counter (hipDeviceAttributeWallClockRate)
    if (s->persistent && !s->xcdBands) {
        int cap = 256;
        while (cap < t.count) cap <<= 1;
        if (int r = s->queue.ensure((size_t)(kQueueHeader + 2 * cap) * sizeof(int))) return r;
        if (int r = s->left.ensure((size_t)t.count * 64 * sizeof(int))) return r;
        P.queue = (int*)s->queue.p; P.queueMask = cap - 1; P.left = (int*)s->left.p;
        P.gridBlocks = s->numCU * P.wavesPerSimd * 4 / wgWaves;      // n waves per SIMD = 4n waves per CU, in workgroups of wgWaves
    }
    s->lastLaunchQueued = P.queue != nullptr;
    s->lastLaunchTiles = t.count;
    P.rng = (uint32_t*)s->rng.p; P.out = (float4*)d_tiles; P.pixCounters = d_pixcnt;
    P.totals = count ? (unsigned long long*)s->totals.p : nullptr;
    P.spill = spillEntries > 0 ? (int32_t*)s->spill.p : nullptr;
    HIP_OK(hipEventRecord(s->ev0, stream));            // HIP events on the launch stream, around the megakernel only
    HIP_OK(launch_megakernel(integrator, count, !(s->deferShadow && !s->armless), P, stream));
    HIP_OK(hipEventRecord(s->ev1, stream));
    s->evPending = true;
    return 0;
}

int pt_render_tiles_device(pt_scene* s, const pt_camera* cam, int w, int h, int spp, int maxDepth, int integrator, int useMIS,
                           uint64_t seed, const pt_tile_range* tiles, void* d_tile_rgba, int count_work, void* stream) {
    if (int r = check_render_args(s, cam, spp, integrator)) return r;
    if (!d_tile_rgba) return fail(-1, "null tile buffer");
    TileSpan t;
    if (int r = resolve_tiles(w, h, tiles, t)) return r;
    return render_tiles(s, cam, w, h, spp, maxDepth, integrator, useMIS, seed, t, d_tile_rgba, nullptr, count_work != 0, (hipStream_t)stream, false);
}

int pt_untile_device(int w, int h, const pt_tile_range* tiles, const void* d_tile_rgba, void* d_colors, void* stream) {
    TileSpan t;
    if (int r = resolve_tiles(w, h, tiles, t)) return r;
    HIP_OK(launch_untile(w, h, t, (const float4*)d_tile_rgba, (float4*)d_colors, (hipStream_t)stream));
    return 0;
}
int pt_tile_device(int w, int h, const pt_tile_range* tiles, const void* d_colors, void* d_tile_rgba, void* stream) {
    TileSpan t;
    if (int r = resolve_tiles(w, h, tiles, t)) return r;
    HIP_OK(launch_tile(w, h, t, (const float4*)d_colors, (float4*)d_tile_rgba, (hipStream_t)stream));
    return 0;
}

// scan-line DEVICE accumulator in / out, blocking; the body of both reference launchers.
static int launch_on_colors(pt_scene* s, const pt_camera* cam, int w, int h, int spp, int maxDepth, int integrator, int useMIS,
                            uint64_t seed, const pt_tile_range* tiles, void* d_colors, uint32_t* d_pixcnt, bool count) {
    if (int r = check_render_args(s, cam, spp, integrator)) return r;
    TileSpan t;
    if (int r = resolve_tiles(w, h, tiles, t)) return r;
    if (int r = s->tilebuf.ensure(std::max<size_t>((size_t)t.count * 64 * sizeof(float4), 16))) return r;
    HIP_OK(launch_tile(w, h, t, (const float4*)d_colors, (float4*)s->tilebuf.p, nullptr));
    if (int r = render_tiles(s, cam, w, h, spp, maxDepth, integrator, useMIS, seed, t, s->tilebuf.p, d_pixcnt, count, nullptr, false)) return r;
    HIP_OK(launch_untile(w, h, t, (const float4*)s->tilebuf.p, (float4*)d_colors, nullptr));
    HIP_OK(hipDeviceSynchronize());                 // cudaDeviceSynchronize, deviceCode.cu:608
    return queue_error(s);                          // a tile-queue timeout means an incomplete frame: never report success
}

// The reference's launchers with their progressive hook (deviceCode.cu:568-606): the sample loop runs
// in chunks of `chunk_spp`; after every chunk `d_colors` holds the sum over `samples_done` samples and
// `progress(samples_done, user)` is called on the calling thread (the reference writes render.bmp /
// renderCSV.csv there every >= 5 s; file I/O stays outside this ABI). The per-pixel streams continue
// across chunks, so the final image is bit-identical to the one-shot launcher. A non-zero return
// from `progress` stops the render after that chunk.
int pt_launch_progressive(int integrator, int maxDepth, pt_camera camera, pt_scene* s, int numSample, int useMIS, int w, int h,
                          void* d_colors, int chunk_spp, pt_progress_fn progress, void* user) {
    if (int r = check_render_args(s, &camera, numSample, integrator)) return r;
    if (chunk_spp <= 0) return fail(-1, "chunk_spp must be positive");
    TileSpan t;
    if (int r = resolve_tiles(w, h, nullptr, t)) return r;
    if (int r = s->tilebuf.ensure(std::max<size_t>((size_t)t.count * 64 * sizeof(float4), 16))) return r;
    HIP_OK(launch_tile(w, h, t, (const float4*)d_colors, (float4*)s->tilebuf.p, nullptr));
    for (int done = 0; done < numSample;) {
        const int n = std::min(chunk_spp, numSample - done);
        if (int r = render_tiles(s, &camera, w, h, n, maxDepth, integrator, useMIS, 103033ull, t, s->tilebuf.p, nullptr, false, nullptr, done > 0)) return r;
        HIP_OK(launch_untile(w, h, t, (const float4*)s->tilebuf.p, (float4*)d_colors, nullptr));
        HIP_OK(hipDeviceSynchronize());
        if (int r = queue_error(s)) return r;
        done += n;
        if (progress && progress(done, user) != 0) break;
    }
    return 0;
}

int pt_launch_unidirectional(int maxDepth, pt_camera camera, pt_scene* scene, int numSample, int useMIS, int w, int h, void* d_colors) {
    return launch_on_colors(scene, &camera, w, h, numSample, maxDepth, PT_UNIDIRECTIONAL, useMIS, 103033ull, nullptr, d_colors, nullptr, false);
}
int pt_launch_naive_unidirectional(int maxDepth, pt_camera camera, pt_scene* scene, int numSample, int useMIS, int w, int h, void* d_colors) {
    return launch_on_colors(scene, &camera, w, h, numSample, maxDepth, PT_NAIVE_UNIDIRECTIONAL, useMIS, 103033ull, nullptr, d_colors, nullptr, false);
}

static int render_host(pt_scene* s, const pt_camera* cam, int w, int h, int spp, int maxDepth, int integrator, int useMIS, uint64_t seed,
                       const pt_tile_range* tiles, float* out, uint32_t* outCounters, bool count) {
    if (!out) return fail(-1, "null output buffer");
    if (int r = check_render_args(s, cam, spp, integrator)) return r;
    TileSpan t;
    if (int r = resolve_tiles(w, h, tiles, t)) return r;
    size_t px = (size_t)w * h;
    if (int r = s->colors.ensure(px * sizeof(float4))) return r;
    HIP_OK(hipMemcpy(s->colors.p, out, px * sizeof(float4), hipMemcpyHostToDevice));
    uint32_t* dpc = nullptr;
    if (outCounters) {
        if (int r = s->pixcnt.ensure(std::max<size_t>((size_t)t.count * 512 * sizeof(uint32_t), 16))) return r;
        dpc = (uint32_t*)s->pixcnt.p;
    }
    if (int r = launch_on_colors(s, cam, w, h, spp, maxDepth, integrator, useMIS, seed, tiles, s->colors.p, dpc, count)) return r;
    HIP_OK(hipMemcpy(out, s->colors.p, px * sizeof(float4), hipMemcpyDeviceToHost));
    if (outCounters) {
        std::vector<uint32_t> tmp((size_t)t.count * 512);
        HIP_OK(hipMemcpy(tmp.data(), dpc, tmp.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (int lt = 0; lt < t.count; lt++) {
            int tile = t.first + lt * t.stride;
            for (int lane = 0; lane < 64; lane++) {
                int x = (tile % t.tilesX) * 8 + (lane & 7), y = (tile / t.tilesX) * 8 + (lane >> 3);
                if (x >= w || y >= h) continue;
                for (int k = 0; k < 8; k++) outCounters[((size_t)y * w + x) * 8 + k] = tmp[(size_t)lt * 512 + k * 64 + lane];
            }
        }
    }
    return 0;
}

// pt_render runs the kernels the bench times; pt_render_counted the counting instantiations (same image, per-pixel
// and total work counters on the side, no time slices).
int pt_render(pt_scene* s, const pt_camera* cam, int w, int h, int spp, int maxDepth, int integrator, int useMIS, uint64_t seed,
              const pt_tile_range* tiles, float* out) {
    return render_host(s, cam, w, h, spp, maxDepth, integrator, useMIS, seed, tiles, out, nullptr, false);
}
int pt_render_counted(pt_scene* s, const pt_camera* cam, int w, int h, int spp, int maxDepth, int integrator, int useMIS, uint64_t seed,
                      const pt_tile_range* tiles, float* out, uint32_t* outCounters) {
    return render_host(s, cam, w, h, spp, maxDepth, integrator, useMIS, seed, tiles, out, outCounters, true);
}

int pt_has_experimental(void) {
#ifdef PT_EXPERIMENTAL
    return 1;
#else
    return 0;
#endif
}

int pt_set_variant(pt_scene* s, int variant) {
    if (!s) return fail(-1, "null scene");
    if (variant != 0 && variant != 1) return fail(-1, "unknown variant %d (0 = megakernel, 1 = wavefront)", variant);
    s->variant = variant;
    return 0;
}

int pt_get_counters(pt_scene* s, pt_counters* out) {
    if (!s || !out) return fail(-1, "null argument");
    HIP_OK(hipMemcpy(out, s->totals.p, sizeof(pt_counters), hipMemcpyDeviceToHost));
    return 0;
}
int pt_reset_counters(pt_scene* s) {
    if (!s) return fail(-1, "null scene");
    HIP_OK(hipMemset(s->totals.p, 0, 16 * sizeof(unsigned long long)));
    return 0;
}
// Diagnostic builds only: the eight diagnostic sums behind the counters (-DPT_STAMPS: six per-phase s_memtime /
// wall-clock sums; -DPT_UTIL: wave- and lane-level trip counts of the traversal loops); zeros otherwise.
int pt_debug_stamps(pt_scene* s, unsigned long long* out8) {
    if (!s || !out8) return fail(-1, "null argument");
    HIP_OK(hipMemcpy(out8, (unsigned long long*)s->totals.p + 8, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
}

// The tile queue's waits are bounded (pt_kernels.hip); a wait that ran out leaves q[3] != 0 and an
// unfinished frame. Read where the host waits for the kernel anyway.
static int queue_error(pt_scene* s) {
    if (!s->queue.p || s->variant != 0 || !s->lastLaunchQueued) return 0;     // (an error word left by an earlier queued launch is not this launch's)
    int q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpy(q, s->queue.p, sizeof(q), hipMemcpyDeviceToHost) != hipSuccess) return fail(-2, "tile queue read-back failed");
    if (q[3] != 0) {
        // A waiter saw no progress for the whole bound (bit 0: a stall, recorded; the waiters stayed), or gave up for good after four
        // more (bit 1; bit 2: a push found no slot). The frame is complete iff every tile has been finished (each wave counts its tile
        // in q[2] after its last state store, and the kernel has ended); only an unfinished tile is an error.
        if (q[2] >= s->lastLaunchTiles) { s->queueStalls++; return 0; }
        return fail(-4, "megakernel tile queue timed out (code %d): %d of %d tiles finished, %d pops and %d pushes claimed; the first stalled waiter saw %d finished after %.1f M ticks "
                        "of the device's steady counter without progress: the frame is incomplete", q[3], q[2], s->lastLaunchTiles, q[0], q[1], q[6], (double)q[7] * 1.048576);
    }
    return 0;
}

// How often did a tile change hands in the last megakernel launch? The ring starts with every tile pushed once (q[1] = tiles);
// every further push is a wave yielding its tile at the end of a time slice for another wave to continue.
int pt_last_tile_handovers(pt_scene* s) {
    if (!s) return fail(-1, "null scene");
    if (!s->queue.p || s->variant != 0 || !s->lastLaunchQueued) return 0;
    int q[4] = {0, 0, 0, 0};
    HIP_OK(hipMemcpy(q, s->queue.p, sizeof(q), hipMemcpyDeviceToHost));
    return std::max(0, q[1] - s->lastLaunchTiles);
}

// The 16 header words of the tile queue after the last queued launch (tests: the words' layout — q[4] / q[5], the issue-priority
// steering's sums, are back at zero when the kernel has ended, q[8..9] hold the wait bound in ticks, q[3] the stall / error bits).
int pt_debug_queue_header(pt_scene* s, int* out16) {
    if (!s || !out16) return fail(-1, "null argument");
    for (int i = 0; i < kQueueHeader; i++) out16[i] = 0;
    if (!s->queue.p || s->variant != 0 || !s->lastLaunchQueued) return 0;
    HIP_OK(hipMemcpy(out16, s->queue.p, kQueueHeader * sizeof(int), hipMemcpyDeviceToHost));
    return 1;
}

// Launches of this scene whose queue waiters gave up (no progress for "queue_timeout_ms") although the frame was complete.
int pt_queue_stalls(pt_scene* s) { return s ? s->queueStalls : 0; }

int pt_set_culling(pt_scene* s, int on) {
    if (!s) return fail(-1, "null scene");
    s->cull = on != 0;
    return 0;
}

// Per-scene kernel-selection and scheduling options (round 1 read these from the environment of the host process; an
// environment variable must not be able to change what a library renders, so they are explicit calls now). Only
// "culling" can reach the image (pt_set_culling); every other option selects between instantiations / schedules that
// are bit-identical by construction and by test.
namespace {
struct OptionRef { const char* name; int lo, hi; };
const OptionRef kOptions[] = {
    {"flat", 0, 2}, {"onchip", 0, 1}, {"waves_hbm", 0, 2}, {"refill", 0, 2}, {"refill_keep", 0, 15}, {"node_keep", 0, 15}, {"tri_keep", 0, 15},
    {"defer_shadow", 0, 1}, {"slice_iters", 0, 1 << 30}, {"slice_always", 0, 1}, {"sched_mask", 0, 1 << 20}, {"lpt_prio", 0, 2},
    {"persistent", 0, 1}, {"xcd_bands", 0, 1}, {"culling", 0, 1}, {"spec", 0, 2}, {"simple", 0, 1}, {"flat2", 0, 1}, {"leaf_boxes", 0, 1}, {"wide", 0, 1},
    {"compact", 0, 1}, {"wf_wide_wg", 0, 2}, {"lean", 0, 1}, {"queue_timeout_ms", 0, 1 << 22},
};
int option_index(const char* name) {
    if (!name) return -1;
    for (size_t i = 0; i < sizeof(kOptions) / sizeof(kOptions[0]); i++) if (!strcmp(kOptions[i].name, name)) return (int)i;
    return -1;
}
}  // namespace

int pt_set_option(pt_scene* s, const char* name, int v) {
    if (!s) return fail(-1, "null scene");
    const int k = option_index(name);
    if (k < 0) return fail(-1, "pt_set_option: unknown option '%s'", name ? name : "(null)");
    if (v < kOptions[k].lo || v > kOptions[k].hi) return fail(-1, "pt_set_option: %s = %d is outside [%d, %d]", name, v, kOptions[k].lo, kOptions[k].hi);
#ifndef PT_EXPERIMENTAL
    // A/B variants that lost their measurements are not in a default build: only their "off" value is accepted.
    {
        const bool exp = ((k == 7 || k == 13 || k == 19 || k == 20) && v != 0) || (k == 3 && v == 2) || (k == 15 && v != 2);
        if (exp) return fail(-3, "pt_set_option: %s = %d selects an experimental kernel; rebuild with `make EXPERIMENTAL=1` (pt_has_experimental() == 0)", name, v);
    }
#endif
    switch (k) {
        case 0: s->flatWanted = v; break;
        case 1: s->onchipOk = v != 0; break;
        case 2: s->wavesHbmOk = v != 0 && PT_WAVES_HBM > 0; s->wavesHbmForce = s->wavesHbmOk && v == 2; break;
        case 3: s->refill = v; break;
        case 4: s->refillKeep = v; s->refillKeepSet = true; break;
        case 5: s->nodeKeep = v; break;
        case 6: s->triKeep = v; break;
        case 7: s->deferShadow = v != 0; break;
        case 8: s->sliceIters = v; break;
        case 9: s->sliceAlways = v != 0; break;
        case 10: if (((v + 1) & v) != 0) return fail(-1, "pt_set_option: sched_mask must be 2^k - 1"); s->schedMask = v; break;
        case 11: s->lptPrio = v; break;
        case 12: s->persistent = v != 0; break;
        case 13: s->xcdBands = v != 0; break;
        case 14: s->cull = v != 0; break;
        case 15: s->spec = v; break;
        case 16: s->simpleWanted = v != 0; break;
        case 17: s->flat2Wanted = v; break;
        case 18: s->leafBoxes = v != 0; break;
        case 19: s->wideWanted = v != 0; break;
        case 20: s->compactWanted = v != 0; break;
        case 21: s->wfWideWg = v; break;
        case 22: s->leanWanted = v != 0; break;
        case 23: s->queueTimeoutMs = v; break;
    }
    return 0;
}

int pt_get_option(pt_scene* s, const char* name, int* out) {
    if (!s || !out) return fail(-1, "null argument");
    switch (option_index(name)) {
        case 0: *out = s->flatWanted; break;
        case 1: *out = s->onchipOk; break;
        case 2: *out = s->wavesHbmForce ? 2 : (s->wavesHbmOk ? 1 : 0); break;
        case 3: *out = s->refill; break;
        case 4: *out = s->refillKeep; break;
        case 5: *out = s->nodeKeep; break;
        case 6: *out = s->triKeep; break;
        case 7: *out = s->deferShadow; break;
        case 8: *out = s->sliceIters; break;
        case 9: *out = s->sliceAlways; break;
        case 10: *out = s->schedMask; break;
        case 11: *out = s->lptPrio; break;
        case 12: *out = s->persistent; break;
        case 13: *out = s->xcdBands; break;
        case 14: *out = s->cull; break;
        case 15: *out = s->spec; break;
        case 16: *out = s->simpleWanted; break;
        case 17: *out = s->flat2Wanted; break;
        case 18: *out = s->leafBoxes; break;
        case 19: *out = s->wideWanted; break;
        case 20: *out = s->compactWanted; break;
        case 21: *out = s->wfWideWg; break;
        case 22: *out = s->leanWanted; break;
        case 23: *out = s->queueTimeoutMs; break;
        default: return fail(-1, "pt_get_option: unknown option '%s'", name ? name : "(null)");
    }
    return 0;
}

int pt_scene_flags(pt_scene* s) {
    if (!s) return 0;
    const bool onchip = scene_onchip(s);
    const bool pers = s->persistent && !s->xcdBands;
    // the kernel the last launch used; before any launch, the one a full 1080p-class frame would get
    const bool hbm = s->lastLaunchHbm >= 0 ? s->lastLaunchHbm == 1 : (!onchip && !(s->deferShadow && !s->armless) && s->wavesHbmOk);
    return (onchip ? 1 : 0) | (pers ? 2 : 0) | ((pers && s->sliceIters > 0) ? 4 : 0) | (hbm ? 8 : 0) | ((s->cull && hbm) ? 16 : 0) | (s->lastLaunchRefill ? 32 : 0) | (s->lastLaunchFlat ? 64 : 0) | (s->lastLaunchSimple ? 128 : 0) | (s->lastLaunchFlat2 ? 256 : 0) | (s->lastLaunchLeafTable ? 512 : 0) | (s->lastLaunchLean ? 1024 : 0);
}

float pt_last_kernel_ms(pt_scene* s) {
    if (!s) return -1.0f;
    if (s->evPending) {                                // waits for the last megakernel launch to finish
        if (hipEventSynchronize(s->ev1) != hipSuccess || hipEventElapsedTime(&s->lastKernelMs, s->ev0, s->ev1) != hipSuccess) return -1.0f;
        s->evPending = false;
        if (queue_error(s)) return -1.0f;
    }
    return s->lastKernelMs;
}

// device-memory helpers for novum_host.cpp (which stays free of HIP headers)
int pt_host_alloc_zero_(void** p, size_t bytes) {
    HIP_OK(hipMalloc(p, std::max<size_t>(bytes, 16)));
    HIP_OK(hipMemset(*p, 0, bytes));
    return 0;
}
int pt_host_download_(void* d, void* h, size_t bytes) {
    HIP_OK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
    return 0;
}
int pt_host_download_free_(void* d, void* h, size_t bytes) {
    hipError_t e = hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(-2, "hipMemcpy D2H failed: %s", hipGetErrorString(e));
    return 0;
}

// ---- probes -------------------------------------------------------------------------------------
struct Scratch {      // RAII device scratch for the probe entry points
    std::vector<void*> ptrs;
    ~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
    void* get(size_t bytes, const void* init = nullptr) {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(bytes, 16)) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        if (init && bytes) { if (hipMemcpy(p, init, bytes, hipMemcpyHostToDevice) != hipSuccess) return nullptr; }
        return p;
    }
};
#define SCRATCH(var, type, bytes, init)                              \
    type* var = (type*)sc.get(bytes, init);                          \
    if (!var) return fail(-2, "probe: device scratch allocation failed")

static int jump_device(Scratch& sc, const uint32_t*& dj) {
    const std::vector<uint32_t>& jt = xorwow_host::jump_table();
    dj = (const uint32_t*)sc.get(jt.size() * 4, jt.data());
    return dj ? 0 : fail(-2, "probe: device scratch allocation failed");
}

int pt_probe_rng(uint64_t seed, int n, const uint32_t* subseq, int nDraws, uint32_t* outState6, uint32_t* outU32, float* outUni) {
    if (n <= 0 || nDraws < 0) return fail(-1, "bad sizes");
    Scratch sc;
    const uint32_t* dj; if (int r = jump_device(sc, dj)) return r;
    SCRATCH(dsub, uint32_t, (size_t)n * 4, subseq);
    SCRATCH(dst, uint32_t, (size_t)n * 24, nullptr);
    SCRATCH(du, uint32_t, (size_t)n * nDraws * 4, nullptr);
    SCRATCH(df, float, (size_t)n * nDraws * 4, nullptr);
    HIP_OK(launch_probe_rng(dj, seed, n, dsub, nDraws, dst, du, df, nullptr));
    HIP_OK(hipDeviceSynchronize());
    if (outState6) HIP_OK(hipMemcpy(outState6, dst, (size_t)n * 24, hipMemcpyDeviceToHost));
    if (outU32 && nDraws) HIP_OK(hipMemcpy(outU32, du, (size_t)n * nDraws * 4, hipMemcpyDeviceToHost));
    if (outUni && nDraws) HIP_OK(hipMemcpy(outUni, df, (size_t)n * nDraws * 4, hipMemcpyDeviceToHost));
    return 0;
}

int pt_probe_math(int n, const float* x, float* oSin, float* oCos, float* oExp, float* oRsqrt, float* oPow5) {
    if (n <= 0) return fail(-1, "bad sizes");
    Scratch sc;
    SCRATCH(dx, float, (size_t)n * 4, x);
    float* d[5]; float* outs[5] = {oSin, oCos, oExp, oRsqrt, oPow5};
    for (int k = 0; k < 5; k++) { d[k] = (float*)sc.get((size_t)n * 4); if (!d[k]) return fail(-2, "probe: device scratch allocation failed"); }
    HIP_OK(launch_probe_math(n, dx, d[0], d[1], d[2], d[3], d[4], nullptr));
    HIP_OK(hipDeviceSynchronize());
    for (int k = 0; k < 5; k++) if (outs[k]) HIP_OK(hipMemcpy(outs[k], d[k], (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

int pt_probe_rcp_exhaustive(unsigned long long* out3, uint32_t* first_bad) {
    if (!out3) return fail(-1, "bad arguments");
    Scratch sc;
    SCRATCH(dout, unsigned long long, 32, nullptr);
    SCRATCH(dfb, uint32_t, 16, nullptr);
    HIP_OK(hipMemset(dout, 0, 32));
    HIP_OK(hipMemset(dfb, 0xff, 16));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(launch_probe_rcp_exhaustive(dout, dfb, nullptr));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(out3, dout, 24, hipMemcpyDeviceToHost));
    if (first_bad) HIP_OK(hipMemcpy(first_bad, dfb, 4, hipMemcpyDeviceToHost));
    return 0;
}

int pt_probe_camera_rays(const pt_camera* cam, uint64_t seed, int n, const int32_t* xy, float* outRays6) {
    if (!cam || n <= 0) return fail(-1, "bad arguments");
    Scratch sc;
    const uint32_t* dj; if (int r = jump_device(sc, dj)) return r;
    std::vector<uint32_t> sub(n);
    for (int i = 0; i < n; i++) sub[i] = (uint32_t)(xy[2 * i + 1] * cam->w + xy[2 * i]);
    SCRATCH(dsub, uint32_t, (size_t)n * 4, sub.data());
    SCRATCH(dst, uint32_t, (size_t)n * 24, nullptr);
    SCRATCH(du, uint32_t, 16, nullptr);
    SCRATCH(df, float, 16, nullptr);
    SCRATCH(dxy, int, (size_t)n * 8, xy);
    SCRATCH(dout, float, (size_t)n * 24, nullptr);
    HIP_OK(launch_probe_rng(dj, seed, n, dsub, 0, dst, du, df, nullptr));
    HIP_OK(launch_probe_camera(dst, cam_to_kernel(*cam), n, dxy, dout, nullptr));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(outRays6, dout, (size_t)n * 24, hipMemcpyDeviceToHost));
    return 0;
}

static int probe_spill(pt_scene* s, int n, int32_t*& spill) {
    spill = nullptr;
    if (s->ds.stackSpill > 0) {
        if (int r = s->spill.ensure((size_t)probe_trace_blocks(n) * s->ds.stackSpill * 64 * sizeof(int32_t))) return r;
        spill = (int32_t*)s->spill.p;
    }
    return 0;
}

int pt_probe_trace_closest(pt_scene* s, int n, const float* rays6, int32_t* outI, float* outF, pt_counters* counters) {
    if (!s || n <= 0) return fail(-1, "bad arguments");
    Scratch sc;
    SCRATCH(dr, float, (size_t)n * 24, rays6);
    SCRATCH(di, int32_t, (size_t)n * 16, nullptr);
    SCRATCH(df, float, (size_t)n * 48, nullptr);
    SCRATCH(dt, unsigned long long, 64, nullptr);
    HIP_OK(hipMemset(dt, 0, 64));
    int32_t* spill; if (int r = probe_spill(s, n, spill)) return r;
    HIP_OK(launch_probe_closest(s->ds, n, dr, di, df, dt, spill, nullptr));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(outI, di, (size_t)n * 16, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(outF, df, (size_t)n * 48, hipMemcpyDeviceToHost));
    if (counters) HIP_OK(hipMemcpy(counters, dt, 64, hipMemcpyDeviceToHost));
    return 0;
}

int pt_probe_trace_shadow(pt_scene* s, int n, const float* rays6, const float* maxT, float* outThr3, pt_counters* counters) {
    if (!s || n <= 0) return fail(-1, "bad arguments");
    Scratch sc;
    SCRATCH(dr, float, (size_t)n * 24, rays6);
    SCRATCH(dm, float, (size_t)n * 4, maxT);
    SCRATCH(df, float, (size_t)n * 12, nullptr);
    SCRATCH(dt, unsigned long long, 64, nullptr);
    HIP_OK(hipMemset(dt, 0, 64));
    int32_t* spill; if (int r = probe_spill(s, n, spill)) return r;
    HIP_OK(launch_probe_shadow(s->ds, n, dr, dm, df, dt, spill, nullptr));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(outThr3, df, (size_t)n * 12, hipMemcpyDeviceToHost));
    if (counters) HIP_OK(hipMemcpy(counters, dt, 64, hipMemcpyDeviceToHost));
    return 0;
}

int pt_probe_bsdf_sample(pt_scene* s, int n, const int32_t* material, const float* wi3, const int32_t* backface, float etaI, float etaT,
                         uint64_t seed, const uint32_t* subseq, float* out8) {
    (void)etaT;      // the reference's sample_f_eval never reads etaT (reflectors.cuh:588-629)
    if (!s || n <= 0) return fail(-1, "bad arguments");
    Scratch sc;
    const uint32_t* dj; if (int r = jump_device(sc, dj)) return r;
    SCRATCH(dsub, uint32_t, (size_t)n * 4, subseq);
    SCRATCH(dst, uint32_t, (size_t)n * 24, nullptr);
    SCRATCH(du, uint32_t, 16, nullptr);
    SCRATCH(dfu, float, 16, nullptr);
    SCRATCH(dm, int, (size_t)n * 4, material);
    SCRATCH(dw, float, (size_t)n * 12, wi3);
    SCRATCH(db, int, (size_t)n * 4, backface);
    SCRATCH(dout, float, (size_t)n * 32, nullptr);
    HIP_OK(launch_probe_rng(dj, seed, n, dsub, 0, dst, du, dfu, nullptr));
    HIP_OK(launch_probe_bsdf_sample(s->ds, n, dm, dw, db, etaI, dst, dout, nullptr));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(out8, dout, (size_t)n * 32, hipMemcpyDeviceToHost));
    return 0;
}

int pt_probe_bsdf_eval(pt_scene* s, int n, const int32_t* material, const float* wi3, const float* wo3, float etaI, float etaT, float* out4) {
    (void)etaT;
    if (!s || n <= 0) return fail(-1, "bad arguments");
    Scratch sc;
    SCRATCH(dm, int, (size_t)n * 4, material);
    SCRATCH(dwi, float, (size_t)n * 12, wi3);
    SCRATCH(dwo, float, (size_t)n * 12, wo3);
    SCRATCH(dout, float, (size_t)n * 16, nullptr);
    HIP_OK(launch_probe_bsdf_eval(s->ds, n, dm, dwi, dwo, etaI, dout, nullptr));
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(out4, dout, (size_t)n * 16, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
