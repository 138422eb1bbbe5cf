// pt_megakernel.h — the megakernel (Li_unidirectional / Li_naive_unidirectional with the reference's host sample loop
// inside, deviceCode.cu:158-205, 285-542, 568-573), its tile queue and its two entry points. Included by the two
// translation units that instantiate it: pt_mk_lds.hip (scenes that live in LDS; VALU-bound, built with
// -fno-slp-vectorize: packed f32 instructions do not issue at twice the scalar rate and cost pair moves, -2.3 % on
// Cornell) and pt_mk_hbm.hip (scenes in HBM; latency-bound, +1 % WITH the vectorizer).
//
// Execution shape: one wave64 owns one 8x8-pixel tile (lane = ly*8+lx); a workgroup is 4 (or 12) independent waves that
// share one LDS scene cache. Every lane walks its pixel's samples in order (the per-pixel stream is sequential by
// construction) and lanes REGENERATE: a lane whose path ended starts its pixel's next sample at the top of the next
// bounce iteration, so the wave keeps 64 live rays for traversal until the pixels run out of samples.
#pragma once
#include "pt_path.h"
#include "pt_params.h"

namespace pt {

PT_DEV void wave_add_total(unsigned long long* totals, int k, uint32_t v) {
    unsigned long long s = v;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(&totals[k], s);
}

// Dynamic LDS of one workgroup: [scene cache: nodes | tris][4 traversal stacks][4 medium stacks].
extern __shared__ __attribute__((aligned(16))) unsigned char pt_smem[];

// All threads of the workgroup copy the cached part of the scene into LDS (16 B per thread per step). attrOff > 0
// (LDS-resident scenes): the records the bounce reads — nA PAttr, nM PMat, nL PLight — go to byte offset attrOff too.
PT_DEV SceneCache stage_scene_cache(const DeviceScene& S, int cacheNodes, int cacheTris, int attrOff = 0, int nA = 0, int nM = 0, int nL = 0) {
    typedef __attribute__((address_space(3))) f4v lds_f4;
    lds_f4* dstN = (lds_f4*)pt_smem;
    lds_f4* dstT = dstN + cacheNodes * 4;
    const f4v* srcN = reinterpret_cast<const f4v*>(S.nodes);
    const f4v* srcT = reinterpret_cast<const f4v*>(S.tris);
    for (int i = threadIdx.x; i < cacheNodes * 4; i += blockDim.x) dstN[i] = srcN[i];
    for (int i = threadIdx.x; i < cacheTris * 3; i += blockDim.x) dstT[i] = srcT[i];
    if (attrOff > 0) {
        lds_f4* dA = (lds_f4*)(pt_smem + attrOff);
        lds_f4* dM = dA + nA * 5;
        lds_f4* dL = dM + nM * 6;
        const f4v* sA = reinterpret_cast<const f4v*>(S.attrs);
        const f4v* sM = reinterpret_cast<const f4v*>(S.mats);
        const f4v* sL = reinterpret_cast<const f4v*>(S.lights);
        for (int i = threadIdx.x; i < nA * 5; i += blockDim.x) dA[i] = sA[i];
        for (int i = threadIdx.x; i < nM * 6; i += blockDim.x) dM[i] = sM[i];
        for (int i = threadIdx.x; i < nL * 4; i += blockDim.x) dL[i] = sL[i];
    }
    __syncthreads();
    SceneCache C;
    C.nodes = (lds_cf4*)dstN; C.nNodes = cacheNodes;
    C.tris = (lds_cf4*)dstT; C.nTris = cacheTris;
    return C;
}

// Diagnostic build only (-DPT_STAMPS): wave-level s_memtime shares, summed into totals[8..9]
// (logic step, traversal). Never quote this build's run time (cdna_hip_programming.md §7).
#ifdef PT_STAMPS
#define PT_STAMP(slot) do { unsigned long long now_ = __builtin_amdgcn_s_memtime(); stamp[slot] += now_ - tprev; tprev = now_; } while (0)
#else
#define PT_STAMP(slot) do {} while (0)
#endif

PT_DEV void path_finish(PathState& ps, V3& acc, bool defer) {
    const bool wait = defer && (ps.flags & kShadowPending) != 0;     // last NEE term still in flight: Li keeps the sum until apply_pending closes it (pt_path.h: PathState)
    const V3 sum = acc + ps.Li;                                        // colors[pixelIdx] += Li, deviceCode.cu:540 / :203
    acc = v3(wait ? acc.x : sum.x, wait ? acc.y : sum.y, wait ? acc.z : sum.z);
    ps.flags = (ps.flags & ~kInPath) | (wait ? kFinishPending : 0u);
}

// ---- tile queue of the persistent megakernel -------------------------------------------------
// A bounded multi-producer / multi-consumer ring in global memory:
//   q[0] pops claimed, q[1] pushes claimed, q[2] tiles finished, q[3] stall / error word (below), q[4] sum of remaining samples
//   of the tiles being worked on and q[5] waves working (issue-priority steering), q[6..7] what the first waiter that saw a stall
//   saw, q[8..9] the wait bound in ticks (64 bit, written by queue_init_kernel, read only by a waiter that sees no progress);
//   from q + kQueueHeader: cap 64-bit slots {sequence, item}
//   (cap = mask + 1 >= tiles; Vyukov's scheme: slot p%cap holds sequence p+1 when push p is in it, and
//   p+cap once pop p has taken it). It starts holding every tile once (kFreshBit). A wave that yields a
//   tile at the end of a time slice pushes it back; a tile is in the ring at most once, so it cannot overflow.
// Memory model. A yielded tile's state (RNG, accumulator, samples left) moves between waves on different
// XCDs, whose L2s are not coherent with each other. Fences at agent scope would do it, but on gfx950 they
// write back / invalidate the whole L2 each time — measured 3x slower on the 263 k-triangle scene. Instead
// every access to queue words and tile state inside the kernel is a relaxed agent-scope atomic (sc1: served
// at the memory side, never from a possibly stale cache line), slot and item travel in ONE 64-bit word, and
// "state before the queue entry" is the wave waiting for its own stores: an explicit `s_waitcnt vmcnt(0)` between the
// last state store and queue_push (see the end of megakernel_body; no fence emits it by itself).
// Waits are bounded by the wall clock: a logic error must surface as an error code, never as a hung GPU. A waiter that sees no
// progress for the bound RECORDS a stall (q[3] bit 0) and keeps waiting — a stall of the device (seen once: four persistent kernels
// co-resident on one device inside a long-lived process, 30 s, every tile finished) must cost time, not tiles, and a waiter that left
// would orphan whatever is pushed to the position it had claimed; from then on no wave yields its tile (the slices stop). Only after
// four more bounds without progress does a waiter give up for good (bit 1; a push that cannot find its slot: bit 2): then every
// waiter leaves, and the host reports the frame as incomplete if a tile is unfinished (q[2] < tiles) — never a silent partial frame.
constexpr int kFreshBit = 1 << 30;
#define PT_QLOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define PT_QSTORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
PT_DEV unsigned long long* queue_slot(int* q, int mask, unsigned pos) { return (unsigned long long*)(q + kQueueHeader) + (pos & (unsigned)mask); }
PT_DEV unsigned long long queue_timeout(int* q) { return PT_QLOAD((unsigned long long*)q + 4); }     // q[8..9]
PT_DEV int queue_pop(int* q, int mask, int nTiles, int lane, bool mayWait) {
    int item = -1;
    if (lane == 0) {
        const unsigned pos = atomicAdd((unsigned*)&q[0], 1u);
        unsigned long long* slot = queue_slot(q, mask, pos);
        unsigned long long t0 = wall_clock64();
        int seen = -1;
        for (unsigned spin = 0;; spin++) {
            const unsigned long long v = PT_QLOAD(slot);
            if ((unsigned)v == pos + 1u) {
                item = (int)(v >> 32);
                PT_QSTORE(slot, (unsigned long long)(pos + (unsigned)mask + 1u));
                break;
            }
            if (!mayWait) break;                                                // without time slices nothing is ever pushed
            if ((spin & 7) == 0) {
                const int done = PT_QLOAD(&q[2]);
                const int err = PT_QLOAD(&q[3]);
                if (done >= nTiles || (err & 6) != 0) break;                    // frame finished, or a HARD give-up (2: a waiter, 4: a push)
                // the clock only runs while nothing moves: every running wave pushes or finishes within one time
                // slice, so the end of a long frame (fewer tiles left than waves) is not a timeout
                const int progress = done + PT_QLOAD(&q[1]);
                const unsigned long long now = wall_clock64();
                const bool moved = progress != seen;
                if (moved) { seen = progress; t0 = now; }
                const unsigned long long tmo = queue_timeout(q);
                if (tmo == 0ull || (!moved && now - t0 > tmo)) {                // (a bound of zero: no waiting at all — the give-up path, deterministically, for the tests)
                    if (!(err & 1)) {
                        // A stall: recorded (bit 0; what this waiter saw in q[6], q[7]: tiles finished, the wait in 2^20 ticks), and the wait
                        // goes on — nobody abandons a claimed position for a stall, or a tile pushed to it later would be lost. From now on
                        // no wave yields its tile any more (megakernel_body reads q[3] at a slice's end), so the frame finishes without them.
                        PT_QSTORE(&q[6], done); PT_QSTORE(&q[7], (int)((now - t0) >> 20));
                        atomicOr((unsigned*)&q[3], 1u);
                        t0 = now;
                    } else if (tmo == 0ull || now - t0 > 4ull * tmo) {          // four more bounds without progress: something is lost for good
                        atomicOr((unsigned*)&q[3], 2u);
                        break;
                    }
                }
            }
            __builtin_amdgcn_s_sleep(64);
        }
    }
    return __builtin_amdgcn_readfirstlane(item);
}
PT_DEV void queue_push(int* q, int mask, int item, int lane) {
    if (lane == 0) {
        const unsigned pos = atomicAdd((unsigned*)&q[1], 1u);
        unsigned long long* slot = queue_slot(q, mask, pos);
        const unsigned long long t0 = wall_clock64();
        while ((unsigned)PT_QLOAD(slot) != pos) {
            if ((PT_QLOAD(&q[3]) & 6) != 0) return;
            if (wall_clock64() - t0 > queue_timeout(q)) { atomicOr((unsigned*)&q[3], 4u); return; }
            __builtin_amdgcn_s_sleep(4);
        }
        PT_QSTORE(slot, ((unsigned long long)(unsigned)item << 32) | (unsigned long long)(pos + 1u));
    }
}
// Tile state words: plain accesses when one wave owns the tile for the whole kernel, memory-side ones when
// tiles can change hands (see above).
PT_DEV uint32_t state_load(const uint32_t* p, bool shared) { return shared ? PT_QLOAD(p) : *p; }
PT_DEV void state_store(uint32_t* p, uint32_t v, bool shared) { if (shared) PT_QSTORE(p, v); else *p = v; }

// INTEG: 0 = Li_unidirectional, 2 = Li_naive_unidirectional. DEFER: see pt_path.h. ONCHIP: the whole packed
// scene is in the LDS cache and the stack never spills (pt_trace.h); the host decides per scene. STACKN: LDS
// stack entries per lane. The body is shared by the two kernels below, which differ in their register cap.
// LEAN: the generic bounce for scenes without MAT_LEAF triangles and without textures (pt_shade.h).
template <int INTEG, bool COUNT, bool DEFER, bool ONCHIP, int STACKN, bool CULL = false, bool REFILL = false, bool FLAT = false, bool SIMPLE = false, int FLATW = 1, int TREE = 0, bool LEAN = false>
PT_DEV void megakernel_body(const KParams& P) {
    // NOLEAF: no triangle of the scene carries a MAT_LEAF material, so a shadow ray's throughput is exactly 0 or 1 — one flag bit
    // of the resumable traversal, and the DEFER record holds the finished NEE term (PRE). True for SIMPLE and LEAN scenes and
    // for every scene the pair form of FLAT is launched on (pt_api.hip: noLeafTris).
    constexpr bool NOLEAF = SIMPLE || LEAN || (DEFER && FLAT);
    // (... and for the 4-wave SIMPLE kernel that small shares of a scene in HBM run — 128 VGPRs, chain-bound: 1/8 shares +4-5 %,
    //  profiles/r03_shards_hbm_final.log; the 8-wave kernel at 64 VGPRs loses 1-6 % to it)
#ifndef PT_TRISEL_SIMPLE4
#define PT_TRISEL_SIMPLE4 1
#endif
    constexpr bool TRISEL = ((PT_TRISEL_LEAN != 0 && LEAN && !SIMPLE) || (PT_TRISEL_SIMPLE4 != 0 && SIMPLE && STACKN == kStackLds)) && !ONCHIP && !COUNT;      // pt_trace.h: moller_trumbore_sel
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nW = blockDim.x >> 6;      // nW waves share this workgroup's scene cache
    DeviceScene S = P.S;
    // ONCHIP kernels: the host guarantees that the bounce's records fit as well (pt_api.hip: `onchip`), so their address
    // space is a compile-time fact and every access below is a ds_read.
    constexpr bool ATTRLDS = ONCHIP && kAttrCacheBytes > 0;
    constexpr int kMedBytes = SIMPLE ? 0 : kMediumMax * 64;                 // SIMPLE kernels have no medium stack (pt_path.h)
    const int attrOff = ATTRLDS ? P.cacheNodes * 64 + P.cacheTris * 48 + nW * (STACKN * 256 + kMedBytes) : 0;
    DeviceScene Sstage = S;
#ifdef PT_EXPERIMENTAL
    constexpr bool WIDE = TREE == 1, COMPACT = TREE == 2;
    if (WIDE) Sstage.nodes = reinterpret_cast<const PNode*>(P.wnodes);       // the LDS scene cache of the WIDE kernel holds wide nodes (P.cacheNodes counts 64-byte halves)
    if (COMPACT) Sstage.nodes = reinterpret_cast<const PNode*>(P.qnodes);    // ... of the COMPACT kernel 32-byte nodes, two per unit
#else
    static_assert(TREE == 0, "the wide / compact trees are -DPT_EXPERIMENTAL builds (pt_trace_experimental.h)");
#endif
    const SceneCache SC = stage_scene_cache(Sstage, P.cacheNodes, P.cacheTris, attrOff, P.cacheAttrs, P.cacheMats, P.cacheLights);      // contains the only barrier
    if constexpr (ATTRLDS) {
        // The bounce reads its records through S; pointing S at the LDS copies makes those loads ds_reads (the address
        // space is visible to the compiler: everything below is inlined into this function).
        typedef __attribute__((address_space(3))) const PAttr lds_attr;
        typedef __attribute__((address_space(3))) const PMat lds_mat;
        typedef __attribute__((address_space(3))) const PLight lds_light;
        lds_attr* a = (lds_attr*)(pt_smem + attrOff);
        lds_mat* m = (lds_mat*)(pt_smem + attrOff + P.cacheAttrs * 80);
        lds_light* l = (lds_light*)(pt_smem + attrOff + P.cacheAttrs * 80 + P.cacheMats * 96);
        S.attrs = (const PAttr*)a; S.mats = (const PMat*)m; S.lights = (const PLight*)l;
    }
    // Workgroups go to the 8 XCDs round-robin (blockIdx % 8) and each XCD has its own L2: with tile = blockIdx order the
    // XCDs interleave over the frame at 32x8-pixel granularity — every XCD gets the same mix of cheap and expensive
    // regions. (One contiguous band of tiles per XCD, so that its L2 holds only that band's geometry, measured no gain on
    // the 263 k scene — secondary rays leave the band at once — and -14 % / -9 % on the 82 k scene / Cornell: bands differ
    // in cost and a static 1/8 split cannot rebalance. Option "xcd_bands" of -DPT_EXPERIMENTAL builds.)
    int vb = blockIdx.x;
#ifdef PT_EXPERIMENTAL
    if (P.xcdBands) {
        const int nB = gridDim.x, q = nB >> 3, r = nB & 7, x = vb & 7;
        vb = x * q + (x < r ? x : r) + (vb >> 3);
    }
#endif
    // Persistent waves (P.queue != null): the grid only fills the chip and every wave takes its next tile
    // from the queue, so a wave slot is never parked behind the slowest of four sibling waves or behind
    // workgroup launch; tiles are independent, so the order does not reach the image.
    for (bool first = true;; first = false) {
    int lt;
    bool fresh = true;
    if (P.queue) {
        const int item = queue_pop(P.queue, P.queueMask, P.tileCount, lane, !COUNT && P.sliceIters > 0);
        if (item < 0) break;
        fresh = (item & kFreshBit) != 0;
        lt = item & ~kFreshBit;
    } else {
        if (!first) break;
        lt = vb * nW + wave;
    }
    if (lt >= P.tileCount) break;
    const int tile = P.tileFirst + lt * P.tileStride;
    // (the pixel coordinates are re-derived from the wave-uniform tile origin where they are needed — the start of a sample —
    // instead of living in two VGPRs across every traversal)
#ifndef PT_XY_VGPR
#define PT_XY_VGPR 0
#endif
#if PT_XY_VGPR
    const int tileX0 = (tile % P.tilesX) * 8 + (lane & 7), tileY0 = (tile / P.tilesX) * 8 + (lane >> 3);      // (per-lane x, y)
    const bool inImage = (tileX0 < P.w) && (tileY0 < P.h);
#define PT_PX (tileX0)
#define PT_PY (tileY0)
#else
    const int tileX0 = (tile % P.tilesX) * 8, tileY0 = (tile / P.tilesX) * 8;
    const bool inImage = (tileX0 + (lane & 7) < P.w) && (tileY0 + (lane >> 3) < P.h);
#define PT_PX (tileX0 + (lane & 7))
#define PT_PY (tileY0 + (lane >> 3))
#endif

    const int cacheBytes = P.cacheNodes * 64 + P.cacheTris * 48;
    Stack<STACKN> st;
    st.lds = (lds_i32*)(pt_smem + cacheBytes) + wave * (STACKN * 64) + lane;
    st.spill = P.spill ? P.spill + ((size_t)(blockIdx.x * nW + wave) * S.stackSpill) * 64 + lane : nullptr;
    st.sp = 0;
    LdsMedium msLds;
    msLds.p = (LdsMedium::lds_u8*)(pt_smem + cacheBytes + nW * STACKN * 256) + wave * kMedBytes + lane;
    NoMedium msNone;
    auto& ms = [&]() -> auto& { if constexpr (SIMPLE) return msNone; else return msLds; }();      // SIMPLE: no LDS behind it (kMedBytes = 0)

    PathState ps;
    const bool shared = P.queue != nullptr && !COUNT && P.sliceIters > 0;       // tiles may change hands
    {
        const uint32_t* r = P.rng + (size_t)lt * 384 + lane;
        ps.rng.v0 = state_load(r, shared); ps.rng.v1 = state_load(r + 64, shared); ps.rng.v2 = state_load(r + 128, shared);
        ps.rng.v3 = state_load(r + 192, shared); ps.rng.v4 = state_load(r + 256, shared); ps.rng.d = state_load(r + 320, shared);
    }
    ps.o = v3(0.0f); ps.d = v3(0.0f); ps.beta = v3(1.0f); ps.Li = v3(0.0f); ps.prevPoint = v3(0.0f); ps.woLocal = v3(0.0f);
    ps.pdf = kEps; ps.etaI = kEps; ps.etaT = kEps; ps.depth = 0; ps.guard = 0; ps.msTop = 1; ps.flags = 0;
    ps.so = v3(0.0f); ps.sd = v3(0.0f); ps.smaxt = 0.0f; ps.neeRaw = v3(0.0f); ps.neeBeta = v3(0.0f); ps.neeW = 0.0f;
    V3 acc;                                                 // colors[pixelIdx].xyz; .w is never touched (`+=` of a Li whose w is 0)
    {
        const uint32_t* o = (const uint32_t*)(P.out + (size_t)lt * 64 + lane);
        acc = v3(__uint_as_float(state_load(o, shared)), __uint_as_float(state_load(o + 1, shared)), __uint_as_float(state_load(o + 2, shared)));
    }
    Ctr c = {};
    c.gnodeFrom = (uint32_t)P.gnodeFrom;
    int samplesLeft = fresh ? (inImage ? P.spp : 0) : (int)state_load((const uint32_t*)P.left + (size_t)lt * 64 + lane, true);
    Hit h; h.tri = -1; h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.material = 0;
    V3 thr = v3(1.0f);
    RayState rs;                                          // REFILL: a lane's traversal state between two visits of the loops
    rs.o = v3(0.0f); rs.d = v3(0.0f); rs.max_t = 0.0f; rs.cur = kRefNone; rs.flags = 0u;
#ifdef PT_EXPERIMENTAL
    rs.inv = v3(0.0f); rs.min_t = 0.0f; rs.pend = kRefNone;
#endif
#ifdef PT_STAMPS
    unsigned long long stamp[4] = {0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
    const unsigned long long wall0 = wall_clock64();     // device-wide 100 MHz clock: slot occupancy (tools/stamps.py)
#endif
    auto shadowSync = [&](V3 ro, V3 wi, float maxt) {
        PT_STAMP(2);
        V3 t_ = trace_shadow<COUNT, STACKN, ONCHIP, CULL, SIMPLE || LEAN>(S, SC, ro, wi, maxt, st, c, Keep{P.nodeKeep, P.triKeep});
#ifdef PT_DIAG_DOUBLE_SHADOW        // cost measurement only (tools/phase_cost.sh): the shadow ray traced twice, same result
        {
            const V3 t2_ = trace_shadow<COUNT, STACKN, ONCHIP, CULL, SIMPLE || LEAN>(S, SC, ro, wi, maxt, st, c, Keep{P.nodeKeep, P.triKeep});
            t_ = v3(fminf_(t_.x, t2_.x), fminf_(t_.y, t2_.y), fminf_(t_.z, t2_.z));
        }
#endif
        PT_STAMP(3);
        return t_;
    };

    // Every iteration: one logic step per lane (finish the previous bounce's NEE, shade the hit,
    // regenerate if the path ended), then one traversal round for the rays the logic produced. A
    // lane whose path ended starts its pixel's next sample in the same step, so the wave keeps 64
    // live rays until the pixels run out of samples; a ballot ends the wave.
    // Longest-remaining-first inside a SIMD (P.lptPrio). Once the tile cursor is exhausted no slot gets
    // new work and the kernel ends with its slowest pixel chain (a pixel's samples are one sequential RNG
    // stream, so a chain cannot be split); a wave that is alone on its SIMD runs latency-bound, far below
    // the SIMD's throughput. Waves publish their remaining samples; a wave with more left than the mean
    // of the waves still running raises its issue priority (s_setprio), so the long chains advance at
    // near single-wave speed while the short ones fill the gaps — the image does not depend on it.
    // Time slices (P.sliceIters). With no fresh tile left, what each SIMD still has to do is whatever its
    // four waves happen to hold, and the sums differ (measured: the worst SIMD carries ~1.3x the mean when
    // every slot holds exactly one tile, the 8-GPU case). So from then on a wave works on a tile for
    // sliceIters bounce iterations, lets its lanes finish the paths in flight (no new samples), writes the
    // tile's state back (RNG, accumulator, samples left per pixel) and queues it again; the next free wave —
    // on any SIMD — continues it. The per-pixel streams continue exactly where they stopped.
    int itc = 0, myRem = 0, sliceEnd = 0x7fffffff;
    bool stopStarting = false;
    const bool lpt = P.queue != nullptr && P.lptPrio != 0;
    if (lpt) {
        myRem = samplesLeft;
        for (int o = 32; o; o >>= 1) myRem = max(myRem, __shfl_xor(myRem, o));
        if (lane == 0) { atomicAdd(&P.queue[5], 1); atomicAdd(&P.queue[4], myRem); }
    }
    while (true) {
        if (P.queue && ((++itc) & P.schedMask) == 0) {
            const bool exhausted = PT_QLOAD(&P.queue[0]) >= P.tileCount;
            if (!COUNT && P.sliceIters > 0 && (exhausted || P.sliceAlways)) {
                if (sliceEnd == 0x7fffffff) sliceEnd = itc + P.sliceIters;
                else if (itc >= sliceEnd) stopStarting = PT_QLOAD(&P.queue[3]) == 0;      // (after a stall was seen nobody yields any more: queue_pop)
            }
            if (lpt) {
                int rem = samplesLeft + ((ps.flags & kInPath) ? 1 : 0);
                for (int o = 32; o; o >>= 1) rem = max(rem, __shfl_xor(rem, o));
                if (lane == 0 && rem != myRem) atomicAdd(&P.queue[4], rem - myRem);
                myRem = rem;
                int prio = 0;
                if (exhausted || P.lptPrio == 2) {
                    const long long sum = PT_QLOAD(&P.queue[4]), a = PT_QLOAD(&P.queue[5]), act = a > 0 ? a : 1;
                    const long long r10 = 10ll * rem * act;
                    prio = r10 >= 13 * sum ? 3 : (r10 >= 11 * sum ? 2 : (r10 >= 9 * sum ? 1 : 0));
                }
                if (prio == 3) __builtin_amdgcn_s_setprio(3);
                else if (prio == 2) __builtin_amdgcn_s_setprio(2);
                else if (prio == 1) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
        }
        if (REFILL) {
            // Lanes whose rays are done take their logic step (DEFER form: the shadow ray is recorded, not traced
            // inside the bounce) and start their next pair of rays; lanes still tracing skip it and resume below.
            if (!(rs.flags & kRayBusy)) {
                if constexpr (NOLEAF) thr = (rs.flags & kRayOccluded) ? v3(0.0f) : v3(1.0f);      // the shadow ray's result is one flag bit (pt_trace.h: RayState)
                apply_pending<NOLEAF>(ps, thr, acc);
                if (ps.flags & kInPath) {
                    bool done = path_bounce<INTEG, COUNT, true, SIMPLE, LEAN, NOLEAF>(S, ps, ms, h, P.maxDepth, P.useMIS, shadowSync, c);
                    if (!done) done = path_exhausted<INTEG>(ps, P.maxDepth);
                    if (done) path_finish(ps, acc, true);
                }
                while (!(ps.flags & kInPath) && samplesLeft > 0 && !stopStarting) {
                    samplesLeft--;
                    path_begin<COUNT>(P.cam, ps, ms, PT_PX, PT_PY, c);
                    if (path_exhausted<INTEG>(ps, P.maxDepth)) path_finish(ps, acc, true);
                }
                const bool hasExt = (ps.flags & kInPath) != 0, hasShadow = (ps.flags & kShadowPending) != 0;
                if (hasExt || hasShadow) {
#ifdef PT_EXPERIMENTAL
                    bool irregular = false;
                    if constexpr (WIDE) irregular = (hasExt && !inv_is_regular(inv3(ps.d))) || (hasShadow && !inv_is_regular(inv3(ps.sd)));
                    if (WIDE && irregular) {
                        // a zero direction component: the monotonicity argument behind the wide tree does not hold (0 * inf = NaN), so
                        // this lane's rays take the reference traversal, here and now (rare: an axis-parallel direction)
                        SceneCache none; none.nodes = nullptr; none.nNodes = 0; none.tris = nullptr; none.nTris = 0;
                        thr = v3(1.0f);
                        if (hasShadow) thr = trace_shadow_plain<false, STACKN, false, false, true>(S, none, ps.so, ps.sd, ps.smaxt, st, c);
                        h.tri = -1; h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.material = 0;
                        if (hasExt) trace_closest_plain<false, STACKN, false, false>(S, none, ps.o, ps.d, 999999.0f, st, h, c);
                        rs.flags = (hasShadow && thr.x == 0.0f) ? kRayOccluded : 0u;      // (a NOLEAF scene: the throughput is 0 or 1; the next logic step reads the bit)
                    } else
#endif
                    ray_start<COUNT, STACKN>(S, st, rs, hasShadow, ps.so, ps.sd, ps.smaxt, hasExt, ps.o, ps.d, thr, h, c);
                }
            }
            PT_STAMP(2);
            const int nBusy = __builtin_popcountll(__ballot((rs.flags & kRayBusy) != 0));
            if (nBusy == 0) break;
#ifdef PT_EXPERIMENTAL
            if constexpr (COMPACT) {
                const Compact K{P.qnodes, reinterpret_cast<const f4v*>(P.leafBox), P.mids};
                trace_resume_q<STACKN>(S, SC, K, st, rs, ps.o, ps.d, (nBusy * P.refillKeep) >> 4, thr, h, Keep{P.nodeKeep, P.triKeep});
                PT_STAMP(1);
                continue;
            }
            if constexpr (WIDE) {
                trace_resume_w4<STACKN>(S, SC, P.wnodes, st, rs, ps.o, ps.d, (nBusy * P.refillKeep) >> 4, thr, h, c, Keep{P.nodeKeep, P.triKeep});
                if (!(rs.flags & kRayBusy) && (rs.flags & kRayTie)) {
                    // two triangles returned the SAME closest t: the reference keeps the one it visits first — its own traversal decides
                    SceneCache none; none.nodes = nullptr; none.nNodes = 0; none.tris = nullptr; none.nTris = 0;
                    trace_closest_plain<false, STACKN, false, false>(S, none, ps.o, ps.d, 999999.0f, st, h, c);
                    rs.flags &= ~kRayTie;
                }
                PT_STAMP(1);
                continue;
            }
#endif
#if !defined(PT_EXPERIMENTAL)
            trace_resume<COUNT, STACKN, ONCHIP, NOLEAF, TRISEL>(S, SC, st, rs, ps.o, ps.d, (nBusy * P.refillKeep) >> 4, thr, h, c, Keep{P.nodeKeep, P.triKeep});
#elif PT_SPEC == 2            // both bodies in the kernel, chosen per launch (A/B only: the second body costs registers)
            if (P.spec) trace_resume_spec<COUNT, STACKN, ONCHIP, NOLEAF>(S, SC, st, rs, ps.o, ps.d, (nBusy * P.refillKeep) >> 4, thr, h, c, Keep{P.nodeKeep, P.triKeep}, P.spec == 2);
            else trace_resume<COUNT, STACKN, ONCHIP, NOLEAF, TRISEL>(S, SC, st, rs, ps.o, ps.d, (nBusy * P.refillKeep) >> 4, thr, h, c, Keep{P.nodeKeep, P.triKeep});
#elif PT_SPEC == 1
            trace_resume_spec<COUNT, STACKN, ONCHIP, NOLEAF>(S, SC, st, rs, ps.o, ps.d, (nBusy * P.refillKeep) >> 4, thr, h, c, Keep{P.nodeKeep, P.triKeep}, P.spec == 2);
#else
            trace_resume<COUNT, STACKN, ONCHIP, NOLEAF, TRISEL>(S, SC, st, rs, ps.o, ps.d, (nBusy * P.refillKeep) >> 4, thr, h, c, Keep{P.nodeKeep, P.triKeep});
#endif
            PT_STAMP(1);
            continue;
        }
        if (DEFER) apply_pending<NOLEAF>(ps, thr, acc);
        if (ps.flags & kInPath) {
            bool done = path_bounce<INTEG, COUNT, DEFER, SIMPLE, LEAN, NOLEAF>(S, ps, ms, h, P.maxDepth, P.useMIS, shadowSync, c);
            if (!done) done = path_exhausted<INTEG>(ps, P.maxDepth);
            if (done) path_finish(ps, acc, DEFER);
        }
        PT_STAMP(2);                                   // slot 2: scheduling check + bounce logic (shading, NEE shadow ray)
        while (!(ps.flags & kInPath) && samplesLeft > 0 && !stopStarting) {
            samplesLeft--;
            path_begin<COUNT>(P.cam, ps, ms, PT_PX, PT_PY, c);
            if (path_exhausted<INTEG>(ps, P.maxDepth)) path_finish(ps, acc, DEFER);
        }
        const bool hasExt = (ps.flags & kInPath) != 0;
        const bool hasShadow = DEFER && (ps.flags & kShadowPending) != 0;
        PT_STAMP(0);
        if (__ballot(hasExt || hasShadow) == 0ull) break;
        if constexpr (DEFER && FLAT) {
            trace_pair_flat<STACKN>(S, SC, st, hasShadow, ps.so, ps.sd, ps.smaxt, hasExt, ps.o, ps.d, thr, h, c, P.cacheNodes, P.leaves, P.nLeaves);
#ifdef PT_DIAG_DOUBLE_PAIR          // cost measurement only: the pair pass run twice, same result
            { Hit h2; V3 thr2; trace_pair_flat<STACKN>(S, SC, st, hasShadow, ps.so, ps.sd, ps.smaxt, hasExt, ps.o, ps.d, thr2, h2, c, P.cacheNodes, P.leaves, P.nLeaves);
              if (hasExt && h2.tri == h.tri) h.t = fminf_(h.t, h2.t); thr.x = fminf_(thr.x, thr2.x); }
#endif
        }
#ifdef PT_EXPERIMENTAL
        else if (DEFER) trace_pair<COUNT, STACKN>(S, SC, st, hasShadow, ps.so, ps.sd, ps.smaxt, hasExt, ps.o, ps.d, thr, h, c);
#endif
        else if constexpr (FLAT) {
            trace_closest_flat<STACKN, FLATW>(S, SC, hasExt, ps.o, ps.d, 999999.0f, st, h, c, P.cacheNodes, P.leaves, P.nLeaves);
#ifdef PT_DIAG_DOUBLE_CLOSEST       // cost measurement only: the closest-hit traversal run twice, same result
            { Hit h2; trace_closest_flat<STACKN, FLATW>(S, SC, hasExt, ps.o, ps.d, 999999.0f, st, h2, c, P.cacheNodes); if (hasExt && h2.tri == h.tri) h.t = fminf_(h.t, h2.t); }
#endif
        }
        else if (hasExt) trace_closest<COUNT, STACKN, ONCHIP, CULL>(S, SC, ps.o, ps.d, 999999.0f, st, h, c, Keep{P.nodeKeep, P.triKeep});
        PT_STAMP(1);
    }

    if (lpt) {
        if (lane == 0) { atomicAdd(&P.queue[4], -myRem); atomicAdd(&P.queue[5], -1); }
        __builtin_amdgcn_s_setprio(0);
    }
#ifdef PT_STAMPS
    if (P.totals && lane == 0) {
        for (int k = 0; k < 3; k++) atomicAdd(&P.totals[8 + k], stamp[k]);   // regeneration, closest-hit traversal, bounce logic
        atomicAdd(&P.totals[14], stamp[3]);                                    // shadow rays traced inside the bounce (SYNC kernels)
        const unsigned long long wall1 = wall_clock64();
        atomicAdd(&P.totals[11], wall1 - wall0);             // sum of wave lifetimes
        atomicMax(&P.totals[12], ~wall0);                     // ~(earliest start)
        atomicMax(&P.totals[13], wall1);                      // latest end
    }
#endif
    if (inImage) {
        uint32_t* o = (uint32_t*)(P.out + (size_t)lt * 64 + lane);
        state_store(o, __float_as_uint(acc.x), shared); state_store(o + 1, __float_as_uint(acc.y), shared);
        state_store(o + 2, __float_as_uint(acc.z), shared);
    }
    {
        uint32_t* r = P.rng + (size_t)lt * 384 + lane;
        state_store(r, ps.rng.v0, shared); state_store(r + 64, ps.rng.v1, shared); state_store(r + 128, ps.rng.v2, shared);
        state_store(r + 192, ps.rng.v3, shared); state_store(r + 256, ps.rng.v4, shared); state_store(r + 320, ps.rng.d, shared);
    }
    if (P.queue) {
        int left = samplesLeft;
        for (int o = 32; o; o >>= 1) left = max(left, __shfl_xor(left, o));
        if (left > 0) {                                // yielded: somebody else continues this tile
            state_store((uint32_t*)P.left + (size_t)lt * 64 + lane, (uint32_t)samplesLeft, true);
            // "State before the queue entry": the wave waits for its OWN stores — every state word above is an sc1 store
            // counted in vmcnt (gfx9 counts stores there, in issue order), so vmcnt(0) means the memory side has them all.
            // An explicit s_waitcnt, not a fence: a workgroup-scope release fence compiles to nothing here (the two sc1
            // stores came out back to back), an agent-scope one writes back the whole L2 (3x slower). The
            // "memory" clobber keeps the compiler from moving the queue accesses above it; tests/test_isa.py checks the
            // instruction order in every non-counting instantiation.
            asm volatile("; PT_YIELD_STATE_STORED\n\ts_waitcnt vmcnt(0)" ::: "memory");
            queue_push(P.queue, P.queueMask, lt, lane);
        } else if (lane == 0) atomicAdd(&P.queue[2], 1);
    }
    if (COUNT) {
        if (P.pixCounters) {
            uint32_t* pc = P.pixCounters + (size_t)lt * 512 + lane;
            pc[0] = c.raysClosest; pc[64] = c.raysShadow; pc[128] = c.pops; pc[192] = c.boxes;
            pc[256] = c.tris; pc[320] = c.hits; pc[384] = c.draws; pc[448] = c.iters;
        }
        if (P.totals) {
            wave_add_total(P.totals, 0, c.raysClosest); wave_add_total(P.totals, 1, c.raysShadow);
            wave_add_total(P.totals, 2, c.pops); wave_add_total(P.totals, 3, c.boxes);
            wave_add_total(P.totals, 4, c.tris); wave_add_total(P.totals, 5, c.hits);
            wave_add_total(P.totals, 6, c.draws); wave_add_total(P.totals, 7, c.iters);
#ifdef PT_UTIL
            for (int k = 0; k < 8; k++) wave_add_total(P.totals, 8 + k, c.u[k]);     // in the slots of the PT_STAMPS diagnostic
#elif !defined(PT_STAMPS)
            wave_add_total(P.totals, 8, c.gnodes);                                    // pt_debug_stamps()[0] in a normal build
#endif
        }
    }
    }   // next tile
}

// Two register budgets (PMC counters): a scene that lives in LDS is VALU-bound and best at
// 4 waves per SIMD with 128 VGPRs; a scene in HBM is latency-bound (waves wait on memory 66 % of their
// cycles at 4 waves) and gains 18 % from 6 waves per SIMD at 80 VGPRs and an 8-entry LDS stack, spills
// included (5: +10 %, 7-8: no better). Both run the same body.
// LDS-resident scenes run in workgroups of 4, 8 or 16 waves (the host picks the smallest whose LDS share holds the scene:
// one copy of the scene per workgroup), hence the launch bound of 1024 for ONCHIP kernels; 4 waves per SIMD either way.
template <int INTEG, bool COUNT, bool DEFER, bool ONCHIP, bool REFILL = false, bool FLAT = false, bool SIMPLE = false, int FLATW = 1, bool LEAN = false>
__global__ void __launch_bounds__(ONCHIP ? 1024 : 256)
#if PT_MIN_WAVES > 0
__attribute__((amdgpu_waves_per_eu(PT_MIN_WAVES)))     // cap VGPRs so that PT_MIN_WAVES waves fit per SIMD
#endif
megakernel(KParams P) { megakernel_body<INTEG, COUNT, DEFER, ONCHIP, kStackLds, false, REFILL, FLAT, SIMPLE, FLATW, 0, LEAN>(P); }

template <int INTEG, bool COUNT, bool CULL, bool REFILL, bool SIMPLE = false, bool LEAN = false>
__global__ void __launch_bounds__(64 * kWgWavesHbm) __attribute__((amdgpu_waves_per_eu(kWavesHbm)))
megakernel_hbm(KParams P) { megakernel_body<INTEG, COUNT, false, false, kStackLdsHbmGen, CULL, REFILL, false, SIMPLE, 1, 0, LEAN>(P); }

// FLAT for both rays of a lane (pt_trace.h: trace_pair_flat): scenes of at most 64 nodes / triangles none of which is a MAT_LEAF
// (any hit occludes a shadow ray) and all of whose materials have a dispatch arm (the DEFER logic step is exact), MIS
// integrator; with the SIMPLE bounce where the scene allows. The wave's "stack" area is the 24 x 64-word scratch of the tests.
constexpr int kStackFlat2 = kStackFlat2Rows;
template <int INTEG, bool SIMPLE, bool LEAN = false>
__global__ void __launch_bounds__(1024)
#if PT_MIN_WAVES > 0
__attribute__((amdgpu_waves_per_eu(PT_MIN_WAVES)))
#endif
megakernel_flat2(KParams P) { megakernel_body<INTEG, false, true, true, kStackFlat2, false, false, true, SIMPLE, 1, 0, LEAN>(P); }

#ifdef PT_EXPERIMENTAL
// ... and the same on the reference tree collapsed to 4-wide nodes (pt_trace_experimental.h: trace_resume_w4).
template <int INTEG>
__global__ void __launch_bounds__(64 * kWgWavesHbmSimple) __attribute__((amdgpu_waves_per_eu(kWavesHbmSimple)))
megakernel_hbm_wide(KParams P) { megakernel_body<INTEG, false, false, false, kStackLdsHbm, false, true, false, true, 1, 1>(P); }

// ... and on 32-byte quantised nodes with exact leaf boxes (pt_trace_experimental.h: trace_resume_q).
template <int INTEG>
__global__ void __launch_bounds__(64 * kWgWavesHbmSimple) __attribute__((amdgpu_waves_per_eu(kWavesHbmSimple)))
megakernel_hbm_compact(KParams P) { megakernel_body<INTEG, false, false, false, kStackLdsHbm, false, true, false, true, 1, 2>(P); }
#endif

// The SIMPLE production kernel for scenes in HBM: 8 waves per SIMD, 16-wave workgroups (pt_params.h).
template <int INTEG>
__global__ void __launch_bounds__(64 * kWgWavesHbmSimple) __attribute__((amdgpu_waves_per_eu(kWavesHbmSimple)))
megakernel_hbm_simple(KParams P) { megakernel_body<INTEG, false, false, false, kStackLdsHbm, false, true, false, true>(P); }


}  // namespace pt
