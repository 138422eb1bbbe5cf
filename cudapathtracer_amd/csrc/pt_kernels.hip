// pt_kernels.hip — gfx950 kernels of the unidirectional path tracer.
//
//   rng_init_kernel     initRNG (deviceCode.cu:53-61): one XORWOW stream per pixel, keyed by the
//                       global pixel index, via 2^67-step GF(2) jump matrices.
//   megakernel          Li_unidirectional (deviceCode.cu:285-542) / Li_naive_unidirectional
//                       (:158-205) with the reference's host sample loop (:568-573) INSIDE the
//                       kernel: RNG state and the accumulator stay in registers for all spp, so
//                       the reference's 128 B/pixel/sample of global traffic becomes 40 B/pixel.
//   tile / untile       8x8 tile-major <-> scan-line framebuffer.
//   probe_*             single stages for known-answer tests.
//
// Execution shape: one wave64 owns one 8x8-pixel tile (lane = ly*8+lx); a 256-thread workgroup
// is four independent waves (no barriers). Every lane walks its pixel's samples in order (the
// per-pixel stream is sequential by construction) and lanes REGENERATE: a lane whose path
// ended starts its pixel's next sample at the top of the next bounce iteration, so the wave
// keeps 64 live rays for traversal until the pixels run out of samples; the wave leaves the loop
// when a ballot finds no live lane.
#include "pt_path.h"
#include "pt_params.h"

namespace pt {


PT_DEV void wave_add_total(unsigned long long* totals, int k, uint32_t v) {
    unsigned long long s = v;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(&totals[k], s);
}

// -------------------------------------------------------------------------------------------
// jump: [32][160][5] words, jump[k] = A^(2^67 * 2^k) in row-image form. Lane state v <- M v is
// the XOR of the rows selected by v's set bits; the row address is wave-uniform (scalar loads).
__global__ void __launch_bounds__(256) rng_init_kernel(const uint32_t* __restrict__ jump, unsigned long long seed, int w, int h,
                                                       int tileFirst, int tileStride, int tileCount, int tilesX, uint32_t* __restrict__ rng) {
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int lt = blockIdx.x * 4 + wave;
    if (lt >= tileCount) return;
    int tile = tileFirst + lt * tileStride;
    int x = (tile % tilesX) * 8 + (lane & 7), y = (tile / tilesX) * 8 + (lane >> 3);
    uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u, s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0, t1 = 2591861531u * s1;
    uint32_t v[5] = {123456789u + t0, 362436069u ^ t0, 521288629u + t1, 88675123u ^ t1, 5783321u + t0};
    uint32_t d = 6615241u + t1 + t0;
    uint32_t idx = (x < w && y < h) ? (uint32_t)(y * w + x) : 0u;
    for (int k = 0; k < 32; k++) {
        if (!__ballot((idx >> k) & 1u)) continue;
        if ((idx >> k) & 1u) {
            const uint32_t* M = jump + (size_t)k * 800;
            uint32_t r[5] = {0, 0, 0, 0, 0};
            for (int i = 0; i < 5; i++) {
                uint32_t word = v[i];
                for (int j = 0; j < 32; j++) {
                    uint32_t m = 0u - ((word >> j) & 1u);
                    const uint32_t* row = M + (i * 32 + j) * 5;
                    r[0] ^= row[0] & m; r[1] ^= row[1] & m; r[2] ^= row[2] & m; r[3] ^= row[3] & m; r[4] ^= row[4] & m;
                }
            }
            for (int i = 0; i < 5; i++) v[i] = r[i];
        }
    }
    uint32_t* o = rng + (size_t)lt * 384 + lane;
    o[0] = v[0]; o[64] = v[1]; o[128] = v[2]; o[192] = v[3]; o[256] = v[4]; o[320] = d;
}

// Dynamic LDS of one workgroup: [scene cache: nodes | tris][4 traversal stacks][4 medium stacks].
extern __shared__ __attribute__((aligned(16))) unsigned char pt_smem[];

// All threads of the workgroup copy the cached part of the scene into LDS (16 B per thread per step).
PT_DEV SceneCache stage_scene_cache(const DeviceScene& S, int cacheNodes, int cacheTris) {
    typedef __attribute__((address_space(3))) f4v lds_f4;
    lds_f4* dstN = (lds_f4*)pt_smem;
    lds_f4* dstT = dstN + cacheNodes * 4;
    const f4v* srcN = reinterpret_cast<const f4v*>(S.nodes);
    const f4v* srcT = reinterpret_cast<const f4v*>(S.tris);
    for (int i = threadIdx.x; i < cacheNodes * 4; i += blockDim.x) dstN[i] = srcN[i];
    for (int i = threadIdx.x; i < cacheTris * 3; i += blockDim.x) dstT[i] = srcT[i];
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
    if (defer && (ps.flags & kShadowPending)) { ps.LiFinish = ps.Li; ps.flags |= kFinishPending; }   // last NEE term still in flight
    else acc = acc + ps.Li;                      // colors[pixelIdx] += Li, deviceCode.cu:540 / :203
    ps.flags &= ~kInPath;
}

// ---- tile queue of the persistent megakernel -------------------------------------------------
// A bounded multi-producer / multi-consumer ring in global memory:
//   q[0] pops claimed, q[1] pushes claimed, q[2] tiles finished, q[3] error, q[4] sum of remaining samples
//   of the tiles being worked on, q[5] waves working; from q + 8: cap 64-bit slots {sequence, item}
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
// Waits are bounded by a wall-clock timeout that raises q[3] and drains every waiter: a logic error must
// surface as an error code, never as a hung GPU.
constexpr int kFreshBit = 1 << 30;
constexpr unsigned long long kQueueTimeout = 3000000000ull;            // 30 s of the 100 MHz wall clock without any progress
#define PT_QLOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define PT_QSTORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
PT_DEV unsigned long long* queue_slot(int* q, int mask, unsigned pos) { return (unsigned long long*)(q + 8) + (pos & (unsigned)mask); }
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
                if (done >= nTiles || PT_QLOAD(&q[3]) != 0) break;              // frame finished, or somebody gave up
                // the clock only runs while nothing moves: every running wave pushes or finishes within one time
                // slice, so the end of a long frame (fewer tiles left than waves) is not a timeout
                const int progress = done + PT_QLOAD(&q[1]);
                const unsigned long long now = wall_clock64();
                if (progress != seen) { seen = progress; t0 = now; }
                else if (now - t0 > kQueueTimeout) { PT_QSTORE(&q[3], 1); break; }
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
            if (PT_QLOAD(&q[3]) != 0) return;
            if (wall_clock64() - t0 > kQueueTimeout) { PT_QSTORE(&q[3], 2); return; }
            __builtin_amdgcn_s_sleep(4);
        }
        PT_QSTORE(slot, ((unsigned long long)(unsigned)item << 32) | (unsigned long long)(pos + 1u));
    }
}
__global__ void queue_init_kernel(int* q, int mask, int nTiles) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { q[0] = 0; q[1] = nTiles; q[2] = 0; q[3] = 0; q[4] = 0; q[5] = 0; q[6] = 0; q[7] = 0; }
    if (i <= mask)
        ((unsigned long long*)(q + 8))[i] = i < nTiles ? (((unsigned long long)(unsigned)(i | kFreshBit) << 32) | (unsigned long long)(i + 1)) : (unsigned long long)i;
}
// Tile state words: plain accesses when one wave owns the tile for the whole kernel, memory-side ones when
// tiles can change hands (see above).
PT_DEV uint32_t state_load(const uint32_t* p, bool shared) { return shared ? PT_QLOAD(p) : *p; }
PT_DEV void state_store(uint32_t* p, uint32_t v, bool shared) { if (shared) PT_QSTORE(p, v); else *p = v; }

// INTEG: 0 = Li_unidirectional, 2 = Li_naive_unidirectional. DEFER: see pt_path.h. ONCHIP: the whole packed
// scene is in the LDS cache and the stack never spills (pt_trace.h); the host decides per scene. STACKN: LDS
// stack entries per lane. The body is shared by the two kernels below, which differ in their register cap.
template <int INTEG, bool COUNT, bool DEFER, bool ONCHIP, int STACKN, bool CULL = false, bool REFILL = false, bool FLAT = false>
PT_DEV void megakernel_body(const KParams& P) {
    const DeviceScene& S = P.S;
    const SceneCache SC = stage_scene_cache(S, P.cacheNodes, P.cacheTris);      // contains the only barrier
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nW = blockDim.x >> 6;      // nW waves share this workgroup's scene cache
    // Workgroups go to the 8 XCDs round-robin (blockIdx % 8) and each XCD has its own L2. Default: tile =
    // blockIdx order, i.e. the XCDs interleave over the frame at 32x8-pixel granularity — every XCD gets
    // the same mix of cheap and expensive regions. xcdBands (PT_XCD_BANDS=1) instead gives each XCD one
    // contiguous band so its L2 holds only that band's geometry; measured: no gain on the 263 k scene
    // (secondary rays leave the band at once), -14 % / -9 % on the 82 k scene / Cornell (bands differ
    // in cost and a static 1/8 split cannot rebalance). Kept as the A/B switch.
    int vb = blockIdx.x;
    if (P.xcdBands) {
        const int nB = gridDim.x, q = nB >> 3, r = nB & 7, x = vb & 7;
        vb = x * q + (x < r ? x : r) + (vb >> 3);
    }
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
    const int x = (tile % P.tilesX) * 8 + (lane & 7), y = (tile / P.tilesX) * 8 + (lane >> 3);
    const bool inImage = (x < P.w) && (y < P.h);

    const int cacheBytes = P.cacheNodes * 64 + P.cacheTris * 48;
    Stack<STACKN> st;
    st.lds = (lds_i32*)(pt_smem + cacheBytes) + wave * (STACKN * 64) + lane;
    st.spill = P.spill ? P.spill + ((size_t)(blockIdx.x * nW + wave) * S.stackSpill) * 64 + lane : nullptr;
    st.sp = 0;
    LdsMedium ms;
    ms.p = (LdsMedium::lds_u8*)(pt_smem + cacheBytes + nW * STACKN * 256) + wave * (kMediumMax * 64) + lane;

    PathState ps;
    const bool shared = P.queue != nullptr && !COUNT && P.sliceIters > 0;       // tiles may change hands
    {
        const uint32_t* r = P.rng + (size_t)lt * 384 + lane;
        ps.rng.v0 = state_load(r, shared); ps.rng.v1 = state_load(r + 64, shared); ps.rng.v2 = state_load(r + 128, shared);
        ps.rng.v3 = state_load(r + 192, shared); ps.rng.v4 = state_load(r + 256, shared); ps.rng.d = state_load(r + 320, shared);
    }
    ps.o = v3(0.0f); ps.d = v3(0.0f); ps.beta = v3(1.0f); ps.Li = v3(0.0f); ps.prevPoint = v3(0.0f); ps.woLocal = v3(0.0f);
    ps.pdf = kEps; ps.etaI = kEps; ps.etaT = kEps; ps.depth = 0; ps.guard = 0; ps.msTop = 1; ps.flags = 0;
    ps.so = v3(0.0f); ps.sd = v3(0.0f); ps.smaxt = 0.0f; ps.neeRaw = v3(0.0f); ps.neeBeta = v3(0.0f); ps.neeW = 0.0f; ps.LiFinish = v3(0.0f);
    float4 acc4;
    {
        const uint32_t* o = (const uint32_t*)(P.out + (size_t)lt * 64 + lane);
        acc4 = make_float4(__uint_as_float(state_load(o, shared)), __uint_as_float(state_load(o + 1, shared)),
                           __uint_as_float(state_load(o + 2, shared)), __uint_as_float(state_load(o + 3, shared)));
    }
    V3 acc = v3(acc4.x, acc4.y, acc4.z);
    Ctr c = {};
    int samplesLeft = fresh ? (inImage ? P.spp : 0) : (int)state_load((const uint32_t*)P.left + (size_t)lt * 64 + lane, true);
    Hit h; h.tri = -1; h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.material = 0;
    V3 thr = v3(1.0f);
    RayState rs;                                          // REFILL: a lane's traversal state between two visits of the loops
    rs.o = v3(0.0f); rs.d = v3(0.0f); rs.inv = v3(0.0f); rs.max_t = 0.0f; rs.min_t = 0.0f; rs.cur = kRefNone; rs.flags = 0u;
#ifdef PT_STAMPS
    unsigned long long stamp[4] = {0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
    const unsigned long long wall0 = wall_clock64();     // device-wide 100 MHz clock: slot occupancy (tools/stamps.py)
#endif
    auto shadowSync = [&](V3 ro, V3 wi, float maxt) {
        PT_STAMP(2);
        const V3 t_ = trace_shadow<COUNT, STACKN, ONCHIP, CULL>(S, SC, ro, wi, maxt, st, c, Keep{P.nodeKeep, P.triKeep});
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
                else if (itc >= sliceEnd) stopStarting = true;
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
                apply_pending(ps, thr, acc);
                if (ps.flags & kInPath) {
                    bool done = path_bounce<INTEG, COUNT, true>(S, ps, ms, h, P.maxDepth, P.useMIS, shadowSync, c);
                    if (!done) done = path_exhausted<INTEG>(ps, P.maxDepth);
                    if (done) path_finish(ps, acc, true);
                }
                while (!(ps.flags & kInPath) && samplesLeft > 0 && !stopStarting) {
                    samplesLeft--;
                    path_begin<COUNT>(P.cam, ps, ms, x, y, c);
                    if (path_exhausted<INTEG>(ps, P.maxDepth)) path_finish(ps, acc, true);
                }
                const bool hasExt = (ps.flags & kInPath) != 0, hasShadow = (ps.flags & kShadowPending) != 0;
                if (hasExt || hasShadow) ray_start<COUNT, STACKN>(S, st, rs, hasShadow, ps.so, ps.sd, ps.smaxt, hasExt, ps.o, ps.d, thr, h, c);
            }
            PT_STAMP(2);
            const int nBusy = __builtin_popcountll(__ballot((rs.flags & kRayBusy) != 0));
            if (nBusy == 0) break;
            trace_resume<COUNT, STACKN, ONCHIP>(S, SC, st, rs, ps.o, ps.d, (nBusy * P.refillKeep) >> 4, thr, h, c, Keep{P.nodeKeep, P.triKeep});
            PT_STAMP(1);
            continue;
        }
        if (DEFER) apply_pending(ps, thr, acc);
        if (ps.flags & kInPath) {
            bool done = path_bounce<INTEG, COUNT, DEFER>(S, ps, ms, h, P.maxDepth, P.useMIS, shadowSync, c);
            if (!done) done = path_exhausted<INTEG>(ps, P.maxDepth);
            if (done) path_finish(ps, acc, DEFER);
        }
        PT_STAMP(2);                                   // slot 2: scheduling check + bounce logic (shading, NEE shadow ray)
        while (!(ps.flags & kInPath) && samplesLeft > 0 && !stopStarting) {
            samplesLeft--;
            path_begin<COUNT>(P.cam, ps, ms, x, y, c);
            if (path_exhausted<INTEG>(ps, P.maxDepth)) path_finish(ps, acc, DEFER);
        }
        const bool hasExt = (ps.flags & kInPath) != 0;
        const bool hasShadow = DEFER && (ps.flags & kShadowPending) != 0;
        PT_STAMP(0);
        if (__ballot(hasExt || hasShadow) == 0ull) break;
        if (DEFER) trace_pair<COUNT, STACKN>(S, SC, st, hasShadow, ps.so, ps.sd, ps.smaxt, hasExt, ps.o, ps.d, thr, h, c);
        else if constexpr (FLAT) trace_closest_flat<STACKN>(S, SC, hasExt, ps.o, ps.d, 999999.0f, st, h, c, P.cacheNodes);
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
        state_store(o + 2, __float_as_uint(acc.z), shared); state_store(o + 3, __float_as_uint(acc4.w), shared);
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
#endif
        }
    }
    }   // next tile
}

// Two register budgets (PMC counters): a scene that lives in LDS is VALU-bound and best at
// 4 waves per SIMD with 128 VGPRs; a scene in HBM is latency-bound (waves wait on memory 66 % of their
// cycles at 4 waves) and gains 18 % from 6 waves per SIMD at 80 VGPRs and an 8-entry LDS stack, spills
// included (5: +10 %, 7-8: no better). Both run the same body.
template <int INTEG, bool COUNT, bool DEFER, bool ONCHIP, bool REFILL = false, bool FLAT = false>
__global__ void __launch_bounds__(256)
#if PT_MIN_WAVES > 0
__attribute__((amdgpu_waves_per_eu(PT_MIN_WAVES)))     // cap VGPRs so that PT_MIN_WAVES waves fit per SIMD
#endif
megakernel(KParams P) { megakernel_body<INTEG, COUNT, DEFER, ONCHIP, kStackLds, false, REFILL, FLAT>(P); }

template <int INTEG, bool COUNT, bool CULL, bool REFILL>
__global__ void __launch_bounds__(64 * kWgWavesHbm) __attribute__((amdgpu_waves_per_eu(kWavesHbm)))
megakernel_hbm(KParams P) { megakernel_body<INTEG, COUNT, false, false, kStackLdsHbm, CULL, REFILL>(P); }

// -------------------------------------------------------------------------------------------
// tile-major [local tile][64] <-> scan-line colors[y*w+x]
__global__ void __launch_bounds__(256) untile_kernel(int w, int h, int tileFirst, int tileStride, int tileCount, int tilesX,
                                                     const float4* __restrict__ tiles, float4* __restrict__ colors) {
    int lt = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (lt >= tileCount) return;
    int tile = tileFirst + lt * tileStride;
    int x = (tile % tilesX) * 8 + (lane & 7), y = (tile / tilesX) * 8 + (lane >> 3);
    if (x < w && y < h) colors[(size_t)y * w + x] = tiles[(size_t)lt * 64 + lane];
}
__global__ void __launch_bounds__(256) tile_kernel(int w, int h, int tileFirst, int tileStride, int tileCount, int tilesX,
                                                   const float4* __restrict__ colors, float4* __restrict__ tiles) {
    int lt = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (lt >= tileCount) return;
    int tile = tileFirst + lt * tileStride;
    int x = (tile % tilesX) * 8 + (lane & 7), y = (tile / tilesX) * 8 + (lane >> 3);
    tiles[(size_t)lt * 64 + lane] = (x < w && y < h) ? colors[(size_t)y * w + x] : make_float4(0, 0, 0, 0);
}

// ---- probes ---------------------------------------------------------------------------------
__global__ void probe_rng_kernel(const uint32_t* __restrict__ jump, unsigned long long seed, int n, const uint32_t* __restrict__ subseq,
                                 int nDraws, uint32_t* outState, uint32_t* outU32, float* outUni) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u, s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0, t1 = 2591861531u * s1;
    uint32_t v[5] = {123456789u + t0, 362436069u ^ t0, 521288629u + t1, 88675123u ^ t1, 5783321u + t0};
    uint32_t d = 6615241u + t1 + t0;
    uint32_t idx = subseq[i];
    for (int k = 0; k < 32; k++) {
        if (!((idx >> k) & 1u)) continue;
        const uint32_t* M = jump + (size_t)k * 800;
        uint32_t r[5] = {0, 0, 0, 0, 0};
        for (int b = 0; b < 160; b++) {
            uint32_t m = 0u - ((v[b >> 5] >> (b & 31)) & 1u);
            for (int q = 0; q < 5; q++) r[q] ^= M[b * 5 + q] & m;
        }
        for (int q = 0; q < 5; q++) v[q] = r[q];
    }
    for (int q = 0; q < 5; q++) outState[i * 6 + q] = v[q];
    outState[i * 6 + 5] = d;
    Rng a = {v[0], v[1], v[2], v[3], v[4], d}, b = a;
    for (int k = 0; k < nDraws; k++) { outU32[i * nDraws + k] = rng_next(a); outUni[i * nDraws + k] = rng_uniform(b); }
}

__global__ void probe_math_kernel(int n, const float* x, float* s, float* cth, float* e, float* rs, float* p5) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float sn, cs; sincos_(x[i], sn, cs);
    s[i] = sn; cth[i] = cs; e[i] = exp_(x[i]); rs[i] = rsqrt_(x[i]); p5[i] = pow5_(x[i]);
}

__global__ void probe_camera_kernel(const uint32_t* __restrict__ state6, CamK cam, int n, const int* xy, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng r = {state6[i * 6], state6[i * 6 + 1], state6[i * 6 + 2], state6[i * 6 + 3], state6[i * 6 + 4], state6[i * 6 + 5]};
    Ctr c = {};
    V3 o, d;
    camera_ray<false>(cam, r, xy[2 * i], xy[2 * i + 1], o, d, c);
    out[6 * i] = o.x; out[6 * i + 1] = o.y; out[6 * i + 2] = o.z; out[6 * i + 3] = d.x; out[6 * i + 4] = d.y; out[6 * i + 5] = d.z;
}

__global__ void __launch_bounds__(64) probe_closest_kernel(DeviceScene S, int n, const float* rays, int32_t* outI, float* outF,
                                                           unsigned long long* totals, int32_t* spill) {
    __shared__ int32_t ldsStack[kStackLds][64];
    int i = blockIdx.x * 64 + threadIdx.x, lane = threadIdx.x;
    Stack<kStackLds> st; st.lds = (lds_i32*)&ldsStack[0][0] + lane; st.sp = 0;
    SceneCache C; C.nodes = nullptr; C.nNodes = 0; C.tris = nullptr; C.nTris = 0;
    st.spill = spill ? spill + ((size_t)blockIdx.x * S.stackSpill) * 64 + lane : nullptr;
    Ctr c = {};
    if (i < n) {
        V3 o = v3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]), d = v3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
        Hit h;
        trace_closest<true, kStackLds>(S, C, o, d, 999999.0f, st, h, c);
        float* f = outF + 12 * i;
        for (int k = 0; k < 12; k++) f[k] = 0.0f;
        if (h.tri >= 0) {
            HitInfo hi; resolve_hit(S, h, o, d, hi);
            outI[4 * i] = 1; outI[4 * i + 1] = hi.tri; outI[4 * i + 2] = hi.material; outI[4 * i + 3] = hi.backface;
            f[0] = h.t; f[1] = h.u; f[2] = h.v; f[3] = hi.point.x; f[4] = hi.point.y; f[5] = hi.point.z;
            f[6] = hi.normal.x; f[7] = hi.normal.y; f[8] = hi.normal.z; f[9] = hi.uvx; f[10] = hi.uvy;
        } else { outI[4 * i] = 0; outI[4 * i + 1] = -1; outI[4 * i + 2] = -1; outI[4 * i + 3] = 0; }
    }
    wave_add_total(totals, 0, c.raysClosest); wave_add_total(totals, 1, c.raysShadow); wave_add_total(totals, 2, c.pops);
    wave_add_total(totals, 3, c.boxes); wave_add_total(totals, 4, c.tris); wave_add_total(totals, 5, c.hits);
}

__global__ void __launch_bounds__(64) probe_shadow_kernel(DeviceScene S, int n, const float* rays, const float* maxT, float* outF,
                                                          unsigned long long* totals, int32_t* spill) {
    __shared__ int32_t ldsStack[kStackLds][64];
    int i = blockIdx.x * 64 + threadIdx.x, lane = threadIdx.x;
    Stack<kStackLds> st; st.lds = (lds_i32*)&ldsStack[0][0] + lane; st.sp = 0;
    SceneCache C; C.nodes = nullptr; C.nNodes = 0; C.tris = nullptr; C.nTris = 0;
    st.spill = spill ? spill + ((size_t)blockIdx.x * S.stackSpill) * 64 + lane : nullptr;
    Ctr c = {};
    if (i < n) {
        V3 o = v3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]), d = v3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
        V3 t = trace_shadow<true, kStackLds>(S, C, o, d, maxT[i], st, c);
        outF[3 * i] = t.x; outF[3 * i + 1] = t.y; outF[3 * i + 2] = t.z;
    }
    wave_add_total(totals, 0, c.raysClosest); wave_add_total(totals, 1, c.raysShadow); wave_add_total(totals, 2, c.pops);
    wave_add_total(totals, 3, c.boxes); wave_add_total(totals, 4, c.tris); wave_add_total(totals, 5, c.hits);
}

__global__ void probe_bsdf_sample_kernel(DeviceScene S, int n, const int* material, const float* wi3, const int* backface, float etaI,
                                         const uint32_t* __restrict__ state6, float* out8) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng r = {state6[i * 6], state6[i * 6 + 1], state6[i * 6 + 2], state6[i * 6 + 3], state6[i * 6 + 4], state6[i * 6 + 5]};
    Ctr c = {};
    V3 wo = v3(0.0f), f = v3(0.0f); float pdf = 0.0f;
    sample_f_eval<true>(r, S.mats[material[i]], S.textures, v3(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]), etaI, backface[i] != 0, wo, f, pdf, 0.0f, 0.0f, c);
    float* o = out8 + 8 * i;
    o[0] = wo.x; o[1] = wo.y; o[2] = wo.z; o[3] = f.x; o[4] = f.y; o[5] = f.z; o[6] = pdf; o[7] = (float)c.draws;
}

__global__ void probe_bsdf_eval_kernel(DeviceScene S, int n, const int* material, const float* wi3, const float* wo3, float etaI, float* out4) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PMat& m = S.mats[material[i]];
    V3 wi = v3(wi3[3 * i], wi3[3 * i + 1], wi3[3 * i + 2]), wo = v3(wo3[3 * i], wo3[3 * i + 1], wo3[3 * i + 2]);
    V3 f = f_eval(m, S.textures, wi, wo, etaI, 0.0f, 0.0f);
    float pdf = 0.0f;
    pdf_eval(m, S.textures, wi, wo, etaI, 0.0f, 0.0f, pdf);
    out4[4 * i] = f.x; out4[4 * i + 1] = f.y; out4[4 * i + 2] = f.z; out4[4 * i + 3] = pdf;
}

}  // namespace pt

// ---- host-callable launchers ---------------------------------------------------------------
namespace pt {

hipError_t launch_rng_init(const uint32_t* jump, unsigned long long seed, int w, int h, TileSpan t, uint32_t* rng, hipStream_t stream) {
    if (t.count <= 0) return hipSuccess;
    hipLaunchKernelGGL(rng_init_kernel, dim3((t.count + 3) / 4), dim3(256), 0, stream, jump, seed, w, h, t.first, t.stride, t.count, t.tilesX, rng);
    return hipGetLastError();
}

hipError_t launch_megakernel(int integrator, bool count, bool syncShadow, const KParams& P, hipStream_t stream) {
    if (P.tileCount <= 0) return hipSuccess;
    int nBlocks = megakernel_blocks(P.tileCount, P.wgWaves);
    if (P.queue && P.gridBlocks > 0) {
        nBlocks = std::min(nBlocks, P.gridBlocks);
        hipLaunchKernelGGL(queue_init_kernel, dim3((P.queueMask + 256) / 256), dim3(256), 0, stream, P.queue, P.queueMask, P.tileCount);
    }
    dim3 grid(nBlocks), block(64 * P.wgWaves);
    const bool hbm = P.wavesPerSimd == kWavesHbm;              // chosen by the host together with the spill layout
    const unsigned lds = (unsigned)megakernel_lds_bytes(P.cacheNodes, P.cacheTris, hbm ? kStackLdsHbm : kStackLds, P.wgWaves);
    // more than 64 KB of dynamic LDS per workgroup has to be asked for (a workgroup may take all 160 KB of its CU)
#define PT_LDS_OK(K) do { if (lds > 65536u) { hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e_ != hipSuccess) return e_; } } while (0)
#define PT_LAUNCH_MK(I, C, D, O) hipLaunchKernelGGL((megakernel<I, C, D, O>), grid, block, lds, stream, P)
#define PT_LAUNCH_HBM1(I, C, CU, RF) do { PT_LDS_OK((megakernel_hbm<I, C, CU, RF>)); hipLaunchKernelGGL((megakernel_hbm<I, C, CU, RF>), grid, block, lds, stream, P); } while (0)
#define PT_LAUNCH_HBM(I, C) do { if (P.cull) PT_LAUNCH_HBM1(I, C, true, false); else if (P.refill) PT_LAUNCH_HBM1(I, C, false, true); \
                                 else PT_LAUNCH_HBM1(I, C, false, false); } while (0)
#define PT_LAUNCH_RF(I, C) do { if (P.onchip) hipLaunchKernelGGL((megakernel<I, C, false, true, true>), grid, block, lds, stream, P); \
                                else hipLaunchKernelGGL((megakernel<I, C, false, false, true>), grid, block, lds, stream, P); } while (0)
#define PT_LAUNCH_MK2(I) do { if (hbm) { if (count) PT_LAUNCH_HBM(I, true); else PT_LAUNCH_HBM(I, false); } \
                              else if (P.refill) { if (count) PT_LAUNCH_RF(I, true); else PT_LAUNCH_RF(I, false); } \
                              else if (P.flat && !count) hipLaunchKernelGGL((megakernel<I, false, false, true, false, true>), grid, block, lds, stream, P); \
                              else if (count) { if (P.onchip) PT_LAUNCH_MK(I, true, false, true); else PT_LAUNCH_MK(I, true, false, false); } \
                              else { if (P.onchip) PT_LAUNCH_MK(I, false, false, true); else PT_LAUNCH_MK(I, false, false, false); } } while (0)
    if (integrator == 2) PT_LAUNCH_MK2(2);
    else if (syncShadow) PT_LAUNCH_MK2(0);
    else { if (count) PT_LAUNCH_MK(0, true, true, false); else PT_LAUNCH_MK(0, false, true, false); }
    return hipGetLastError();
}

hipError_t launch_untile(int w, int h, TileSpan t, const float4* tiles, float4* colors, hipStream_t stream) {
    if (t.count <= 0) return hipSuccess;
    hipLaunchKernelGGL(untile_kernel, dim3((t.count + 3) / 4), dim3(256), 0, stream, w, h, t.first, t.stride, t.count, t.tilesX, tiles, colors);
    return hipGetLastError();
}
hipError_t launch_tile(int w, int h, TileSpan t, const float4* colors, float4* tiles, hipStream_t stream) {
    if (t.count <= 0) return hipSuccess;
    hipLaunchKernelGGL(tile_kernel, dim3((t.count + 3) / 4), dim3(256), 0, stream, w, h, t.first, t.stride, t.count, t.tilesX, colors, tiles);
    return hipGetLastError();
}
hipError_t launch_probe_rng(const uint32_t* jump, unsigned long long seed, int n, const uint32_t* subseq, int nDraws,
                            uint32_t* outState, uint32_t* outU32, float* outUni, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(probe_rng_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, jump, seed, n, subseq, nDraws, outState, outU32, outUni);
    return hipGetLastError();
}
hipError_t launch_probe_math(int n, const float* x, float* s, float* c, float* e, float* rs, float* p5, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(probe_math_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, x, s, c, e, rs, p5);
    return hipGetLastError();
}
hipError_t launch_probe_camera(const uint32_t* state6, const CamK& cam, int n, const int* xy, float* out, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(probe_camera_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, state6, cam, n, xy, out);
    return hipGetLastError();
}
hipError_t launch_probe_closest(const DeviceScene& S, int n, const float* rays, int32_t* outI, float* outF,
                                unsigned long long* totals, int32_t* spill, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(probe_closest_kernel, dim3(probe_trace_blocks(n)), dim3(64), 0, stream, S, n, rays, outI, outF, totals, spill);
    return hipGetLastError();
}
hipError_t launch_probe_shadow(const DeviceScene& S, int n, const float* rays, const float* maxT, float* outF,
                               unsigned long long* totals, int32_t* spill, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(probe_shadow_kernel, dim3(probe_trace_blocks(n)), dim3(64), 0, stream, S, n, rays, maxT, outF, totals, spill);
    return hipGetLastError();
}
hipError_t launch_probe_bsdf_sample(const DeviceScene& S, int n, const int* material, const float* wi3, const int* backface, float etaI,
                                    const uint32_t* state6, float* out8, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(probe_bsdf_sample_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, S, n, material, wi3, backface, etaI, state6, out8);
    return hipGetLastError();
}
hipError_t launch_probe_bsdf_eval(const DeviceScene& S, int n, const int* material, const float* wi3, const float* wo3, float etaI,
                                  float* out4, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(probe_bsdf_eval_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, S, n, material, wi3, wo3, etaI, out4);
    return hipGetLastError();
}

}  // namespace pt
