// pt_wavefront.hip — stream-compacted ("wavefront") variant of the unidirectional integrators
// (BASELINE config 5): the divergence A/B against the megakernel.
//
// Same arithmetic, same per-pixel XORWOW streams, same results bit for bit; what changes is how
// lanes are kept busy. The path state lives in HBM (SoA, one slot per pixel of the rank's tiles)
// and every bounce is two kernels:
//
//   wf_logic   one thread per path slot: applies the previous bounce's deferred NEE term, shades
//              the new hit (pt_path.h, DEFER policy), regenerates finished paths, and appends the
//              rays it produced — the next extension ray and, after NEE, a shadow ray — to a
//              compacted ray queue (one wave-aggregated atomic per wave).
//   wf_trace   persistent waves. A lane that finishes its ray takes the NEXT ray from the queue
//              instead of idling until the slowest lane of its wave is done (lane-level refill),
//              which is exactly what the megakernel cannot do: there a lane is tied to its pixel.
//
// The queue order only affects which lane traces which ray; results are written to per-ray slots,
// so the image does not depend on it.
#include "pt_path.h"
#include "pt_params.h"

namespace pt {

// Field indices of the SoA state (word arrays of stride n = number of path slots).
enum WfF : int {   // float fields
    F_OX, F_OY, F_OZ, F_DX, F_DY, F_DZ, F_BX, F_BY, F_BZ, F_LIX, F_LIY, F_LIZ, F_PPX, F_PPY, F_PPZ, F_WOX, F_WOY, F_WOZ,
    F_PDF, F_ETAI, F_ETAT,
    F_SOX, F_SOY, F_SOZ, F_SDX, F_SDY, F_SDZ, F_SMAXT,
    F_NRX, F_NRY, F_NRZ, F_NBX, F_NBY, F_NBZ, F_NW, F_ACX, F_ACY, F_ACZ,
    F_COUNT
};
enum WfU : int {   // uint32 fields
    U_R0, U_R1, U_R2, U_R3, U_R4, U_RD, U_DEPTH, U_GUARD, U_MSTOP, U_FLAGS, U_SAMPLES, U_M0, U_M1, U_M2, U_M3,
    U_COUNT
};
// results per RAY slot (extension ray of path p: slot p; shadow ray of path p: slot n + p)
enum WfR : int { R_T, R_U, R_V, R_TRI, R_MAT, R_THRX, R_THRY, R_THRZ, R_POPS, R_BOXES, R_TRIS, R_COUNT };

// SIMPLE (diffuse-only scenes, pt_path.h): the medium stack words, its top, the two indices of refraction, the sampled local
// direction and — the bounce records the finished NEE term (apply_pending<PRE>) — the term's beta and weight are never read.
template <bool SIMPLE = false>
PT_DEV void wf_load(const WfParams& W, int p, PathState& ps, RegMedium& ms, V3& acc, int& samplesLeft) {
    const float* F = W.F + p; const uint32_t* U = W.U + p; const size_t n = W.n;
    ps.o = v3(F[F_OX * n], F[F_OY * n], F[F_OZ * n]); ps.d = v3(F[F_DX * n], F[F_DY * n], F[F_DZ * n]);
    ps.beta = v3(F[F_BX * n], F[F_BY * n], F[F_BZ * n]); ps.Li = v3(F[F_LIX * n], F[F_LIY * n], F[F_LIZ * n]);
    ps.prevPoint = v3(F[F_PPX * n], F[F_PPY * n], F[F_PPZ * n]);
    ps.woLocal = SIMPLE ? v3(0.0f) : v3(F[F_WOX * n], F[F_WOY * n], F[F_WOZ * n]);
    ps.pdf = F[F_PDF * n];
    if (SIMPLE) { ps.etaI = kEps; ps.etaT = kEps; } else { ps.etaI = F[F_ETAI * n]; ps.etaT = F[F_ETAT * n]; }
    ps.so = v3(F[F_SOX * n], F[F_SOY * n], F[F_SOZ * n]); ps.sd = v3(F[F_SDX * n], F[F_SDY * n], F[F_SDZ * n]); ps.smaxt = F[F_SMAXT * n];
    ps.neeRaw = v3(F[F_NRX * n], F[F_NRY * n], F[F_NRZ * n]);
    if (SIMPLE) { ps.neeBeta = v3(0.0f); ps.neeW = 0.0f; } else { ps.neeBeta = v3(F[F_NBX * n], F[F_NBY * n], F[F_NBZ * n]); ps.neeW = F[F_NW * n]; }
    acc = v3(F[F_ACX * n], F[F_ACY * n], F[F_ACZ * n]);
    ps.rng.v0 = U[U_R0 * n]; ps.rng.v1 = U[U_R1 * n]; ps.rng.v2 = U[U_R2 * n]; ps.rng.v3 = U[U_R3 * n]; ps.rng.v4 = U[U_R4 * n]; ps.rng.d = U[U_RD * n];
    ps.depth = (int)U[U_DEPTH * n]; ps.guard = (int)U[U_GUARD * n]; ps.flags = U[U_FLAGS * n];
    samplesLeft = (int)U[U_SAMPLES * n];
    if (SIMPLE) { ps.msTop = 1; ms.w0 = ms.w1 = ms.w2 = ms.w3 = 0u; }
    else { ps.msTop = (int)U[U_MSTOP * n]; ms.w0 = U[U_M0 * n]; ms.w1 = U[U_M1 * n]; ms.w2 = U[U_M2 * n]; ms.w3 = U[U_M3 * n]; }
}

template <bool SIMPLE = false>
PT_DEV void wf_store(const WfParams& W, int p, const PathState& ps, const RegMedium& ms, V3 acc, int samplesLeft) {
    float* F = W.F + p; uint32_t* U = W.U + p; const size_t n = W.n;
    F[F_OX * n] = ps.o.x; F[F_OY * n] = ps.o.y; F[F_OZ * n] = ps.o.z; F[F_DX * n] = ps.d.x; F[F_DY * n] = ps.d.y; F[F_DZ * n] = ps.d.z;
    F[F_BX * n] = ps.beta.x; F[F_BY * n] = ps.beta.y; F[F_BZ * n] = ps.beta.z; F[F_LIX * n] = ps.Li.x; F[F_LIY * n] = ps.Li.y; F[F_LIZ * n] = ps.Li.z;
    F[F_PPX * n] = ps.prevPoint.x; F[F_PPY * n] = ps.prevPoint.y; F[F_PPZ * n] = ps.prevPoint.z;
    if (!SIMPLE) { F[F_WOX * n] = ps.woLocal.x; F[F_WOY * n] = ps.woLocal.y; F[F_WOZ * n] = ps.woLocal.z; }
    F[F_PDF * n] = ps.pdf;
    if (!SIMPLE) { F[F_ETAI * n] = ps.etaI; F[F_ETAT * n] = ps.etaT; }
    F[F_SOX * n] = ps.so.x; F[F_SOY * n] = ps.so.y; F[F_SOZ * n] = ps.so.z; F[F_SDX * n] = ps.sd.x; F[F_SDY * n] = ps.sd.y; F[F_SDZ * n] = ps.sd.z; F[F_SMAXT * n] = ps.smaxt;
    F[F_NRX * n] = ps.neeRaw.x; F[F_NRY * n] = ps.neeRaw.y; F[F_NRZ * n] = ps.neeRaw.z;
    if (!SIMPLE) { F[F_NBX * n] = ps.neeBeta.x; F[F_NBY * n] = ps.neeBeta.y; F[F_NBZ * n] = ps.neeBeta.z; F[F_NW * n] = ps.neeW; }
    F[F_ACX * n] = acc.x; F[F_ACY * n] = acc.y; F[F_ACZ * n] = acc.z;
    U[U_R0 * n] = ps.rng.v0; U[U_R1 * n] = ps.rng.v1; U[U_R2 * n] = ps.rng.v2; U[U_R3 * n] = ps.rng.v3; U[U_R4 * n] = ps.rng.v4; U[U_RD * n] = ps.rng.d;
    U[U_DEPTH * n] = (uint32_t)ps.depth; U[U_GUARD * n] = (uint32_t)ps.guard; U[U_FLAGS * n] = ps.flags;
    U[U_SAMPLES * n] = (uint32_t)samplesLeft;
    if (!SIMPLE) { U[U_MSTOP * n] = (uint32_t)ps.msTop; U[U_M0 * n] = ms.w0; U[U_M1 * n] = ms.w1; U[U_M2 * n] = ms.w2; U[U_M3 * n] = ms.w3; }
}

// Slot p = local tile * 64 + lane, the same mapping as the megakernel's tile buffer.
__global__ void __launch_bounds__(256) wf_init_kernel(WfParams W, int spp) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= W.n) return;
    const int lt = p >> 6, lane = p & 63;
    const int tile = W.tileFirst + lt * W.tileStride;
    const int x = (tile % W.tilesX) * 8 + (lane & 7), y = (tile / W.tilesX) * 8 + (lane >> 3);
    PathState ps;
    ps.o = v3(0.0f); ps.d = v3(0.0f); ps.beta = v3(1.0f); ps.Li = v3(0.0f); ps.prevPoint = v3(0.0f); ps.woLocal = v3(0.0f);
    ps.pdf = kEps; ps.etaI = kEps; ps.etaT = kEps; ps.depth = 0; ps.guard = 0; ps.msTop = 1; ps.flags = 0;
    ps.so = v3(0.0f); ps.sd = v3(0.0f); ps.smaxt = 0.0f; ps.neeRaw = v3(0.0f); ps.neeBeta = v3(0.0f); ps.neeW = 0.0f;
    const uint32_t* r = W.rng + (size_t)lt * 384 + lane;
    ps.rng.v0 = r[0]; ps.rng.v1 = r[64]; ps.rng.v2 = r[128]; ps.rng.v3 = r[192]; ps.rng.v4 = r[256]; ps.rng.d = r[320];
    float4 a = W.out[p];
    RegMedium ms; ms.w0 = ms.w1 = ms.w2 = ms.w3 = 0u;
    wf_store(W, p, ps, ms, v3(a.x, a.y, a.z), (x < W.w && y < W.h) ? spp : 0);
    if (W.pathCtr) for (int k = 0; k < 8; k++) W.pathCtr[(size_t)k * W.n + p] = 0u;
}

__global__ void __launch_bounds__(256) wf_finish_kernel(WfParams W) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= W.n) return;
    const int lt = p >> 6, lane = p & 63;
    const size_t n = W.n;
    float4 a = W.out[p];
    W.out[p] = make_float4(W.F[F_ACX * n + p], W.F[F_ACY * n + p], W.F[F_ACZ * n + p], a.w);
    uint32_t* r = W.rng + (size_t)lt * 384 + lane;
    r[0] = W.U[U_R0 * n + p]; r[64] = W.U[U_R1 * n + p]; r[128] = W.U[U_R2 * n + p]; r[192] = W.U[U_R3 * n + p]; r[256] = W.U[U_R4 * n + p]; r[320] = W.U[U_RD * n + p];
}

// One logic step per path slot. `it` selects the queue counter pair of this iteration.
template <int INTEG, bool COUNT, bool SIMPLE = false>
__global__ void __launch_bounds__(256) wf_logic_kernel(WfParams W, DeviceScene S, CamK cam, int maxDepth, int useMIS, int it) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool hasExt = false, hasShadow = false;
    if (p < W.n) {
        const size_t n = W.n;
        PathState ps; RegMedium ms; V3 acc; int samplesLeft;
        wf_load<SIMPLE>(W, p, ps, ms, acc, samplesLeft);
        const bool live = (ps.flags & (kInPath | kShadowPending)) != 0 || samplesLeft > 0;
        if (live) {
            Ctr c = {0, 0, 0, 0, 0, 0, 0, 0};
            const int lt = p >> 6, pl = p & 63;
            const int tile = W.tileFirst + lt * W.tileStride;
            const int x = (tile % W.tilesX) * 8 + (pl & 7), y = (tile / W.tilesX) * 8 + (pl >> 3);
            // results of the rays this path put on the queue in its previous step
            Hit h; h.tri = -1; h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.material = 0;
            V3 thr = v3(1.0f);
            if (ps.flags & kInPath) {
                const float* R = W.R + p;
                h.t = R[R_T * 2 * n]; h.u = R[R_U * 2 * n]; h.v = R[R_V * 2 * n];
                h.tri = __builtin_bit_cast(int32_t, R[R_TRI * 2 * n]); h.material = __builtin_bit_cast(int32_t, R[R_MAT * 2 * n]);
                if (COUNT) { c.raysClosest++; c.pops += __builtin_bit_cast(uint32_t, R[R_POPS * 2 * n]); c.boxes += __builtin_bit_cast(uint32_t, R[R_BOXES * 2 * n]); c.tris += __builtin_bit_cast(uint32_t, R[R_TRIS * 2 * n]); if (h.tri >= 0) c.hits++; }
            }
            if (ps.flags & kShadowPending) {
                const float* R = W.R + n + p;
                thr = v3(R[R_THRX * 2 * n], R[R_THRY * 2 * n], R[R_THRZ * 2 * n]);
                if (COUNT) { c.raysShadow++; c.pops += __builtin_bit_cast(uint32_t, R[R_POPS * 2 * n]); c.boxes += __builtin_bit_cast(uint32_t, R[R_BOXES * 2 * n]); c.tris += __builtin_bit_cast(uint32_t, R[R_TRIS * 2 * n]); }
            }
            auto noShadow = [](V3, V3, float) { return v3(1.0f); };
            apply_pending<SIMPLE>(ps, thr, acc);
            if (ps.flags & kInPath) {
                bool done = path_bounce<INTEG, COUNT, true, SIMPLE>(S, ps, ms, h, maxDepth, useMIS, noShadow, c);
                if (!done) done = path_exhausted<INTEG>(ps, maxDepth);
                if (done) {
                    if (ps.flags & kShadowPending) ps.flags |= kFinishPending;      // Li keeps the sum until the last NEE term is in (pt_path.h: PathState)
                    else acc = acc + ps.Li;
                    ps.flags &= ~kInPath;
                }
            }
            while (!(ps.flags & kInPath) && samplesLeft > 0) {
                samplesLeft--;
                path_begin<COUNT>(cam, ps, ms, x, y, c);
                if (path_exhausted<INTEG>(ps, maxDepth)) { acc = acc + ps.Li; ps.flags &= ~kInPath; }
            }
            hasExt = (ps.flags & kInPath) != 0;
            hasShadow = (ps.flags & kShadowPending) != 0;
            wf_store<SIMPLE>(W, p, ps, ms, acc, samplesLeft);
            if (COUNT && W.pathCtr) {
                uint32_t* pc = W.pathCtr + p;
                pc[0] += c.raysClosest; pc[n] += c.raysShadow; pc[2 * n] += c.pops; pc[3 * n] += c.boxes;
                pc[4 * n] += c.tris; pc[5 * n] += c.hits; pc[6 * n] += c.draws; pc[7 * n] += c.iters;
            }
        }
    }
    // append this wave's rays to the queue: ray id p (extension) / n + p (shadow)
    const unsigned long long mE = __ballot(hasExt), mS = __ballot(hasShadow);
    const int nE = __popcll(mE), nS = __popcll(mS);
    if (nE + nS) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&W.qctl[(it & 1) * 2 + 0], (uint32_t)(nE + nS));
        base = __shfl(base, 0, 64);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (hasExt) W.queue[base + __popcll(mE & below)] = (uint32_t)p;
        if (hasShadow) W.queue[base + nE + __popcll(mS & below)] = (uint32_t)(W.n + p);
    }
}

// Persistent traversal with lane-level refill. qctl[(it&1)*2] = number of queued rays,
// qctl[(it&1)*2+1] = fetch cursor. The kernel also clears the OTHER counter pair for the next
// iteration (nothing else touches it while this kernel runs).
// WGW: waves per workgroup — 4 (eight workgroups per CU, 12 KB of scene cache each) or, for scenes in HBM, 16 (two per CU sharing
// a 48 KB copy of the top of the tree: 768 nodes instead of 192, as the megakernel's kernel for such scenes does).
template <bool COUNT, int WGW>
__global__ void __launch_bounds__(64 * WGW) __attribute__((amdgpu_waves_per_eu(8)))
wf_trace_kernel(WfParams W, DeviceScene S, int cacheNodes, int cacheTris, int32_t* spill, int spillPerLane, int it) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wf_smem[];
    // stage the scene cache exactly like the megakernel
    typedef __attribute__((address_space(3))) f4v lds_f4;
    lds_f4* dstN = (lds_f4*)wf_smem;
    lds_f4* dstT = dstN + cacheNodes * 4;
    {
        const f4v* srcN = reinterpret_cast<const f4v*>(S.nodes);
        const f4v* srcT = reinterpret_cast<const f4v*>(S.tris);
        for (int i = threadIdx.x; i < cacheNodes * 4; i += blockDim.x) dstN[i] = srcN[i];
        for (int i = threadIdx.x; i < cacheTris * 3; i += blockDim.x) dstT[i] = srcT[i];
        __syncthreads();
    }
    SceneCache SC; SC.nodes = (lds_cf4*)dstN; SC.nNodes = cacheNodes; SC.tris = (lds_cf4*)dstT; SC.nTris = cacheTris;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cacheBytes = cacheNodes * 64 + cacheTris * 48;
    Stack<kWfStackLds> st;
    st.lds = (lds_i32*)(wf_smem + cacheBytes) + wave * (kWfStackLds * 64) + lane;
    st.spill = spill ? spill + ((size_t)(blockIdx.x * WGW + wave) * spillPerLane) * 64 + lane : nullptr;
    st.sp = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { W.qctl[((it + 1) & 1) * 2 + 0] = 0u; W.qctl[((it + 1) & 1) * 2 + 1] = 0u; }

    const uint32_t total = W.qctl[(it & 1) * 2 + 0];
    uint32_t* cursor = &W.qctl[(it & 1) * 2 + 1];
    const size_t n = W.n;
    Trav<kWfStackLds> tr;
    Ctr c = {0, 0, 0, 0, 0, 0, 0, 0};
    bool active = false;
    // A refill claims exactly the idle lanes' worth of ray ids with one wave-level atomic
    // (kWfChunk == 0). Claiming larger chunks per atomic (kWfChunk > 0) measured SLOWER: one launch
    // only holds a few hundred rays per resident wave, so chunks unbalance the waves.
    uint32_t chunkPos = 0, chunkEnd = 0;
    bool drained = false;           // wave-uniform: global queue exhausted and local chunk empty
    uint32_t ray = 0;
    while (true) {
        // ---- refill idle lanes (when enough are idle to amortise the ray fetch, or all are) ----
        const unsigned long long idle = __ballot(!active);
        const int nIdle = __popcll(idle);
        if (!drained && (nIdle >= kWfRefillLanes || nIdle == 64)) {
            if (chunkPos == chunkEnd) {
                uint32_t base = 0;
                const uint32_t claim = kWfChunk > 0 ? (uint32_t)kWfChunk : (uint32_t)nIdle;
                if (lane == 0) base = atomicAdd(cursor, claim);
                base = __shfl(base, 0, 64);
                chunkPos = base < total ? base : total;
                chunkEnd = (base + claim < total) ? base + claim : total;
                if (chunkPos == chunkEnd) drained = true;
            }
            const uint32_t avail = chunkEnd - chunkPos;
            const uint32_t take = avail < (uint32_t)nIdle ? avail : (uint32_t)nIdle;
            if (!active) {
                const uint32_t k = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                if (k < take) {
                    ray = W.queue[chunkPos + k];
                    const bool sh = ray >= (uint32_t)n;
                    const uint32_t p = sh ? ray - (uint32_t)n : ray;
                    const float* F = W.F + p;
                    V3 o, d; float maxt;
                    if (sh) { o = v3(F[F_SOX * n], F[F_SOY * n], F[F_SOZ * n]); d = v3(F[F_SDX * n], F[F_SDY * n], F[F_SDZ * n]); maxt = F[F_SMAXT * n]; }
                    else { o = v3(F[F_OX * n], F[F_OY * n], F[F_OZ * n]); d = v3(F[F_DX * n], F[F_DY * n], F[F_DZ * n]); maxt = 999999.0f; }
                    c.pops = 0; c.boxes = 0; c.tris = 0;
                    tr.template start<false>(S, st, o, d, maxt, sh, c);
                    active = true;
                }
            }
            chunkPos += take;
        }
        if (__ballot(active) == 0ull) { if (drained) break; else continue; }
        // ---- one traversal step (to and through the next leaf) for every lane that holds a ray ----
        if (active && tr.template step<COUNT>(S, SC, st, c, Keep{W.nodeKeep, W.triKeep})) {
            float* R = W.R + ray;
            if (tr.shadow) { R[R_THRX * 2 * n] = tr.thr.x; R[R_THRY * 2 * n] = tr.thr.y; R[R_THRZ * 2 * n] = tr.thr.z; }
            else {
                R[R_T * 2 * n] = tr.hit.t; R[R_U * 2 * n] = tr.hit.u; R[R_V * 2 * n] = tr.hit.v;
                R[R_TRI * 2 * n] = __builtin_bit_cast(float, tr.hit.tri); R[R_MAT * 2 * n] = __builtin_bit_cast(float, tr.hit.material);
            }
            if (COUNT) { R[R_POPS * 2 * n] = __builtin_bit_cast(float, c.pops); R[R_BOXES * 2 * n] = __builtin_bit_cast(float, c.boxes); R[R_TRIS * 2 * n] = __builtin_bit_cast(float, c.tris); }
            active = false;
        }
    }
}

// per-path counters [8][n] -> the megakernel's per-pixel layout [tile][8][64] (+ totals)
__global__ void __launch_bounds__(256) wf_counters_kernel(WfParams W, uint32_t* pixCounters, unsigned long long* totals) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    for (int k = 0; k < 8; k++) {
        uint32_t v = (p < W.n) ? W.pathCtr[(size_t)k * W.n + p] : 0u;
        if (pixCounters && p < W.n) pixCounters[(size_t)(p >> 6) * 512 + k * 64 + (p & 63)] = v;
        if (totals) {
            unsigned long long s = v;
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0 && s) atomicAdd(&totals[k], s);
        }
    }
}

// ---- host-callable launchers ----------------------------------------------------------------
size_t wf_state_bytes(int n) { return (size_t)n * 4 * ((size_t)F_COUNT + (size_t)U_COUNT + 2 * (size_t)R_COUNT + 2 /*queue*/ ); }

void wf_carve(WfParams& W, void* base) {
    const size_t n = W.n;
    float* f = (float*)base;
    W.F = f; f += (size_t)F_COUNT * n;
    W.U = (uint32_t*)f; f += (size_t)U_COUNT * n;
    W.R = f; f += 2 * (size_t)R_COUNT * n;
    W.queue = (uint32_t*)f;
}

hipError_t launch_wf_init(const WfParams& W, int spp, hipStream_t s) {
    hipLaunchKernelGGL(wf_init_kernel, dim3((W.n + 255) / 256), dim3(256), 0, s, W, spp);
    return hipGetLastError();
}
hipError_t launch_wf_finish(const WfParams& W, hipStream_t s) {
    hipLaunchKernelGGL(wf_finish_kernel, dim3((W.n + 255) / 256), dim3(256), 0, s, W);
    return hipGetLastError();
}
hipError_t launch_wf_counters(const WfParams& W, uint32_t* pixCounters, unsigned long long* totals, hipStream_t s) {
    hipLaunchKernelGGL(wf_counters_kernel, dim3((W.n + 255) / 256), dim3(256), 0, s, W, pixCounters, totals);
    return hipGetLastError();
}
hipError_t launch_wf_logic(int integrator, bool count, bool simple, const WfParams& W, const DeviceScene& S, const CamK& cam, int maxDepth, int useMIS, int it, hipStream_t s) {
    dim3 g((W.n + 255) / 256), b(256);
    if (integrator == 2) {
        if (count) hipLaunchKernelGGL((wf_logic_kernel<2, true>), g, b, 0, s, W, S, cam, maxDepth, useMIS, it);
        else if (simple) hipLaunchKernelGGL((wf_logic_kernel<2, false, true>), g, b, 0, s, W, S, cam, maxDepth, useMIS, it);
        else hipLaunchKernelGGL((wf_logic_kernel<2, false>), g, b, 0, s, W, S, cam, maxDepth, useMIS, it);
    } else {
        if (count) hipLaunchKernelGGL((wf_logic_kernel<0, true>), g, b, 0, s, W, S, cam, maxDepth, useMIS, it);
        else if (simple) hipLaunchKernelGGL((wf_logic_kernel<0, false, true>), g, b, 0, s, W, S, cam, maxDepth, useMIS, it);
        else hipLaunchKernelGGL((wf_logic_kernel<0, false>), g, b, 0, s, W, S, cam, maxDepth, useMIS, it);
    }
    return hipGetLastError();
}
hipError_t launch_wf_trace(bool count, int wgWaves, int blocks, const WfParams& W, const DeviceScene& S, int cacheNodes, int cacheTris, int32_t* spill, int spillPerLane, int it, hipStream_t s) {
    const unsigned lds = (unsigned)((size_t)cacheNodes * 64 + (size_t)cacheTris * 48 + (size_t)wgWaves * kWfStackLds * 256);
#define PT_WF_LAUNCH(C, G) do { if (lds > 65536u) { hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&wf_trace_kernel<C, G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                                                    if (e_ != hipSuccess) return e_; } \
                                hipLaunchKernelGGL((wf_trace_kernel<C, G>), dim3(blocks), dim3(64 * G), lds, s, W, S, cacheNodes, cacheTris, spill, spillPerLane, it); } while (0)
    if (wgWaves == 16) { if (count) PT_WF_LAUNCH(true, 16); else PT_WF_LAUNCH(false, 16); }
    else { if (count) PT_WF_LAUNCH(true, 4); else PT_WF_LAUNCH(false, 4); }
#undef PT_WF_LAUNCH
    return hipGetLastError();
}

}  // namespace pt
