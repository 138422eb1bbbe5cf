// pt_params.h — kernel parameter blocks and host-callable launchers shared by pt_kernels.hip
// (definitions) and pt_api.hip (C ABI).
#pragma once
#include "pt_device.h"

namespace pt {

// Build-time knobs (A/B-able with -D...):
#ifndef PT_STACK_LDS
#define PT_STACK_LDS 16           // LDS traversal-stack entries per lane (256 B each per wave); deeper trees spill to global
#endif
#ifndef PT_MIN_WAVES
#define PT_MIN_WAVES 4            // amdgpu_waves_per_eu for the megakernel: caps VGPRs at 128 (0 = unconstrained)
#endif
#ifndef PT_CACHE_BYTES
#define PT_CACHE_BYTES 12288      // LDS bytes per workgroup for the scene cache (top PNodes / all PTris)
#endif
#ifndef PT_SPEC
#define PT_SPEC 0                 // resumable traversal of the REFILL kernels: 0 trace_resume, 1 trace_resume_spec (speculative descent), 2 both (A/B)
#endif
constexpr int kStackLds = PT_STACK_LDS;
#ifndef PT_WAVES_HBM
#define PT_WAVES_HBM 6             // waves per SIMD of the instantiation for scenes that do not fit the LDS cache (0 = use the 4-wave kernel)
#endif
constexpr int kWavesHbm = PT_WAVES_HBM > 0 ? PT_WAVES_HBM : 6;
#ifndef PT_STACK_LDS_HBM
#define PT_STACK_LDS_HBM 8
#endif
#ifndef PT_STACK_LDS_HBM_GEN
#define PT_STACK_LDS_HBM_GEN 12
#endif
constexpr int kStackLdsHbmGen = PT_STACK_LDS_HBM_GEN;     // ... of the GENERAL bounce's kernel (megakernel_hbm, six waves per SIMD): twelve entries measure +2.5 % on the glass blob over
                                                           // eight (fewer stack entries in the global spill area for 64 cached nodes less); the SIMPLE kernel is flat from 6 to 12 (profiles/r03_ab_stack_rows_hbm.log)
constexpr int kStackLdsHbm = PT_STACK_LDS_HBM;    // its LDS stack entries per lane: 8 KB + 4 KB medium stacks + 12 KB cache = 24 KB, six workgroups per CU
constexpr int kMediumMax = 16;    // mediumStack[16], deviceCode.cu:306
constexpr int kRefillKeepSmall = 10;  // "refill_keep" of the 4-wave kernel for scenes in HBM while the option is unset (sixteenths of the busy lanes)
constexpr int kQueueHeader = 16;     // ints of the tile queue before its slots (pt_megakernel.h: the queue's words)
constexpr int kStackFlat2Rows = 24;   // 256-byte rows of LDS per wave of the pair form of FLAT (pt_trace.h: trace_pair_flat's scratch)
constexpr int kCacheBytes = PT_CACHE_BYTES;
// LDS-resident scenes also stage what the bounce reads — hit attributes (80 B per triangle), materials (96 B each) and lights
// (64 B each) — when those fit this budget, so that the logic step has no global load (Cornell: 36 x 80 + 24 x 96 + 2 x 64 =
// 5.3 KB; a bounce otherwise waits on five dependent L2 round trips at 4 waves per SIMD). 0 = never (A/B).
#ifndef PT_ATTR_CACHE_BYTES
#define PT_ATTR_CACHE_BYTES 8192
#endif
constexpr int kAttrCacheBytes = PT_ATTR_CACHE_BYTES;
// The kernel for scenes in HBM runs in workgroups of PT_WG_WAVES_HBM waves (default 12: two workgroups per CU at 6
// waves per SIMD) so that its waves SHARE one large LDS copy of the top of the tree instead of six small ones:
// 80 KB per workgroup - 12 x (2 KB stack + 1 KB medium stack) = 44 KB = the first 704 PNodes, which take 45 % of all
// node visits on the 263 k-triangle scene (the first 192 of the 12 KB cache: 32 %). 4 = the old shape.
#ifndef PT_WG_WAVES_HBM
#define PT_WG_WAVES_HBM 12
#endif
constexpr int kWgWavesHbm = (PT_WAVES_HBM > 0 && (kWavesHbm * 4) % PT_WG_WAVES_HBM == 0 && PT_WG_WAVES_HBM <= 16) ? PT_WG_WAVES_HBM : 4;
constexpr int kCacheBytesHbm = kWgWavesHbm == 4 ? kCacheBytes
                                                : ((160 * 1024) / ((kWavesHbm * 4) / kWgWavesHbm) - kWgWavesHbm * (kStackLdsHbmGen * 256 + kMediumMax * 64)) / 64 * 64;

// The SIMPLE instantiation of the kernel for scenes in HBM (pt_path.h: diffuse-only scenes) is a third of the generic
// kernel's code and holds less state: it runs best at 8 waves per SIMD (64 VGPRs) in workgroups of 16 waves — one LDS copy
// of the top of the tree per 16 waves, and no medium stacks, so the copy is 48 KB (768 PNodes). The generic kernel is 13 %
// SLOWER at 8 (spills) and stays at PT_WAVES_HBM (profiles/r02_ab_waves_simple.log).
#ifndef PT_WAVES_HBM_SIMPLE
#define PT_WAVES_HBM_SIMPLE 8
#endif
#ifndef PT_WG_WAVES_HBM_SIMPLE
#define PT_WG_WAVES_HBM_SIMPLE 16
#endif
constexpr int kWavesHbmSimple = PT_WAVES_HBM_SIMPLE, kWgWavesHbmSimple = PT_WG_WAVES_HBM_SIMPLE;
static_assert((kWavesHbmSimple * 4) % kWgWavesHbmSimple == 0 && kWgWavesHbmSimple <= 16, "the workgroups of the SIMPLE kernel must tile a CU's wave slots");
constexpr int kWgPerCuHbmSimple = (kWavesHbmSimple * 4) / kWgWavesHbmSimple;        // a CU's 160 KB of LDS divided among them
constexpr int kCacheBytesHbmSimple = ((160 * 1024) / kWgPerCuHbmSimple - kWgWavesHbmSimple * (kStackLdsHbm * 256)) / 64 * 64;

struct KParams {
    DeviceScene S;
    CamK cam;
    int w, h, spp, maxDepth, useMIS;
    int tileFirst, tileStride, tileCount, tilesX;
    int cacheNodes, cacheTris;     // scene-cache extent (PNodes / PTris staged in LDS per workgroup)
    int cacheAttrs, cacheMats, cacheLights;   // ONCHIP kernels: PAttr / PMat / PLight records staged behind the stacks (all or none; 0 = read from global memory)
    int wide;                      // 1: megakernel_hbm_wide — the SIMPLE kernel for scenes in HBM on the 4-wide collapsed tree `wnodes`
    const WNode* wnodes;
    int compact;                   // 1: megakernel_hbm_compact — the SIMPLE kernel for scenes in HBM on 32-byte quantised nodes (pt_trace.h: trace_resume_q)
    const QNode* qnodes; const void* leafBox; const int32_t* mids;
    int gnodeFrom;                 // counting launches: node indices from here on count as global fetches — the scene-cache extent of the TIMED instantiation for this launch
    int nLeaves;                   // FLAT kernels: PLeaf records in global memory, read through the scalar cache (0 = none: the lockstep node walk)
    const PLeaf* leaves;
    int cull;                      // opt-in box culling (pt_trace.h: CULL); only the kernel for scenes in HBM has the instantiation
    int triKeep;                   // ... and its triangle loop once no more than entered * triKeep / 16 lanes still have triangles in their leaf
    int nodeKeep;                  // a wave leaves its node loop once no more than active * nodeKeep / 16 lanes are still descending (pt_trace.h); 0 = when none is
    int lean;                      // 1: the LEAN instantiation of the generic bounce — no MAT_LEAF triangle, no textures (pt_shade.h); timed launches of the REFILL / pair kernels
    int simple;                    // 1 (with flat): the SIMPLE instantiation — every triangle an untextured, non-boundary MAT_DIFFUSE in non-absorbing air (pt_path.h)
    int flat;                      // 1: the FLAT instantiation (pt_trace.h: trace_closest_flat) — LDS-resident scenes with at most 64 internal nodes / 64 triangles
    int refill;                    // 1: the REFILL instantiation (pt_trace.h: trace_resume) — finished lanes shade and come back while the rest keep tracing
    int refillKeep;                // the wave leaves the traversal when no more than busy * refillKeep / 16 lanes are still tracing
    int spec;                      // REFILL kernels: 0 trace_resume, 1 speculative descent for closest-hit rays (pt_trace.h: trace_resume_spec), 2 also for shadow rays
    int wgWaves;                   // waves per workgroup of this launch (4, or kWgWavesHbm for megakernel_hbm)
    int wavesPerSimd;              // PT_MIN_WAVES (megakernel), kWavesHbm (megakernel_hbm) or kWavesHbmSimple (its SIMPLE instantiation)
    int hbm;                       // 1: a megakernel_hbm launch (8-entry LDS stacks, the spill area laid out for them)
    int onchip;                    // 1: every PNode / PTri is in the LDS cache and the stack fits LDS -> ONCHIP kernels
    int xcdBands;                  // 1: workgroups of one XCD take a contiguous run of tiles (one L2 per XCD, MI355X_MICROARCH.md)
    uint32_t* rng;                 // [tile][6][64]
    float4* out;                   // [tile][64], += semantics
    uint32_t* pixCounters;         // [tile][8][64] or null
    unsigned long long* totals;    // 8 x u64 or null
    int32_t* spill;                // [wave slot][entry][64] or null
    int gridBlocks;                // persistent waves: workgroups that fill the chip (numCU x resident workgroups per CU)
    int* queue;                    // persistent waves: tile queue (pt_kernels.hip: queue_pop / queue_push); null = one tile per wave
    int queueMask;                 // ring capacity - 1 (power of two >= tileCount)
    unsigned long long queueTimeout;   // ticks of the 100 MHz steady counter a wait on the queue may see no progress before it raises q[3] (default 30 s); the kernels read it from the queue header, where queue_init_kernel puts it
    int sliceIters;                // bounce iterations a wave keeps a tile once no fresh tile is left; 0 = until it is finished
    int sliceAlways;               // 1: slices from the first tile on (round-robin over all tiles), 0: only once no fresh tile is left
    int* left;                     // [tile][64] samples left per pixel of a yielded tile
    int schedMask;                 // the wave looks at the queue / its priority every schedMask + 1 iterations (31)
    int lptPrio;                   // longest-remaining-first issue priority once no fresh tile is left
};

struct TileSpan { int first, stride, count, tilesX; };

// ---- wavefront variant (pt_wavefront.hip) ----
#ifndef PT_WF_REFILL_LANES
#define PT_WF_REFILL_LANES 32     // a wave refills from the ray queue once this many lanes are idle
#endif
#ifndef PT_WF_STACK_LDS
#define PT_WF_STACK_LDS 8         // LDS stack entries per lane in the (register-light) trace kernel
#endif
#ifndef PT_WF_CACHE_BYTES
#define PT_WF_CACHE_BYTES 12288   // its scene-cache budget: 8 KB stacks + 12 KB cache = 8 workgroups per CU
#endif
#ifndef PT_WF_BLOCKS_PER_CU
#define PT_WF_BLOCKS_PER_CU 8     // persistent grid = CUs x this
#endif
#ifndef PT_WF_CHUNK
#define PT_WF_CHUNK 0             // ray ids claimed per atomic; 0 = exactly the idle lanes (fastest measured)
#endif
constexpr int kWfChunk = PT_WF_CHUNK;
constexpr int kWfRefillLanes = PT_WF_REFILL_LANES;
constexpr int kWfStackLds = PT_WF_STACK_LDS;
constexpr int kWfCacheBytes = PT_WF_CACHE_BYTES;
constexpr int kWfBlocksPerCU = PT_WF_BLOCKS_PER_CU;
struct WfParams {
    float* F;            // [F_COUNT][n] path state, float fields
    uint32_t* U;         // [U_COUNT][n] path state, integer fields
    float* R;            // [R_COUNT][2n] per-ray results (extension ray of path p: p; shadow ray: n + p)
    uint32_t* queue;     // [2n] compacted ray ids of the current iteration
    uint32_t* qctl;      // {count, cursor} x 2 (ping-pong by iteration parity)
    uint32_t* pathCtr;   // [8][n] per-path work counters (COUNT builds) or null
    uint32_t* rng;       // [tile][6][64] as the megakernel
    float4* out;         // [tile][64] tile buffer
    int n, w, h, tileFirst, tileStride, tilesX;
    int nodeKeep, triKeep;   // loop exits of the trace kernel (pt_trace.h: LoopExit), as KParams
};
size_t wf_state_bytes(int n);
void wf_carve(WfParams& W, void* base);
hipError_t launch_wf_init(const WfParams& W, int spp, hipStream_t s);
hipError_t launch_wf_finish(const WfParams& W, hipStream_t s);
hipError_t launch_wf_logic(int integrator, bool count, bool simple, const WfParams& W, const DeviceScene& S, const CamK& cam, int maxDepth, int useMIS, int it, hipStream_t s);
hipError_t launch_wf_trace(bool count, int wgWaves, int blocks, const WfParams& W, const DeviceScene& S, int cacheNodes, int cacheTris, int32_t* spill, int spillPerLane, int it, hipStream_t s);
hipError_t launch_wf_counters(const WfParams& W, uint32_t* pixCounters, unsigned long long* totals, hipStream_t s);

// Launchers (asynchronous on `stream`); defined in pt_kernels.hip.
hipError_t launch_rng_init(const uint32_t* jump, unsigned long long seed, int w, int h, TileSpan t, uint32_t* rng, hipStream_t stream);
// syncShadow: trace the NEE shadow ray inside the bounce (needed only for materials without a dispatch arm)
hipError_t launch_megakernel(int integrator, bool count, bool syncShadow, const KParams& P, hipStream_t stream);
hipError_t launch_untile(int w, int h, TileSpan t, const float4* tiles, float4* colors, hipStream_t stream);
hipError_t launch_tile(int w, int h, TileSpan t, const float4* colors, float4* tiles, hipStream_t stream);
hipError_t launch_probe_rng(const uint32_t* jump, unsigned long long seed, int n, const uint32_t* subseq, int nDraws,
                            uint32_t* outState, uint32_t* outU32, float* outUni, hipStream_t stream);
hipError_t launch_probe_math(int n, const float* x, float* s, float* c, float* e, float* rs, float* p5, hipStream_t stream);
hipError_t launch_probe_camera(const uint32_t* state6, const CamK& cam, int n, const int* xy, float* out, hipStream_t stream);
hipError_t launch_probe_rcp_exhaustive(unsigned long long* out3, uint32_t* firstBad, hipStream_t stream);
hipError_t launch_probe_closest(const DeviceScene& S, int n, const float* rays, int32_t* outI, float* outF,
                                unsigned long long* totals, int32_t* spill, hipStream_t stream);
hipError_t launch_probe_shadow(const DeviceScene& S, int n, const float* rays, const float* maxT, float* outF,
                               unsigned long long* totals, int32_t* spill, hipStream_t stream);
hipError_t launch_probe_bsdf_sample(const DeviceScene& S, int n, const int* material, const float* wi3, const int* backface, float etaI,
                                    const uint32_t* state6, float* out8, hipStream_t stream);
hipError_t launch_probe_bsdf_eval(const DeviceScene& S, int n, const int* material, const float* wi3, const float* wo3, float etaI,
                                  float* out4, hipStream_t stream);
// waves a probe_closest/shadow launch of n rays uses (spill sizing)
inline int probe_trace_blocks(int n) { return (n + 63) / 64; }
inline int megakernel_blocks(int tileCount, int wgWaves = 4) { return (tileCount + wgWaves - 1) / wgWaves; }
inline size_t attr_cache_bytes(int nAttrs, int nMats, int nLights) { return (size_t)nAttrs * 80 + (size_t)nMats * 96 + (size_t)nLights * 64; }
inline size_t megakernel_lds_bytes(int cacheNodes, int cacheTris, int stackEntries = kStackLds, int wgWaves = 4, size_t attrBytes = 0, bool mediumStacks = true) {
    return (size_t)cacheNodes * 64 + (size_t)cacheTris * 48 + (size_t)wgWaves * ((size_t)stackEntries * 256 + (mediumStacks ? (size_t)kMediumMax * 64 : 0)) + attrBytes;
}

}  // namespace pt
