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
#include "pt_megakernel.h"

namespace pt {



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

__global__ void queue_init_kernel(int* q, int mask, int nTiles, unsigned long long timeout) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { q[0] = 0; q[1] = nTiles; q[2] = 0; q[3] = 0; q[4] = 0; q[5] = 0; q[6] = 0; q[7] = 0; ((unsigned long long*)q)[4] = timeout; }
    if (i <= mask)
        ((unsigned long long*)(q + kQueueHeader))[i] = i < nTiles ? (((unsigned long long)(unsigned)(i | kFreshBit) << 32) | (unsigned long long)(i + 1)) : (unsigned long long)i;
}

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

// Every binary32 bit pattern through rcp_exact (pt_device.h) against the IEEE division sequence it replaces in
// triangleIntersect's `f = 1.0 / a` (integratorUtilities.cuh:22), in aabbIntersect's 1 / dir (:50-55) and in normalize's
// rsqrtf (util.cuh:129). rcp_exact's fast arm is v_rcp_f32 + one Newton step; that it equals the correctly rounded quotient
// for 1e-12 <= |a| <= 1e30 is a property of THIS GPU's v_rcp_f32 table, so every box that renders re-proves it (tests, -m gpu).
// out[0]: inputs where rcp_exact(a) differs from 1.0f / a (the contract: 0), out[1]: inputs inside the fast arm's range,
// out[2]: inputs OUTSIDE that range where the bare v_rcp + Newton sequence would differ (why the range guard exists; > 0).
__global__ void __launch_bounds__(256) probe_rcp_exhaustive_kernel(unsigned long long* out, uint32_t* firstBad) {
    const unsigned long long i0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad = 0, inRange = 0, badOutside = 0;
    for (unsigned long long i = i0; i < (1ull << 32); i += stride) {
        const uint32_t u = (uint32_t)i;
        const float a = __builtin_bit_cast(float, u);
        const float ref = 1.0f / a;                                   // v_div_scale / v_div_fmas / v_div_fixup: correctly rounded, denormals kept
        const float got = rcp_exact(a);
        const uint32_t rb = __builtin_bit_cast(uint32_t, ref), gb = __builtin_bit_cast(uint32_t, got);
        const bool same = rb == gb || (ref != ref && got != got);
        if (!same) { bad++; atomicMin(firstBad, u); }
        const float m = __builtin_fabsf(a);
        if (m >= 1e-12f && m <= 1.0e30f) inRange++;
        else {
            const float r0 = __builtin_amdgcn_rcpf(a);
            const float bare = __builtin_fmaf(r0, __builtin_fmaf(-a, r0, 1.0f), r0);
            if (!(__builtin_bit_cast(uint32_t, bare) == rb || (bare != bare && ref != ref))) badOutside++;
        }
    }
    for (int off = 32; off > 0; off >>= 1) { bad += __shfl_down(bad, off, 64); inRange += __shfl_down(inRange, off, 64); badOutside += __shfl_down(badOutside, off, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], bad); atomicAdd(&out[1], inRange); atomicAdd(&out[2], badOutside); }
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

// The megakernel instantiations live in two translation units with different code-generation flags (pt_megakernel.h):
// scenes that live in LDS (ONCHIP kernels) and scenes in HBM (megakernel_hbm and the general 4-wave kernel).
hipError_t launch_megakernel_lds(int integrator, bool count, const KParams& P, dim3 grid, dim3 block, unsigned lds, hipStream_t stream);
hipError_t launch_megakernel_hbm(int integrator, bool count, bool syncShadow, bool hbm, const KParams& P, dim3 grid, dim3 block, unsigned lds, hipStream_t stream);

hipError_t launch_megakernel(int integrator, bool count, bool syncShadow, const KParams& P, hipStream_t stream) {
    if (P.tileCount <= 0) return hipSuccess;
    int nBlocks = megakernel_blocks(P.tileCount, P.wgWaves);
    if (P.queue && P.gridBlocks > 0) {
        nBlocks = std::min(nBlocks, P.gridBlocks);
        hipLaunchKernelGGL(queue_init_kernel, dim3((P.queueMask + 256) / 256), dim3(256), 0, stream, P.queue, P.queueMask, P.tileCount, P.queueTimeout);
    }
    dim3 grid(nBlocks), block(64 * P.wgWaves);
    const bool hbm = P.hbm != 0;                               // chosen by the host together with the spill layout
    const bool flat2 = P.flat == 3 && integrator == 0 && !count;
    const bool ldsScene = P.onchip && !hbm && (integrator == 2 || syncShadow || flat2);
    const bool simpleKernel = P.simple && !count && ((ldsScene && P.flat) || (!ldsScene && P.refill && !P.cull));      // the instantiations without medium stacks
    const unsigned lds = (unsigned)megakernel_lds_bytes(P.cacheNodes, P.cacheTris, hbm ? ((P.simple && P.refill && !count) ? kStackLdsHbm : kStackLdsHbmGen) : (flat2 ? kStackFlat2 : kStackLds), P.wgWaves,
                                                        ldsScene ? attr_cache_bytes(P.cacheAttrs, P.cacheMats, P.cacheLights) : 0, !simpleKernel);
    if (ldsScene) return launch_megakernel_lds(integrator, count, P, grid, block, lds, stream);
    return launch_megakernel_hbm(integrator, count, syncShadow, hbm, P, grid, block, lds, stream);
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
hipError_t launch_probe_rcp_exhaustive(unsigned long long* out3, uint32_t* firstBad, hipStream_t stream) {
    hipLaunchKernelGGL(probe_rcp_exhaustive_kernel, dim3(8192), dim3(256), 0, stream, out3, firstBad);
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
