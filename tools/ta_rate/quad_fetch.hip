// Micro-benchmark behind DESIGN.md §6 "who pays for a divergent 64-byte node fetch".
// Every lane chases pointers through an array of 64-byte records (the PNode size), 6 waves per SIMD:
//   mode 0  each lane reads its own record with four 16-byte loads (what the traversal does)
//   mode 1  the four lanes of a quad read each of their four records together, one 16-byte
//           quarter per lane (a quad's load is one contiguous 64-byte line); no transpose,
//           the chain only needs the quarter the lane already holds
//   mode 2  as 1, plus the 4x4 transpose of the quarters inside the quad (DPP) so every lane
//           ends with its whole record, as the traversal would need
//   mode 3  each lane reads ONE 16-byte quarter of its own record (a quarter of mode 0's requests)
// build: hipcc -O3 --offload-arch=gfx950 quad_fetch.hip -o quad_fetch ; run: ./quad_fetch [MB] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

struct alignas(64) Rec { uint4 q[4]; };

__device__ inline uint32_t fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

template <int CTRL> __device__ inline uint32_t qperm(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL> __device__ inline uint4 qperm4(uint4 v) {
    return make_uint4(qperm<CTRL>(v.x), qperm<CTRL>(v.y), qperm<CTRL>(v.z), qperm<CTRL>(v.w));
}
__device__ inline uint4 sel(bool c, uint4 a, uint4 b) {
    return make_uint4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}

template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 6)))
void chase(const Rec* __restrict__ recs, uint32_t mask, int iters, uint32_t* out) {
    const uint32_t lane = threadIdx.x, ql = lane & 3;
    uint32_t idx = (blockIdx.x * 64u + lane) * 2654435761u & mask, acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t v;
        if (MODE == 0) {
            const uint4* p = recs[idx].q;
            uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            v = fold(a) ^ fold(b) + fold(c) ^ fold(d);
        } else if (MODE == 3) {
            v = fold(recs[idx].q[ql]);
        } else {
            uint32_t i0 = qperm<0x00>(idx), i1 = qperm<0x55>(idx), i2 = qperm<0xAA>(idx), i3 = qperm<0xFF>(idx);
            uint4 r0 = recs[i0].q[ql], r1 = recs[i1].q[ql], r2 = recs[i2].q[ql], r3 = recs[i3].q[ql];
            if (MODE == 2) {
                const bool b0 = ql & 1, b1 = ql & 2;
                // xor-1 exchange inside pairs (0,1) and (2,3), then xor-2 inside (0,2) and (1,3)
                uint4 s = sel(b0, r0, r1), t = qperm4<0xB1>(s);
                r0 = sel(b0, t, r0); r1 = sel(b0, r1, t);
                s = sel(b0, r2, r3); t = qperm4<0xB1>(s);
                r2 = sel(b0, t, r2); r3 = sel(b0, r3, t);
                s = sel(b1, r0, r2); t = qperm4<0x4E>(s);
                r0 = sel(b1, t, r0); r2 = sel(b1, r2, t);
                s = sel(b1, r1, r3); t = qperm4<0x4E>(s);
                r1 = sel(b1, t, r1); r3 = sel(b1, r3, t);
                v = fold(r0) ^ fold(r1) + fold(r2) ^ fold(r3);
            } else {
                uint4 mine = ql == 0 ? r0 : ql == 1 ? r1 : ql == 2 ? r2 : r3;
                v = fold(mine) + (fold(r0) ^ fold(r1) ^ fold(r2) ^ fold(r3)) * 0u;
            }
        }
        acc += v;
        idx = (v * 2246822519u + acc) & mask;
    }
    out[blockIdx.x * 64 + lane] = acc;
}

int main(int argc, char** argv) {
    size_t mb = argc > 1 ? atoi(argv[1]) : 16;
    int iters = argc > 2 ? atoi(argv[2]) : 2000;
    size_t n = mb * 1024 * 1024 / sizeof(Rec);
    uint32_t mask = 1; while ((size_t)mask * 2 <= n) mask *= 2; mask -= 1;
    std::vector<Rec> h(mask + 1ull);
    uint32_t s = 12345;
    for (auto& r : h) for (auto& q : r.q) { s = s * 1664525u + 1013904223u; q.x = s; s = s * 1664525u + 1013904223u; q.y = s;
                                             s = s * 1664525u + 1013904223u; q.z = s; s = s * 1664525u + 1013904223u; q.w = s; }
    Rec* d; uint32_t* o;
    const int blocks = 256 * 4 * 6 * 4;
    hipMalloc(&d, h.size() * sizeof(Rec)); hipMalloc(&o, blocks * 64 * 4);
    hipMemcpy(d, h.data(), h.size() * sizeof(Rec), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) chase<0><<<blocks, 64>>>(d, mask, iters, o);
            if (mode == 1) chase<1><<<blocks, 64>>>(d, mask, iters, o);
            if (mode == 2) chase<2><<<blocks, 64>>>(d, mask, iters, o);
            if (mode == 3) chase<3><<<blocks, 64>>>(d, mask, iters, o);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
        double fetches = (double)blocks * 64 * iters;
        printf("{\"array_mb\": %zu, \"mode\": %d, \"ms\": %.3f, \"Gfetch_per_s\": %.2f, \"TB_per_s_64B\": %.2f}\n",
               (size_t)((mask + 1ull) * 64 >> 20), mode, best, fetches / best * 1e-6, fetches * 64 / best * 1e-9);
    }
    return 0;
}
