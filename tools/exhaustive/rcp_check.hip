// Exhaustive check (all 2^32 binary32 inputs) of cheaper sequences against the compiler's IEEE
// division, on the hardware they would run on: 1.0f / a and x / kPi. Prints mismatch counts.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math rcp_check.hip -o rcp_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ inline float rcp_candidate(float a) {
    float r0 = __builtin_amdgcn_rcpf(a);
    float e = __builtin_fmaf(-a, r0, 1.0f);
    return __builtin_fmaf(r0, e, r0);
}
__device__ inline float rcp_candidate2(float a) {           // two refinement steps
    float r0 = __builtin_amdgcn_rcpf(a);
    float e = __builtin_fmaf(-a, r0, 1.0f);
    float r1 = __builtin_fmaf(r0, e, r0);
    float e1 = __builtin_fmaf(-a, r1, 1.0f);
    return __builtin_fmaf(r1, e1, r1);
}
__device__ inline float divpi_candidate(float x) {
    const float pi = 3.14159265358979323846f, c = 1.0f / 3.14159265358979323846f;
    float q = x * c;
    float r = __builtin_fmaf(-q, pi, x);
    return __builtin_fmaf(r, c, q);
}
__device__ inline bool same(float a, float b) {
    unsigned x, y; __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4);
    return x == y || (a != a && b != b);
}
__global__ void check(unsigned long long* bad, unsigned* firstBad) {
    unsigned long long i0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    for (unsigned long long i = i0; i < (1ull << 32); i += stride) {
        unsigned u = (unsigned)i; float a; __builtin_memcpy(&a, &u, 4);
        float ref = 1.0f / a;
        bool inRange = __builtin_fabsf(a) >= 1e-12f && __builtin_fabsf(a) <= 1.0e30f;      // moller_trumbore rejects |a| < 1e-12; above 1e30 a guard would take the IEEE path
        if (!same(ref, rcp_candidate(a))) { b0++; if (inRange) { b1++; atomicMin(&firstBad[0], u); } }
        if (inRange && !same(ref, rcp_candidate2(a))) { b2++; atomicMin(&firstBad[1], u); }
        const float pi = 3.14159265358979323846f;
        bool mid = __builtin_fabsf(a) >= 1e-30f && __builtin_fabsf(a) <= 1.0e30f;
        if (mid && !same(a / pi, divpi_candidate(a))) { b3++; atomicMin(&firstBad[2], u); }
    }
    atomicAdd(&bad[0], b0); atomicAdd(&bad[1], b1); atomicAdd(&bad[2], b2); atomicAdd(&bad[3], b3);
}
int main() {
    unsigned long long* bad; unsigned* fb;
    hipMalloc(&bad, 32); hipMalloc(&fb, 16);
    hipMemset(bad, 0, 32); hipMemset(fb, 0xff, 16);
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, bad, fb);
    unsigned long long h[4]; unsigned f[4];
    hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost); hipMemcpy(f, fb, 16, hipMemcpyDeviceToHost);
    printf("1/a one-step: %llu mismatches overall, %llu with 1e-12 <= |a| <= 1e30 (first 0x%08x)\n", h[0], h[1], f[0]);
    printf("1/a two-step: %llu mismatches in range (first 0x%08x)\n", h[2], f[1]);
    printf("x/pi  fma-refined: %llu mismatches with 1e-30 <= |x| <= 1e30 (first 0x%08x)\n", h[3], f[2]);
    return 0;
}
