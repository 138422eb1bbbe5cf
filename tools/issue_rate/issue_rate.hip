// Micro-benchmark: what one CU issues per clock of scalar (SALU), vector (VALU) and exec-mask / branch instructions at
// 1, 2, 4, 8 waves per SIMD — the traversal loop for scenes in HBM issues 0.8-0.9 scalar instructions per vector one
// (profiles/r03_v3_pmc_blob.json), and a CU has ONE scalar unit for its four SIMDs (MI355X_MICROARCH.md).
//   mode 0  64 independent s_add_u32 per iteration
//   mode 1  64 independent v_add_u32 per iteration
//   mode 2  32 + 32 interleaved (do they issue side by side?)
//   mode 3  16 x { s_and_saveexec_b64 ; s_or_b64 exec } pairs around one v_add (a divergent `if` without the skip branch)
//   mode 4  16 x the same with the compiler's `s_cbranch_execz` skip (never taken)
//   mode 5  65 v_add + 63 s_add interleaved (the traversal loop's mix)
//   modes 6-9  48 v_add + 16 of {ds_read_b32, s_waitcnt with nothing pending, s_nop 0, s_cbranch to the next instruction}
// build: hipcc -O3 --offload-arch=gfx950 issue_rate.hip -o issue_rate ; run: ./issue_rate [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int MODE>
__global__ __launch_bounds__(64) void spin(int iters, uint32_t* out) {
    uint32_t v = threadIdx.x, s = blockIdx.x;
    if (MODE == 6) asm volatile("v_lshlrev_b32 v24, 2, %0" :: "v"(v) : "v24");
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) asm volatile(REP16("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n") ::: "s20", "s21", "s22", "s23", "scc");
        if (MODE == 1) asm volatile(REP16("v_add_u32 v20, v20, 1\n v_add_u32 v21, v21, 1\n v_add_u32 v22, v22, 1\n v_add_u32 v23, v23, 1\n") ::: "v20", "v21", "v22", "v23");
        if (MODE == 2) asm volatile(REP16("s_add_u32 s20, s20, 1\n v_add_u32 v20, v20, 1\n s_add_u32 s21, s21, 1\n v_add_u32 v21, v21, 1\n") ::: "s20", "s21", "v20", "v21", "scc");
        if (MODE == 3) asm volatile(REP16("v_cmp_eq_u32 vcc, v20, v20\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 v21, v21, 1\n s_or_b64 exec, exec, s[20:21]\n") ::: "s20", "s21", "v21", "vcc", "scc");
        if (MODE == 4) asm volatile(REP16("v_cmp_eq_u32 vcc, v20, v20\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1\n v_add_u32 v21, v21, 1\n s_or_b64 exec, exec, s[20:21]\n") ::: "s20", "s21", "v21", "vcc", "scc");
        if (MODE == 5) asm volatile(REP4("v_add_u32 v20, v20, 1\n s_add_u32 s20, s20, 1\n v_add_u32 v21, v21, 1\n s_add_u32 s21, s21, 1\n v_add_u32 v22, v22, 1\n s_add_u32 s22, s22, 1\n v_add_u32 v23, v23, 1\n s_add_u32 s23, s23, 1\n"
                                         "v_add_u32 v20, v20, 1\n s_add_u32 s20, s20, 1\n v_add_u32 v21, v21, 1\n s_add_u32 s21, s21, 1\n v_add_u32 v22, v22, 1\n s_add_u32 s22, s22, 1\n v_add_u32 v23, v23, 1\n s_add_u32 s23, s23, 1\n"
                                         "v_add_u32 v20, v20, 1\n s_add_u32 s20, s20, 1\n v_add_u32 v21, v21, 1\n s_add_u32 s21, s21, 1\n v_add_u32 v22, v22, 1\n s_add_u32 s22, s22, 1\n v_add_u32 v23, v23, 1\n s_add_u32 s23, s23, 1\n"
                                         "v_add_u32 v20, v20, 1\n s_add_u32 s20, s20, 1\n v_add_u32 v21, v21, 1\n s_add_u32 s21, s21, 1\n v_add_u32 v22, v22, 1\n s_add_u32 s22, s22, 1\n v_add_u32 v23, v23, 1\n v_add_u32 v23, v23, 1\n") ::: "s20", "s21", "s22", "s23", "v20", "v21", "v22", "v23", "scc");
        if (MODE == 6) asm volatile(REP16("v_add_u32 v20, v20, 1\n v_add_u32 v21, v21, 1\n v_add_u32 v22, v22, 1\n ds_read_b32 v23, v24\n") "s_waitcnt lgkmcnt(0)\n" ::: "v20", "v21", "v22", "v23", "memory");
        if (MODE == 7) asm volatile(REP16("v_add_u32 v20, v20, 1\n v_add_u32 v21, v21, 1\n v_add_u32 v22, v22, 1\n s_waitcnt vmcnt(0) lgkmcnt(0)\n") ::: "v20", "v21", "v22");
        if (MODE == 8) asm volatile(REP16("v_add_u32 v20, v20, 1\n v_add_u32 v21, v21, 1\n v_add_u32 v22, v22, 1\n s_nop 0\n") ::: "v20", "v21", "v22");
        if (MODE == 9) asm volatile(REP16("v_add_u32 v20, v20, 1\n v_add_u32 v21, v21, 1\n v_add_u32 v22, v22, 1\n s_cbranch_scc1 0\n") ::: "v20", "v21", "v22");
    }
    if (iters < 0) out[blockIdx.x * 64 + threadIdx.x] = v + s;
}

template <int MODE>
static void run(int iters, uint32_t* d, int cus, const char* what, int instPerIter) {
    for (int w : {1, 2, 4, 6, 8}) {
        const int blocks = cus * 4 * w;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(spin<MODE>, dim3(blocks), dim3(64), 0, 0, iters / 8, d);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(spin<MODE>, dim3(blocks), dim3(64), 0, 0, iters, d);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.0f;
        hipEventElapsedTime(&ms, e0, e1);
        const double inst = (double)blocks * iters * instPerIter;
        printf("%-44s waves/SIMD %d  %8.3f ms  %6.3f inst/clk/CU (2.4 GHz)  %5.2f clk per inst per wave\n", what, w, ms,
               inst / (ms * 1e-3) / cus / 2.4e9, ms * 1e-3 * 2.4e9 / ((double)iters * instPerIter));
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    uint32_t* d;
    hipMalloc(&d, 4);
    printf("%s, %d CUs, %d iterations\n", p.name, cus, iters);
    run<0>(iters, d, cus, "SALU: 64 s_add_u32", 64);
    run<1>(iters, d, cus, "VALU: 64 v_add_u32", 64);
    run<2>(iters, d, cus, "32 s_add + 32 v_add interleaved", 64);
    run<3>(iters, d, cus, "16 x {v_cmp, saveexec, v_add, s_or exec}", 64);
    run<4>(iters, d, cus, "16 x {v_cmp, saveexec, cbranch_execz, v_add, s_or}", 80);
    run<5>(iters, d, cus, "65 v_add + 63 s_add interleaved", 128);
    run<6>(iters, d, cus, "48 v_add + 16 ds_read_b32", 64);
    run<7>(iters, d, cus, "48 v_add + 16 s_waitcnt (nothing pending)", 64);
    run<8>(iters, d, cus, "48 v_add + 16 s_nop 0", 64);
    run<9>(iters, d, cus, "48 v_add + 16 s_cbranch_scc1 (to the next)", 64);
    return 0;
}
