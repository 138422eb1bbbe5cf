// pt_mk_hbm.hip — megakernel instantiations for scenes in HBM: megakernel_hbm (6 waves per SIMD, 12-wave workgroups
// sharing a 44 KB copy of the top of the tree; loop exits, REFILL, opt-in culling) and the general 4-wave kernel
// (launches with too few tiles to fill its waves; -DPT_EXPERIMENTAL builds add the A/B instantiations of DESIGN.md §6).
// Built without the SLP vectorizer since round 3 (Makefile: at 64 VGPRs its packed pairs cost spills, -3.4 / -3.8 %), and with the
// per-lane range test in rcp_exact: the wave-uniform form that pays for the issue-bound LDS-resident kernels measures 0.7-1.0 %
// slower here (profiles/r03_ab_rcp_uniform_hbm.log).
#ifndef PT_RCP_UNIFORM
#define PT_RCP_UNIFORM 0
#endif
#include "pt_megakernel.h"

namespace pt {

hipError_t launch_megakernel_hbm(int integrator, bool count, bool syncShadow, bool hbm, const KParams& P, dim3 grid, dim3 block, unsigned lds, hipStream_t stream) {
    // more than 64 KB of dynamic LDS per workgroup has to be asked for (a workgroup may take all 160 KB of its CU)
#define PT_LDS_OK(K) do { if (lds > 65536u) { hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e_ != hipSuccess) return e_; } } while (0)
#define PT_LAUNCH_MK(I, C, D, RF) hipLaunchKernelGGL((megakernel<I, C, D, false, RF>), grid, block, lds, stream, P)
#define PT_LAUNCH_HBM1(I, C, CU, RF) do { PT_LDS_OK((megakernel_hbm<I, C, CU, RF>)); hipLaunchKernelGGL((megakernel_hbm<I, C, CU, RF>), grid, block, lds, stream, P); } while (0)
#ifdef PT_EXPERIMENTAL
#define PT_LAUNCH_HBM_TREES(I, C) if (P.refill && P.simple && P.wide && !C) { PT_LDS_OK((megakernel_hbm_wide<I>)); \
                                                                             hipLaunchKernelGGL((megakernel_hbm_wide<I>), grid, block, lds, stream, P); } \
                                  else if (P.refill && P.simple && P.compact && !C) { PT_LDS_OK((megakernel_hbm_compact<I>)); \
                                                                                      hipLaunchKernelGGL((megakernel_hbm_compact<I>), grid, block, lds, stream, P); } else
#else
#define PT_LAUNCH_HBM_TREES(I, C)
#endif
#define PT_LAUNCH_HBM(I, C) do { if (P.cull) PT_LAUNCH_HBM1(I, C, true, false); \
                                 else PT_LAUNCH_HBM_TREES(I, C) \
                                 if (P.refill && P.simple && !C) { PT_LDS_OK((megakernel_hbm_simple<I>)); \
                                                                   hipLaunchKernelGGL((megakernel_hbm_simple<I>), grid, block, lds, stream, P); } \
                                 else if (P.refill && P.lean && !C) { PT_LDS_OK((megakernel_hbm<I, false, false, true, false, true>)); \
                                                                      hipLaunchKernelGGL((megakernel_hbm<I, false, false, true, false, true>), grid, block, lds, stream, P); } \
                                 else if (P.refill) PT_LAUNCH_HBM1(I, C, false, true); \
                                 else PT_LAUNCH_HBM1(I, C, false, false); } while (0)
#define PT_PICK(I) do { if (hbm) { if (count) PT_LAUNCH_HBM(I, true); else PT_LAUNCH_HBM(I, false); } \
                        else if (P.refill && P.simple && !count) hipLaunchKernelGGL((megakernel<I, false, false, false, true, false, true>), grid, block, lds, stream, P); \
                        else if (P.refill && P.lean && !count) hipLaunchKernelGGL((megakernel<I, false, false, false, true, false, false, 1, true>), grid, block, lds, stream, P); \
                        else if (P.refill) { if (count) PT_LAUNCH_MK(I, true, false, true); else PT_LAUNCH_MK(I, false, false, true); } \
                        else if (count) PT_LAUNCH_MK(I, true, false, false); \
                        else PT_LAUNCH_MK(I, false, false, false); } while (0)
    if (integrator == 2) PT_PICK(2);
    else if (syncShadow) PT_PICK(0);
#ifdef PT_EXPERIMENTAL
    else { if (count) PT_LAUNCH_MK(0, true, true, false); else PT_LAUNCH_MK(0, false, true, false); }      // option "defer_shadow": the DEFER pair walk
#else
    else return hipErrorInvalidValue;                                     // pt_api.hip never asks for it in a default build
#endif
#undef PT_PICK
#undef PT_LAUNCH_HBM
#undef PT_LAUNCH_HBM_TREES
#undef PT_LAUNCH_HBM1
#undef PT_LAUNCH_MK
#undef PT_LDS_OK
    return hipGetLastError();
}

}  // namespace pt
