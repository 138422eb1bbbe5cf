// pt_mk_lds.hip — megakernel instantiations for scenes that live in LDS (ONCHIP: every PNode / PTri / PAttr / PMat /
// PLight of the scene staged per workgroup; Cornell-class, BASELINE C1 / C2). These kernels are VALU-bound; this file is
// built with -fno-slp-vectorize (Makefile): the SLP vectorizer's v_pk_mul_f32 / v_pk_add_f32 pairs do not issue at twice
// the scalar rate on gfx950 and cost register-pair moves (+2.3 % on C2 without them, profiles/r02_ab_noslp.log).
#include "pt_megakernel.h"

namespace pt {

hipError_t launch_megakernel_lds(int integrator, bool count, const KParams& P, dim3 grid, dim3 block, unsigned lds, hipStream_t stream) {
    // workgroups of 8 or 16 waves may need more than the default 64 KB of dynamic LDS: raised per kernel below
#define PT_LDS_OK(K) do { if (lds > 65536u) { hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e_ != hipSuccess) return e_; } } while (0)
#define PT_LAUNCH(I, C, RF, FL) do { PT_LDS_OK((megakernel<I, C, false, true, RF, FL>)); hipLaunchKernelGGL((megakernel<I, C, false, true, RF, FL>), grid, block, lds, stream, P); } while (0)
    if (P.flat == 3 && integrator == 0 && !count) {          // both rays of a lane in one FLAT pass (SIMPLE scenes, MIS)
        if (P.simple) { PT_LDS_OK((megakernel_flat2<0, true>)); hipLaunchKernelGGL((megakernel_flat2<0, true>), grid, block, lds, stream, P); }
        else if (P.lean) { PT_LDS_OK((megakernel_flat2<0, false, true>)); hipLaunchKernelGGL((megakernel_flat2<0, false, true>), grid, block, lds, stream, P); }
        else { PT_LDS_OK((megakernel_flat2<0, false>)); hipLaunchKernelGGL((megakernel_flat2<0, false>), grid, block, lds, stream, P); }
        return hipGetLastError();
    }
#ifdef PT_EXPERIMENTAL      // option "refill" 2: the resumable traversal for LDS-resident scenes too (-32 % on Cornell)
#define PT_PICK_REFILL(I) if (P.refill) { if (count) PT_LAUNCH(I, true, true, false); else PT_LAUNCH(I, false, true, false); } else
#else
#define PT_PICK_REFILL(I)
#endif
#define PT_PICK(I) do { PT_PICK_REFILL(I) \
                        if (P.flat == 2 && !count && P.simple) { PT_LDS_OK((megakernel<I, false, false, true, false, true, true, 2>)); hipLaunchKernelGGL((megakernel<I, false, false, true, false, true, true, 2>), grid, block, lds, stream, P); } \
                        else if (P.flat == 2 && !count) { PT_LDS_OK((megakernel<I, false, false, true, false, true, false, 2>)); hipLaunchKernelGGL((megakernel<I, false, false, true, false, true, false, 2>), grid, block, lds, stream, P); } \
                        else if (P.flat && !count && P.simple) { PT_LDS_OK((megakernel<I, false, false, true, false, true, true>)); hipLaunchKernelGGL((megakernel<I, false, false, true, false, true, true>), grid, block, lds, stream, P); } \
                        else if (P.flat && !count) PT_LAUNCH(I, false, false, true); \
                        else if (count) PT_LAUNCH(I, true, false, false); \
                        else PT_LAUNCH(I, false, false, false); } while (0)
    if (integrator == 2) PT_PICK(2); else PT_PICK(0);
#undef PT_PICK
#undef PT_PICK_REFILL
#undef PT_LAUNCH
#undef PT_LDS_OK
    return hipGetLastError();
}

}  // namespace pt
