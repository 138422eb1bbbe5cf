// pt_mk_lds.hip — megakernel instantiations for scenes that live in LDS (ONCHIP: every PNode / PTri / PAttr / PMat /
// PLight of the scene staged per workgroup; Cornell-class, BASELINE C1 / C2). These kernels are VALU-bound; this file is
// built with -fno-slp-vectorize (Makefile): the SLP vectorizer's v_pk_mul_f32 / v_pk_add_f32 pairs do not issue at twice
// the scalar rate on gfx950 and cost register-pair moves (+2.3 % on C2 without them, profiles/r02_ab_noslp.log).
#include "pt_megakernel.h"

namespace pt {

hipError_t launch_megakernel_lds(int integrator, bool count, const KParams& P, dim3 grid, dim3 block, unsigned lds, hipStream_t stream) {
#define PT_LAUNCH(I, C, RF, FL) hipLaunchKernelGGL((megakernel<I, C, false, true, RF, FL>), grid, block, lds, stream, P)
#define PT_PICK(I) do { if (P.refill) { if (count) PT_LAUNCH(I, true, true, false); else PT_LAUNCH(I, false, true, false); } \
                        else if (P.flat && !count && P.simple) hipLaunchKernelGGL((megakernel<I, false, false, true, false, true, true>), grid, block, lds, stream, P); \
                        else if (P.flat && !count) PT_LAUNCH(I, false, false, true); \
                        else if (count) PT_LAUNCH(I, true, false, false); \
                        else PT_LAUNCH(I, false, false, false); } while (0)
    if (integrator == 2) PT_PICK(2); else PT_PICK(0);
#undef PT_PICK
#undef PT_LAUNCH
    return hipGetLastError();
}

}  // namespace pt
