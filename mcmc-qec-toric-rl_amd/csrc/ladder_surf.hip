// XZZX, rotated and planar codes, depolarizing rule, random scan.
#include "ladder_kernel.hpp"

namespace qecmc {

hipError_t launch_ladder_surf(const LadderArgs &a, hipStream_t stream)
{
    constexpr int X = kCodeXzzx, R = kCodeRotated, P = kCodePlanar;
    const unsigned block = (unsigned)a.Nc * 64u;
    const bool conv = a.conv_mode != 0;
    const void *fn;
#define QECMC_K(maxt, minw, code) (conv ? (const void *)ladder_rs_toric_kernel<maxt, minw, true, false, code, false, false, true> \
                                        : (const void *)ladder_rs_toric_kernel<maxt, minw, false, false, code, false, false, true>)
#define QECMC_KP(maxt, code) (conv ? (const void *)ladder_rs_toric_kernel<maxt, 4, true, false, code, false, false, true, false, false, true> \
                                   : (const void *)ladder_rs_toric_kernel<maxt, 4, false, false, code, false, false, true, false, false, true>)
    if (a.queue != nullptr && conv && ((a.acc_all_mask >> (a.Nc - 1)) & 1u)) {
        // runs that stop by the criterion: the persistent-grid kernels with the work queue
#define QECMC_KQ(maxt, minw, code) (const void *)ladder_rs_toric_kernel<maxt, minw, true, false, code, false, false, true, false, false, false, false, true>
        if (block <= 512) fn = a.code == X ? QECMC_KQ(512, 8, X) : a.code == R ? QECMC_KQ(512, 8, R) : a.code == P ? QECMC_KQ(512, 8, P) : nullptr;
        else fn = a.code == X ? QECMC_KQ(1024, 4, X) : a.code == R ? QECMC_KQ(1024, 4, R) : a.code == P ? QECMC_KQ(1024, 4, P) : nullptr;
#undef QECMC_KQ
    } else
    if (ladder_wants_pre(a) && ((a.acc_all_mask >> (a.Nc - 1)) & 1u)) {   // (a top chain at p = 0.75: the blind path the blocks feed)
        if (block <= 512) fn = a.code == X ? QECMC_KP(512, X) : a.code == R ? QECMC_KP(512, R) : a.code == P ? QECMC_KP(512, P) : nullptr;
        else fn = a.code == X ? QECMC_KP(1024, X) : a.code == R ? QECMC_KP(1024, R) : a.code == P ? QECMC_KP(1024, P) : nullptr;
    }
    else
#undef QECMC_KP
    if (!conv && block <= 512 && 4 * ladder_launch_lds(a) <= 160 * 1024 && !(a.tune & 8u)) {
        // four workgroups per CU: the VALU-bound shapes take the swap sweep run once by wave 0 (SSW, ladder_kernel.hpp)
#define QECMC_KS(code) (const void *)ladder_rs_toric_kernel<512, 8, false, false, code, false, false, true, false, false, false, false, false, true>
        fn = a.code == X ? QECMC_KS(X) : a.code == R ? QECMC_KS(R) : a.code == P ? QECMC_KS(P) : nullptr;
#undef QECMC_KS
    } else
    if (a.code == X) fn = block <= 512 ? QECMC_K(512, 8, X) : QECMC_K(1024, 4, X);
    else if (a.code == R) fn = block <= 512 ? QECMC_K(512, 8, R) : QECMC_K(1024, 4, R);
    else if (a.code == P) fn = block <= 512 ? QECMC_K(512, 8, P) : QECMC_K(1024, 4, P);
    else return hipErrorInvalidValue;
#undef QECMC_K
    if (!fn) return hipErrorInvalidValue;
    return launch_ladder_fn(fn, a, stream);
}

}  // namespace qecmc
