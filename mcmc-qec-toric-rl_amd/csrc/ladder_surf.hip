// XZZX, rotated and planar codes, depolarizing rule, random scan.
#include "ladder_kernel.hpp"

namespace qecmc {

hipError_t launch_ladder_surf(const LadderArgs &a, hipStream_t stream)
{
    constexpr int X = kCodeXzzx, R = kCodeRotated, P = kCodePlanar;
    const unsigned block = (unsigned)a.Nc * 64u;
    const bool conv = a.conv_mode != 0;
    // dE of a proposal from the look-up table behind the expanded generator table (DELUT, one row per Pauli pattern)
    // (measured: +8 % xzzx L = 9, +7.7 % rotated L = 9, +5.6 % rotated L = 13, +1 % planar L = 9 -- four workgroups per CU, bound by VALU
    // issue; 0 % rotated L = 21 -- two; -4 % xzzx L = 15 -- three per CU, where the LDS pipe is the busier one)
    const bool lut = !(a.tune & 4u) && a.gen_type != nullptr && a.n_types > 0 && a.n_types <= kLutTypes && (160 * 1024) / ladder_launch_lds(a) != 3;
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
#define QECMC_KPL(maxt, code) (conv ? (const void *)ladder_rs_toric_kernel<maxt, 4, true, false, code, false, false, true, false, false, true, true> \
                                    : (const void *)ladder_rs_toric_kernel<maxt, 4, false, false, code, false, false, true, false, false, true, true>)
        if (lut) {
            if (block <= 512) fn = a.code == X ? QECMC_KPL(512, X) : a.code == R ? QECMC_KPL(512, R) : a.code == P ? QECMC_KPL(512, P) : nullptr;
            else fn = a.code == X ? QECMC_KPL(1024, X) : a.code == R ? QECMC_KPL(1024, R) : a.code == P ? QECMC_KPL(1024, P) : nullptr;
        } else
        if (block <= 512) fn = a.code == X ? QECMC_KP(512, X) : a.code == R ? QECMC_KP(512, R) : a.code == P ? QECMC_KP(512, P) : nullptr;
        else fn = a.code == X ? QECMC_KP(1024, X) : a.code == R ? QECMC_KP(1024, R) : a.code == P ? QECMC_KP(1024, P) : nullptr;
#undef QECMC_KPL
    }
    else
#undef QECMC_KP
    if (!conv && block <= 512 && 4 * ladder_launch_lds(a) <= 160 * 1024 && !(a.tune & 8u)) {
        // four workgroups per CU: the VALU-bound shapes take the swap sweep run once by wave 0 (SSW, ladder_kernel.hpp)
#define QECMC_KS(code) (const void *)ladder_rs_toric_kernel<512, 8, false, false, code, false, false, true, false, false, false, false, false, true>
#define QECMC_KSL(code) (const void *)ladder_rs_toric_kernel<512, 8, false, false, code, false, false, true, false, false, false, true, false, true>
        if (lut) fn = a.code == X ? QECMC_KSL(X) : a.code == R ? QECMC_KSL(R) : a.code == P ? QECMC_KSL(P) : nullptr;
        else
        fn = a.code == X ? QECMC_KS(X) : a.code == R ? QECMC_KS(R) : a.code == P ? QECMC_KS(P) : nullptr;
#undef QECMC_KSL
#undef QECMC_KS
    } else
#define QECMC_KL(maxt, minw, code) (conv ? (const void *)ladder_rs_toric_kernel<maxt, minw, true, false, code, false, false, true, false, false, false, true> \
                                         : (const void *)ladder_rs_toric_kernel<maxt, minw, false, false, code, false, false, true, false, false, false, true>)
    if (lut && a.code == X) fn = block <= 512 ? QECMC_KL(512, 8, X) : QECMC_KL(1024, 4, X);
    else if (lut && a.code == R) fn = block <= 512 ? QECMC_KL(512, 8, R) : QECMC_KL(1024, 4, R);
    else if (lut && a.code == P) fn = block <= 512 ? QECMC_KL(512, 8, P) : QECMC_KL(1024, 4, P);
    else
#undef QECMC_KL
    if (a.code == X) fn = block <= 512 ? QECMC_K(512, 8, X) : QECMC_K(1024, 4, X);
    else if (a.code == R) fn = block <= 512 ? QECMC_K(512, 8, R) : QECMC_K(1024, 4, R);
    else if (a.code == P) fn = block <= 512 ? QECMC_K(512, 8, P) : QECMC_K(1024, 4, P);
    else return hipErrorInvalidValue;
#undef QECMC_K
    if (!fn) return hipErrorInvalidValue;
    return launch_ladder_fn(fn, a, stream);
}

}  // namespace qecmc
