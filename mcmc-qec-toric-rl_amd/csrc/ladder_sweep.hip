// scan = 1: the systematic generator sweep (depolarizing rule, every code).
#include "ladder_kernel.hpp"

namespace qecmc {

hipError_t launch_ladder_sweep(const LadderArgs &a, hipStream_t stream)
{
    constexpr int T = kCodeToric, X = kCodeXzzx, R = kCodeRotated, P = kCodePlanar;
    const unsigned block = (unsigned)a.Nc * 64u;
    const bool conv = a.conv_mode != 0;
    const void *fn;
#define QECMC_K(maxt, minw, code, gentop) (conv ? (const void *)ladder_rs_toric_kernel<maxt, minw, true, false, code, false, true, gentop> \
                                                : (const void *)ladder_rs_toric_kernel<maxt, minw, false, false, code, false, true, gentop>)
    if (a.code == T) {
        const bool gentop = a.thr_logical != 0 && (a.L > 16 || !((a.acc_all_mask >> (a.Nc - 1)) & 1u));
        if (gentop) fn = block <= 512 ? QECMC_K(512, 8, T, true) : QECMC_K(1024, 4, T, true);
        else fn = block <= 512 ? QECMC_K(512, 8, T, false) : QECMC_K(1024, 4, T, false);
    } else if (a.code == X) fn = block <= 512 ? QECMC_K(512, 8, X, true) : QECMC_K(1024, 4, X, true);
    else if (a.code == R) fn = block <= 512 ? QECMC_K(512, 8, R, true) : QECMC_K(1024, 4, R, true);
    else if (a.code == P) fn = block <= 512 ? QECMC_K(512, 8, P, true) : QECMC_K(1024, 4, P, true);
    else return hipErrorInvalidValue;
#undef QECMC_K
    return launch_ladder_fn(fn, a, stream);
}

}  // namespace qecmc
