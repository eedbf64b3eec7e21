// The biased (src/mcmc_biased.py) and alpha (src/mcmc_alpha.py) acceptance rules on the xzzx / rotated codes.
#include "ladder_kernel.hpp"

namespace qecmc {

hipError_t launch_ladder_biased(const LadderArgs &a, hipStream_t stream)
{
    constexpr int X = kCodeXzzx, R = kCodeRotated;
    const bool conv = a.conv_mode != 0;
    const void *fn;
#define QECMC_KB(maxt, minw, code, alpha) (conv ? (const void *)ladder_rs_toric_kernel<maxt, minw, true, false, code, true, false, true, false, alpha> \
                                                : (const void *)ladder_rs_toric_kernel<maxt, minw, false, false, code, true, false, true, false, alpha>)
    const unsigned block = (unsigned)a.Nc * 64u;
    const bool alpha = a.noise == 2;
    if (a.code == X) fn = block <= 512 ? (alpha ? QECMC_KB(512, 8, X, true) : QECMC_KB(512, 8, X, false)) : (alpha ? QECMC_KB(1024, 4, X, true) : QECMC_KB(1024, 4, X, false));
    else if (a.code == R) fn = block <= 512 ? (alpha ? QECMC_KB(512, 8, R, true) : QECMC_KB(512, 8, R, false)) : (alpha ? QECMC_KB(1024, 4, R, true) : QECMC_KB(1024, 4, R, false));
    else return hipErrorInvalidValue;
#undef QECMC_KB
    return launch_ladder_fn(fn, a, stream);
}

}  // namespace qecmc
