// The biased (src/mcmc_biased.py) and alpha (src/mcmc_alpha.py) acceptance rules on the xzzx / rotated codes.
#include "ladder_kernel.hpp"

namespace qecmc {

hipError_t launch_ladder_biased(const LadderArgs &a, hipStream_t stream)
{
    constexpr int X = kCodeXzzx, R = kCodeRotated;
    const bool big = (unsigned)a.Nc * 64u > 512;
    const uint32_t want = kBiased | kGentop | (a.conv_mode != 0 ? kConv : 0u) | (a.noise == 2 ? kAlpha : 0u);
    const void *fn = big ? LadderKernels<1024, 4, kBiased | kGentop, kBiased | kGentop | kConv, kBiased | kGentop | kAlpha, kBiased | kGentop | kAlpha | kConv>::of<X, R>(a.code, want)
                         : LadderKernels<512, 8, kBiased | kGentop, kBiased | kGentop | kConv, kBiased | kGentop | kAlpha, kBiased | kGentop | kAlpha | kConv>::of<X, R>(a.code, want);
    if (!fn) return hipErrorInvalidValue;
    return launch_ladder_fn(fn, a, stream, false);
}

}  // namespace qecmc
