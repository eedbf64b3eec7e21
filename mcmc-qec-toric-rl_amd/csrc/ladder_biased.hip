// The biased (src/mcmc_biased.py) and alpha (src/mcmc_alpha.py) acceptance rules on the xzzx / rotated codes.
#include "ladder_kernel.hpp"

namespace qecmc {

hipError_t launch_ladder_biased(const LadderArgs &a, hipStream_t stream)
{
    constexpr int X = kCodeXzzx, R = kCodeRotated;
    const bool big = (unsigned)a.Nc * 64u > 512;
    constexpr uint32_t B = kBiased | kGentop;
    uint32_t want = B | (a.conv_mode != 0 ? kConv : 0u) | (a.noise == 2 ? kAlpha : 0u);
    // four workgroups per CU, fixed-length runs: the swap sweep run once by wave 0 (SSW, as on the depolarizing kernels)
    if (!big && !(want & kConv) && 4 * ladder_launch_lds(a) <= 160 * 1024 && !(a.tune & 8u)) want |= kSsw;
    const void *fn = big ? LadderKernels<1024, 4, B, B | kConv, B | kAlpha, B | kAlpha | kConv>::of<X, R>(a.code, want)
                         : LadderKernels<512, 8, B, B | kConv, B | kAlpha, B | kAlpha | kConv, B | kSsw, B | kAlpha | kSsw>::of<X, R>(a.code, want);
    if (!fn) return hipErrorInvalidValue;
    return launch_ladder_fn(fn, a, stream, false);
}

}  // namespace qecmc
