// The biased (src/mcmc_biased.py) and alpha (src/mcmc_alpha.py) acceptance rules on the xzzx / rotated codes.
#include "ladder_kernel.hpp"

namespace qecmc {

hipError_t launch_ladder_biased(const LadderArgs &a, hipStream_t stream)
{
    constexpr int X = kCodeXzzx, R = kCodeRotated;
    const bool big = (unsigned)a.Nc * 64u > 512;
    constexpr uint32_t B = kBiased | kGentop;
    uint32_t want = B | (a.conv_mode != 0 ? kConv : 0u) | (a.noise == 2 ? kAlpha : 0u);
    // runs that stop by the criterion: the persistent-grid kernels with the work queue (capi.hip decides, ladder_uses_queue)
    if (a.queue != nullptr) want |= kQueue;
    // four workgroups per CU, fixed-length runs: the swap sweep run once by wave 0 (SSW, as on the depolarizing kernels)
    if (!big && !(want & kConv) && 4 * ladder_launch_lds(a) <= 160 * 1024 && !(a.tune & 8u)) want |= kSsw;
    constexpr uint32_t Q = kConv | kQueue;
    // (the queue kernels carry the criterion's window sums, the refill state and the biased rule's counts: at 64 VGPRs they would
    // spill 130-250 B per lane into their hot loops, so they run at 128 VGPRs / 4 waves per SIMD whatever the workgroup size)
    const void *fn = big ? LadderKernels<1024, 4, B, B | kConv, B | kAlpha, B | kAlpha | kConv, B | Q, B | kAlpha | Q>::of<X, R>(a.code, want)
                   : (want & kQueue) ? LadderKernels<512, 4, B | Q, B | kAlpha | Q>::of<X, R>(a.code, want)
                         : LadderKernels<512, 8, B, B | kConv, B | kAlpha, B | kAlpha | kConv, B | kSsw, B | kAlpha | kSsw>::of<X, R>(a.code, want);
    if (!fn) return hipErrorInvalidValue;
    return launch_ladder_fn(fn, a, stream, (want & kQueue) != 0);
}

}  // namespace qecmc
