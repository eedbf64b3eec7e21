// scan = 1: the systematic generator sweep (depolarizing rule, every code).
#include "ladder_kernel.hpp"

namespace qecmc {

hipError_t launch_ladder_sweep(const LadderArgs &a, hipStream_t stream)
{
    constexpr int T = kCodeToric, X = kCodeXzzx, R = kCodeRotated, P = kCodePlanar;
    const bool big = (unsigned)a.Nc * 64u > 512;
    // toric: the general top-chain path only for L > 16 or a top chain below p = 0.75; the plaquette codes always keep it
    const bool gentop = a.code != T || (a.thr_logical != 0 && (a.L > 16 || !((a.acc_all_mask >> (a.Nc - 1)) & 1u)));
    const uint32_t want = kScan | (a.conv_mode != 0 ? kConv : 0u) | (gentop ? kGentop : 0u);
    const void *fn;
    if (a.code == T)
        fn = big ? select_ladder_kernel<1024, 4, T, kScan, kScan | kConv, kScan | kGentop, kScan | kGentop | kConv>(want)
                 : select_ladder_kernel<512, 8, T, kScan, kScan | kConv, kScan | kGentop, kScan | kGentop | kConv>(want);
    else
        fn = big ? LadderKernels<1024, 4, kScan | kGentop, kScan | kGentop | kConv>::of<X, R, P>(a.code, want)
                 : LadderKernels<512, 8, kScan | kGentop, kScan | kGentop | kConv>::of<X, R, P>(a.code, want);
    if (!fn) return hipErrorInvalidValue;
    return launch_ladder_fn(fn, a, stream, false);
}

}  // namespace qecmc
