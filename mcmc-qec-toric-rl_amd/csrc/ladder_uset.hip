// USET instantiations: the ladder kernel with the unique-chain set insertion of PTDC / STDC / PTRC / STRC compiled in.
#include "ladder_kernel.hpp"

namespace qecmc {

hipError_t launch_ladder_uset(const LadderArgs &a, hipStream_t stream)
{
    constexpr int T = kCodeToric, X = kCodeXzzx, R = kCodeRotated, P = kCodePlanar;
    const bool big = (unsigned)a.Nc * 64u > 512;
    const void *fn;
    // direct-counting runs: depolarizing random scan without logical moves (no general top path), fixed length
    if ((a.noise && a.noise != 2) || a.scan || a.conv_mode != 0 || a.thr_logical != 0) return hipErrorInvalidValue;
    if (a.xyz_thr != nullptr && (a.code == T || a.Nc != 1 || a.noise)) return hipErrorInvalidValue;   // Chain_xyz: single chains, table-driven codes
    if (a.noise == 2) {
        // STDC_droplet_alpha (decoders.py:510-534): single Chain_alpha chains
        if (a.Nc != 1) return hipErrorInvalidValue;
        fn = LadderKernels<1024, 4, kUset | kBiased | kAlpha | kGentop>::of<X, R>(a.code, kUset | kBiased | kAlpha | kGentop);
    } else {
        const uint32_t want = kUset | ((int)a.n_gen <= kGenSplit ? kGsplit : 0u);
        fn = big ? LadderKernels<1024, 4, kUset, kUset | kGsplit>::of<T, X, R, P>(a.code, want)
                 : LadderKernels<512, 8, kUset, kUset | kGsplit>::of<T, X, R, P>(a.code, want);
    }
    if (!fn) return hipErrorInvalidValue;
    return launch_ladder_fn(fn, a, stream, false);
}

}  // namespace qecmc
