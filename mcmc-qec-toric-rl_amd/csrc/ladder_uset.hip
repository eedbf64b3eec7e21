// USET instantiations: the ladder kernel with the unique-chain set insertion of PTDC / STDC / PTRC / STRC compiled in.
#include "ladder_kernel.hpp"

namespace qecmc {

hipError_t launch_ladder_uset(const LadderArgs &a, hipStream_t stream)
{
    constexpr int T = kCodeToric, X = kCodeXzzx, R = kCodeRotated, P = kCodePlanar;
    const unsigned block = (unsigned)a.Nc * 64u;
    const void *fn;
    // direct-counting runs: depolarizing random scan without logical moves (no general top path), fixed length
    if ((a.noise && a.noise != 2) || a.scan || a.conv_mode != 0 || a.thr_logical != 0) return hipErrorInvalidValue;
    if (a.xyz_thr != nullptr && (a.code == T || a.Nc != 1 || a.noise)) return hipErrorInvalidValue;   // Chain_xyz: single chains, table-driven codes
    if (a.noise == 2) {
        // STDC_droplet_alpha (decoders.py:510-534): single Chain_alpha chains
        if (a.Nc != 1 || (a.code != X && a.code != R)) return hipErrorInvalidValue;
        fn = a.code == X ? (const void *)ladder_rs_toric_kernel<1024, 4, false, false, X, true, false, true, true, true>
                         : (const void *)ladder_rs_toric_kernel<1024, 4, false, false, R, true, false, true, true, true>;
    } else {
        const bool gsplit = (int)a.n_gen <= kGenSplit;
#define QECMC_KU(code) (block <= 512 ? (gsplit ? (const void *)ladder_rs_toric_kernel<512, 8, false, true, code, false, false, false, true>   \
                                              : (const void *)ladder_rs_toric_kernel<512, 8, false, false, code, false, false, false, true>)  \
                                     : (gsplit ? (const void *)ladder_rs_toric_kernel<1024, 4, false, true, code, false, false, false, true>  \
                                              : (const void *)ladder_rs_toric_kernel<1024, 4, false, false, code, false, false, false, true>))
        fn = a.code == T ? QECMC_KU(T) : a.code == X ? QECMC_KU(X) : a.code == R ? QECMC_KU(R) : QECMC_KU(P);
#undef QECMC_KU
    }
    return launch_ladder_fn(fn, a, stream);
}

}  // namespace qecmc
