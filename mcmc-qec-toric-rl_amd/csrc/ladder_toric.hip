// Toric code, depolarizing rule, the reference's random scan: the headline kernel family.
#include "ladder_kernel.hpp"

namespace qecmc {

hipError_t launch_ladder_toric(const LadderArgs &a, hipStream_t stream)
{
    constexpr int T = kCodeToric;
    const unsigned block = (unsigned)a.Nc * 64u;
    const bool conv = a.conv_mode != 0;
    // the general top-chain path is needed only for L > 16 or a top chain below p = 0.75 (1-chain ladder)
    const bool gentop = a.thr_logical != 0 && (a.L > 16 || !((a.acc_all_mask >> (a.Nc - 1)) & 1u));
    const bool gsplit = (int)a.n_gen <= kGenSplit;      // the table layout of ladder_gen_dwords()
    // the swap sweep run once by wave 0 (SSW) pays where four workgroups share a CU: the VALU-bound shapes
    const bool ssw = block <= 512 && 4 * ladder_launch_lds(a) <= 160 * 1024 && !(a.tune & 8u);
    // the dE look-up table (DELUT): between the halves of a split table (L <= 9) or behind an unsplit one (L >= 12), ladder_gen_dwords
    // (three workgroups per CU -- L = 10 ... 12 at 8 temperatures -- are the LDS-bound shapes: -1.5 % with the table, see ladder_surf.hip)
    const bool lut = !(a.tune & 4u) && (!gsplit || (int)a.n_gen + 64 <= kGenSplit) && (160 * 1024) / ladder_launch_lds(a) != 3;
    const void *fn;
#define QECMC_K(maxt, minw, g, gentop) (conv ? (const void *)ladder_rs_toric_kernel<maxt, minw, true, g, T, false, false, gentop> \
                                             : (const void *)ladder_rs_toric_kernel<maxt, minw, false, g, T, false, false, gentop>)
    if (a.queue != nullptr && conv && !gentop) {
        // runs that stop by the criterion: the persistent-grid kernels with the work queue (capi.hip decides, ladder_uses_queue)
#define QECMC_KQ(maxt, minw, g, lut) (const void *)ladder_rs_toric_kernel<maxt, minw, true, g, T, false, false, false, false, false, false, lut, true>
        const bool qlut = gsplit && (int)a.n_gen + 64 <= kGenSplit;
        if (block <= 512) fn = qlut ? QECMC_KQ(512, 8, true, true) : gsplit ? QECMC_KQ(512, 8, true, false) : QECMC_KQ(512, 8, false, false);
        else fn = gsplit ? QECMC_KQ(1024, 4, true, false) : QECMC_KQ(1024, 4, false, false);
#undef QECMC_KQ
    } else
    if (gentop) {
        if (block <= 512) fn = gsplit ? QECMC_K(512, 8, true, true) : QECMC_K(512, 8, false, true);
        else fn = gsplit ? QECMC_K(1024, 4, true, true) : QECMC_K(1024, 4, false, true);
    } else if (ladder_wants_pre(a)) {
#define QECMC_KP(maxt, g, lut) (conv ? (const void *)ladder_rs_toric_kernel<maxt, 4, true, g, T, false, false, false, false, false, true, lut> \
                                     : (const void *)ladder_rs_toric_kernel<maxt, 4, false, g, T, false, false, false, false, false, true, lut>)
        if (block <= 512) fn = gsplit ? (lut ? QECMC_KP(512, true, true) : QECMC_KP(512, true, false)) : (lut ? QECMC_KP(512, false, true) : QECMC_KP(512, false, false));
        else fn = gsplit ? (lut ? QECMC_KP(1024, true, true) : QECMC_KP(1024, true, false)) : (lut ? QECMC_KP(1024, false, true) : QECMC_KP(1024, false, false));
#undef QECMC_KP
    } else if (!conv && ssw && gsplit) {
        // (four workgroups per CU: small lattices, a split table)
        fn = lut ? (const void *)ladder_rs_toric_kernel<512, 8, false, true, T, false, false, false, false, false, false, true, false, true>
                 : (const void *)ladder_rs_toric_kernel<512, 8, false, true, T, false, false, false, false, false, false, false, false, true>;
    } else if (lut) {
#define QECMC_KL(maxt, minw, g) (conv ? (const void *)ladder_rs_toric_kernel<maxt, minw, true, g, T, false, false, false, false, false, false, true> \
                                      : (const void *)ladder_rs_toric_kernel<maxt, minw, false, g, T, false, false, false, false, false, false, true>)
        if (block <= 512) fn = gsplit ? QECMC_KL(512, 8, true) : QECMC_KL(512, 8, false);
        else fn = gsplit ? QECMC_KL(1024, 4, true) : QECMC_KL(1024, 4, false);
#undef QECMC_KL
    } else {
        if (block <= 512) fn = gsplit ? QECMC_K(512, 8, true, false) : QECMC_K(512, 8, false, false);
        else fn = gsplit ? QECMC_K(1024, 4, true, false) : QECMC_K(1024, 4, false, false);
    }
#undef QECMC_K
    return launch_ladder_fn(fn, a, stream);
}

}  // namespace qecmc
