// Toric code, depolarizing rule, the reference's random scan: the headline kernel family.
#include "ladder_kernel.hpp"

namespace qecmc {

hipError_t launch_ladder_toric(const LadderArgs &a, hipStream_t stream)
{
    constexpr int T = kCodeToric;
    const bool big = (unsigned)a.Nc * 64u > 512;        // 9 .. 16 rungs: 1024-thread workgroups at 4 waves per SIMD
    // the general top-chain path is needed only for L > 16 or a top chain below p = 0.75 (1-chain ladder)
    const bool gentop = a.thr_logical != 0 && (a.L > 16 || !((a.acc_all_mask >> (a.Nc - 1)) & 1u));
    const bool gsplit = (int)a.n_gen <= kGenSplit;      // the table layout of ladder_gen_dwords()
    // the swap sweep run once by wave 0 (SSW) pays where four workgroups share a CU: the VALU-bound shapes
    const bool ssw = !big && 4 * ladder_launch_lds(a) <= 160 * 1024 && !(a.tune & 8u);
    // the dE look-up table (DELUT): between the halves of a split table (L <= 9) or behind an unsplit one (L >= 12), ladder_gen_dwords
    // (three workgroups per CU -- L = 10 ... 12 at 8 temperatures -- are the LDS-bound shapes: -1.5 % with the table, see ladder_surf.hip)
    const bool lut = !(a.tune & 4u) && (!gsplit || (int)a.n_gen + 64 <= kGenSplit) && (160 * 1024) / ladder_launch_lds(a) != 3;

    uint32_t want = (a.conv_mode != 0 ? kConv : 0u) | (gsplit ? kGsplit : 0u);
    bool pre = false;
    if (a.queue != nullptr) {
        // runs that stop by the criterion: the persistent-grid kernels with the work queue (capi.hip decides, ladder_uses_queue);
        // no instantiation for a shape outside ladder_uses_queue() -> an error below, never a plain kernel on the capped grid
        want |= kQueue | ((!big && gsplit && (int)a.n_gen + 64 <= kGenSplit) ? kDelut : 0u);
        if (gentop) want |= kGentop;                    // (not instantiated)
    } else if (gentop) {
        want |= kGentop;
    } else if (ladder_wants_pre(a)) {
        pre = true;
        want |= kPre | (lut ? kDelut : 0u);
    } else if (!(want & kConv) && ssw && gsplit) {
        want |= kSsw | (lut ? kDelut : 0u);             // (four workgroups per CU: small lattices, a split table)
    } else if (lut) {
        want |= kDelut;
    }
    const void *fn;
    if (pre)   // 4 waves per SIMD by their LDS footprint: 128 VGPRs, the top chain's blocks drawn ahead
        fn = big ? nullptr      // (ladder_wants_pre: up to 8 rungs)
                 : select_ladder_kernel<512, 4, T, kPre, kPre | kConv, kPre | kGsplit, kPre | kGsplit | kConv, kPre | kDelut, kPre | kDelut | kConv,
                                        kPre | kDelut | kGsplit, kPre | kDelut | kGsplit | kConv>(want);
    else if (big)
        fn = select_ladder_kernel<1024, 4, T, 0u, kConv, kGsplit, kGsplit | kConv,                                          // plain
                                  kGentop, kGentop | kConv, kGentop | kGsplit, kGentop | kGsplit | kConv,                   // general top chain
                                  kDelut, kDelut | kConv, kDelut | kGsplit, kDelut | kGsplit | kConv,                       // dE table
                                  kQueue | kConv, kQueue | kConv | kGsplit>(want);                                          // work queue
    else
        fn = select_ladder_kernel<512, 8, T, 0u, kConv, kGsplit, kGsplit | kConv,
                                  kGentop, kGentop | kConv, kGentop | kGsplit, kGentop | kGsplit | kConv,
                                  kDelut, kDelut | kConv, kDelut | kGsplit, kDelut | kGsplit | kConv,
                                  kSsw | kGsplit, kSsw | kGsplit | kDelut,                                                  // swap sweep by wave 0
                                  kQueue | kConv, kQueue | kConv | kGsplit, kQueue | kConv | kGsplit | kDelut>(want);
    if (!fn) return hipErrorInvalidValue;
    return launch_ladder_fn(fn, a, stream, (want & kQueue) != 0);
}

}  // namespace qecmc
