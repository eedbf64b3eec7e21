// XZZX, rotated and planar codes, depolarizing rule, random scan.
#include "ladder_kernel.hpp"

namespace qecmc {

hipError_t launch_ladder_surf(const LadderArgs &a, hipStream_t stream)
{
    constexpr int X = kCodeXzzx, R = kCodeRotated, P = kCodePlanar;
    const bool big = (unsigned)a.Nc * 64u > 512;
    const bool top_blind = (a.acc_all_mask >> (a.Nc - 1)) & 1u;     // a top chain at p = 0.75: the blind path
    // dE of a proposal from the look-up table behind the expanded generator table (DELUT, one row per Pauli pattern)
    // (measured: +8 % xzzx L = 9, +7.7 % rotated L = 9, +5.6 % rotated L = 13, +1 % planar L = 9 -- four workgroups per CU, bound by VALU
    // issue; 0 % rotated L = 21 -- two; -4 % xzzx L = 15 -- three per CU, where the LDS pipe is the busier one)
    const bool lut = !(a.tune & 4u) && a.gen_type != nullptr && a.n_types > 0 && a.n_types <= kLutTypes && (160 * 1024) / ladder_launch_lds(a) != 3;

    // (these codes always keep the table-driven general top-chain path)
    uint32_t want = kGentop | (a.conv_mode != 0 ? kConv : 0u);
    bool pre = false;
    if (a.queue != nullptr) {
        // runs that stop by the criterion: the persistent-grid kernels with the work queue (built on the blind top chain)
        if (!top_blind) return hipErrorInvalidValue;
        want |= kQueue;
    } else if (ladder_wants_pre(a) && top_blind) {                   // (the blind path is what the blocks drawn ahead feed)
        pre = true;
        want |= kPre | (lut ? kDelut : 0u);
    } else if (!(want & kConv) && !big && 4 * ladder_launch_lds(a) <= 160 * 1024 && !(a.tune & 8u)) {
        want |= kSsw | (lut ? kDelut : 0u);                          // four workgroups per CU: the swap sweep run once by wave 0
    } else if (lut) {
        want |= kDelut;
    }
    const void *fn;
    if (pre)
        fn = big ? nullptr      // (ladder_wants_pre: up to 8 rungs)
                 : LadderKernels<512, 4, kGentop | kPre, kGentop | kPre | kConv, kGentop | kPre | kDelut, kGentop | kPre | kDelut | kConv>::of<X, R, P>(a.code, want);
    else if (big)
        fn = LadderKernels<1024, 4, kGentop, kGentop | kConv, kGentop | kDelut, kGentop | kDelut | kConv, kGentop | kQueue | kConv>::of<X, R, P>(a.code, want);
    else
        fn = LadderKernels<512, 8, kGentop, kGentop | kConv, kGentop | kDelut, kGentop | kDelut | kConv, kGentop | kSsw, kGentop | kSsw | kDelut,
                           kGentop | kQueue | kConv>::of<X, R, P>(a.code, want);
    if (!fn) return hipErrorInvalidValue;
    return launch_ladder_fn(fn, a, stream, (want & kQueue) != 0);
}

}  // namespace qecmc
