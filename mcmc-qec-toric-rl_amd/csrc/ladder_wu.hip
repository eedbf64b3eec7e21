// scan = 3 (wave-uniform generator picks, states in registers): dispatcher and the toric instantiations.  The kernel: ladder_wu.hpp.
#include "ladder_wu.hpp"

namespace qecmc {

size_t wu_lds_bytes(int Nc, int W, int ncls, int L, bool conv, bool alpha) { return sizeof(uint32_t) * (size_t)wu_lds(Nc, W, ncls, L, conv, alpha).total; }

// the shapes scan = 3 is built for: depolarizing rule, a ladder whose top rung accepts every move (Nc >= 2, p_top = 0.75), up to
// 16 state words per ladder rung (toric / planar L <= 11, xzzx / rotated L <= 16: where the states fit the registers of 8 waves per
// SIMD), fixed-length runs of up to 8 rungs also up to 32 words (toric L <= 16, xzzx / rotated L <= 22: 6 waves per SIMD, the exchange in two
// halves), 1 <= iters <= 128, rungs at distinct temperatures (32-bit swap thresholds)
// ... and the alpha noise model's ladder (noise = 2, whose top rung sits at pz_tilde = 1 and accepts every move) on the xzzx / rotated codes up to
// 8 state words, where the plan allows the single-precision estimate of the acceptance ratio on every rung below the top
bool wu_supported(const LadderArgs &a)
{
    if (a.noise == 2)
        return (a.code == kCodeXzzx || a.code == kCodeRotated) && a.Nc >= 2 && a.Nc <= 16 && a.W <= 8 && a.n_gen <= 1023u && a.iters >= 1u && a.iters <= 128u &&
               (a.bias_f32ok & ((1u << (a.Nc - 1)) - 1u)) == ((1u << (a.Nc - 1)) - 1u) && a.bias_tbl != nullptr && a.alpha_lnb != nullptr &&
               a.uset_tab == nullptr && a.swap_acc == nullptr && !a.resume && a.neff == nullptr &&
               wu_lds_bytes(a.Nc, a.W, a.ncls, a.L, a.conv_mode != 0, true) <= 160 * 1024;
    return a.noise == 0 && a.Nc >= 2 && ((a.acc_all_mask >> (a.Nc - 1)) & 1u) && !(a.acc_all_mask & ((1u << (a.Nc - 1)) - 1u)) &&
           (a.W <= 16 || (a.W <= 32 && a.conv_mode == 0 && a.Nc <= 8 && a.code != kCodePlanar)) && a.n_gen <= 1023u && a.iters >= 1u && a.iters <= 128u && a.swap_fast_ok != 0 &&
           a.uset_tab == nullptr && a.swap_acc == nullptr && wu_lds_bytes(a.Nc, a.W, a.ncls, a.L, a.conv_mode != 0, false) <= 160 * 1024;
}

const void *wu_kernel_toric(int variant, int Nc, int W, uint32_t iters) { return wu_pick<kCodeToric>(variant, Nc, W, iters); }

// the persistent grid of the criterion runs: a.grid_cap workgroups (capi.hip), each owning a.wu_chunk ladders of the batch
hipError_t launch_ladder_wu(const LadderArgs &a, hipStream_t stream)
{
    if (!wu_supported(a) || a.wu_desc == nullptr || (a.first_syndrome & 63u)) return hipErrorInvalidValue;
    if (a.conv_mode != 0 && a.nlog == nullptr) return hipErrorInvalidValue;
    const bool queue = a.conv_mode != 0;
    if (queue && (a.resume || a.write_states || a.wu_chunk < 64u || (a.wu_chunk & 63u))) return hipErrorInvalidValue;
    const int variant = queue ? 2 : 0;
    const void *fn = a.noise == 2 ? wu_kernel_alpha(a.code, variant, a.Nc, a.W, a.iters) : a.code == kCodeToric ? wu_kernel_toric(variant, a.Nc, a.W, a.iters) : a.code == kCodeXzzx ? wu_kernel_xzzx(variant, a.Nc, a.W, a.iters)
                   : a.code == kCodeRotated ? wu_kernel_rotated(variant, a.Nc, a.W, a.iters) : a.code == kCodePlanar ? wu_kernel_planar(variant, a.Nc, a.W, a.iters) : nullptr;
    if (!fn) return hipErrorInvalidValue;
    const size_t lds = wu_lds_bytes(a.Nc, a.W, a.ncls, a.L, a.conv_mode != 0, a.noise == 2);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const uint64_t per = queue ? a.wu_chunk : 64u;
    void *kargs[] = {const_cast<LadderArgs *>(&a)};
    hipError_t e = hipLaunchKernel(fn, dim3((unsigned)((a.N + per - 1) / per)), dim3((unsigned)a.Nc * 64u), kargs, lds, stream);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

}  // namespace qecmc
