// scan = 3 (wave-uniform generator picks, states in registers): dispatcher and the toric instantiations.  The kernel: ladder_wu.hpp.
#include "ladder_wu.hpp"

namespace qecmc {

size_t wu_lds_bytes(int Nc, int W, int ncls, int L) { return sizeof(uint32_t) * (size_t)wu_lds(Nc, W, ncls, L).total; }

// the shapes scan = 3 is built for: depolarizing rule, a ladder whose top rung accepts every move (Nc >= 2, p_top = 0.75), up to
// 32 state words per ladder rung (toric L <= 16, xzzx / rotated L <= 22), descriptor offsets that fit 16 bits, 1 <= iters <= 128,
// rungs at distinct temperatures (32-bit swap thresholds)
bool wu_supported(const LadderArgs &a)
{
    return a.noise == 0 && a.Nc >= 2 && ((a.acc_all_mask >> (a.Nc - 1)) & 1u) && !(a.acc_all_mask & ((1u << (a.Nc - 1)) - 1u)) &&
           a.W <= 32 && a.n_gen <= 1023u && a.iters >= 1u && a.iters <= 128u && a.swap_fast_ok != 0 && a.uset_tab == nullptr && a.swap_acc == nullptr && a.queue == nullptr &&
           wu_lds_bytes(a.Nc, a.W, a.ncls, a.L) <= 160 * 1024;
}

const void *wu_kernel_toric(bool conv, int Nc, int W) { return conv ? wu_pick<kCodeToric, true>(Nc, W) : wu_pick<kCodeToric, false>(Nc, W); }

hipError_t launch_ladder_wu(const LadderArgs &a, hipStream_t stream)
{
    if (!wu_supported(a) || a.wu_desc == nullptr || (a.first_syndrome & 63u)) return hipErrorInvalidValue;
    if (a.conv_mode != 0 && a.nlog == nullptr) return hipErrorInvalidValue;
    const bool conv = a.conv_mode != 0;
    const void *fn = a.code == kCodeToric ? wu_kernel_toric(conv, a.Nc, a.W) : wu_kernel_surf(a.code, conv, a.Nc, a.W);
    if (!fn) return hipErrorInvalidValue;
    const size_t lds = wu_lds_bytes(a.Nc, a.W, a.ncls, a.L);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    void *kargs[] = {const_cast<LadderArgs *>(&a)};
    hipError_t e = hipLaunchKernel(fn, dim3((unsigned)((a.N + 63) / 64)), dim3((unsigned)a.Nc * 64u), kargs, lds, stream);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

}  // namespace qecmc
