// scan = 3 (ladder_wu.hpp): the xzzx, rotated and planar instantiations.
#include "ladder_wu.hpp"

namespace qecmc {

const void *wu_kernel_surf(int code, bool conv, int Nc, int W)
{
#define QECMC_WU(c) (conv ? wu_pick<c, true>(Nc, W) : wu_pick<c, false>(Nc, W))
    return code == kCodeXzzx ? QECMC_WU(kCodeXzzx) : code == kCodeRotated ? QECMC_WU(kCodeRotated) : code == kCodePlanar ? QECMC_WU(kCodePlanar) : nullptr;
#undef QECMC_WU
}

}  // namespace qecmc
