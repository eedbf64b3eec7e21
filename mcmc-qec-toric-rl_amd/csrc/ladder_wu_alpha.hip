// scan = 3 under the alpha noise model (src/mcmc_alpha.py): the xzzx / rotated instantiations.  The kernel: ladder_wu.hpp.
#include "ladder_wu.hpp"

namespace qecmc {

const void *wu_kernel_alpha(int code, int variant, int Nc, int W, uint32_t iters)
{
    if (iters == 10u)
        return code == kCodeXzzx ? wu_pick_alpha<kCodeXzzx, 10>(variant, Nc, W) : code == kCodeRotated ? wu_pick_alpha<kCodeRotated, 10>(variant, Nc, W) : nullptr;
    return code == kCodeXzzx ? wu_pick_alpha<kCodeXzzx, 0>(variant, Nc, W) : code == kCodeRotated ? wu_pick_alpha<kCodeRotated, 0>(variant, Nc, W) : nullptr;
}

}  // namespace qecmc
