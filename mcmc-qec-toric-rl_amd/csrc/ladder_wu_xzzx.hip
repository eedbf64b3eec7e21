// scan = 3 (ladder_wu.hpp): the xzzx instantiations.
#include "ladder_wu.hpp"

namespace qecmc {

const void *wu_kernel_xzzx(int variant, int Nc, int W, uint32_t iters) { return wu_pick<kCodeXzzx>(variant, Nc, W, iters); }

}  // namespace qecmc
