// scan = 3 (ladder_wu.hpp): the planar instantiations.
#include "ladder_wu.hpp"

namespace qecmc {

const void *wu_kernel_planar(int variant, int Nc, int W, uint32_t iters) { return wu_pick<kCodePlanar>(variant, Nc, W, iters); }

}  // namespace qecmc
