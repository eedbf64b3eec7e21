// scan = 3 (ladder_wu.hpp): the rotated instantiations.
#include "ladder_wu.hpp"

namespace qecmc {

const void *wu_kernel_rotated(int variant, int Nc, int W, uint32_t iters) { return wu_pick<kCodeRotated>(variant, Nc, W, iters); }

}  // namespace qecmc
