// Dispatcher of the parallel-tempering ladder kernel (ladder_kernel.hpp): the instantiations live in one translation unit per
// kernel family (ladder_toric / ladder_sweep / ladder_surf / ladder_biased / ladder_uset .hip) so that they build in parallel.
#include "ladder_kernel.hpp"

namespace qecmc {

size_t ladder_lds_bytes(int L, int Nc, int W, int ncls, int gen_dwords)
{
    (void)L;   // per group of 64 syndromes
    return sizeof(uint32_t) * (size_t)ladder_group_dwords(Nc, W, ncls, gen_dwords);
}

hipError_t launch_ladder_rs_toric(const LadderArgs &a, hipStream_t stream)
{
    if (a.N == 0) return hipSuccess;
    // One 64-syndrome group (Nc waves) per workgroup.  (Two groups per workgroup, sharing only the barrier, paid off while a
    // one-round grid ended in a long tail; with the current proposal loop the 8-wave workgroups are faster at every batch
    // size: +1.6 % at 65 536 syndromes, +5 % at 262 144.)
    if (a.scan == 3) return launch_ladder_wu(a, stream);
    if (a.scan == 2) return a.uset_tab != nullptr ? hipErrorInvalidValue : launch_ladder_colour(a, stream);
    if (a.uset_tab != nullptr) return launch_ladder_uset(a, stream);
    if (a.noise) return a.scan ? hipErrorInvalidValue : launch_ladder_biased(a, stream);   // the sweep is built for the depolarizing rule only
    if (a.scan) return launch_ladder_sweep(a, stream);
    return a.code == kCodeToric ? launch_ladder_toric(a, stream) : launch_ladder_surf(a, stream);
}

}  // namespace qecmc
