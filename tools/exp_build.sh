#!/bin/bash
# Timing experiments (diagnostic): builds variants of the library with parts of the ladder step compiled out
# (QECMC_EXP_* in csrc/ladder_rs.hip; their results are wrong by construction) into tools/exp_libs/.
set -e
cd "$(dirname "$0")/../mcmc-qec-toric-rl_amd/csrc"
mkdir -p ../../tools/exp_libs
for v in ${EXP_VARIANTS:-NOCASCADE NOBARRIER NOPHILOX NOSTATE NOTHR NOXOR NOTOP NOTOPFLUSH NOSWAPDRAW}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DQECMC_EXP_$v -shared -o ../../tools/exp_libs/libqecmc_$v.so capi.hip ladder_rs.hip primitives.hip &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DQECMC_EXP_NOBARRIER -DQECMC_EXP_NOCASCADE -shared -o ../../tools/exp_libs/libqecmc_NOBARRIER_NOCASCADE.so capi.hip ladder_rs.hip primitives.hip
