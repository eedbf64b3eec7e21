// Batched byte-state kernels behind the Toric_code / Chain API surface
// (src/toric_model.py:33-56, src/mcmc.py:19-43).  One thread per state; these
// are the compatibility path (drop-in single calls), not the throughput path.
#include "kernels.hpp"
#include "philox.hpp"
#include "stencil_bytes.hpp"

namespace qecmc {

namespace {
constexpr int kBlock = 64;
inline unsigned grid_for(uint64_t N) { return (unsigned)((N + kBlock - 1) / kBlock); }

__device__ __forceinline__ void copy_state(uint8_t *dst, const uint8_t *src, int nq)
{
    if (dst != src)
        for (int i = 0; i < nq; ++i) dst[i] = src[i];
}
}  // namespace

__global__ void k_apply_stabilizer(int L, uint64_t N, const uint8_t *in, uint8_t *out, const int32_t *rows,
                                   const int32_t *cols, const int32_t *ops, int32_t *dE)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int nq = 2 * L * L;
    copy_state(out + i * nq, in + i * nq, nq);
    dE[i] = toric_apply_stabilizer_b(L, out + i * nq, rows[i], cols[i], ops[i]);
}

__global__ void k_apply_logical(int L, uint64_t N, const uint8_t *in, uint8_t *out, const int32_t *ops,
                                const int32_t *layers, const int32_t *xpos, const int32_t *zpos, int32_t *dE)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int nq = 2 * L * L;
    copy_state(out + i * nq, in + i * nq, nq);
    dE[i] = toric_apply_logical_b(L, out + i * nq, ops[i], layers[i], xpos[i], zpos[i]);
}

__global__ void k_count_errors(int nq, uint64_t N, const uint8_t *in, int64_t *n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    n[i] = count_errors_b(nq, in + i * nq);
}

__global__ void k_eq_class(int L, uint64_t N, const uint8_t *in, int32_t *cls)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    cls[i] = toric_eq_class_b(L, in + i * (uint64_t)(2 * L * L));
}

__global__ void k_to_class(int L, uint64_t N, const uint8_t *in, uint8_t *out, const int32_t *eq)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int nq = 2 * L * L;
    copy_state(out + i * nq, in + i * nq, nq);
    toric_to_class_b(L, out + i * nq, eq[i]);
}

__global__ void k_syndrome(int L, uint64_t N, const uint8_t *in, uint8_t *defects)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int nq = 2 * L * L;
    toric_syndrome_b(L, in + i * nq, defects + i * nq);
}

// Chain.update_chain(iters), src/mcmc.py:19-43, one thread per chain, state in HBM.
// Draw addressing is the oracle's (DESIGN.md "RNG addressing").
__global__ void k_chain_update(const ChainArgs a)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N) return;
    const int L = a.L, nq = 2 * L * L;
    uint8_t *m = a.states + i * nq;
    const uint32_t syn = a.first_syndrome + (uint32_t)i;
    auto thr = [&](int dE) { return a.acc_tbl[dE]; };       // ceil(f^dE * 2^32), dE in [1, nq]
    for (uint64_t j = 0; j < a.iters; ++j) {
        const uint64_t k = a.k0 + j;
        const u32x4 x = philox_block(k, 0, syn, a.slot, a.seed_lo, a.seed_hi);
        if (a.thr_logical == 0) {                                  // mcmc.py:37-43
            const int row = scale_u32(x.x, L), col = scale_u32(x.y, L), op = (x.z >> 31) ? 1 : 3;
            const int dE = toric_apply_stabilizer_b(L, m, row, col, op);
            const bool acc = dE <= 0 || a.acc_all || x.w < thr(dE);
            if (!acc) toric_apply_stabilizer_b(L, m, row, col, op);   // undo (XOR is an involution)
        } else {                                                    // mcmc.py:20-35
            int dE;
            int row = 0, col = 0, op = 0, op0 = 0, op1 = 0, x0 = 0, z0 = 0, x1 = 0, z1 = 0;
            const bool logical = (uint64_t)x.x < a.thr_logical;
            if (logical) {
                op0 = x.y >> 30; op1 = x.z >> 30;                       // toric_model.py:234
                if (op0 == 1 || op0 == 2) x0 = scale_low30(x.y, L);     // :241-248 (positions share block (k,0))
                if (op0 == 3 || op0 == 2) z0 = scale_u16(x.w >> 16, L);
                if (op1 == 1 || op1 == 2) x1 = scale_low30(x.z, L);
                if (op1 == 3 || op1 == 2) z1 = scale_u16(x.w & 0xFFFFu, L);
                dE = toric_apply_logical_b(L, m, op0, 0, x0, z0) + toric_apply_logical_b(L, m, op1, 1, x1, z1);
            } else {
                row = scale_u32(x.y, L); col = scale_u32(x.z, L); op = (x.w >> 31) ? 1 : 3;
                dE = toric_apply_stabilizer_b(L, m, row, col, op);
            }
            bool acc = a.acc_all || dE <= 0;                         // mcmc.py:30
            if (!acc) {
                const u32x4 c = philox_block(k, 2, syn, a.slot, a.seed_lo, a.seed_hi);
                acc = c.x < thr(dE);                                 // mcmc.py:34
            }
            if (!acc) {
                if (logical) { toric_apply_logical_b(L, m, op1, 1, x1, z1); toric_apply_logical_b(L, m, op0, 0, x0, z0); }
                else toric_apply_stabilizer_b(L, m, row, col, op);
            }
        }
    }
}

#define QECMC_LAUNCH(kern, N, s, ...)                                                    \
    do {                                                                                 \
        if ((N) == 0) return hipSuccess;                                                 \
        hipLaunchKernelGGL(kern, dim3(grid_for(N)), dim3(kBlock), 0, s, __VA_ARGS__);    \
        return hipGetLastError();                                                        \
    } while (0)

hipError_t launch_apply_stabilizer(int L, uint64_t N, const uint8_t *in, uint8_t *out, const int32_t *rows,
                                   const int32_t *cols, const int32_t *ops, int32_t *dE, hipStream_t s)
{
    QECMC_LAUNCH(k_apply_stabilizer, N, s, L, N, in, out, rows, cols, ops, dE);
}
hipError_t launch_apply_logical(int L, uint64_t N, const uint8_t *in, uint8_t *out, const int32_t *ops,
                                const int32_t *layers, const int32_t *xpos, const int32_t *zpos, int32_t *dE,
                                hipStream_t s)
{
    QECMC_LAUNCH(k_apply_logical, N, s, L, N, in, out, ops, layers, xpos, zpos, dE);
}
hipError_t launch_count_errors(int nq, uint64_t N, const uint8_t *in, int64_t *n, hipStream_t s)
{
    QECMC_LAUNCH(k_count_errors, N, s, nq, N, in, n);
}
hipError_t launch_eq_class(int L, uint64_t N, const uint8_t *in, int32_t *cls, hipStream_t s)
{
    QECMC_LAUNCH(k_eq_class, N, s, L, N, in, cls);
}
hipError_t launch_to_class(int L, uint64_t N, const uint8_t *in, uint8_t *out, const int32_t *eq, hipStream_t s)
{
    QECMC_LAUNCH(k_to_class, N, s, L, N, in, out, eq);
}
hipError_t launch_syndrome(int L, uint64_t N, const uint8_t *in, uint8_t *defects, hipStream_t s)
{
    QECMC_LAUNCH(k_syndrome, N, s, L, N, in, defects);
}
hipError_t launch_chain_update(const ChainArgs &a, hipStream_t s)
{
    QECMC_LAUNCH(k_chain_update, a.N, s, a);
}

}  // namespace qecmc
