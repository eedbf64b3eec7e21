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

__device__ __forceinline__ int code_nq(int code, int L) { return code_nq_of(code, L); }

__global__ void k_apply_stabilizer(int code, int L, uint64_t N, const uint8_t *in, uint8_t *out, const int32_t *rows,
                                   const int32_t *cols, const int32_t *ops, int32_t *dE)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int nq = code_nq(code, L);
    copy_state(out + i * nq, in + i * nq, nq);
    dE[i] = code == kCodeToric ? toric_apply_stabilizer_b(L, out + i * nq, rows[i], cols[i], ops[i])
                               : surf_apply_stabilizer_b(code, L, out + i * nq, rows[i], cols[i], ops[i]);
}

__global__ void k_apply_logical(int code, int L, uint64_t N, const uint8_t *in, uint8_t *out, const int32_t *ops,
                                const int32_t *layers, const int32_t *xpos, const int32_t *zpos, int32_t *dE)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int nq = code_nq(code, L);
    copy_state(out + i * nq, in + i * nq, nq);
    dE[i] = code == kCodeToric ? toric_apply_logical_b(L, out + i * nq, ops[i], layers[i], xpos[i], zpos[i])
                               : surf_apply_logical_b(code, L, out + i * nq, ops[i], xpos[i], zpos[i]);
}

__global__ void k_count_errors(int nq, uint64_t N, const uint8_t *in, int64_t *n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    n[i] = count_errors_b(nq, in + i * nq);
}

__global__ void k_eq_class(int code, int L, uint64_t N, const uint8_t *in, int32_t *cls)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint8_t *m = in + i * (uint64_t)code_nq(code, L);
    cls[i] = code == kCodeToric ? toric_eq_class_b(L, m) : surf_eq_class_b(code, L, m);
}

__global__ void k_to_class(int L, uint64_t N, const uint8_t *in, uint8_t *out, const int32_t *eq)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int nq = 2 * L * L;
    copy_state(out + i * nq, in + i * nq, nq);
    toric_to_class_b(L, out + i * nq, eq[i]);
}

__global__ void k_syndrome(int code, int L, uint64_t N, const uint8_t *in, uint8_t *defects)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (code == kCodeToric) toric_syndrome_b(L, in + i * (uint64_t)(2 * L * L), defects + i * (uint64_t)(2 * L * L));
    else if (code == kCodePlanar) planar_syndrome_b(L, in + i * (uint64_t)(2 * L * L), defects + i * (uint64_t)(2 * L * (L - 1)));
    else surf_syndrome_b(code, L, in + i * (uint64_t)(L * L), defects + i * (uint64_t)((L + 1) * (L + 1)));
}

// one word picks one of the generators of a plaquette code (xzzx_model.py:439-452 draws five uniforms, planar_model.py:343-352
// three, for the same uniform choice), in table order (surf_gen_rco)
__device__ __forceinline__ void surf_pick(int code, int L, uint32_t w, bool top, int &row, int &col, int &op)
{
    const uint32_t G = (uint32_t)surf_ngen(code, L);
    surf_gen_rco(code, L, (int)(top ? scale_u32(w, G) : pick_top20(w, G)), row, col, op);
}

// p_x^nx p_y^ny p_z^nz p_I^nI from the host-built power tables (mcmc_biased.py:31,43): IEEE products in the
// reference's left-to-right order, so the value is bit-identical to the CPU's
__device__ __forceinline__ double biased_weight_b(const double *tbl, int nq, const uint8_t *m)
{
    int nx = 0, ny = 0, nz = 0;
    for (int i = 0; i < nq; ++i) { nx += m[i] == 1; ny += m[i] == 2; nz += m[i] == 3; }
    const int T = nq + 1;
    return tbl[nx] * tbl[T + ny] * tbl[2 * T + nz] * tbl[3 * T + (nq - nx - ny - nz)];
}

// Chain.update_chain(iters), src/mcmc.py:19-43, and Chain_biased.update_chain, src/mcmc_biased.py:20-59,
// one thread per chain, state in HBM.  Draw addressing is the oracle's (DESIGN.md "RNG addressing").
__global__ void k_chain_update(const ChainArgs a)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.N) return;
    const int L = a.L, code = a.code, nq = code_nq(code, L);
    uint8_t *m = a.states + i * nq;
    const uint32_t syn = a.first_syndrome + (uint32_t)i;
    auto thr = [&](int dE) { return a.acc_tbl[dE]; };       // ceil(f^dE * 2^32), dE in [1, nq]
    const bool top = a.thr_logical != 0;
    const double pb = a.noise ? biased_weight_b(a.bias_tbl, nq, m) : 0.0;       // mcmc_biased.py:28-31 (never refreshed: Q3)
    bool any_acc = false;
    for (uint64_t j = 0; j < a.iters; ++j) {
        const uint64_t k = a.k0 + j;
        // top: block (k, 0).  non-top: ONE word, word k&3 of block (k>>2, 1) -- top 20 bits pick the generator, low 12 bits
        // lead the 44-bit acceptance uniform that word k&3 of block (k>>2, kSubRefine) completes
        u32x4 x = philox_block(top ? k : k >> 2, top ? 0u : 1u, syn, a.slot, a.seed_lo, a.seed_hi);
        uint64_t v44 = 0;
        if (!top) {
            const u32x4 r = philox_block(k >> 2, kSubRefine, syn, a.slot, a.seed_lo, a.seed_hi);
            const int w = (int)(k & 3);
            x.x = w == 0 ? x.x : w == 1 ? x.y : w == 2 ? x.z : x.w;
            v44 = ((uint64_t)(x.x & 0xFFFu) << 32) | (w == 0 ? r.x : w == 1 ? r.y : w == 2 ? r.z : r.w);
        }
        // ---- propose (in place; XOR moves are involutions, so a rejected move is undone by re-applying it)
        int dE, row = 0, col = 0, op = 0, op0 = 0, op1 = 0, x0 = 0, z0 = 0, x1 = 0, z1 = 0;
        const bool logical = top && (uint64_t)x.x < a.thr_logical;                 // mcmc.py:23
        if (logical) {
            if (code == kCodeToric) {
                op0 = x.y >> 30; op1 = x.z >> 30;                                   // toric_model.py:234
                if (op0 == 1 || op0 == 2) x0 = scale_low30(x.y, L);                 // :241-248 (positions share block (k,0))
                if (op0 == 3 || op0 == 2) z0 = scale_u16(x.w >> 16, L);
                if (op1 == 1 || op1 == 2) x1 = scale_low30(x.z, L);
                if (op1 == 3 || op1 == 2) z1 = scale_u16(x.w & 0xFFFFu, L);
                dE = toric_apply_logical_b(L, m, op0, 0, x0, z0) + toric_apply_logical_b(L, m, op1, 1, x1, z1);
            } else {
                op0 = x.y >> 30;                                                    // xzzx_model.py:346-355
                if (op0 == 1 || op0 == 2) x0 = scale_low30(x.y, L);
                if (op0 == 3 || op0 == 2) z0 = scale_u16(x.w >> 16, L);
                dE = surf_apply_logical_b(code, L, m, op0, x0, z0);
            }
        } else if (code == kCodeToric) {
            // one word picks one of the 2L^2 generators (toric_model.py:291-295): X plaquettes first, row-major
            const uint32_t g = top ? scale_u32(x.y, 2u * L * L) : pick_top20(x.x, 2u * L * L), rc = g < (uint32_t)(L * L) ? g : g - L * L;
            row = rc / L; col = rc % L; op = g < (uint32_t)(L * L) ? 1 : 3;
            dE = toric_apply_stabilizer_b(L, m, row, col, op);
        } else {
            surf_pick(code, L, top ? x.y : x.x, top, row, col, op);
            dE = surf_apply_stabilizer_b(code, L, m, row, col, op);
        }
        // ---- accept?
        bool acc;
        if (a.noise) {                                                              // mcmc_biased.py:40-46 / :53-59
            // top: word 2 of the proposal's own block (the plaquette codes' logical draws leave it unused)
            const double u = top ? (double)x.z * (1.0 / 4294967296.0)
                                 : (double)v44 * (1.0 / 17592186044416.0);         // 2^-44, exact
            acc = u < biased_weight_b(a.bias_tbl, nq, m) / pb;
        } else if (top) {                                                           // mcmc.py:30-34
            acc = a.acc_all || dE <= 0;
            if (!acc) acc = philox_block(k, 2, syn, a.slot, a.seed_lo, a.seed_hi).x < thr(dE);
        } else {
            acc = dE <= 0 || a.acc_all || v44 < a.acc44[dE];                        // mcmc.py:42 (a generator: dE <= 4)
        }
        any_acc |= acc;
        if (!acc) {
            if (logical) {
                if (code == kCodeToric) { toric_apply_logical_b(L, m, op1, 1, x1, z1); toric_apply_logical_b(L, m, op0, 0, x0, z0); }
                else surf_apply_logical_b(code, L, m, op0, x0, z0);
            } else if (code == kCodeToric) toric_apply_stabilizer_b(L, m, row, col, op);
            else surf_apply_stabilizer_b(code, L, m, row, col, op);
        }
    }
    if (a.accepted != nullptr) a.accepted[i] = any_acc;
}

#define QECMC_LAUNCH(kern, N, s, ...)                                                    \
    do {                                                                                 \
        if ((N) == 0) return hipSuccess;                                                 \
        hipLaunchKernelGGL(kern, dim3(grid_for(N)), dim3(kBlock), 0, s, __VA_ARGS__);    \
        return hipGetLastError();                                                        \
    } while (0)

hipError_t launch_apply_stabilizer(int code, int L, uint64_t N, const uint8_t *in, uint8_t *out, const int32_t *rows,
                                   const int32_t *cols, const int32_t *ops, int32_t *dE, hipStream_t s)
{
    QECMC_LAUNCH(k_apply_stabilizer, N, s, code, L, N, in, out, rows, cols, ops, dE);
}
hipError_t launch_apply_logical(int code, int L, uint64_t N, const uint8_t *in, uint8_t *out, const int32_t *ops,
                                const int32_t *layers, const int32_t *xpos, const int32_t *zpos, int32_t *dE,
                                hipStream_t s)
{
    QECMC_LAUNCH(k_apply_logical, N, s, code, L, N, in, out, ops, layers, xpos, zpos, dE);
}
hipError_t launch_count_errors(int nq, uint64_t N, const uint8_t *in, int64_t *n, hipStream_t s)
{
    QECMC_LAUNCH(k_count_errors, N, s, nq, N, in, n);
}
hipError_t launch_eq_class(int code, int L, uint64_t N, const uint8_t *in, int32_t *cls, hipStream_t s)
{
    QECMC_LAUNCH(k_eq_class, N, s, code, L, N, in, cls);
}
hipError_t launch_to_class(int L, uint64_t N, const uint8_t *in, uint8_t *out, const int32_t *eq, hipStream_t s)
{
    QECMC_LAUNCH(k_to_class, N, s, L, N, in, out, eq);
}
hipError_t launch_syndrome(int code, int L, uint64_t N, const uint8_t *in, uint8_t *defects, hipStream_t s)
{
    QECMC_LAUNCH(k_syndrome, N, s, code, L, N, in, defects);
}
hipError_t launch_chain_update(const ChainArgs &a, hipStream_t s)
{
    QECMC_LAUNCH(k_chain_update, a.N, s, a);
}

}  // namespace qecmc
