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
    const bool xyz = a.xyz_thr != nullptr;                                      // Chain_xyz (mcmc.py:106-114,162-173): acceptance by the change of (n_x, n_y, n_z)
    const double pb = (a.noise && !xyz) ? biased_weight_b(a.bias_tbl, nq, m) : 0.0;       // mcmc_biased.py:28-31 (never refreshed: Q3)
    int cx = 0, cy = 0, cz = 0;                                                 // qubit_errors of the chain (:111), carried like the reference does (:172)
    if (xyz) for (int q = 0; q < nq; ++q) { cx += m[q] == 1; cy += m[q] == 2; cz += m[q] == 3; }
    bool any_acc = false;
    for (uint64_t j = 0; j < a.iters; ++j) {
        const uint64_t k = a.k0 + j;
        // top: the packed block (k >> 1, kSubTopPair), two proposals per block: words A, B = 2 (k & 1), 2 (k & 1) + 1 (philox.hpp).
        // non-top: ONE word, word k&3 of block (k>>2, 1) -- top 20 bits pick the generator, low 12 bits lead the 44-bit
        // acceptance uniform that word k&3 of block (k>>2, kSubRefine) completes.  Under the biased / alpha rules the top chain's
        // word B is such a word too: its top 20 bits pick the generator (a logical operator: Z_pos = B[31:16]), its low 12 bits
        // lead the acceptance uniform that word 2 (k & 1) + 1 of block (k >> 1, kSubRefine) completes.
        const bool packed = top;
        u32x4 x = philox_block(packed ? k >> 1 : k >> 2, packed ? kSubTopPair : 1u, syn, a.slot, a.seed_lo, a.seed_hi);
        const uint32_t pA = (k & 1) ? x.z : x.x, pB = (k & 1) ? x.w : x.y;
        uint64_t v44 = 0;
        if (!top) {
            const u32x4 r = philox_block(k >> 2, kSubRefine, syn, a.slot, a.seed_lo, a.seed_hi);
            const int w = (int)(k & 3);
            x.x = w == 0 ? x.x : w == 1 ? x.y : w == 2 ? x.z : x.w;
            v44 = ((uint64_t)(x.x & 0xFFFu) << 32) | (w == 0 ? r.x : w == 1 ? r.y : w == 2 ? r.z : r.w);
        } else if (a.noise) {
            const u32x4 r = philox_block(k >> 1, kSubRefine, syn, a.slot, a.seed_lo, a.seed_hi);
            v44 = ((uint64_t)(pB & 0xFFFu) << 32) | ((k & 1) ? r.w : r.y);
        }
        // ---- propose (in place; XOR moves are involutions, so a rejected move is undone by re-applying it)
        int dE, row = 0, col = 0, op = 0, op0 = 0, op1 = 0, x0 = 0, z0 = 0, x1 = 0, z1 = 0;
        const bool logical = packed && (uint64_t)(pA >> 16) < ((a.thr_logical + 65535u) >> 16);   // mcmc.py:23 with a 16-bit select: A[31:16] < ceil(p_logical * 2^16)
        if (logical) {
            if (packed && code != kCodeToric) {
                // A = select[31:16] | op[15:14] | X_pos[13:0];  Z_pos = B[31:16]
                op0 = (pA >> 14) & 3u;                                              // xzzx_model.py:346-355
                if (op0 == 1 || op0 == 2) x0 = ((pA & 0x3FFFu) * (uint32_t)L) >> 14;
                if (op0 == 3 || op0 == 2) z0 = scale_u16(pB >> 16, L);
                dE = surf_apply_logical_b(code, L, m, op0, x0, z0);
            } else {
                // A = select[31:16] | op0[15:14] | op1[13:12] | X_pos0[11:0];  B = Z_pos0[31:21] | X_pos1[20:10] | Z_pos1[9:0]
                op0 = (pA >> 14) & 3u; op1 = (pA >> 12) & 3u;                       // toric_model.py:234
                if (op0 == 1 || op0 == 2) x0 = ((pA & 0xFFFu) * (uint32_t)L) >> 12; // :241-248
                if (op0 == 3 || op0 == 2) z0 = ((pB >> 21) * (uint32_t)L) >> 11;
                if (op1 == 1 || op1 == 2) x1 = (((pB >> 10) & 0x7FFu) * (uint32_t)L) >> 11;
                if (op1 == 3 || op1 == 2) z1 = ((pB & 0x3FFu) * (uint32_t)L) >> 10;
                dE = toric_apply_logical_b(L, m, op0, 0, x0, z0) + toric_apply_logical_b(L, m, op1, 1, x1, z1);
            }
        } else if (code == kCodeToric) {
            // one word picks one of the 2L^2 generators (toric_model.py:291-295): X plaquettes first, row-major
            const uint32_t g = packed ? scale_u32(pB, 2u * L * L) : pick_top20(x.x, 2u * L * L), rc = g < (uint32_t)(L * L) ? g : g - L * L;
            row = rc / L; col = rc % L; op = g < (uint32_t)(L * L) ? 1 : 3;
            dE = toric_apply_stabilizer_b(L, m, row, col, op);
        } else {
            surf_pick(code, L, packed ? pB : x.x, top && !a.noise, row, col, op);   // (depolarizing top chain: the whole word B picks)
            dE = surf_apply_stabilizer_b(code, L, m, row, col, op);
        }
        // ---- accept?
        bool acc;
        int nx = 0, ny = 0, nz = 0;
        if (xyz) {                                                                  // mcmc.py:166-170: u < (factors ** change).prod()
            for (int q = 0; q < nq; ++q) { nx += m[q] == 1; ny += m[q] == 2; nz += m[q] == 3; }     // _count_errors_xyz of the proposal
            acc = v44 < a.xyz_thr[((nx - cx + 4) * 9 + (ny - cy + 4)) * 9 + (nz - cz + 4)];
            if (acc) { cx = nx; cy = ny; cz = nz; }
        } else
        if (a.noise) {                                                              // mcmc_biased.py:40-46 / :53-59
            const double u = (double)v44 * (1.0 / 17592186044416.0);               // 2^-44, exact
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
// ---- syndrome generation on the device (generate_data.py:57-60,110-131) --------------------------------------------
// Toric_code.generate_random_error(p) (toric_model.py:15-23: a qubit errs with probability p, its Pauli uniform on {1,2,3}) and
// xzzx_code / RotSurCode / Planar_code.generate_random_error(p_x, p_y, p_z) (xzzx_model.py:16-30, rotated_surface_model.py:25-38,
// planar_model.py:18-40: one uniform r per qubit: r < p_z -> Z, < p_z + p_x -> X, < p_z + p_x + p_y -> Y).  Qubit q of syndrome s
// draws words (2 (q & 1), 2 (q & 1) + 1) of Philox block (q >> 1, sub 0) of stream kGenStream: two qubits per block, one
// thread per block; HBM-bound byte work (one byte written per qubit, twice when the raw errors are kept).
__global__ void k_generate_errors(const GenArgs a)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t npairs = (uint32_t)(a.nq + 1) / 2;
    if (i >= a.N * (uint64_t)npairs) return;
    const uint64_t s = i / npairs;
    const uint32_t b = (uint32_t)(i - s * npairs);
    const u32x4 x = philox_block(b, 0, a.first_syndrome + (uint32_t)s, kGenStream, a.seed_lo, a.seed_hi);
    const uint32_t u1[2] = {x.x, x.z}, u2[2] = {x.y, x.w};
    for (int h = 0; h < 2; ++h) {
        const uint32_t q = 2 * b + h;
        if (q >= (uint32_t)a.nq) break;
        uint8_t v = 0;
        if (a.code == kCodeToric) {
            if ((uint64_t)u1[h] < a.thr_z) v = (uint8_t)(1u + scale_u32(u2[h], 3u));            // thr_z = ceil(p 2^32)
        } else {
            const uint64_t r = u1[h];
            v = r < a.thr_z ? 3 : r < a.thr_zx ? 1 : r < a.thr_zxy ? 2 : 0;
            if (a.code == kCodePlanar && q >= (uint32_t)(a.L * a.L)) {                            // layer 1 lives on its first L-1 rows / columns
                const uint32_t rc = q - (uint32_t)(a.L * a.L), row = rc / (uint32_t)a.L, col = rc - row * (uint32_t)a.L;
                if (row == (uint32_t)a.L - 1u || col == (uint32_t)a.L - 1u) v = 0;
            }
        }
        a.out[s * (uint64_t)a.nq + q] = v;
        if (a.raw) a.raw[s * (uint64_t)a.nq + q] = v;
    }
}

// define_equivalence_class of the raw errors (generate_data.py:121-122) and one apply_random_logical on top (:131;
// toric_model.py:228-253, xzzx_model.py:340-357): the operator fields of block (0, 1) of the generation stream, laid out as in
// a top-chain proposal block (word 1: op / X position, word 2: layer 1's, word 3: the Z positions)
__global__ void k_hide_class(const GenArgs a)
{
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.N) return;
    uint8_t *m = a.out + s * (uint64_t)a.nq;
    const int L = a.L;
    if (a.eq_true) a.eq_true[s] = a.code == kCodeToric ? toric_eq_class_b(L, m) : surf_eq_class_b(a.code, L, m);
    if (!a.hide) return;
    const u32x4 x = philox_block(0, 1, a.first_syndrome + (uint32_t)s, kGenStream, a.seed_lo, a.seed_hi);
    if (a.code == kCodeToric) {
        const int op0 = (int)(x.y >> 30), op1 = (int)(x.z >> 30);
        const int x0 = (op0 == 1 || op0 == 2) ? (int)scale_low30(x.y, L) : 0, z0 = (op0 == 3 || op0 == 2) ? (int)scale_u16(x.w >> 16, L) : 0;
        const int x1 = (op1 == 1 || op1 == 2) ? (int)scale_low30(x.z, L) : 0, z1 = (op1 == 3 || op1 == 2) ? (int)scale_u16(x.w & 0xFFFFu, L) : 0;
        toric_apply_logical_b(L, m, op0, 0, x0, z0);
        toric_apply_logical_b(L, m, op1, 1, x1, z1);
    } else {
        const int op = (int)(x.y >> 30);
        const int xp = (op == 1 || op == 2) ? (int)scale_low30(x.y, L) : 0, zp = (op == 3 || op == 2) ? (int)scale_u16(x.w >> 16, L) : 0;
        surf_apply_logical_b(a.code, L, m, op, xp, zp);
    }
}

hipError_t launch_generate(const GenArgs &a, hipStream_t s)
{
    const uint64_t threads = a.N * (uint64_t)((a.nq + 1) / 2);
    if (threads == 0) return hipSuccess;
    hipLaunchKernelGGL(k_generate_errors, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (a.hide || a.eq_true) {
        hipLaunchKernelGGL(k_hide_class, dim3(grid_for(a.N)), dim3(kBlock), 0, s, a);
        e = hipGetLastError();
    }
    return e;
}

hipError_t launch_chain_update(const ChainArgs &a, hipStream_t s)
{
    QECMC_LAUNCH(k_chain_update, a.N, s, a);
}

}  // namespace qecmc
