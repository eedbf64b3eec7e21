// Random-scan parallel-tempering ladder kernel (gfx950): toric, XZZX, rotated and planar codes; depolarizing, biased and
// alpha acceptance rules; the reference's random scan and the systematic sweep.
//
// This is the reference's Markov chain (src/mcmc.py:19-43 Chain.update_chain,
// :94-103 Ladder.step, decoders.py:55-68 PTEQ bookkeeping) laid out for CDNA4:
//
//   * one workgroup = 64 syndromes x Nc ladder slots; wavefront w owns slot w
//     (temperature p_ladder[w]) of all 64 syndromes, lane l owns syndrome l.
//     Acceptance thresholds are therefore wave-uniform and the top slot's
//     logical-operator branch (mcmc.py:23) never diverges against the
//     stabilizer-only slots.
//   * every chain's qubit_matrix lives in LDS for the whole run, packed 2 bits
//     per qubit: word w of state s of lane l sits at dword (s*W + w)*64 + l, so a
//     wave's ds_read_b32 / ds_xor_b32 hit bank (l mod 32) whatever (s, w) each
//     lane picks: random-scan access with zero bank conflicts, and addresses
//     need one add only.
//   * proposals: one Philox4x32-10 block feeds TWO non-top proposals (a word that picks one of the G generators,
//     g = (x * G) >> 32, and the acceptance word, each); the generator's sites come from an LDS table; accept tests
//     are integer compares against host-built thresholds ceil(f^dE * 2^32), so results are bit-identical to
//     the CPU oracle fed the same Philox stream.
//   * the top slot sits at p = 0.75 where every proposal is accepted
//     (mcmc.py:30): its moves are fire-and-forget LDS XORs (no reads, no dE) and
//     its error count is recounted once per ladder step.
//   * swaps (mcmc.py:96-103) move slot->state indices, not data; error counts and
//     equivalence classes are carried incrementally per state (n += dE; class ^=
//     logical delta) instead of recounted (mcmc.py:88-89, toric_model.py:317).
//   * one barrier per ladder step: every wave publishes its slot record (error
//     count, state id, class, flag) and then replays the whole top-down swap
//     cascade (mcmc.py:96-103) itself from those records to learn which state
//     lands in its own slot -- a few integer ops per rung (the swap uniform is turned into the largest accepted
//     error-count difference before the barrier), no serial section
//     that leaves seven waves idle, no second barrier.
//   * HBM traffic is compulsory only: nq bytes in, ncls counters out per syndrome.
#pragma once
#include "kernels.hpp"
#include "philox.hpp"
#include "stencil_bytes.hpp"   // kCode* constants

namespace qecmc {

// LDS carve-up in dwords (keep in sync with the kernel)
__host__ __device__ inline int ladder_group_dwords(int Nc, int W, int ncls, int gen_dwords)
{
    // st + info[2] + swx[2] + hist + thrT + swapT + stop flag (16) + generator table
    int d = Nc * W * 64 + 4 * Nc * 64 + ncls * 64 + Nc * 9 + Nc * kSwapFast + 16;
    d = (d + 3) & ~3;          // 16-byte aligned generator table (ds_read_b128 entries)
    return d + ((gen_dwords + 3) & ~3);
}

__device__ __forceinline__ uint32_t nnz2(uint32_t x) { return __popc((x | (x >> 1)) & 0x55555555u); }

__device__ __forceinline__ uint32_t sel4(const u32x4 &b, int i)
{
    return i == 0 ? b.x : i == 1 ? b.y : i == 2 ? b.z : b.w;
}

// single-instruction forms: the hardware uses only bits [4:0] of a shift amount / field offset, so the upper bits of the
// table entry that supplies them need no masking
__device__ __forceinline__ uint32_t bfe2_lo5(uint32_t w, uint32_t e)
{
    return __builtin_amdgcn_ubfe(w, e, 2u);                 // v_bfe_u32 w, e, 2
}
// (a << k) | b in one instruction (left to itself the compiler merges three fields with three shifts and two ors)
__device__ __forceinline__ uint32_t lshl_or(uint32_t a, int k, uint32_t b)
{
    uint32_t r;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(k), "v"(b));
    return r;
}

// a ^ byte 1 of b / (byte 1 of v) << (e & 31): the byte select rides on the instruction (SDWA), no shift or mask of its own
__device__ __forceinline__ uint32_t xor_byte1(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_xor_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t shl_byte1(uint32_t e, uint32_t v)
{
    uint32_t r;
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(e), "v"(v));
    return r;
}

__device__ __forceinline__ uint32_t shl_lo5(uint32_t v, uint32_t e)
{
    return v << (e & 31u);                                  // v_lshlrev_b32: the mask folds away
}

__device__ __forceinline__ void lds_xor(uint32_t *p, uint32_t v)
{
    __hip_atomic_fetch_xor(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_xor_b32, no return
}

// The four qubits of stabilizer (row, col, op), toric_model.py:261-269, as flat
// indices into uint8[2][L][L]:  X: (1,r,c) (0,r,c) (1,r,c-1) (0,r-1,c)
//                               Z: (1,r,c) (0,r,c) (0,r,c+1) (1,r+1,c)
__device__ __forceinline__ void toric_sites(uint32_t L, uint32_t LL, uint32_t row, uint32_t col, uint32_t isX,
                                            uint32_t q[4])
{
    const uint32_t rL = row * L;
    const uint32_t cm = (col == 0 ? L : col) - 1, cp = (col + 1 == L ? 0 : col + 1);
    const uint32_t rm = (row == 0 ? L : row) - 1, rp = (row + 1 == L ? 0 : row + 1);
    const uint32_t cn = isX ? cm : cp, rn = isX ? rm : rp;
    q[0] = LL + rL + col;
    q[1] = rL + col;
    q[2] = rL + cn + (isX ? LL : 0u);
    q[3] = rn * L + col + (isX ? 0u : LL);
}

// class of a packed state, toric_model.py:317-351: X component = b0^b1 (values 1,2),
// Z component = b1 (values 2,3); parity is linear, so XOR the words first.
__device__ __forceinline__ uint32_t toric_class_packed(const uint32_t *sb, int W, int LL)
{
    const int wb = LL >> 4;
    const uint32_t lowmask = (1u << ((LL & 15) * 2)) - 1u;    // layer-0 fields of the boundary word
    uint32_t acc0 = 0, acc1 = 0;
    for (int w = 0; w < W; ++w) {
        const uint32_t x = sb[w * 64];
        if (w < wb) acc0 ^= x;
        else if (w > wb) acc1 ^= x;
        else { acc0 ^= x & lowmask; acc1 ^= x & ~lowmask; }
    }
    const uint32_t x1 = __popc((acc0 ^ (acc0 >> 1)) & 0x55555555u) & 1u, z1 = __popc(acc0 & 0xAAAAAAAAu) & 1u;
    const uint32_t x2 = __popc((acc1 ^ (acc1 >> 1)) & 0x55555555u) & 1u, z2 = __popc(acc1 & 0xAAAAAAAAu) & 1u;
    return x1 + 2u * z1 + 4u * x2 + 8u * z2;
}

// ---- XZZX / rotated codes on the packed state (one L x L layer) ---------------------------------
// internal class value v = xparity | zparity << 1; rotated class = v (rotated_surface_model.py:411-420),
// XZZX class = v ^ (v >> 1) (xzzx_model.py:474-486 maps (1,0)->1, (1,1)->2, (0,1)->3)
__device__ __forceinline__ uint32_t surf_class_packed(int code, const uint32_t *sb, int L)
{
    uint32_t x = 0, z = 0;
    for (int i = 0; i < L; ++i) {
        const uint32_t qa = (uint32_t)i, qb = (uint32_t)(i * L);
        const uint32_t fa = (sb[(qa >> 4) * 64] >> ((qa & 15u) * 2u)) & 3u;    // row 0
        const uint32_t fb = (sb[(qb >> 4) * 64] >> ((qb & 15u) * 2u)) & 3u;    // column 0
        const uint32_t xa = (fa ^ (fa >> 1)) & 1u, za = fa >> 1, xb = (fb ^ (fb >> 1)) & 1u, zb = fb >> 1;   // X / Z components
        if (code == kCodeXzzx) {            // row 0 counts Y, X at even i, Z at odd i; column 0 the other way round
            x ^= (i & 1) ? za : xa;
            z ^= (i & 1) ? xb : zb;
        } else if (code == kCodePlanar) {   // X/Y parity of layer 0's first column, Z/Y parity of its first row (planar_model.py:379-390)
            x ^= xb;
            z ^= za;
        } else {
            x ^= xa;
            z ^= zb;
        }
    }
    return x | (z << 1);
}

// counts of X, Y, Z fields in one packed word
__device__ __forceinline__ void count_xyz(uint32_t w, int &nx, int &ny, int &nz)
{
    const uint32_t b0 = w & 0x55555555u, b1 = (w >> 1) & 0x55555555u;
    nx += __popc(b0 & ~b1); ny += __popc(b1 & ~b0); nz += __popc(b0 & b1);
}

// slot record published once per ladder step: error count | state id << 16 | class << 24 | flag << 31
// (flag = "has been at the top since it last reached the bottom", Chain.flag, mcmc.py:75,99-103)
__device__ __forceinline__ uint32_t pack_info(uint32_t n, uint32_t sid, uint32_t cls, uint32_t flag)
{
    return n | (sid << 16) | (cls << 24) | (flag << 31);
}

// CONV: build with the error_based convergence criterion (its per-lane window sums cost ~10 VGPRs,
// so fixed-step runs use the instantiation without it)
// GSPLIT: the expanded generator table of the toric random-scan path is stored as two halves kGenSplit entries apart
// (sites 0,1 | sites 2,3): one ds_read2_b64 with a constant second offset costs ~15 LDS cycles for random entries,
// 16 adjacent bytes ~21 (tools/ubench_lds.hip).  Needs n_gen <= kGenSplit (toric L <= 11); unused by the BIASED instantiations.
// exp(y) for y <= 0 from IEEE multiply / add / fma only: the same operation sequence as the oracle's orc_det_exp, so the
// swap decision of the alpha ladder is bit-identical on both sides.  y >= 0 returns 1; below 2^-1022 flushes to 0.
__device__ inline double det_exp(double y)
{
#pragma clang fp contract(off)
    if (!(y < 0.0)) return 1.0;
    if (y < -745.0) return 0.0;
    const double t = y * 1.4426950408889634;
    const int k = (int)(t - 0.5);
    double r = __builtin_fma(-(double)k, 6.93147180369123816490e-01, y);
    r = __builtin_fma(-(double)k, 1.90821492927058770002e-10, r);
    double q = 1.0 / 6227020800.0;
    q = __builtin_fma(q, r, 1.0 / 479001600.0);
    q = __builtin_fma(q, r, 1.0 / 39916800.0);
    q = __builtin_fma(q, r, 1.0 / 3628800.0);
    q = __builtin_fma(q, r, 1.0 / 362880.0);
    q = __builtin_fma(q, r, 1.0 / 40320.0);
    q = __builtin_fma(q, r, 1.0 / 5040.0);
    q = __builtin_fma(q, r, 1.0 / 720.0);
    q = __builtin_fma(q, r, 1.0 / 120.0);
    q = __builtin_fma(q, r, 1.0 / 24.0);
    q = __builtin_fma(q, r, 1.0 / 6.0);
    q = __builtin_fma(q, r, 0.5);
    q = __builtin_fma(q, r, 1.0);
    q = __builtin_fma(q, r, 1.0);
    if (k < -1022) return 0.0;
    return q * __longlong_as_double((long long)(k + 1023) << 52);
}

// u < (pz_lo / pz_hi) ** (n_eff_hi - n_eff_lo), mcmc_alpha.py:118-123, with n_eff = n_z + alpha (n_x + n_y) rebuilt from the
// packed counts exactly as the reference forms it (:58) and the power taken as det_exp(e * ln(base))
__device__ inline bool alpha_flip(uint32_t x, uint32_t hi, uint32_t lo, double alpha, double lnb)
{
#pragma clang fp contract(off)
    const double ne_hi = (double)(hi & 0xFFFFu) + alpha * (double)(hi >> 16);
    const double ne_lo = (double)(lo & 0xFFFFu) + alpha * (double)(lo >> 16);
    const double e = ne_hi - ne_lo;
    return (double)x * (1.0 / 4294967296.0) < det_exp(e * lnb);
}

// conv_crit_error_based_PT_alpha, decoders_biasednoise.py:229-238: |mean Q2 - mean Q4| < eps on the n_eff series, each mean
// formed as (sum n_z + alpha sum n_xy) / len from exact integer sums
__device__ inline bool alpha_series_close(uint64_t z2, uint64_t xy2, uint32_t den2, uint64_t z4, uint64_t xy4, uint32_t den4,
                                          double alpha, double eps)
{
#pragma clang fp contract(off)
    const double q2 = ((double)z2 + alpha * (double)xy2) / (double)den2;
    const double q4 = ((double)z4 + alpha * (double)xy4) / (double)den4;
    return fabs(q2 - q4) < eps;
}

// CODE / BIASED: code model (toric, xzzx, rotated) and acceptance rule (src/mcmc.py or src/mcmc_biased.py).
// The tuned paths are toric + depolarizing; the other combinations share the staging, cascade and bookkeeping.
// SCAN: false = the reference's random scan; true = systematic sweep (proposal k tests generator k mod G):
// sites are wave-uniform scalars and one Philox block feeds four proposals.
// GENTOP: keep the table-driven general top-chain path (toric L > 16, or a 1-chain ladder whose top sits below
// p = 0.75; always needed by the plaquette codes and the biased rule).  The common toric configurations compile it out,
// which keeps its registers out of the hot loop.
// ALPHA (with BIASED): the alpha noise model's ladder (slot-bound n_eff records, floating-point swap test, mcmc_alpha.py)
// DELUT (toric; the table sits in the >= 64 idle entries between the halves of a split generator table, 2 L^2 <= 191, or behind
// an unsplit one, 2 L^2 > 255): a proposal's dE comes from a 512-byte LDS table indexed by the four old fields and the generator's
// type instead of seven VALU instructions -- where the kernel is bound by VALU issue and the LDS array has room (L = 9: +4 %,
// L = 15: +3.3 %; L = 10, 11 -- three workgroups per CU, LDS-bound -- measured -1.5 % and keep the popcount).
// SSW: the swap sweep (mcmc.py:96-103) is run once, by wave 0, on the published records -- the raw swap uniforms against the
// threshold table -- and its result handed to the other waves through the idle half of the record buffer behind a second
// barrier, instead of every wave replaying the cascade on acceptance bounds two waves prepared.  Fewer VALU instructions
// (the kernel's bound) for a short serial section the other workgroups of the CU cover: +7 % where four 8-wave workgroups
// share a CU (toric L <= 9), -10 ... -14 % where the LDS footprint leaves two or three (measured on every family), so only the
// former are instantiated with it.
// QUEUE (with CONV; depolarizing rule, random scan, the framed top chain at p = 0.75): a persistent grid with a work queue for the runs
// that stop by the convergence criterion (decoders.py:74-82).  Stopping times spread over a decade (4e4 ... 3e5
// ladder steps at L = 9), and a lane whose syndrome has converged would otherwise idle until the slowest of its 64 finishes.
// Here a finished lane writes its results out at once, takes the next unassigned ladder from a global counter and starts
// it in place: every wave re-stages its slot's state for that lane, and the lane's Philox addresses are offset by the step it
// started at (aligned so that the four-proposal blocks of all lanes stay in phase).  A ladder's trajectory depends on its
// global index only, so the results are those of the one-ladder-per-lane launch, bit for bit.
// PRE: the top chain's Philox blocks are drawn ahead by the wave that will take the top role -- half of them two steps before,
// while it works on slot 1, the other half one step before on slot 0 -- and wait in registers (48 VGPRs: only for the shapes
// whose LDS footprint leaves 4 waves per SIMD anyway).  The draws do not depend on the state, so nothing changes but who
// is the step's longest wave: at L = 15 the top wave's 10 blocks + frame flush were 2.6 x a non-top wave's step.
//
// The variant is named, not positional: ladder_kernel<MAXT, MINW, CODE, kConv | kDelut | ...> (LadderFlag below).  A
// translation unit lists the flag sets it instantiates in one select_ladder_kernel<...>() call and asks for one of them at
// run time; a set that is not on the list yields no kernel (an error), never a neighbouring one.
enum LadderFlag : uint32_t {
    kConv = 1u << 0, kGsplit = 1u << 1, kBiased = 1u << 2, kScan = 1u << 3, kGentop = 1u << 4, kUset = 1u << 5, kAlpha = 1u << 6,
    kPre = 1u << 7, kDelut = 1u << 8, kQueue = 1u << 9, kSsw = 1u << 10,
};

// Diagnostic build only (tools/steptrace.hip): shader-clock stamps of one workgroup's waves at the phase boundaries of 32 ladder steps,
// behind the per-workgroup stamps of QECMC_TIMELINE in a.dbg.
#ifdef QECMC_STEPTRACE
#define QECMC_STAMP(k)                                                                                                         \
    do {                                                                                                                       \
        if (a.dbg && blockIdx.x == gridDim.x / 2 && t >= 2000 && t < 2032 && (threadIdx.x & 63) == 0)                          \
            a.dbg[(size_t)gridDim.x * 4 + (((t - 2000) * 16 + (threadIdx.x >> 6)) * 8 + (k))] = (k) == 5 ? (uint64_t)slot_u : (uint64_t)clock64(); \
    } while (0)
#else
#define QECMC_STAMP(k) ((void)0)
#endif

template <int MAXT, int MINW, int CODE, uint32_t FLAGS>
__global__ __launch_bounds__(MAXT, MINW) void ladder_kernel(const LadderArgs a)
{
    constexpr bool CONV = (FLAGS & kConv) != 0, GSPLIT = (FLAGS & kGsplit) != 0, BIASED = (FLAGS & kBiased) != 0, SCAN = (FLAGS & kScan) != 0;
    constexpr bool GENTOP = (FLAGS & kGentop) != 0, USET = (FLAGS & kUset) != 0, ALPHA = (FLAGS & kAlpha) != 0, PRE = (FLAGS & kPre) != 0;
    constexpr bool DELUT = (FLAGS & kDelut) != 0, QUEUE = (FLAGS & kQueue) != 0, SSW = (FLAGS & kSsw) != 0;
    static_assert(!ALPHA || BIASED, "the alpha model is a variant of the biased rule");
    static_assert(!QUEUE || CONV, "the work queue serves the runs that stop by the criterion");
    extern __shared__ uint32_t lds_all[];
    const int NC = a.Nc, W = a.W, L = a.L, LL = L * L, nq = a.nq, ncls = a.ncls;
    const int nthreads = NC * 64;                 // threads of one group
    const int tid = (int)threadIdx.x, lane = tid & 63, slot = tid >> 6;
    // generator table in LDS: 4 x u16 per generator as the plan stores it, or -- for the toric random-scan hot path --
    // expanded to 4 x u32 (byte offset << 16 | Pauli x 0x55 << 8 | Pauli << 5 | bit shift) so a site costs one add (its high
    // half, SDWA) and one bfe
    constexpr bool kWideGen = !SCAN;                           // every code and rule: the random-scan proposals read the expanded table
    constexpr bool kSplitGen = kWideGen && GSPLIT && !BIASED;  // (GSPLIT means something else in the BIASED instantiations)
    constexpr bool kNarrowGen = !kWideGen || CODE != kCodeToric;   // the plan's form: sweep, biased rule, plaquette-code top / general paths
    const int narrow_dw = kNarrowGen ? (kWideGen ? (2 * (int)a.n_gen + 3) & ~3 : 2 * (int)a.n_gen) : 0;
    // toric: 128 dwords behind an unsplit table for the dE look-up table (a split one keeps it in the idle entries between its halves);
    // plaquette codes: 256 bytes per Pauli pattern of their generators
    const int lut_tail = (!kWideGen || BIASED) ? 0 : CODE == kCodeToric ? ((int)a.n_gen > kGenSplit ? 128 : 0) : 64 * kLutTypes;
    const int wide_dw = kWideGen ? (kSplitGen ? 2 * (kGenSplit + (int)a.n_gen) : 4 * (int)a.n_gen) + lut_tail : 0;
    const int gen_dw = narrow_dw + wide_dw;
    constexpr bool alpha_noise = BIASED && ALPHA;               // mcmc_alpha.py: biased rule + slot-bound n_eff swap test
    // biased / alpha rules: the count-change table [n_types][256] and the packed counts of every state [NC][64]
    const int lm_dw = BIASED ? 2 * (L + 1) * W : 0;             // ... and the X / Z logical-operator masks [2][L+1][W] (row L = identity)
    // (xzzx: its operators do not depend on a position, so rows 0 .. 3 of the X table are overwritten with the four products
    // I, X, Z, XZ indexed by the class change -- one look-up per word instead of two and an xor)
    // (xzzx: mtab[n_gen], the logical operators' fields at every generator's four sites -- the top chain's frame, below)
    // ... and rk[3][64], the top chain's packed counts under the three logical products
    const int mtab_dw = (BIASED && CODE == kCodeXzzx) ? (((int)a.n_gen + 1) & ~1) + 3 * 64 : 0;
    const int neff_dw = alpha_noise ? 2 * NC * 64 : 0, bias_dw = BIASED ? 512 * a.n_types + NC * 64 + lm_dw + mtab_dw : 0;
    const int gen_region = (neff_dw || bias_dw) ? ((gen_dw + 3) & ~3) + neff_dw + bias_dw : gen_dw;
    const int gdw = ladder_group_dwords(NC, W, ncls, gen_region);   // dwords per group
    const int gen_off = gdw - ((gen_region + 3) & ~3);           // start of the generator table
    uint32_t *lds = lds_all;
    [[maybe_unused]] uint32_t *neffb = lds + gen_off + ((gen_dw + 3) & ~3);   // [2][NC][64] n_z | (n_x+n_y) << 16 per slot, by step parity
    // [n_types][256] x 8 bytes: .x = dx + (dz << 10) + ((dx + dy) << 20) (wrapping), .y = the same change as two halves (dx + dy | dz << 16, fp16)
    [[maybe_unused]] uint2 *xlut = reinterpret_cast<uint2 *>(lds + gen_off + ((gen_dw + 3) & ~3) + neff_dw);
    [[maybe_unused]] uint32_t *xyc = reinterpret_cast<uint32_t *>(xlut) + (BIASED ? 512 * a.n_types : 0);   // [NC][64] n_x | n_z << 10 | (n_x + n_y) << 20 of state s
    [[maybe_unused]] uint32_t *lml = xyc + (BIASED ? NC * 64 : 0);                    // [2][L+1][W] LDS copy of the plan's logical masks
    [[maybe_unused]] uint32_t *mtab = lml + lm_dw;                                    // [n_gen] 0 | M_X << 8 | M_Z << 16 | M_XZ << 24 (xzzx)
    [[maybe_unused]] uint32_t *rk = mtab + (((int)a.n_gen + 1) & ~1);                 // [3][64] (xzzx, the wave in the top role)

    uint32_t *st = lds;                           // [NC][W][64]   packed states
    uint32_t *info = st + (size_t)NC * W * 64;    // [2][NC][64]   slot records, double-buffered by step parity
    uint32_t *swx = info + 2 * NC * 64;           // [2][NC][64]   swap uniform of rung pair i, same parity
    uint32_t *hist = swx + 2 * NC * 64;           // [ncls][64]
    uint32_t *thrT = hist + ncls * 64;            // [NC][9]       sweep: accept iff x <= thrT[slot][dE+4]; random scan: the leading
                                                  //               12 bits of the 44-bit threshold, accept iff a12 < thrT[slot][dE+4]
    uint32_t *swapT = thrT + NC * 9;              // [NC][kSwapFast]  swap iff x < swapT[i][d]
    [[maybe_unused]] uint32_t *thrF = swapT + (NC - 1) * kSwapFast;   // [NC][4] (the idle last row of swapT): the low 32 bits of the
                                                  //               44-bit threshold of dE = 1..4, looked at when a12 == thrT[..]
    volatile uint32_t *stopf = swapT + NC * kSwapFast;   // [1]  every syndrome of the workgroup has converged
    [[maybe_unused]] const uint2 *gtab = reinterpret_cast<const uint2 *>(lds + gen_off);   // [n_gen] generator table (LDS copy)
    [[maybe_unused]] const uint4 *gtab4 = reinterpret_cast<const uint4 *>(lds + gen_off + narrow_dw);  // wide form (kWideGen)
    [[maybe_unused]] const uint2 *gtabw = reinterpret_cast<const uint2 *>(lds + gen_off + narrow_dw);  // ... as two halves
    [[maybe_unused]] auto gen_entry = [&](uint32_t g) -> uint4 {                          // one ds_read2_b64 either way
        if constexpr (kSplitGen) {
            const uint2 lo = gtabw[g], hi = gtabw[g + kGenSplit];
            return uint4{lo.x, lo.y, hi.x, hi.y};
        } else {
            return gtab4[g];
        }
    };

#if defined(QECMC_TIMELINE) || defined(QECMC_STEPTRACE)   // diagnostic build only (tools/timeline.hip, tools/steptrace.hip): per-workgroup start/end stamps and placement
    if (a.dbg && threadIdx.x == 0) {
        a.dbg[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memrealtime();
        a.dbg[blockIdx.x * 4 + 1] = ((uint64_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) |   // XCC_ID
                                    __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));                       // HW_ID
    }
#endif
    const uint64_t s0 = (uint64_t)blockIdx.x * 64u;
    const int cnt = a.N > s0 ? (int)((a.N - s0) < 64u ? (a.N - s0) : 64u) : 0;   // 0: a group past the end of the batch
    uint32_t syn = a.first_syndrome + (uint32_t)s0 + (uint32_t)lane;   // Philox ctr[2] (QUEUE: of the ladder the lane works on now)
    [[maybe_unused]] uint32_t qi = (uint32_t)s0 + (uint32_t)lane;      // QUEUE: that ladder's index in this launch (0xFFFFFFFF: none left)
    [[maybe_unused]] uint32_t t0 = 0;                                  // ... the workgroup step it started at
    [[maybe_unused]] uint64_t kq = 0;                                  // ... t0 * iters: its proposal indices are (t * iters + j) - kq

    for (int i = tid; i < NC * W * 64; i += nthreads) st[i] = 0;
    for (int i = tid; i < ncls * 64; i += nthreads) hist[i] = 0;
    if (tid == 0) { stopf[0] = 0; stopf[1] = 0; stopf[2] = 0; stopf[3] = 0; }   // [0], [1] stop; [2], [3] QUEUE: refill requested -- each by step parity:
                                                                                //  wave 0 writes the flag of step t+1 during step t, every wave reads it behind step t+1's barrier
    if (a.swap_acc != nullptr)
        for (int i = tid; i < 2 * NC * 64; i += nthreads) lds_all[gdw + i] = 0;
    if constexpr (USET) {
        for (int i = tid; i < 3 * 64; i += nthreads) hist[i] = i < 64 ? 2u * (uint32_t)LL : 0xFFFFFFFFu;   // decoders.py:140,242; "never"
    }
    if constexpr (kWideGen) {
        for (int i = tid; i < 4 * (int)a.n_gen; i += nthreads) {
            const uint32_t e = reinterpret_cast<const uint16_t *>(a.gen)[i], q = e >> 2;
            const int g = i >> 2, k = i & 3;                                       // generator, site
            // byte offset of the state dword [31:16] | in site 0: the four Paulis as 2-bit fields [15:8] | Pauli [6:5] | bit shift [4:0]
            uint32_t ops = 0;
            if (k == 0)
                for (int u = 0; u < 4; ++u) ops |= (uint32_t)(reinterpret_cast<const uint16_t *>(a.gen)[4 * g + u] & 3u) << (2 * u);
            if (BIASED) ops = k == 1 ? a.gen_type[g] : (e & 3u);                    // site 1, bits [11:8]: the generator's Pauli-pattern id; the others: their own Pauli
            if (!BIASED && CODE == kCodeToric && k == 1) ops = e & 3u;              // toric: site 1, byte 1 = the generator's one Pauli
            if (DELUT && CODE != kCodeToric && !BIASED && (k & 1)) ops = e & 3u;      // plaquette codes with the dE table: sites 1, 3: byte 1 = the site's own Pauli
            if (DELUT && k == 2) ops = CODE == kCodeToric ? (e & 3u) == 3u : a.gen_type[g];   // ... site 2, byte 1 = the row of the dE table (toric: 1 for a Z generator)
            (lds + gen_off + narrow_dw)[kSplitGen ? 2 * (g + (k >> 1) * kGenSplit) + (k & 1) : i] =
                (((q >> 4) * 256u) << 16) | (ops << 8) | ((e & 3u) << 5) | ((q & 15u) * 2u);
        }
    }
    if constexpr (kNarrowGen) {
        for (int i = tid; i < 2 * (int)a.n_gen; i += nthreads) (lds + gen_off)[i] = reinterpret_cast<const uint32_t *>(a.gen)[i];
    }
    // 4 (dE + 4) for old fields F (bits 0-7) under an X (index bit 8 clear) or Z generator: toric_model.py:275-282 tabulated.
    // It sits in the idle entries between the two halves of the split generator table, or behind the unsplit one.
    const int delut_dw = gen_off + narrow_dw + (lut_tail ? wide_dw - lut_tail : 2 * (int)a.n_gen);
    [[maybe_unused]] const uint8_t *delut = reinterpret_cast<const uint8_t *>(lds + delut_dw);
    if constexpr (DELUT) {
        static_assert(kWideGen && !BIASED, "no room for the dE look-up table in this layout");
        for (int t = 0; t < (CODE == kCodeToric ? 2 : a.n_types); ++t) {           // one row per Pauli pattern (toric: X, Z)
            const uint32_t ops = CODE == kCodeToric ? (t ? 0xFFu : 0x55u) : (uint32_t)a.type_ops[t];
            for (int i = tid; i < 256; i += nthreads) {
                const uint32_t F = (uint32_t)i, G = F ^ ops;
                const uint32_t v = __popc((G | (G >> 1)) & 0x55u) + __popc(~(F | (F >> 1)) & 0x55u);
                reinterpret_cast<uint8_t *>(lds + delut_dw)[256 * t + i] = (uint8_t)(4u * v);
            }
        }
    }
    for (int i = tid; i < (NC - 1) * kSwapFast; i += nthreads) {
        // u < p_diff^d  <=>  x < thr; d = 0 always swaps and is never looked up (mcmc.py:146-149)
        const int pr = i / kSwapFast, d = i - pr * kSwapFast;
        swapT[i] = (d >= 1 && d <= nq) ? (uint32_t)a.swap_thr[(size_t)pr * (nq + 1) + d] : 0u;
    }
    if constexpr (BIASED) {
        for (int i = tid; i < 256 * a.n_types; i += nthreads) {
            // the change of (n_x + n_y, n_z) once more, as two fp16 numbers: what the fast acceptance test multiplies (each in [-4, 4])
            const uint32_t d = a.xyz_lut[i], v = d + (512u | (512u << 10) | (512u << 20));
            const _Float16 hxy = (_Float16)((int)(v >> 20) - 512), hz = (_Float16)((int)((v >> 10) & 1023u) - 512);
            xlut[i] = uint2{d, (uint32_t)__builtin_bit_cast(uint16_t, hxy) | ((uint32_t)__builtin_bit_cast(uint16_t, hz) << 16)};
        }
        if constexpr (CODE == kCodeXzzx) {
            // the fields of the X (anti-diagonal) and Z (diagonal) logical operators at every generator's sites (a null site: 0)
            for (int g = tid; g < (int)a.n_gen; g += nthreads) {
                uint32_t mx = 0, mz = 0;
                for (int k = 0; k < 4; ++k) {
                    const uint32_t e = reinterpret_cast<const uint16_t *>(a.gen)[4 * g + k], q = e >> 2;
                    if (e == 0) continue;
                    mx |= ((a.lmask[q >> 4] >> ((q & 15u) * 2u)) & 3u) << (2 * k);
                    mz |= ((a.lmask[(L + 1) * W + (q >> 4)] >> ((q & 15u) * 2u)) & 3u) << (2 * k);
                }
                mtab[g] = (mx << 8) | (mz << 16) | ((mx ^ mz) << 24);
            }
        }
        for (int i = tid; i < lm_dw; i += nthreads) lml[i] = a.lmask[i];              // kinds 0 (X) and 1 (Z) are the first two tables
        if (CODE == kCodeXzzx && L >= 3) {
            __syncthreads();                                                         // (rows 0 .. 3 of the copy above are replaced)
            for (int i = tid; i < 4 * W; i += nthreads) {                            // (a 1-rung ladder has 64 threads: 4 W can exceed them)
                const int c = i / W, w = i - c * W;                                  // product c = ax | az << 1, word w
                lml[i] = ((c & 1) ? a.lmask[w] : 0u) ^ ((c & 2) ? a.lmask[(L + 1) * W + w] : 0u);
            }
        }
    }
    if (tid < NC * 9) {
        // u < f^dE  <=>  x < thr  <=>  x <= thr-1;  dE <= 0 (f^dE >= 1) and f >= 1 always accept (mcmc.py:30,42)
        const int c = tid / 9, d = tid - c * 9 - 4;
        const bool always = d <= 0 || ((a.acc_all_mask >> c) & 1u);
        if constexpr (SCAN) {
            thrT[tid] = always ? 0xFFFFFFFFu : a.acc_thr[c][d - 1] - 1u;
        } else {
            // non-top random-scan proposals compare a 44-bit uniform (a12 * 2^32 + w) with T44 = ceil(f^dE * 2^44)
            thrT[tid] = always ? 4096u : (uint32_t)(a.acc_thr44[c][d - 1] >> 32);
            if (d >= 1) thrF[c * 4 + d - 1] = (uint32_t)a.acc_thr44[c][d - 1];
        }
    }
    __syncthreads();

    // ---- stage the batch: coalesced byte stream -> 2-bit fields in LDS -------------
    const uint32_t R = a.replicas;                // ladders per syndrome (>= 1)
    if (!a.resume) {
        const uint8_t *src = a.init + (R > 1 ? 0ull : s0 * (uint64_t)nq);
        const int total = cnt * nq;
        for (int o = tid; o < total; o += nthreads) {
            const int j = o / nq, q = o - j * nq;
            const uint32_t v = (R > 1 ? src[((s0 + (uint64_t)j) / R) * (uint64_t)nq + q] : src[o]) & 3u;   // ladder l starts from init row l / R
            if (v) {
                const uint32_t bits = v << ((q & 15) * 2);
                uint32_t *p = st + (q >> 4) * 64 + j;
                for (int s = 0; s < NC; ++s)      // Ladder.__init__ deep-copies init into every slot (mcmc.py:72)
                    __hip_atomic_fetch_or(p + s * W * 64, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    } else {
        const uint8_t *src = a.states + s0 * (uint64_t)NC * nq;
        const int per = NC * nq, total = cnt * per;
        for (int o = tid; o < total; o += nthreads) {
            const uint32_t v = src[o] & 3u;
            if (v) {
                const int j = o / per, rem = o - j * per, s = rem / nq, q = rem - s * nq;
                __hip_atomic_fetch_or(st + (s * W + (q >> 4)) * 64 + j, v << ((q & 15) * 2), __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();

    // every wave carries its slot's current state id, error count, class and flag in registers
    uint32_t sid = (uint32_t)slot, n, cls, flag = (slot == NC - 1);   // chains[-1].flag = 1 (mcmc.py:75)
    {
        const uint32_t *sb = st + slot * W * 64 + lane;
        n = 0;
        for (int w = 0; w < W; ++w) n += nnz2(sb[w * 64]);
        cls = CODE == kCodeToric ? toric_class_packed(sb, W, LL) : surf_class_packed(CODE, sb, L);
    }
    uint32_t tops0 = 0, samples = 0;              // per-syndrome counters live in wave 0
    // convergence criterion of decoders.py:74-82,93-105 (wave 0 only): window sums over the logged
    // bottom-chain error counts, Q2 = series[l/4 : l/2], Q4 = series[3l/4 : l]
    uint32_t burn = 0, conv_start = 0, conv_streak = 0, done = 0, steps_done = 0, conv_ok = 0;
    // conv_mult early stop of the unique-chain droplets (USET): replicated on every wave of the ladder
    [[maybe_unused]] uint32_t cm_done = 0, cm_steps = 0, cm_last = 0;
    [[maybe_unused]] uint32_t *cm_short = hist;          // [64]     shortest chain the droplet has seen (the histogram rows are idle in USET runs)
    [[maybe_unused]] uint32_t *cm_trig = hist + 64;      // [2][64]  last step (of each parity) that found a new chain no longer than that
    uint64_t sumA = 0, sumB = 0;
    [[maybe_unused]] uint64_t sumAxy = 0, sumBxy = 0;          // alpha noise: window sums of n_x + n_y (sumA / sumB hold n_z)
    if (a.resume && lane < cnt) {
        flag = a.flags[(s0 + lane) * NC + slot] != 0;
        if (slot == 0) tops0 = a.tops0[s0 + lane];
    }
    // kLdsCounters (the register-starved instantiations): wave 0's per-syndrome counters wait in LDS between its bookkeeping
    // blocks -- the idle last row of each parity's swap-uniform buffer (rung pairs 0 .. NC-2 exist; the QUEUE kernels keep their
    // refill records there) -- instead of occupying two registers of every wave for the whole run
    // (every 64-VGPR instantiation but the toric fixed-length family, which fits its registers as it is)
    constexpr bool kLaunderLane = MINW >= 8 && (BIASED || CONV || GENTOP || USET || SCAN || CODE != kCodeToric);
    constexpr bool kLdsCounters = kLaunderLane && !QUEUE;
    [[maybe_unused]] uint32_t *ctrT = swx + (NC - 1) * 64, *ctrS = swx + (2 * NC - 1) * 64;
    if constexpr (kLdsCounters) {
        if (slot == 0) { ctrT[lane] = tops0; ctrS[lane] = 0; }
        tops0 = 0;
    }
    if constexpr (BIASED) {
        // every state's (n_x, n_y, n_z), packed; carried per state from here on (accepted moves add their change)
        int nx = 0, ny = 0, nz = 0;
        for (int w = 0; w < W; ++w) count_xyz(st[(slot * W + w) * 64 + lane], nx, ny, nz);
        xyc[slot * 64 + lane] = (uint32_t)nx | ((uint32_t)nz << 10) | ((uint32_t)(nx + ny) << 20);
        if (alpha_noise) {
            // Chain_alpha.__init__ (mcmc_alpha.py:18-22) on a fresh ladder; the carried attributes on resume.
            // Parity 1 is "the step before step 0".
            uint32_t v;
            if (a.resume) v = lane < cnt ? a.neff[(s0 + lane) * NC + slot] : 0u;
            else v = (uint32_t)nz | ((uint32_t)(nx + ny) << 16);
            neffb[(NC + slot) * 64 + lane] = v;
        }
        __syncthreads();
    }

    // (the slot record waits in one register while the wave-uniform set-up below is formed)
    [[maybe_unused]] uint32_t rec0 = pack_info(n, sid, cls, flag);
    if constexpr (kLaunderLane) asm volatile("" : "+v"(rec0));

    // Roles rotate: at every ladder step each wave moves on to the next slot, so the heavier top
    // slot (frame flush + recount) visits every SIMD in turn instead of loading one of them for
    // the whole run.  Results do not depend on which wave computes a slot.
    const uint32_t wave_u = __builtin_amdgcn_readfirstlane(slot);
    uint32_t slot_u = wave_u;                                   // the slot this wave works on in the current step
    const uint32_t iters = a.iters;
    // the wave's diagonal stream (philox.hpp): slot_u + T is constant while roles rotate downwards (QUEUE: per lane, T counts
    // from the step the lane's ladder started at), and the last non-top block drawn, for the step that continues it
    uint32_t dstrm = kDiagStream + (wave_u + (uint32_t)(a.step0 % (uint64_t)NC)) % (uint32_t)NC;
    // table entries fetched one proposal ahead (random_scan_loop): same-box A/B +1.9 % at rotated L = 21 and at toric L = 13 with 9 rungs,
    // +0.5 % at toric L = 15 with 15 rungs, -2.4 % at toric L = 15 with 8 (512 threads, dE table) -- which keeps the plain order
    constexpr bool kAhead = MINW < 8 && (CODE != kCodeToric || MAXT > 512);
    // (the carry, measured against a build without it on the same box: +1 % on the 8-waves-per-SIMD shapes, +1.5 ... +1.8 % on the 4-wave ones)
    [[maybe_unused]] u32x4 carry{0, 0, 0, 0};
    [[maybe_unused]] uint64_t carry_kb = ~0ull;
    const uint32_t thrL1 = (uint32_t)(a.thr_logical - 1);      // x < thr_logical <=> x <= thr_logical-1 (thr in [1, 2^32])
    // the packed toric top chain's 16-bit select: A[31:16] < thr16 = ceil(p_logical * 2^16)  <=>  A <= (thr16 << 16) - 1
    [[maybe_unused]] const uint32_t thrA1 = (uint32_t)((((a.thr_logical + 65535u) >> 16) << 16) - 1u);
    const uint32_t Lodd = L & 1;                                // a row/column operator flips L parities
    const uint32_t rowbits = 2u * (uint32_t)L;                  // bits of one lattice row in the packed stream
    const uint32_t rowmask = rowbits >= 32 ? 0xFFFFFFFFu : (1u << rowbits) - 1u;
    const bool swap_fast = a.swap_fast_ok != 0;

    constexpr int kPre = 12;                                    // blocks drawn ahead (a step of more proposals draws the rest in place)
    [[maybe_unused]] u32x4 pre[PRE ? kPre : 1];
    // QUEUE: finished lanes take new ladders until the counter runs out; the loop ends by the stop flag.  A new ladder may
    // start only at a step that keeps its four-proposal blocks in phase with the others': t0 * iters = 0 (mod 4).
    [[maybe_unused]] const uint32_t q_period = (iters & 3u) == 0 ? 1u : (iters & 1u) == 0 ? 2u : 4u;
    [[maybe_unused]] uint32_t *qidx = swx + (NC - 1) * 64, *qt0 = swx + (2 * NC - 1) * 64;   // the idle last rows of swx: [64] each
    [[maybe_unused]] bool q_dead = lane >= cnt, q_flushed = false;                           // (wave 0) no ladder left for this lane / its results are written
    [[maybe_unused]] bool q_empty = a.N <= (uint64_t)gridDim.x * 64u;                        // ... the counter is exhausted (uniform)
    if constexpr (kLaunderLane) {
        asm volatile("" : "+v"(rec0));
        n = rec0 & 0xFFFFu; sid = (rec0 >> 16) & 0xFFu; cls = (rec0 >> 24) & 0x3Fu; flag = rec0 >> 31;
    }
    for (uint64_t t = 0; QUEUE || t < a.nsteps; ++t) {
        // The lane index of this step.  In the register-starved instantiations (kLaunderLane) it is opaque to the compiler, so the
        // dozen LDS addresses derived from it (records, swap uniforms, histogram rows, count tables ...) are formed where a step
        // uses them -- one add each -- instead of being hoisted out of the step loop into registers the 64-VGPR cap then spills.
        int lane_t = lane;
        if constexpr (kLaunderLane) { lane_t = (int)__lane_id(); asm volatile("" : "+v"(lane_t)); }   // (v_mbcnt: nothing derived from threadIdx stays live)
        // Issue arbitration between co-resident workgroups is oldest-first, which lets the first one
        // race ahead and leaves the last one alone (latency-bound, 2 waves per SIMD) at the end of a
        // launch.  Lowering a workgroup's priority as it advances (cyclically, every 8 steps) narrows
        // that spread: +6 % on a one-round grid (measured), neutral otherwise.
        // (Tried in round 2: the top-role wave at the highest priority instead: -10 % at L = 9, +3 % at L = 15, 0 at rotated L = 21.)
        switch (3u - (uint32_t)((t >> 3) & 3)) {    // s_setprio takes an immediate
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
        QECMC_STAMP(5); QECMC_STAMP(0);
        // ---------------- Chain.update_chain(iters) on every slot (mcmc.py:81-83) -----------
        uint32_t *stw = st + sid * W * 64 + lane_t;
        const uint64_t kbase = a.prop0 + t * iters;
        const bool acc_all = (a.acc_all_mask >> slot_u) & 1u;
        const bool top_logical = (slot_u == (uint32_t)(NC - 1)) && a.thr_logical != 0;
        const uint32_t strm = top_logical ? slot_u : dstrm;                         // Philox stream of this step's proposals (philox.hpp)
        // (the top-role wave at the highest priority: same-box A/B +2.2 % / +1.6 % in the 1024-thread PRE kernels -- L = 15 with 15 rungs, L = 13
        // with 9 --, -1.6 % in the 512-thread ones, configs 3 and 5, which keep the workgroup's level)
        if constexpr (PRE && MAXT > 512) { if (top_logical) __builtin_amdgcn_s_setprio(3); }
        // The blind top chain's proposals (random scan, p = 0.75): move(A, B) for each proposal of the step, in order.  The step's
        // proposals [kbase, kbase + iters) lie in blocks b0 .. b0 + nblk - 1, two per block: the first block may start at its
        // second proposal and the last may end at its first (all wave-uniform).
        [[maybe_unused]] auto top_blocks = [&](auto &&move) {
            const uint64_t b0 = (kbase >> 1) - (kq >> 1);                           // (kq is a multiple of 4)
            const bool skip_first = (kbase & 1) != 0, skip_last = ((kbase + iters) & 1) != 0;
            const uint32_t nblk = (uint32_t)(((kbase + iters - 1) >> 1) - (kbase >> 1)) + 1u;
            auto both = [&](const u32x4 &x, uint32_t bi) {
                if (!(skip_first && bi == 0)) move(x.x, x.y);
                if (!(skip_last && bi == nblk - 1)) move(x.z, x.w);
            };
            uint32_t bi = 0;
            if constexpr (PRE) {
                if (t >= 2 && NC >= 3) {                                            // (the first two top steps of a launch had no earlier role)
                    const uint32_t pn = nblk < (uint32_t)kPre ? nblk : (uint32_t)kPre;
#pragma unroll
                    for (int jj = 0; jj < kPre; ++jj)
                        if ((uint32_t)jj < pn) both(pre[jj], (uint32_t)jj);
                    bi = pn;
                }
            }
            // two blocks' (four proposals') Philox chains in flight: this wave is the step's longest and often runs alone
            for (; bi + 1 < nblk; bi += 2) {
                const u32x4 xa = philox_block(b0 + bi, kSubTopPair, syn, slot_u, a.seed_lo, a.seed_hi);
                const u32x4 xb = philox_block(b0 + bi + 1, kSubTopPair, syn, slot_u, a.seed_lo, a.seed_hi);
                both(xa, bi);
                both(xb, bi + 1);
            }
            if (bi < nblk) both(philox_block(b0 + bi, kSubTopPair, syn, slot_u, a.seed_lo, a.seed_hi), bi);
        };
        const uint32_t *myT = thrT + slot_u * 9 + 4;
        // sweep (scan = 1) of a top chain at f = 1 with table-driven logical masks: used by the plaquette codes and by
        // toric L > 16 (the L <= 16 toric top chain has the frame-based fast path below)
        [[maybe_unused]] auto blind_sweep_tables = [&]() {
            const uint32_t *lmask = a.lmask;
            const int LW = (L + 1) * W;
            uint32_t gs = (uint32_t)(kbase % a.n_gen), cdelta = 0;
            u32x4 coins{0, 0, 0, 0};
            uint64_t cb_cur = ~0ull;
            for (uint32_t j = 0; j < iters; ++j) {
                const uint64_t k = kbase + j;
                if ((k & 7) == 0) {                                                 // one random logical operator
                    const u32x4 x = philox_block(k, 0, syn, strm, a.seed_lo, a.seed_hi);
                    const uint32_t *m0 = lmask + L * W, *m1 = m0, *m2 = m0, *m3 = m0;   // identity rows
                    if (CODE == kCodeToric) {
                        const uint32_t op0 = x.y >> 30, op1 = x.z >> 30;
                        const uint32_t dx0 = (op0 ^ (op0 >> 1)) & 1u, dz0 = op0 >> 1, dx1 = (op1 ^ (op1 >> 1)) & 1u, dz1 = op1 >> 1;
                        if (dx0) m0 = lmask + scale_low30(x.y, L) * W;
                        if (dz0) m1 = lmask + LW + scale_u16(x.w >> 16, L) * W;
                        if (dx1) m2 = lmask + 2 * LW + scale_low30(x.z, L) * W;
                        if (dz1) m3 = lmask + 3 * LW + scale_u16(x.w & 0xFFFFu, L) * W;
                        cdelta ^= (L & 1) ? (dx0 | (dz0 << 1) | (dx1 << 2) | (dz1 << 3)) : 0u;
                    } else {
                        const uint32_t op = x.y >> 30;
                        const uint32_t xp = ((op ^ (op >> 1)) & 1u) ? scale_low30(x.y, L) : 0u, zp = (op >> 1) ? scale_u16(x.w >> 16, L) : 0u;
                        const uint32_t ax = CODE == kCodeXzzx ? ((op ^ (op >> 1)) & 1u) : (op & 1u), az = op >> 1;
                        if (ax) m0 = lmask + xp * W;
                        if (az) m1 = lmask + LW + zp * W;
                        cdelta ^= ax | (az << 1);
                    }
                    for (int w = 0; w < W; ++w) lds_xor(stw + w * 64, m0[w] ^ m1[w] ^ m2[w] ^ m3[w]);
                }
                if ((k >> 7) != cb_cur) { cb_cur = k >> 7; coins = philox_block(cb_cur, 3, syn, strm, a.seed_lo, a.seed_hi); }
                const uint2 ev = gtab[gs];
                gs = gs + 1 == a.n_gen ? 0u : gs + 1;
                if ((sel4(coins, (int)((k >> 5) & 3)) >> (k & 31)) & 1u) {
                    const uint32_t ent[4] = {ev.x & 0xFFFFu, ev.x >> 16, ev.y & 0xFFFFu, ev.y >> 16};
                    for (int i = 0; i < 4; ++i) lds_xor(stw + (ent[i] >> 6) * 64, (ent[i] & 3u) << (((ent[i] >> 2) & 15u) * 2u));
                }
            }
            uint32_t cnt_n = 0;
            for (int w = 0; w < W; ++w) cnt_n += nnz2(stw[w * 64]);
            n = cnt_n;
            cls ^= cdelta;
        };

        // ---------- the non-top random-scan loop (every code, depolarizing rule) -------------------------------------------
        // ONE Philox word per proposal (word k&3 of block (k>>2, 1), so a block feeds four): its top 20 bits pick the
        // generator, g = floor(x20 * G / 2^20) (the G generators as equally likely as 20 bits allow: G 2^-20; the proposal stays
        // symmetric, so the stationary law is untouched) and its low 12 bits lead the 44-bit acceptance uniform.  Word k&3 of
        // the refinement block (k>>2, kSubRefine) supplies the other 32 bits, and is computed only when some lane's 12 bits
        // tie with its threshold's (once in 4096 proposals per lane).
        [[maybe_unused]] auto random_scan_loop = [&]() {
            int ni = DELUT ? (int)(4u * n) : (int)n;
            const uint32_t *myF = thrF + slot_u * 4 - 5;                            // indexed by dE + 4 = 5..8
            auto fetch = [&](uint32_t xw) -> uint4 {
                // (20-bit field x 11-bit count: a full-rate 24-bit multiply)
                uint32_t gi = (uint32_t)__mul24((int)(xw >> 12), (int)a.n_gen) >> 20;
                asm("" : "+v"(gi));                                                 // (keeps index and address as shift + shift-add)
                return gen_entry(gi);                                               // the (up to) four sites; an unused entry is 0
            };
            auto propose_e = [&](uint32_t xw, const uint4 ev, uint64_t kb, auto wsel) {
                const uint32_t sh[4] = {ev.x, ev.y, ev.z, ev.w};                    // byte offset << 16 | ... | Pauli << 5 | bit shift
                uint32_t *ad[4];
                uint32_t f[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ad[i] = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(stw) + (sh[i] >> 16));   // byte offset: one SDWA add
                    f[i] = bfe2_lo5(*ad[i], sh[i]);
                }
                const uint32_t F = lshl_or(lshl_or(f[3], 2, f[2]), 4, lshl_or(f[1], 2, f[0]));
                if constexpr (DELUT) {
                    // 4 (dE + 4) from the table; the thresholds' rows and the running count (ni = 4 n) take it as a byte offset
                    const uint32_t v = delut[(ev.z & 0xF00u) | F];
                    const uint32_t a12 = xw & 0xFFFu, tI = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(myT - 4) + v);
                    bool acc = a12 < tI;                                            // mcmc.py:42 (dE <= 0: tI = 4096)
                    if (a12 == tI) {                                                // rare (a lane in 4096): the next 32 bits decide
                        constexpr int WI = decltype(wsel)::value;
                        const u32x4 r = philox_block(kb - (kq >> 2), kSubRefine, syn, strm, a.seed_lo, a.seed_hi);
                        acc = (WI == 0 ? r.x : WI == 1 ? r.y : WI == 2 ? r.z : r.w) < *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(myF) + v);
                    }
                    if (acc) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            lds_xor(ad[i], CODE == kCodeToric ? shl_byte1(sh[i], ev.y) : (i & 1) ? shl_byte1(sh[i], sh[i]) : shl_lo5((sh[i] >> 5) & 3u, sh[i]));
                        ni += (int)v - 16;
                    }
                    return;
                }
                const uint32_t G = xor_byte1(F, ev.x);                              // the four new values (byte 1 of site 0: the Paulis as 2-bit fields)
                // dE + 4 = #(new != 0) + #(old == 0) (toric_model.py:275-282) as ONE popcount: the nonzero new fields marked on the
                // even bits, the zero old fields on the odd bits ((F << 1) | 0x55..55 has every even bit set and F's low field bits
                // moved up, so ~(u | F) is 1 on odd bit 2i+1 iff field i is 0 -- and on the 12 odd bits above the byte, a constant
                // the threshold row's base absorbs).  An unused entry reads site 0 into both and counts 1.
                const uint32_t nzG = (G | (G >> 1)) & 0x55u;
                const uint32_t uF = lshl_or(F, 1, 0x55555555u);
                const uint32_t dE16 = __popc(__builtin_amdgcn_bitop3_b32(uF, F, nzG, 0xAB));   // ~(uF | F) | nzG;  = dE + 16
                const uint32_t a12 = xw & 0xFFFu, tI = (myT - 16)[dE16];
                bool acc = a12 < tI;                                                // mcmc.py:42 (dE <= 0: tI = 4096)
                if (a12 == tI) {                                                    // rare (a lane in 4096): the next 32 bits decide
                    constexpr int WI = decltype(wsel)::value;
                    const u32x4 r = philox_block(kb - (kq >> 2), kSubRefine, syn, strm, a.seed_lo, a.seed_hi);
                    acc = (WI == 0 ? r.x : WI == 1 ? r.y : WI == 2 ? r.z : r.w) < (myF - 12)[dE16];
                }
                if (acc) {
                    if constexpr (CODE == kCodeToric) {                            // one Pauli for the whole generator: byte 1 of site 1
#pragma unroll
                        for (int i = 0; i < 4; ++i) lds_xor(ad[i], shl_byte1(sh[i], ev.y));
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) lds_xor(ad[i], shl_lo5((sh[i] >> 5) & 3u, sh[i]));
                    }
                    ni += (int)dE16 - 16;
                }
            };
            auto propose = [&](uint32_t xw, uint64_t kb, auto wsel) { propose_e(xw, fetch(xw), kb, wsel); };
            // the blocks that overlap [kbase, kbase + iters)
            uint64_t kb = kbase >> 2;
            for (int jb = -(int)((uint32_t)kbase & 3u); jb < (int)iters; jb += 4, ++kb) {
                // (the current block lives in `carry`: the one a step ends in is still there for the next step, which starts in it
                // unless the top role came between)
                if (kb != carry_kb) { carry = philox_block(kb - (kq >> 2), 1, syn, strm, a.seed_lo, a.seed_hi); carry_kb = kb; }
                const u32x4 &xa = carry;
                if (jb >= 0 && jb + 4 <= (int)iters) {                             // a whole block: no per-proposal range tests
                    if constexpr (kAhead) {
                        // the next proposal's table entry is fetched before this one's test (its LDS round trip is then off the chain
                        // entry -> sites -> dE -> threshold, which 4 waves per SIMD cover badly)
                        const uint4 e0 = fetch(xa.x), e1 = fetch(xa.y);
                        propose_e(xa.x, e0, kb, std::integral_constant<int, 0>{});
                        const uint4 e2 = fetch(xa.z);
                        propose_e(xa.y, e1, kb, std::integral_constant<int, 1>{});
                        const uint4 e3 = fetch(xa.w);
                        propose_e(xa.z, e2, kb, std::integral_constant<int, 2>{});
                        propose_e(xa.w, e3, kb, std::integral_constant<int, 3>{});
                    } else {
                    propose(xa.x, kb, std::integral_constant<int, 0>{});
                    propose(xa.y, kb, std::integral_constant<int, 1>{});
                    propose(xa.z, kb, std::integral_constant<int, 2>{});
                    propose(xa.w, kb, std::integral_constant<int, 3>{});
                    }
                } else {
                    if ((uint32_t)jb < iters) propose(xa.x, kb, std::integral_constant<int, 0>{});
                    if ((uint32_t)(jb + 1) < iters) propose(xa.y, kb, std::integral_constant<int, 1>{});
                    if ((uint32_t)(jb + 2) < iters) propose(xa.z, kb, std::integral_constant<int, 2>{});
                    if ((uint32_t)(jb + 3) < iters) propose(xa.w, kb, std::integral_constant<int, 3>{});
                }
            }
            n = DELUT ? (uint32_t)ni >> 2 : (uint32_t)ni;
        };

        if constexpr (CODE != kCodeToric || BIASED) {
            // ---------- XZZX / rotated codes and the biased acceptance rule ----------------------------
            const bool top = top_logical;
            if constexpr (BIASED) {
                // ---------- the biased / alpha acceptance rule (mcmc_biased.py:20-59, mcmc_alpha.py:27-70) ------------------------
                // accept iff u < p_n / p_b with p = px^nx py^ny pz^nz pI^nI from the reference's power tables and p_b frozen at
                // loop entry (quirk Q3).  The chain's counts travel packed (n_x | n_z << 10 | (n_x + n_y) << 20, carried per
                // state across steps); a generator move's count change comes from an LDS table indexed by the generator's Pauli
                // pattern and the four old fields.  The exact test costs eight table look-ups, six fp64 products and a
                // division, and almost no proposal needs it: log2(p_n / p_b) is lxy (D_x + D_y) + lz D_z with D the count change
                // since loop entry -- exactly, up to the roundings of the power tables (1e-15) -- so two fmas and one v_exp_f32 give
                // 2^12 p_n / p_b to 1e-2 of a unit, and only a proposal whose 12 leading uniform bits lie within one unit of
                // it (3 cells in 4096) evaluates the reference's expression.  Decisions are those of the exact test in every
                // case, hence bit-identical to the CPU.
                const int T1 = nq + 1;
                const double *bt = a.bias_tbl + (size_t)slot_u * 4 * T1;
                auto weight = [&](uint32_t P) -> double {                           // mcmc_biased.py:28-31 on packed counts
                    const int cx = (int)(P & 1023u), cz = (int)((P >> 10) & 1023u), cxy = (int)(P >> 20);
                    return bt[cx] * bt[T1 + (cxy - cx)] * bt[2 * T1 + cz] * bt[3 * T1 + (nq - cxy - cz)];
                };
                constexpr uint32_t kFieldBias = 512u | (512u << 10) | (512u << 20);
                uint32_t Np = xyc[sid * 64 + lane_t];
                // (state id, class and flag wait in one register while the proposals run: the loops below are the kernel's register peak)
                uint32_t rec3 = sid | (cls << 8) | (flag << 16);
                if constexpr (kLaunderLane) { asm volatile("" : "+v"(rec3)); sid = rec3 & 0xFFu; }
                // (the counts at loop entry -- p_b's -- stay in xyc until the step ends: the rare paths read them back from there)
                auto entry_counts = [&]() -> uint32_t { uint32_t v = xyc[sid * 64 + lane_t]; asm volatile("" : "+v"(v)); return v; };
                const float lxyf = a.bias_l2f[slot_u][0], lzf = a.bias_l2f[slot_u][1];
                // (a count changes by at most 4 per proposal since loop entry: the changes stay exact as fp16 integers up to 2048;
                // the host has checked iters <= 512 and 4 iters max|l| <= 2000 for this rung)
                const bool f32ok = (a.bias_f32ok >> slot_u) & 1u;
                uint32_t cdelta = 0;
                bool any_acc = false;
                // 2^12 p_n / p_b, to 1e-2 of a unit, from the packed counts of the proposal
                auto ratio12_packed = [&](uint32_t Nn, uint32_t Nb) -> float {
                    const uint32_t cand = Nn - (Nb - kFieldBias);                   // the counts' change since loop entry, every field + 512
                    const uint32_t cx = cand ^ kFieldBias;                          // ... as 10-bit two's complement fields
                    const int dxyi = __builtin_amdgcn_sbfe((int)cx, 20u, 10u), dzi = __builtin_amdgcn_sbfe((int)cx, 10u, 10u);
                    if (f32ok) {
                        // single precision is enough for the 12-bit comparison once the offsets are out (a field + 512 with its top bit
                        // flipped is the change in 10-bit two's complement): the products are exact inside the fmas, the roundings are
                        // those of l as a float (6e-8 |l d|) and of a partial sum of magnitude <= |l d| + 12, so with |l d| <= 2000 the
                        // error of the exponent stays below 2.5e-4 / 0.7 of a unit at e = 4096; a unit is what the margins below allow
                        return __builtin_amdgcn_exp2f(__builtin_fmaf(lxyf, (float)dxyi, __builtin_fmaf(lzf, (float)dzi, 12.0f)));
                    }
                    const double tl = __builtin_fma(a.bias_l2[slot_u][0], (double)dxyi, __builtin_fma(a.bias_l2[slot_u][1], (double)dzi, 12.0));
                    return __builtin_amdgcn_exp2f((float)tl);                       // relative error < 1e-6
                };
                // The test u < p_n / p_b on the 44-bit uniform u = (a12 2^32 + w) 2^-44 (philox.hpp): a12 = the proposal word's low 12
                // bits, e = 2^12 p_n / p_b within a unit.  u lies in [a12, a12 + 1) 2^-12: below the ratio for certain when
                // a12 + 2 <= e, above when a12 - 1 > e (for an integer i, i <= floor(e) <=> i <= e: the comparisons stay in floating
                // point, exact on 12-bit integers).  In between (3 cells in 4096) the reference's expression decides, and only a true
                // 12-bit tie draws w = refine(), the proposal's word of the refinement block.
                // (Both margins come from ONE subtraction: e - a12 is exact -- a12 is a multiple of e's ulp whenever e < 2^24 -- except when e is
                // far below a12, where its rounding moves the difference by < 1e-3 of a unit around -1: inside the 0.3 of a unit the
                // exponent's error budget leaves.  The proposal's packed counts are formed only where the exact expression needs them.)
                auto test12 = [&](float e, uint32_t word, auto &&counts, auto &&refine) -> bool {
                    const uint32_t a12 = word & 0xFFFu;
                    const float dm = e - (float)a12;
                    bool acc = dm >= 2.0f;
                    if (!acc && dm >= -1.0f) {
                        const double ratio = weight(counts()) / weight(entry_counts());   // mcmc_biased.py:44-46 (p_b's table addresses stay out of the hot loop)
                        const double ulo = (double)a12 * (1.0 / 4096.0);
                        acc = ulo + (1.0 / 4096.0) <= ratio;
                        if (!acc && ulo < ratio) {
                            const uint64_t v44 = ((uint64_t)a12 << 32) | refine();
                            acc = (double)v44 * (1.0 / 17592186044416.0) < ratio;
                        }
                    }
                    return acc;
                };
                // a generator's four sites: LDS addresses, table words (shift / Pauli), old fields as F = f3 f2 f1 f0
                auto sites = [&](const uint4 ev, uint32_t *(&sad)[4], uint32_t (&ssh)[4]) -> uint32_t {
                    ssh[0] = ev.x; ssh[1] = ev.y; ssh[2] = ev.z; ssh[3] = ev.w;
                    uint32_t f[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        sad[i] = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(stw) + (ssh[i] >> 16));   // the expanded entry: one add
                        f[i] = bfe2_lo5(*sad[i], ssh[i]);                           // (an unused entry reads site 0 and applies no Pauli)
                    }
                    return lshl_or(lshl_or(f[3], 2, f[2]), 4, lshl_or(f[1], 2, f[0]));
                };
                auto apply_gen = [&](uint32_t *const (&sad)[4], const uint32_t (&ssh)[4]) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)                                     // (byte 1 of sites 0, 2, 3 is the site's Pauli: one SDWA shift)
                        lds_xor(sad[i], i == 1 ? shl_lo5((ssh[i] >> 5) & 3u, ssh[i]) : shl_byte1(ssh[i], ssh[i]));
                };
                if (!top) {
                    // ---- non-top chains: word k&3 of block (k>>2, 1): 20 bits pick the generator, 12 lead the acceptance uniform.
                    // The count change since loop entry rides along as two fp16 integers (D_xy | D_z): one packed add forms the
                    // proposal's, two mixed-precision fmas its exponent.
                    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
                    half2_t Dh = {(_Float16)0.0f, (_Float16)0.0f};
                    auto one = [&](uint32_t xw, uint64_t kb, auto wsel) {
                        uint32_t gi = (uint32_t)__mul24((int)(xw >> 12), (int)a.n_gen) >> 20;   // (20-bit field x 11-bit count: the full-rate multiply)
                        asm("" : "+v"(gi));
                        const uint4 ev = gen_entry(gi);
                        uint32_t *sad[4];
                        uint32_t ssh[4];
                        const uint32_t F = sites(ev, sad, ssh);
                        const uint2 d = xlut[(ev.y & 0xF00u) | F];                  // pattern id << 8 | old fields -> the count change
                        const half2_t pD = Dh + __builtin_bit_cast(half2_t, d.y);
                        const float e = f32ok ? __builtin_amdgcn_exp2f(__builtin_fmaf((float)pD.x, lxyf, __builtin_fmaf((float)pD.y, lzf, 12.0f)))
                                              : ratio12_packed(Np + d.x, entry_counts());
                        const bool acc = test12(e, xw, [&]() -> uint32_t { return Np + d.x; }, [&]() -> uint32_t {
                            constexpr int WI = decltype(wsel)::value;
                            const u32x4 r = philox_block(kb - (kq >> 2), kSubRefine, syn, strm, a.seed_lo, a.seed_hi);
                            return WI == 0 ? r.x : WI == 1 ? r.y : WI == 2 ? r.z : r.w;
                        });
                        if (acc) {
                            apply_gen(sad, ssh);
                            Np += d.x;
                            Dh = pD;
                            any_acc = true;
                        }
                    };
                    // the blocks that overlap [kbase, kbase + iters), a static word index per proposal (+1 % against a rolled loop with a
                    // dynamic word select); the current block lives in `carry`, so the one a step ends in is still there for the next step
                    uint64_t kb = kbase >> 2;
                    for (int jb = -(int)((uint32_t)kbase & 3u); jb < (int)iters; jb += 4, ++kb) {
                        // (QUEUE: a ladder's proposal indices count from the step it started at, kq = t0 * iters, a multiple of 4)
                        if (kb != carry_kb) { carry = philox_block(kb - (kq >> 2), 1, syn, strm, a.seed_lo, a.seed_hi); carry_kb = kb; }
                        if ((uint32_t)jb < iters) one(carry.x, kb, std::integral_constant<int, 0>{});
                        if ((uint32_t)(jb + 1) < iters) one(carry.y, kb, std::integral_constant<int, 1>{});
                        if ((uint32_t)(jb + 2) < iters) one(carry.z, kb, std::integral_constant<int, 2>{});
                        if ((uint32_t)(jb + 3) < iters) one(carry.w, kb, std::integral_constant<int, 3>{});
                    }
                } else {
                    // ---- the top chain (mcmc_biased.py:32-46): two proposals per block (k >> 1, kSubTopPair), words A, B (philox.hpp):
                    // A[31:16] selects logical / generator, op = A[15:14], X_pos from A[13:0], Z_pos from B[31:16]; B's top 20 bits pick
                    // the generator and its low 12 bits lead the acceptance uniform, as in a non-top word.
                    //
                    // xzzx: the logical operators do not depend on a position (X on the anti-diagonal, Z on the diagonal,
                    // xzzx_model.py:291-311), so a logical proposal is O(1): the wave keeps R[d] = the packed counts of (state ^ mask_d)
                    // for the four products d = I, X, Z, XZ relative to the state the chain is in -- recounted when the step begins,
                    // updated by every accepted generator move with three more look-ups (the fields under mask d are F ^ M_d, M_d =
                    // the operator's fields at the generator's sites, mtab) -- so the proposal's counts are R[cd] and an accepted one
                    // permutes R.  The accepted operators are collected in a frame `fr` (the LDS state lags by mask_fr, so the fields
                    // read from it are corrected by M_fr) and applied once, when the step ends.
                    [[maybe_unused]] uint32_t fr8 = 0;                                // the frame, times 8 (a byte select of mtab's word)
                    [[maybe_unused]] uint32_t *rkl = rk + lane_t;                      // R[1 .. 3] at rkl[0], rkl[64], rkl[128]
                    const uint32_t Nb = Np;
                    [[maybe_unused]] const int LW = (L + 1) * W;
                    if constexpr (CODE == kCodeXzzx) {
                        uint32_t c1x = 0, c1z = 0, c1s = 0, c2x = 0, c2z = 0, c2s = 0, c3x = 0, c3z = 0, c3s = 0;
                        auto cnt = [](uint32_t v, uint32_t &cx, uint32_t &cz, uint32_t &cs) {
                            const uint32_t h = v >> 1;
                            cx += __popc(__builtin_amdgcn_bitop3_b32(v, h, 0x55555555u, 0x20));   // v & ~h & m: X
                            cz += __popc(__builtin_amdgcn_bitop3_b32(v, h, 0x55555555u, 0x80));   // v & h & m:  Z
                            cs += __popc(__builtin_amdgcn_bitop3_b32(v, h, 0x55555555u, 0x28));   // (v ^ h) & m: X or Y
                        };
                        for (int w = 0; w < W; ++w) {
                            const uint32_t v = stw[w * 64], mx = lml[W + w], mz = lml[2 * W + w];   // rows 0 .. 3: I, X, Z, XZ
                            cnt(v ^ mx, c1x, c1z, c1s);
                            cnt(v ^ mz, c2x, c2z, c2s);
                            cnt(v ^ mx ^ mz, c3x, c3z, c3s);
                        }
                        rkl[0] = c1x | (c1z << 10) | (c1s << 20); rkl[64] = c2x | (c2z << 10) | (c2s << 20); rkl[128] = c3x | (c3z << 10) | (c3s << 20);
                    }
                    auto one = [&](uint32_t A, uint32_t B, uint64_t kb, auto wsel) {
                        const bool logical = A <= thrA1;                            // mcmc.py:23 (A[31:16] < ceil(p_logical 2^16))
                        const uint32_t op = (A >> 14) & 3u;                         // xzzx_model.py:346 / rotated_surface_model.py:334
                        // applied operators: xzzx X iff op in {1,2}, Z iff op in {3,2}; rotated X iff op in {1,3}, Z iff op in {2,3}
                        const uint32_t ax = CODE == kCodeXzzx ? ((op ^ (op >> 1)) & 1u) : (op & 1u), az = op >> 1;
                        const uint32_t cd = ax | (az << 1);
                        uint32_t gi = (uint32_t)__mul24((int)(B >> 12), (int)a.n_gen) >> 20;
                        asm("" : "+v"(gi));
                        uint32_t *sad[4] = {stw, stw, stw, stw};
                        uint32_t ssh[4] = {0, 0, 0, 0};
                        uint32_t Nn, idx0 = 0, mw = 0;
                        [[maybe_unused]] const uint32_t *m0 = lml + L * W, *m1 = m0;   // identity rows (LDS copy of the plan's masks)
                        if constexpr (CODE == kCodeXzzx) {
                            // branch-free: every lane reads its generator (harmless where the proposal is a logical operator)
                            const uint4 ev = gen_entry(gi);
                            mw = mtab[gi];
                            const uint32_t F = sites(ev, sad, ssh);
                            idx0 = (ev.y & 0xF00u) | (F ^ __builtin_amdgcn_ubfe(mw, fr8, 8u));   // the fields of the chain's state: LDS ^ M_fr
                            const uint32_t Ngen = Np + xlut[idx0].x;
                            const uint32_t Rcd = rkl[((cd ? cd : 1u) - 1u) * 64u];
                            const uint32_t Nlog = cd ? Rcd : Np;
                            Nn = logical ? Nlog : Ngen;
                        } else if (logical) {
                            const uint32_t xp = ((op ^ (op >> 1)) & 1u) ? ((A & 0x3FFFu) * (uint32_t)L) >> 14 : 0u;   // drawn iff op in {1,2}
                            const uint32_t zp = (op >> 1) ? scale_u16(B >> 16, L) : 0u;                               // drawn iff op in {3,2}
                            if (ax) m0 = lml + xp * W;                              // kind 0: X on column X_pos
                            if (az) m1 = lml + LW + zp * W;                         // kind 1: Z on row Z_pos
                            int cx = 0, cy = 0, cz = 0;                             // the operator moves O(L) sites: recount the result
                            for (int w = 0; w < W; ++w) count_xyz(stw[w * 64] ^ m0[w] ^ m1[w], cx, cy, cz);
                            Nn = (uint32_t)cx | ((uint32_t)cz << 10) | ((uint32_t)(cx + cy) << 20);
                        } else {
                            const uint4 ev = gen_entry(gi);
                            const uint32_t F = sites(ev, sad, ssh);
                            Nn = Np + xlut[(ev.y & 0xF00u) | F].x;
                        }
                        const bool acc = test12(ratio12_packed(Nn, Nb), B, [&]() -> uint32_t { return Nn; }, [&]() -> uint32_t {
                            constexpr int WI = decltype(wsel)::value;               // 1 or 3: the word B's index
                            const u32x4 r = philox_block(kb - (kq >> 1), kSubRefine, syn, strm, a.seed_lo, a.seed_hi);
                            return WI == 1 ? r.y : r.w;
                        });
                        if (acc) {
                            any_acc = true;
                            if (logical) {
                                if constexpr (CODE == kCodeXzzx) {
                                    // R'[d] = R[d ^ cd]: two conditional pair swaps
                                    const bool s1 = (cd & 1u) != 0, s2 = (cd & 2u) != 0;
                                    uint32_t r0 = Np, r1 = rkl[0], r2 = rkl[64], r3 = rkl[128];
                                    { const uint32_t t0 = s1 ? r1 : r0, t1 = s1 ? r0 : r1, t2 = s1 ? r3 : r2, t3 = s1 ? r2 : r3; r0 = t0; r1 = t1; r2 = t2; r3 = t3; }
                                    { const uint32_t t0 = s2 ? r2 : r0, t1 = s2 ? r3 : r1, t2 = s2 ? r0 : r2, t3 = s2 ? r1 : r3; r0 = t0; r1 = t1; r2 = t2; r3 = t3; }
                                    Np = r0; rkl[0] = r1; rkl[64] = r2; rkl[128] = r3;
                                    fr8 ^= cd << 3;
                                } else {
                                    for (int w = 0; w < W; ++w) lds_xor(stw + w * 64, m0[w] ^ m1[w]);
                                    Np = Nn;
                                }
                                cdelta ^= cd;
                            } else {
                                apply_gen(sad, ssh);
                                Np = Nn;
                                if constexpr (CODE == kCodeXzzx) {
                                    // idx0 carries the chain's fields: those under mask d are F ^ M_d
                                    __hip_atomic_fetch_add(rkl, xlut[idx0 ^ ((mw >> 8) & 0xFFu)].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    __hip_atomic_fetch_add(rkl + 64, xlut[idx0 ^ ((mw >> 16) & 0xFFu)].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    __hip_atomic_fetch_add(rkl + 128, xlut[idx0 ^ (mw >> 24)].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                }
                            }
                        }
                    };
                    // the packed blocks that overlap [kbase, kbase + iters): the first may start at its second proposal, the last may
                    // end at its first (wave-uniform)
                    // (QUEUE: the block index is reduced by the ladder's own start, kq >> 1, where the block is drawn)
                    const uint64_t b0 = kbase >> 1;
                    const bool skip_first = (kbase & 1) != 0, skip_last = ((kbase + iters) & 1) != 0;
                    const uint32_t nblk = (uint32_t)(((kbase + iters - 1) >> 1) - b0) + 1u;
                    for (uint32_t bi = 0; bi < nblk; ++bi) {
                        const u32x4 x = philox_block(b0 + bi - (kq >> 1), kSubTopPair, syn, strm, a.seed_lo, a.seed_hi);
                        if (!(skip_first && bi == 0)) one(x.x, x.y, b0 + bi, std::integral_constant<int, 1>{});
                        if (!(skip_last && bi == nblk - 1)) one(x.z, x.w, b0 + bi, std::integral_constant<int, 3>{});
                    }
                    if constexpr (CODE == kCodeXzzx) {
                        // the step's accepted logical operators, applied at once (row fr of the product masks; row 0 = identity)
                        const uint32_t *mf = lml + (fr8 >> 3) * W;
                        for (int w = 0; w < W; ++w) lds_xor(stw + w * 64, mf[w]);
                    }
                }
                if constexpr (kLaunderLane) { asm volatile("" : "+v"(rec3)); sid = rec3 & 0xFFu; cls = (rec3 >> 8) & 0xFFu; flag = rec3 >> 16; }
                xyc[sid * 64 + lane_t] = Np;
                n = (Np >> 20) + ((Np >> 10) & 1023u);
                cls ^= cdelta;
                // the slot's n_eff is refreshed by accepted moves only (mcmc_alpha.py:58,70); otherwise it keeps the
                // value it had, possibly that of a configuration since swapped away (quirk Q4)
                if (alpha_noise)
                    neffb[((t & 1) * NC + slot_u) * 64 + lane_t] = any_acc ? (((Np >> 10) & 1023u) | ((Np >> 20) << 16))
                                                                         : neffb[(((t & 1) ^ 1) * NC + slot_u) * 64 + lane_t];
            } else {
            const bool xyz_rule = USET && a.xyz_thr != nullptr;                     // Chain_xyz: the general path with its own table
            if (!SCAN && !top && CODE != kCodeToric && !xyz_rule) {
                if constexpr (kWideGen) random_scan_loop();
            } else if (SCAN && !top && CODE != kCodeToric) {
                // sweep on a plaquette code: generator table lookup, 2 to 4 sites
                int ni = (int)n;
                uint32_t gs = SCAN ? (uint32_t)(kbase % a.n_gen) : 0u;
                u32x4 blk{0, 0, 0, 0};
                uint64_t kb_cur = ~0ull;
                for (uint32_t j = 0; j < iters; ++j) {
                    u32x4 x;
                    uint32_t g;
                    if constexpr (SCAN) {                                          // generator k mod G, accept word k&3 of block k>>2
                        const uint64_t k = kbase + j;
                        if ((k >> 2) != kb_cur) { kb_cur = k >> 2; blk = philox_block(kb_cur, 3, syn, strm, a.seed_lo, a.seed_hi); }
                        x.w = sel4(blk, (int)(k & 3));
                        g = gs;
                        gs = gs + 1 == a.n_gen ? 0u : gs + 1;
                    } else {
                        g = 0; x.w = 0;                                             // (SCAN is part of the branch condition)
                    }
                    const uint2 e = gtab[g];                                      // 4 x (site << 2 | pauli), 0 = no site
                    const uint32_t ent[4] = {e.x & 0xFFFFu, e.x >> 16, e.y & 0xFFFFu, e.y >> 16};
                    uint32_t *ad[4];
                    uint32_t sh[4], F = 0, OPS = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t q = ent[i] >> 2;
                        ad[i] = stw + (q >> 4) * 64;
                        sh[i] = (q & 15u) * 2u;
                        F |= ((*ad[i] >> sh[i]) & 3u) << (2 * i);
                        OPS |= (ent[i] & 3u) << (2 * i);
                    }
                    const uint32_t G = F ^ OPS;
                    const int dE = __popc((G | (G >> 1)) & 0x55u) - __popc((F | (F >> 1)) & 0x55u);
                    if (x.w <= myT[dE]) {                                           // mcmc.py:42
#pragma unroll
                        for (int i = 0; i < 4; ++i) lds_xor(ad[i], (ent[i] & 3u) << sh[i]);
                        ni += dE;
                    }
                }
                n = (uint32_t)ni;
            } else if (SCAN && top && acc_all) {
                blind_sweep_tables();
            } else if (top && acc_all) {
                // random scan, top chain at f = 1 (mcmc.py:30): every proposal is applied blindly, n recounted once.  With a row
                // of the lattice inside one word (L <= 16) the logical operators are collected in a per-lane frame and applied
                // once per step, as on the toric code: xzzx -- parities of the (position-independent) anti-diagonal X and
                // diagonal Z; rotated -- the set of X columns and of Z rows; planar -- the X rows and Z columns of layer 0.
                uint32_t cdelta = 0, frX = 0, frZ = 0;
                const bool framed = L <= 32;                                       // (a row of up to 64 bits: two words)
                auto top_move = [&](uint32_t A, uint32_t B) {                       // the packed words (philox.hpp, kSubTopPair)
                    if (A <= thrA1) {                                               // logical (xzzx_model.py:340-357)
                        const uint32_t op = (A >> 14) & 3u;
                        const uint32_t xp = ((op ^ (op >> 1)) & 1u) ? ((A & 0x3FFFu) * (uint32_t)L) >> 14 : 0u, zp = (op >> 1) ? scale_u16(B >> 16, L) : 0u;
                        const uint32_t ax = CODE == kCodeXzzx ? ((op ^ (op >> 1)) & 1u) : (op & 1u), az = op >> 1;
                        if (framed) {
                            frX ^= CODE == kCodeXzzx ? ax : ax << xp;
                            frZ ^= CODE == kCodeXzzx ? az : az << zp;
                        } else {
                            // the operator's L + L sites, generated on the fly (per-lane positions would make the plan's mask rows
                            // 2 W scattered global loads per proposal): xzzx -- X on the anti-diagonal, Z on the diagonal
                            // (xzzx_model.py:291-311); rotated -- X on column X_pos, Z on row Z_pos (rotated_surface_model.py:260-280);
                            // planar -- X on row X_pos, Z on column Z_pos of layer 0 (planar_model.py:264-268)
                            for (uint32_t i = 0; i < (uint32_t)L; ++i) {
                                uint32_t qx, qz;
                                if (CODE == kCodeXzzx) { qx = i * L + ((uint32_t)L - 1u - i); qz = i * L + i; }
                                else if (CODE == kCodeRotated) { qx = i * L + xp; qz = zp * L + i; }
                                else { qx = xp * L + i; qz = i * L + zp; }
                                lds_xor(stw + (qx >> 4) * 64, ax << ((qx & 15u) * 2u));
                                lds_xor(stw + (qz >> 4) * 64, (az * 3u) << ((qz & 15u) * 2u));
                            }
                        }
                        cdelta ^= ax | (az << 1);
                    } else if constexpr (kWideGen) {
                        const uint4 ev = gen_entry(scale_u32(B, a.n_gen));          // word B picks the generator; the expanded entry
                        const uint32_t e4[4] = {ev.x, ev.y, ev.z, ev.w};            // gives address and shift directly (a null site is 0)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            lds_xor(reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(stw) + (e4[i] >> 16)), shl_lo5((e4[i] >> 5) & 3u, e4[i]));
                    } else {
                        const uint2 e = gtab[scale_u32(B, a.n_gen)];                // word B picks the generator
                        const uint32_t ent[4] = {e.x & 0xFFFFu, e.x >> 16, e.y & 0xFFFFu, e.y >> 16};
#pragma unroll
                        for (int i = 0; i < 4; ++i) lds_xor(stw + (ent[i] >> 6) * 64, (ent[i] & 3u) << (((ent[i] >> 2) & 15u) * 2u));
                    }
                };
                top_blocks(top_move);
                uint32_t cnt_n = 0;
                if (framed && L > 16) {
                    // rows of 34 .. 64 bits (rotated L = 21 is BASELINE config 5's shape): the same stream with two-word patterns
                    auto spread16 = [](uint32_t c) {
                        c = (c | (c << 8)) & 0x00FF00FFu; c = (c | (c << 4)) & 0x0F0F0F0Fu;
                        c = (c | (c << 2)) & 0x33333333u; return (c | (c << 1)) & 0x55555555u;
                    };
                    const uint32_t hibits = rowbits - 32u;                          // 2 .. 32 bits of a row live in its second word
                    const uint32_t rmhi = hibits >= 32u ? 0xFFFFFFFFu : (1u << hibits) - 1u;
                    const uint32_t csrc = CODE == kCodeRotated ? frX : CODE == kCodePlanar ? frZ : 0u;
                    const uint32_t cmul = CODE == kCodePlanar ? 3u : 1u;
                    const uint32_t cplo = spread16(csrc & 0xFFFFu) * cmul, cphi = spread16(csrc >> 16) * cmul;
                    const uint32_t rplo = CODE == kCodeRotated ? 0xFFFFFFFFu : 0x55555555u, rphi = (CODE == kCodeRotated ? 0xFFFFFFFFu : 0x55555555u) & rmhi;
                    const int rowsel = (int)(CODE == kCodeRotated ? frZ : frX);
                    const uint32_t mX = (uint32_t)__builtin_amdgcn_sbfe((int)frX, 0u, 1u), mZ = (uint32_t)__builtin_amdgcn_sbfe((int)frZ, 0u, 1u);
                    uint32_t acc = 0, fill = 0;
                    uint32_t *wp = stw;
                    const int nrows = CODE == kCodePlanar ? 2 * L : L;
                    for (int r = 0; r < nrows; ++r) {
                        uint32_t plo, phi;
                        if (CODE == kCodeXzzx) {
                            const uint64_t cx = 1ull << (2 * (L - 1 - r)), cz = 3ull << (2 * r);        // wave-uniform
                            plo = (mX & (uint32_t)cx) ^ (mZ & (uint32_t)cz);
                            phi = (mX & (uint32_t)(cx >> 32)) ^ (mZ & (uint32_t)(cz >> 32));
                        } else if (CODE == kCodePlanar && r >= L) {
                            plo = phi = 0u;
                        } else {
                            const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe(rowsel, (uint32_t)r, 1u);
                            plo = __builtin_amdgcn_bitop3_b32(m, rplo, cplo, 0x6A);
                            phi = __builtin_amdgcn_bitop3_b32(m, rphi, cphi, 0x6A);
                        }
                        // the row's 96-bit window at bit `fill`: lo | mid | hi
                        const uint32_t lo = acc | (plo << fill);
                        const uint32_t mid = fill ? (plo >> (32u - fill)) | (phi << fill) : phi;
                        const uint32_t hi = fill ? phi >> (32u - fill) : 0u;
                        const uint32_t total = fill + rowbits;                      // 34 .. 95
                        cnt_n += nnz2(__hip_atomic_fetch_xor(wp, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ^ lo);
                        wp += 64;
                        if (total >= 64u) {
                            cnt_n += nnz2(__hip_atomic_fetch_xor(wp, mid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ^ mid);
                            wp += 64;
                            acc = hi; fill = total - 64u;
                        } else {
                            acc = mid; fill = total - 32u;
                        }
                    }
                    if (fill) cnt_n += nnz2(__hip_atomic_fetch_xor(wp, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ^ acc);
                } else
                if (framed) {
                    // flush the frame as one bit stream of 2L-bit rows (a returning ds_xor per word; the recount rides along)
                    auto spread16 = [](uint32_t c) {                                // bit i -> bit 2i
                        c = (c | (c << 8)) & 0x00FF00FFu; c = (c | (c << 4)) & 0x0F0F0F0Fu;
                        c = (c | (c << 2)) & 0x33333333u; return (c | (c << 1)) & 0x55555555u;
                    };
                    const uint32_t colpat = CODE == kCodeRotated ? spread16(frX) : CODE == kCodePlanar ? spread16(frZ) * 3u : 0u;
                    const uint32_t rowpat = CODE == kCodeRotated ? rowmask : 0x55555555u & rowmask;   // Z (11) / X (01) along a chosen row
                    const int rowsel = (int)(CODE == kCodeRotated ? frZ : frX);
                    const int mX = __builtin_amdgcn_sbfe((int)frX, 0u, 1u), mZ = __builtin_amdgcn_sbfe((int)frZ, 0u, 1u);   // xzzx parities as masks
                    uint32_t acc = 0, fill = 0, pend = 0;
                    uint32_t *wp = stw;
                    const int nrows = CODE == kCodePlanar ? 2 * L : L;
                    for (int r = 0; r < nrows; ++r) {
                        uint32_t pat;
                        if (CODE == kCodeXzzx)
                            pat = ((uint32_t)mX & (1u << (2 * (L - 1 - r)))) ^ ((uint32_t)mZ & (3u << (2 * r)));
                        else if (CODE == kCodePlanar && r >= L)
                            pat = 0u;                                               // layer 1 carries no logical operator
                        else
                            pat = __builtin_amdgcn_bitop3_b32((uint32_t)__builtin_amdgcn_sbfe(rowsel, (uint32_t)r, 1u), rowpat, colpat, 0x6A);
                        acc |= pat << fill;
                        if (fill + rowbits >= 32u) {
                            cnt_n += nnz2(pend);
                            pend = __hip_atomic_fetch_xor(wp, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ^ acc;
                            wp += 64;
                            const uint32_t rem = fill + rowbits - 32u;
                            acc = rem ? pat >> (rowbits - rem) : 0u;
                            fill = rem;
                        } else {
                            fill += rowbits;
                        }
                    }
                    cnt_n += nnz2(pend);
                    if (fill) cnt_n += nnz2(__hip_atomic_fetch_xor(wp, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ^ acc);
                } else {
                    for (int w = 0; w < W; ++w) cnt_n += nnz2(stw[w * 64]);
                }
                n = cnt_n;
                cls ^= cdelta;
            } else {
                // general path: top chains below p = 0.75 (logical proposals with the full Metropolis test, mcmc.py:20-35) and
                // Chain_xyz (mcmc.py:106-114,162-173)
                int nx = 0, ny = 0, nz = 0;
                for (int w = 0; w < W; ++w) count_xyz(stw[w * 64], nx, ny, nz);
                uint32_t cdelta = 0;
                const uint32_t *lmask = a.lmask;
                const int LW = (L + 1) * W;
                uint32_t gs = SCAN ? (uint32_t)(kbase % a.n_gen) : 0u;
                u32x4 pair{0, 0, 0, 0};                                            // the block four non-top proposals share
                uint64_t kb_pair = ~0ull;
                for (uint32_t j = 0; j < iters; ++j, gs = gs + 1 == a.n_gen ? 0u : gs + 1) {
                    const uint64_t k = kbase + j;
                    // non-top: word k&3 of block (k>>2, 1) (+ its refinement); top:
                    u32x4 x;
                    uint64_t v44 = 0;                                               // non-top: the 44-bit acceptance uniform
                    // (random scan: the top chain's packed words A, B of block (k >> 1, kSubTopPair), philox.hpp; sweep: block (k, 0))
                    const bool packed = top && !SCAN;
                    uint32_t pA = 0, pB = 0;
                    if (packed) {
                        x = philox_block(k >> 1, kSubTopPair, syn, strm, a.seed_lo, a.seed_hi);
                        pA = (k & 1) ? x.z : x.x; pB = (k & 1) ? x.w : x.y;
                    } else if (top) {
                        x = philox_block(k, 0, syn, strm, a.seed_lo, a.seed_hi);
                    } else {
                        if ((k >> 2) != kb_pair) {
                            kb_pair = k >> 2;
                            pair = philox_block(kb_pair, 1, syn, strm, a.seed_lo, a.seed_hi);
                        }
                        x.x = sel4(pair, (int)(k & 3));                             // the proposal's word: generator | 12 leading accept bits
                        v44 = (uint64_t)(x.x & 0xFFFu) << 32;                       // ... the low 32 bits are drawn only when they decide
                        x.y = x.z = x.w = 0;
                    }
                    const bool logical = packed ? pA <= thrA1 : (top && x.x <= thrL1);   // mcmc.py:23
                    int dx = 0, dy = 0, dz = 0;                                    // change of the X / Y / Z counts
                    uint32_t ent[4] = {0, 0, 0, 0}, cd = 0;
                    uint32_t *sad[4] = {stw, stw, stw, stw};                        // a stabilizer's sites: LDS word, bit shift (low 5 bits), Paulis
                    uint32_t ssh[4] = {0, 0, 0, 0}, sops = 0;
                    const uint32_t *m0 = lmask + L * W, *m1 = m0, *m2 = m0, *m3 = m0;   // identity rows
                    if (logical) {
                        if (CODE == kCodeToric) {
                            const uint32_t op0 = x.y >> 30, op1 = x.z >> 30;
                            const uint32_t dx0 = (op0 ^ (op0 >> 1)) & 1u, dz0 = op0 >> 1, dx1 = (op1 ^ (op1 >> 1)) & 1u, dz1 = op1 >> 1;
                            if (dx0) m0 = lmask + scale_low30(x.y, L) * W;
                            if (dz0) m1 = lmask + LW + scale_u16(x.w >> 16, L) * W;
                            if (dx1) m2 = lmask + 2 * LW + scale_low30(x.z, L) * W;
                            if (dz1) m3 = lmask + 3 * LW + scale_u16(x.w & 0xFFFFu, L) * W;
                            cd = Lodd ? (dx0 | (dz0 << 1) | (dx1 << 2) | (dz1 << 3)) : 0u;
                        } else {
                            const uint32_t op = packed ? (pA >> 14) & 3u : x.y >> 30;   // xzzx_model.py:346 / rotated_surface_model.py:334
                            const uint32_t xpos = packed ? ((pA & 0x3FFFu) * (uint32_t)L) >> 14 : scale_low30(x.y, L);
                            const uint32_t zpos = scale_u16((packed ? pB : x.w) >> 16, L);
                            const uint32_t xp = ((op ^ (op >> 1)) & 1u) ? xpos : 0u;   // drawn iff op in {1,2}
                            const uint32_t zp = (op >> 1) ? zpos : 0u;                 // drawn iff op in {3,2}
                            // applied operators: xzzx X iff op in {1,2}, Z iff op in {3,2}; rotated X iff op in {1,3}, Z iff op in {2,3}
                            const uint32_t ax = CODE == kCodeXzzx ? ((op ^ (op >> 1)) & 1u) : (op & 1u);
                            const uint32_t az = op >> 1;
                            if (ax) m0 = lmask + xp * W;                            // kind 0: X (xzzx: anti-diagonal at every pos)
                            if (az) m1 = lmask + LW + zp * W;                       // kind 1: Z
                            cd = ax | (az << 1);
                        }
                        for (int w = 0; w < W; ++w) {
                            const uint32_t old = stw[w * 64], neu = old ^ m0[w] ^ m1[w] ^ m2[w] ^ m3[w];
                            int ox = 0, oy = 0, oz = 0;
                            count_xyz(old, ox, oy, oz);
                            count_xyz(neu, dx, dy, dz);
                            dx -= ox; dy -= oy; dz -= oz;
                        }
                    } else {
                        const uint32_t wa = packed ? pB : top ? x.y : x.x;          // the generator word
                        if constexpr (SCAN) {
                            const uint2 e = gtab[gs];
                            ent[0] = e.x & 0xFFFFu; ent[1] = e.x >> 16; ent[2] = e.y & 0xFFFFu; ent[3] = e.y >> 16;
                        } else if (CODE == kCodeToric) {
                            uint32_t q[4];
                            const uint32_t g = top ? scale_u32(wa, 2u * (uint32_t)LL) : pick_top20(wa, 2u * (uint32_t)LL);
                            const uint32_t isX = g < (uint32_t)LL, rc = isX ? g : g - (uint32_t)LL;
                            toric_sites(L, LL, rc / (uint32_t)L, rc % (uint32_t)L, isX, q);
                            for (int i = 0; i < 4; ++i) ent[i] = (q[i] << 2) | (isX ? 1u : 3u);
                        } else if constexpr (!kWideGen) {
                            const uint2 e = gtab[top ? scale_u32(wa, a.n_gen) : pick_top20(wa, a.n_gen)];
                            ent[0] = e.x & 0xFFFFu; ent[1] = e.x >> 16; ent[2] = e.y & 0xFFFFu; ent[3] = e.y >> 16;
                        }
                        if constexpr (kWideGen && CODE != kCodeToric) {
                            // the expanded entry gives each site's LDS address with one add and its field with one bfe
                            const uint4 ev = gen_entry(top ? scale_u32(wa, a.n_gen) : pick_top20(wa, a.n_gen));
                            ssh[0] = ev.x; ssh[1] = ev.y; ssh[2] = ev.z; ssh[3] = ev.w;
                            sops = (ev.x >> 8) & 0xFFu;                             // the four Paulis as 2-bit fields (a null site: 0)
#pragma unroll
                            for (int i = 0; i < 4; ++i) sad[i] = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(stw) + (ssh[i] >> 16));
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const uint32_t q = ent[i] >> 2;
                                sad[i] = stw + (q >> 4) * 64;
                                ssh[i] = (q & 15u) * 2u;
                                sops |= (ent[i] & 3u) << (2 * i);
                            }
                        }
                        // old and new values of the (up to) four sites as 2-bit fields; an unused entry reads site 0 into
                        // both and cancels
                        uint32_t F = 0;
                        const uint32_t OPS = sops;
#pragma unroll
                        for (int i = 0; i < 4; ++i) F |= bfe2_lo5(*sad[i], ssh[i]) << (2 * i);
                        int ox = 0, oy = 0, oz = 0;
                        count_xyz(F, ox, oy, oz);
                        count_xyz(F ^ OPS, dx, dy, dz);
                        dx -= ox; dy -= oy; dz -= oz;
                    }
                    const int dE = dx + dy + dz;
                    bool acc;
                    // non-top: u = (v44 + w) 2^-44 with w the proposal's word of the refinement block, needed only when the 12
                    // leading bits do not decide (u lies in [v44, v44 + 2^32) 2^-44): drawn by the lanes that tie
                    auto refinement = [&]() { return (uint64_t)sel4(philox_block(k >> 2, kSubRefine, syn, strm, a.seed_lo, a.seed_hi), (int)(k & 3)); };
                    if (top) {
                        acc = acc_all || dE <= 0;                                   // mcmc.py:30
                        if (!acc) acc = philox_block(k, 2, syn, strm, a.seed_lo, a.seed_hi).x < a.acc_tbl_top[dE];   // :34
                    } else {
                        // mcmc.py:170 (Chain_xyz: a generator moves <= 4 sites) / mcmc.py:42 (a generator: dE <= 4)
                        const bool always = !xyz_rule && (acc_all || dE <= 0);
                        const uint64_t T = xyz_rule ? a.xyz_thr[((dx + 4) * 9 + (dy + 4)) * 9 + (dz + 4)]
                                                    : a.acc_thr44[slot_u][dE > 4 ? 3 : dE < 1 ? 0 : dE - 1];
                        acc = always || v44 + 0x100000000ull <= T;
                        if (!acc && v44 < T) acc = (v44 | refinement()) < T;
                    }
                    if (acc) {
                        if (logical) {
                            for (int w = 0; w < W; ++w) lds_xor(stw + w * 64, m0[w] ^ m1[w] ^ m2[w] ^ m3[w]);
                            cdelta ^= cd;
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i) lds_xor(sad[i], shl_lo5((sops >> (2 * i)) & 3u, ssh[i]));
                        }
                        nx += dx; ny += dy; nz += dz;
                    }
                }
                n = (uint32_t)(nx + ny + nz);
                cls ^= cdelta;
            }
            }   // !BIASED
        } else
        if (!top_logical) {
            [[maybe_unused]] int ni = (int)n;
            if constexpr (SCAN) {
                // systematic sweep: generator k mod G -- its sites are wave-uniform scalars from the plan's table --
                // and one Philox block per four proposals (word k&3 of block k>>2): walk the blocks that overlap
                // [kbase, kbase+iters) with a static word index
                const uint64_t kend = kbase + iters;
                uint32_t gs = (uint32_t)(kbase % a.n_gen);
                uint2 ev = gtab[gs];                                               // entry of the next proposal, fetched one ahead
                for (uint64_t kb = kbase >> 2; (kb << 2) < kend; ++kb) {
                    const u32x4 blk = philox_block(kb, 3, syn, strm, a.seed_lo, a.seed_hi);
                    const uint32_t xs[4] = {blk.x, blk.y, blk.z, blk.w};
#pragma unroll
                    for (int wi = 0; wi < 4; ++wi) {
                        const uint64_t k = (kb << 2) + wi;
                        if (k < kbase || k >= kend) continue;                       // uniform
                        const uint32_t e0 = __builtin_amdgcn_readfirstlane(ev.x), e1 = __builtin_amdgcn_readfirstlane(ev.y);
                        gs = gs + 1 == a.n_gen ? 0u : gs + 1;
                        ev = gtab[gs];
                        const uint32_t ent[4] = {e0 & 0xFFFFu, e0 >> 16, e1 & 0xFFFFu, e1 >> 16};
                        const uint32_t op = e0 & 3u;
                        uint32_t *ad[4];
                        uint32_t sh[4], wv[4], F = 0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            ad[i] = stw + (ent[i] >> 6) * 64;                      // scalar offsets
                            sh[i] = ((ent[i] >> 2) & 15u) * 2u;
                            wv[i] = *ad[i];                                        // all four reads in flight before the first use
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) F |= ((wv[i] >> sh[i]) & 3u) << (2 * i);
                        const uint32_t G = F ^ (op * 0x55u);
                        const int dE = __popc((G | (G >> 1)) & 0x55u) - __popc((F | (F >> 1)) & 0x55u);
                        if (xs[wi] <= myT[dE]) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) lds_xor(ad[i], op << sh[i]);
                            ni += dE;
                        }
                    }
                }
            }
            if constexpr (SCAN) n = (uint32_t)ni;
            else random_scan_loop();
        } else if (acc_all && L <= 16) {
            // Top chain at p = 0.75: every proposal is accepted (mcmc.py:30), so moves are blind
            // XORs and commute.  Stabilizers go straight to LDS; logical operators are collected
            // in a per-lane frame (which rows / columns carry an operator) and flushed once.
            //   fr0: bit r      = X on row r of layer 0      bit 16+c = Z on column c of layer 0
            //   fr1: bit c      = X on column c of layer 1   bit 16+r = Z on row r of layer 1
            uint32_t fr0 = 0, fr1 = 0, cdelta = 0;
            auto frame_logical = [&](uint32_t op0, uint32_t op1, uint32_t x0, uint32_t z0, uint32_t x1, uint32_t z1) {
                const uint32_t dx0 = (op0 ^ (op0 >> 1)) & 1u, dz0 = op0 >> 1;       // X iff op in {1,2}; Z iff op in {2,3}
                const uint32_t dx1 = (op1 ^ (op1 >> 1)) & 1u, dz1 = op1 >> 1;
                fr0 ^= (dx0 << x0) | (dz0 << (16 + z0));
                fr1 ^= (dx1 << x1) | (dz1 << (16 + z1));
                cdelta ^= dx0 | (dz0 << 1) | (dx1 << 2) | (dz1 << 3);
            };
            [[maybe_unused]] auto add_logical = [&](const u32x4 &x) {               // _apply_random_logical, toric_model.py:228-253 (sweep: block (k, 0))
                frame_logical(x.y >> 30, x.z >> 30, scale_low30(x.y, L), scale_u16(x.w >> 16, L), scale_low30(x.z, L), scale_u16(x.w & 0xFFFFu, L));
            };
            [[maybe_unused]] auto add_logical2 = [&](uint32_t A, uint32_t B) {       // random scan: the packed words (philox.hpp, kSubTopPair)
                frame_logical((A >> 14) & 3u, (A >> 12) & 3u, ((A & 0xFFFu) * (uint32_t)L) >> 12, ((B >> 21) * (uint32_t)L) >> 11,
                              (((B >> 10) & 0x7FFu) * (uint32_t)L) >> 11, ((B & 0x3FFu) * (uint32_t)L) >> 10);
            };
            if constexpr (SCAN) {
                // sweep at f = 1: generator k mod G with probability 1/2 (coin bit k&31 of word (k>>5)&3 of block
                // k>>7; a coin-less sweep composes to the identity) and one random logical every 8th proposal
                uint32_t gs = (uint32_t)(kbase % a.n_gen);
                u32x4 coins{0, 0, 0, 0};
                uint64_t cb_cur = ~0ull;
                for (uint32_t j = 0; j < iters; ++j) {
                    const uint64_t k = kbase + j;
                    if ((k & 7) == 0) add_logical(philox_block(k, 0, syn, strm, a.seed_lo, a.seed_hi));
                    if ((k >> 7) != cb_cur) { cb_cur = k >> 7; coins = philox_block(cb_cur, 3, syn, strm, a.seed_lo, a.seed_hi); }
                    const uint2 ev = gtab[gs];
                    const uint32_t e0 = __builtin_amdgcn_readfirstlane(ev.x), e1 = __builtin_amdgcn_readfirstlane(ev.y);
                    gs = gs + 1 == a.n_gen ? 0u : gs + 1;
                    if ((sel4(coins, (int)((k >> 5) & 3)) >> (k & 31)) & 1u) {
                        const uint32_t ent[4] = {e0 & 0xFFFFu, e0 >> 16, e1 & 0xFFFFu, e1 >> 16};
#pragma unroll
                        for (int i = 0; i < 4; ++i) lds_xor(stw + (ent[i] >> 6) * 64, (ent[i] & 3u) << (((ent[i] >> 2) & 15u) * 2u));
                    }
                }
            } else
            {
                auto blind = [&](uint32_t A, uint32_t B) {
                    if (A <= thrA1) {                                               // mcmc.py:23 (A[31:16] < thr16)
                        add_logical2(A, B);
                    } else {
                        const uint4 ev = gen_entry(scale_u32(B, 2u * (uint32_t)LL));     // word B picks the generator
                        const uint32_t e4[4] = {ev.x, ev.y, ev.z, ev.w};
                        const uint32_t op = (ev.x >> 5) & 3u;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            lds_xor(reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(stw) + (e4[i] >> 16)), shl_lo5(op, e4[i]));
                    }
                };
                top_blocks(blind);
            }
            // flush the frame: lattice row r of layer l is the 2L-bit span at bit 2*(l*LL + r*L)
            {
                uint32_t c0 = fr0 >> 16, c1 = fr1 & 0xFFFFu;       // column sets -> one 2-bit field per column
                c0 = (c0 | (c0 << 8)) & 0x00FF00FFu; c0 = (c0 | (c0 << 4)) & 0x0F0F0F0Fu;
                c0 = (c0 | (c0 << 2)) & 0x33333333u; c0 = (c0 | (c0 << 1)) & 0x55555555u;
                c1 = (c1 | (c1 << 8)) & 0x00FF00FFu; c1 = (c1 | (c1 << 4)) & 0x0F0F0F0Fu;
                c1 = (c1 | (c1 << 2)) & 0x33333333u; c1 = (c1 | (c1 << 1)) & 0x55555555u;
                const uint32_t colpat0 = c0 * 3u;                   // Z (11) on the chosen columns of layer 0
                const uint32_t colpat1 = c1;                        // X (01) on the chosen columns of layer 1
                // The 2L rows of the two layers are one bit stream (row rr at bit rr * rowbits): each row's pattern is appended
                // to a 32-bit accumulator and a word goes out (one ds_xor) whenever it fills.  A row's pattern is the layer's
                // column pattern, toggled by the row operator where the row is chosen: (mask & rowpattern) ^ colpattern in one
                // v_bitop3, the mask being the row's frame bit sign-extended (v_bfe_i32).
                const uint32_t rowX = 0x55555555u & rowmask;           // X (01) along a chosen row of layer 0
                // The xor returns the word it found, so the step's recount rides along: a finished word is counted while the
                // next one is being assembled.
                uint32_t acc = 0, fill = 0;                            // fill: bits of the accumulator in use (wave-uniform)
                uint32_t *wp = stw;
                uint32_t pend = 0, cnt_n = 0;                          // the last word written, not counted yet
#define QECMC_PUT_ROW(pat_expr)                                                              \
                {                                                                            \
                    const uint32_t pat = (pat_expr);                                         \
                    acc |= pat << fill;                                                      \
                    if (fill + rowbits >= 32u) {                                             \
                        cnt_n += nnz2(pend);                                                 \
                        pend = __hip_atomic_fetch_xor(wp, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ^ acc;   \
                        wp += 64;                                                            \
                        const uint32_t rem = fill + rowbits - 32u;                           \
                        acc = rem ? pat >> (rowbits - rem) : 0u;                             \
                        fill = rem;                                                          \
                    } else {                                                                 \
                        fill += rowbits;                                                     \
                    }                                                                        \
                }
                for (int r = 0; r < L; ++r)                            // layer 0: Z (11) on chosen columns, X (01) along chosen rows
                    QECMC_PUT_ROW(__builtin_amdgcn_bitop3_b32((uint32_t)__builtin_amdgcn_sbfe((int)fr0, (uint32_t)r, 1u), rowX, colpat0, 0x6A))
                for (int r = 0; r < L; ++r)                            // layer 1: X (01) on chosen columns, Z (11) along chosen rows
                    QECMC_PUT_ROW(__builtin_amdgcn_bitop3_b32((uint32_t)__builtin_amdgcn_sbfe((int)fr1, 16u + (uint32_t)r, 1u), rowmask, colpat1, 0x6A))
#undef QECMC_PUT_ROW
                cnt_n += nnz2(pend);
                if (fill) cnt_n += nnz2(__hip_atomic_fetch_xor(wp, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ^ acc);
                n = cnt_n;
            }
            if (Lodd) cls ^= cdelta;
        } else if (SCAN && acc_all) {
            if constexpr (SCAN && GENTOP) blind_sweep_tables();                     // toric L > 16 at f = 1
        } else if constexpr (GENTOP) {
            // general top chain (L > 16, or a 1-chain ladder whose top sits below p = 0.75):
            // table-driven logical operators and the full Metropolis test, mcmc.py:20-35
            int ni = (int)n;
            uint32_t cdelta = 0;
            const uint32_t *lmask = a.lmask;
            for (uint32_t j = 0; j < iters; ++j) {
                const uint64_t k = kbase + j;
                // random scan: the packed words A, B of block (k >> 1, kSubTopPair); sweep: block (k, 0)
                const u32x4 x = SCAN ? philox_block(k, 0, syn, strm, a.seed_lo, a.seed_hi)
                                     : philox_block(k >> 1, kSubTopPair, syn, strm, a.seed_lo, a.seed_hi);
                [[maybe_unused]] const uint32_t pA = (k & 1) ? x.z : x.x, pB = (k & 1) ? x.w : x.y;
                if (SCAN ? x.x <= thrL1 : pA <= thrA1) {
                    const uint32_t op0 = SCAN ? x.y >> 30 : (pA >> 14) & 3u, op1 = SCAN ? x.z >> 30 : (pA >> 12) & 3u;
                    const uint32_t dx0 = (op0 ^ (op0 >> 1)) & 1u, dz0 = op0 >> 1, dx1 = (op1 ^ (op1 >> 1)) & 1u, dz1 = op1 >> 1;
                    const uint32_t px0 = SCAN ? scale_low30(x.y, L) : ((pA & 0xFFFu) * (uint32_t)L) >> 12;
                    const uint32_t pz0 = SCAN ? scale_u16(x.w >> 16, L) : ((pB >> 21) * (uint32_t)L) >> 11;
                    const uint32_t px1 = SCAN ? scale_low30(x.z, L) : (((pB >> 10) & 0x7FFu) * (uint32_t)L) >> 11;
                    const uint32_t pz1 = SCAN ? scale_u16(x.w & 0xFFFFu, L) : ((pB & 0x3FFu) * (uint32_t)L) >> 10;
                    const uint32_t ix0 = dx0 ? px0 : L, iz0 = dz0 ? pz0 : L;       // row L = identity
                    const uint32_t ix1 = dx1 ? px1 : L, iz1 = dz1 ? pz1 : L;
                    const int LW = (L + 1) * W;
                    const uint32_t *m0 = lmask + ix0 * W, *m1 = lmask + LW + iz0 * W, *m2 = lmask + 2 * LW + ix1 * W,
                                   *m3 = lmask + 3 * LW + iz1 * W;
                    int dE = 0;
                    for (int w = 0; w < W; ++w) {
                        const uint32_t old = stw[w * 64];
                        dE += (int)nnz2(old ^ m0[w] ^ m1[w] ^ m2[w] ^ m3[w]) - (int)nnz2(old);
                    }
                    bool acc = true;
                    if (!acc_all && dE > 0) acc = philox_block(k, 2, syn, strm, a.seed_lo, a.seed_hi).x < a.acc_tbl_top[dE];
                    if (acc) {
                        for (int w = 0; w < W; ++w) lds_xor(stw + w * 64, m0[w] ^ m1[w] ^ m2[w] ^ m3[w]);
                        ni += dE;
                        cdelta ^= dx0 | (dz0 << 1) | (dx1 << 2) | (dz1 << 3);
                    }
                } else {
                    uint32_t q[4], op;
                    if constexpr (SCAN) {                                          // generator k mod G from the plan's table
                        const uint2 e = gtab[(uint32_t)(k % a.n_gen)];
                        q[0] = (e.x & 0xFFFFu) >> 2; q[1] = e.x >> 18; q[2] = (e.y & 0xFFFFu) >> 2; q[3] = e.y >> 18;
                        op = e.x & 3u;
                    } else {
                        const uint32_t g = scale_u32(pB, 2u * (uint32_t)LL), isX = g < (uint32_t)LL, rc = isX ? g : g - (uint32_t)LL;
                        toric_sites(L, LL, rc / (uint32_t)L, rc % (uint32_t)L, isX, q);
                        op = isX ? 1u : 3u;
                    }
                    int dE = 0;
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t f = (stw[(q[i] >> 4) * 64] >> ((q[i] & 15u) * 2u)) & 3u;
                        dE += (int)(f == 0u) - (int)(f == op);
                    }
                    bool acc = true;
                    if (!acc_all && dE > 0) acc = philox_block(k, 2, syn, strm, a.seed_lo, a.seed_hi).x < a.acc_tbl_top[dE];
                    if (acc) {
                        for (int i = 0; i < 4; ++i) lds_xor(stw + (q[i] >> 4) * 64, op << ((q[i] & 15u) * 2u));
                        ni += dE;
                    }
                }
            }
            n = (uint32_t)ni;
            if (Lodd) cls ^= cdelta;
        }

        // ---------------- swap sweep, Ladder.step mcmc.py:96-103 --------------------------------
        // (double-buffered by step parity: a fast wave may publish step t+1 while a slow one still reads step t)
        QECMC_STAMP(1);
        uint32_t *cur = info + (t & 1) * NC * 64 + lane_t, *sx = swx + (t & 1) * NC * 64 + lane_t;
        cur[slot_u * 64] = pack_info(n, sid, cls, flag);
        if constexpr (PRE) {
            // the wave on slot 1 becomes the top chain two steps from now, the wave on slot 0 next step: each draws half of that
            // step's top-chain blocks (state-independent: the packed blocks of the top slot's stream) while the others finish
            if (NC >= 3 && slot_u <= 1u && a.thr_logical != 0) {
                const uint64_t kb1 = a.prop0 + (t + 1 + slot_u) * iters;
                {
                    // the packed blocks (two proposals each) of that step, the first kPre of them
                    const uint32_t nb = (uint32_t)(((kb1 + iters - 1) >> 1) - (kb1 >> 1)) + 1u, pn = nb < (uint32_t)kPre ? nb : (uint32_t)kPre, ph = pn / 2;
#pragma unroll
                    for (int jj = 0; jj < kPre; ++jj) {
                        const bool mine = slot_u == 1u ? (uint32_t)jj < ph : ((uint32_t)jj >= ph && (uint32_t)jj < pn);
                        if (mine) pre[jj] = philox_block((kb1 >> 1) + jj, kSubTopPair, syn, (uint32_t)(NC - 1), a.seed_lo, a.seed_hi);
                    }
                }
            }
        }
        const int swb = NC - 2 - (int)slot_u;                     // Philox block of swap uniforms this wave draws (if any)
        if (swb >= 0 && swb < 4 && swb * 4 < NC - 1) {
            // the sweep's uniforms do not depend on the state: the slots just below the top (never the
            // heavier top slot itself) draw one Philox block each
            const u32x4 b = philox_block(a.step0 + t - t0, (uint32_t)swb, syn, kSwapStream, a.seed_lo, a.seed_hi);
            uint32_t *p = sx + swb * 4 * 64;                       // rows 4b .. 4b+3 = rung pairs, NC-1 of them exist
            const int left = NC - 1 - swb * 4;
            // The swap test u < p_diff[i]^d (mcmc.py:149) does not need the records: thresholds fall with d, so it reads
            // d <= dmax with dmax = the largest d whose threshold exceeds x.  dmax is found here, off the cascade's serial
            // path: a log2 guess, then the exact table moves it up or down (so rounding in the guess cannot matter).
            auto swap_dmax = [&](uint32_t x, int i) -> uint32_t {
                if (SSW || (BIASED && alpha_noise)) return x;       // SSW: wave 0 tests the raw uniform; Ladder_alpha compares in floating point (below)
                auto below = [&](int dd) -> bool {                  // x < ceil(p_diff[i]^dd * 2^32), dd in [1, nq]
                    return (swap_fast && dd < kSwapFast) ? x < swapT[i * kSwapFast + dd]
                                                         : (uint64_t)x < a.swap_thr[(size_t)i * (nq + 1) + dd];
                };
                const float inv = a.swap_inv_log2[i];               // 0: p_diff[i] >= 1 (coinciding rungs), every d passes
                int d = inv == 0.0f ? nq : (int)((__log2f((float)x + 0.5f) - 32.0f) * inv);
                d = d < 0 ? 0 : d > nq ? nq : d;
                if (swap_fast && d + 2 < kSwapFast) {
                    // the guess is within one of the answer: look at the four thresholds around it at once (independent LDS
                    // reads) instead of walking the table; entries past nq are 0 and never pass
                    const int w0 = d > 1 ? d - 1 : 1;
                    const uint32_t *T = swapT + i * kSwapFast + w0;
                    const int c = (int)(x < T[0]) + (int)(x < T[1]) + (int)(x < T[2]) + (int)(x < T[3]);   // a prefix passes
                    d = w0 - 1 + c;
                    if (c < 4 && (c > 0 || w0 == 1)) return (uint32_t)d;   // the answer lies inside the window
                }
                while (d < nq && below(d + 1)) ++d;
                while (d > 0 && !below(d)) --d;
                return (uint32_t)d;
            };
            // (results first, stores after: the four look-ups are independent and overlap)
            const uint32_t r0 = swap_dmax(b.x, swb * 4);
            const uint32_t r1 = left > 1 ? swap_dmax(b.y, swb * 4 + 1) : 0u;
            const uint32_t r2 = left > 2 ? swap_dmax(b.z, swb * 4 + 2) : 0u;
            const uint32_t r3 = left > 3 ? swap_dmax(b.w, swb * 4 + 3) : 0u;
            p[0] = r0;
            if (left > 1) p[64] = r1;
            if (left > 2) p[128] = r2;
            if (left > 3) p[192] = r3;
        }
        QECMC_STAMP(2);
        __syncthreads();
        QECMC_STAMP(3);
        [[maybe_unused]] bool q_refill = false;
        if (CONV || USET) {                                 // flags set one step earlier: uniform for the workgroup
            volatile uint32_t *f0 = lds_all + (NC * W * 64 + 4 * NC * 64 + ncls * 64 + NC * 9 + NC * kSwapFast);
            if (f0[t & 1]) break;
            if constexpr (QUEUE) q_refill = f0[2 + (t & 1)] != 0;   // (written by wave 0 before this step's barrier: double-buffered by parity)
        }
        {
            // every wave replays the top-down cascade on the published records; `car` is the record
            // being carried down, `mine` the one that ends in the slot this wave takes over next
            slot_u = slot_u == 0 ? (uint32_t)(NC - 1) : slot_u - 1;   // downwards: the wave leaving the top slot needs one rung only
            // _r_flip (mcmc.py:146-149) of rung pair i for the carried record and the one below: d <= 0, or u < rel_p**d
            auto swap_flip = [&](int i, uint32_t hi, uint32_t lo, uint32_t xi) -> bool {
                const int d = (int)(hi & 0xFFFFu) - (int)(lo & 0xFFFFu);           // ne_hi - ne_lo
                if (BIASED && alpha_noise) {
                    // Ladder_alpha.r_flip, mcmc_alpha.py:118-123: slot-bound n_eff, always draws
                    const uint32_t *ne = neffb + (t & 1) * NC * 64 + lane_t;
                    const int ic = i < 0 ? 0 : i;
                    return alpha_flip(xi, ne[(ic + 1) * 64], ne[ic * 64], a.alpha, a.alpha_lnb[ic]);
                }
                if constexpr (SSW) {
                    // xi is the raw uniform: x < ceil(p_diff[i]^d 2^32) from the LDS table (d < 64; entry 0 never passes), else HBM
                    const int ic = i < 0 ? 0 : i, dd = d < 1 ? 1 : d;
                    bool lt = xi < swapT[ic * kSwapFast + (dd < kSwapFast ? dd : 0)];
                    if (!swap_fast || dd >= kSwapFast) lt = (uint64_t)xi < a.swap_thr[(size_t)ic * (nq + 1) + dd];
                    return d <= 0 || lt;
                } else {
                    return d <= (int)xi;                                            // xi = the largest accepted difference (thresholds fall with d)
                }
            };
            uint32_t car = cur[(NC - 1) * 64], mine = car;
            if constexpr (SSW) {
                uint32_t *nxt = info + ((t + 1) & 1) * NC * 64 + lane_t;             // the other parity's records: idle until the waves publish step t+1
                if (wave_u == 0) {
                    // the other seven waves wait for this sweep: it runs at the highest issue priority.  Afterwards wave 0 drops to the lowest
                    // until the next step sets the workgroup's level again on the toric / planar codes (its bookkeeping then yields to
                    // the other waves' proposals), and returns to that level at once on the xzzx / rotated codes.  Same-box A/B: toric L = 9
                    // +1.7 %, L = 5 +1.1 %, planar L = 9 +0.8 %, xzzx / rotated L = 9 +0.8 % (and -2.4 ... -3 % with the toric variant).
                    __builtin_amdgcn_s_setprio(3);
                    for (int i = NC - 2; i >= 0; --i) {                             // mcmc.py:96
                        const uint32_t lo = cur[i * 64];
                        const bool flip = swap_flip(i, car, lo, sx[i * 64]);
                        nxt[(i + 1) * 64] = flip ? lo : car;                        // what slot i+1 now holds (:98-99)
                        car = flip ? car : lo;
                    }
                    nxt[0] = car;
                    if constexpr (CODE == kCodeToric || CODE == kCodePlanar) {
                        __builtin_amdgcn_s_setprio(0);
                    } else {
                        switch (3u - (uint32_t)((t >> 3) & 3)) {
                            case 0: __builtin_amdgcn_s_setprio(0); break;
                            case 1: __builtin_amdgcn_s_setprio(1); break;
                            case 2: __builtin_amdgcn_s_setprio(2); break;
                            default: __builtin_amdgcn_s_setprio(3); break;
                        }
                    }
                }
                __syncthreads();
                mine = nxt[slot_u * 64];
            } else {
            // a wave only needs the cascade down to the rung that fills its own next slot (wave 0 also does the
            // slot-0 bookkeeping and runs it to the bottom).  (Compiling the cascade twice -- wave 0 capturing its record on the way,
            // the others taking what their last rung leaves -- saves a v_cndmask per rung and measured 2 % slower.)
            const int i_stop = (wave_u == 0 || slot_u == 0) ? 0 : (int)slot_u - 1;
            // one rung at a time.  (Fetching four rungs' records and bounds together took the LDS round trips off the serial path when
            // every wave ran the whole cascade; a wave now stops at its own rung -- 4.4 rungs on average at 8 temperatures, one for the
            // wave that moves to the top -- and the eight look-ups per group cost more than they hid: one at a time is +1.8 % at
            // toric L = 15, +2.8 % at rotated L = 21, +3.5 % at L = 10, same-box A/B against groups of 2, 3, 4 and 8.)
            for (int i = NC - 2; i >= i_stop; --i) {                                // mcmc.py:96
                const uint32_t lo = cur[i * 64], xi = sx[i * 64];
                const bool flip = swap_flip(i, car, lo, xi);
                const uint32_t into = flip ? lo : car;                              // what slot i+1 now holds (:98-99)
                car = flip ? car : lo;
                if ((int)slot_u == i + 1) mine = into;
            }
            }   // !SSW
            if (!SSW && slot_u == 0) mine = car;
            n = mine & 0xFFFFu; sid = (mine >> 16) & 0xFFu; cls = (mine >> 24) & 0x3Fu; flag = mine >> 31;
            QECMC_STAMP(4);
            if ((int)slot_u == NC - 1) flag = 1;                                    // chains[-1].flag = 1, mcmc.py:100
            if (wave_u == 0 && !done) {                                             // ladder + PTEQ bookkeeping on slot 0's new state
                if constexpr (kLdsCounters) { tops0 = ctrT[lane_t]; samples = ctrS[lane_t]; }
                tops0 += (NC == 1) | (car >> 31);                                   // chains[0].flag == 1, :101-102
                const uint32_t n0 = car & 0xFFFFu;
                if (a.counts != nullptr && tops0 >= a.tops_burn) {                  // decoders.py:60-67
                    const uint32_t v = (car >> 24) & 0x3Fu;
                    hist[(CODE == kCodeXzzx ? (v ^ (v >> 1)) : v) * 64 + lane_t] += 1;
                    samples++;
                    if (CONV && BIASED && alpha_noise) {
                        // nbr_errors_bottom_chain[since_burn] = chains[0].n_eff (decoders_biasednoise.py:204): slot 0's
                        // attribute, logged as its two counts; the window sums stay exact integers
                        if (QUEUE ? !q_dead : lane_t < cnt) {
                            // (QUEUE: one log column per lane of the persistent grid, rows = the ladder's own steps)
                            const size_t lN = QUEUE ? (size_t)gridDim.x * 64u : (size_t)a.N;
                            uint32_t *mylog = reinterpret_cast<uint32_t *>(a.nlog) + (s0 + lane_t);
                            const uint32_t v0 = neffb[(t & 1) * NC * 64 + lane_t];
                            mylog[(size_t)(t - t0) * lN] = v0;
                            const uint32_t l = samples, lo1 = l - 1;
                            const uint32_t a0 = lo1 >> 2, b0 = lo1 >> 1, c0 = (3u * lo1) >> 2, a1 = l >> 2, b1 = l >> 1, c1 = (3u * l) >> 2;
                            sumB += v0 & 0xFFFFu; sumBxy += v0 >> 16;
                            if (c1 != c0) { const uint32_t v = mylog[(size_t)(burn + c0) * lN]; sumB -= v & 0xFFFFu; sumBxy -= v >> 16; }
                            if (b1 != b0) { const uint32_t v = mylog[(size_t)(burn + b0) * lN]; sumA += v & 0xFFFFu; sumAxy += v >> 16; }
                            if (a1 != a0) { const uint32_t v = mylog[(size_t)(burn + a0) * lN]; sumA -= v & 0xFFFFu; sumAxy -= v >> 16; }
                        }
                    } else
                    if (CONV && (QUEUE ? !q_dead : lane_t < cnt)) {
                        // nbr_errors_bottom_chain[since_burn] = count_errors (:68); series index i lives in log row burn+i
                        // (QUEUE: one log column per lane of the persistent grid, rows = the ladder's own steps)
                        const size_t lN = QUEUE ? (size_t)gridDim.x * 64u : (size_t)a.N;
                        uint16_t *mylog = a.nlog + (s0 + lane_t);
                        mylog[(size_t)(t - t0) * lN] = (uint16_t)n0;
                        const uint32_t l = samples, lo1 = l - 1;
                        const uint32_t a0 = lo1 >> 2, b0 = lo1 >> 1, c0 = (3u * lo1) >> 2, a1 = l >> 2, b1 = l >> 1, c1 = (3u * l) >> 2;
                        sumB += n0;
                        if (c1 != c0) sumB -= mylog[(size_t)(burn + c0) * lN];
                        if (b1 != b0) sumA += mylog[(size_t)(burn + b0) * lN];
                        if (a1 != a0) sumA -= mylog[(size_t)(burn + a0) * lN];
                    }
                } else {
                    burn++;                                                         // resulting_burn_in, :71
                }
                if (CONV && tops0 >= a.TOPS) {                               // :74
                    const uint32_t l = samples ? samples : 1u;
                    const uint32_t den2 = (l >> 1) - (l >> 2), den4 = l - ((3u * l) >> 2);
                    bool accept = false;                                            // empty slice -> nan -> not accepted
                    if (samples && den2 && den4) {
                        if (BIASED && alpha_noise)
                            accept = alpha_series_close(sumA, sumAxy, den2, sumB, sumBxy, den4, a.alpha, a.eps);
                        else
                            accept = fabs((double)sumA / (double)den2 - (double)sumB / (double)den4) < a.eps;   // :96-102
                    }
                    if (accept) {
                        if (conv_streak >= a.SEQ) { done = 1; conv_ok = 1; steps_done = (uint32_t)(t - t0) + 1; }   // :77-78
                        else conv_streak = tops0 - conv_start;                      // :79
                    } else {
                        conv_streak = 0;                                            // :81-82
                        conv_start = tops0;
                    }
                }
                if constexpr (kLdsCounters) { ctrT[lane_t] = tops0; ctrS[lane_t] = samples; tops0 = 0; samples = 0; }
            }
            if (a.swap_acc != nullptr && wave_u == 0) {
                // equilibrium observables (qecmc_plan_set_stats): the cascade once more, with every rung's decision and the error
                // count each rung ends the step with added to per-lane LDS counters (off the hot path: one scalar branch when off)
                uint32_t *sacc = lds_all + gdw + lane_t, *nsum = sacc + NC * 64;
                uint32_t c2 = cur[(NC - 1) * 64];
                for (int i = NC - 2; i >= 0; --i) {
                    const uint32_t lo = cur[i * 64], xi = sx[i * 64];
                    const bool flip = swap_flip(i, c2, lo, xi);
                    if (!done) {                                                   // (a converged syndrome stops counting)
                        sacc[i * 64] += flip;
                        nsum[(i + 1) * 64] += (flip ? lo : c2) & 0xFFFFu;
                    }
                    c2 = flip ? c2 : lo;
                }
                if (!done) nsum[0] += c2 & 0xFFFFu;
            }
            if constexpr (QUEUE) {
                if (wave_u == 0) {
                    // a ladder ends by the criterion or at the horizon of `nsteps` of its own steps; its results go out at once
                    if (!q_dead && !done && (t - t0) + 1 >= a.nsteps) { done = 1; steps_done = (uint32_t)a.nsteps; }
                    if (!q_dead && done && !q_flushed) {
                        const uint64_t row = (uint64_t)qi / R;
                        for (int c = 0; c < ncls; ++c) {
                            const uint32_t v = hist[c * 64 + lane_t];
                            hist[c * 64 + lane_t] = 0;
                            if (R > 1) { if (v) atomicAdd(a.counts + row * ncls + c, v); }
                            else a.counts[row * ncls + c] = v;
                        }
                        if (R > 1) {
                            atomicAdd(a.samples + row, samples);
                            if (a.tops0 != nullptr) atomicAdd(a.tops0 + row, tops0);
                            if (a.steps_done != nullptr) atomicMax(a.steps_done + row, steps_done);
                            if (a.converged != nullptr && !conv_ok) a.converged[row] = 0;
                        } else {
                            a.samples[row] = samples;
                            if (a.tops0 != nullptr) a.tops0[row] = tops0;
                            if (a.steps_done != nullptr) a.steps_done[row] = steps_done;
                            if (a.converged != nullptr) a.converged[row] = (uint8_t)conv_ok;
                        }
                        q_flushed = true;
                        if (q_empty) q_dead = true;
                    }
                    volatile uint32_t *f0 = lds_all + (NC * W * 64 + 4 * NC * 64 + ncls * 64 + NC * 9 + NC * kSwapFast);
                    if (__all(q_dead)) f0[(t + 1) & 1] = 1;
                    // ask for a refill at the end of the next step if lanes wait for work and that step's successor keeps the blocks in phase
                    f0[2 + ((t + 1) & 1)] = (!q_empty && __any(done && !q_dead) && ((t + 2) % q_period) == 0) ? 1u : 0u;
                }
            } else
            if (CONV && wave_u == 0 && __all(done || lane_t >= cnt)) stopf[(t + 1) & 1] = 1;
            if (slot_u == 0) flag = 0;                                              // :103
            if constexpr (USET) {
                const bool cm = a.uset_conv_mult != 0.0;
                if (cm && t > 0 && !cm_done) {
                    // the stop test that ends step t-1 (decoders.py:159-162, :261-262, :825-826), now that every rung's
                    // insertion of that step is behind this step's barrier
                    const uint32_t tp = (uint32_t)t - 1u;
                    if (cm_trig[(tp & 1u) * 64 + lane_t] == tp) cm_last = tp;         // stop = step * conv_mult, :156
                    if ((double)tp >= (double)cm_last * a.uset_conv_mult && (uint64_t)tp * 100u >= a.nsteps) { cm_done = 1; cm_steps = (uint32_t)t; }
                }
                // PTDC_droplet / PTRC_droplet (decoders.py:146-152, :596-618): the configuration now in this wave's rung goes into
                // the set of chains seen so far.  Key = FNV-1a over the packed words (any collision-free key gives the same N(n)).
                if (lane_t < cnt && !cm_done) {
                    const uint32_t *sw = st + sid * W * 64 + lane_t;
                    uint64_t h = 0xCBF29CE484222325ull;
                    for (int w = 0; w < W; ++w) h = (h ^ sw[w * 64]) * 0x100000001B3ull;
                    h ^= h >> 32;
                    const unsigned long long key = h ? h : 1ull;
                    const uint64_t ladder = s0 + (uint64_t)lane_t;
                    const uint64_t set = a.uset_per_rung ? ladder * (uint64_t)NC + slot_u : ladder / a.uset_D;
                    if (a.uset_mhist != nullptr) atomicAdd(a.uset_mhist + set * (uint64_t)(nq + 1) + n, 1u);
                    unsigned long long *tb = a.uset_tab + set * a.uset_cap;
                    uint64_t idx = ((key * 0x9E3779B97F4A7C15ull) >> 20) & (a.uset_cap - 1);
                    bool fresh = false;
                    for (uint64_t probes = 0; probes < a.uset_cap; ++probes) {     // the table holds twice the insertions it can see
                        const unsigned long long old = atomicCAS(tb + idx, 0ull, key);
                        if (old == 0ull) { atomicAdd(a.uset_hist + set * (uint64_t)(nq + 1) + n, 1u); fresh = true; break; }
                        if (old == key) break;
                        idx = (idx + 1) & (a.uset_cap - 1);
                    }
                    if (fresh && a.uset_xyz != nullptr) {
                        int cx = 0, cy = 0, cz = 0;
                        for (int w = 0; w < W; ++w) count_xyz(sw[w * 64], cx, cy, cz);
                        const uint32_t pos = atomicAdd(a.uset_xyz_cnt + set, 1u);
                        if (pos < a.uset_xyz_stride) a.uset_xyz[set * a.uset_xyz_stride + pos] = (uint32_t)cx | ((uint32_t)cy << 10) | ((uint32_t)cz << 20);
                    }
                    if (cm) {
                        if (a.uset_own != nullptr) {
                            // the stop looks at the droplet's own dictionary (one process per droplet in the reference, :213-219),
                            // the class set above is the union over the droplets (:220-226)
                            unsigned long long *ob = a.uset_own + ladder * a.uset_own_cap;
                            uint64_t oi = ((key * 0x9E3779B97F4A7C15ull) >> 20) & (a.uset_own_cap - 1);
                            fresh = false;
                            for (uint64_t probes = 0; probes < a.uset_own_cap; ++probes) {
                                const unsigned long long old = atomicCAS(ob + oi, 0ull, key);
                                if (old == 0ull) { fresh = true; break; }
                                if (old == key) break;
                                oi = (oi + 1) & (a.uset_own_cap - 1);
                            }
                        }
                        // "if conv_mult and length <= shortest" (:153-156) on the rungs in any order: the step extends the run
                        // iff its shortest new chain is no longer than the shortest seen before, and that rung always passes
                        if (fresh && n <= atomicMin(&cm_short[lane_t], n)) cm_trig[((uint32_t)t & 1u) * 64 + lane_t] = (uint32_t)t;
                    }
                }
                if (cm && wave_u == 0 && __all(cm_done || lane_t >= cnt)) stopf[(t + 1) & 1] = 1;
            }
            if constexpr (QUEUE) {
                if (q_refill) {                                                     // uniform for the workgroup (read behind the barrier)
                    carry_kb = ~0ull;                                               // (a new ladder starts on a block boundary)
                    if (wave_u == 0) {
                        // the finished lanes take the next ladders of the batch: one atomic per wave, ranks by prefix count
                        const bool want = done && !q_dead;
                        const uint64_t m = __ballot(want);
                        const uint32_t nw = (uint32_t)__popcll(m);
                        uint32_t base = 0;
                        if (lane_t == 0 && nw) base = atomicAdd(a.queue, nw);
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + gridDim.x * 64u;   // the first grid x 64 ladders were handed out at launch
                        const uint32_t mine = base + (uint32_t)__popcll(m & ((1ull << lane_t) - 1ull));
                        uint32_t nqi = qi, nt0 = t0;
                        if (want) {
                            if ((uint64_t)mine < a.N) { nqi = mine; nt0 = (uint32_t)t + 1u; }
                            else { nqi = 0xFFFFFFFFu; q_dead = true; }
                        }
                        qidx[lane_t] = nqi;
                        qt0[lane_t] = nt0;
                        if ((uint64_t)base + nw >= a.N) q_empty = true;             // (later finishers retire when they write their results)
                    }
                    __syncthreads();
                    // every wave adopts the assignment and stages its next slot's chain for the lanes that changed ladder
                    const uint32_t nqi = qidx[lane_t], nt0 = qt0[lane_t];
                    if (nqi != 0xFFFFFFFFu && (nqi != qi || nt0 != t0)) {
                        qi = nqi; t0 = nt0; kq = (uint64_t)t0 * iters;
                        syn = a.first_syndrome + qi;
                        dstrm = kDiagStream + slot_u;                                // its step 0: the slot this wave has just moved to
                        const uint8_t *src = a.init + (uint64_t)(qi / R) * (uint64_t)nq;
                        const int dbase = (int)slot_u * W * 64 + lane_t;               // Ladder.__init__: every slot starts from the seed (mcmc.py:72)
                        uint32_t cn = 0;
                        for (int w = 0; w < W; ++w) {
                            uint32_t word = 0;
                            for (int b = 0; b < 16; ++b) {
                                const int q = w * 16 + b;
                                if (q < nq) word |= (uint32_t)(src[q] & 3u) << (2 * b);
                            }
                            st[dbase + w * 64] = word;
                            cn += nnz2(word);
                        }
                        sid = slot_u; n = cn; flag = slot_u == (uint32_t)(NC - 1);
                        if constexpr (CODE == kCodeToric) cls = toric_class_packed(st + dbase, W, LL);
                        else cls = surf_class_packed(CODE, st + dbase, L);
                        if constexpr (BIASED) {
                            // the packed counts of the fresh state and, for alpha noise, the slot's n_eff record of both parities
                            // (Chain_alpha.__init__, mcmc_alpha.py:18-22)
                            int nx = 0, ny = 0, nz = 0;
                            for (int w = 0; w < W; ++w) count_xyz(st[dbase + w * 64], nx, ny, nz);
                            xyc[sid * 64 + lane_t] = (uint32_t)nx | ((uint32_t)nz << 10) | ((uint32_t)(nx + ny) << 20);
                            if (alpha_noise) {
                                const uint32_t v = (uint32_t)nz | ((uint32_t)(nx + ny) << 16);
                                neffb[slot_u * 64 + lane_t] = v;
                                neffb[(NC + slot_u) * 64 + lane_t] = v;
                            }
                        }
                        if (wave_u == 0) {
                            tops0 = 0; samples = 0; burn = 0; conv_start = 0; conv_streak = 0; done = 0; steps_done = 0; conv_ok = 0;
                            sumA = 0; sumB = 0; sumAxy = 0; sumBxy = 0; q_flushed = false;
                        }
                    }
                }
            }
        }
    }
#if defined(QECMC_TIMELINE) || defined(QECMC_STEPTRACE)
    if (a.dbg && threadIdx.x == 0) a.dbg[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime();
#endif
    // (the epilogue's addresses are formed here, not carried through the step loop)
    const int tid_e0 = tid;
    const int slot_e = kLaunderLane ? (int)wave_u : slot;        // (the wave's index, from the scalar copy)
    int lane_e = lane;
    if constexpr (kLaunderLane) { lane_e = (int)__lane_id(); asm volatile("" : "+v"(lane_e)); }
    const int tid_e = kLaunderLane ? (int)(wave_u * 64u) + lane_e : tid_e0;
    uint32_t *fin = info + (a.nsteps & 1) * NC * 64;
    fin[slot_u * 64 + lane_e] = pack_info(n, sid, cls, flag);
    __syncthreads();
    if constexpr (kLdsCounters) {
        if (slot_e == 0) { tops0 = ctrT[lane_e]; samples = ctrS[lane_e]; }
    }

    // ---- results: coalesced stores ---------------------------------------------------
    if constexpr (QUEUE) return;                  // (every ladder wrote its results when it finished)
    if (a.counts != nullptr)
#pragma unroll 1
        for (int i = tid_e; i < cnt * ncls; i += nthreads) {
            const int j = i / ncls, c = i - j * ncls;
            const uint32_t v = hist[c * 64 + j];
            if (R > 1) { if (v) atomicAdd(a.counts + ((s0 + (uint64_t)j) / R) * ncls + c, v); }   // the syndrome's R ladders, summed
            else if (a.accumulate) a.counts[s0 * ncls + i] += v;
            else a.counts[s0 * ncls + i] = v;
        }
    if (a.swap_acc != nullptr) {
#pragma unroll 1
        for (int i = tid_e; i < cnt * (2 * NC - 1); i += nthreads) {
            const int j = i / (2 * NC - 1), c = i - j * (2 * NC - 1);
            if (c < NC - 1) a.swap_acc[(s0 + j) * (NC - 1) + c] = lds_all[gdw + c * 64 + j];
            else if (a.nerr_sum != nullptr) a.nerr_sum[(s0 + j) * NC + (c - (NC - 1))] = lds_all[gdw + (NC + c - (NC - 1)) * 64 + j];
        }
    }
    if (slot_e == 0 && lane_e < cnt && R > 1) {
        const uint64_t row = (s0 + lane_e) / R;
        if (a.samples != nullptr) atomicAdd(a.samples + row, samples);
        if (a.tops0 != nullptr) atomicAdd(a.tops0 + row, tops0);
        if (a.steps_done != nullptr) atomicMax(a.steps_done + row, done ? steps_done : (uint32_t)a.nsteps);
        if (a.converged != nullptr && !conv_ok) a.converged[row] = 0;               // the caller presets 1: all R ladders converged
    } else
    if (slot_e == 0 && lane_e < cnt) {
        if (a.samples != nullptr) a.samples[s0 + lane_e] = a.accumulate ? a.samples[s0 + lane_e] + samples : samples;
        if (a.steps_done != nullptr) a.steps_done[s0 + lane_e] = USET ? (cm_done ? cm_steps : (uint32_t)a.nsteps) : done ? steps_done : (uint32_t)a.nsteps;
        if (a.converged != nullptr) a.converged[s0 + lane_e] = (uint8_t)conv_ok;
        if (a.tops0 != nullptr) a.tops0[s0 + lane_e] = tops0;
        if (a.flags != nullptr)
            for (int c = 0; c < NC; ++c) a.flags[(s0 + lane_e) * NC + c] = (uint8_t)(fin[c * 64 + lane_e] >> 31);
    }
    if constexpr (BIASED) {
        if (alpha_noise && a.neff != nullptr && lane_e < cnt)
            a.neff[(s0 + lane_e) * NC + slot_e] = neffb[((((uint32_t)a.nsteps & 1u) ^ 1u) * NC + slot_e) * 64 + lane_e];
    }
    if (a.write_states && a.states != nullptr) {
        uint8_t *dst = a.states + s0 * (uint64_t)NC * nq;
        const int per = NC * nq, total = cnt * per;
#pragma unroll 1      // (unrolled, its index divisions spill a register of the whole kernel to scratch)
        for (int o = tid_e; o < total; o += nthreads) {
            const int j = o / per, rem = o - j * per, c = rem / nq, q = rem - c * nq;
            const uint32_t sidc = (fin[c * 64 + j] >> 16) & 0xFFu;
            dst[o] = (uint8_t)((st[(sidc * W + (q >> 4)) * 64 + j] >> ((q & 15) * 2)) & 3u);
        }
    }
}

// dynamic LDS of one workgroup
inline size_t ladder_launch_lds(const LadderArgs &a)
{
    size_t lds = sizeof(uint32_t) * (size_t)ladder_group_dwords(a.Nc, a.W, a.ncls, ladder_gen_dwords(a.code, a.noise, a.scan, a.n_gen, a.Nc, a.nq, a.n_types));
    if (a.swap_acc != nullptr) lds += ladder_stats_lds_bytes(a.Nc);   // per-lane counters behind the group's region
    return lds;
}
// a workgroup whose LDS footprint lets at most two of them share a CU (or which has more than 8 waves) runs at most 4 waves
// per SIMD whatever its register count: such shapes take the 128-VGPR instantiations that draw the top chain's Philox blocks
// ahead (PRE)
inline bool ladder_wants_pre(const LadderArgs &a)
{
    // (up to 8 rungs whose LDS footprint leaves a CU two workgroups anyway: the PRE kernels' 105-119 VGPRs mean 4 waves per SIMD, which is also two
    // workgroups of 8 waves.  A longer ladder would get ONE: measured at toric L = 9, Nc = 9 / 12 / 16: 0.31 / 0.39 / 0.44 with PRE against
    // 0.37 / 0.59 / 0.45 without -- round 4; those shapes took PRE until then)
    return a.Nc >= 3 && a.Nc * 64 <= 512 && 3 * ladder_launch_lds(a) > 160 * 1024 && a.thr_logical != 0 && !(a.tune & 2u);
}

// launch `fn` (one of the instantiations above) on the grid the arguments imply
// (`persistent`: fn is a QUEUE instantiation -- only those run on the capped grid and take the rest of the batch from the
// counter; a.queue without such a kernel is an error here, so a plain kernel can never run on a grid that skips ladders)
inline hipError_t launch_ladder_fn(const void *fn, const LadderArgs &a, hipStream_t stream, bool persistent)
{
    unsigned grid = (unsigned)((a.N + 63) / 64);
    const unsigned block = (unsigned)a.Nc * 64u;
    if ((a.queue != nullptr) != persistent) return hipErrorInvalidValue;
    if (persistent && a.grid_cap && grid > a.grid_cap) grid = a.grid_cap;
    const size_t lds = ladder_launch_lds(a);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {   // beyond the default dynamic-LDS window (160 KiB per CU on gfx950)
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    void *kargs[] = {const_cast<LadderArgs *>(&a)};
    hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(block), kargs, lds, stream);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

// the instantiation <MAXT, MINW, CODE, want> if `want` is one of the listed flag sets Fs..., else nullptr
template <int MAXT, int MINW, int CODE, uint32_t... Fs>
inline const void *select_ladder_kernel(uint32_t want)
{
    const void *fn = nullptr;
    ((want == Fs ? (void)(fn = (const void *)ladder_kernel<MAXT, MINW, CODE, Fs>) : (void)0), ...);
    return fn;
}
// ... the same over the code models Cs... (`code` picks)
template <int MAXT, int MINW, uint32_t... Fs>
struct LadderKernels {
    template <int... Cs>
    static const void *of(int code, uint32_t want)
    {
        const void *fn = nullptr;
        ((code == Cs ? (void)(fn = select_ladder_kernel<MAXT, MINW, Cs, Fs...>(want)) : (void)0), ...);
        return fn;
    }
};

// one translation unit per kernel family (parallel builds): each picks among its own instantiations
hipError_t launch_ladder_toric(const LadderArgs &a, hipStream_t stream);      // ladder_toric.hip: toric, depolarizing, random scan
hipError_t launch_ladder_sweep(const LadderArgs &a, hipStream_t stream);      // ladder_sweep.hip: scan = 1, every code
hipError_t launch_ladder_surf(const LadderArgs &a, hipStream_t stream);       // ladder_surf.hip: xzzx / rotated / planar, depolarizing, random scan
hipError_t launch_ladder_biased(const LadderArgs &a, hipStream_t stream);     // ladder_biased.hip: biased and alpha rules
hipError_t launch_ladder_uset(const LadderArgs &a, hipStream_t stream);       // ladder_uset.hip: the unique-chain estimators' set insertion
hipError_t launch_ladder_colour(const LadderArgs &a, hipStream_t stream);     // ladder_colour.hip: scan = 2, one workgroup per ladder, colour-parallel phases
hipError_t launch_ladder_wu(const LadderArgs &a, hipStream_t stream);         // ladder_wu.hip: scan = 3, wave-uniform generator picks, states in registers

}  // namespace qecmc
