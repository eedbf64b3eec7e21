// scan = 3 (QECMC_SCAN_WAVE): the reference's random-scan chain with a generator pick that the 64 ladders of a wavefront SHARE.
//
// Chain.update_chain (src/mcmc.py:19-43) picks a stabilizer generator uniformly and independently of the state
// (toric_model.py:287-296), and the ladders of different syndromes never interact.  So the 64 ladders a wavefront runs may all
// test the SAME generator at proposal k while every ladder keeps its own acceptance uniform: each ladder's chain has exactly the
// reference's law, only the noise of different syndromes becomes correlated.  What that buys on CDNA4:
//   * the four sites of a proposal are wave-uniform: word index and bit shift are scalars, so every rung's qubit_matrix can live
//     in REGISTERS (2 bits per qubit, W dwords per lane) and be addressed with the VGPR index mode (s_set_gpr_idx_on): one
//     VALU instruction reads a site (shift by a scalar, SDWA byte placement) and one updates it (v_xor under the accept mask);
//     no LDS traffic, no per-lane table gathers, no address arithmetic in the proposal loop;
//   * dE from the four old fields by one v_perm_b32 (a four-entry table per Pauli, applied to the four bytes at once) and one
//     v_sad_u8, which also adds the rung's threshold-row address: 4 (dE + 4) + base in a single instruction;
//   * the pick costs one Philox block per TWO proposals per wavefront (lane l draws the block of proposals 2l, 2l + 1 of a
//     window of up to 128 proposals), the per-ladder acceptance uniform 12 bits: one block per TEN proposals per lane; the 44-bit
//     uniform is completed (a 32-bit refinement word) only when a lane's 12 bits tie with its threshold's;
//   * the top rung (p = 0.75: every move is accepted, mcmc.py:30) applies its stabilizers blindly and collects its logical
//     operators -- wave-uniform now -- in one frame that is applied once per step.
// States move through LDS once per ladder step, when the swap sweep (mcmc.py:94-103) has decided who goes where: every wave
// writes its rung's W words, and after the cascade reads the W words of the rung whose state it receives.
//
// RNG addressing of scan = 3 (the CPU oracle restates it independently, as its scan = 3): by ladder step T and proposal j of the step
// (1 <= iters <= 128), slot c:
//   pick      S = 128 / iters steps share a window: w = T / S, P = (T % S) iters + j; words A, B = 2 (P & 1), 2 (P & 1) + 1 of block
//             (64 w + (P >> 1), sub 9) with ctr[2] = (global ladder index) >> 6 and stream 0x800 + c -- lane P >> 1 of the wavefront draws
//             it --:  g = floor(B G / 2^32); top rung: logical iff A[31:16] < ceil(p_logical 2^16), its fields cut from A[15:0] and B as
//             in the packed layout of scan = 0 (philox.hpp)
//   accept    ten proposals share block (T ceil(iters / 10) + j / 10, sub 10) of the ladder's own index, stream c: a12 = field j % 10
//             (wu_field below); w32 = word j & 3 of block (T ceil(iters / 4) + (j >> 2), sub 11);
//             accept iff a12 2^32 + w32 < ceil(f^dE 2^44)
//   swaps     as in the other scans (block (T, i >> 2) of stream 0x100)
// Batches must start on a multiple of 64 (first_syndrome & 63 == 0) so that a wavefront is one pick group.
#pragma once
#include "ladder_kernel.hpp"

namespace qecmc {

constexpr uint32_t kWuPickStream = 0x800u;
constexpr uint32_t kSubWuPick = 9u, kSubWuAcc = 10u, kSubWuRefine = 11u;

template <int WV> struct WuVec;
template <> struct WuVec<4> { typedef uint32_t type __attribute__((ext_vector_type(4))); };
template <> struct WuVec<8> { typedef uint32_t type __attribute__((ext_vector_type(8))); };
template <> struct WuVec<12> { typedef uint32_t type __attribute__((ext_vector_type(12))); };
template <> struct WuVec<16> { typedef uint32_t type __attribute__((ext_vector_type(16))); };
template <> struct WuVec<32> { typedef uint32_t type __attribute__((ext_vector_type(32))); };

// The state registers are PINNED (the index mode addresses v[base + M0]) -- WV dwords ending at v63 (64-VGPR kernels) or at v127 --
// and every access to them is an asm statement that names the pinned tuple as an operand: the compiler then keeps the value where
// it is (any C++-level element access makes it a value of its own that is copied in and out of the pinned registers around every
// statement).  An "i" operand gives the register number of a static element, v[%c[r]].
template <int WV> constexpr int wu_base() { return WV == 32 ? 96 : 64 - WV; }
#define WU_BY_WV(M)                                                                                                            \
    if constexpr (WV == 4) { M("{v[60:63]}") } else if constexpr (WV == 8) { M("{v[56:63]}") }                                 \
    else if constexpr (WV == 12) { M("{v[52:63]}") } else if constexpr (WV == 16) { M("{v[48:63]}") } else { M("{v[96:127]}") }
#define WU_EACH(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16) M(17) M(18) M(19) \
    M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31)

template <int WV> __device__ __forceinline__ void wu_def(typename WuVec<WV>::type &st)
{
#define M(PIN) asm volatile("" : "=" PIN(st));
    WU_BY_WV(M)
#undef M
}
template <int WV, int w> __device__ __forceinline__ void wu_set(typename WuVec<WV>::type &st, uint32_t v)
{
#define M(PIN) asm volatile("v_mov_b32 v[%c[r]], %[v]" : "+" PIN(st) : [r] "i"(wu_base<WV>() + w), [v] "v"(v));
    WU_BY_WV(M)
#undef M
}
template <int WV, int w> __device__ __forceinline__ uint32_t wu_get(typename WuVec<WV>::type &st)
{
    uint32_t v;
#define M(PIN) asm volatile("v_mov_b32 %[v], v[%c[r]]" : [v] "=v"(v), "+" PIN(st) : [r] "i"(wu_base<WV>() + w));
    WU_BY_WV(M)
#undef M
    return v;
}
// word w ^= m (m wave-uniform)
template <int WV, int w> __device__ __forceinline__ void wu_xor_s(typename WuVec<WV>::type &st, uint32_t m)
{
#define M(PIN) asm volatile("v_xor_b32 v[%c[r]], %[m], v[%c[r]]" : "+" PIN(st) : [r] "i"(wu_base<WV>() + w), [m] "s"(m));
    WU_BY_WV(M)
#undef M
}
// acc += number of non-identity fields of word w (toric_model.py:174-176 on the packed word)
template <int WV, int w> __device__ __forceinline__ void wu_count(typename WuVec<WV>::type &st, uint32_t &acc, uint32_t m55)
{
    uint32_t t;
#define M(PIN)                                                                                                                 \
    asm volatile("v_lshrrev_b32 %[t], 1, v[%c[r]]\n\t"                                                                         \
                 "v_bitop3_b32 %[t], %[t], v[%c[r]], %[m] bitop3:0xa8\n\t"                                                     \
                 "v_bcnt_u32_b32 %[acc], %[t], %[acc]"                                                                         \
                 : [t] "=&v"(t), [acc] "+v"(acc), "+" PIN(st) : [r] "i"(wu_base<WV>() + w), [m] "s"(m55));
    WU_BY_WV(M)
#undef M
}
template <int WV, int w> __device__ __forceinline__ void wu_ds_write(typename WuVec<WV>::type &st, uint32_t addr)
{
#define M(PIN) asm volatile("ds_write_b32 %[a], v[%c[r]] offset:%c[o]" : "+" PIN(st) : [a] "v"(addr), [r] "i"(wu_base<WV>() + w), [o] "i"(w * 256) : "memory");
    WU_BY_WV(M)
#undef M
}
template <int WV, int w> __device__ __forceinline__ void wu_ds_read(typename WuVec<WV>::type &st, uint32_t addr)
{
#define M(PIN) asm volatile("ds_read_b32 v[%c[r]], %[a] offset:%c[o]" : "+" PIN(st) : [a] "v"(addr), [r] "i"(wu_base<WV>() + w), [o] "i"(w * 256) : "memory");
    WU_BY_WV(M)
#undef M
}
template <int WV> __device__ __forceinline__ void wu_ds_wait(typename WuVec<WV>::type &st)
{
#define M(PIN) asm volatile("s_waitcnt lgkmcnt(0)" : "+" PIN(st) : : "memory");
    WU_BY_WV(M)
#undef M
}
// the four old fields of a generator's sites, one per byte of F (junk above bit 1 of every byte): site i sits in word d_i[7:0]
// at bit d_i[12:8]
template <int WV>
__device__ __forceinline__ uint32_t wu_read(typename WuVec<WV>::type &st, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3)
{
    uint32_t F;
#define M(PIN)                                                                                                                 \
    asm volatile("s_set_gpr_idx_on %[d0], 0x2\n\t"                                                                             \
                 "v_lshrrev_b32_sdwa %[F], %[d0], v[%c[b]] dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t"      \
                 "s_set_gpr_idx_idx %[d1]\n\t"                                                                                 \
                 "v_lshrrev_b32_sdwa %[F], %[d1], v[%c[b]] dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:DWORD\n\t" \
                 "s_set_gpr_idx_idx %[d2]\n\t"                                                                                 \
                 "v_lshrrev_b32_sdwa %[F], %[d2], v[%c[b]] dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:DWORD\n\t" \
                 "s_set_gpr_idx_idx %[d3]\n\t"                                                                                 \
                 "v_lshrrev_b32_sdwa %[F], %[d3], v[%c[b]] dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:DWORD\n\t" \
                 "s_set_gpr_idx_off"                                                                                           \
                 : [F] "=&v"(F), "+" PIN(st)                                                                                   \
                 : [d0] "s"(d0), [d1] "s"(d1), [d2] "s"(d2), [d3] "s"(d3), [b] "i"(wu_base<WV>()));
    WU_BY_WV(M)
#undef M
    return F;
}
// the accepted move: word d_i[7:0] ^= x_i (under the caller's exec mask; sites may share a word: four read-modify-writes in order)
template <int WV>
__device__ __forceinline__ void wu_xor(typename WuVec<WV>::type &st, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3,
                                       uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3)
{
#define M(PIN)                                                                                                                 \
    asm volatile("s_set_gpr_idx_on %[d0], 0xA\n\t"                                                                             \
                 "v_xor_b32 v[%c[b]], %[x0], v[%c[b]]\n\t"                                                                     \
                 "s_set_gpr_idx_idx %[d1]\n\t"                                                                                 \
                 "v_xor_b32 v[%c[b]], %[x1], v[%c[b]]\n\t"                                                                     \
                 "s_set_gpr_idx_idx %[d2]\n\t"                                                                                 \
                 "v_xor_b32 v[%c[b]], %[x2], v[%c[b]]\n\t"                                                                     \
                 "s_set_gpr_idx_idx %[d3]\n\t"                                                                                 \
                 "v_xor_b32 v[%c[b]], %[x3], v[%c[b]]\n\t"                                                                     \
                 "s_set_gpr_idx_off"                                                                                           \
                 : "+" PIN(st)                                                                                                 \
                 : [d0] "s"(d0), [d1] "s"(d1), [d2] "s"(d2), [d3] "s"(d3), [x0] "s"(x0), [x1] "s"(x1), [x2] "s"(x2), [x3] "s"(x3),  \
                   [b] "i"(wu_base<WV>()));
    WU_BY_WV(M)
#undef M
}

// LDS carve-up of one workgroup (dwords): the exchange buffer, records, swap bounds, histogram, acceptance rows, swap rows,
// logical masks (+ 64: the frame reads a row with all 64 lanes), stop flag
struct WuLds { int xbuf, rec, swd, hist, thr, swapT, lml, stop, total; };
__host__ __device__ inline int wu_words(int W) { return W <= 4 ? 4 : W <= 8 ? 8 : W <= 12 ? 12 : W <= 16 ? 16 : 32; }   // WV: state words per rung, padded
__host__ __device__ inline WuLds wu_lds(int Nc, int W, int ncls, int L)
{
    W = wu_words(W);               // (the kernel moves and masks whole WV-word states: the padding words stay zero)
    WuLds o;
    o.xbuf = 0;
    o.rec = o.xbuf + Nc * W * 64;
    o.swd = o.rec + Nc * 64;
    o.hist = o.swd + Nc * 64;
    o.thr = o.hist + ncls * 64;               // [Nc][2][9]: high 13 / low 32 bits of ceil(f^dE 2^44), dE + 4 = 0 .. 8
    o.swapT = o.thr + Nc * 18;
    o.lml = o.swapT + Nc * kSwapFast;
    o.stop = o.lml + 4 * (L + 1) * W + 64;
    o.total = o.stop + 4;
    return o;
}

// A Philox block whose key schedule is formed where it is used (two scalar adds per round) instead of being hoisted out of the
// step loop into twenty scalar registers per call site: the kernel runs at 8 waves per SIMD on ~80 SGPRs.
__device__ __forceinline__ u32x4 wu_philox(uint64_t k, uint32_t sub, uint32_t syndrome, uint32_t stream, uint32_t seed_lo, uint32_t seed_hi)
{
    asm volatile("" : "+s"(seed_lo), "+s"(seed_hi));
    return philox_block(k, sub, syndrome, stream, seed_lo, seed_hi);
}

typedef const uint32_t __attribute__((address_space(4))) *wu_const_ptr;
typedef const uint32_t __attribute__((address_space(3))) *wu_lds_ptr;

struct WuCtx { uint32_t n4, cls, flag, tops0, samples, done, conv_ok, steps_done; };
struct WuEnv {
    uint32_t *xbuf, *rec, *swd, *hist, *swapT, *lml;
    volatile uint32_t *stopf;
    uint32_t lds0, thr_off, slot, syn, grp;
    int lane, cnt;
    uint64_t s0;
};

// field f (0 .. 9) of an acceptance block: twelve leading bits of a 44-bit uniform each -- two per word in bits 0-23, the last two
// gathered from the four top bytes
__device__ __forceinline__ uint32_t wu_field(const u32x4 &b, int f)
{
    if (f < 8) {
        const uint32_t w = sel4(b, f >> 1);
        return (f & 1) ? __builtin_amdgcn_ubfe(w, 12u, 12u) : (w & 0xFFFu);
    }
    const uint32_t lo = f == 8 ? b.x : b.z, hi = f == 8 ? b.y : b.w;
    return __builtin_amdgcn_perm(hi, lo, 0x0C0C0703u) & 0xFFFu;      // byte 3 of lo | byte 3 of hi << 8
}

// the step loop of one wave: TOP = the rung that accepts every move (its stabilizers unseen, its logical operators through a frame);
// IT = 10: `iters` known at compile time (decoders.py:25 iters=10), the proposal loop unrolled without guards; IT = 0: any 1 <= iters <= 128
template <int CODE, int WV, bool CONV, bool TOP, int IT>
__device__ __forceinline__ void wu_run(const LadderArgs &a, typename WuVec<WV>::type &st, WuCtx &cx, const WuEnv &ev)
{
    const int L = a.L;
    uint32_t *const lml = ev.lml;
    const uint32_t lds0 = ev.lds0, slot = ev.slot, syn = ev.syn, grp = ev.grp;
    const int lane = ev.lane;
    const uint64_t s0 = ev.s0;
    const bool live = lane < ev.cnt;
    constexpr bool top = TOP;
    const uint32_t G = a.n_gen;
    const uint32_t m55 = 0x55555555u;
    uint32_t n4 = cx.n4, cls = cx.cls, flag = cx.flag, tops0 = cx.tops0, samples = cx.samples;
    [[maybe_unused]] uint32_t burn = 0, conv_start = 0, conv_streak = 0, done = 0, conv_ok = 0, steps_done = 0;
    [[maybe_unused]] uint64_t sumA = 0, sumB = 0;
    const wu_const_ptr desc = (wu_const_ptr)a.wu_desc;
    const uint32_t iters = IT ? (uint32_t)IT : a.iters;
    const uint32_t nch = (iters + 9u) / 10u, nc4 = (iters + 3u) / 4u;             // acceptance / refinement blocks per step
    const uint32_t S = 128u / iters;                                              // ladder steps per pick window (iters <= 128)
    const uint32_t thr16 = (uint32_t)((a.thr_logical + 65535u) >> 16);            // logical iff A[31:16] < thr16
    const uint32_t thr_base = lds0 + ev.thr_off;                                  // LDS byte address of this rung's threshold row
    const uint32_t nbias = 0u - (thr_base + 16u);
    uint32_t pk = 0;                                                              // packed descriptor offsets of a pick window
    [[maybe_unused]] uint32_t pa0 = 0, pa1 = 0, pb0 = 0, pb1 = 0;                 // top rung: the window's words A, B
    uint32_t ws = (uint32_t)(a.step0 % S);                                        // this step's index within its pick window
    uint64_t wi = a.step0 / S;                                                    // ... and the window's
    auto refresh = [&]() {
        // lane l: the picks of proposals 2l, 2l + 1 of the window (one Philox block), as descriptor offsets
        const u32x4 b = wu_philox(wi * 64u + (uint64_t)lane, kSubWuPick, grp, kWuPickStream + slot, a.seed_lo, a.seed_hi);
        const uint32_t g0 = scale_u32(b.y, G), g1 = scale_u32(b.w, G);
        pk = (g0 << 6) | (g1 << 22);
        if (top) { pa0 = b.x; pb0 = b.y; pa1 = b.z; pb1 = b.w; }
    };
    refresh();

    for (uint64_t t = 0; t < a.nsteps; ++t) {
        const uint64_t T = a.step0 + t;
        [[maybe_unused]] uint32_t maskv = 0, cdelta = 0;                           // top rung: the step's frame of logical operators (lane w: word w), class change
        const uint32_t pbase = ws * iters;                                         // the step's first proposal within the window
        for (uint32_t c = 0; c < nch; ++c) {
            [[maybe_unused]] u32x4 ab{0, 0, 0, 0};                                 // this ladder's block of ten 12-bit acceptance uniforms
            if (!top) ab = wu_philox(T * nch + c, kSubWuAcc, syn, slot, a.seed_lo, a.seed_hi);
            uint32_t left = iters - c * 10u;
            if constexpr (IT == 0) asm volatile("" : "+s"(left));                  // (a scalar bound for the guards below)
#pragma unroll
            for (int f = 0; f < 10; ++f) {
                if constexpr (IT == 0) { if ((uint32_t)f >= left) break; }
                const uint32_t P = pbase + c * 10u + (uint32_t)f;                  // proposal of the window: lane P >> 1, half P & 1
                const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)pk, (int)(P >> 1));
                const uint32_t off = (P & 1u) ? r >> 16 : r & 0xFFFFu;
                const wu_const_ptr e = desc + (off >> 2);
                if constexpr (top) {
                    const uint32_t A = (uint32_t)__builtin_amdgcn_readlane((int)((P & 1u) ? pa1 : pa0), (int)(P >> 1));
                    if (thr16 != 0 && (A >> 16) < thr16) {
                        // a logical operator (mcmc.py:23-24; toric_model.py:228-253, xzzx_model.py:340-357): into the frame
                        const uint32_t B = (uint32_t)__builtin_amdgcn_readlane((int)((P & 1u) ? pb1 : pb0), (int)(P >> 1));
                        const int LW = (L + 1) * WV;
                        if constexpr (CODE == kCodeToric) {
                            const uint32_t op0 = (A >> 14) & 3u, op1 = (A >> 12) & 3u;
                            const uint32_t dx0 = (op0 ^ (op0 >> 1)) & 1u, dz0 = op0 >> 1, dx1 = (op1 ^ (op1 >> 1)) & 1u, dz1 = op1 >> 1;
                            if (dx0) maskv ^= lml[(((A & 0xFFFu) * (uint32_t)L) >> 12) * WV + lane];
                            if (dz0) maskv ^= lml[LW + (((B >> 21) * (uint32_t)L) >> 11) * WV + lane];
                            if (dx1) maskv ^= lml[2 * LW + ((((B >> 10) & 0x7FFu) * (uint32_t)L) >> 11) * WV + lane];
                            if (dz1) maskv ^= lml[3 * LW + (((B & 0x3FFu) * (uint32_t)L) >> 10) * WV + lane];
                            if (L & 1) cdelta ^= dx0 | (dz0 << 1) | (dx1 << 2) | (dz1 << 3);
                        } else {
                            const uint32_t op = (A >> 14) & 3u;
                            const uint32_t hx = (op ^ (op >> 1)) & 1u, hz = op >> 1;
                            const uint32_t xp = hx ? ((A & 0x3FFFu) * (uint32_t)L) >> 14 : 0u, zp = hz ? ((B >> 16) * (uint32_t)L) >> 16 : 0u;
                            const uint32_t ax = CODE == kCodeXzzx ? hx : (op & 1u), az = hz;
                            if (ax) maskv ^= lml[xp * WV + lane];
                            if (az) maskv ^= lml[LW + zp * WV + lane];
                            cdelta ^= ax | (az << 1);
                        }
                    } else {
                        // a stabilizer, accepted unseen (mcmc.py:30)
                        const uint32_t d0 = e[0], d1 = e[1], d2 = e[2], d3 = e[3], x0 = e[4], x1 = e[5], x2 = e[6], x3 = e[7];
                        wu_xor<WV>(st, d0, d1, d2, d3, x0, x1, x2, x3);
                    }
                } else {
                    const uint32_t d0 = e[0], d1 = e[1], d2 = e[2], d3 = e[3], x0 = e[4], x1 = e[5], x2 = e[6], x3 = e[7];
                    const uint32_t F = wu_read<WV>(st, d0, d1, d2, d3);
                    uint32_t pv;
                    if constexpr (CODE == kCodeToric) {
                        const uint32_t tlo = e[8];
                        pv = __builtin_amdgcn_perm(tlo, tlo, F & 0x03030303u);      // byte i: 4 (1 + change of the error count at site i)
                    } else {
                        const uint32_t tlo = e[8], thi = e[9];
                        pv = __builtin_amdgcn_perm(thi, tlo, (F & e[11]) | e[10]);  // (null sites and the second Pauli's half of the table)
                    }
                    const uint32_t addr = __builtin_amdgcn_sad_u8(pv, 0u, thr_base); // LDS address of this rung's threshold for dE
                    const uint32_t Th = *(wu_lds_ptr)(uintptr_t)addr;
                    const uint32_t a12 = wu_field(ab, f);
                    bool acc = a12 < Th;
                    if (__builtin_amdgcn_uicmp(a12, Th, 32) != 0) {
                        // the 12 leading bits tie with the threshold's in some lane: the refinement word decides there
                        const uint32_t j = c * 10u + (uint32_t)f;
                        const u32x4 rb = wu_philox(T * nc4 + (j >> 2), kSubWuRefine, syn, slot, a.seed_lo, a.seed_hi);
                        const uint32_t Tl = *(wu_lds_ptr)(uintptr_t)(addr + 36u);
                        if (a12 == Th) acc = sel4(rb, (int)(j & 3u)) < Tl;
                    }
                    if (acc) {
                        wu_xor<WV>(st, d0, d1, d2, d3, x0, x1, x2, x3);
                        n4 = n4 + addr + nbias;                                     // n += dE
                    }
                }
            }
        }
        if (top) {
            // the step's logical operators at once, then the error count (the blind moves did not keep it)
            cls ^= cdelta;
            n4 = 0;
#define QECMC_WU_FLUSH(w)                                                                                    \
            if constexpr (w < WV) {                                                                          \
                wu_xor_s<WV, w>(st, (uint32_t)__builtin_amdgcn_readlane((int)maskv, w));                     \
                wu_count<WV, w>(st, n4, m55);                                                                \
            }
            WU_EACH(QECMC_WU_FLUSH)
#undef QECMC_WU_FLUSH
            n4 <<= 2;
        }
        // the next step's pick window (state-independent; before the barrier, where the other waves are still busy)
        if (++ws == S) { ws = 0; ++wi; refresh(); }

        // ---- Ladder.step's swap sweep (mcmc.py:96-103)
        // (per-step work: its table addresses are formed here, from laundered copies of the shape, instead of being hoisted out of the
        // step loop and kept -- spilled -- in scalar registers across the proposal loop, which runs at 8 waves per SIMD on ~80 SGPRs)
        int NCl = a.Nc, nql = a.nq, ncl = a.ncls, Ll = a.L;
        uint32_t slotl = ev.slot, lds0l = ev.lds0;
        uint32_t *ldsl = ev.xbuf;
        asm volatile("" : "+s"(NCl), "+s"(nql), "+s"(ncl), "+s"(Ll), "+s"(slotl), "+s"(lds0l), "+s"(ldsl));
        const int NC = NCl, nq = nql;
        const uint32_t slot = slotl;
        const WuLds ol = wu_lds(NC, WV, ncl, Ll);
        uint32_t *const rec = ldsl + ol.rec, *const swd = ldsl + ol.swd, *const hist = ldsl + ol.hist, *const swapT = ldsl + ol.swapT;
        volatile uint32_t *const stopf = ldsl + ol.stop;
        const uint32_t xaddr = lds0l + (uint32_t)lane * 4u;
        const int swb = NC - 1 - (int)slot;                 // the top rungs draw the swap uniforms: block swb = pairs 4 swb .. 4 swb + 3
        u32x4 sb{0, 0, 0, 0};
        const bool duty = swb >= 0 && swb < 4 && swb * 4 < NC - 1;
        if (duty) sb = wu_philox(T, (uint32_t)swb, syn, kSwapStream, a.seed_lo, a.seed_hi);
        __syncthreads();                                   // (everybody has read the exchange buffer and the swap uniforms of the step before)
        {
            const uint32_t xo = xaddr + slot * (uint32_t)(WV * 256);
#define QECMC_WU_PUT(w) if constexpr (w < WV) wu_ds_write<WV, w>(st, xo);
            WU_EACH(QECMC_WU_PUT)
            rec[slot * 64u + (uint32_t)lane] = pack_info(n4 >> 2, slot, cls, flag);
            if (duty) {
                uint32_t *p = swd + (uint32_t)(swb * 4) * 64u + (uint32_t)lane;
                const int left = NC - 1 - swb * 4;
                p[0] = sb.x;
                if (left > 1) p[64] = sb.y;
                if (left > 2) p[128] = sb.z;
                if (left > 3) p[192] = sb.w;
            }
        }
        wu_ds_wait<WV>(st);                                 // (the asm stores of the exchange are not in the compiler's count)
        __syncthreads();
        if constexpr (CONV) { if (stopf[t & 1]) break; }    // (written by wave 0 during the step before: uniform for the workgroup)
        {
            // every wave replays the top-down cascade on the published records down to the rung that fills its own slot
            const uint32_t *cur = rec + (uint32_t)lane, *sx = swd + (uint32_t)lane;
            uint32_t car = cur[(NC - 1) * 64], mine = car;
            const int i_stop = slot == 0 ? 0 : (int)slot - 1;
            for (int i = NC - 2; i >= i_stop; --i) {                                 // mcmc.py:96
                const uint32_t lo = cur[i * 64], x = sx[i * 64];
                const int d = (int)(car & 0xFFFFu) - (int)(lo & 0xFFFFu);            // ne_hi - ne_lo, _r_flip :146-149
                // x < ceil(p_diff[i]^d 2^32): the LDS table for d < 64 (entry 0 never passes: d <= 0 flips anyway), else the plan's
                const int dd = d < 1 ? 1 : d;
                bool lt = x < swapT[i * kSwapFast + (dd < kSwapFast ? dd : 0)];
                if (dd >= kSwapFast) lt = (uint64_t)x < a.swap_thr[(size_t)i * (nq + 1) + dd];
                const bool flip = d <= 0 || lt;
                const uint32_t into = flip ? lo : car;                               // what slot i+1 now holds (:98-99)
                car = flip ? car : lo;
                if ((int)slot == i + 1) mine = into;
            }
            if (slot == 0) mine = car;
            // this rung's new state: the W words of the rung it comes from
            const uint32_t xin = xaddr + ((mine >> 16) & 0xFFu) * (uint32_t)(WV * 256);
#define QECMC_WU_TAKE(w) if constexpr (w < WV) wu_ds_read<WV, w>(st, xin);
            WU_EACH(QECMC_WU_TAKE)
            wu_ds_wait<WV>(st);
            n4 = (mine & 0xFFFFu) << 2; cls = (mine >> 24) & 0x3Fu; flag = mine >> 31;
            if (top) flag = 1;                                                       // chains[-1].flag = 1, mcmc.py:100
            if (slot == 0 && !done) {                                                // ladder + PTEQ bookkeeping on rung 0's new state
                tops0 += (NC == 1) | flag;                                           // :101-102
                const uint32_t n0 = mine & 0xFFFFu;
                if (a.counts != nullptr && tops0 >= a.tops_burn) {                   // decoders.py:60-67
                    hist[(CODE == kCodeXzzx ? (cls ^ (cls >> 1)) : cls) * 64 + lane] += 1;
                    samples++;
                    if (CONV && live) {
                        // nbr_errors_bottom_chain[since_burn] = count_errors (:68), logged in HBM: series index i in row burn + i
                        const size_t lN = (size_t)a.N;
                        uint16_t *mylog = a.nlog + (s0 + lane);
                        mylog[(size_t)t * lN] = (uint16_t)n0;
                        const uint32_t l = samples, lo1 = l - 1;
                        const uint32_t a0 = lo1 >> 2, b0 = lo1 >> 1, c0 = (3u * lo1) >> 2, a1 = l >> 2, b1 = l >> 1, c1 = (3u * l) >> 2;
                        sumB += n0;
                        if (c1 != c0) sumB -= mylog[(size_t)(burn + c0) * lN];
                        if (b1 != b0) sumA += mylog[(size_t)(burn + b0) * lN];
                        if (a1 != a0) sumA -= mylog[(size_t)(burn + a0) * lN];
                    }
                } else {
                    burn++;                                                          // resulting_burn_in, :71
                }
                if (CONV && tops0 >= a.TOPS) {                                       // :74
                    const uint32_t l = samples ? samples : 1u;
                    const uint32_t den2 = (l >> 1) - (l >> 2), den4 = l - ((3u * l) >> 2);
                    bool accept = false;                                             // empty slice -> nan -> not accepted
                    if (samples && den2 && den4) accept = fabs((double)sumA / (double)den2 - (double)sumB / (double)den4) < a.eps;   // :96-102
                    if (accept) {
                        if (conv_streak >= a.SEQ) { done = 1; conv_ok = 1; steps_done = (uint32_t)t + 1; }   // :77-78
                        else conv_streak = tops0 - conv_start;                       // :79
                    } else {
                        conv_streak = 0;                                             // :81-82
                        conv_start = tops0;
                    }
                }
            }
            if (CONV && slot == 0 && __all(done || !live)) stopf[(t + 1) & 1] = 1;
            if (slot == 0) flag = 0;                                                 // :103
        }
    }
    cx.n4 = n4; cx.cls = cls; cx.flag = flag; cx.tops0 = tops0; cx.samples = samples; cx.done = done; cx.conv_ok = conv_ok; cx.steps_done = steps_done;
}

template <int MAXT, int MINW, int CODE, int WV, bool CONV>
__global__ __launch_bounds__(MAXT, MINW) void ladder_wu_kernel(const LadderArgs a)
{
    typedef typename WuVec<WV>::type vec_t;
    extern __shared__ uint32_t lds[];
    const int NC = a.Nc, W = a.W, L = a.L, nq = a.nq, ncls = a.ncls;
    const int nthreads = NC * 64;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const uint32_t slot = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);      // this wave's rung (fixed: states move)
    const WuLds o = wu_lds(NC, W, ncls, L);
    uint32_t *xbuf = lds + o.xbuf, *rec = lds + o.rec, *swd = lds + o.swd, *hist = lds + o.hist, *thrT = lds + o.thr;
    uint32_t *swapT = lds + o.swapT, *lml = lds + o.lml;
    volatile uint32_t *stopf = lds + o.stop;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(wu_lds_ptr)lds;                    // LDS byte address of the dynamic segment
    const uint32_t R = a.replicas;
    const uint64_t s0 = (uint64_t)blockIdx.x * 64u;
    const int cnt = a.N > s0 ? (int)((a.N - s0) < 64u ? (a.N - s0) : 64u) : 0;
    const uint32_t syn = a.first_syndrome + (uint32_t)s0 + (uint32_t)lane;        // Philox ctr[2] of this ladder
    const uint32_t grp = (a.first_syndrome + (uint32_t)s0) >> 6;                  // ... of the wavefront's shared picks
    const bool live = lane < cnt;
    const bool top = slot == (uint32_t)(NC - 1);                                  // (the launcher guarantees that this rung accepts every move)

    // ---- tables
    for (int i = tid; i < ncls * 64; i += nthreads) hist[i] = 0;
    if (tid < 4) stopf[tid] = 0;
    for (int i = tid; i < NC * 18; i += nthreads) {
        const int c = i / 18, r = i - c * 18, hi = r < 9, idx = hi ? r : r - 9;
        // dE <= 0 (idx <= 4): always accepted -- a high part no 12-bit uniform reaches; dE = 1..4: ceil(f^dE 2^44)
        const uint64_t t44 = idx <= 4 ? (1ull << 44) : a.acc_thr44[c][idx - 5];
        thrT[i] = hi ? (uint32_t)(t44 >> 32) : (uint32_t)t44;
    }
    for (int i = tid; i < (NC - 1) * kSwapFast; i += nthreads) {
        const int pr = i / kSwapFast, d = i - pr * kSwapFast;
        swapT[i] = (d >= 1 && d <= nq) ? (uint32_t)a.swap_thr[(size_t)pr * (nq + 1) + d] : 0u;
    }
    for (int i = tid; i < 4 * (L + 1) * WV + 64; i += nthreads) {          // rows padded to WV words
        const int row = i / WV, w = i - row * WV;
        lml[i] = (row < 4 * (L + 1) && w < W) ? a.lmask[row * W + w] : 0u;
    }

    // ---- stage this rung's state into registers: Ladder.__init__ copies the seed into every rung (mcmc.py:72), or resume
    vec_t st;
    wu_def<WV>(st);
    uint32_t n4 = 0, cls = 0, flag = top ? 1u : 0u;
    {
        const uint64_t ladder = s0 + (uint64_t)(live ? lane : 0);
        const uint8_t *src = cnt > 0 ? (a.resume ? a.states + (ladder * NC + slot) * (uint64_t)nq : a.init + (ladder / R) * (uint64_t)nq) : nullptr;
#define QECMC_WU_STAGE(w)                                                                                    \
        if constexpr (w < WV) {                                                                              \
            uint32_t word = 0;                                                                               \
            if (w < W && cnt > 0)                                                                            \
                for (int b = 0; b < 16; ++b) {                                                               \
                    const int q = w * 16 + b;                                                                \
                    if (q < nq) word |= (uint32_t)(src[q] & 3u) << (2 * b);                                  \
                }                                                                                            \
            wu_set<WV, w>(st, word);                                                                         \
            n4 += 4u * nnz2(word);                                                                           \
        }
        WU_EACH(QECMC_WU_STAGE)
#undef QECMC_WU_STAGE
        if (cnt > 0) {
            cls = (uint32_t)(CODE == kCodeToric ? toric_eq_class_b(L, src) : surf_eq_class_b(CODE, L, src));
            if (CODE == kCodeXzzx) cls = cls == 0 ? 0u : cls == 1 ? 1u : cls == 2 ? 3u : 2u;   // the internal value v with class = v ^ (v >> 1)
            if (a.resume) flag = a.flags[ladder * NC + slot];
        }
    }
    uint32_t tops0 = 0, samples = 0;                                  // wave 0's per-ladder bookkeeping
    if (slot == 0 && a.resume && live) tops0 = a.tops0[s0 + lane];
    __syncthreads();

    WuCtx cx{n4, cls, flag, tops0, samples, 0u, 0u, 0u};
    WuEnv ev;
    ev.xbuf = xbuf; ev.rec = rec; ev.swd = swd; ev.hist = hist; ev.swapT = swapT; ev.lml = lml; ev.stopf = stopf;
    ev.lds0 = lds0; ev.thr_off = (uint32_t)((o.thr + (int)slot * 18) * 4); ev.slot = slot; ev.syn = syn; ev.grp = grp;
    ev.lane = lane; ev.cnt = cnt; ev.s0 = s0;
    // (the two roles are separate loops: they meet at the step's barriers)
    if (a.iters == 10u) {
        if (top) wu_run<CODE, WV, CONV, true, 10>(a, st, cx, ev);
        else wu_run<CODE, WV, CONV, false, 10>(a, st, cx, ev);
    } else {
        if (top) wu_run<CODE, WV, CONV, true, 0>(a, st, cx, ev);
        else wu_run<CODE, WV, CONV, false, 0>(a, st, cx, ev);
    }
    n4 = cx.n4; cls = cx.cls; flag = cx.flag; tops0 = cx.tops0; samples = cx.samples;
    const uint32_t done = cx.done, conv_ok = cx.conv_ok, steps_done = cx.steps_done;
    const uint32_t xaddr = lds0 + (uint32_t)lane * 4u;
    // ---- results
    __syncthreads();
    {
        const uint32_t xo = xaddr + slot * (uint32_t)(WV * 256);
        WU_EACH(QECMC_WU_PUT)
        wu_ds_wait<WV>(st);
        rec[slot * 64u + (uint32_t)lane] = pack_info(n4 >> 2, slot, cls, flag);
    }
#undef QECMC_WU_PUT
#undef QECMC_WU_TAKE
    __syncthreads();
    if (a.counts != nullptr)
#pragma unroll 1
        for (int i = tid; i < cnt * ncls; i += nthreads) {
            const int j = i / ncls, c = i - j * ncls;
            const uint32_t v = hist[c * 64 + j];
            if (R > 1) { if (v) atomicAdd(a.counts + ((s0 + (uint64_t)j) / R) * ncls + c, v); }
            else if (a.accumulate) a.counts[s0 * ncls + i] += v;
            else a.counts[s0 * ncls + i] = v;
        }
    if (slot == 0 && live && R > 1) {
        const uint64_t row = (s0 + lane) / R;
        if (a.samples != nullptr) atomicAdd(a.samples + row, samples);
        if (a.tops0 != nullptr) atomicAdd(a.tops0 + row, tops0);
        if (a.steps_done != nullptr) atomicMax(a.steps_done + row, done ? steps_done : (uint32_t)a.nsteps);
        if (a.converged != nullptr && !conv_ok) a.converged[row] = 0;
    } else if (slot == 0 && live) {
        if (a.samples != nullptr) a.samples[s0 + lane] = a.accumulate ? a.samples[s0 + lane] + samples : samples;
        if (a.steps_done != nullptr) a.steps_done[s0 + lane] = done ? steps_done : (uint32_t)a.nsteps;
        if (a.converged != nullptr) a.converged[s0 + lane] = (uint8_t)conv_ok;
        if (a.tops0 != nullptr) a.tops0[s0 + lane] = tops0;
        if (a.flags != nullptr)
            for (int c = 0; c < NC; ++c) a.flags[(s0 + lane) * NC + c] = (uint8_t)(rec[c * 64 + lane] >> 31);
    }
    if (a.write_states && a.states != nullptr) {
        uint8_t *dst = a.states + s0 * (uint64_t)NC * nq;
        const int per = NC * nq, total = cnt * per;
#pragma unroll 1
        for (int i = tid; i < total; i += nthreads) {
            const int j = i / per, rem = i - j * per, c = rem / nq, q = rem - c * nq;
            dst[i] = (uint8_t)((xbuf[(c * WV + (q >> 4)) * 64 + j] >> ((q & 15) * 2)) & 3u);
        }
    }
}

template <int CODE, bool CONV>
inline const void *wu_pick(int Nc, int W)
{
    const bool big = Nc * 64 > 512;
#ifdef QECMC_WU_DEV     // development builds: the headline shape only
    return (!big && W > 8 && W <= 12) ? (const void *)ladder_wu_kernel<512, 8, CODE, 12, CONV> : nullptr;
#else
    if (W <= 4) return big ? (const void *)ladder_wu_kernel<1024, 4, CODE, 4, CONV> : (const void *)ladder_wu_kernel<512, 8, CODE, 4, CONV>;
    if (W <= 8) return big ? (const void *)ladder_wu_kernel<1024, 4, CODE, 8, CONV> : (const void *)ladder_wu_kernel<512, 8, CODE, 8, CONV>;
    if (W <= 12) return big ? (const void *)ladder_wu_kernel<1024, 4, CODE, 12, CONV> : (const void *)ladder_wu_kernel<512, 8, CODE, 12, CONV>;
    if (W <= 16) return big ? (const void *)ladder_wu_kernel<1024, 4, CODE, 16, CONV> : (const void *)ladder_wu_kernel<512, 8, CODE, 16, CONV>;
    return big ? (const void *)ladder_wu_kernel<1024, 4, CODE, 32, CONV> : (const void *)ladder_wu_kernel<512, 4, CODE, 32, CONV>;
#endif
}

// one translation unit per code family (parallel builds)
const void *wu_kernel_toric(bool conv, int Nc, int W);       // ladder_wu.hip
const void *wu_kernel_surf(int code, bool conv, int Nc, int W);   // ladder_wu_surf.hip

}  // namespace qecmc
