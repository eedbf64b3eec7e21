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
// (32 words -- toric L <= 16, the one-layer codes L <= 22 --: v48 .. v79 of an 80-VGPR kernel, 6 waves per SIMD)
template <int WV> constexpr int wu_base() { return WV == 32 ? 48 : 64 - WV; }
#define WU_BY_WV(M)                                                                                                            \
    if constexpr (WV == 4) { M("{v[60:63]}") } else if constexpr (WV == 8) { M("{v[56:63]}") }                                 \
    else if constexpr (WV == 12) { M("{v[52:63]}") } else if constexpr (WV == 16) { M("{v[48:63]}") }                          \
    else { static_assert(WV == 32, "state widths: 4, 8, 12, 16, 32 words"); M("{v[48:79]}") }
#define WU_EACH(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) \
                   M(16) M(17) M(18) M(19) M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31)

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
// (row: the word's row of the rung's region in the exchange buffer -- w itself, or w - 16 for the upper half of a 32-word state, which
// passes through the buffer in two halves)
template <int WV, int w, int row = w> __device__ __forceinline__ void wu_ds_write(typename WuVec<WV>::type &st, uint32_t addr)
{
#define M(PIN) asm volatile("ds_write_b32 %[a], v[%c[r]] offset:%c[o]" : "+" PIN(st) : [a] "v"(addr), [r] "i"(wu_base<WV>() + w), [o] "i"(row * 256) : "memory");
    WU_BY_WV(M)
#undef M
}
template <int WV, int w, int row = w> __device__ __forceinline__ void wu_ds_read(typename WuVec<WV>::type &st, uint32_t addr)
{
#define M(PIN) asm volatile("ds_read_b32 v[%c[r]], %[a] offset:%c[o]" : "+" PIN(st) : [a] "v"(addr), [r] "i"(wu_base<WV>() + w), [o] "i"(row * 256) : "memory");
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

// LDS carve-up of one workgroup (dwords): the exchange buffer (W words per rung), records, swap uniforms, histogram, acceptance rows,
// swap rows, logical masks (rows padded to WV words, + 64: the frame reads a row with all 64 lanes), stop / refill flags, and -- the
// criterion kernels -- wave 0's per-ladder bookkeeping [kWuBk][64] and the refill mailbox [2][64]
struct WuLds { int xbuf, rec, swd, hist, thr, swapT, lml, stop, bk, mail, bot, cht, nef, lnb, bot2, total; };
constexpr int kWuBk = 13;      // tops0, samples, burn, conv_start, conv_streak, sumA lo / hi, sumB lo / hi, state (done | pending << 1 | has << 3),
                               // steps_done, converged, the lane's ladder (QUEUE)
constexpr int kWuBkAlpha = 17; // ... and the alpha rule's second pair of window sums (n_x + n_y): sumAxy lo / hi, sumBxy lo / hi
__host__ __device__ inline int wu_words(int W) { return W <= 4 ? 4 : W <= 8 ? 8 : W <= 12 ? 12 : W <= 16 ? 16 : 32; }   // WV: state words per rung, padded
__host__ __device__ inline int wu_words_min(int WV) { return WV == 4 ? 1 : WV == 32 ? 17 : WV - 3; }      // the narrowest W a WV-word kernel serves
constexpr int kWuHalf = 16;                                                                               // rows per rung of a 32-word kernel's exchange buffer
// rows per rung of the exchange buffer.  The fixed-length kernels give a rung the kernel's padded width (16 per half of a 32-word state): the padding
// words travel with the rest (they are zero), so that no transfer tests the lattice's width at run time on the scalar unit, the busiest unit of these
// kernels (-30 tests per step at 29 words).  The criterion kernels keep the tight layout -- W rows, a transfer of a word at or beyond wu_words_min
// tests the width --: padded rows cost the headline shape's criterion kernel its fourth workgroup per CU (42 KB instead of 39.9) and the route 11 %.
__host__ __device__ inline int wu_rows(int W, bool conv) { return W > 16 ? kWuHalf : conv ? W : wu_words(W); }
// (alpha rule: the 9 x 9 table of a proposal's count change as two fp16 numbers, the slots' n_eff attributes as doubles [Nc][64], ln(pz_i / pz_i+1),
// and -- criterion runs -- slot 0's n_eff record by step parity)
__host__ __device__ inline WuLds wu_lds(int Nc, int W, int ncls, int L, bool conv, bool alpha = false)
{
    const int WV = wu_words(W);
    WuLds o;
    o.xbuf = 0;
    o.rec = o.xbuf + Nc * wu_rows(W, conv) * 64;
    o.swd = o.rec + Nc * 64;
    o.hist = o.swd + Nc * 64;
    o.thr = o.hist + ncls * 64;               // [Nc][2][9]: high 13 / low 32 bits of ceil(f^dE 2^44), dE + 4 = 0 .. 8
    o.swapT = o.thr + Nc * 18;
    o.lml = o.swapT + Nc * kSwapFast;
    o.stop = o.lml + 4 * (L + 1) * WV + 64;
    o.bk = o.stop + 4;
    o.mail = o.bk + (conv ? (alpha ? kWuBkAlpha : kWuBk) * 64 : 0);   // [2][64] refill orders by step parity
    o.bot = o.mail + (conv ? 2 * 64 : 0);                    // [2][64] the record that landed in rung 0, by step parity
    o.cht = o.bot + (conv ? 2 * 64 : 0);
    o.nef = (o.cht + (alpha ? 84 : 0) + 1) & ~1;             // (doubles: 8-byte aligned)
    o.lnb = o.nef + (alpha ? Nc * 128 : 0);
    o.bot2 = o.lnb + (alpha ? 2 * Nc : 0);
    o.total = o.bot2 + (alpha && conv ? 2 * 64 : 0);
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
typedef uint32_t __attribute__((address_space(3))) *wu_lds_rw;

constexpr uint32_t kWuDead = 0xFFFFFFFFu, kWuKeep = 0xFFFFFFFEu;      // refill mailbox: no ladder left for the lane / the lane keeps its ladder

struct WuCtx { uint32_t n4, cls, flag, tops0, samples, done, conv_ok, steps_done, nef; };
struct WuEnv {
    uint32_t lds0, thr_off, lml_off, cht_off, slot, grp, lad;    // lad: the lane's first ladder of this launch (kWuDead: none)
    int lane;
    uint64_t chunk_hi;                                  // QUEUE: end of the workgroup's share of the batch
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

// one Metropolis proposal of a rung below the top (mcmc.py:38-42) on the 64 ladders of the wave: the generator of descriptor e[0..11],
// a12 = the lanes' leading acceptance bits.  The 12 bits decide unless they tie with the threshold's in some lane (once in 4096 per lane).
template <int CODE, int WV>
__device__ __forceinline__ void wu_propose(typename WuVec<WV>::type &st, uint32_t &n4, uint32_t a12, uint32_t thr_base, uint32_t nbias,
                                           uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3,
                                           uint32_t tlo, uint32_t thi, uint32_t omask, uint32_t amask,
                                           uint64_t refine_k, uint32_t refine_w, uint32_t syn, uint32_t slot, uint32_t seed_lo, uint32_t seed_hi)
{
    const uint32_t F = wu_read<WV>(st, d0, d1, d2, d3);
    uint32_t pv;
    if constexpr (CODE == kCodeToric) pv = __builtin_amdgcn_perm(tlo, tlo, F & 0x03030303u);   // byte i: 4 (1 + change of the error count at site i)
    else pv = __builtin_amdgcn_perm(thi, tlo, (F & amask) | omask);                            // (null sites and the second Pauli's half of the table)
    const uint32_t addr = __builtin_amdgcn_sad_u8(pv, 0u, thr_base);                           // LDS address of this rung's threshold for dE
    const uint32_t Th = *(wu_lds_ptr)(uintptr_t)addr;
    bool acc = a12 < Th;
    if (__builtin_amdgcn_uicmp(a12, Th, 32) != 0) {
        const u32x4 rb = wu_philox(refine_k, kSubWuRefine, syn, slot, seed_lo, seed_hi);
        const uint32_t Tl = *(wu_lds_ptr)(uintptr_t)(addr + 36u);
        if (a12 == Th) acc = sel4(rb, (int)refine_w) < Tl;
    }
    if (acc) {
        wu_xor<WV>(st, d0, d1, d2, d3, x0, x1, x2, x3);
        n4 = n4 + addr + nbias;                                                                 // n += dE
    }
}

// ---- the alpha rule (src/mcmc_alpha.py) -----------------------------------------------------------------------------------------
typedef _Float16 wu_half2 __attribute__((ext_vector_type(2)));
typedef const double __attribute__((address_space(3))) *wu_lds_dptr;
typedef double __attribute__((address_space(3))) *wu_lds_drw;

// nx / nz / nxy += the fields of word w that are 1 / 3 / 1 or 2 (Chain_alpha.__init__, mcmc_alpha.py:18-22, on the packed word)
template <int WV, int w> __device__ __forceinline__ void wu_count3(typename WuVec<WV>::type &st, uint32_t &nx, uint32_t &nz, uint32_t &nxy, uint32_t m55)
{
    uint32_t t, u;
#define M(PIN)                                                                                                                 \
    asm volatile("v_lshrrev_b32 %[t], 1, v[%c[r]]\n\t"                                                                         \
                 "v_bitop3_b32 %[u], %[t], v[%c[r]], %[m] bitop3:0x08\n\t"                                                     \
                 "v_bcnt_u32_b32 %[nx], %[u], %[nx]\n\t"                                                                       \
                 "v_bitop3_b32 %[u], %[t], v[%c[r]], %[m] bitop3:0x80\n\t"                                                     \
                 "v_bcnt_u32_b32 %[nz], %[u], %[nz]\n\t"                                                                       \
                 "v_bitop3_b32 %[u], %[t], v[%c[r]], %[m] bitop3:0x28\n\t"                                                     \
                 "v_bcnt_u32_b32 %[nxy], %[u], %[nxy]"                                                                         \
                 : [t] "=&v"(t), [u] "=&v"(u), [nx] "+v"(nx), [nz] "+v"(nz), [nxy] "+v"(nxy), "+" PIN(st)                      \
                 : [r] "i"(wu_base<WV>() + w), [m] "s"(m55));
    WU_BY_WV(M)
#undef M
}
template <int WV, int w> __device__ __forceinline__ void wu_count_x(typename WuVec<WV>::type &st, uint32_t &nx, uint32_t m55)
{
    uint32_t t;
#define M(PIN)                                                                                                                 \
    asm volatile("v_lshrrev_b32 %[t], 1, v[%c[r]]\n\t"                                                                         \
                 "v_bitop3_b32 %[t], %[t], v[%c[r]], %[m] bitop3:0x08\n\t"                                                     \
                 "v_bcnt_u32_b32 %[nx], %[t], %[nx]"                                                                           \
                 : [t] "=&v"(t), [nx] "+v"(nx), "+" PIN(st) : [r] "i"(wu_base<WV>() + w), [m] "s"(m55));
    WU_BY_WV(M)
#undef M
}
// the state's counts, packed n_x | n_z << 10 | (n_x + n_y) << 20
// (W: the words in use -- wave-uniform; the padding words of a WV-word kernel are zero and need not be looked at)
template <int WV> __device__ __forceinline__ uint32_t wu_counts_packed(typename WuVec<WV>::type &st, uint32_t m55, int W = WV)
{
    uint32_t nx = 0, nz = 0, nxy = 0;
#define QECMC_WU_C3(w) if constexpr (w < WV) { if (w < wu_words_min(WV) || w < W) wu_count3<WV, w>(st, nx, nz, nxy, m55); }
    WU_EACH(QECMC_WU_C3)
#undef QECMC_WU_C3
    return nx | (nz << 10) | (nxy << 20);
}
// n_eff = n_z + alpha (n_x + n_y) as Chain_alpha forms it (mcmc_alpha.py:22,58) from the record n_z | n_xy << 16
__device__ __forceinline__ double wu_neff(uint32_t rec, double alpha)
{
#pragma clang fp contract(off)
    return (double)(rec & 0xFFFFu) + alpha * (double)(rec >> 16);
}
// Ladder_alpha.r_flip (mcmc_alpha.py:118-123) on the slots' attributes: u < (pz_lo / pz_hi) ** (n_eff_hi - n_eff_lo), the power as
// det_exp(e ln b) like the CPU oracle -- behind a single-precision estimate of 2^32 times it that settles all but ~6e-5 of the tests
// (the estimate's relative error stays below 5e-6: the exponent is rounded to a float of magnitude <= 32 wherever the outcome is open)
__device__ __forceinline__ bool wu_alpha_flip(uint32_t x, double ne_hi, double ne_lo, double lnb)
{
#pragma clang fp contract(off)
    const double e = ne_hi - ne_lo;
    const double y = e * lnb;
    if (!(y < 0.0)) return true;                                       // (det_exp returns 1: every u passes)
    const float ef = __builtin_amdgcn_exp2f(__builtin_fmaf((float)y, 1.44269504f, 32.0f));
    const float xf = (float)x, band = __builtin_fmaf(ef, 3.0e-5f, 2.0f);
    if (xf < ef - band) return true;
    if (xf > ef + band) return false;
    return (double)x * (1.0 / 4294967296.0) < det_exp(y);
}

struct WuAl { uint32_t Dh, any, Nb; };     // the counts' change since the step began (D_xy | D_z << 16, fp16 integers), "a move was accepted", the counts then

// one proposal of a rung below the top under the alpha rule (mcmc_alpha.py:61-70): accept iff u < p_n / p_b with p_b frozen when the step
// began (Q3).  log2 of the ratio is lxy D_xy + lz D_z with D the count change since then, so two fmas and a v_exp_f32 give 2^12 p_n / p_b to
// well within a unit (the plan has checked 4 iters max|l| <= 2000: ladder_kernel.hpp, the same estimate), and only a lane whose twelve
// leading uniform bits lie within a cell or two of it evaluates the reference's expression -- on the power tables, in its order.
template <int CODE, int WV>
__device__ __forceinline__ void wu_propose_alpha(const LadderArgs &a, typename WuVec<WV>::type &st, WuAl &al, uint32_t a12, uint32_t cht_addr,
                                                 float lxy, float lz, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3,
                                                 uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t tlo, uint32_t thi, uint32_t omask, uint32_t amask,
                                                 uint64_t refine_k, uint32_t refine_w, uint32_t syn, uint32_t slot, uint32_t m55)
{
    const uint32_t F = wu_read<WV>(st, d0, d1, d2, d3);
    const uint32_t pv = __builtin_amdgcn_perm(thi, tlo, (F & amask) | omask);      // byte i: 4 ((dz + 1) + 9 (dxy + 1)) of site i (a missing site: 0)
    const uint32_t addr = __builtin_amdgcn_sad_u8(pv, 0u, cht_addr);               // LDS address of the proposal's (D_xy, D_z)
    const uint32_t dch = *(wu_lds_ptr)(uintptr_t)addr;
    const wu_half2 pD = __builtin_bit_cast(wu_half2, al.Dh) + __builtin_bit_cast(wu_half2, dch);
    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf((float)pD.x, lxy, __builtin_fmaf((float)pD.y, lz, 12.0f)));
    const float dm = e - (float)a12;
    bool acc = dm >= 2.0f;
    if (!acc && dm >= -1.0f) {
        // the reference's expression (mcmc_alpha.py:64-68): counts of the proposal and of the state the step began with
        const int nq = a.nq, T1 = nq + 1;
        const double *bt = a.bias_tbl + (size_t)slot * 4 * T1;
        uint32_t nxc = 0;
#define QECMC_WU_CX(w) if constexpr (w < WV) wu_count_x<WV, w>(st, nxc, m55);
        WU_EACH(QECMC_WU_CX)
#undef QECMC_WU_CX
        int dx = 0;
        {
            const uint32_t P0 = x0 >> ((d0 >> 8) & 31u), P1 = x1 >> ((d1 >> 8) & 31u), P2 = x2 >> ((d2 >> 8) & 31u), P3 = x3 >> ((d3 >> 8) & 31u);
            const uint32_t f0 = F & 3u, f1 = (F >> 8) & 3u, f2 = (F >> 16) & 3u, f3 = (F >> 24) & 3u;
            dx += (int)((f0 ^ P0) == 1u) - (int)(f0 == 1u);
            dx += (int)((f1 ^ P1) == 1u) - (int)(f1 == 1u);
            dx += (int)((f2 ^ P2) == 1u) - (int)(f2 == 1u);
            dx += (int)((f3 ^ P3) == 1u) - (int)(f3 == 1u);
        }
        const int bx = (int)(al.Nb & 1023u), bz = (int)((al.Nb >> 10) & 1023u), bxy = (int)(al.Nb >> 20);
        const int cx = (int)nxc + dx, cz = bz + (int)(float)pD.y, cxy = bxy + (int)(float)pD.x;
        const double pn = bt[cx] * bt[T1 + (cxy - cx)] * bt[2 * T1 + cz] * bt[3 * T1 + (nq - cxy - cz)];
        const double pb = bt[bx] * bt[T1 + (bxy - bx)] * bt[2 * T1 + bz] * bt[3 * T1 + (nq - bxy - bz)];
        const double ratio = pn / pb;
        const double ulo = (double)a12 * (1.0 / 4096.0);
        acc = ulo + (1.0 / 4096.0) <= ratio;
        if (!acc && ulo < ratio) {
            const u32x4 rb = wu_philox(refine_k, kSubWuRefine, syn, slot, a.seed_lo, a.seed_hi);
            const uint64_t v44 = ((uint64_t)a12 << 32) | sel4(rb, (int)refine_w);
            acc = (double)v44 * (1.0 / 17592186044416.0) < ratio;
        }
    }
    if (acc) {
        wu_xor<WV>(st, d0, d1, d2, d3, x0, x1, x2, x3);
        al.Dh = __builtin_bit_cast(uint32_t, pD);
        al.any = 1u;
    }
}

// this rung's seed configuration of ladder `lad`, packed 2 bits per qubit, into the lane's column of the rung's rows of the exchange
// buffer (Ladder.__init__ copies the seed into every rung, mcmc.py:72; resume: the rung's own state) -- a short runtime loop; the caller
// then takes the words into its state registers with the exchange's own static reads.  Returns 4 x the error count and the class.
// (a 32-word kernel stages words [w0, w1) = [0, 16) and [16, W) in two calls, each into rows 0 .. of the rung's region; the class comes with the first)
template <int CODE>
// (rows: the rows of the region to fill -- those beyond the words of the range get zeros: the padding words of the state)
__device__ __forceinline__ void wu_stage_lds(const LadderArgs &a, uint64_t lad, uint32_t slot, wu_lds_rw xrow, uint32_t &n4, uint32_t &cls, int w0 = 0, int w1 = -1, int rows = 0)
{
    const int NC = a.Nc, W = a.W, L = a.L, nq = a.nq;
    const uint8_t *src = a.resume ? a.states + (lad * NC + slot) * (uint64_t)nq : a.init + (lad / a.replicas) * (uint64_t)nq;
    uint32_t cnt = 0;
    if (w1 < 0) w1 = W;
#pragma unroll 1
    for (int w = w0; w < w1; ++w) {
        uint32_t word = 0;
#pragma unroll 4
        for (int b = 0; b < 16; ++b) {
            const int q = w * 16 + b;
            if (q < nq) word |= (uint32_t)(src[q] & 3u) << (2 * b);
        }
        xrow[(w - w0) * 64] = word;
        cnt += nnz2(word);
    }
#pragma unroll 1
    for (int r = w1 - w0; r < rows; ++r) xrow[r * 64] = 0u;
    if (w0 != 0) { n4 += 4u * cnt; return; }
    n4 = 4u * cnt;
    cls = (uint32_t)(CODE == kCodeToric ? toric_eq_class_b(L, src) : surf_eq_class_b(CODE, L, src));
    if (CODE == kCodeXzzx) cls = cls == 0 ? 0u : cls == 1 ? 1u : cls == 2 ? 3u : 2u;   // the internal value v with class = v ^ (v >> 1)
}

// the step loop of one wave: TOP = the rung that accepts every move (its stabilizers unseen, its logical operators through a frame);
// IT = 10: `iters` known at compile time (decoders.py:25 iters=10), the proposal loop unrolled without guards; IT = 0: any 1 <= iters <= 128.
// QUEUE (with CONV): a persistent grid; a lane whose ladder has ended takes the next one of its workgroup's share of the batch (in lane
// order among the lanes that end together, so the assignment does not depend on timing): the acceptance and swap uniforms follow
// the ladder (its index, its own step), the generator picks the lane's position (group, workgroup step).
// ALPHA: the alpha noise model's ladder (src/mcmc_alpha.py; xzzx / rotated codes): wu_propose_alpha, slot-bound n_eff attributes (Q4) -- a wave IS
// a slot here, so its attribute is a register --, the floating-point swap test, the criterion on the logged count pairs.
template <int CODE, int WV, bool CONV, bool QUEUE, bool TOP, int IT, bool ALPHA>
__device__ __forceinline__ void wu_run(const LadderArgs &a, typename WuVec<WV>::type &st, WuCtx &cx, const WuEnv &ev)
{
    static_assert(!QUEUE || CONV, "the work queue serves the runs that stop by the criterion");
    static_assert(!ALPHA || CODE != kCodeToric, "the alpha rule: xzzx / rotated codes");
    const int L = a.L;
    wu_lds_ptr const lml = (wu_lds_ptr)(uintptr_t)(ev.lds0 + ev.lml_off);
    const uint32_t lds0 = ev.lds0, slot = ev.slot, grp = ev.grp;
    const int lane = ev.lane;
    constexpr bool top = TOP;
    const uint32_t G = a.n_gen;
    const uint32_t m55 = 0x55555555u;
    uint32_t n4 = cx.n4, cls = cx.cls, flag = cx.flag;
    [[maybe_unused]] uint32_t nef = cx.nef;                                       // ALPHA: this slot's n_eff attribute as n_z | (n_x + n_y) << 16
    [[maybe_unused]] float lxyf = 0.0f, lzf = 0.0f;
    if constexpr (ALPHA) { lxyf = a.bias_l2f[ev.slot][0]; lzf = a.bias_l2f[ev.slot][1]; }
    [[maybe_unused]] const uint32_t cht_base = ev.lds0 + ev.cht_off;
    uint32_t syn = a.first_syndrome + ev.lad;                                     // Philox ctr[2] of the lane's ladder
    [[maybe_unused]] uint32_t t0 = 0;                                             // QUEUE: the workgroup step the lane's ladder started at
    [[maybe_unused]] uint32_t tops0 = cx.tops0, samples = 0;                      // wave 0 of the fixed-length kernels: in registers
    const wu_const_ptr desc = (wu_const_ptr)a.wu_desc;
    const uint32_t iters = IT ? (uint32_t)IT : a.iters;
    const uint32_t nch = (iters + 9u) / 10u, nc4 = (iters + 3u) / 4u;             // acceptance / refinement blocks per step
    const uint32_t S = 128u / iters;                                              // ladder steps per pick window (iters <= 128)
    const uint32_t thr16 = (uint32_t)((a.thr_logical + 65535u) >> 16);            // logical iff A[31:16] < thr16
    const uint32_t thr_base = lds0 + ev.thr_off;                                  // LDS byte address of this rung's threshold row
    uint32_t nbias = 0u - (thr_base + 16u);
    asm volatile("" : "+s"(nbias));                                               // (one v_add3_u32 per accepted move: n4 + address + nbias)
    uint32_t pk = 0;                                                              // packed descriptor offsets of a pick window
    [[maybe_unused]] uint32_t pa0 = 0, pa1 = 0, pb0 = 0, pb1 = 0;                 // top rung: the window's words A, B
    [[maybe_unused]] uint32_t pf_a = 0, pf_b = 0, pf_c = 0, pf_l = 0;             // the booking wave: log entries fetched ahead, for sample pf_l
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

    // (the criterion kernels book a step behind the barrier of the next one: without the queue one more, unbooked, step follows the last)
    for (uint64_t t = 0; QUEUE || t < a.nsteps + (CONV ? 1u : 0u); ++t) {
        // the ladder's own step: what addresses its acceptance and swap uniforms
        const uint64_t T = QUEUE ? (uint64_t)((uint32_t)t - t0) : a.step0 + t;
        [[maybe_unused]] uint32_t maskv = 0, cdelta = 0;                           // top rung: the step's frame of logical operators (lane w: word w), class change
        const uint32_t pbase = ws * iters;                                         // the step's first proposal within the window
        [[maybe_unused]] WuAl al{0u, 0u, 0u};
        if constexpr (ALPHA && !top) al.Nb = wu_counts_packed<WV>(st, m55, a.W);   // p_b's counts (mcmc_alpha.py:38-41)
        for (uint32_t c = 0; c < nch; ++c) {
            [[maybe_unused]] u32x4 ab{0, 0, 0, 0};                                 // this ladder's block of ten 12-bit acceptance uniforms
            if (!top) ab = wu_philox(T * nch + c, kSubWuAcc, syn, slot, a.seed_lo, a.seed_hi);
            uint32_t left = iters - c * 10u;
            if constexpr (IT == 0) asm volatile("" : "+s"(left));                  // (a scalar bound for the guards below)
#pragma unroll
            for (int f = 0; f < 10; ++f) {
                if constexpr (IT == 0) { if ((uint32_t)f >= left) continue; }     // (skipped, not left: a loop with one exit unrolls)
                const uint32_t P = pbase + c * 10u + (uint32_t)f;                  // proposal of the window: lane P >> 1, half P & 1
                if constexpr (!top && IT == 10) { if (f & 1) continue; }           // (done with its pair)
                // (iters = 10: a step's first proposal sits at an even index of its window, so the lane that holds proposal P is half the step's
                // base plus a constant)
                const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)pk, IT == 10 ? (int)((pbase >> 1) + ((c * 10u + (uint32_t)f) >> 1)) : (int)(P >> 1));
                const uint32_t off = (P & 1u) ? r >> 16 : r & 0xFFFFu;
                // (byte offsets: multiples of 64 -- added to the table's address as they are)
                typedef const char __attribute__((address_space(4))) *wu_const_bytes;
                const wu_const_ptr e = (wu_const_ptr)((wu_const_bytes)desc + off);
                if constexpr (top) {
                    const uint32_t A = (uint32_t)__builtin_amdgcn_readlane((int)((P & 1u) ? pa1 : pa0), (int)(P >> 1));
                    // a logical operator (mcmc.py:23-24; toric_model.py:228-253, xzzx_model.py:340-357) goes into the frame; a stabilizer is applied
                    // unseen (mcmc.py:30)
#define QECMC_WU_LOGICAL()                                                                                                       \
                        const uint32_t B = (uint32_t)__builtin_amdgcn_readlane((int)((P & 1u) ? pb1 : pb0), (int)(P >> 1)); \
                        const int LW = (L + 1) * WV; \
                        if constexpr (CODE == kCodeToric) { \
                            const uint32_t op0 = (A >> 14) & 3u, op1 = (A >> 12) & 3u; \
                            const uint32_t dx0 = (op0 ^ (op0 >> 1)) & 1u, dz0 = op0 >> 1, dx1 = (op1 ^ (op1 >> 1)) & 1u, dz1 = op1 >> 1; \
                            if (dx0) maskv ^= lml[(((A & 0xFFFu) * (uint32_t)L) >> 12) * WV + lane]; \
                            if (dz0) maskv ^= lml[LW + (((B >> 21) * (uint32_t)L) >> 11) * WV + lane]; \
                            if (dx1) maskv ^= lml[2 * LW + ((((B >> 10) & 0x7FFu) * (uint32_t)L) >> 11) * WV + lane]; \
                            if (dz1) maskv ^= lml[3 * LW + (((B & 0x3FFu) * (uint32_t)L) >> 10) * WV + lane]; \
                            if (L & 1) cdelta ^= dx0 | (dz0 << 1) | (dx1 << 2) | (dz1 << 3); \
                        } else { \
                            const uint32_t op = (A >> 14) & 3u; \
                            const uint32_t hx = (op ^ (op >> 1)) & 1u, hz = op >> 1; \
                            const uint32_t xp = hx ? ((A & 0x3FFFu) * (uint32_t)L) >> 14 : 0u, zp = hz ? ((B >> 16) * (uint32_t)L) >> 16 : 0u; \
                            const uint32_t ax = CODE == kCodeXzzx ? hx : (op & 1u), az = hz; \
                            if (ax) maskv ^= lml[xp * WV + lane]; \
                            if (az) maskv ^= lml[LW + zp * WV + lane]; \
                            cdelta ^= ax | (az << 1); \
                        }
                    if constexpr (WV == 32) {
                        // (32 words: a logical proposal passes through the stabilizer's XOR too, with zeros, so that the state tuple has ONE path
                        // through the loop body -- a branch around an asm statement that updates the tuple makes the compiler merge two copies of it
                        // where the paths meet, which at 32 words is a round trip of the whole state through scratch per step.  The narrower
                        // kernels keep the branch: their merge costs nothing, and the unified form costs the headline kernel 8 B of scratch.)
                        const bool logical = thr16 != 0 && (A >> 16) < thr16;
                        if (logical) { QECMC_WU_LOGICAL() }
                        const uint32_t d0 = e[0], d1 = e[1], d2 = e[2], d3 = e[3];
                        const uint32_t x0 = logical ? 0u : e[4], x1 = logical ? 0u : e[5], x2 = logical ? 0u : e[6], x3 = logical ? 0u : e[7];
                        wu_xor<WV>(st, d0, d1, d2, d3, x0, x1, x2, x3);
                    } else {
                        if (thr16 != 0 && (A >> 16) < thr16) {
                            QECMC_WU_LOGICAL()
                        } else {
                            const uint32_t d0 = e[0], d1 = e[1], d2 = e[2], d3 = e[3], x0 = e[4], x1 = e[5], x2 = e[6], x3 = e[7];
                            wu_xor<WV>(st, d0, d1, d2, d3, x0, x1, x2, x3);
                        }
                    }
#undef QECMC_WU_LOGICAL
                } else {
                    const uint32_t j = c * 10u + (uint32_t)f;
                    if constexpr (IT == 10) {
                        // (iters even: proposal parity = field parity; a lane of the window holds a pair: one readlane, both descriptors
                        // fetched together, the second one's latency hidden behind the first proposal)
                        if constexpr (ALPHA) {
                          if ((f & 1) == 0) {
                            const wu_const_ptr eb = (wu_const_ptr)((wu_const_bytes)desc + (r >> 16));
                            const uint32_t d0 = e[0], d1 = e[1], d2 = e[2], d3 = e[3], x0 = e[4], x1 = e[5], x2 = e[6], x3 = e[7];
                            const uint32_t am = e[11], tlo = e[12], thi = e[13], om = e[14], coff = e[15];
                            const uint32_t b0 = eb[0], b1 = eb[1], b2 = eb[2], b3 = eb[3], y0 = eb[4], y1 = eb[5], y2 = eb[6], y3 = eb[7];
                            const uint32_t an = eb[11], ulo = eb[12], uhi = eb[13], on = eb[14], cofn = eb[15];
                            __builtin_amdgcn_sched_barrier(0);
                            wu_propose_alpha<CODE, WV>(a, st, al, wu_field(ab, f), cht_base + coff, lxyf, lzf, d0, d1, d2, d3, x0, x1, x2, x3, tlo, thi, om, am,
                                                       T * nc4 + (j >> 2), j & 3u, syn, slot, m55);
                            wu_propose_alpha<CODE, WV>(a, st, al, wu_field(ab, f + 1), cht_base + cofn, lxyf, lzf, b0, b1, b2, b3, y0, y1, y2, y3, ulo, uhi, on, an,
                                                       T * nc4 + ((j + 1u) >> 2), (j + 1u) & 3u, syn, slot, m55);
                          }
                        } else
                        if ((f & 1) == 0) {
                            const wu_const_ptr eb = (wu_const_ptr)((wu_const_bytes)desc + (r >> 16));
                            const uint32_t d0 = e[0], d1 = e[1], d2 = e[2], d3 = e[3], x0 = e[4], x1 = e[5], x2 = e[6], x3 = e[7], tlo = e[8];
                            const uint32_t thi = CODE == kCodeToric ? 0u : e[9], om = CODE == kCodeToric ? 0u : e[10], am = CODE == kCodeToric ? 0u : e[11];
                            const uint32_t b0 = eb[0], b1 = eb[1], b2 = eb[2], b3 = eb[3], y0 = eb[4], y1 = eb[5], y2 = eb[6], y3 = eb[7], ulo = eb[8];
                            const uint32_t uhi = CODE == kCodeToric ? 0u : eb[9], on = CODE == kCodeToric ? 0u : eb[10], an = CODE == kCodeToric ? 0u : eb[11];
                            __builtin_amdgcn_sched_barrier(0);
                            wu_propose<CODE, WV>(st, n4, wu_field(ab, f), thr_base, nbias, d0, d1, d2, d3, x0, x1, x2, x3, tlo, thi, om, am,
                                                 T * nc4 + (j >> 2), j & 3u, syn, slot, a.seed_lo, a.seed_hi);
                            wu_propose<CODE, WV>(st, n4, wu_field(ab, f + 1), thr_base, nbias, b0, b1, b2, b3, y0, y1, y2, y3, ulo, uhi, on, an,
                                                 T * nc4 + ((j + 1u) >> 2), (j + 1u) & 3u, syn, slot, a.seed_lo, a.seed_hi);
                        }
                    } else if constexpr (ALPHA) {
                        const uint32_t d0 = e[0], d1 = e[1], d2 = e[2], d3 = e[3], x0 = e[4], x1 = e[5], x2 = e[6], x3 = e[7];
                        const uint32_t am = e[11], tlo = e[12], thi = e[13], om = e[14], coff = e[15];
                        wu_propose_alpha<CODE, WV>(a, st, al, wu_field(ab, f), cht_base + coff, lxyf, lzf, d0, d1, d2, d3, x0, x1, x2, x3, tlo, thi, om, am,
                                                   T * nc4 + (j >> 2), j & 3u, syn, slot, m55);
                    } else {
                        const uint32_t d0 = e[0], d1 = e[1], d2 = e[2], d3 = e[3], x0 = e[4], x1 = e[5], x2 = e[6], x3 = e[7], tlo = e[8];
                        const uint32_t thi = CODE == kCodeToric ? 0u : e[9], om = CODE == kCodeToric ? 0u : e[10], am = CODE == kCodeToric ? 0u : e[11];
                        wu_propose<CODE, WV>(st, n4, wu_field(ab, f), thr_base, nbias, d0, d1, d2, d3, x0, x1, x2, x3, tlo, thi, om, am,
                                             T * nc4 + (j >> 2), j & 3u, syn, slot, a.seed_lo, a.seed_hi);
                    }
                }
            }
        }
        if (top) {
            // the step's logical operators at once, then the error count (the blind moves did not keep it)
            cls ^= cdelta;
            n4 = 0;
            [[maybe_unused]] uint32_t tx = 0, tz = 0, txy = 0;
#define QECMC_WU_FLUSH(w)                                                                                    \
            if constexpr (w < WV) {                                                                          \
                wu_xor_s<WV, w>(st, (uint32_t)__builtin_amdgcn_readlane((int)maskv, w));                     \
                if constexpr (ALPHA) wu_count3<WV, w>(st, tx, tz, txy, m55);                                 \
                else wu_count<WV, w>(st, n4, m55);                                                           \
            }
            WU_EACH(QECMC_WU_FLUSH)
#undef QECMC_WU_FLUSH
            if constexpr (ALPHA) { n4 = tz + txy; nef = tz | (txy << 16); }           // (every move of this rung is accepted: the attribute follows, :58)
            n4 <<= 2;
        } else if constexpr (ALPHA) {
            // the counts the step ends with; the slot's attribute follows them if a move was accepted (mcmc_alpha.py:70)
            const wu_half2 D = __builtin_bit_cast(wu_half2, al.Dh);
            const uint32_t ez = (uint32_t)((int)((al.Nb >> 10) & 1023u) + (int)(float)D.y), exy = (uint32_t)((int)(al.Nb >> 20) + (int)(float)D.x);
            n4 = (ez + exy) << 2;
            if (al.any) nef = ez | (exy << 16);
        }
        // the next step's pick window (state-independent; before the barrier, where the other waves are still busy)
        if (++ws == S) { ws = 0; ++wi; refresh(); }

        // ---- Ladder.step's swap sweep (mcmc.py:96-103)
        // (per-step work: its table addresses are formed here, from laundered copies of the shape, instead of being hoisted out of the
        // step loop and kept -- spilled -- in scalar registers across the proposal loop, which runs at 8 waves per SIMD on ~80 SGPRs)
        int NCl = a.Nc, nql = a.nq, ncl = a.ncls, Ll = a.L, Wl = a.W;
        uint32_t slotl = ev.slot, lds0l = ev.lds0;
        asm volatile("" : "+s"(NCl), "+s"(nql), "+s"(ncl), "+s"(Ll), "+s"(slotl), "+s"(lds0l), "+s"(Wl));
        const int NC = NCl, nq = nql, ncls = ncl;
        const uint32_t slot = slotl;
        const WuLds ol = wu_lds(NC, Wl, ncls, Ll, CONV, ALPHA);
        // (LDS pointers by address space: a laundered generic pointer would turn every access below into a flat load)
        wu_lds_rw const ldsl = (wu_lds_rw)(uintptr_t)lds0l;
        wu_lds_rw const rec = ldsl + ol.rec, swd = ldsl + ol.swd, hist = ldsl + ol.hist, swapT = ldsl + ol.swapT;
        volatile __attribute__((address_space(3))) uint32_t *const stopf = ldsl + ol.stop;
        [[maybe_unused]] wu_lds_rw const bk = ldsl + ol.bk + (uint32_t)lane, mail = ldsl + ol.mail, bot = ldsl + ol.bot, bot2 = ldsl + ol.bot2;
        [[maybe_unused]] wu_lds_drw const nefd = (wu_lds_drw)(ldsl + ol.nef) + (uint32_t)lane;
        [[maybe_unused]] wu_lds_dptr const lnbd = (wu_lds_dptr)(ldsl + ol.lnb);
        const uint32_t xaddr = lds0l + (uint32_t)lane * 4u;
        constexpr bool PAD = WV == 32 || !CONV;                                      // (wu_rows: padded rows, transfers without width tests)
        const uint32_t xstride = (uint32_t)(WV == 32 ? kWuHalf : PAD ? WV : Wl) * 256u;  // bytes of one rung in the exchange buffer
#ifdef QECMC_WU_TOP_SWAPS
        // the swap uniforms -- block b = pairs 4 b .. 4 b + 3 -- are all drawn by the TOP rung's wave, whose step is the shortest
        const int nblk = (NC - 1 + 3) >> 2;
        u32x4 sbs[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        if constexpr (TOP) {
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (b < nblk) sbs[b] = wu_philox(T, (uint32_t)b, syn, kSwapStream, a.seed_lo, a.seed_hi);
        }
#else
        const int swb = NC - 1 - (int)slot;                 // the top rungs draw the swap uniforms: block swb = pairs 4 swb .. 4 swb + 3
        u32x4 sb{0, 0, 0, 0};
        const bool duty = swb >= 0 && swb < 4 && swb * 4 < NC - 1;
        if (duty) sb = wu_philox(T, (uint32_t)swb, syn, kSwapStream, a.seed_lo, a.seed_hi);
#endif
        __syncthreads();                                   // (everybody has read the exchange buffer and the swap uniforms of the step before)
        {
            const uint32_t xo = xaddr + slot * xstride;
            // (a 32-word state passes through the buffer in two halves: words 0-15 here, the rest behind two more barriers below)
#define QECMC_WU_PUT(w) if constexpr (w < WV && w < kWuHalf) { if (WV == 32 || !CONV || w < wu_words_min(WV) || w < Wl) wu_ds_write<WV, w>(st, xo); }
#define QECMC_WU_PUT_HI(w) if constexpr (w < WV && w >= kWuHalf) wu_ds_write<WV, w, w - kWuHalf>(st, xo);
            WU_EACH(QECMC_WU_PUT)
            rec[slot * 64u + (uint32_t)lane] = pack_info(n4 >> 2, slot, cls, flag);
            if constexpr (ALPHA) nefd[slot * 64u] = wu_neff(nef, a.alpha);
#ifdef QECMC_WU_TOP_SWAPS
            if constexpr (TOP) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (b >= nblk) break;
                    wu_lds_rw p = swd + (uint32_t)(b * 4) * 64u + (uint32_t)lane;
                    const int left = NC - 1 - b * 4;
                    p[0] = sbs[b].x;
                    if (left > 1) p[64] = sbs[b].y;
                    if (left > 2) p[128] = sbs[b].z;
                    if (left > 3) p[192] = sbs[b].w;
                }
            }
#else
            if (duty) {
                wu_lds_rw p = swd + (uint32_t)(swb * 4) * 64u + (uint32_t)lane;
                const int left = NC - 1 - swb * 4;
                p[0] = sb.x;
                if (left > 1) p[64] = sb.y;
                if (left > 2) p[128] = sb.z;
                if (left > 3) p[192] = sb.w;
            }
#endif
        }
        wu_ds_wait<WV>(st);                                 // (the asm stores of the exchange are not in the compiler's count)
        __syncthreads();
        if constexpr (CONV) { if (stopf[t & 1]) break; }    // (written by wave 0 during the step before: uniform for the workgroup)
        {
            // every wave replays the top-down cascade on the published records down to the rung that fills its own slot
            wu_lds_ptr cur = rec + (uint32_t)lane, sx = swd + (uint32_t)lane;
            uint32_t car = cur[(NC - 1) * 64], mine = car;
            const int i_stop = slot == 0 ? 0 : (int)slot - 1;
            for (int i = NC - 2; i >= i_stop; --i) {                                 // mcmc.py:96
                const uint32_t lo = cur[i * 64], x = sx[i * 64];
                bool flip;
                if constexpr (ALPHA) {
                    flip = wu_alpha_flip(x, nefd[(i + 1) * 64], nefd[i * 64], lnbd[i]);   // the slots' attributes: they stay where they are (Q4)
                } else {
                const int d = (int)(car & 0xFFFFu) - (int)(lo & 0xFFFFu);            // ne_hi - ne_lo, _r_flip :146-149
                // x < ceil(p_diff[i]^d 2^32): the LDS table for d < 64 (entry 0 never passes: d <= 0 flips anyway), else the plan's
                const int dd = d < 1 ? 1 : d;
                bool lt = x < swapT[i * kSwapFast + (dd < kSwapFast ? dd : 0)];
                if (dd >= kSwapFast) lt = (uint64_t)x < a.swap_thr[(size_t)i * (nq + 1) + dd];
                flip = d <= 0 || lt;
                }
                const uint32_t into = flip ? lo : car;                               // what slot i+1 now holds (:98-99)
                car = flip ? car : lo;
                if ((int)slot == i + 1) mine = into;
            }
            if (slot == 0) mine = car;
            // this rung's new state: the W words of the rung it comes from
            const uint32_t xin = xaddr + ((mine >> 16) & 0xFFu) * xstride;
#define QECMC_WU_TAKE(w) if constexpr (w < WV && w < kWuHalf) { if (WV == 32 || !CONV || w < wu_words_min(WV) || w < Wl) wu_ds_read<WV, w>(st, xin); }
#define QECMC_WU_TAKE_HI(w) if constexpr (w < WV && w >= kWuHalf) wu_ds_read<WV, w, w - kWuHalf>(st, xin);
            WU_EACH(QECMC_WU_TAKE)
            wu_ds_wait<WV>(st);
            if constexpr (WV == 32) {
                static_assert(WV != 32 || (!CONV && !QUEUE), "the 32-word kernels: fixed-length runs");
                const uint32_t xo = xaddr + slot * xstride;
                __syncthreads();                               // (every wave has taken the lower half of its new state)
                WU_EACH(QECMC_WU_PUT_HI)
                wu_ds_wait<WV>(st);
                __syncthreads();
                WU_EACH(QECMC_WU_TAKE_HI)
                wu_ds_wait<WV>(st);
            }
            n4 = (mine & 0xFFFFu) << 2; cls = (mine >> 24) & 0x3Fu; flag = mine >> 31;
            if (top) flag = 1;                                                       // chains[-1].flag = 1, mcmc.py:100
            if constexpr (!CONV) {
                if (slot == 0) {                                                     // ladder + PTEQ bookkeeping on rung 0's new state
                    tops0 += (NC == 1) | flag;                                       // :101-102
                    if (a.counts != nullptr && tops0 >= a.tops_burn) {               // decoders.py:60-67
                        hist[(CODE == kCodeXzzx ? (cls ^ (cls >> 1)) : cls) * 64 + lane] += 1;
                        samples++;
                    }
                }
            } else if (slot == 0) {
                bot[((uint32_t)t & 1u) * 64u + (uint32_t)lane] = mine;              // (booked by the top rung's wave behind the next step's barrier)
                if constexpr (ALPHA) bot2[((uint32_t)t & 1u) * 64u + (uint32_t)lane] = nef;   // chains[0].n_eff, decoders_biasednoise.py:204
            }
            if constexpr (CONV && TOP) {
                // ---- ladder + PTEQ bookkeeping with the error_based criterion (decoders.py:60-82,93-105), by the wave of the TOP rung -- the
                // one with the shortest step -- and one step behind: rung 0's wave leaves the record that landed in rung 0 at step tb in LDS
                // (by step parity), and this wave books it behind the barrier of step tb + 1, off the critical path of the workgroup.  The
                // per-ladder state lives in LDS between steps.
                auto book = [&](uint64_t tb, uint64_t Tb) {
                    const uint32_t recw = bot[((uint32_t)tb & 1u) * 64u + (uint32_t)lane];
            // the same with the error_based criterion (decoders.py:74-82,93-105); the per-ladder state lives in LDS between steps
            uint32_t b_tops0 = bk[0], b_samples = bk[64], b_burn = bk[128], b_cstart = bk[192], b_cstreak = bk[256], b_state = bk[576];
            uint64_t sumA = (uint64_t)bk[320] | ((uint64_t)bk[384] << 32), sumB = (uint64_t)bk[448] | ((uint64_t)bk[512] << 32);
            [[maybe_unused]] uint64_t sumAxy = 0, sumBxy = 0;
            if constexpr (ALPHA) { sumAxy = (uint64_t)bk[832] | ((uint64_t)bk[896] << 32); sumBxy = (uint64_t)bk[960] | ((uint64_t)bk[1024] << 32); }
            uint32_t has = (b_state >> 3) & 1u, pending = (b_state >> 1) & 3u, done = b_state & 1u;
            const uint32_t Town = (uint32_t)Tb;                                  // the ladder's own step (of the step being booked)
            bool ended = false;
            uint32_t conv_ok = 0;
            if (has && !done && !pending) {
                b_tops0 += (NC == 1) | (recw >> 31);                                     // :101-102
                const uint32_t n0 = recw & 0xFFFFu, cls0 = (recw >> 24) & 0x3Fu;
                if (a.counts != nullptr && b_tops0 >= a.tops_burn) {             // decoders.py:60-67
                    hist[(CODE == kCodeXzzx ? (cls0 ^ (cls0 >> 1)) : cls0) * 64 + lane] += 1;
                    b_samples++;
                    // nbr_errors_bottom_chain[since_burn] = count_errors (:68), logged in HBM: series index i in row burn + i of the
                    // lane's column (QUEUE: a column per lane of the grid, rows = the ladder's own steps)
                    const size_t lN = QUEUE ? (size_t)gridDim.x * 64u : (size_t)a.N;
                    // (ALPHA: chains[0].n_eff -- slot 0's attribute, decoders_biasednoise.py:204 -- logged as its two counts, 4 B)
                    typedef typename std::conditional<ALPHA, uint32_t, uint16_t>::type log_t;
                    log_t *mylog = reinterpret_cast<log_t *>(a.nlog) + ((size_t)blockIdx.x * 64u + (size_t)lane);
                    const uint32_t v0 = ALPHA ? bot2[((uint32_t)tb & 1u) * 64u + (uint32_t)lane] : n0;
                    mylog[(size_t)Town * lN] = (log_t)v0;
                    const uint32_t l = b_samples, lo1 = l - 1;
                    const uint32_t a0 = lo1 >> 2, b0 = lo1 >> 1, c0 = (3u * lo1) >> 2, a1 = l >> 2, b1 = l >> 1, c1 = (3u * l) >> 2;
                    // the (up to three) entries that leave / enter the windows Q2 = series[l/4 : l/2], Q4 = series[3l/4 : l]: old rows of the
                    // log, i.e. HBM round trips -- fetched one sample ahead (below), so that they travel during the proposals of a step
                    // instead of standing on this wave's path (samples below 8 read them in place: their rows are only just written)
                    const bool ahead = pf_l == l && l >= 8u;
                    uint32_t vc = pf_c, vb = pf_b, va = pf_a;
                    if (!ahead) {
                        vc = c1 != c0 ? mylog[(size_t)(b_burn + c0) * lN] : 0u;
                        vb = b1 != b0 ? mylog[(size_t)(b_burn + b0) * lN] : 0u;
                        va = a1 != a0 ? mylog[(size_t)(b_burn + a0) * lN] : 0u;
                    }
                    if constexpr (ALPHA) {
                        sumB += v0 & 0xFFFFu; sumBxy += v0 >> 16;
                        sumB -= vc & 0xFFFFu; sumBxy -= vc >> 16;
                        sumA += vb & 0xFFFFu; sumAxy += vb >> 16;
                        sumA -= va & 0xFFFFu; sumAxy -= va >> 16;
                    } else {
                    sumB += n0;
                    sumB -= vc;
                    sumA += vb;
                    sumA -= va;
                    }
                    {   // ... and the next sample's (the burn-in is over: its offset stays)
                        const uint32_t l2 = l + 1u;
                        const uint32_t a2 = l2 >> 2, b2 = l2 >> 1, c2 = (3u * l2) >> 2;
                        pf_c = c2 != c1 ? mylog[(size_t)(b_burn + c1) * lN] : 0u;
                        pf_b = b2 != b1 ? mylog[(size_t)(b_burn + b1) * lN] : 0u;
                        pf_a = a2 != a1 ? mylog[(size_t)(b_burn + a1) * lN] : 0u;
                        pf_l = l2;
                    }
                } else {
                    b_burn++;                                                    // resulting_burn_in, :71
                }
                if (b_tops0 >= a.TOPS) {                                         // :74
                    const uint32_t l = b_samples ? b_samples : 1u;
                    const uint32_t den2 = (l >> 1) - (l >> 2), den4 = l - ((3u * l) >> 2);
                    bool accept = false;                                         // empty slice -> nan -> not accepted
                    if (b_samples && den2 && den4) {
                        if constexpr (ALPHA) accept = alpha_series_close(sumA, sumAxy, den2, sumB, sumBxy, den4, a.alpha, a.eps);   // decoders_biasednoise.py:229-238
                        else accept = fabs((double)sumA / (double)den2 - (double)sumB / (double)den4) < a.eps;   // :96-102
                    }
                    if (accept) {
                        if (b_cstreak >= a.SEQ) { ended = true; conv_ok = 1; }   // :77-78
                        else b_cstreak = b_tops0 - b_cstart;                     // :79
                    } else {
                        b_cstreak = 0;                                           // :81-82
                        b_cstart = b_tops0;
                    }
                }
                if (QUEUE && !ended && (uint64_t)Town + 1u >= a.nsteps) ended = true;   // the horizon: `steps` of the ladder's own steps
            }
            if constexpr (QUEUE) {
                if (pending) pending -= 1;                                       // (a lane between two ladders: the steps it idles are not booked)
                uint32_t give = kWuKeep;
                const uint64_t em = __ballot(ended);
                if (ended) {
                    // ---- a ladder that ended writes its results at once; its lane takes the workgroup's next one -- in lane order
                    // among the lanes that end at the same step, so the assignment does not depend on timing
                    const uint64_t row = (uint64_t)bk[768] / a.replicas;
                    for (int cc = 0; cc < ncls; ++cc) {
                        const uint32_t v = hist[cc * 64 + lane];
                        hist[cc * 64 + lane] = 0;
                        if (a.replicas > 1) { if (v) atomicAdd(a.counts + row * ncls + cc, v); }
                        else a.counts[row * ncls + cc] = v;
                    }
                    const uint32_t sd = Town + 1u;
                    if (a.replicas > 1) {
                        atomicAdd(a.samples + row, b_samples);
                        if (a.tops0 != nullptr) atomicAdd(a.tops0 + row, b_tops0);
                        if (a.steps_done != nullptr) atomicMax(a.steps_done + row, sd);
                        if (a.converged != nullptr && !conv_ok) a.converged[row] = 0;
                    } else {
                        a.samples[row] = b_samples;
                        if (a.tops0 != nullptr) a.tops0[row] = b_tops0;
                        if (a.steps_done != nullptr) a.steps_done[row] = sd;
                        if (a.converged != nullptr) a.converged[row] = (uint8_t)conv_ok;
                    }
                    b_tops0 = b_samples = b_burn = b_cstart = b_cstreak = 0; sumA = sumB = 0; sumAxy = sumBxy = 0;
                    const uint32_t cand = (uint32_t)stopf[2] + (uint32_t)__popcll(em & ((1ull << lane) - 1ull));
                    give = (uint64_t)cand < ev.chunk_hi ? cand : kWuDead;
                    if (give == kWuDead) has = 0; else { pending = 2; bk[768] = give; }
                }
                // this step's orders: read by every wave behind the NEXT step's barrier (mailbox and flag by step parity)
                mail[((uint32_t)tb & 1u) * 64u + (uint32_t)lane] = give;
                if (lane == 0) { stopf[2] = (uint32_t)stopf[2] + (uint32_t)__popcll(em); }
                if (__all(!has)) stopf[(tb + 2) & 1] = 1;
            } else {
                if (ended) { done = 1; bk[640] = Town + 1u; bk[704] = conv_ok; }   // steps_done, converged
                if (__all(done || !has)) stopf[(tb + 2) & 1] = 1;
            }
            bk[0] = b_tops0; bk[64] = b_samples; bk[128] = b_burn; bk[192] = b_cstart; bk[256] = b_cstreak;
            bk[320] = (uint32_t)sumA; bk[384] = (uint32_t)(sumA >> 32); bk[448] = (uint32_t)sumB; bk[512] = (uint32_t)(sumB >> 32);
            bk[576] = done | (pending << 1) | (has << 3);
            if constexpr (ALPHA) { bk[832] = (uint32_t)sumAxy; bk[896] = (uint32_t)(sumAxy >> 32); bk[960] = (uint32_t)sumBxy; bk[1024] = (uint32_t)(sumBxy >> 32); }
                };
                if (t > 0) book(t - 1, QUEUE ? (uint64_t)((uint32_t)(t - 1) - t0) : a.step0 + t - 1);
            }

            if (slot == 0) flag = 0;                                                 // :103
        }
        if constexpr (QUEUE) {
            // ---- refill: the orders wave 0 wrote during the step before take effect here, behind this step's barrier: every wave stages
            // its own rung of the lane's new ladder (the lane idled this step; its ladder's first step is the next one)
            if (t > 1) {
                const uint32_t give = mail[((uint32_t)t & 1u) * 64u + (uint32_t)lane];          // (the orders of the booking of step t - 2)
                if (__builtin_amdgcn_uicmp(give, kWuKeep, 36) != 0) {                // (some lane has an order: give < kWuKeep -- ICMP_ULT)
                    if (give < kWuKeep) {
                        // (through the lane's own column of this rung's rows of the exchange buffer: whoever still reads that column reads it
                        // for the same lane, whose state is being replaced in every wave)
                        syn = a.first_syndrome + give;
                        t0 = (uint32_t)t + 1u;
                        wu_stage_lds<CODE>(a, (uint64_t)give, slot, ldsl + (slot * (uint32_t)Wl) * 64u + (uint32_t)lane, n4, cls);
                        const uint32_t xme = xaddr + slot * xstride;
#define QECMC_WU_MINE(w) if constexpr (w < WV) { if (w < wu_words_min(WV) || w < Wl) wu_ds_read<WV, w>(st, xme); }
                        WU_EACH(QECMC_WU_MINE)
#undef QECMC_WU_MINE
                        wu_ds_wait<WV>(st);
                        flag = top ? 1u : 0u;
                        if constexpr (ALPHA) { const uint32_t c3 = wu_counts_packed<WV>(st, m55); nef = ((c3 >> 10) & 1023u) | ((c3 >> 20) << 16); }   // Chain_alpha.__init__
                    }
                }
            }
        }
    }
    cx.n4 = n4; cx.cls = cls; cx.flag = flag; cx.tops0 = tops0; cx.samples = samples;
    if constexpr (ALPHA) cx.nef = nef;
}

template <int MAXT, int MINW, int CODE, int WV, bool CONV, bool QUEUE, int IT, bool ALPHA = false>
__global__ __launch_bounds__(MAXT, MINW) void ladder_wu_kernel(const LadderArgs a)
{
    typedef typename WuVec<WV>::type vec_t;
    extern __shared__ uint32_t lds[];
    const int NC = a.Nc, W = a.W, L = a.L, nq = a.nq, ncls = a.ncls;
    const int nthreads = NC * 64;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const uint32_t slot = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);      // this wave's rung (fixed: states move)
    const WuLds o = wu_lds(NC, W, ncls, L, CONV, ALPHA);
    uint32_t *xbuf = lds + o.xbuf, *rec = lds + o.rec, *hist = lds + o.hist, *thrT = lds + o.thr;
    uint32_t *swapT = lds + o.swapT, *lml = lds + o.lml;
    volatile uint32_t *stopf = lds + o.stop;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(wu_lds_ptr)lds;                    // LDS byte address of the dynamic segment
    const uint32_t R = a.replicas;
    // the workgroup's share of the batch: 64 ladders, or -- QUEUE -- a.wu_chunk of them, taken 64 at a time
    const uint64_t s0 = (uint64_t)blockIdx.x * (QUEUE ? (uint64_t)a.wu_chunk : 64u);
    const uint64_t s1 = QUEUE ? (s0 + a.wu_chunk < a.N ? s0 + a.wu_chunk : a.N) : a.N;
    const int cnt = s1 > s0 ? (int)((s1 - s0) < 64u ? (s1 - s0) : 64u) : 0;
    const bool live = lane < cnt;
    const bool top = slot == (uint32_t)(NC - 1);                                  // (the launcher guarantees that this rung accepts every move)

    // ---- tables
    for (int i = tid; i < ncls * 64; i += nthreads) hist[i] = 0;
    if (tid < 4) stopf[tid] = tid == 2 ? (uint32_t)s0 + 64u : 0u;   // [0], [1]: stop, by step parity; [2]: the workgroup's queue (next unassigned ladder)
    if constexpr (CONV) {
        uint32_t *bk = lds + o.bk, *mail = lds + o.mail;
        for (int i = tid; i < (ALPHA ? kWuBkAlpha : kWuBk) * 64; i += nthreads) {
            const int row = i >> 6, l = i & 63;
            bk[i] = row == 9 ? (l < cnt ? 8u : 0u) : row == 12 ? (uint32_t)s0 + (uint32_t)l : 0u;                   // state: has; the lane's ladder
        }
        for (int i = tid; i < 128; i += nthreads) mail[i] = kWuKeep;
    }
    if constexpr (ALPHA) {
        // (D_xy, D_z) of a proposal as two fp16 integers, at byte offset 4 ((D_z + 4) + 9 (D_xy + 4)); ln(pz_i / pz_i+1) of the rung pairs
        for (int i = tid; i < 81; i += nthreads) {
            const wu_half2 h = {(_Float16)(float)(i / 9 - 4), (_Float16)(float)(i % 9 - 4)};
            lds[o.cht + i] = __builtin_bit_cast(uint32_t, h);
        }
        for (int i = tid; i < NC - 1; i += nthreads) reinterpret_cast<double *>(lds + o.lnb)[i] = a.alpha_lnb[i];
    }
    for (int i = tid; i < NC * 18 && !ALPHA; i += nthreads) {
        const int c = i / 18, r = i - c * 18, hi = r < 9, idx = hi ? r : r - 9;
        // dE <= 0 (idx <= 4): always accepted -- a high part no 12-bit uniform reaches; dE = 1..4: ceil(f^dE 2^44)
        const uint64_t t44 = idx <= 4 ? (1ull << 44) : a.acc_thr44[c][idx - 5];
        thrT[i] = hi ? (uint32_t)(t44 >> 32) : (uint32_t)t44;
    }
    for (int i = tid; i < (NC - 1) * kSwapFast && !ALPHA; i += nthreads) {
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
    const uint64_t ladder = s0 + (uint64_t)(live ? lane : 0);
#define QECMC_WU_ZERO(w) if constexpr (w < WV) wu_set<WV, w>(st, 0u);
    WU_EACH(QECMC_WU_ZERO)
#undef QECMC_WU_ZERO
    if (cnt > 0) {
        const int rows = wu_rows(W, CONV);                            // rows of a rung's region of the exchange buffer
        const wu_lds_rw xcol = (wu_lds_rw)(uintptr_t)lds0 + (slot * (uint32_t)rows) * 64u + (uint32_t)lane;
        wu_stage_lds<CODE>(a, ladder, slot, xcol, n4, cls, 0, WV == 32 ? kWuHalf : W, rows);
        const uint32_t xme = lds0 + (uint32_t)lane * 4u + slot * (uint32_t)(rows * 256);
        const int Wl = W;
#define QECMC_WU_MINE(w) if constexpr (w < WV && w < kWuHalf) { if (WV == 32 || !CONV || w < wu_words_min(WV) || w < Wl) wu_ds_read<WV, w>(st, xme); }
        WU_EACH(QECMC_WU_MINE)
#undef QECMC_WU_MINE
        wu_ds_wait<WV>(st);
        if constexpr (WV == 32) {
            // (the upper half through the same rows: this lane's own column, which nobody else reads)
            wu_stage_lds<CODE>(a, ladder, slot, xcol, n4, cls, kWuHalf, W, rows);
#define QECMC_WU_MINE_HI(w) if constexpr (w < WV && w >= kWuHalf) wu_ds_read<WV, w, w - kWuHalf>(st, xme);
            WU_EACH(QECMC_WU_MINE_HI)
#undef QECMC_WU_MINE_HI
            wu_ds_wait<WV>(st);
        }
        if (a.resume) flag = a.flags[ladder * NC + slot];
    }
    uint32_t tops0 = 0;                                               // wave 0's per-ladder bookkeeping (the criterion kernels: in LDS)
    if (slot == 0 && a.resume && live) tops0 = a.tops0[s0 + lane];
    __syncthreads();

    uint32_t nef0 = 0;
    if constexpr (ALPHA) { const uint32_t c3 = wu_counts_packed<WV>(st, 0x55555555u); nef0 = ((c3 >> 10) & 1023u) | ((c3 >> 20) << 16); }   // Chain_alpha.__init__, mcmc_alpha.py:18-22
    WuCtx cx{n4, cls, flag, tops0, 0u, 0u, 0u, 0u, nef0};
    WuEnv ev;
    ev.lds0 = lds0; ev.thr_off = (uint32_t)((o.thr + (int)slot * 18) * 4); ev.lml_off = (uint32_t)(o.lml * 4); ev.cht_off = (uint32_t)(o.cht * 4); ev.slot = slot;
    ev.grp = (a.first_syndrome >> 6) + (uint32_t)blockIdx.x;         // the wavefront's shared picks: its position in the grid
    ev.lad = live ? (uint32_t)ladder : kWuDead;
    ev.lane = lane; ev.chunk_hi = s1;
    // (the two roles are separate loops: they meet at the step's barriers)
    if (top) wu_run<CODE, WV, CONV, QUEUE, true, IT, ALPHA>(a, st, cx, ev);
    else wu_run<CODE, WV, CONV, QUEUE, false, IT, ALPHA>(a, st, cx, ev);
    if constexpr (QUEUE) return;                                      // (every ladder wrote its results when it ended)
    n4 = cx.n4; cls = cx.cls; flag = cx.flag; tops0 = cx.tops0;
    uint32_t samples = cx.samples, done = 0, conv_ok = 0, steps_done = 0;
    const uint32_t xaddr = lds0 + (uint32_t)lane * 4u;
    // ---- results
    __syncthreads();
    if constexpr (CONV) {
        const uint32_t *bk = lds + o.bk + lane;
        tops0 = bk[0]; samples = bk[64]; done = bk[576] & 1u; steps_done = bk[640]; conv_ok = bk[704];
    }
    const int rows = wu_rows(W, CONV);
    {
        const int Wl = W;
        const uint32_t xo = xaddr + slot * (uint32_t)(rows * 256);
        WU_EACH(QECMC_WU_PUT)
        wu_ds_wait<WV>(st);
        rec[slot * 64u + (uint32_t)lane] = pack_info(n4 >> 2, slot, cls, flag);
    }
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
        // (the words the exchange buffer holds: all of them, or -- 32-word kernels -- the lower half, then the upper one)
        uint8_t *dst = a.states + s0 * (uint64_t)NC * nq;
        const int per = NC * nq, total = cnt * per;
        auto copy = [&](int w_lo, int w_hi) {
#pragma unroll 1
            for (int i = tid; i < total; i += nthreads) {
                const int j = i / per, rem = i - j * per, c = rem / nq, q = rem - c * nq, w = q >> 4;
                if (w >= w_lo && w < w_hi) dst[i] = (uint8_t)((xbuf[(c * rows + (w - w_lo)) * 64 + j] >> ((q & 15) * 2)) & 3u);
            }
        };
        copy(0, rows);
        if constexpr (WV == 32) {
            const int Wl = W;
            const uint32_t xo = xaddr + slot * (uint32_t)(rows * 256);
            __syncthreads();
            WU_EACH(QECMC_WU_PUT_HI)
            wu_ds_wait<WV>(st);
            __syncthreads();
            copy(kWuHalf, W);
        }
    }
#undef QECMC_WU_PUT
#undef QECMC_WU_TAKE
#undef QECMC_WU_PUT_HI
#undef QECMC_WU_TAKE_HI
}

// the kernel for a shape: the padded state width (4, 8, 12 or 16 words: toric L <= 11, the one-layer codes L <= 16), 8 waves per SIMD up
// to 8 rungs and 4 beyond; IT = 10: the unrolled proposal loop of `iters` = 10 (decoders.py:25) on up to 8 rungs; QUEUE: the criterion runs (every criterion launch takes the queue kernel: a batch no larger than the grid gives each ladder
// a lane of its own)
template <int CODE, bool CONV, bool QUEUE, int IT>
inline const void *wu_pick_it(int Nc, int W)
{
    const bool big = Nc * 64 > 512;
#ifdef QECMC_WU_DEV     // development builds: the headline shape only
#ifndef QECMC_WU_DEV_QUEUE_MINW
#define QECMC_WU_DEV_QUEUE_MINW 8
#endif
    return (!big && W > 8 && W <= 12) ? (const void *)ladder_wu_kernel<512, (QUEUE ? QECMC_WU_DEV_QUEUE_MINW : 8), CODE, 12, CONV, QUEUE, IT> : nullptr;
#else
    // (9 .. 16 rungs: the same 64-VGPR code with a launch bound of 1 024 threads -- three workgroups of 9 waves, two of 12 .. 16 per CU.  The round's
    // first build had given them 4 waves per SIMD: one workgroup per CU and 0.37 at toric L = 9 with the reference's default Nc = L = 9.)
    if (W <= 4) return big ? (const void *)ladder_wu_kernel<1024, 8, CODE, 4, CONV, QUEUE, IT> : (const void *)ladder_wu_kernel<512, 8, CODE, 4, CONV, QUEUE, IT>;
    if (W <= 8) return big ? (const void *)ladder_wu_kernel<1024, 8, CODE, 8, CONV, QUEUE, IT> : (const void *)ladder_wu_kernel<512, 8, CODE, 8, CONV, QUEUE, IT>;
    if (W <= 12) return big ? (const void *)ladder_wu_kernel<1024, 8, CODE, 12, CONV, QUEUE, IT> : (const void *)ladder_wu_kernel<512, 8, CODE, 12, CONV, QUEUE, IT>;
    if (W <= 16) return big ? (const void *)ladder_wu_kernel<1024, 6, CODE, 16, CONV, QUEUE, IT> : (const void *)ladder_wu_kernel<512, 6, CODE, 16, CONV, QUEUE, IT>;
    // 17 .. 32 words (toric L <= 16, the one-layer codes L <= 22): fixed-length runs of up to 8 rungs, 80 VGPRs at 6 waves per SIMD, the state
    // through the exchange buffer in two halves (52 KB of LDS: three workgroups per CU); not built for the planar code
    if constexpr (!CONV && !QUEUE && CODE != kCodePlanar) { if (!big) return (const void *)ladder_wu_kernel<512, 6, CODE, 32, false, false, IT>; }
    return nullptr;
#endif
}
// variant: 0 fixed length, 2 criterion on the persistent grid
template <int CODE>
inline const void *wu_pick(int variant, int Nc, int W, uint32_t iters)
{
    if (W > 32 || (variant != 0 && variant != 2)) return nullptr;
    // iters = 10 (decoders.py:25): the unrolled proposal loop, for every code and ladder length (same-box A/B at L = 9, config 2's shape: xzzx 0.85 against
    // 0.61 with the general loop -- and 0.73 with the random scan's kernel --, rotated 0.85 / 0.61 / 0.72, planar 0.82 / 0.60 / 0.68)
    if (iters == 10u) return variant == 2 ? wu_pick_it<CODE, true, true, 10>(Nc, W) : wu_pick_it<CODE, false, false, 10>(Nc, W);
    return variant == 2 ? wu_pick_it<CODE, true, true, 0>(Nc, W) : wu_pick_it<CODE, false, false, 0>(Nc, W);
}

// the alpha rule's kernels: xzzx / rotated codes up to 8 state words per rung (L <= 11)
// (IT = 10: PTEQ_alpha's default iters, decoders_biasednoise.py:175)
template <int CODE, int IT>
inline const void *wu_pick_alpha(int variant, int Nc, int W)
{
    if (W > 8 || (variant != 0 && variant != 2)) return nullptr;
    const bool big = Nc * 64 > 512;          // (9 .. 16 rungs -- Ladder_alpha's default is Nc = L, decoders_biasednoise.py:175 --: the same code under a 1 024-thread bound)
    // (the criterion kernels of the alpha rule at 6 waves per SIMD -- 80 VGPRs, 102 SGPRs --: 96-120 B of scratch reloaded every step at 8, 28-40 B at 6,
    // and 4 % faster on the PTEQ_alpha route, same-box A/B; kWuAlphaQueueWaves tells the plan how many workgroups a CU then holds)
#ifndef QECMC_WU_ALPHA_CONV_MINW
#define QECMC_WU_ALPHA_CONV_MINW kWuAlphaQueueWaves
#endif
    if (variant == 2) {
        if (big) return W <= 4 ? (const void *)ladder_wu_kernel<1024, QECMC_WU_ALPHA_CONV_MINW, CODE, 4, true, true, IT, true> : (const void *)ladder_wu_kernel<1024, QECMC_WU_ALPHA_CONV_MINW, CODE, 8, true, true, IT, true>;
        return W <= 4 ? (const void *)ladder_wu_kernel<512, QECMC_WU_ALPHA_CONV_MINW, CODE, 4, true, true, IT, true> : (const void *)ladder_wu_kernel<512, QECMC_WU_ALPHA_CONV_MINW, CODE, 8, true, true, IT, true>;
    }
    if (big) return W <= 4 ? (const void *)ladder_wu_kernel<1024, 8, CODE, 4, false, false, IT, true> : (const void *)ladder_wu_kernel<1024, 8, CODE, 8, false, false, IT, true>;
    return W <= 4 ? (const void *)ladder_wu_kernel<512, 8, CODE, 4, false, false, IT, true> : (const void *)ladder_wu_kernel<512, 8, CODE, 8, false, false, IT, true>;
}

// one translation unit per code family (parallel builds)
const void *wu_kernel_toric(int variant, int Nc, int W, uint32_t iters);            // ladder_wu.hip
const void *wu_kernel_xzzx(int variant, int Nc, int W, uint32_t iters);             // ladder_wu_xzzx.hip
const void *wu_kernel_rotated(int variant, int Nc, int W, uint32_t iters);          // ladder_wu_rotated.hip
const void *wu_kernel_planar(int variant, int Nc, int W, uint32_t iters);           // ladder_wu_planar.hip
const void *wu_kernel_alpha(int code, int variant, int Nc, int W, uint32_t iters);   // ladder_wu_alpha.hip

}  // namespace qecmc
