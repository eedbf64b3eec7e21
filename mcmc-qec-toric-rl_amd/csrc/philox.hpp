// Philox4x32-10 counter-based RNG (Salmon et al., SC'11) for gfx950.
// One call = one 128-bit block: the draws of one top-chain proposal, or of FOUR non-top proposals (one word each: 20 bits
// pick the generator, 12 bits lead the acceptance uniform, which a refinement block completes to 44 bits on demand) --
// DESIGN.md "RNG addressing".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qecmc {

struct u32x4 { uint32_t x, y, z, w; };

// Counter layout (DESIGN.md "RNG addressing"):
//   c0 = k[31:0], c1 = k[47:32] | sub<<16, c2 = global syndrome index, c3 = stream id
//   key = 64-bit seed.  k = proposal index of the slot (>> 2 for the non-top blocks of four; ladder-step index for the
//   swap stream), sub = 0 top-chain proposal, 1 four non-top proposals, 2 top-chain acceptance, 3 sweep mode,
//   4 acceptance refinement of the four non-top proposals; for the swap stream the block number within one swap sweep.
constexpr uint32_t kSwapStream = 0x100u;

__host__ __device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);   // one v_bitop3_b32 instead of two v_xor_b32
#else
    return a ^ b ^ c;
#endif
}

__host__ __device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                        uint32_t k0, uint32_t k1)
{
#ifndef QECMC_PHILOX_ROUNDS
#define QECMC_PHILOX_ROUNDS 10      // Philox4x32-10, the Random123 default (tools/ builds a 7-round library for the ceiling analysis of DESIGN.md only)
#endif
#pragma unroll
    for (int r = 0; r < QECMC_PHILOX_ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
        const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

__host__ __device__ __forceinline__ u32x4 philox_block(uint64_t k, uint32_t sub, uint32_t syndrome,
                                                       uint32_t stream, uint32_t seed_lo, uint32_t seed_hi)
{
    return philox4x32_10((uint32_t)k, (uint32_t)((k >> 32) & 0xFFFFu) | (sub << 16), syndrome, stream,
                         seed_lo, seed_hi);
}

// Logical-operator positions of a top-chain proposal share its block (k,0) with the
// operator draws (DESIGN.md "RNG addressing"): X_pos of layer 0 / 1 = low 30 bits of
// word 1 / 2, Z_pos of layer 0 / 1 = high / low half of word 3.
__host__ __device__ __forceinline__ uint32_t scale_low30(uint32_t w, uint32_t n) { return (uint32_t)(((uint64_t)(w << 2) * n) >> 32); }
__host__ __device__ __forceinline__ uint32_t scale_u16(uint32_t h, uint32_t n) { return (h * n) >> 16; }

// Non-top proposal word (word k&3 of block (k>>2, 1)): generator g = floor(x20 * G / 2^20) from its top 20 bits (G < 2^11:
// a 24-bit multiply), a12 = its low 12 bits = the leading bits of the 44-bit acceptance uniform (a12 * 2^32 + w) * 2^-44,
// w = word k&3 of block (k>>2, kSubRefine).  accept iff that integer < T44 = ceil(v * 2^44):  a12 < T44>>32, or equal and
// w < (uint32_t)T44 -- the second word is needed once in 4096 proposals.
constexpr uint32_t kSubRefine = 4u;
// The depolarizing top chain (random scan) spends TWO words per proposal, so one block feeds two proposals: proposal k
// owns words A, B = 2 (k & 1), 2 (k & 1) + 1 of block (k >> 1, kSubTopPair); logical iff A[31:16] < ceil(p_logical * 2^16).
//   toric:      A = select[31:16] | op0[15:14] | op1[13:12] | X_pos0[11:0]
//               B = the generator word (g = floor(B * 2 L^2 / 2^32)), or for a logical  Z_pos0[31:21] | X_pos1[20:10] | Z_pos1[9:0]
//   plaquette:  A = select[31:16] | op[15:14] | X_pos[13:0];   B = the generator word, or for a logical  Z_pos = B[31:16]
// (pos = (field * L) >> bits)
constexpr uint32_t kSubTopPair = 5u;
// Chains updated by the non-top rule (mcmc.py:37-43) draw from the DIAGONAL streams: the chain on slot c at ladder step T uses
// stream kDiagStream + (c + T) mod Nc.  Roles rotate downwards in the kernel (a wave works on slot (w - T) mod Nc), so a wave
// stays on one stream for the whole run and the words a step leaves over in its last four-proposal block (two of twelve at
// iters = 10) are still in its registers when the next step needs them.  The top rule keeps stream c = Nc - 1.
constexpr uint32_t kDiagStream = 0x400u;
__host__ __device__ __forceinline__ uint32_t pick_top20(uint32_t x, uint32_t n) { return ((x >> 12) * n) >> 20; }

// int(u * n) for u = x * 2^-32, exactly (toric_model.py:291 `int(random() * size)`)
__host__ __device__ __forceinline__ uint32_t scale_u32(uint32_t x, uint32_t n)
{
    return (uint32_t)(((uint64_t)x * n) >> 32);
}

}  // namespace qecmc
