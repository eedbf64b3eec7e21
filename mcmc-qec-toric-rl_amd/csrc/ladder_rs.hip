// Random-scan parallel-tempering ladder kernel for the toric code (gfx950).
//
// This is the reference's Markov chain (src/mcmc.py:19-43 Chain.update_chain,
// :94-103 Ladder.step, decoders.py:55-68 PTEQ bookkeeping) laid out for CDNA4:
//
//   * one workgroup = 64 syndromes x Nc ladder slots; wavefront w owns slot w
//     (temperature p_ladder[w]) of all 64 syndromes, lane l owns syndrome l.
//     Acceptance thresholds are therefore wave-uniform (SGPRs) and the top
//     slot's logical-operator branch (mcmc.py:23) never diverges against the
//     stabilizer-only slots.
//   * every chain's qubit_matrix lives in LDS for the whole run, packed 2 bits
//     per qubit: word w of state s of lane l sits at dword (w*Nc + s)*64 + l, so
//     a wave's ds_read_b32 / ds_xor_b32 hit bank (l mod 32) whatever (w, s) each
//     lane picks: random-scan access with zero bank conflicts.
//   * proposals: one Philox4x32-10 block = the four uniforms (row, col, op,
//     accept) of one proposal; accept tests are integer compares against
//     host-built thresholds ceil(f^dE * 2^32), so results are bit-identical to
//     the CPU oracle fed the same Philox stream.
//   * swaps (mcmc.py:96-103) move slot->state indices, not data; error counts are
//     carried incrementally (n += dE) instead of recounted (mcmc.py:88-89).
//   * HBM traffic is compulsory only: nq bytes in, ncls counters out per syndrome.
#include "kernels.hpp"
#include "philox.hpp"

namespace qecmc {

size_t ladder_lds_bytes(int L, int Nc, int W, int ncls)
{
    return sizeof(uint32_t) * ((size_t)W * Nc * 64 + 2 * (size_t)Nc * 64 + (size_t)ncls * 64 + 4 * (size_t)(L + 1) * W);
}

__device__ __forceinline__ uint32_t nnz2(uint32_t x) { return __popc((x | (x >> 1)) & 0x55555555u); }

__device__ __forceinline__ uint32_t sel4(const u32x4 &b, int i)
{
    return i == 0 ? b.x : i == 1 ? b.y : i == 2 ? b.z : b.w;
}

// One stabilizer proposal on the packed state at `stw` (word stride `ws` dwords).
// Sites: toric_model.py:261-269.  Returns dE; applies the move iff `accept(dE)`.
template <class Accept>
__device__ __forceinline__ int toric_stab_proposal(uint32_t *stw, uint32_t ws, int L, int LL, uint32_t row,
                                                   uint32_t col, bool isX, Accept accept)
{
    const uint32_t rL = row * L, rc = rL + col;
    const uint32_t cm = col == 0 ? L - 1 : col - 1, cp = col + 1 == (uint32_t)L ? 0 : col + 1;
    const uint32_t rm = row == 0 ? L - 1 : row - 1, rp = row + 1 == (uint32_t)L ? 0 : row + 1;
    uint32_t q[4];
    q[0] = LL + rc;                                   // (1, r, c)
    q[1] = rc;                                        // (0, r, c)
    q[2] = isX ? LL + rL + cm : rL + cp;              // (1, r, c-1) | (0, r, c+1)
    q[3] = isX ? rm * L + col : LL + rp * L + col;    // (0, r-1, c) | (1, r+1, c)
    const uint32_t op = isX ? 1u : 3u;
    uint32_t *ad[4];
    uint32_t sh[4];
    int dE = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ad[i] = stw + (q[i] >> 4) * ws;
        sh[i] = (q[i] & 15u) * 2u;
        const uint32_t f = (*ad[i] >> sh[i]) & 3u;
        dE += (int)(f == 0u) - (int)(f == op);
    }
    if (accept(dE)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __hip_atomic_fetch_xor(ad[i], op << sh[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return dE;
    }
    return 0;
}

template <int MAXT>
__global__ __launch_bounds__(MAXT) void ladder_rs_toric_kernel(const LadderArgs a)
{
    extern __shared__ uint32_t lds[];
    const int NC = a.Nc, W = a.W, L = a.L, LL = L * L, nq = a.nq, ncls = a.ncls;
    const int tid = threadIdx.x, lane = tid & 63, slot = tid >> 6;
    const int nthreads = NC * 64;
    const uint32_t ws = (uint32_t)NC * 64u;       // dword stride between words of one state

    uint32_t *st = lds;                           // [W][NC][64]
    uint32_t *nerr = st + (size_t)W * NC * 64;    // [NC][64]   error count of state s
    uint32_t *perm = nerr + NC * 64;              // [NC][64]   state held by slot c
    uint32_t *hist = perm + NC * 64;              // [ncls][64]
    uint32_t *lmask = hist + ncls * 64;           // [4][L+1][W]

    const uint64_t s0 = (uint64_t)blockIdx.x * 64u;
    const int cnt = (int)((a.N - s0) < 64u ? (a.N - s0) : 64u);
    const uint32_t syn = a.first_syndrome + (uint32_t)s0 + (uint32_t)lane;   // Philox ctr[2]

    for (int i = tid; i < W * NC * 64; i += nthreads) st[i] = 0;
    for (int i = tid; i < ncls * 64; i += nthreads) hist[i] = 0;
    for (int i = tid; i < 4 * (L + 1) * W; i += nthreads) lmask[i] = a.lmask[i];
    __syncthreads();

    // ---- stage the batch: coalesced byte stream -> 2-bit fields in LDS -------------
    if (!a.resume) {
        const uint8_t *src = a.init + s0 * (uint64_t)nq;
        const int total = cnt * nq;
        for (int o = tid; o < total; o += nthreads) {
            const uint32_t v = src[o] & 3u;
            if (v) {
                const int j = o / nq, q = o - j * nq;
                const uint32_t bits = v << ((q & 15) * 2);
                uint32_t *p = st + (size_t)(q >> 4) * ws + j;
                for (int s = 0; s < NC; ++s)      // Ladder.__init__ deep-copies init into every slot (mcmc.py:72)
                    __hip_atomic_fetch_or(p + s * 64, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    } else {
        const uint8_t *src = a.states + s0 * (uint64_t)NC * nq;
        const int per = NC * nq, total = cnt * per;
        for (int o = tid; o < total; o += nthreads) {
            const uint32_t v = src[o] & 3u;
            if (v) {
                const int j = o / per, rem = o - j * per, s = rem / nq, q = rem - s * nq;
                __hip_atomic_fetch_or(st + (size_t)(q >> 4) * ws + s * 64 + j, v << ((q & 15) * 2), __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();
    {
        uint32_t n = 0;
        for (int w = 0; w < W; ++w) n += nnz2(st[(size_t)w * ws + slot * 64 + lane]);
        nerr[slot * 64 + lane] = n;
        perm[slot * 64 + lane] = (uint32_t)slot;
    }
    // per-syndrome ladder bookkeeping lives in the registers of wave 0
    uint32_t flagbits = 1u << (NC - 1);           // chains[-1].flag = 1 (mcmc.py:75)
    uint32_t tops0 = 0, samples = 0;
    if (a.resume && slot == 0 && lane < cnt) {
        flagbits = 0;
        for (int c = 0; c < NC; ++c) flagbits |= (uint32_t)(a.flags[(s0 + lane) * NC + c] != 0) << c;
        tops0 = a.tops0[s0 + lane];
    }
    __syncthreads();

    // wave-uniform acceptance data of this slot
    const uint32_t slot_u = __builtin_amdgcn_readfirstlane(slot);
    const uint32_t t1 = a.acc_thr[slot_u][0], t2 = a.acc_thr[slot_u][1], t3 = a.acc_thr[slot_u][2],
                   t4 = a.acc_thr[slot_u][3];
    const bool acc_all = (a.acc_all_mask >> slot_u) & 1u;
    const bool top_logical = (slot_u == (uint32_t)(NC - 1)) && a.thr_logical != 0;
    const uint32_t iters = a.iters;

    for (uint64_t t = 0; t < a.nsteps; ++t) {
        // ---------------- Chain.update_chain(iters) on every slot (mcmc.py:81-83) -----------
        const uint32_t sid = perm[slot * 64 + lane];
        uint32_t *stw = st + sid * 64 + lane;
        int n = (int)nerr[sid * 64 + lane];
        const uint64_t kbase = a.prop0 + t * iters;
        if (!top_logical) {
            for (uint32_t j = 0; j < iters; ++j) {
                const u32x4 x = philox_block(kbase + j, 0, syn, slot_u, a.seed_lo, a.seed_hi);
                const uint32_t row = scale_u32(x.x, L), col = scale_u32(x.y, L);   // toric_model.py:291-292
                const bool isX = x.z >> 31;                                        // :293-295
                n += toric_stab_proposal(stw, ws, L, LL, row, col, isX, [&](int dE) {
                    const uint32_t thr = dE == 1 ? t1 : dE == 2 ? t2 : dE == 3 ? t3 : t4;
                    return dE <= 0 || acc_all || x.w < thr;                        // mcmc.py:42
                });
            }
        } else {
            for (uint32_t j = 0; j < iters; ++j) {
                const uint64_t k = kbase + j;
                const u32x4 x = philox_block(k, 0, syn, slot_u, a.seed_lo, a.seed_hi);
                if ((uint64_t)x.x < a.thr_logical) {                               // mcmc.py:23
                    // _apply_random_logical, toric_model.py:228-253
                    const uint32_t op0 = x.y >> 30, op1 = x.z >> 30;
                    const u32x4 b = philox_block(k, 1, syn, slot_u, a.seed_lo, a.seed_hi);
                    int nb = 0;
                    uint32_t ix0 = L, iz0 = L, ix1 = L, iz1 = L;                   // row L of each table = identity
                    if (op0 == 1 || op0 == 2) ix0 = scale_u32(sel4(b, nb++), L);
                    if (op0 == 3 || op0 == 2) iz0 = scale_u32(sel4(b, nb++), L);
                    if (op1 == 1 || op1 == 2) ix1 = scale_u32(sel4(b, nb++), L);
                    if (op1 == 3 || op1 == 2) iz1 = scale_u32(sel4(b, nb++), L);
                    const int LW = (L + 1) * W;
                    const uint32_t *m0 = lmask + ix0 * W, *m1 = lmask + LW + iz0 * W, *m2 = lmask + 2 * LW + ix1 * W,
                                   *m3 = lmask + 3 * LW + iz1 * W;
                    bool acc = true;
                    if (!acc_all) {                                                 // only a 1-chain ladder has a top chain below p = 0.75
                        int dE = 0;
                        for (int w = 0; w < W; ++w) {
                            const uint32_t old = stw[(size_t)w * ws];
                            dE += (int)nnz2(old ^ m0[w] ^ m1[w] ^ m2[w] ^ m3[w]) - (int)nnz2(old);
                        }
                        if (dE > 0)                                                 // mcmc.py:30-34
                            acc = philox_block(k, 2, syn, slot_u, a.seed_lo, a.seed_hi).x < a.acc_tbl_top[dE];
                    }
                    if (acc)
                        for (int w = 0; w < W; ++w) {                               // p >= 0.75 accepts all (mcmc.py:30)
                            const uint32_t m = m0[w] ^ m1[w] ^ m2[w] ^ m3[w];
                            const uint32_t old = stw[(size_t)w * ws], neu = old ^ m;
                            stw[(size_t)w * ws] = neu;
                            n += (int)nnz2(neu) - (int)nnz2(old);
                        }
                } else {
                    const uint32_t row = scale_u32(x.y, L), col = scale_u32(x.z, L);
                    n += toric_stab_proposal(stw, ws, L, LL, row, col, (bool)(x.w >> 31), [&](int dE) {
                        if (acc_all || dE <= 0) return true;
                        return philox_block(k, 2, syn, slot_u, a.seed_lo, a.seed_hi).x < a.acc_tbl_top[dE];
                    });
                }
            }
        }
        nerr[sid * 64 + lane] = (uint32_t)n;
        __syncthreads();

        // ---------------- swap sweep + PTEQ bookkeeping, one lane per syndrome ---------------
        if (slot == 0) {
            const uint64_t tstep = a.step0 + t;
            u32x4 blk{0, 0, 0, 0};
            int cur = -1;
            uint32_t hi = perm[(NC - 1) * 64 + lane];
            for (int i = NC - 2; i >= 0; --i) {                                    // mcmc.py:96
                if ((i >> 2) != cur) {
                    cur = i >> 2;
                    blk = philox_block(tstep, (uint32_t)cur, syn, kSwapStream, a.seed_lo, a.seed_hi);
                }
                const uint32_t lo = perm[i * 64 + lane];
                const int d = (int)nerr[hi * 64 + lane] - (int)nerr[lo * 64 + lane];   // ne_hi - ne_lo
                bool flip = d < 0;                                                  // _r_flip, mcmc.py:146
                if (!flip) flip = (uint64_t)sel4(blk, i & 3) < a.swap_thr[(size_t)i * (nq + 1) + d];   // :149
                if (flip) {                                                          // :98-99
                    perm[i * 64 + lane] = hi;
                    perm[(i + 1) * 64 + lane] = lo;
                    const uint32_t fl = (flagbits >> i) & 1u, fh = (flagbits >> (i + 1)) & 1u;
                    flagbits = (flagbits & ~(3u << i)) | (fh << i) | (fl << (i + 1));
                } else {
                    hi = lo;
                }
            }
            flagbits |= 1u << (NC - 1);                                             // mcmc.py:100
            if (flagbits & 1u) { tops0++; flagbits &= ~1u; }                        // :101-103
            if (a.counts != nullptr && tops0 >= a.tops_burn) {                      // decoders.py:60-67
                const uint32_t b0 = perm[lane];
                const uint32_t *sb = st + b0 * 64 + lane;
                const int wb = LL >> 4;
                const uint32_t lowmask = (1u << ((LL & 15) * 2)) - 1u;              // layer-0 fields of the boundary word
                uint32_t acc0 = 0, acc1 = 0;
                for (int w = 0; w < W; ++w) {
                    const uint32_t x = sb[(size_t)w * ws];
                    if (w < wb) acc0 ^= x;
                    else if (w > wb) acc1 ^= x;
                    else { acc0 ^= x & lowmask; acc1 ^= x & ~lowmask; }
                }
                // X component = b0^b1 (values 1,2), Z component = b1 (values 2,3); toric_model.py:317-351
                const uint32_t x1 = __popc((acc0 ^ (acc0 >> 1)) & 0x55555555u) & 1u, z1 = __popc(acc0 & 0xAAAAAAAAu) & 1u;
                const uint32_t x2 = __popc((acc1 ^ (acc1 >> 1)) & 0x55555555u) & 1u, z2 = __popc(acc1 & 0xAAAAAAAAu) & 1u;
                const uint32_t cls = x1 + 2u * z1 + 4u * x2 + 8u * z2;
                hist[cls * 64 + lane] += 1;
                samples++;
            }
        }
        __syncthreads();
    }

    // ---- results: coalesced stores ---------------------------------------------------
    if (a.counts != nullptr)
        for (int i = tid; i < cnt * ncls; i += nthreads) {
            const int j = i / ncls, c = i - j * ncls;
            a.counts[s0 * ncls + i] = hist[c * 64 + j];
        }
    if (slot == 0 && lane < cnt) {
        if (a.samples != nullptr) a.samples[s0 + lane] = samples;
        if (a.tops0 != nullptr) a.tops0[s0 + lane] = tops0;
        if (a.flags != nullptr)
            for (int c = 0; c < NC; ++c) a.flags[(s0 + lane) * NC + c] = (flagbits >> c) & 1u;
    }
    if (a.write_states && a.states != nullptr) {
        uint8_t *dst = a.states + s0 * (uint64_t)NC * nq;
        const int per = NC * nq, total = cnt * per;
        for (int o = tid; o < total; o += nthreads) {
            const int j = o / per, rem = o - j * per, c = rem / nq, q = rem - c * nq;
            const uint32_t sidc = perm[c * 64 + j];
            dst[o] = (uint8_t)((st[(size_t)(q >> 4) * ws + sidc * 64 + j] >> ((q & 15) * 2)) & 3u);
        }
    }
}

hipError_t launch_ladder_rs_toric(const LadderArgs &a, hipStream_t stream)
{
    const unsigned grid = (unsigned)((a.N + 63) / 64);
    const unsigned block = (unsigned)a.Nc * 64u;
    const size_t lds = ladder_lds_bytes(a.L, a.Nc, a.W, a.ncls);
    if (grid == 0) return hipSuccess;
    const void *fn = block <= 512 ? (const void *)ladder_rs_toric_kernel<512> : (const void *)ladder_rs_toric_kernel<1024>;
    if (lds > 64 * 1024) {   // beyond the default dynamic-LDS window (160 KiB per CU on gfx950)
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (block <= 512)
        hipLaunchKernelGGL(ladder_rs_toric_kernel<512>, dim3(grid), dim3(block), lds, stream, a);
    else
        hipLaunchKernelGGL(ladder_rs_toric_kernel<1024>, dim3(grid), dim3(block), lds, stream, a);
    return hipGetLastError();
}

}  // namespace qecmc
