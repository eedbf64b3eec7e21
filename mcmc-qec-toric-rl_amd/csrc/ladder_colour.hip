// scan = 2 (QECMC_SCAN_COLOUR): the latency layout.  One workgroup per LADDER (one syndrome), one wavefront per rung, the lanes
// of a wavefront = the stabilizer generators of one colour phase, proposed all at once.
//
// The reference decodes ONE syndrome per call (decoders.py:25, generate_data.py:136).  The throughput kernels (ladder_kernel.hpp)
// give every chain one lane and advance it one proposal at a time -- 64 syndromes per wavefront; a single syndrome then waits for
// 162 sequential proposals per sweep at L = 9.  Here the parallelism is inside the chain (BASELINE.json's north star:
// "thread-block per syndrome, checkerboard-parallel stabilizer flips, wavefront reductions for the dE count"): generators that
// share no qubit commute and their Metropolis tests are independent, so a whole phase of them is one wavefront instruction
// stream; a sweep is n_phases of those (7 at toric L = 9) instead of 162 proposals.
//
// This is NOT the reference's Markov chain (a systematic scan, like scan = 1): every single-generator Metropolis kernel keeps
// the rung's stationary law, hence so does their composition.  The rule, which the CPU oracle restates (scan = 2 of its model):
//   * the plan cuts the generators into phases (tables.hpp colour_phases: greedy colouring in table order, chunks of <= 64);
//     phase index K of a rung counts from prop0 = step0 * iters; a ladder step = `iters` phases of every rung, phase K uses
//     phase (K mod n_phases) of the table;
//   * top rung (slot Nc - 1, p_logical > 0; it sits at p = 0.75 where every move is accepted, mcmc.py:30): before the phase, with
//     probability p_logical (word 0 of block (K, 0) < ceil(p_logical 2^32)) one uniformly random logical operator drawn from
//     words 1-3 of that block exactly as scan = 1 draws it (toric_model.py:228-253 / xzzx_model.py:340-357);
//   * generator i of the phase draws u = word (K & 3) of block (K >> 2, 8 + i) of the rung's stream (the slot's own for the
//     top rule, the diagonal stream kDiagStream + (slot + step) mod Nc otherwise, as in the other scans);
//     a rung with f < 1 accepts iff u < ceil(f^dE 2^32) (dE <= 0: always; mcmc.py:42); a rung with f >= 1 (where a
//     coin-less sweep would compose to the identity) applies the generator iff the top bit of u is set;
//   * swap sweep, tops0 / class histogram bookkeeping: mcmc.py:94-103 and decoders.py:60-68 unchanged (swap uniforms: word i & 3
//     of block (t, i >> 2) of the swap stream, as in the other scans).
// conv_mode = error_based runs the reference's criterion on wave 0 (the workgroup leaves when its ladder has converged).  In
// fixed-length runs steps_done reports the first ladder step after which tops0 >= TOPS (or `steps`), converged whether it was
// reached: the "time to tops0 >= 10" the latency table of profiles/ quotes.
//
// RULE = 1 / 2: the biased (src/mcmc_biased.py) and alpha (src/mcmc_alpha.py) noise models on the xzzx / rotated codes.  The members of
// a phase are tested at once, so Q3's frozen p_b cannot be carried (a member's ratio would depend on what the members before it did):
// every generator is a Metropolis move for the model's own weight, u < (px / pI)^dxy (pz / pI)^dz with (dxy, dz) the change of
// n_x + n_y and n_z of that generator alone -- an 81-entry threshold table per rung (a.col_thr) --, the law the reference's rule has at
// iters = 1.  The biased top rung is not uniform: it runs the same rule and tests its logical operators (word 0 of block (K, 1) against
// the ratio of the power tables' products, the reference's expression); Ladder_alpha's top rung (pz_tilde = 1: every ratio is 1) takes
// the coin and its logical operators unseen, like the depolarizing one.  RULE = 2 also: the swap test on the slots' n_eff attributes
// (mcmc_alpha.py:118-123), which stay with the slot (Q4: a wave IS a slot here, the attribute is a scalar of the wave) and follow the
// counts only when a move was accepted (:58,:70); the criterion on the logged attribute of slot 0 (decoders_biasednoise.py:204,229-238).
#include "ladder_kernel.hpp"

namespace qecmc {

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// CONV: the error_based convergence criterion of decoders.py:74-82,93-105 on wave 0 (one ladder per workgroup: the workgroup leaves when
// its ladder has converged); steps_done / converged are then the criterion's, as in the other scans.
// Diagnostic build only (tools/steptrace.hip): shader-clock stamps of workgroup 0's waves at the phase boundaries of ladder steps 2000 .. 2031
#ifdef QECMC_STEPTRACE
#define QECMC_CSTAMP(k)                                                                                                        \
    do {                                                                                                                       \
        if (a.dbg && blockIdx.x == 0 && t >= 2000 && t < 2032 && lane == 0)                                                    \
            a.dbg[(size_t)gridDim.x * 4 + (((t - 2000) * 16 + slot) * 8 + (k))] = (k) == 5 ? (uint64_t)slot : (uint64_t)clock64(); \
    } while (0)
#else
#define QECMC_CSTAMP(k) ((void)0)
#endif

// MAXT / MINW: 1 024 threads at 4 waves per SIMD (a development build can give ladders of up to 8 rungs 512 threads at 6 or 8: launch_ladder_colour)
template <int CODE, bool CONV, int RULE = 0, int MAXT = 1024, int MINW = 4>
__global__ __launch_bounds__(MAXT, MINW) void ladder_colour_kernel(const LadderArgs a)
{
    static_assert(RULE == 0 || CODE == kCodeXzzx || CODE == kCodeRotated, "the biased / alpha rules: xzzx and rotated codes");
    extern __shared__ uint32_t lds[];
    const int NC = a.Nc, W = a.W, L = a.L, LL = L * L, nq = a.nq, ncls = a.ncls;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const uint32_t slot = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6);      // this wave's rung: fixed (states move by id)
    uint32_t *st = lds;                                  // [NC][W]   packed states, 2 bits per qubit
    uint32_t *rec = st + NC * W;                         // [2][NC]   slot records by step parity (pack_info)
    uint32_t *swu = rec + 2 * NC;                        // [2][NC]   swap uniforms by step parity
    uint32_t *hist = swu + 2 * NC;                       // [ncls]
    uint16_t *ptab = reinterpret_cast<uint16_t *>(hist + ncls);                       // [n_phases][64]
    uint2 *gtab = reinterpret_cast<uint2 *>(hist + ncls + 32 * a.n_phases + ((ncls + 32 * a.n_phases + NC * W + 4 * NC) & 1));   // [n_gen], 8-byte aligned
    uint32_t *lml = reinterpret_cast<uint32_t *>(gtab + a.n_gen);                      // [4][L+1][W]
    uint32_t *swt = lml + 4 * (a.L + 1) * W;                                          // [NC-1][nq+1] swap thresholds (u32, or u64 as two dwords)
    const bool swap32 = a.swap_fast_ok != 0;
    volatile uint32_t *stopf = swt + (swap32 ? 1 : 2) * (NC > 1 ? NC - 1 : 0) * (a.nq + 1);   // [2] "the ladder has converged", by step parity
    [[maybe_unused]] uint32_t *cthr = const_cast<uint32_t *>(stopf) + 2;                  // RULE != 0: [NC][81] accept iff u <= cthr[9 (dz + 4) + dxy + 4]
    [[maybe_unused]] uint32_t *nefr = cthr + (RULE ? NC * 81 : 0);                          // RULE == 2: [2][NC] the slots' n_eff records (n_z | n_xy << 16) by step parity
    const uint32_t R = a.replicas;
    const uint64_t ladder = blockIdx.x;                  // one workgroup per ladder
    if (ladder >= a.N) return;
    const uint32_t syn = a.first_syndrome + (uint32_t)ladder;
    const uint64_t row = ladder / R;

    // ---- stage: this wave packs the seed configuration into its own slot (Ladder.__init__ copies it into every rung, mcmc.py:72)
    const uint8_t *src = a.init + row * (uint64_t)nq;
    int cnt0 = 0;
    for (int w = lane; w < W; w += 64) {
        uint32_t word = 0;
        for (int b = 0; b < 16; ++b) {
            const int q = w * 16 + b;
            if (q < nq) word |= (uint32_t)(src[q] & 3u) << (2 * b);
        }
        st[slot * W + w] = word;
        cnt0 += (int)nnz2(word);
    }
    for (int c = tid; c < ncls; c += NC * 64) hist[c] = 0;
    if (tid < 2) stopf[tid] = 0;
    for (int i = tid; i < (int)a.n_phases * 64; i += NC * 64) ptab[i] = a.phase_tab[i];
    for (int i = tid; i < (int)a.n_gen; i += NC * 64) gtab[i] = a.gen[i];
    for (int i = tid; i < 4 * (L + 1) * W; i += NC * 64) lml[i] = a.lmask[i];
    if constexpr (RULE != 0) { for (int i = tid; i < NC * 81; i += NC * 64) cthr[i] = a.col_thr[i]; }
    for (int i = tid; i < (NC - 1) * (nq + 1); i += NC * 64) {
        if (swap32) swt[i] = (uint32_t)a.swap_thr[i];      // (entry d = 0 -- 2^32 -- is never looked up: d <= 0 always swaps)
        else { swt[2 * i] = (uint32_t)a.swap_thr[i]; swt[2 * i + 1] = (uint32_t)(a.swap_thr[i] >> 32); }
    }
    __syncthreads();
    // wave-uniform slot state: error count, state id, class, flag (Chain.flag, mcmc.py:75)
    uint32_t n = (uint32_t)wave_sum(cnt0), sid = slot, flag = slot == (uint32_t)(NC - 1);
    uint32_t cls;
    {
        uint32_t c = 0;
        if (lane == 0) {                                 // W words, once: the serial class functions of ladder_kernel.hpp with a lane stride of 1
            if constexpr (CODE == kCodeToric) {
                const int wb = LL >> 4;
                const uint32_t lowmask = (1u << ((LL & 15) * 2)) - 1u;
                uint32_t acc0 = 0, acc1 = 0;
                for (int w = 0; w < W; ++w) {
                    const uint32_t x = st[slot * W + w];
                    if (w < wb) acc0 ^= x;
                    else if (w > wb) acc1 ^= x;
                    else { acc0 ^= x & lowmask; acc1 ^= x & ~lowmask; }
                }
                c = (__popc((acc0 ^ (acc0 >> 1)) & 0x55555555u) & 1u) + 2u * (__popc(acc0 & 0xAAAAAAAAu) & 1u) +
                    4u * (__popc((acc1 ^ (acc1 >> 1)) & 0x55555555u) & 1u) + 8u * (__popc(acc1 & 0xAAAAAAAAu) & 1u);
            } else {
                uint32_t x = 0, z = 0;
                const uint32_t *sb = st + slot * W;
                for (int i = 0; i < L; ++i) {
                    const uint32_t qa = (uint32_t)i, qb = (uint32_t)(i * L);
                    const uint32_t fa = (sb[qa >> 4] >> ((qa & 15u) * 2u)) & 3u, fb = (sb[qb >> 4] >> ((qb & 15u) * 2u)) & 3u;
                    const uint32_t xa = (fa ^ (fa >> 1)) & 1u, za = fa >> 1, xb = (fb ^ (fb >> 1)) & 1u, zb = fb >> 1;
                    if (CODE == kCodeXzzx) { x ^= (i & 1) ? za : xa; z ^= (i & 1) ? xb : zb; }
                    else if (CODE == kCodePlanar) { x ^= xb; z ^= za; }
                    else { x ^= xa; z ^= zb; }
                }
                c = x | (z << 1);
            }
        }
        cls = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
    }
    n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n);
    // RULE == 2: n_z | n_xy << 16 of a state (wave-parallel over its words), and this slot's n_eff attribute as those counts (Chain_alpha.__init__)
    [[maybe_unused]] auto counts_zxy = [&](const uint32_t *sbw) -> uint32_t {
        int cz = 0, cxy = 0;
        for (int w = lane; w < W; w += 64) {
            const uint32_t x = sbw[w];
            cz += __popc(x & (x >> 1) & 0x55555555u);
            cxy += __popc((x ^ (x >> 1)) & 0x55555555u);
        }
        return (uint32_t)__builtin_amdgcn_readfirstlane(wave_sum(cz)) | ((uint32_t)__builtin_amdgcn_readfirstlane(wave_sum(cxy)) << 16);
    };
    [[maybe_unused]] uint32_t nef = 0;
    if constexpr (RULE == 2) nef = counts_zxy(st + slot * W);
    [[maybe_unused]] uint64_t sumAxy = 0, sumBxy = 0;
    uint32_t tops0 = 0, samples = 0, t_reached = 0;     // wave 0's bookkeeping (uniform)
    [[maybe_unused]] uint32_t burn = 0, conv_start = 0, conv_streak = 0, done = 0, steps_done = 0;   // decoders.py:37-48
    [[maybe_unused]] uint64_t sumA = 0, sumB = 0;       // window sums of the logged bottom-chain error counts: Q2 = series[l/4 : l/2], Q4 = series[3l/4 : l]
    const bool acc_all = (a.acc_all_mask >> slot) & 1u;
    const bool top_logical = slot == (uint32_t)(NC - 1) && a.thr_logical != 0;
    const uint32_t thrL1 = (uint32_t)(a.thr_logical - 1);
    // accept iff u <= thr[dE + 4] (dE <= 0 or f >= 1: always -- a rung with f >= 1 takes the coin instead)
    const uint32_t thr1 = a.acc_thr[slot][0] - 1u, thr2 = a.acc_thr[slot][1] - 1u, thr3 = a.acc_thr[slot][2] - 1u, thr4 = a.acc_thr[slot][3] - 1u;
    [[maybe_unused]] const uint32_t *mythr = cthr + slot * 81u;
    const uint32_t iters = a.iters, P = a.n_phases;
    const uint32_t *lmask = lml;
    const int LW = (L + 1) * W;

    // running indices instead of 64-bit remainders on the serial path: the phase of the table (K mod P) and the diagonal stream's
    // offset ((slot + step) mod Nc); the generator entry of the NEXT phase is fetched while the current one is tested -- it does
    // not depend on the state -- which takes two of the three LDS round trips of a phase off the critical path
    uint32_t ph = (uint32_t)(a.prop0 % (uint64_t)P), dg = (uint32_t)(((uint64_t)slot + a.step0) % (uint64_t)NC);
    // ... and so does everything else about a phase except the four state words: word indices, bit shifts, the values to xor in, the
    // Pauli pattern, and -- the uniform being known before dE is -- the largest dE the lane would accept (thresholds fall with dE; a rung
    // with f >= 1 takes the coin: every dE or none).  All of it is prepared one phase ahead, in the shadow of the current phase's chain
    // state words -> fields -> dE -> compare -> xor, which is what a lone workgroup's step time is made of.
    struct Prepared { bool act; uint32_t wi[4], sh[4], xv[4], ops, u; int dmax; };
    uint2 e_next;                                                             // the entry of the phase after the prepared one
    bool act_next;
    u32x4 ub{0, 0, 0, 0};                                                     // this lane's block of uniforms (four consecutive phases)
    auto fetch_entry = [&]() {
        const uint32_t gn = ptab[ph * 64u + (uint32_t)lane];
        ph = ph + 1u == P ? 0u : ph + 1u;
        act_next = gn != 0xFFFFu;
        e_next = gtab[act_next ? gn : 0u];
    };
    // prepare phase Kn of a step on stream strm_n from the entry fetched last (and fetch the one after it)
    auto prepare = [&](uint64_t Kn, uint32_t strm_n, bool first_of_step) -> Prepared {
        Prepared q;
        q.act = act_next;
        const uint2 e = e_next;                                               // 4 x (site << 2 | pauli), 0 = no site
        fetch_entry();
        const uint32_t ent[4] = {e.x & 0xFFFFu, e.x >> 16, e.y & 0xFFFFu, e.y >> 16};
        q.ops = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t qb = ent[i] >> 2;
            q.wi[i] = qb >> 4;
            q.sh[i] = (qb & 15u) * 2u;
            q.xv[i] = (ent[i] & 3u) << q.sh[i];
            q.ops |= (ent[i] & 3u) << (2 * i);                                // (an unused entry reads site 0 into both and cancels)
        }
        // (a member's block serves four consecutive phases of a step: drawn at the step's first phase -- the stream is the step's -- and
        // whenever the phase index enters a new group of four)
        if (first_of_step || (Kn & 3u) == 0) ub = philox_block(Kn >> 2, 8u + (uint32_t)lane, syn, strm_n, a.seed_lo, a.seed_hi);
        const uint32_t u = sel4(ub, (int)(Kn & 3u));
        q.u = u;
        // mcmc.py:42 / :30 with the coin: accept iff dE <= dmax
        q.dmax = acc_all ? ((u >> 31) != 0u ? 127 : -127) : (int)(u <= thr1) + (int)(u <= thr2) + (int)(u <= thr3) + (int)(u <= thr4);
        return q;
    };
    fetch_entry();
    Prepared nxt = prepare(a.prop0, top_logical ? slot : kDiagStream + dg, true);
    for (uint64_t t = 0; t < a.nsteps; ++t) {
        QECMC_CSTAMP(5); QECMC_CSTAMP(0);
        uint32_t *sb = st + sid * W;
        bool recount = false;
        // the rung's Philox stream at this step: the slot's own for the top rule, the diagonal one otherwise (philox.hpp)
        const uint32_t strm = top_logical ? slot : kDiagStream + dg;
        dg = dg + 1u == (uint32_t)NC ? 0u : dg + 1u;
        [[maybe_unused]] u32x4 topb{0, 0, 0, 0};
        int dn = 0;                                                           // this lane's accepted dE of the step
        [[maybe_unused]] bool any_acc = false, any_lane = false;              // RULE == 2: a move was accepted this step (wave-uniform / this lane's)
        for (uint32_t j = 0; j < iters; ++j) {
            const uint64_t K = a.prop0 + t * iters + j;
            if (top_logical) {
                // the top rule's blocks (K, 0) are wave-uniform: lane l draws the one of phase j + l, 64 phases at a time, and the phase
                // that needs it reads that lane -- one Philox evaluation per step instead of one per phase on the ladder's longest wave
                if ((j & 63u) == 0) topb = philox_block(K + (uint64_t)lane, 0, syn, strm, a.seed_lo, a.seed_hi);
                const int jl = (int)(j & 63u);
                const u32x4 x{(uint32_t)__builtin_amdgcn_readlane((int)topb.x, jl), (uint32_t)__builtin_amdgcn_readlane((int)topb.y, jl),
                              (uint32_t)__builtin_amdgcn_readlane((int)topb.z, jl), (uint32_t)__builtin_amdgcn_readlane((int)topb.w, jl)};
                if (x.x <= thrL1) {
                    const uint32_t *m0 = lmask + L * W, *m1 = m0, *m2 = m0, *m3 = m0;    // identity rows
                    uint32_t cdelta;
                    if constexpr (CODE == kCodeToric) {
                        const uint32_t op0 = x.y >> 30, op1 = x.z >> 30;
                        const uint32_t dx0 = (op0 ^ (op0 >> 1)) & 1u, dz0 = op0 >> 1, dx1 = (op1 ^ (op1 >> 1)) & 1u, dz1 = op1 >> 1;
                        if (dx0) m0 = lmask + scale_low30(x.y, L) * W;
                        if (dz0) m1 = lmask + LW + scale_u16(x.w >> 16, L) * W;
                        if (dx1) m2 = lmask + 2 * LW + scale_low30(x.z, L) * W;
                        if (dz1) m3 = lmask + 3 * LW + scale_u16(x.w & 0xFFFFu, L) * W;
                        cdelta = (L & 1) ? (dx0 | (dz0 << 1) | (dx1 << 2) | (dz1 << 3)) : 0u;
                    } else {
                        const uint32_t op = x.y >> 30;
                        const uint32_t xp = ((op ^ (op >> 1)) & 1u) ? scale_low30(x.y, L) : 0u, zp = (op >> 1) ? scale_u16(x.w >> 16, L) : 0u;
                        const uint32_t ax = CODE == kCodeXzzx ? ((op ^ (op >> 1)) & 1u) : (op & 1u), az = op >> 1;
                        if (ax) m0 = lmask + xp * W;
                        if (az) m1 = lmask + LW + zp * W;
                        cdelta = ax | (az << 1);
                    }
                    bool take = true;
                    if constexpr (RULE == 1) {
                        // the biased top rung tests the operator like every move (mcmc_biased.py:32-46): u < w(new) / w(old), the power
                        // tables' products in the reference's order; u = word 0 of block (K, 1)
                        int ox = 0, oz = 0, oxy = 0, qx = 0, qz = 0, qxy = 0;
                        for (int w = lane; w < W; w += 64) {
                            const uint32_t xo = sb[w], xn = xo ^ m0[w] ^ m1[w] ^ m2[w] ^ m3[w];
                            ox += __popc(xo & ~(xo >> 1) & 0x55555555u); oz += __popc(xo & (xo >> 1) & 0x55555555u); oxy += __popc((xo ^ (xo >> 1)) & 0x55555555u);
                            qx += __popc(xn & ~(xn >> 1) & 0x55555555u); qz += __popc(xn & (xn >> 1) & 0x55555555u); qxy += __popc((xn ^ (xn >> 1)) & 0x55555555u);
                        }
                        ox = wave_sum(ox); oz = wave_sum(oz); oxy = wave_sum(oxy); qx = wave_sum(qx); qz = wave_sum(qz); qxy = wave_sum(qxy);
                        const int T1 = nq + 1;
                        const double *bt = a.bias_tbl + (size_t)slot * 4 * T1;
                        const double wn = bt[qx] * bt[T1 + (qxy - qx)] * bt[2 * T1 + qz] * bt[3 * T1 + (nq - qxy - qz)];
                        const double wo = bt[ox] * bt[T1 + (oxy - ox)] * bt[2 * T1 + oz] * bt[3 * T1 + (nq - oxy - oz)];
                        const u32x4 ab = philox_block(K, 1u, syn, strm, a.seed_lo, a.seed_hi);
                        take = (double)ab.x * (1.0 / 4294967296.0) < wn / wo;
                        take = __builtin_amdgcn_readfirstlane((int)take) != 0;
                    }
                    if (take) {
                        for (int w = lane; w < W; w += 64) sb[w] ^= m0[w] ^ m1[w] ^ m2[w] ^ m3[w];
                        cls ^= cdelta;
                        recount = true;
                        any_acc = true;
                    }
                }
            }
            // ---- one phase: every lane its generator (prepared during the phase before)
            const Prepared cu = nxt;
            {
                // the next phase: of this step, or the first of the next one (whose stream is the next diagonal)
                const bool last = j + 1u == iters;
                nxt = prepare(K + 1u, top_logical ? slot : (last ? kDiagStream + dg : strm), last);
            }
            uint32_t *ad[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) ad[i] = sb + cu.wi[i];
            int dE = 0;
            bool acc;
            if (acc_all) {
                // a rung with f >= 1 is blind: the coin decides, the state is not read, and the error count is taken again at the step's end
                acc = cu.act && cu.dmax > 0;
                recount = true;
            } else {
                uint32_t F = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) F |= ((*ad[i] >> cu.sh[i]) & 3u) << (2 * i);
                const uint32_t G = F ^ cu.ops;
                dE = (int)__popc((G | (G >> 1)) & 0x55u) - (int)__popc((F | (F >> 1)) & 0x55u);            // toric_model.py:275-282
                if constexpr (RULE != 0) {
                    // the model's own weight ratio for this generator: the changes of n_z (fields = 3) and n_x + n_y (fields 1, 2)
                    const int dz = (int)__popc(G & (G >> 1) & 0x55u) - (int)__popc(F & (F >> 1) & 0x55u);
                    const int dxy = (int)__popc((G ^ (G >> 1)) & 0x55u) - (int)__popc((F ^ (F >> 1)) & 0x55u);
                    acc = cu.act && cu.u <= mythr[9 * (dz + 4) + (dxy + 4)];
                } else
                acc = cu.act && dE <= cu.dmax;
            }
            if (acc) {
#pragma unroll
                for (int i = 0; i < 4; ++i) lds_xor(ad[i], cu.xv[i]);                   // (same-word updates of different lanes: LDS atomics)
            }
            dn += acc ? dE : 0;                                                 // (the rule never reads n inside a step: summed over the wave once, below)
            if constexpr (RULE == 2) any_lane |= acc;
        }
        QECMC_CSTAMP(1);
        if (recount) {                                                        // the logical operators moved O(L) sites
            int c = 0;
            for (int w = lane; w < W; w += 64) c += (int)nnz2(sb[w]);
            n = (uint32_t)__builtin_amdgcn_readfirstlane(wave_sum(c));
        } else {
            n = (uint32_t)((int)n + __builtin_amdgcn_readfirstlane(wave_sum(dn)));
        }
        // ---- Ladder.step's swap sweep (mcmc.py:96-103): records and uniforms out, one barrier, every wave replays the cascade
        uint32_t *cur = rec + (t & 1) * NC, *sx = swu + (t & 1) * NC;
        if (lane == 0) cur[slot] = pack_info(n, sid, cls, flag);
        if constexpr (RULE == 2) {
            // the slot's attribute follows its chain's counts if a move was accepted this step (mcmc_alpha.py:58,:70)
            if (any_acc || __any(any_lane)) nef = counts_zxy(sb);
            if (lane == 0) nefr[(t & 1) * NC + slot] = nef;
        }
        if ((int)slot * 4 < NC - 1) {
            const u32x4 b = philox_block(a.step0 + t, slot, syn, kSwapStream, a.seed_lo, a.seed_hi);
            if (lane < 4 && (int)slot * 4 + lane < NC - 1) sx[slot * 4 + lane] = sel4(b, lane);
        }
        QECMC_CSTAMP(2);
        __syncthreads();
        QECMC_CSTAMP(3);
        if constexpr (CONV) { if (stopf[t & 1]) break; }                                 // (set by wave 0 one step earlier: uniform for the workgroup)
        // The cascade (mcmc.py:96-99) carries one record down the rungs: which one depends on every decision above.  Its decisions do
        // not: the record carried into rung pair i is one of those of slots i+1 .. NC-1, so lane (c, i) tests "record c against slot i" for
        // every pair at once (one threshold look-up each, NC^2 <= 256 tests in at most four passes) and the serial part walks a bit table
        // with scalar instructions -- a lone workgroup has nobody to hide seven dependent look-ups behind.
        uint64_t fm[4] = {0, 0, 0, 0};
        const uint32_t rec_l = cur[lane < NC ? lane : 0];                              // lane l: slot l's record (read back by the walk's result)
        [[maybe_unused]] uint64_t fa = 0;                                              // RULE == 2: bit i = rung pair i flips (the attributes stay put: Q4)
        if constexpr (RULE == 2) {
            const uint32_t *ne = nefr + (t & 1) * NC;
            const int i = lane < NC - 1 ? lane : 0;
            fa = __ballot(lane < NC - 1 && alpha_flip(sx[i], ne[i + 1], ne[i], a.alpha, a.alpha_lnb[i]));
        } else
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const int c0 = ch * 64;
            if (c0 >= NC * NC) break;
            const int c = c0 + lane, ca = c / NC, ci = c - ca * NC;
            const bool valid = ca < NC && ci < NC - 1 && ca > ci;
            const uint32_t hi = cur[valid ? ca : 0], lo = cur[valid ? ci : 0], xi = sx[valid ? ci : 0];
            const int d = (int)(hi & 0xFFFFu) - (int)(lo & 0xFFFFu);                   // ne_hi - ne_lo, _r_flip mcmc.py:146-149
            const int e = (valid ? ci : 0) * (nq + 1) + (d > 0 ? d : 0);
            const bool flip = d <= 0 || (swap32 ? xi < swt[e] : (uint64_t)xi < (((uint64_t)swt[2 * e + 1] << 32) | swt[2 * e]));
            fm[ch] = __ballot(valid && flip);
        }
        int carried = NC - 1, mine_s = NC - 1;                                         // slots whose step-t records are carried / end up here
        for (int i = NC - 2; i >= 0; --i) {
            const int b = carried * NC + i;
            const uint64_t w = b < 64 ? fm[0] : b < 128 ? fm[1] : b < 192 ? fm[2] : fm[3];
            const bool flip = RULE == 2 ? ((fa >> i) & 1ull) != 0 : ((w >> (b & 63)) & 1ull) != 0;
            const int into = flip ? i : carried;                                       // what slot i+1 now holds (:98-99)
            carried = flip ? carried : i;
            if ((int)slot == i + 1) mine_s = into;
        }
        if (slot == 0) mine_s = carried;
        uint32_t mine = (uint32_t)__builtin_amdgcn_readlane((int)rec_l, mine_s);
        mine = (uint32_t)__builtin_amdgcn_readfirstlane((int)mine);
        n = mine & 0xFFFFu; sid = (mine >> 16) & 0xFFu; cls = (mine >> 24) & 0x3Fu; flag = mine >> 31;
        if ((int)slot == NC - 1) flag = 1;                                               // mcmc.py:100
        QECMC_CSTAMP(4);
        if (slot == 0 && !done) {
            tops0 += (NC == 1) | flag;                                                   // :101-102
            if (a.counts != nullptr && tops0 >= a.tops_burn) {                           // decoders.py:60-67
                if (lane == 0) hist[CODE == kCodeXzzx ? (cls ^ (cls >> 1)) : cls] += 1;
                samples++;
                if constexpr (CONV) {
                    // nbr_errors_bottom_chain[since_burn] = count_errors (:68): logged in HBM, series index i in row burn + i; the three
                    // entries that leave / enter the windows are independent loads (one round trip on wave 0's path per step)
                    const size_t lN = (size_t)a.N;
                    const uint32_t l = samples, lo1 = l - 1;
                    const uint32_t a0 = lo1 >> 2, b0 = lo1 >> 1, c0 = (3u * lo1) >> 2, a1 = l >> 2, b1 = l >> 1, c1 = (3u * l) >> 2;
                    if constexpr (RULE == 2) {
                        // chains[0].n_eff (decoders_biasednoise.py:204): slot 0's attribute -- this wave's -- logged as its two counts
                        uint32_t *mylog = reinterpret_cast<uint32_t *>(a.nlog) + ladder;
                        mylog[(size_t)t * lN] = nef;
                        sumB += nef & 0xFFFFu; sumBxy += nef >> 16;
                        if (c1 != c0) { const uint32_t v = mylog[(size_t)(burn + c0) * lN]; sumB -= v & 0xFFFFu; sumBxy -= v >> 16; }
                        if (b1 != b0) { const uint32_t v = mylog[(size_t)(burn + b0) * lN]; sumA += v & 0xFFFFu; sumAxy += v >> 16; }
                        if (a1 != a0) { const uint32_t v = mylog[(size_t)(burn + a0) * lN]; sumA -= v & 0xFFFFu; sumAxy -= v >> 16; }
                    } else {
                    uint16_t *mylog = a.nlog + ladder;
                    mylog[(size_t)t * lN] = (uint16_t)n;
                    sumB += n;
                    if (c1 != c0) sumB -= mylog[(size_t)(burn + c0) * lN];
                    if (b1 != b0) sumA += mylog[(size_t)(burn + b0) * lN];
                    if (a1 != a0) sumA -= mylog[(size_t)(burn + a0) * lN];
                    }
                }
            } else {
                burn++;                                                                  // resulting_burn_in, :71
            }
            if (!t_reached && tops0 >= a.TOPS) t_reached = (uint32_t)t + 1u;
            if constexpr (CONV) {
                if (tops0 >= a.TOPS) {                                                   // :74
                    const uint32_t l = samples ? samples : 1u;
                    const uint32_t den2 = (l >> 1) - (l >> 2), den4 = l - ((3u * l) >> 2);
                    bool accept = false;                                                 // empty slice -> nan -> not accepted
                    if (samples && den2 && den4) {
                        if constexpr (RULE == 2) accept = alpha_series_close(sumA, sumAxy, den2, sumB, sumBxy, den4, a.alpha, a.eps);   // decoders_biasednoise.py:229-238
                        else accept = fabs((double)sumA / (double)den2 - (double)sumB / (double)den4) < a.eps;   // :96-102
                    }
                    if (accept) {
                        if (conv_streak >= a.SEQ) { done = 1; steps_done = (uint32_t)t + 1u; }   // :77-78
                        else conv_streak = tops0 - conv_start;                           // :79
                    } else {
                        conv_streak = 0;                                                 // :81-82
                        conv_start = tops0;
                    }
                }
                if (done && lane == 0) stopf[(t + 1) & 1] = 1;
            }
        }
        if (slot == 0) flag = 0;                                                         // :103
    }
    __syncthreads();
    // ---- results
    uint32_t *fin = rec;                                                                 // every wave's final record, for the state dump
    if (lane == 0) fin[slot] = pack_info(n, sid, cls, flag);
    __syncthreads();
    if (slot == 0) {
        if (a.counts != nullptr)
            for (int c = lane; c < ncls; c += 64) {
                if (R > 1) { if (hist[c]) atomicAdd(a.counts + row * ncls + c, hist[c]); }
                else a.counts[row * ncls + c] = hist[c];
            }
        if (lane == 0) {
            // steps_done / converged: the criterion's; without it, the first step with tops0 >= TOPS
            const uint32_t sd = CONV ? (done ? steps_done : (uint32_t)a.nsteps) : (t_reached ? t_reached : (uint32_t)a.nsteps);
            const bool reached = CONV ? done != 0 : t_reached != 0;
            if (R > 1) {
                if (a.samples != nullptr) atomicAdd(a.samples + row, samples);
                if (a.tops0 != nullptr) atomicAdd(a.tops0 + row, tops0);
                if (a.steps_done != nullptr) atomicMax(a.steps_done + row, sd);
                if (a.converged != nullptr && !reached) a.converged[row] = 0;
            } else {
                if (a.samples != nullptr) a.samples[row] = samples;
                if (a.tops0 != nullptr) a.tops0[row] = tops0;
                if (a.steps_done != nullptr) a.steps_done[row] = sd;
                if (a.converged != nullptr) a.converged[row] = reached;
            }
        }
    }
    if (a.write_states && a.states != nullptr) {
        uint8_t *dst = a.states + (ladder * NC + slot) * (uint64_t)nq;                   // slot order
        const uint32_t sidc = (fin[slot] >> 16) & 0xFFu;
        for (int q = lane; q < nq; q += 64) dst[q] = (uint8_t)((st[sidc * W + (q >> 4)] >> ((q & 15) * 2)) & 3u);
    }
    if (a.flags != nullptr && lane == 0) a.flags[ladder * NC + slot] = (uint8_t)(fin[slot] >> 31);
}

hipError_t launch_ladder_colour(const LadderArgs &a, hipStream_t stream)
{
    if (a.phase_tab == nullptr || a.n_phases == 0 || a.noise < 0 || a.noise > 2 || a.resume) return hipErrorInvalidValue;
    if (a.conv_mode != 0 && a.nlog == nullptr) return hipErrorInvalidValue;
    if (a.noise != 0 && (a.col_thr == nullptr || a.bias_tbl == nullptr || (a.code != kCodeXzzx && a.code != kCodeRotated) || (a.noise == 2 && a.alpha_lnb == nullptr)))
        return hipErrorInvalidValue;
    const bool conv = a.conv_mode != 0;
    // (development knob: -DQECMC_COLOUR_SMALL_MINW=6|8 builds ladders of up to 8 rungs for 512 threads at that many waves per SIMD -- three / four
    // workgroups per CU instead of two.  Measured, profiles/r04_colour_occupancy_ab.json: +15 ... 28 % ladder steps per second from 1 024 ladders on,
    // -8 % at 256 and below (20-160 B of scratch on a lone workgroup's path): the layout exists for few syndromes, so the default stays 4.)
#ifndef QECMC_COLOUR_SMALL_MINW
#define QECMC_COLOUR_SMALL_MINW 4
#endif
#if QECMC_COLOUR_SMALL_MINW != 4
    const bool small = a.Nc <= 8;
#define QECMC_KR(code, rule) (small ? (conv ? (const void *)ladder_colour_kernel<code, true, rule, 512, QECMC_COLOUR_SMALL_MINW> : (const void *)ladder_colour_kernel<code, false, rule, 512, QECMC_COLOUR_SMALL_MINW>) \
                                    : (conv ? (const void *)ladder_colour_kernel<code, true, rule> : (const void *)ladder_colour_kernel<code, false, rule>))
#else
#define QECMC_KR(code, rule) (conv ? (const void *)ladder_colour_kernel<code, true, rule> : (const void *)ladder_colour_kernel<code, false, rule>)
#endif
#define QECMC_KC(code) QECMC_KR(code, 0)
    const void *fn = a.noise == 1 ? (a.code == kCodeXzzx ? QECMC_KR(kCodeXzzx, 1) : QECMC_KR(kCodeRotated, 1))
                   : a.noise == 2 ? (a.code == kCodeXzzx ? QECMC_KR(kCodeXzzx, 2) : QECMC_KR(kCodeRotated, 2))
                   : a.code == kCodeToric ? QECMC_KC(kCodeToric) : a.code == kCodeXzzx ? QECMC_KC(kCodeXzzx)
                   : a.code == kCodeRotated ? QECMC_KC(kCodeRotated) : a.code == kCodePlanar ? QECMC_KC(kCodePlanar) : nullptr;
#undef QECMC_KC
#undef QECMC_KR
    if (!fn) return hipErrorInvalidValue;
    const size_t lds = sizeof(uint32_t) * colour_lds_dwords(a.Nc, a.W, a.ncls, a.n_phases, a.n_gen, a.L, a.nq, a.swap_fast_ok != 0, a.noise);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    void *kargs[] = {const_cast<LadderArgs *>(&a)};
    hipError_t e = hipLaunchKernel(fn, dim3((unsigned)a.N), dim3((unsigned)a.Nc * 64u), kargs, lds, stream);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

}  // namespace qecmc
