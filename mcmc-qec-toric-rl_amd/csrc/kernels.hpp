// Internal interface between the C-ABI layer (capi.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qecmc {

constexpr int kMaxNc = 16;          // one wavefront per ladder slot, <= 1024 threads per workgroup
constexpr int kSynPerBlock = 64;    // one lane per syndrome
constexpr int kSwapFast = 64;       // swap-threshold entries per rung pair kept in LDS (d < kSwapFast)

// Arguments of the random-scan ladder kernel (passed by value; fits the kernarg segment).
struct LadderArgs {
    // inputs / outputs (device pointers)
    const uint8_t *init;      // [N][nq]           fresh start (resume == 0)
    uint8_t *states;          // [N][Nc][nq]       resume input and/or final-state output (nullable)
    uint8_t *flags;           // [N][Nc]           resume in/out (nullable)
    uint32_t *tops0;          // [N]               resume in / out (nullable)
    uint32_t *counts;         // [N][ncls]         out (nullable)
    uint32_t *samples;        // [N]               out (nullable)
    uint32_t *steps_done;     // [N]               out (nullable): ladder steps until convergence (or nsteps)
    uint8_t *converged;       // [N]               out (nullable)
    uint16_t *nlog;           // [nsteps][N]       bottom-chain error counts (conv_mode workspace)
    const uint64_t *swap_thr; // [Nc-1][nq+1]      ceil(p_diff[i]^d * 2^32)
    const uint32_t *lmask;    // [4][L+1][W]       logical-operator XOR masks (row L = identity)
    const uint2 *gen;         // [(L-1)^2 + 2(L-1)]  xzzx / rotated generators: 4 x u16 (site << 2 | pauli), 0 = unused
    const double *bias_tbl;   // [Nc][4][nq+1]      px^n, py^n, pz^n, pI^n per rung (biased and alpha noise)
    // unique-chain set of the direct-counting estimators (PTDC_droplet, decoders.py:146-152), filled in-kernel after every
    // ladder step when uset_tab != nullptr: set of (ladder, rung) = ladder * Nc + rung if uset_per_rung, else ladder / uset_D
    unsigned long long *uset_tab;   // [sets][uset_cap] keys, 0 = empty
    uint64_t uset_cap;              // power of two
    uint32_t *uset_hist;            // [sets][nq+1] N(n): distinct chains by length
    uint32_t *uset_mhist;           // [sets][nq+1] m(n): observations by length (nullable)
    uint32_t uset_D;                // droplets per set
    int uset_per_rung;
    // conv_mult early stop (decoders.py:153-162): 0 = off.  uset_own (nullable): one private set per ladder, needed when a
    // class set is shared by several droplets (the stop looks at the droplet's own dictionary)
    double uset_conv_mult;
    unsigned long long *uset_own;   // [N][uset_own_cap]
    uint64_t uset_own_cap;
    // (n_x, n_y, n_z) of every distinct chain of a set, n_x | n_y << 10 | n_z << 20, appended in the order the insertions
    // land (STDC_droplet_general_noise's dict values, decoders.py:339-340); nullable
    uint32_t *uset_xyz;             // [sets][uset_xyz_stride]
    uint32_t *uset_xyz_cnt;         // [sets]
    uint64_t uset_xyz_stride;
    // Chain_xyz (mcmc.py:106-114,162-173; 1-chain ladders): accept iff v44 < xyz_thr[dx+4][dy+4][dz+4]; nullable
    const uint64_t *xyz_thr;        // [9][9][9]   ceil(w * 2^44): the 44-bit acceptance uniform of a non-top proposal
    // biased / alpha rules: the count change of a generator move from an LDS table, and the fast acceptance test
    const uint32_t *xyz_lut;  // [n_types][256]     dx + (dz << 10) + ((dx + dy) << 20) (wrapping) of applying a generator of Pauli pattern
                              //                    `type` to four sites holding the 2-bit fields of the index
    const uint8_t *gen_type;  // [n_gen]            Pauli-pattern id of every generator
    int n_types;              //                    distinct Pauli patterns among the generators (<= 16)
    uint8_t type_ops[16];     //                    ... each as four 2-bit Paulis (site 0 in bits 1:0; 0 = no site): the plaquette codes' dE table
    double bias_l2[kMaxNc][2];//                    log2(px / pI), log2(pz / pI) per rung (px = py in both noise models)
    float bias_l2f[kMaxNc][2];//                    ... rounded to single precision
    uint32_t bias_f32ok;      //                    bit c: rung c's fast test may use them (4 iters max|l| <= 2000, iters <= 512)
    const double *alpha_lnb;  // [Nc-1]             ln(pz_tilde[i] / pz_tilde[i+1]) (alpha noise, mcmc_alpha.py:123)
    double alpha;             //                    mcmc_alpha.py:11
    uint32_t *neff;           // [N][Nc]            alpha noise: the slots' n_eff attributes as n_z | (n_x+n_y) << 16; resume in / out
    int code, noise;          // qecmc_code, qecmc_noise
    int scan;                 // qecmc_scan
    const uint16_t *phase_tab;//                    scan = 2: [n_phases][64] generator indices of every colour phase, 0xFFFF = idle lane
    uint32_t n_phases;
    uint32_t n_gen;           // number of stabilizer generators G (sweep order = table order)
    const uint32_t *acc_tbl_top; // [nq+1]          ceil(f_top^dE * 2^32): top slot below p = 0.75 (Nc == 1 only)
    uint64_t N;
    uint64_t step0, prop0, nsteps;
    uint64_t thr_logical;     // ceil(p_logical * 2^32)
    uint32_t iters;
    uint32_t first_syndrome;
    uint32_t seed_lo, seed_hi;
    uint32_t tops_burn;
    uint32_t TOPS, SEQ;       // decoders.py:74,78
    double eps;               // decoders.py:102
    int conv_mode;            // 0 = fixed steps, 1 = error_based
    uint32_t acc_all_mask;    // bit c: slot c accepts every proposal (f >= 1, mcmc.py:30)
    uint32_t acc_thr[kMaxNc][4];   // ceil(f_c^dE * 2^32), dE = 1..4 (sweep mode: one 32-bit word per acceptance)
    uint64_t acc_thr44[kMaxNc][4]; // ceil(f_c^dE * 2^44): random scan, the 44-bit acceptance uniform of a non-top proposal
    const uint32_t *col_thr;  // [Nc][81]           scan = 2 under the biased / alpha rules: accept iff u <= col_thr[c][9 (dz + 4) + dxy + 4] (capi.hip)
    const uint32_t *wu_desc;       // scan = 3: [n_gen][16] generator descriptors (tables.hpp wave_descriptors)
    uint32_t wu_chunk;             // scan = 3, criterion runs on a persistent grid: ladders per workgroup (a multiple of 64; 0: one ladder per lane, no queue)
    float swap_inv_log2[kMaxNc];   // 1 / log2(p_diff[i]): first guess of the largest d with u < p_diff[i]^d (the table decides)
    int32_t swap_fast_ok;          // every swap threshold with d >= 1 fits 32 bits (false only if two rungs coincide)
    int L, Nc, W, nq, ncls;
#if defined(QECMC_TIMELINE) || defined(QECMC_STEPTRACE)
    uint64_t *dbg;            // [grid][4] diagnostic stamps (tools/timeline.hip, tools/steptrace.hip only)
#endif
    int resume;               // 0: replicate init into every slot (mcmc.py:72); 1: load states/flags/tops0
    int write_states;
    // replicas R >= 1: a.N counts LADDERS (syndromes x R); ladder l starts from init row l / R and adds its class counts, samples
    // and tops0 to the outputs of syndrome l / R (atomics; the caller zeroes them) -- decoders.py:215-225 "droplets"
    uint32_t *queue;          // QUEUE kernels: the batch's counter of ladders handed out after launch (zeroed by the caller)
    uint32_t grid_cap;        // ... the persistent grid: at most this many workgroups (0: one per 64 ladders)
    uint32_t tune;            // development knobs (qecmc_params.flags bits 0-15, include/qecmc.h qecmc_flag; 0 in production): bit 1 = no PRE instantiations, bit 2 = no dE table, bit 3 = no SSW instantiations
    uint32_t replicas;
    int accumulate;           // counts / samples are added to (qecmc_pteq_resume_dev)
    // equilibrium observables (qecmc_plan_set_stats; nullable): accepted swaps per rung pair, sum of error counts per rung
    uint32_t *swap_acc;       // [N][Nc-1]
    uint32_t *nerr_sum;       // [N][Nc]
};

size_t ladder_lds_bytes(int L, int Nc, int W, int ncls, int gen_dwords);
inline size_t ladder_stats_lds_bytes(int Nc) { return sizeof(uint32_t) * 64u * (size_t)(2 * Nc); }   // [Nc] swap accepts (row Nc-1 idle) + [Nc] error sums
constexpr uint32_t kMaxGenLds = 2048;   // generator tables up to this many entries are staged in LDS
// dwords of the LDS generator table: the toric random-scan kernels expand each generator to 4 x u32
// (byte offset << 16 | pauli fields | bit shift), the other paths keep the plan's 4 x u16 form.  Up to kGenSplit
// generators the expanded table is stored as two halves kGenSplit entries apart (sites 0,1 | sites 2,3).
// alpha noise appends the double-buffered n_eff records [2][Nc][64] to the region
constexpr int kLutTypes = 8;        // rows of the plaquette codes' dE look-up table (Pauli patterns of their generators; more: no table)
constexpr int kGenSplit = 255;      // ds_read2_b64's second offset is an 8-bit count of 8-byte units
// ... and the biased / alpha rules' count-change table uint2[n_types][256] and packed per-state counts uint32[Nc][64]
// lattice size of a plaquette code from its qubit count (xzzx / rotated: L x L; planar: 2 L^2 with an idle row and column)
inline int nq_L(int code, int nq) { int L = 1; while ((code == 3 ? 2 * L * L : L * L) < nq) ++L; return L; }
inline int ladder_gen_dwords(int code, int noise, int scan, uint32_t n_gen, int Nc, int nq = 0, int n_types = 0)
{
    // depolarizing random scan: the expanded table of the non-top proposal loop; the plaquette codes also keep the plan's
    // form for their top-chain / general paths
    const bool wide = !scan;                                   // (biased / alpha kernels: unsplit, next to the plan's form)
    // (toric, unsplit: 128 dwords behind the table for the dE look-up table, which the split form keeps between its halves)
    // (plaquette codes: 256 bytes per Pauli pattern behind their expanded table)
    const int lut_tail = noise ? 0 : code == 0 ? ((int)n_gen > kGenSplit ? 128 : 0) : 64 * kLutTypes;
    const int wide_dw = ((!noise && (int)n_gen <= kGenSplit) ? 2 * (kGenSplit + (int)n_gen) : 4 * (int)n_gen) + lut_tail;
    int d = wide ? (code == 0 ? wide_dw : ((2 * (int)n_gen + 3) & ~3) + wide_dw) : 2 * (int)n_gen;
    if (noise == 2) d = ((d + 3) & ~3) + 2 * Nc * 64;
    // (8 bytes per count-change entry; + the X / Z logical masks [2][L+1][W]; xzzx: + the logical operators' fields per generator)
    if (noise) d = ((d + 3) & ~3) + 512 * n_types + Nc * 64 + 2 * (nq_L(code, nq) + 1) * ((nq + 15) / 16) + (code == 1 ? (((int)n_gen + 1) & ~1) + 3 * 64 : 0);
    return d;
}
// the shapes the work-queue kernels exist for: depolarizing rule, random scan, error_based criterion, the framed top chain
// (toric L <= 16, plaquette codes L <= 32; a top rung at p = 0.75, i.e. Nc >= 2, with logical moves)
// ... and the biased / alpha rules on the xzzx / rotated codes (any L, Nc: no framed top chain there)
inline bool ladder_uses_queue(int code, int noise, int scan, int conv_mode, int L, int Nc, double p_logical)
{
    if (noise != 0) return scan == 0 && conv_mode != 0 && (code == 1 || code == 2);
    return scan == 0 && conv_mode != 0 && L <= (code == 0 ? 16 : 32) && Nc >= 2 && p_logical > 0.0;
}
// LDS of one workgroup (dwords): states, records, swap uniforms, histogram, then the tables every phase reads -- the phase
// table, the generator table, the logical masks, the swap thresholds (32-bit where they fit) -- so that the serial path of a
// step never waits for global memory
// (noise != 0: the rule's 81 thresholds per rung; the alpha rule's n_eff records by step parity)
inline size_t colour_lds_dwords(int Nc, int W, int ncls, uint32_t n_phases, uint32_t n_gen, int L, int nq, bool swap32, int noise = 0)
{
    return (size_t)Nc * W + 4 * (size_t)Nc + ncls + 32u * n_phases + 2u * n_gen + 4u * (L + 1) * W +
           (swap32 ? 1u : 2u) * (size_t)(Nc > 1 ? Nc - 1 : 0) * (nq + 1) + 2 + 4 +   // (+ 2: the stop flag by step parity)
           (noise ? (size_t)Nc * 81 + 2 * (size_t)Nc : 0);
}

hipError_t launch_ladder_rs_toric(const LadderArgs &a, hipStream_t stream);
// scan = 3 (ladder_wu.hip): what it is built for, and the LDS of one workgroup
bool wu_supported(const LadderArgs &a);
size_t wu_lds_bytes(int Nc, int W, int ncls, int L, bool conv, bool alpha);
constexpr int kWuAlphaQueueWaves = 6;   // waves per SIMD the alpha rule's criterion kernels of scan = 3 are built for (ladder_wu.hpp wu_pick_alpha): sizes their persistent grid

// byte-state primitive kernels (primitives.hip); all pointers are device pointers
hipError_t launch_apply_stabilizer(int code, int L, uint64_t N, const uint8_t *in, uint8_t *out, const int32_t *rows,
                                   const int32_t *cols, const int32_t *ops, int32_t *dE, hipStream_t s);
hipError_t launch_apply_logical(int code, int L, uint64_t N, const uint8_t *in, uint8_t *out, const int32_t *ops,
                                const int32_t *layers, const int32_t *xpos, const int32_t *zpos, int32_t *dE,
                                hipStream_t s);
hipError_t launch_count_errors(int nq, uint64_t N, const uint8_t *in, int64_t *n, hipStream_t s);
hipError_t launch_eq_class(int code, int L, uint64_t N, const uint8_t *in, int32_t *cls, hipStream_t s);
hipError_t launch_to_class(int L, uint64_t N, const uint8_t *in, uint8_t *out, const int32_t *eq, hipStream_t s);
hipError_t launch_syndrome(int code, int L, uint64_t N, const uint8_t *in, uint8_t *defects, hipStream_t s);

struct ChainArgs {
    uint8_t *states;          // [N][nq] in/out
    uint64_t N, iters, k0;
    uint64_t thr_logical;     // 0 => p_logical == 0 (non-top branch, mcmc.py:37)
    const uint32_t *acc_tbl;  // [nq+1] ceil(f^dE * 2^32) for dE >= 1 (entry 0 unused): top-chain acceptance words
    uint64_t acc44[5];        // ceil(f^dE * 2^44), dE = 1..4: the 44-bit acceptance uniform of non-top proposals
    uint32_t acc_all;         // f >= 1: every proposal is accepted (mcmc.py:30)
    uint32_t first_syndrome, slot, seed_lo, seed_hi;
    int L;
    int code;                 // 0 toric, 1 xzzx, 2 rotated
    int noise;                // 0 depolarizing (mcmc.py), 1 biased (mcmc_biased.py), 2 alpha (mcmc_alpha.py; same rule, other table)
    uint8_t *accepted;        // [N] out (nullable): 1 iff at least one proposal was accepted (Chain_alpha refreshes n_eff then)
    const double *bias_tbl;   // [4][nq+1] px^n, py^n, pz^n, pI^n (biased noise)
    const uint64_t *xyz_thr;  // [9][9][9] ceil(w 2^44), w = prod_i (p_i / (1 - sum p))^(change of n_i): Chain_xyz (mcmc.py:106-114,162-173); nullable
};
hipError_t launch_chain_update(const ChainArgs &a, hipStream_t s);

// device-side syndrome generation (primitives.hip)
constexpr uint32_t kGenStream = 0x200u;   // Philox stream id of the error / logical-operator draws of the generator
struct GenArgs {
    uint8_t *out;             // [N][nq] seed configurations (errors, then the random logical operator if `hide`)
    uint8_t *raw;             // [N][nq] the errors before the logical operator (nullable)
    int32_t *eq_true;         // [N] class of the raw errors (nullable)
    uint64_t N;
    uint64_t thr_z, thr_zx, thr_zxy;   // ceil(p_z 2^32), ceil((p_z + p_x) 2^32), ceil((p_z + p_x + p_y) 2^32); toric: thr_z = ceil(p 2^32)
    uint32_t first_syndrome, seed_lo, seed_hi;
    int code, L, nq, hide;
};
hipError_t launch_generate(const GenArgs &a, hipStream_t s);

}  // namespace qecmc
