/*
 * qecmc.h -- C-ABI of the MI355X-native MCMC equivalence-class sampler.
 *
 * This is the drop-in boundary for ONE hot path of
 * QEC-project-2020/MCMC-QEC-toric-RL: the Metropolis / parallel-tempering
 * sampler of src/mcmc.py, the code-model stencils it calls
 * (src/toric_model.py ...) and its caller decoders.PTEQ.  The reference has no
 * FFI of its own (it is pure Python + numba); the boundary below is what a
 * `ctypes.CDLL` binding on the reference side would load (INTEGRATION.md shows
 * that binding).  Every entry point cites the reference interface it replaces
 * (file:line relative to the reference tree).
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every buffer and the
 *     library keeps no pointer after a call returns (plans excepted);
 *   - return 0 on success, a negative qecmc_status on error with a message in
 *     qecmc_last_error() (thread-local); no C++ exception crosses the ABI;
 *   - Pauli encoding 0=I 1=X 2=Y 3=Z, composition = XOR (toric_model.py:277);
 *     toric state = uint8[2][L][L] C-order (toric_model.py:12), nq = 2*L*L; xzzx / rotated state = uint8[L][L];
 *     planar state = uint8[2][L][L] with layer 1 on its first L-1 rows / columns (planar_model.py:14,38-39);
 *   - every compute entry point runs on the GPU (HIP, gfx950).  There is no
 *     CPU fallback: without a device the call fails with QECMC_ERR_NO_DEVICE.
 *   - `_dev` entry points take DEVICE pointers and a hipStream_t (as void*),
 *     enqueue asynchronously and allocate nothing; the others take HOST
 *     pointers and do H2D + kernel + D2H + synchronise themselves.
 */
#ifndef QECMC_H
#define QECMC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QECMC_ABI_VERSION 4

typedef enum qecmc_status {
    QECMC_OK = 0,
    QECMC_ERR_INVALID = -1,     /* bad argument (message says which) */
    QECMC_ERR_NO_DEVICE = -2,   /* no HIP device / device index out of range */
    QECMC_ERR_HIP = -3,         /* a HIP runtime call failed */
    QECMC_ERR_UNSUPPORTED = -4  /* valid request this build has no kernel for */
} qecmc_status;

typedef enum qecmc_code { QECMC_TORIC = 0, QECMC_XZZX = 1, QECMC_ROTATED = 2, QECMC_PLANAR = 3 } qecmc_code;
/* SWEEP: systematic sweep over the generators in table order (proposal k tests generator k mod G); no colouring is needed
 * because a lane owns a whole chain.  Same stationary law, not the reference's chain. */
/* COLOUR: the latency layout -- one workgroup per ladder, one wavefront per rung, the lanes of a wavefront = the generators of
 * one colour phase (mutually disjoint, so their Metropolis tests are independent), proposed at once: a sweep is n_phases wavefront
 * passes instead of G sequential proposals.  A systematic scan like SWEEP (same stationary law, not the reference's chain);
 * `iters` counts phases; fixed-length runs report the first step with tops0 >= TOPS in steps_done / converged,
 * conv_mode error_based runs the reference's criterion.  Under the biased and alpha rules (xzzx / rotated codes) every
 * generator is a Metropolis move for the noise model's own weight px^nx py^ny pz^nz pI^nI -- the reference's rule at iters = 1: the members
 * of a phase are tested at once, so the p_b that mcmc_biased.py:28-31 / mcmc_alpha.py:38-41 freeze per update_chain call cannot be
 * carried --; the biased top rung tests its logical operators, Ladder_alpha's top rung (pz_tilde = 1) takes the coin. */
/* WAVE: the reference's random scan (src/mcmc.py:19-43) with ONE generator pick per proposal shared by the 64 ladders of a wavefront
 * (global ladder indices that agree above bit 6), every ladder keeping its own acceptance uniform.  toric_model.py:287-296 picks the
 * generator independently of the state and syndromes never interact, so each ladder's chain has exactly the reference's law -- unlike
 * SWEEP / COLOUR this IS the reference's Markov chain per syndrome; only the noise of different syndromes is correlated.  What it buys:
 * a proposal's sites are wave-uniform, so the rungs' states live in registers.  Depolarizing rule with a top rung at
 * p = 0.75 (Nc >= 2), at most 16 packed state words per rung (toric / planar L <= 11, xzzx / rotated L <= 16) -- fixed-length runs of
 * up to 8 rungs also 17 .. 32 words (toric L <= 16, xzzx / rotated L <= 22) --, and the alpha rule (xzzx / rotated L <= 11);
 * first_syndrome a multiple of 64, 1 <= iters <= 128.  With conv_mode error_based the launch runs on a persistent grid
 * whose workgroups own contiguous shares of the batch and reuse the lane of a stopped ladder for the next one of their share: the
 * generator picks then belong to the lane's position in the grid, so results are reproducible for a given (batch size, grid,
 * first_syndrome) and equal to the one-ladder-per-lane layout whenever the batch fits the grid; no final states in that mode. */
typedef enum qecmc_scan { QECMC_SCAN_RANDOM = 0, QECMC_SCAN_SWEEP = 1, QECMC_SCAN_COLOUR = 2, QECMC_SCAN_WAVE = 3 } qecmc_scan;
typedef enum qecmc_noise { QECMC_NOISE_DEPOLARIZING = 0, QECMC_NOISE_BIASED = 1, QECMC_NOISE_ALPHA = 2 } qecmc_noise;
typedef enum qecmc_conv { QECMC_CONV_NONE = 0, QECMC_CONV_ERROR_BASED = 1 } qecmc_conv;

/* Mirrors the keyword arguments of decoders.PTEQ (decoders.py:25) plus what the
 * batched GPU call needs.  Set abi_size = sizeof(qecmc_params). */
typedef struct qecmc_params {
    uint32_t abi_size;
    int32_t  code;          /* qecmc_code */
    int32_t  L;             /* system_size (toric_model.py:10).  The ladder calls keep every rung's state, the generator table and the
                               acceptance tables in one workgroup's LDS, and refuse (QECMC_ERR_UNSUPPORTED, with the numbers) what does not fit:
                               Nc * ceil(nq / 16) * 256 B of states + tables <= 160 KiB (toric L = 20 at Nc = 8, L = 31 at Nc = 2), at most 2048
                               generators (toric L <= 32, xzzx / rotated L <= 45), nq <= 511 for the biased / alpha rules (L <= 21) */
    int32_t  Nc;            /* number of chains in the ladder (decoders.py:30; Q7: default L) */
    int32_t  noise;         /* qecmc_noise */
    int32_t  scan;          /* qecmc_scan: RANDOM = the reference's chain (src/mcmc.py:19-43) */
    int32_t  conv_mode;     /* qecmc_conv (decoders.py:74) */
    int32_t  device;        /* HIP device ordinal */
    uint64_t iters;         /* proposals per chain between swap sweeps (decoders.py:25 iters=10) */
    uint64_t steps;         /* ladder steps (decoders.py:25 steps) */
    int32_t  tops_burn;     /* decoders.py:63 */
    int32_t  TOPS;          /* decoders.py:74 */
    int32_t  SEQ;           /* decoders.py:78 */
    int32_t  replicas;      /* 0 or 1: one ladder per syndrome.  R > 1: R independent ladders per syndrome (Philox syndrome
                               index first_syndrome + s*R + r), class counts / samples / tops0 summed over the R ladders on the
                               device -- the reference's "droplets" pattern (decoders.py:215-225) for small batches */
    double   eps;           /* decoders.py:102 */
    double   p;             /* bottom-chain error rate (mcmc.py:50 p_bottom); pz_tilde_bottom for alpha noise (mcmc_alpha.py:76) */
    double   eta;           /* bias (mcmc_biased.py:11); unused otherwise */
    double   alpha;         /* alpha noise exponent (mcmc_alpha.py:11, decoders_biasednoise.py:175); unused otherwise */
    double   p_logical;     /* top-chain logical proposal rate (decoders.py:52 passes 0.5); the ladder kernels' random scan selects with
                               16 bits, i.e. acts as ceil(p_logical * 2^16) / 2^16 (exact for 0.5; the CPU oracle does the same) */
    uint64_t seed;          /* Philox key */
    uint32_t first_syndrome;/* global index of syndrome 0 of this call: results do not
                               depend on how a batch is sharded over GPUs */
    uint32_t flags;         /* 0 in production.  Developer switches (results are identical with any value; they select among kernel
                               variants that compute the same thing): QECMC_FLAG_* below in bits 0-15, and in bits 16-31 the size of
                               the work-queue kernels' persistent grid in workgroups (0: what fits the chip; the tests force 1-2
                               workgroups so that every lane runs several ladders) */
} qecmc_params;

enum qecmc_flag {
    QECMC_FLAG_NO_PRE   = 2,   /* no instantiations that draw the top chain's Philox blocks ahead */
    QECMC_FLAG_NO_DELUT = 4,   /* no dE look-up table (popcount form) */
    QECMC_FLAG_NO_SSW   = 8    /* no swap sweep run once by wave 0 (every wave replays the cascade) */
};
#define QECMC_FLAGS_QUEUE_GRID(n) ((uint32_t)(n) << 16)

typedef struct qecmc_stats {
    uint64_t proposals;     /* Metropolis trials executed (all chains, all syndromes) */
    uint64_t swap_tests;
    double   kernel_ms;     /* HIP-event time of the sampler kernel(s) */
    double   total_ms;      /* wall time incl. H2D/D2H for host-pointer calls */
} qecmc_stats;

int         qecmc_abi_version(void);
const char *qecmc_last_error(void);
int         qecmc_device_count(void);      /* 0 when no GPU is visible */

/* ---- stencil primitives (batched; N states per call, host pointers) -------
 * Each runs the same __device__ stencil code the sampler kernels use. */

/* Toric_code.apply_stabilizer -> _apply_stabilizer, toric_model.py:40-41,256-284 (xzzx_model.py:360-436,
 * rotated_surface_model.py:349-392: operator 1 = plaquette (row,col), 3 = half plaquette `row` on side `col`;
 * planar_model.py:292-339: operator 1 at (row < L-1, col < L), operator 3 at (row < L, col < L-1)).
 * out[i] = in[i] with stabilizer (rows[i], cols[i], ops[i]) applied; dE[i] = error-count change. */
int qecmc_apply_stabilizer(int code, int L, uint64_t N, const uint8_t *in, uint8_t *out,
                           const int32_t *rows, const int32_t *cols, const int32_t *ops, int32_t *dE);
/* _apply_logical, toric_model.py:179-225 (layer is ignored by non-toric codes). */
int qecmc_apply_logical(int code, int L, uint64_t N, const uint8_t *in, uint8_t *out,
                        const int32_t *ops, const int32_t *layers, const int32_t *xpos,
                        const int32_t *zpos, int32_t *dE);
/* Toric_code.count_errors -> _count_errors, toric_model.py:33-34,174-176. */
int qecmc_count_errors(int code, int L, uint64_t N, const uint8_t *in, int64_t *n);
/* define_equivalence_class, toric_model.py:52-53,317-351. */
int qecmc_eq_class(int code, int L, uint64_t N, const uint8_t *in, int32_t *cls);
/* to_class, toric_model.py:55-56,354-377. */
int qecmc_to_class(int code, int L, uint64_t N, const uint8_t *in, uint8_t *out, const int32_t *eq);
/* Toric_code.syndrom, toric_model.py:58-101: defects_out uint8[N][2][L][L]; xzzx_code.syndrome /
 * RotSurCode.syndrome (xzzx_model.py:60-83): defects_out uint8[N][L+1][L+1] (plaquette_defects); Planar_code.syndrom
 * (planar_model.py:134-153): defects_out uint8[N][2 L (L-1)] = vertex_defects [L-1][L] then plaquette_defects [L][L-1]. */
int qecmc_syndrome(int code, int L, uint64_t N, const uint8_t *in, uint8_t *defects_out);

/* ---- syndrome generation (the data-generation recipe of generate_data.py:57-60,110-131, batched) --------------------
 * N random error chains -- Toric_code.generate_random_error(p) (toric_model.py:15-23; pass p_x = p_y = p_z = p / 3) or
 * generate_random_error(p_x, p_y, p_z) of the xzzx / rotated / planar models (xzzx_model.py:16-30, planar_model.py:18-40) --
 * then, if hide_class != 0, one apply_random_logical on each (generate_data.py:131), all on the device.
 * init_out uint8[N][nq] = the seed configurations for the decoder; raw_out (nullable) = the errors themselves
 * (generate_data.py:120 keeps them); eq_true_out (nullable) int32[N] = their equivalence class (:121-122), the decoding target.
 * Syndrome s draws from Philox (seed, first_syndrome + s): independent of how a data set is cut into batches.
 * The _dev form takes device pointers and a hipStream_t, allocates nothing and does not synchronise. */
int qecmc_generate_syndromes(int code, int L, uint64_t N, double p_x, double p_y, double p_z, int hide_class,
                             uint64_t seed, uint32_t first_syndrome, uint8_t *init_out, uint8_t *raw_out,
                             int32_t *eq_true_out);
int qecmc_generate_syndromes_dev(int code, int L, uint64_t N, double p_x, double p_y, double p_z, int hide_class,
                                 uint64_t seed, uint32_t first_syndrome, void *d_init_out, void *d_raw_out,
                                 void *d_eq_true_out, void *hip_stream);

/* ---- chain / ladder on caller-owned state (host pointers) ----------------- */

/* Chain.update_chain(iters), src/mcmc.py:19-43, on N independent chains.
 * Chain i draws from Philox stream (syndrome = first_syndrome+i, slot, proposals
 * k0 .. k0+iters-1).  p_logical != 0 selects the top-chain branch (mcmc.py:20). */
int qecmc_chain_update(int code, int L, uint64_t N, uint8_t *states_inout, double p, double p_logical,
                       uint64_t iters, uint64_t seed, uint32_t first_syndrome, uint32_t slot, uint64_t k0);

/* Chain_biased.update_chain(iters), src/mcmc_biased.py:20-59 (acceptance pn/pb from the full
 * (nx,ny,nz) counts, pb frozen at loop entry as in the reference, quirk Q3). */
int qecmc_chain_update_biased(int code, int L, uint64_t N, uint8_t *states_inout, double p, double eta,
                              double p_logical, uint64_t iters, uint64_t seed, uint32_t first_syndrome,
                              uint32_t slot, uint64_t k0);

/* Chain_alpha.update_chain(iters), src/mcmc_alpha.py:27-70: the biased rule with (p_x, p_y, p_z) derived from
 * (pz_tilde, alpha).  accepted_out uint8[N] (nullable) = 1 iff chain i accepted at least one proposal -- the
 * reference refreshes the chain's n_eff attribute on accepted moves only (:58,:70). */
int qecmc_chain_update_alpha(int code, int L, uint64_t N, uint8_t *states_inout, double pz_tilde, double alpha,
                             double p_logical, uint64_t iters, uint64_t seed, uint32_t first_syndrome,
                             uint32_t slot, uint64_t k0, uint8_t *accepted_out);

/* Chain_xyz.update_chain_fast(iters), src/mcmc.py:106-114,162-173 (what decoders.py:352,442 sample STDC_general_noise with): a
 * generator proposal accepted with probability prod_i (p_i / (1 - p_x - p_y - p_z))^(change of n_i), i = x, y, z; p_xyz double[3]. */
int qecmc_chain_update_xyz(int code, int L, uint64_t N, uint8_t *states_inout, const double *p_xyz, uint64_t iters,
                           uint64_t seed, uint32_t first_syndrome, uint32_t slot, uint64_t k0);

/* Ladder.step(iters) x nsteps, src/mcmc.py:94-103, on N ladders in slot order.
 * states uint8[N][Nc][nq], flags uint8[N][Nc], tops0 uint32[N]; step0 / prop0 =
 * ladder steps / proposals per slot already done (Philox addressing: at ladder step T the top chain draws from
 * stream Nc-1, the chain on a lower slot c from the "diagonal" stream 0x400 + (c + T) mod Nc).  Uses
 * params->{code,L,Nc,p,p_logical,seed,first_syndrome,device}. */
int qecmc_ladder_step(const qecmc_params *params, uint64_t N, uint8_t *states_inout, uint8_t *flags_inout,
                      uint32_t *tops0_inout, uint64_t iters, uint64_t nsteps, uint64_t step0, uint64_t prop0);

/* Ladder_alpha.step(iters) x nsteps, src/mcmc_alpha.py:127-137 (params->noise = QECMC_NOISE_ALPHA, params->p =
 * pz_tilde_bottom, params->alpha).  neff_counts_inout uint16[N][Nc][2] carries each SLOT's n_eff attribute as the
 * pair (n_z, n_x + n_y) it was last computed from (n_eff = n_z + alpha (n_x + n_y), :58): the reference swaps codes
 * and flags but not n_eff, so the attribute can lag the slot's configuration and is state the
 * caller must carry between calls. */
int qecmc_ladder_step_alpha(const qecmc_params *params, uint64_t N, uint8_t *states_inout, uint8_t *flags_inout,
                            uint32_t *tops0_inout, uint16_t *neff_counts_inout, uint64_t iters, uint64_t nsteps,
                            uint64_t step0, uint64_t prop0);

/* ---- the batched hot path: decoders.PTEQ (decoders.py:25-89) on N syndromes */

/* Host-pointer form.  init uint8[N][nq] (one seed configuration per syndrome,
 * generate_data.py:131-138); counts_out uint32[N][ncls] = eq[since_burn]
 * (decoders.py:66-67); samples_out uint32[N] = since_burn+1 or 0 when the burn-in
 * never ended (A10 "burn-in trap"); tops0_out uint32[N] (nullable);
 * steps_done_out uint32[N] (nullable) = ladder steps run until the convergence
 * criterion fired (decoders.py:74-82), or params->steps; converged_out uint8[N]
 * (nullable); final_states_out uint8[N][Nc][nq] in slot order (nullable; only
 * meaningful with conv_mode NONE); stats nullable.
 * conv_mode ERROR_BASED needs device workspace for the per-step bottom-chain error counts the criterion averages
 * (decoders.py:68,93-105): 2*N*steps bytes (alpha noise: 4, the two counts behind n_eff), or -- for the ladders that run on a
 * persistent grid with a work queue (a lane whose ladder has stopped takes the next one of the batch in place: the depolarizing
 * random-scan ladders of every code with a top rung at p = 0.75, toric L <= 16 and the others L <= 32, and the biased / alpha rules
 * on the xzzx / rotated codes) -- the same per min(N, grid*64) columns.  A plan's queue counter is its own: one launch of a
 * conv_mode plan at a time. */
int qecmc_pteq_batch(const qecmc_params *params, const uint8_t *init, uint64_t N, uint32_t *counts_out,
                     uint32_t *samples_out, uint32_t *tops0_out, uint32_t *steps_done_out,
                     uint8_t *converged_out, uint8_t *final_states_out, qecmc_stats *stats_out);

/* PTDC (decoders.py:168-233, conv_mult = 0) on N syndromes: the direct-counting estimator.  For every syndrome and
 * class, `droplets` independent ladders WITHOUT logical moves (decoders.py:182,196) run params->steps ladder steps of
 * params->iters proposals at p = params->p (p_sampling); after every step every rung's configuration goes into the
 * (syndrome, class) set of chains seen so far (PTDC_droplet, decoders.py:146-152; the droplets' sets are merged,
 * :220-226).  init uint8[N][ncls][nq] = one representative per class (what the reference's list form of init_code /
 * to_class provides); hist_out uint32[N][ncls][nq+1] = N(n), the number of distinct chains of each length found.
 * The caller forms Z_E = sum_n N(n) exp(-beta n) with beta from p_error (:208,229-233).  PTDC itself passes
 * steps // Nc (:201).  Ladder l = (s * ncls + c) * droplets + d draws from Philox syndrome first_syndrome + l.
 * Uses params->{code,L,Nc,p,iters,steps,seed,first_syndrome,device}; p_logical is ignored (0).
 * STDC / STDC_droplet (decoders.py:236-322) is the same computation on 1-chain ladders: Nc = 1, iters = 5
 * (`chain.update_chain_fast(5)`, :250).
 * flags: QECMC_PTDC_INIT_PER_DROPLET -- init is uint8[N][ncls][droplets][nq], a start of its own for every droplet
 *        (STDC's "rain", `apply_stabilizers_uniform` per droplet, :246-247);
 *        QECMC_PTDC_SET_PER_RUNG -- one set per (ladder, rung) instead of per (syndrome, class): PTRC_droplet's
 *        per-rung dictionaries (decoders.py:584-631); the outputs are then uint32[N][ncls][droplets][Nc][nq+1].
 * m_out (nullable, shaped like hist_out): m(n), the number of OBSERVATIONS of chains of length n (the len_counts of
 * PTRC_droplet :606-618 and STRC_droplet :768-776), from which the host forms the STRC / PTRC estimates. */
enum { QECMC_PTDC_INIT_PER_DROPLET = 1, QECMC_PTDC_SET_PER_RUNG = 2 };
int qecmc_ptdc_batch(const qecmc_params *params, const uint8_t *init, uint64_t N, int32_t droplets,
                     uint32_t flags, uint32_t *hist_out, uint32_t *m_out, qecmc_stats *stats_out);

/* The same with the `conv_mult` early stop of PTDC_droplet (decoders.py:153-162), STDC_droplet (:256-262) and STRC_droplet
 * (:783-826): whenever a ladder step finds a chain that is new to the DROPLET's own dictionary and no longer than the
 * shortest it has seen (initially 2 L^2, :140), stop = step * conv_mult; the droplet records nothing after the first step
 * with step >= stop and step * 100 >= params->steps.  conv_mult = 0 is qecmc_ptdc_batch.  With QECMC_PTDC_SET_PER_RUNG
 * conv_mult is ignored, as PTRC_droplet's stop is commented out in the reference (:627-630).
 * steps_done_out (nullable) uint32[N][ncls][droplets]: ladder steps each droplet recorded.  A workgroup of 64 droplets
 * leaves the step loop once all of them have stopped. */
int qecmc_ptdc_batch_conv(const qecmc_params *params, const uint8_t *init, uint64_t N, int32_t droplets,
                          uint32_t flags, double conv_mult, uint32_t *hist_out, uint32_t *m_out,
                          uint32_t *steps_done_out, qecmc_stats *stats_out);

/* The same sampling for the general-noise estimators STDC_general_noise / STDC_general_noise_shortest (decoders.py:345-507)
 * and STDC_Nall_n_alpha's weighting (:537-581): xyz_out (nullable) uint32[N][ncls][params->steps * Nc * droplets] receives,
 * for every (syndrome, class) set, n_x | n_y << 10 | n_z << 20 of each DISTINCT chain (count_errors_xyz,
 * planar_model.py:225-229 -- the values of STDC_droplet_general_noise's dict, decoders.py:339-340), in no particular
 * order, the unused tail 0xFFFFFFFF; xyz_count_out (nullable) uint32[N][ncls] their number.  Not with
 * QECMC_PTDC_SET_PER_RUNG.
 * p_xyz_sampling (nullable) double[3]: sample with Chain_xyz (src/mcmc.py:106-114,162-173) instead of Chain -- a single
 * chain (params->Nc must be 1; params->p is ignored) whose proposals are accepted with probability
 * prod_i (p_i / (1 - sum p))^(change of n_i); planar, xzzx and rotated codes (the reference's runs the planar stencil).
 * params->noise = QECMC_NOISE_ALPHA (with params->alpha, params->p = pz_tilde_sampling, Nc = 1, iters = 5) samples with
 * Chain_alpha instead: STDC_droplet_alpha / STDC_Nall_n_alpha (decoders.py:510-581), whose weights
 * n_z + alpha (n_x + n_y) the caller forms from xyz_out. */
int qecmc_ptdc_batch_xyz(const qecmc_params *params, const uint8_t *init, uint64_t N, int32_t droplets,
                         uint32_t flags, double conv_mult, const double *p_xyz_sampling, uint32_t *hist_out,
                         uint32_t *m_out, uint32_t *steps_done_out, uint32_t *xyz_out, uint32_t *xyz_count_out,
                         qecmc_stats *stats_out);

/* Plan + device-pointer form: build once (validates, uploads threshold tables),
 * then launch asynchronously on a caller stream with buffers already in HBM.
 * d_workspace / workspace_bytes: the criterion runs' log (decoders.py:68: nbr_errors_bottom_chain), qecmc_plan_workspace_bytes() bytes
 * -- 0 for conv_mode NONE, then NULL / 0 is fine.  ABI 4: the size travels with the pointer and a buffer smaller than the launch
 * needs is refused (QECMC_ERR_INVALID) before anything is enqueued; the library has ONE formula for it (capi.hip workspace_need). */
typedef struct qecmc_plan qecmc_plan;
int qecmc_plan_create(const qecmc_params *params, qecmc_plan **plan_out);
int qecmc_plan_destroy(qecmc_plan *plan);
/* the workspace a launch of N syndromes needs: with_final_states != 0 for a launch that passes d_final_states (one log column per ladder:
 * 2 N steps bytes, 4 for alpha noise); 0 for one that does not -- the plans with a work queue then log one column per lane of their
 * persistent grid, however large the batch */
int qecmc_plan_workspace_bytes(const qecmc_plan *plan, uint64_t N, int with_final_states, uint64_t *bytes_out);
int qecmc_pteq_launch_dev(qecmc_plan *plan, const void *d_init, uint64_t N, uint32_t first_syndrome,
                          void *d_counts, void *d_samples, void *d_tops0 /*nullable*/,
                          void *d_steps_done /*nullable*/, void *d_converged /*nullable*/,
                          void *d_final_states /*nullable*/, void *d_workspace /*nullable*/, uint64_t workspace_bytes,
                          void *hip_stream);
/* Optional per-syndrome equilibrium observables of the following launches of `plan` (device pointers, either nullable;
 * NULL, NULL switches them off again):
 *   d_swap_accepts uint32[N][Nc-1]: accepted swap tests of rung pair (i, i+1) (Ladder.step's r_flip, src/mcmc.py:96-99);
 *   d_nerr_sums    uint32[N][Nc]:   sum over the ladder steps of count_errors() of the chain in rung c after the step's
 *                                   swaps (what mcmc.py:88-89 reads) -- divide by steps for the time average.
 * Both count every ladder step of the launch (no burn-in) and need nq * steps < 2^32.  Not with replicas > 1. */
int qecmc_plan_set_stats(qecmc_plan *plan, void *d_swap_accepts, void *d_nerr_sums);
/* Continue N ladders for params->steps more ladder steps from caller-held device state (exact continuation: Philox is
 * addressed by step0 = ladder steps already done): d_states uint8[N][Nc][nq] slot order in/out, d_flags uint8[N][Nc] in/out,
 * d_tops0 uint32[N] in/out; d_counts uint32[N][ncls] and d_samples uint32[N] are ADDED to (decoders.py:60-67 bookkeeping
 * continues).  A fresh ladder is d_states[s][c] = init[s] for every rung c, d_flags[s][c] = (c == Nc-1), d_tops0[s] = 0,
 * step0 = 0 (Ladder.__init__, mcmc.py:72-77); chunked runs then reproduce one long run bit for bit.  conv_mode must be NONE. */
int qecmc_pteq_resume_dev(qecmc_plan *plan, void *d_states, void *d_flags, void *d_tops0, uint64_t N,
                          uint32_t first_syndrome, uint64_t step0, void *d_counts, void *d_samples, void *hip_stream);
/* Host-pointer qecmc_pteq_batch with the observables of qecmc_plan_set_stats (swap_accepts_out uint32[N][Nc-1],
 * nerr_sums_out uint32[N][Nc]; either nullable). */
int qecmc_pteq_batch_stats(const qecmc_params *params, const uint8_t *init, uint64_t N, uint32_t *counts_out,
                           uint32_t *samples_out, uint32_t *tops0_out, uint32_t *steps_done_out,
                           uint8_t *converged_out, uint8_t *final_states_out, uint32_t *swap_accepts_out,
                           uint32_t *nerr_sums_out, qecmc_stats *stats_out);
/* Bytes of dynamic LDS and threads per workgroup the plan's kernel uses (for reports). */
int qecmc_plan_info(const qecmc_plan *plan, uint32_t *lds_bytes, uint32_t *block_threads,
                    uint32_t *syndromes_per_block);

#ifdef __cplusplus
}
#endif
#endif
