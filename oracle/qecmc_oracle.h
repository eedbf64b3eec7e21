/*
 * qecmc_oracle.h -- CPU oracle for the MCMC equivalence-class sampler hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * algorithm (QEC-project-2020/MCMC-QEC-toric-RL: src/toric_model.py,
 * src/mcmc.py, decoders.py:25-105), used as the checker in tests/, in
 * __graft_entry__.smoke() and as bench.py's `cpu_baseline` leg.  Nothing in
 * the product package (mcmc-qec-toric-rl_amd/) may include, link or call it.
 *
 * Parity status: PINNED.  Every function below is checked against vectors
 * captured from the reference itself (imported in the build container with an
 * identity-decorator numba stub, see tests/golden/gen_golden.py):
 *   F1  deterministic known-answer vectors for all stencil functions,
 *   F2  stream-injected exact trajectories of Chain.update_chain / Ladder.step
 *       / the PTEQ bookkeeping loop (uniform stream injected into `random`),
 *   F4  the BASELINE config-1 plumbing vector.
 *
 * Pauli encoding 0=I 1=X 2=Y 3=Z, composition = XOR (toric_model.py:277).
 * Toric state: uint8[2][L][L] C-order (toric_model.py:12).
 */
#ifndef QECMC_ORACLE_H
#define QECMC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- uniform source -------------------------------------------------------
 * mode 0: injected stream of doubles, consumed strictly in reference call
 *         order (this is how F2 pins the draw order of src/mcmc.py:19-43).
 * mode 1: Philox4x32-10 counter RNG, u = word * 2^-32, with the
 *         (stream, k, sub, word) addressing documented in DESIGN.md §RNG.
 */
typedef struct orc_rng {
    int mode;
    const double *stream;
    uint64_t pos, len;
    uint64_t consumed;       /* number of uniforms drawn (both modes) */
    uint64_t seed;
    uint32_t syndrome;       /* global syndrome index (Philox ctr[2]) */
    /* block cache for mode 1: entry 0, and entry 1 for the acceptance-refinement blocks (sub 4) */
    uint32_t c_stream, c_sub;
    uint64_t c_k;
    int c_valid;
    uint32_t c_w[4];
    uint32_t c2_stream, c2_sub;
    uint64_t c2_k;
    int c2_valid;
    uint32_t c2_w[4];
    /* scan = 3 work-queue runs: the generator picks are addressed by the lane's position (group, workgroup step = ladder step + t0) */
    int wave_override;
    uint32_t wave_group;
    uint64_t wave_t0;
} orc_rng;

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void orc_rng_init_stream(orc_rng *r, const double *stream, uint64_t len);
void orc_rng_init_philox(orc_rng *r, uint64_t seed, uint32_t syndrome);

enum { ORC_TORIC = 0, ORC_XZZX = 1, ORC_ROTATED = 2, ORC_PLANAR = 3 };
enum { ORC_NOISE_DEPOLARIZING = 0, ORC_NOISE_BIASED = 1, ORC_NOISE_ALPHA = 2, ORC_NOISE_XYZ = 3 };

/* which code model / acceptance rule a chain uses (duck typing in the reference) */
typedef struct orc_model {
    int code;      /* ORC_TORIC: uint8[2][L][L], 16 classes; ORC_XZZX / ORC_ROTATED: uint8[L][L], 4 classes;
                      ORC_PLANAR: uint8[2][L][L] (layer 1 uses its first L-1 rows / columns), 4 classes */
    int L;
    int noise;     /* ORC_NOISE_DEPOLARIZING: src/mcmc.py; ORC_NOISE_BIASED: src/mcmc_biased.py */
    double eta;    /* bias, mcmc_biased.py:11 */
    double alpha;  /* "alpha" noise exponent, mcmc_alpha.py:11 (noise = ORC_NOISE_ALPHA; the ladder variable is pz_tilde) */
    int det_pow;   /* alpha swap test: 0 = libm pow as the reference (mcmc_alpha.py:123); 1 = the deterministic
                      exp(e*ln(base)) the GPU uses (same decision unless u falls within ~1e-16 of the threshold) */
    int scan;      /* 0: the reference's random scan; 1: systematic sweep over the generators (NOT the reference's
                      chain: the deterministic-scan variant the GPU offers as scan=1, same stationary law); 2: the same idea one
                      colour phase -- a set of mutually disjoint generators -- at a time (the GPU's latency layout, scan=2);
                      3: the reference's random scan again, with a generator pick shared by the 64 ladders of a GPU wavefront (scan=3:
                      every ladder keeps the reference's law; chain_update_wave in qecmc_oracle.c states the Philox addressing) */
    double pxyz[3]; /* ORC_NOISE_XYZ: Chain_xyz's (p_x, p_y, p_z), src/mcmc.py:106-114 -- a single chain without logical
                      moves whose proposals are accepted with prod_i (p_i / (1 - sum p))^(change of n_i), :162-173 */
} orc_model;

int orc_nq(int code, int L);
int orc_ncls(int code);
int orc_eq_class(int code, int L, const uint8_t *m);

/* ---- XZZX / rotated stencils (src/xzzx_model.py, src/rotated_surface_model.py) ---- */
int  orc_surf_generator(int code, int L, int row, int col, int op, int sites[4], int paulis[4]);
int  orc_surf_apply_stabilizer(int code, int L, const uint8_t *in, uint8_t *out, int row, int col, int op);
int  orc_surf_apply_logical(int code, int L, const uint8_t *in, uint8_t *out, int op, int xpos, int zpos);
int  orc_surf_eq_class(int code, int L, const uint8_t *m);
void orc_surf_syndrome(int code, int L, const uint8_t *in, uint8_t *defects /*[L+1][L+1]*/);
/* number of generators, and generator g in the order the sweep / the one-word pick use (-> row, col, op) */
int  orc_colour_phases(int code, int L, int *tab_out, int cap);   /* scan = 2: [n_phases][64] generator indices, -1 = none; returns n_phases */
int  orc_surf_ngen(int code, int L);
void orc_surf_gen_rco(int code, int L, int g, int *row, int *col, int *op);
/* Planar_code.syndrom, planar_model.py:134-153: vertex_defects uint8[L-1][L] then plaquette_defects uint8[L][L-1] */
void orc_planar_syndrome(int L, const uint8_t *in, uint8_t *vertex, uint8_t *plaquette);

/* ---- toric stencils (src/toric_model.py) -------------------------------- */
int     orc_toric_apply_stabilizer(int L, const uint8_t *in, uint8_t *out, int row, int col, int op);
int     orc_toric_apply_logical(int L, const uint8_t *in, uint8_t *out, int op, int layer, int xpos, int zpos);
int64_t orc_count_errors(size_t nq, const uint8_t *in);
int     orc_toric_eq_class(int L, const uint8_t *in);
void    orc_toric_to_class(int L, const uint8_t *in, uint8_t *out, int eq);
void    orc_toric_syndrome(int L, const uint8_t *in, uint8_t *defects_out);

/* ---- chain / ladder / PTEQ (src/mcmc.py, decoders.py) -------------------- */
typedef struct orc_ladder {
    orc_model model;
    int L, Nc, nq;
    double p_logical;
    double *p_ladder;    /* [Nc]   */
    double *p_diff;      /* [Nc-1] */
    uint8_t *states;     /* [Nc][nq], slot order (chains[i].code.qubit_matrix) */
    uint8_t *flags;      /* [Nc] */
    double *n_eff;       /* [Nc] Chain_alpha.n_eff: stays with the SLOT when codes are swapped (quirk Q4) */
    uint32_t *n_eff_cnt; /* [Nc][2] the (n_z, n_x + n_y) each n_eff was formed from */
    uint64_t tops0;
    uint64_t step_index; /* ladder steps done so far (Philox addressing) */
    uint8_t *scratch;    /* [nq] */
    /* equilibrium observables (fixture F5): accepted swap tests per rung pair, and the sum over the steps of every
     * rung's count_errors() after the step's swaps */
    uint64_t *swap_acc;  /* [Nc-1] */
    uint64_t *nerr_sum;  /* [Nc] */
} orc_ladder;

/* Chain.update_chain(iters), src/mcmc.py:19-43.  `slot`/`k0` only address the
 * Philox stream (mode 1): proposal j of this call is proposal k0+j of `slot`. */
void orc_toric_chain_update(int L, uint8_t *state, double p, double p_logical,
                            uint64_t iters, orc_rng *rng, uint32_t slot, uint64_t k0,
                            uint8_t *scratch);

orc_ladder *orc_toric_ladder_new(int L, const uint8_t *init, double p_bottom, int Nc, double p_logical);
void        orc_ladder_free(orc_ladder *ld);
void        orc_toric_ladder_step(orc_ladder *ld, uint64_t iters, orc_rng *rng);

typedef struct orc_pteq_result {
    uint32_t counts[16];   /* eq[since_burn]   (decoders.py:42,66-67) */
    uint64_t samples;      /* since_burn + 1 if anything was recorded, else 0 */
    uint64_t tops0;
    uint64_t steps_done;
    int converged;
    uint8_t percent[16];   /* decoders.py:89 */
} orc_pteq_result;

/* decoders.PTEQ (decoders.py:25-89).  conv_mode 0 = conv_criteria None (fixed
 * `steps`), 1 = 'error_based'.  final_states (nullable) receives [Nc][nq]. */
void orc_toric_pteq(int L, const uint8_t *init, double p, int Nc, int SEQ, int TOPS, int tops_burn,
                    double eps, uint64_t steps, uint64_t iters, int conv_mode, orc_rng *rng,
                    orc_pteq_result *res, uint8_t *final_states);

/* scan = 3 with the error_based criterion on a persistent grid of `grid` workgroups (the GPU's deterministic work queue, restated in
 * qecmc_oracle.c): counts_out [N][16], the other outputs [N] */
void orc_pteq_wave_queue(const orc_model *m, const uint8_t *init, uint64_t N, uint32_t first_syndrome, double p, int Nc, int SEQ,
                         int TOPS, int tops_burn, double eps, uint64_t steps, uint64_t iters, uint64_t seed, uint32_t grid, int n_threads,
                         uint32_t *counts_out, uint64_t *samples_out, uint64_t *tops0_out, uint64_t *steps_done_out, uint8_t *converged_out);

/* code / noise generic forms of the above */
void orc_chain_update(const orc_model *m, uint8_t *state, double p, double p_logical, uint64_t iters,
                      orc_rng *rng, uint32_t slot, uint64_t k0, uint8_t *scratch);
orc_ladder *orc_ladder_new(const orc_model *m, const uint8_t *init, double p_bottom, int Nc, double p_logical);
void orc_ladder_step(orc_ladder *ld, uint64_t iters, orc_rng *rng);
void orc_pteq(const orc_model *m, const uint8_t *init, double p, int Nc, int SEQ, int TOPS, int tops_burn,
              double eps, uint64_t steps, uint64_t iters, int conv_mode, orc_rng *rng,
              orc_pteq_result *res, uint8_t *final_states);
void orc_pteq_batch(const orc_model *m, const uint8_t *init, uint64_t N, uint32_t first_syndrome, double p,
                    int Nc, int SEQ, int TOPS, int tops_burn, double eps, uint64_t steps,
                    uint64_t iters, int conv_mode, uint64_t seed, int n_threads,
                    uint32_t *counts_out /*[N][16]*/, uint64_t *samples_out, uint64_t *tops0_out,
                    uint64_t *steps_done_out /*nullable*/, uint8_t *converged_out /*nullable*/,
                    uint8_t *final_states /*nullable*/);

/* Chain_alpha.update_chain (mcmc_alpha.py:27-70); *n_eff is the chain's n_eff attribute (updated on accepted moves only) */
int orc_chain_update_alpha(const orc_model *m, uint8_t *state, double pz_tilde, double p_logical, uint64_t iters,
                            orc_rng *rng, uint32_t slot, uint64_t k0, uint8_t *scratch, double *n_eff);
double orc_det_exp(double y);

/* ---- unique-chain estimators (decoders.py:138-233, PTDC) ----
 * 64-bit key of a configuration (FNV-1a over the bytes, never 0) -- the role of hash(qubit_matrix.tobytes()), decoders.py:148 */
uint64_t orc_state_key(const uint8_t *state, size_t nq);
/* open-addressing set of keys: tab[cap] (cap a power of two, 0 = empty); returns 1 if the key was new */
int orc_uset_insert(uint64_t *tab, uint64_t cap, uint64_t key);
/* PTDC_droplet (decoders.py:138-164, conv_mult = 0): `steps` x Ladder.step(iters) on a ladder WITHOUT logical moves
 * (PTDC builds Ladder(p_sampling, code, Nc), p_logical = 0), every rung's configuration recorded after every step;
 * hist[n] += 1 for every configuration not seen before in `tab` (its length n = count_errors).
 * per_rung != 0 is PTRC_droplet (decoders.py:584-631): one set per rung -- tab = [Nc][cap], hist = [Nc][nq+1].
 * mhist (nullable, same shape as hist): m(n), EVERY observation of a chain of length n (the len_counts of :606-618 / STRC_droplet :768-776). */
/* STDC_droplet_general_noise (decoders.py:325-342) / STDC_droplet_alpha (:510-534) keep, for every distinct chain, its
 * (n_x, n_y, n_z) (count_errors_xyz, planar_model.py:225-229; chain_lengths, xzzx_model.py:39-43): a sink (nullable) that
 * receives n_x | n_y << 10 | n_z << 20 of every chain that is new to the set `tab`, in the order found. */
typedef struct { uint32_t *out; uint64_t n; } orc_xyz_sink;
void orc_ptdc_droplet(const orc_model *m, const uint8_t *init, double p_sampling, int Nc, uint64_t steps, uint64_t iters,
                      orc_rng *rng, uint64_t *tab, uint64_t cap, uint32_t *hist /*[nq+1]*/, int per_rung, uint32_t *mhist,
                      double conv_mult, uint64_t *steps_done, orc_xyz_sink *xyz);
/* conv_mult != 0: the early stop of PTDC_droplet / STDC_droplet / STRC_droplet (decoders.py:153-162, :256-262, :783-826):
 * whenever a chain that is new to the droplet's own dictionary has length <= the shortest seen so far, stop = step * conv_mult;
 * sampling ends after the first step with step >= stop and step * 100 >= steps.  (PTRC_droplet's stop is commented out in
 * the reference, :627-630.)  steps_done (nullable): the number of steps run. */
/* N syndromes x ncls class representatives x D droplets; droplet ladders of one (syndrome, class) share a set (the dict
 * merge of decoders.py:220-226).  Ladder l = (s * ncls + c) * D + d draws from Philox syndrome first_syndrome + l.
 * init uint8[N][ncls][nq]; hist_out uint32[N][ncls][nq+1] = number of unique chains of each length, N(n). */
void orc_ptdc_batch(const orc_model *m, const uint8_t *init, uint64_t N, int ncls, int D, int init_per_droplet,
                    uint32_t first_syndrome, double p_sampling, int Nc, uint64_t steps, uint64_t iters, uint64_t seed,
                    int n_threads, uint32_t *hist_out, int per_rung, uint32_t *mhist_out, double conv_mult,
                    uint32_t *xyz_out /* nullable: [N*ncls][steps*Nc*D], unused tail 0xFFFFFFFF; not with per_rung */);
/* init_per_droplet: init is [N][ncls][D][nq] (STDC's rain).  per_rung = 0: one set per (syndrome, class), outputs
 * [N][ncls][nq+1]; per_rung != 0 (PTRC): one set per (ladder, rung), outputs [N][ncls][D][Nc][nq+1].  mhist_out nullable. */   /* deterministic exp for y <= 0 (IEEE +,*,fma only): bit-identical on CPU and GPU */

/* N independent PTEQ runs (one per syndrome, Philox keyed by first_syndrome+i),
 * spread over `n_threads` OpenMP threads.  This is the timed CPU baseline. */
void orc_toric_pteq_batch(int L, const uint8_t *init /*[N][nq]*/, uint64_t N, uint32_t first_syndrome,
                          double p, int Nc, int tops_burn, uint64_t steps, uint64_t iters,
                          uint64_t seed, int n_threads,
                          uint32_t *counts_out /*[N][16]*/, uint64_t *samples_out /*[N]*/,
                          uint64_t *tops0_out /*[N]*/, uint8_t *final_states /*nullable [N][Nc][nq]*/);

/* As above with the convergence criterion of decoders.py:74-82,93-105 (conv_mode 1). */
void orc_toric_pteq_batch_conv(int L, const uint8_t *init, uint64_t N, uint32_t first_syndrome, double p,
                               int Nc, int SEQ, int TOPS, int tops_burn, double eps, uint64_t steps,
                               uint64_t iters, int conv_mode, uint64_t seed, int n_threads,
                               uint32_t *counts_out, uint64_t *samples_out, uint64_t *tops0_out,
                               uint64_t *steps_done_out /*nullable*/, uint8_t *converged_out /*nullable*/,
                               uint8_t *final_states /*nullable*/);

#ifdef __cplusplus
}
#endif
/* Syndrome generation in Philox mode (the checker of qecmc_generate_syndromes): error chains by the models'
 * generate_random_error, their equivalence class, and one apply_random_logical on top (generate_data.py:110-131). */
void orc_generate_syndromes(int code, int L, uint64_t N, double p_x, double p_y, double p_z, int hide, uint64_t seed,
                            uint32_t first_syndrome, uint8_t *init_out, uint8_t *raw_out, int32_t *eq_true_out);

#endif
