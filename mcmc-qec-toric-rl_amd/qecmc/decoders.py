"""Host-side mirror of the reference's `PTEQ` (decoders.py:25-89) and the new
batched call `pteq_batch` that the hot path is built around."""
import ctypes as C

import numpy as np

from . import _lib as L_
from .mcmc import _code_id, _fresh_seed


def percent_from_counts(counts, samples):
    """decoders.py:89: (eq[since_burn] / (since_burn + 1) * 100).astype(uint8) --
    truncating, in float64, with an all-zero row when nothing was recorded (A10)."""
    counts = np.asarray(counts)
    den = np.maximum(np.asarray(samples, dtype=np.float64), 1.0)
    return (np.divide(counts, den[..., None] if counts.ndim > 1 else den) * 100).astype(np.uint8)


def pteq_batch(init, p, Nc=None, steps=1000, iters=10, tops_burn=2, p_logical=0.5, seed=0, first_syndrome=0,
               device=0, return_states=False, return_stats=False, conv_criteria=None, SEQ=2, TOPS=10, eps=0.1,
               code=L_.TORIC, eta=None, scan="random", alpha=None, replicas=1, return_swap_stats=False, flags=0):
    """decoders.PTEQ (decoders.py:25-89) on N syndromes at once.

    init: uint8[N, 2, L, L] (toric, code=PLANAR) or uint8[N, L, L] (code=XZZX / ROTATED) seed configurations, one per
    syndrome; eta selects the biased chain of src/mcmc_biased.py (PTEQ_biased), alpha the "alpha" noise ladder of
    src/mcmc_alpha.py (PTEQ_alpha; `p` is then pz_tilde); scan="random" is the reference's
    chain, scan="sweep" the systematic generator sweep (same stationary law), scan="colour" the latency layout for few
    syndromes: one workgroup per ladder, a whole colour phase of mutually disjoint generators per wavefront pass (`iters` counts
    phases; steps_done / converged report the first step with tops0 >= TOPS; depolarizing rule, fixed-length runs).  conv_criteria None runs exactly
    `steps` ladder steps; 'error_based' stops each syndrome by the reference's criterion (:74-105).
    replicas=R > 1 runs R independent ladders per syndrome (Philox index first_syndrome + s*R + r) and sums their class
    counts, samples and tops0 on the device (steps_done: the slowest ladder; converged: all of them) -- the droplets
    pattern of decoders.py:215-225, which is how a call with few syndromes fills the GPU; states are then [N*R, Nc, ...].
    return_swap_stats=True adds swap_accepts uint32[N,Nc-1] (accepted swap tests per rung pair over all `steps` ladder steps)
    and nerr_sums uint32[N,Nc] (sum over the steps of each rung's error count after the swaps), the per-batch mixing
    metrics of SURVEY.md 5.  flags: qecmc_params.flags (developer switches between equivalent kernel variants, _lib.dev_flags).
    Returns dict(counts uint32[N,16], samples uint32[N], tops0 uint32[N], steps_done uint32[N],
    converged bool[N], percent uint8[N,16] [, states uint8[N,Nc,2,L,L]] [, stats]).
    """
    if conv_criteria not in (None, 'error_based'):
        raise ValueError(f"conv_criteria={conv_criteria!r}: only None and 'error_based' exist for PTEQ")
    a, _ = L_.as_states(init, 3 if code in (L_.TORIC, L_.PLANAR) else 2)
    N, size = a.shape[0], a.shape[-1]
    Nc = Nc or size
    ncls = 16 if code == L_.TORIC else 4
    pr = L_.make_params(code=code, L=size, Nc=Nc, p=float(p), p_logical=float(p_logical), iters=int(iters),
                        steps=int(steps), tops_burn=int(tops_burn), TOPS=int(TOPS), SEQ=int(SEQ), eps=float(eps),
                        seed=seed, first_syndrome=first_syndrome, device=device,
                        conv_mode=L_.CONV_ERROR_BASED if conv_criteria else L_.CONV_NONE,
                        noise=L_.NOISE_ALPHA if alpha is not None else L_.NOISE_DEPOLARIZING if eta is None else L_.NOISE_BIASED,
                        eta=0.0 if eta is None else float(eta), alpha=0.0 if alpha is None else float(alpha),
                        scan=L_.SCANS[scan], replicas=int(replicas), flags=int(flags))
    R = max(int(replicas), 1)
    counts = np.zeros((N, ncls), dtype=np.uint32)
    samples = np.zeros(N, dtype=np.uint32)
    tops0 = np.zeros(N, dtype=np.uint32)
    steps_done = np.zeros(N, dtype=np.uint32)
    converged = np.zeros(N, dtype=np.uint8)
    states = np.empty((N * R, Nc) + a.shape[1:], dtype=np.uint8) if return_states else None
    swap_acc = np.zeros((N, max(Nc - 1, 1)), dtype=np.uint32) if return_swap_stats else None
    nerr = np.zeros((N, Nc), dtype=np.uint32) if return_swap_stats else None
    stats = L_.Stats()
    L_.check(L_.lib().qecmc_pteq_batch_stats(pr, L_.u8(a), N, L_.u32(counts), L_.u32(samples), L_.u32(tops0),
                                             L_.u32(steps_done), L_.u8(converged),
                                             L_.u8(states) if return_states else None,
                                             L_.u32(swap_acc) if return_swap_stats else None,
                                             L_.u32(nerr) if return_swap_stats else None, stats))
    out = dict(counts=counts, samples=samples, tops0=tops0, steps_done=steps_done, converged=converged.astype(bool),
               percent=percent_from_counts(counts, samples))
    if return_states:
        out["states"] = states
    if return_swap_stats:
        out["swap_accepts"], out["nerr_sums"] = swap_acc[:, :Nc - 1], nerr
    if return_stats:
        out["stats"] = dict(proposals=int(stats.proposals), swap_tests=int(stats.swap_tests),
                            kernel_ms=float(stats.kernel_ms), total_ms=float(stats.total_ms))
    return out


# Ladders per PTEQ(...) call.  1 = the reference's estimator: one ladder, its own stopping rule (decoders.py:25-89), which is what a
# caller that "drops in unchanged" must get.  A wavefront per temperature is 64 lanes wide, so replicas=64 costs what one costs and
# gives a lower-variance estimate (summed class counts; `converged` then means all 64 have converged, so the run ends with the
# slowest of them): opt in per call (replicas=64) or for a whole script (decoders.PTEQ_REPLICAS = 64).
PTEQ_REPLICAS = 1
# The criterion runs of the one-syndrome drop-ins: first horizon (ladder steps) and its growth when a ladder outlasts it (see _pteq)
PTEQ_FIRST_HORIZON = 1 << 20
PTEQ_HORIZON_GROWTH = 16
LAST_RUN = {}


def PTEQ(init_code, p, Nc=None, SEQ=2, TOPS=10, tops_burn=2, eps=0.1, steps=50000000, iters=10,
         conv_criteria='error_based', seed=None, replicas=None, scan="random"):
    """Drop-in for decoders.PTEQ (decoders.py:25): same arguments, returns the uint8 percent vector of
    the equivalence classes.  With the convergence criterion the run is issued with a growing horizon (2^20
    ladder steps, x16 until it converges or `steps` is reached) so that the default `steps = 5e7` never allocates
    a 5e7-entry error-count log up front.  Philox is counter-based, so a longer horizon replays the same
    trajectory: the answer equals that of a single run with the full `steps`; a ladder that stops within the first
    horizon -- every default-criterion run measured -- is one launch with no repeated step (decoders.LAST_RUN).

    The reference decodes ONE syndrome per call (generate_data.py:136) with one ladder, which occupies one lane of each of the
    Nc wavefronts; that is the default here too (PTEQ_REPLICAS = 1).  `replicas=R` runs R independent ladders -- each with the
    reference's bookkeeping and, if asked for, its own convergence stop -- in the lanes that would otherwise idle, and forms the
    percent vector from their summed class counts: the "droplets" pattern the reference itself uses for its other estimators
    (decoders.py:215-225).  R = 64 fills the wavefronts at the price of one; it is a different (lower-variance) estimator
    whose run ends when the slowest of the R ladders has converged, so it is opt-in.
    scan="colour" (with conv_criteria=None) decodes the one syndrome in the latency layout: a workgroup per ladder, a colour
    phase of generators per wavefront pass.  A ladder step of `iters` = 10 there is 10 PHASES -- about 1.4 sweeps of every rung
    at toric L = 9 -- against 10 proposals (0.06 sweep) in the lane-per-chain layout, at 1.3 x the time per step; that work
    ratio is why it reaches tops0 >= 10 in 17-41 ms instead of 187-366 ms (profiles/r03_latency.json).  The stop itself
    (tops0 >= 10, or the eps = 0.1 heuristic) is the reference's and says nothing about decoding quality at equal wall time.
    scan="wave": the reference's own chain per syndrome with a generator pick shared by the 64 ladders of a wavefront (the
    batched throughput layout, qecmc.pteq_batch; a one-syndrome call gains nothing from it)."""
    return _pteq(init_code, p, None, Nc, SEQ, TOPS, tops_burn, eps, steps, iters, conv_criteria, seed, replicas=replicas, scan=scan)


def conv_crit_error_based_PT(nbr_errors_bottom_chain, since_burn, tops_accepted, SEQ, eps):
    """decoders.py:93-105 (the host form of the criterion the kernels run in place): |mean of the 2nd quarter - mean of the 4th| < eps on
    the bottom chain's error-count series of length since_burn + 1 -> (accept, converged = tops_accepted >= SEQ)"""
    l = since_burn + 1
    with np.errstate(invalid="ignore"), __import__("warnings").catch_warnings():
        __import__("warnings").simplefilter("ignore")
        q2 = np.average(nbr_errors_bottom_chain[(l // 4): (l // 2)])
        q4 = np.average(nbr_errors_bottom_chain[(3 * l // 4): l])
    if abs(q2 - q4) < eps:
        return True, tops_accepted >= SEQ
    return False, False


def _pteq(init_code, p, eta, Nc, SEQ, TOPS, tops_burn, eps, steps, iters, conv_criteria, seed, alpha=None, replicas=None, scan="random"):
    if tops_burn >= TOPS:
        print('tops_burn has to be smaller than TOPS')
    seed = _fresh_seed() if seed is None else seed
    kw = dict(Nc=Nc or init_code.system_size, iters=iters, tops_burn=tops_burn, p_logical=0.5, seed=seed,
              code=_code_id(init_code), eta=eta, alpha=alpha, replicas=PTEQ_REPLICAS if replicas is None else int(replicas), scan=scan)
    if conv_criteria is None:
        return pteq_batch(init_code.qubit_matrix, p, steps=steps, **kw)["percent"][0]
    # The horizon only sizes the error-count log (2 B -- alpha rule 4 B -- per ladder and step of it; a converged ladder ends its launch at
    # once), so it starts at 2^20 steps -- beyond where the reference's defaults stop at every lattice size measured (L = 9: 3e4 ... 3.4e5
    # steps, profiles/r03_latency.json) -- and grows x16: a run that does outlast it replays at most 1/15 of its steps (round 3: 65 536 and
    # x4, up to a third).  LAST_RUN tells what happened.
    horizon = min(int(steps), PTEQ_FIRST_HORIZON)
    launches = replayed = 0
    while True:
        res = pteq_batch(init_code.qubit_matrix, p, steps=horizon, conv_criteria=conv_criteria, SEQ=SEQ, TOPS=TOPS,
                         eps=eps, **kw)
        launches += 1
        if res["converged"][0] or horizon >= int(steps):
            break
        replayed += horizon
        horizon = min(int(steps), horizon * PTEQ_HORIZON_GROWTH)
    LAST_RUN.update(launches=launches, horizon=horizon, steps_done=int(res["steps_done"][0]), replayed_steps=replayed)
    if not res["converged"][0]:
        print('\n\nWARNING: PTEQ hit max number of steps before convergence:\t', horizon, '\n\n')
    return res["percent"][0]


def ptdc_batch(init, p_sampling, Nc=None, steps=2000, droplets=1, iters=10, seed=0, first_syndrome=0, device=0, code=L_.TORIC,
               return_stats=False, per_rung=False, with_m=False, conv_mult=0.0, return_steps=False, with_xyz=False, alpha=None):
    """The sampling half of PTDC (decoders.py:168-233) on N syndromes at once.

    init: uint8[N, ncls, ...] -- one representative per equivalence class for every syndrome (what `to_class` / the list
    form of init_code provides) -- or uint8[N, ncls, droplets, ...] with a start of its own for every droplet (STDC's rain).
    For every (syndrome, class), `droplets` ladders without logical moves run `steps` ladder steps; returns N(n)
    uint32[N, ncls, nq+1], the number of DISTINCT chains of each length seen by any rung of any droplet (PTDC_droplet's
    dict, decoders.py:146-152,220-226).  `steps` is per ladder: PTDC passes steps // Nc (:201).  Nc = 1, iters = 5 is
    STDC_droplet / STRC_droplet (:236-265, :745-830).  per_rung=True keeps one set per (ladder, rung) as PTRC_droplet
    does (:584-631): shape [N, ncls, droplets, Nc, nq+1].  with_m=True also returns m(n), all observations by length.
    conv_mult != 0 is the early stop of PTDC_droplet / STDC_droplet / STRC_droplet (:153-162, :256-262, :783-826), per
    droplet; return_steps=True appends steps_done uint32[N, ncls, droplets], the steps each droplet recorded.
    with_xyz=True appends a list[N][ncls] of int64[k, 3] arrays: (n_x, n_y, n_z) of the k distinct chains of each set, sorted
    (the values of STDC_droplet_general_noise's dict, :325-342).  p_sampling may be an array (p_x, p_y, p_z): the chains
    are then Chain_xyz (src/mcmc.py:106-114), which needs Nc = 1.  alpha: the chains are Chain_alpha at
    pz_tilde = p_sampling (src/mcmc_alpha.py; STDC_droplet_alpha, :510-534: Nc = 1, iters = 5; xzzx / rotated codes)."""
    nd = 3 if code in (L_.TORIC, L_.PLANAR) else 2
    a = np.ascontiguousarray(init, dtype=np.uint8)
    per_droplet = a.ndim == nd + 3
    if a.ndim not in (nd + 2, nd + 3) or (per_droplet and a.shape[2] != droplets):
        raise ValueError(f"expected init of shape [N, classes, (droplets,) ...state], got {a.shape}")
    N, ncls, size = a.shape[0], a.shape[1], a.shape[-1]
    if ncls != (16 if code == L_.TORIC else 4):
        raise ValueError("one representative per equivalence class is needed")
    nq = int(np.prod(a.shape[-nd:]))
    Nc = Nc or size
    pxyz = None
    if np.ndim(p_sampling) != 0:
        pxyz = (C.c_double * 3)(*[float(v) for v in p_sampling])
        p_sampling = 0.1                                   # ignored by the Chain_xyz rule
    noise = dict(noise=L_.NOISE_ALPHA, alpha=float(alpha)) if alpha is not None else {}
    pr = L_.make_params(code=code, L=size, Nc=Nc, p=float(p_sampling), iters=int(iters), steps=int(steps), seed=seed,
                        first_syndrome=first_syndrome, device=device, **noise)
    shape = (N, ncls, int(droplets), Nc, nq + 1) if per_rung else (N, ncls, nq + 1)
    hist = np.zeros(shape, dtype=np.uint32)
    mh = np.zeros(shape, dtype=np.uint32) if with_m else None
    stats = L_.Stats()
    flags = (L_.PTDC_INIT_PER_DROPLET if per_droplet else 0) | (L_.PTDC_SET_PER_RUNG if per_rung else 0)
    sd = np.zeros((N, ncls, int(droplets)), dtype=np.uint32) if return_steps else None
    xv = np.zeros((N, ncls, int(steps) * Nc * int(droplets)), dtype=np.uint32) if with_xyz else None
    L_.check(L_.lib().qecmc_ptdc_batch_xyz(pr, L_.u8(a), N, int(droplets), flags, float(conv_mult or 0.0), pxyz, L_.u32(hist),
                                           L_.u32(mh) if with_m else None, L_.u32(sd) if return_steps else None,
                                           L_.u32(xv) if with_xyz else None, None, stats))
    out = (hist, mh) if with_m else hist
    if return_steps:
        out = (out if isinstance(out, tuple) else (out,)) + (sd,)
    if with_xyz:
        out = (out if isinstance(out, tuple) else (out,)) + ([[unpack_xyz(xv[s, c]) for c in range(ncls)] for s in range(N)],)
    if return_stats:
        return out, dict(proposals=int(stats.proposals), kernel_ms=float(stats.kernel_ms), total_ms=float(stats.total_ms))
    return out


def ptdc_distribution(hist, p_error):
    """decoders.py:208,229-233: Z_E = sum over the unique chains of exp(-beta n), normalised over the classes, x 100."""
    beta = -np.log((p_error / 3) / (1 - p_error))
    n = np.arange(hist.shape[-1], dtype=np.float64)
    Z = (hist.astype(np.float64) * np.exp(-beta * n)).sum(axis=-1)
    return Z / Z.sum(axis=-1, keepdims=True) * 100


def PTDC(init_code, p_error, p_sampling=None, droplets=4, Nc=None, steps=20000, conv_mult=0, seed=None):
    """Drop-in for decoders.PTDC (decoders.py:168): same arguments, returns the uint8 percent vector.  init_code is a code
    with `to_class` (toric) or a list with one code per class.  conv_mult: the early stop of :153-162."""
    p_sampling = p_sampling or p_error
    if isinstance(init_code, list):
        assert len(init_code) == init_code[0].nbr_eq_classes, 'if init_code is a list, it has to contain one code for each class'
        code0, reps = init_code[0], [c.qubit_matrix for c in init_code]
    else:
        code0, reps = init_code, [init_code.to_class(eq) for eq in range(init_code.nbr_eq_classes)]
    Nc = Nc or code0.system_size
    hist = ptdc_batch(np.stack(reps)[None], p_sampling, Nc=Nc, steps=steps // Nc, droplets=droplets, conv_mult=conv_mult,
                      seed=_fresh_seed() if seed is None else seed, code=_code_id(code0))
    return ptdc_distribution(hist[0], p_error).astype(np.uint8)


def STDC(init_code, p_error, p_sampling=None, droplets=10, steps=20000, conv_mult=0, seed=None):
    """Drop-in for decoders.STDC (decoders.py:268-322): single chains at p_sampling, `update_chain_fast(5)` per step,
    Z_E from the distinct chains found; returns the float percent vector.  init_code: a list with one code per class
    (no rain, :279), or a code with `to_class` (then every droplet starts from `apply_stabilizers_uniform()`, :246-247,
    :292).  The reference's jitted loop is hard-wired to the planar stencil (quirk Q1), which is the only model it is
    correct for; here the chain runs on the stencil of the code it is given.  conv_mult: the early stop of :256-262."""
    p_sampling = p_sampling or p_error
    import copy
    if isinstance(init_code, list):
        assert len(init_code) == init_code[0].nbr_eq_classes, 'if init_code is a list, it has to contain one code for each class'
        code0 = init_code[0]
        init = np.stack([c.qubit_matrix for c in init_code])[None]
    else:
        code0 = init_code
        reps = []
        for eq in range(init_code.nbr_eq_classes):
            c = copy.deepcopy(init_code)
            c.qubit_matrix = c.to_class(eq)
            reps.append(np.stack([c.apply_stabilizers_uniform() for _ in range(droplets)]))     # rain, one per droplet
        init = np.stack(reps)[None]
    hist = ptdc_batch(init, p_sampling, Nc=1, steps=steps, droplets=droplets, iters=5, seed=_fresh_seed() if seed is None else seed,
                      code=_code_id(code0), conv_mult=conv_mult)
    return ptdc_distribution(hist[0], p_error)


def unpack_xyz(vals):
    """n_x | n_y << 10 | n_z << 20 words (qecmc_ptdc_batch_xyz) -> sorted int64[k, 3]; the unused 0xFFFFFFFF tail is dropped."""
    v = np.asarray(vals, dtype=np.uint32)
    v = np.sort(v[v != 0xFFFFFFFF])
    return np.stack([v & 1023, (v >> 10) & 1023, (v >> 20) & 1023], axis=-1).astype(np.int64)


def general_noise_distribution(xyz, p_xyz, shortest_only=False):
    """STDC_general_noise's estimate (decoders.py:390-432; shortest_only / STDC_general_noise_shortest :494-507) from
    xyz[c] = int[k_c, 3], the (n_x, n_y, n_z) of the distinct chains found in class c: Z_c = sum exp(-sum_i beta_i n_i) over
    them -- with shortest_only over the chains whose weighted length is `np.isclose` to the smallest -- normalised, x 100."""
    p_xyz = np.asarray(p_xyz, dtype=np.float64)
    with np.errstate(divide="ignore"):
        beta = -np.log((p_xyz / 3) / (1 - p_xyz))                                              # :385
    Z = np.zeros(len(xyz))
    for c, q in enumerate(xyz):
        q = np.asarray(q).reshape(-1, 3)
        with np.errstate(invalid="ignore"):
            wl = np.sum(beta * q, axis=1, where=(q > 0))                                       # :403
        if shortest_only:
            wl = wl[np.isclose(wl, np.min(wl))]                                                # :407-408
        Z[c] = np.sum(np.exp(-wl))                                                             # :411
    return np.divide(Z, sum(Z)) * 100


def _general_noise_xyz(init_code, p_xyz, p_sampling, droplets, steps, seed):
    """the sampling of STDC_general_noise / _shortest (decoders.py:345-401): Chain at a scalar p_sampling (default
    p_xyz.sum()), Chain_xyz at an array; no rain in either form of init_code (`randomize = False`, :364,:378)"""
    p_xyz = np.asarray(p_xyz, dtype=np.float64)
    if p_sampling is None:
        p_sampling = p_xyz.sum()
    code0, init = _class_starts(init_code, droplets, rain=False)
    _, xyz = ptdc_batch(init, p_sampling, Nc=1, steps=steps, droplets=droplets, iters=5, with_xyz=True,
                        seed=_fresh_seed() if seed is None else seed, code=_code_id(code0))
    return p_xyz, xyz[0]


def STDC_general_noise(init_code, p_xyz, p_sampling=None, droplets=10, steps=20000, shortest_only=False, seed=None):
    """Drop-in for decoders.STDC_general_noise (decoders.py:345-432): single chains (`update_chain_fast(5)` per step) sampled
    by Chain, or by Chain_xyz when p_sampling is an array; Z_E from the (n_x, n_y, n_z) of the distinct chains with one
    beta per Pauli type.  Returns the float percent vector."""
    p_xyz, xyz = _general_noise_xyz(init_code, p_xyz, p_sampling, droplets, steps, seed)
    return general_noise_distribution(xyz, p_xyz, shortest_only)


def STDC_general_noise_shortest(init_code, p_xyz, p_sampling=None, droplets=10, steps=20000, seed=None):
    """Drop-in for decoders.STDC_general_noise_shortest (decoders.py:435-507): both estimates from one sampling run."""
    p_xyz, xyz = _general_noise_xyz(init_code, p_xyz, p_sampling, droplets, steps, seed)
    return general_noise_distribution(xyz, p_xyz), general_noise_distribution(xyz, p_xyz, shortest_only=True)


def nall_n_alpha_distribution(xyz, alpha, pz_tilde):
    """STDC_Nall_n_alpha's estimate (decoders.py:569-581) from xyz[c] = int[k_c, 3]: Z_c = sum over the distinct chains of
    exp(-beta (n_z + alpha (n_x + n_y))) with beta = -ln(pz_tilde) (:522, :569, :578), normalised, x 100."""
    beta = -np.log(pz_tilde)
    Z = np.zeros(len(xyz))
    for c, q in enumerate(xyz):
        q = np.asarray(q).reshape(-1, 3)
        Z[c] = np.sum(np.exp(-beta * (q[:, 2] + alpha * (q[:, 0] + q[:, 1]))))
    return np.divide(Z, sum(Z)) * 100


def STDC_Nall_n_alpha(init_code, pz_tilde_sampling=None, alpha=1, pz_tilde=0.1, steps=20000, seed=None):
    """Drop-in for decoders.STDC_Nall_n_alpha (decoders.py:537-581; `generate_data.py:190-196`, method "STDC_N_n"): one
    Chain_alpha per class at pz_tilde_sampling, `update_chain(5)` per step, Z_E from the effective lengths of the distinct
    chains.  init_code: a list with one code per class, or a code (moved to class eq by the logical operator
    `eq_class ^ eq`, :557-561; no rain either way -- STDC_droplet_alpha takes no `randomize`)."""
    import copy
    if isinstance(init_code, list):
        assert len(init_code) == init_code[0].nbr_eq_classes, 'if init_code is a list, it has to contain one code for each class'
        code0, reps = init_code[0], [c.qubit_matrix for c in init_code]
    else:
        code0, reps = init_code, []
        for eq in range(init_code.nbr_eq_classes):
            c = copy.deepcopy(init_code)
            reps.append(c.apply_logical(c.define_equivalence_class() ^ eq)[0])
    _, xyz = ptdc_batch(np.stack(reps)[None], pz_tilde_sampling, Nc=1, steps=steps, droplets=1, iters=5, with_xyz=True, alpha=alpha,
                        seed=_fresh_seed() if seed is None else seed, code=_code_id(code0))
    return nall_n_alpha_distribution(xyz[0], alpha, pz_tilde)


def strc_distribution(n_unique, m_obs, p_error, p_sampling):
    """STRC's estimate (decoders.py:863-949) from the counts of one syndrome: n_unique[c, n] = N(n), the distinct chains of
    length n of class c (union over the droplets); m_obs[c, n] = m(n), all observations (summed over the droplets).
    Z_e = sum_l m(l) exp(-beta_s n0 + d_beta l) x mean fraction of distinct chains at the two shortest lengths."""
    from math import exp, log
    beta_error = -log((p_error / 3) / (1 - p_error))
    beta_sampling = -log((p_sampling / 3) / (1 - p_sampling))
    d_beta = beta_sampling - beta_error
    Z = np.zeros(n_unique.shape[0])
    for c in range(n_unique.shape[0]):
        lengths = np.flatnonzero(m_obs[c])
        shortest = int(lengths[0])
        fraction = n_unique[c, shortest] / m_obs[c, shortest]                                  # :925-926
        if len(lengths) > 1:                                                                   # :931-933
            nxt = int(lengths[1])
            fraction = 0.5 * (fraction + n_unique[c, nxt] / m_obs[c, nxt] * exp(-beta_sampling * (nxt - shortest)))
        Z[c] = sum(int(m_obs[c, l]) * exp(-beta_sampling * shortest + d_beta * int(l)) for l in lengths) * fraction   # :939
    return Z / np.sum(Z) * 100


def ptrc_distribution(n_unique, m_obs, p_error, p_sampling):
    """PTRC's estimate (decoders.py:684-742) from the per-rung counts of one syndrome, n_unique / m_obs[c, droplet, rung, n]
    (N(n) and m(n) are summed over the droplets, :705-717; the top rung is left out, :720)."""
    from math import log
    Nc = n_unique.shape[2]
    p_ladder = np.linspace(p_sampling, 0.75, Nc)
    beta_error = -log((p_error / 3) / (1 - p_error))
    beta_ladder = -np.log((p_ladder[:-1] / 3) / (1 - p_ladder[:-1]))
    d_beta = beta_ladder - beta_error
    Nn, mn = n_unique.sum(axis=1).astype(np.float64), m_obs.sum(axis=1).astype(np.float64)
    Z = np.zeros(n_unique.shape[0])
    for c in range(n_unique.shape[0]):
        for i in range(Nc - 1):
            lengths = np.flatnonzero(mn[c, i])
            counts = np.stack([Nn[c, i, lengths], mn[c, i, lengths]], axis=1)
            C_mean = np.mean(counts[:2, 0] / counts[:2, 1] * np.exp(-beta_ladder[i] * (lengths[:2] - lengths[0])))     # :732
            Z[c] += C_mean * (counts[:, 1] * np.exp(lengths * d_beta[i] - beta_ladder[i] * lengths[0])).sum()          # :735
    return (Z / np.sum(Z) * 100).astype(np.uint8)


def _class_starts(init_code, droplets, rain):
    """[1, classes, (droplets,) ...] starts from a list with one code per class, or from a code with `to_class`
    (then, with rain, every droplet starts from apply_stabilizers_uniform(), decoders.py:246-247)."""
    import copy
    if isinstance(init_code, list):
        assert len(init_code) == init_code[0].nbr_eq_classes, 'if init_code is a list, it has to contain one code for each class'
        return init_code[0], np.stack([c.qubit_matrix for c in init_code])[None]
    reps = []
    for eq in range(init_code.nbr_eq_classes):
        c = copy.deepcopy(init_code)
        c.qubit_matrix = c.to_class(eq)
        reps.append(np.stack([c.apply_stabilizers_uniform() for _ in range(droplets)]) if rain else c.qubit_matrix)
    return init_code, np.stack(reps)[None]


def STRC(init_code, p_error, p_sampling=None, droplets=10, steps=20000, conv_mult=0, seed=None):
    """Drop-in for decoders.STRC (decoders.py:835-949): single chains as in STDC; the estimate uses m(n) and the number of
    distinct chains at the two shortest lengths.  Returns the float percent vector.  conv_mult: the early stop of :783-826."""
    p_sampling = p_sampling or p_error
    code0, init = _class_starts(init_code, droplets, rain=True)
    n_u, m_o = ptdc_batch(init, p_sampling, Nc=1, steps=steps, droplets=droplets, iters=5, with_m=True, conv_mult=conv_mult,
                          seed=_fresh_seed() if seed is None else seed, code=_code_id(code0))
    return strc_distribution(n_u[0], m_o[0], p_error, p_sampling)


def PTRC(init_code, p_error, p_sampling=None, droplets=4, Nc=None, steps=20000, conv_mult=2.0, seed=None):
    """Drop-in for decoders.PTRC (decoders.py:638-742): class ladders as in PTDC, per-rung N(n) and m(n), curve estimate
    per rung.  Returns the uint8 percent vector.  (conv_mult is accepted and, as in the reference -- whose early stop is
    commented out, :627-630 -- has no effect.)"""
    p_sampling = p_sampling or p_error
    code0, init = _class_starts(init_code, droplets, rain=False)
    Nc = Nc or code0.system_size
    n_u, m_o = ptdc_batch(init, p_sampling, Nc=Nc, steps=steps // Nc, droplets=droplets, per_rung=True, with_m=True,
                          seed=_fresh_seed() if seed is None else seed, code=_code_id(code0))
    return ptrc_distribution(n_u[0], m_o[0], p_error, p_sampling)


def single_temp(init_code, p, max_iters, seed=None):
    """Drop-in for decoders.single_temp (decoders.py:108-135): one chain per class at p, `update_chain_fast(5)` per step; the
    mean number of errors over the first max_iters - 1 steps (`np.average(nbr_errors_chain[eq, :j])` with j = max_iters - 1)."""
    code0, init = _class_starts(init_code, 1, rain=False)
    _, m_o = ptdc_batch(init, p, Nc=1, steps=max_iters - 1, droplets=1, iters=5, with_m=True,
                        seed=_fresh_seed() if seed is None else seed, code=_code_id(code0))
    n = np.arange(m_o.shape[-1], dtype=np.float64)
    return (m_o[0] * n).sum(axis=-1) / (max_iters - 1)
