"""Batched counterpart of the reference's data-generation harness (`generate_data.generate`,
generate_data.py:20-269).

The reference makes one random error, hides its class with a random logical operator, decodes it
with one `PTEQ` call and appends (qubit_matrix, distribution) to a pandas pickle, one syndrome at a
time.  Here the seed configurations of a whole batch are drawn on the host (NumPy, the recipe of
toric_model.py:15-24 / xzzx_model.py:16-30 and generate_data.py:121-131), decoded in ONE batched
GPU call, and returned / saved as plain arrays (npz instead of the MultiIndex pickle).
"""
import numpy as np

from . import _lib as L_
from .decoders import (nall_n_alpha_distribution, pteq_batch, ptdc_batch, ptdc_distribution, ptrc_distribution,
                       strc_distribution)

_CODES = {"toric": L_.TORIC, "xzzx": L_.XZZX, "rotated": L_.ROTATED, "planar": L_.PLANAR}


def _class_of(code, m):
    from . import _surf, planar_model, toric_model
    if code == L_.PLANAR:
        return planar_model.eq_class(m)
    return toric_model.eq_class(m) if code == L_.TORIC else _surf.eq_class(code, m)


def alpha_rates(pz_tilde, alpha):
    """(p_x, p_y, p_z) of the alpha noise model (generate_data.py:84-91, mcmc_alpha.py:31-36)."""
    p_tilde = pz_tilde + 2 * pz_tilde ** alpha
    p = p_tilde / (1 + p_tilde)
    px = pz_tilde ** alpha * (1 - p)
    return px, px, pz_tilde * (1 - p)


def biased_as_alpha(p, eta):
    """generate_data.py:145-146: the (pz_tilde, alpha) that `generate` hands to PTEQ_alpha for noise = 'biased'."""
    pz_tilde = (p / (1 + 1 / eta)) / (1 - p)
    return pz_tilde, np.log(pz_tilde / (2 * eta)) / np.log(pz_tilde)


def draw_errors(code, size, n, p_error, rng, eta=None, rates=None):
    """n random error chains: toric -- each qubit errs w.p. p_error, Pauli uniform (toric_model.py:15-23);
    xzzx / rotated -- one uniform per qubit against (p_z, p_x, p_y) (xzzx_model.py:16-30), with
    p_x = p_y = p_z = p/3 (generate_data.py:116-118) or the Z-biased split p_z = p eta/(eta+1),
    p_x = p_y = p/(2(eta+1)) (generate_data.py:78-83)."""
    code = _CODES.get(code, code)
    if code == L_.TORIC:
        m = np.zeros((n, 2, size, size), dtype=np.uint8)
        err = rng.random(m.shape) < p_error
        m[err] = rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)
        return m
    if rates is not None:
        px, py, pz = rates
    elif eta is None:
        px = py = pz = p_error / 3
    else:
        pz, px = p_error * eta / (eta + 1), p_error / (2 * (eta + 1))
        py = px
    shape = (n, 2, size, size) if code == L_.PLANAR else (n, size, size)
    r = rng.random(shape)
    m = np.zeros(shape, dtype=np.uint8)
    m[r < pz] = 3
    m[(r > pz) & (r < pz + px)] = 1
    m[(r > pz + px) & (r < pz + px + py)] = 2
    if code == L_.PLANAR:                 # layer 1 lives on its first L-1 rows / columns (planar_model.py:38-39)
        m[:, 1, -1, :] = 0
        m[:, 1, :, -1] = 0
    return m


def hide_class(code, m, rng):
    """`init_code.qubit_matrix, _ = init_code.apply_random_logical()` (generate_data.py:131) on a batch:
    the operator draws of toric_model.py:234-248 / xzzx_model.py:346-355, applied by the device stencil."""
    from . import _surf, toric_model
    code = _CODES.get(code, code)
    n, size = m.shape[0], m.shape[-1]
    if code == L_.TORIC:
        for layer in range(2):
            ops = rng.integers(0, 4, size=n)
            xpos = np.where(np.isin(ops, (1, 2)), rng.integers(0, size, size=n), 0)
            zpos = np.where(np.isin(ops, (3, 2)), rng.integers(0, size, size=n), 0)
            m, _ = toric_model.apply_logical(m, ops, layer, xpos, zpos)
        return m
    ops = rng.integers(0, 4, size=n)
    xpos = np.where(np.isin(ops, (1, 2)), rng.integers(0, size, size=n), 0)
    zpos = np.where(np.isin(ops, (3, 2)), rng.integers(0, size, size=n), 0)
    if code == L_.PLANAR:
        from . import planar_model
        return planar_model.apply_logical(m, ops, xpos, zpos)[0]
    m, _ = _surf.apply_logical(code, m, ops, xpos, zpos)
    return m


def class_representatives(code, m):
    """uint8[n, ...] -> uint8[n, ncls, ...]: the configuration with the same syndrome in every equivalence class, indexed by
    class (`to_class(eq)` for the toric code, toric_model.py:354; for the 4-class codes the four logical operators, as
    STDC_Nall_n_alpha does with `apply_logical(eq_class ^ eq)`, decoders.py:557-561)."""
    from . import _surf, planar_model, toric_model
    code = _CODES.get(code, code)
    m = np.ascontiguousarray(m, dtype=np.uint8)
    if code == L_.TORIC:
        return np.stack([toric_model.to_class(m, eq) for eq in range(16)], axis=1)
    out = np.empty((m.shape[0], 4) + m.shape[1:], dtype=np.uint8)
    for op in range(4):
        r = planar_model.apply_logical(m, op)[0] if code == L_.PLANAR else _surf.apply_logical(code, m, op)[0]
        out[np.arange(m.shape[0]), np.asarray(_class_of(code, r))] = r
    return out


def rain(code, m, rng, p=0.5):
    """`apply_stabilizers_uniform` (toric_model.py:299-314, planar_model.py:355-376) on a batch: every generator is applied
    to every configuration independently with probability p (STDC's / STRC's high-energy start, decoders.py:246-247)."""
    from . import planar_model, toric_model
    code = _CODES.get(code, code)
    if code not in (L_.TORIC, L_.PLANAR):
        raise ValueError("rain is defined for the toric and planar codes (the only models with apply_stabilizers_uniform)")
    mod = toric_model if code == L_.TORIC else planar_model
    m = np.ascontiguousarray(m, dtype=np.uint8).copy()
    n, size = m.shape[0], m.shape[-1]
    pick = rng.random((n, 2, size, size)) < p
    if code == L_.PLANAR:                       # no X-type generator in the last row, no Z-type in the last column
        pick[:, 1, size - 1, :] = False
        pick[:, 0, :, size - 1] = False
    for o in range(2):
        for r in range(size):
            for c in range(size):
                sel = np.flatnonzero(pick[:, o, r, c])
                if sel.size:
                    m[sel] = mod.apply_stabilizer(m[sel], r, c, 3 if o == 0 else 1)[0]
    return m


def _estimate(method, code, reps, p_error, p_sampling, Nc, steps, droplets, conv_mult, seed, first, rng, alpha):
    """the unique-chain estimators of generate_data.py:168-196 on a batch of class representatives [n, ncls, ...] -> float[n, ncls]"""
    n = reps.shape[0]
    if method == "PTDC":                                                       # decoders.py:168-233
        hist = ptdc_batch(reps, p_sampling, Nc=Nc, steps=steps // Nc, droplets=droplets, conv_mult=conv_mult, seed=seed,
                          first_syndrome=first, code=code)
        return ptdc_distribution(hist, p_error)
    if method == "PTRC":                                                       # decoders.py:638-742
        n_u, m_o = ptdc_batch(reps, p_sampling, Nc=Nc, steps=steps // Nc, droplets=droplets, per_rung=True, with_m=True, seed=seed,
                              first_syndrome=first, code=code)
        return np.stack([ptrc_distribution(n_u[i], m_o[i], p_error, p_sampling) for i in range(n)]).astype(np.float64)
    if method in ("STDC", "STRC"):                                             # decoders.py:268-322, :835-949: rain per droplet
        starts = np.stack([rain(code, reps.reshape((-1,) + reps.shape[2:]), rng).reshape(reps.shape) for _ in range(droplets)], axis=2)
        if method == "STDC":
            hist = ptdc_batch(starts, p_sampling, Nc=1, steps=steps, droplets=droplets, iters=5, conv_mult=conv_mult, seed=seed,
                              first_syndrome=first, code=code)
            return ptdc_distribution(hist, p_error)
        n_u, m_o = ptdc_batch(starts, p_sampling, Nc=1, steps=steps, droplets=droplets, iters=5, with_m=True, conv_mult=conv_mult,
                              seed=seed, first_syndrome=first, code=code)
        return np.stack([strc_distribution(n_u[i], m_o[i], p_error, p_sampling) for i in range(n)])
    if method == "STDC_N_n":                                                   # decoders.py:537-581 (generate_data.py:190-196)
        _, xyz = ptdc_batch(reps, p_sampling, Nc=1, steps=steps, droplets=1, iters=5, with_xyz=True, alpha=alpha, seed=seed,
                            first_syndrome=first, code=code)
        return np.stack([nall_n_alpha_distribution(xyz[i], alpha, p_error) for i in range(n)])
    raise ValueError(f"method={method!r}")


def generate(params, nbr_datapoints, seed=0, file_path=None, steps=100000, conv_criteria="error_based", biased_decoder="alpha",
             **pteq_kw):
    """params: dict like generate_data.py:276-296 ({'code','size','p_error','noise'[,'eta','alpha']}), method PTEQ.
    noise 'depolarizing' -> PTEQ (:136); 'biased' -> errors from the eta split (:78-83) decoded by PTEQ_alpha with
    (pz_tilde, alpha) derived from (p, eta) exactly as :142-150 does (biased_decoder="biased" decodes with PTEQ_biased
    instead); 'alpha' -> p_error is pz_tilde, errors and decoder from (pz_tilde, alpha) (:84-91,:151-160).
    params['method'] (default "PTEQ") may also be "PTDC", "PTRC", "STDC", "STRC" or "STDC_N_n" (generate_data.py:168-196): the
    unique-chain estimators on one representative per class of every syndrome, with params['p_sampling'] (default p_error),
    params['droplets'], params['conv_mult'] and `steps` as the estimator's own `steps`; `batch` syndromes go into one launch
    (default 256: the sets of visited chains live in HBM).  They return distr float64[n, ncls] and no counts.
    Returns (and optionally saves as npz) qubit_matrix uint8[n,...] (the raw errors, generate_data.py:120),
    eq_true int32[n], counts uint32[n,ncls], distr uint8[n,ncls] (what PTEQ returns), success bool[n]
    (argmax(distr) == eq_true, generate_data.py:139), steps_done, converged."""
    code = _CODES[params["code"]]
    size, p = params["size"], params["p_error"]
    noise = params.get("noise", "depolarizing")
    if noise not in ("depolarizing", "biased", "alpha"):
        raise ValueError(f"noise={noise!r}")
    eta = params.get("eta") if noise == "biased" else None
    rng = np.random.default_rng(seed)
    raw = draw_errors(code, size, nbr_datapoints, p, rng, eta, rates=alpha_rates(p, params["alpha"]) if noise == "alpha" else None)
    eq_true = np.asarray(_class_of(code, raw), dtype=np.int32)
    init = hide_class(code, raw, rng)
    method = params.get("method", "PTEQ")
    if method != "PTEQ":
        if noise != ("alpha" if method == "STDC_N_n" else "depolarizing"):
            raise ValueError(f"method {method} is defined for {'alpha' if method == 'STDC_N_n' else 'depolarizing'} noise (generate_data.py:168-196)")
        batch = int(pteq_kw.pop("batch", 256))
        ncls = 16 if code == L_.TORIC else 4
        distr = np.empty((nbr_datapoints, ncls), dtype=np.float64)
        droplets = params.get("droplets", 1 if method == "STDC_N_n" else 4)
        for lo in range(0, nbr_datapoints, batch):
            reps = class_representatives(code, init[lo:lo + batch])
            # every (syndrome, class, droplet) ladder needs a Philox syndrome index of its own
            distr[lo:lo + batch] = _estimate(method, code, reps, p, params.get("p_sampling") or p, params.get("Nc") or size, steps, droplets,
                                             params.get("conv_mult", 0), seed, lo * ncls * droplets, rng, params.get("alpha"))
        out = dict(qubit_matrix=raw, eq_true=eq_true, distr=distr, success=np.argmax(distr, axis=1) == eq_true)
        if file_path is not None:
            np.savez_compressed(file_path, params=np.array([repr(params)]), **out)
        return out
    dec = dict(eta=eta)
    p_dec = p
    if noise == "alpha":
        dec = dict(alpha=params["alpha"])
    elif noise == "biased" and biased_decoder == "alpha":
        p_dec, a = biased_as_alpha(p, eta)
        dec = dict(alpha=float(a))
    res = pteq_batch(init, p_dec, Nc=params.get("Nc"), steps=steps, conv_criteria=conv_criteria, seed=seed, code=code, **dec,
                     **pteq_kw)
    out = dict(qubit_matrix=raw, eq_true=eq_true, counts=res["counts"], distr=res["percent"],
               success=np.argmax(res["percent"], axis=1) == eq_true, steps_done=res["steps_done"],
               converged=res["converged"], samples=res["samples"])
    if file_path is not None:
        np.savez_compressed(file_path, params=np.array([repr(params)]), **out)
    return out


def convergence_study(init, p, checkpoints, Nc=None, iters=10, tops_burn=0, seed=0, code=L_.TORIC, **pteq_kw):
    """Class histograms of the same chains at increasing run lengths (BASELINE config 5: "long-chain convergence study").
    Philox is counter-based, so a run of `s` ladder steps is the exact prefix of any longer run with the same seed: the
    batch is simply re-run to every checkpoint (total cost < 2x the longest run for log-spaced checkpoints).
    Returns dict(steps int64[k], counts uint32[k, N, ncls], samples uint32[k, N], percent uint8[k, N, ncls],
    tv float64[k]) with tv = mean total-variation distance of each checkpoint's class distribution to the last one."""
    cps = sorted(int(c) for c in checkpoints)
    runs = [pteq_batch(init, p, Nc=Nc, steps=c, iters=iters, tops_burn=tops_burn, seed=seed, code=code, **pteq_kw) for c in cps]
    counts = np.stack([r["counts"] for r in runs])
    samples = np.stack([r["samples"] for r in runs])
    frac = counts / np.maximum(samples, 1)[..., None].astype(np.float64)
    tv = 0.5 * np.abs(frac - frac[-1]).sum(axis=-1).mean(axis=-1)
    return dict(steps=np.array(cps, dtype=np.int64), counts=counts, samples=samples,
                percent=np.stack([r["percent"] for r in runs]), tv=tv)
