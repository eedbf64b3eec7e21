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


def generate_syndromes(code, size, n, p_error=None, eta=None, rates=None, hide=True, seed=0, first_syndrome=0):
    """The generation half of the recipe ON THE DEVICE (qecmc_generate_syndromes): n error chains by the model's
    generate_random_error, their equivalence class, and -- hide=True -- one apply_random_logical on top (generate_data.py:110-131).
    Rates as in draw_errors: toric / depolarizing p_x = p_y = p_z = p_error / 3, the Z-biased split for eta, or explicit `rates`.
    Syndrome s draws from Philox (seed, first_syndrome + s).  Returns (init uint8[n, ...], raw uint8[n, ...], eq_true int32[n])."""
    code = _CODES.get(code, code)
    if rates is not None:
        px, py, pz = rates
    elif eta is None or code == L_.TORIC:
        px = py = pz = p_error / 3
    else:
        pz, px = p_error * eta / (eta + 1), p_error / (2 * (eta + 1))
        py = px
    shape = (n, 2, size, size) if code in (L_.TORIC, L_.PLANAR) else (n, size, size)
    init = np.zeros(shape, dtype=np.uint8)
    raw = np.zeros(shape, dtype=np.uint8)
    eq = np.zeros(n, dtype=np.int32)
    L_.check(L_.lib().qecmc_generate_syndromes(code, size, n, float(px), float(py), float(pz), int(bool(hide)), seed & 0xFFFFFFFFFFFFFFFF,
                                               first_syndrome, L_.u8(init), L_.u8(raw), L_.i32(eq)))
    return init, raw, eq


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
             rng=None, device_generation=False, metrics="basic", **pteq_kw):
    """params: dict like generate_data.py:276-296 ({'code','size','p_error','noise'[,'eta','alpha']}), method PTEQ.
    noise 'depolarizing' -> PTEQ (:136); 'biased' -> errors from the eta split (:78-83) decoded by PTEQ_alpha with
    (pz_tilde, alpha) derived from (p, eta) exactly as :142-150 does (biased_decoder="biased" decodes with PTEQ_biased
    instead); 'alpha' -> p_error is pz_tilde, errors and decoder from (pz_tilde, alpha) (:84-91,:151-160).
    params['method'] (default "PTEQ") may also be "PTDC", "PTRC", "STDC", "STRC" or "STDC_N_n" (generate_data.py:168-196): the
    unique-chain estimators on one representative per class of every syndrome, with params['p_sampling'] (default p_error),
    params['droplets'], params['conv_mult'] and `steps` as the estimator's own `steps`; `batch` syndromes go into one launch
    (default 256: the sets of visited chains live in HBM).  They return distr float64[n, ncls] and no counts.
    device_generation=True draws the errors and the hiding logical operator on the GPU (`generate_syndromes`) instead of NumPy.
    metrics="basic" (default) costs nothing: throughput, convergence, burn-in and success figures.  metrics="full" also attaches the
    mixing counters (swap acceptance per rung pair, mean error count per rung) -- which rule out the work-queue kernels
    (their lanes run several ladders) and need nq * steps < 2^32, so they are dropped, not failed on, where they do not fit.
    Returns (and optionally saves as npz) qubit_matrix uint8[n,...] (the raw errors, generate_data.py:120),
    eq_true int32[n], counts uint32[n,ncls], distr uint8[n,ncls] (what PTEQ returns), success bool[n]
    (argmax(distr) == eq_true, generate_data.py:139), steps_done, converged."""
    code = _CODES[params["code"]]
    size, p = params["size"], params["p_error"]
    noise = params.get("noise", "depolarizing")
    if noise not in ("depolarizing", "biased", "alpha"):
        raise ValueError(f"noise={noise!r}")
    eta = params.get("eta") if noise == "biased" else None
    rng = np.random.default_rng(seed) if rng is None else rng         # (shards pass a generator keyed by (seed, shard id))
    rates = alpha_rates(p, params["alpha"]) if noise == "alpha" else None
    if device_generation:
        # errors, true class and the hiding logical operator drawn on the GPU (Philox keyed by the global syndrome index: the data
        # set does not depend on how it is cut into shards)
        init, raw, eq_true = generate_syndromes(code, size, nbr_datapoints, p, eta, rates, True, seed, int(pteq_kw.get("first_syndrome", 0)))
    else:
        raw = draw_errors(code, size, nbr_datapoints, p, rng, eta, rates=rates)
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
    import time
    Nc = params.get("Nc") or size
    t0 = time.perf_counter()
    if metrics not in ("basic", "full"):
        raise ValueError(f"metrics={metrics!r}")
    nq = int(np.prod(init.shape[1:]))
    swap_stats = metrics == "full" and Nc > 1 and int(pteq_kw.get("replicas", 1)) <= 1 and nq * int(steps) < 2 ** 32
    res = pteq_batch(init, p_dec, Nc=Nc, steps=steps, conv_criteria=conv_criteria, seed=seed, code=code, return_stats=True,
                     return_swap_stats=swap_stats, **dec, **pteq_kw)
    wall = time.perf_counter() - t0
    out = dict(qubit_matrix=raw, eq_true=eq_true, counts=res["counts"], distr=res["percent"],
               success=np.argmax(res["percent"], axis=1) == eq_true, steps_done=res["steps_done"],
               converged=res["converged"], samples=res["samples"], tops0=res["tops0"])
    if file_path is not None:
        np.savez_compressed(file_path, params=np.array([repr(params)]), **out)
    out["metrics"] = batch_metrics(code, size, Nc, int(pteq_kw.get("iters", 10)), res, wall, out["success"],
                                   replicas=int(pteq_kw.get("replicas", 1)))
    return out


def batch_metrics(code, size, Nc, iters, res, wall_s, success=None, replicas=1):
    """The per-batch metrics line (SURVEY.md 5 "Metrics / logging"; the reference only prints a progress counter,
    generate_data.py:266): proposals, seconds, chain-sweeps/s, swap acceptance per rung pair, mean error count per rung,
    the tops0 histogram, and the success rate when the true classes are known.  JSON-serialisable."""
    n_gen = 2 * size * size if code == L_.TORIC else 2 * size * (size - 1) if code == L_.PLANAR else size * size - 1
    steps_run = res["steps_done"].astype(np.float64)
    proposals = float(steps_run.sum()) * Nc * iters * max(int(replicas), 1)
    k_ms = res.get("stats", {}).get("kernel_ms", float("nan"))
    m = dict(syndromes=int(res["counts"].shape[0]), Nc=int(Nc), iters=int(iters), proposals=proposals, wall_s=float(wall_s),
             kernel_ms=float(k_ms), chain_sweeps_per_s_kernel=float(proposals / n_gen / (k_ms * 1e-3)) if k_ms == k_ms and k_ms > 0 else None,
             chain_sweeps_per_s_wall=float(proposals / n_gen / wall_s) if wall_s > 0 else None,
             converged_frac=float(np.mean(res["converged"])), mean_steps=float(steps_run.mean()) if steps_run.size else 0.0,
             frac_past_burn_in=float(np.mean(res["samples"] > 0)) if res["samples"].size else 0.0,
             tops0_hist=np.bincount(np.minimum(res["tops0"], 20).astype(np.int64), minlength=21).tolist())
    if "swap_accepts" in res and steps_run.sum() > 0:
        m["swap_acceptance"] = (res["swap_accepts"].sum(axis=0) / steps_run.sum()).tolist()
        m["mean_errors_per_rung"] = (res["nerr_sums"].sum(axis=0) / steps_run.sum()).tolist()
    if success is not None and len(success):
        k, n = int(np.sum(success)), len(success)
        m["success_rate"] = k / n
        m["success_rate_err"] = float(np.sqrt(max(k / n * (1 - k / n), 0.0) / n))
    return m


def shard_name(prefix, seed, shard):
    return f"{prefix}_seed{int(seed)}_shard{int(shard):05d}.npz"


def generate_shards(params, n_total, shard_size, out_dir, seed=0, prefix="data", log=None, **gen_kw):
    """The reference's array job (generate_data.py:251-256,274-276; generate_data_noise_models.py:27-32): the data set is cut
    into shards keyed by (seed, shard id); a shard whose file exists is skipped, so an interrupted job resumes where it
    stopped, and any subset of shards can be (re)made on any GPU in any order with the same result -- shard k draws its
    errors from default_rng([seed, k]) and its ladders from Philox syndrome indices k*shard_size ....
    One JSON metrics line per shard made is appended to <out_dir>/<prefix>_metrics.jsonl (and to `log`, a list, if given).
    Returns the list of shard paths (all of them, made or found)."""
    import json
    import os
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    n_shards = (int(n_total) + int(shard_size) - 1) // int(shard_size)
    for k in range(n_shards):
        path = os.path.join(out_dir, shard_name(prefix, seed, k))
        paths.append(path)
        if os.path.exists(path):                                   # resume = skip existing shards
            continue
        n = min(int(shard_size), int(n_total) - k * int(shard_size))
        tmp = path + ".tmp.npz"
        out = generate(params, n, seed=seed, file_path=tmp, rng=np.random.default_rng([int(seed), k]),
                       first_syndrome=k * int(shard_size) * max(int(gen_kw.get("replicas", 1)), 1), **gen_kw)
        os.replace(tmp, path)                                      # a shard file is complete or absent
        line = dict(shard=k, seed=int(seed), file=os.path.basename(path), **out.get("metrics", {}))
        with open(os.path.join(out_dir, f"{prefix}_metrics.jsonl"), "a") as f:
            f.write(json.dumps(line) + "\n")
        if log is not None:
            log.append(line)
    return paths


def threshold_curve(params, p_list, n, seed=0, **gen_kw):
    """Logical success rate against the physical error rate (the p_error scan of generate_data.py:57-60,121-141,276-296 --
    the reference loops p over [0.05, 0.20] in its job script and evaluates argmax(distr) == true class offline):
    for every p in p_list, n syndromes at p_error = p decoded in one batched call.
    Returns dict(p, n, success_rate, err (binomial standard error), converged_frac, metrics [one dict per p]) and, beside the raw
    rate, success_rate_sampled / err_sampled / frac_sampled: the rate among the syndromes whose ladder got past the burn-in
    (samples > 0).  The reference's burn-in trap (decoders.py:63,89: a ladder that never sees tops0 >= tops_burn returns an all-zero
    vector, argmax 0) counts as a failure in the raw rate; at low p and a short horizon it, not the decoder, sets that number."""
    rate, err, conv, met, rate_s, err_s, frac_s = [], [], [], [], [], [], []
    for i, p in enumerate(p_list):
        out = generate(dict(params, p_error=float(p)), n, seed=seed + i, **gen_kw)
        k = float(np.mean(out["success"]))
        rate.append(k); err.append(float(np.sqrt(max(k * (1 - k), 0.0) / n)))
        conv.append(float(np.mean(out["converged"])) if "converged" in out else float("nan"))
        met.append(out.get("metrics"))
        sampled = out["samples"] > 0 if "samples" in out else np.ones(len(out["success"]), dtype=bool)
        ns = int(sampled.sum())
        ks = float(np.mean(out["success"][sampled])) if ns else float("nan")
        rate_s.append(ks); err_s.append(float(np.sqrt(max(ks * (1 - ks), 0.0) / ns)) if ns else float("nan")); frac_s.append(ns / max(n, 1))
    return dict(p=np.asarray(p_list, dtype=np.float64), n=int(n), success_rate=np.array(rate), err=np.array(err),
                converged_frac=np.array(conv), metrics=met, success_rate_sampled=np.array(rate_s), err_sampled=np.array(err_s),
                frac_sampled=np.array(frac_s))


class LadderRun:
    """N ladders resident in HBM, advanced in chunks by qecmc_pteq_resume_dev: chunked runs reproduce one long run bit for
    bit (Philox is addressed by the steps already done), so a 10^6-sweep study costs one pass, checkpoints are free, and no
    single launch runs for minutes.  Device memory through torch (plumbing)."""

    def __init__(self, init, p, Nc=None, iters=10, tops_burn=2, p_logical=0.5, seed=0, first_syndrome=0, device=0,
                 code=L_.TORIC, eta=None):
        import torch
        a, _ = L_.as_states(init, 3 if code in (L_.TORIC, L_.PLANAR) else 2)
        self.N, size = a.shape[0], a.shape[-1]
        self.Nc = Nc or size
        self.ncls = 16 if code == L_.TORIC else 4
        self.shape = a.shape[1:]
        nq = int(np.prod(self.shape))
        self._mk = lambda steps: L_.make_params(code=code, L=size, Nc=self.Nc, p=float(p), p_logical=float(p_logical), iters=int(iters),
                                                steps=int(steps), tops_burn=int(tops_burn), seed=seed, device=device,
                                                noise=L_.NOISE_DEPOLARIZING if eta is None else L_.NOISE_BIASED, eta=float(eta or 0.0))
        self.first = int(first_syndrome)
        dev = torch.device("cuda", device)
        st = np.broadcast_to(a.reshape(self.N, 1, nq), (self.N, self.Nc, nq))            # Ladder.__init__, mcmc.py:72
        self.states = torch.from_numpy(np.array(st, order="C")).to(dev)                  # (a writable copy of the broadcast view)
        fl = np.zeros((self.N, self.Nc), dtype=np.uint8); fl[:, -1] = 1                    # mcmc.py:75
        self.flags = torch.from_numpy(fl).to(dev)
        self.tops0 = torch.zeros(self.N, dtype=torch.int32, device=dev)
        self.counts = torch.zeros((self.N, self.ncls), dtype=torch.int32, device=dev)
        self.samples = torch.zeros(self.N, dtype=torch.int32, device=dev)
        self.steps = 0
        self._plans = {}
        self._torch = torch

    def advance(self, steps):
        import ctypes as C
        steps = int(steps)
        if steps <= 0:
            return self
        if steps not in self._plans:
            pl = C.c_void_p()
            L_.check(L_.lib().qecmc_plan_create(self._mk(steps), C.byref(pl)))
            self._plans[steps] = pl
        stream = self._torch.cuda.current_stream(self.states.device)
        L_.check(L_.lib().qecmc_pteq_resume_dev(self._plans[steps], self.states.data_ptr(), self.flags.data_ptr(), self.tops0.data_ptr(),
                                                self.N, self.first, self.steps, self.counts.data_ptr(), self.samples.data_ptr(),
                                                C.c_void_p(stream.cuda_stream)))
        self.steps += steps
        return self

    def snapshot(self, states=False):
        self._torch.cuda.synchronize()
        out = dict(steps=self.steps, counts=self.counts.cpu().numpy().view(np.uint32), samples=self.samples.cpu().numpy().view(np.uint32),
                   tops0=self.tops0.cpu().numpy().view(np.uint32))
        if states:
            out["states"] = self.states.cpu().numpy().reshape((self.N, self.Nc) + self.shape)
            out["flags"] = self.flags.cpu().numpy()
        return out

    def close(self):
        for pl in self._plans.values():
            L_.lib().qecmc_plan_destroy(pl)
        self._plans = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def convergence_study(init, p, checkpoints, Nc=None, iters=10, tops_burn=0, seed=0, code=L_.TORIC, chunk=None, **run_kw):
    """Class histograms of the same chains at increasing run lengths (BASELINE config 5: "long-chain convergence study").
    The ladders stay in HBM (LadderRun) and are advanced from checkpoint to checkpoint -- an exact continuation, so the whole
    study costs one run to the last checkpoint; `chunk` bounds the ladder steps of a single launch (default 2^20).
    Returns dict(steps int64[k], counts uint32[k, N, ncls], samples uint32[k, N], percent uint8[k, N, ncls],
    tv float64[k]) with tv = mean total-variation distance of each checkpoint's class distribution to the last one."""
    from .decoders import percent_from_counts
    cps = sorted(int(c) for c in checkpoints)
    chunk = int(chunk or (1 << 20))
    run = LadderRun(init, p, Nc=Nc, iters=iters, tops_burn=tops_burn, seed=seed, code=code, **run_kw)
    snaps = []
    for c in cps:
        while run.steps < c:
            run.advance(min(chunk, c - run.steps))
        snaps.append(run.snapshot())
    run.close()
    counts = np.stack([r["counts"] for r in snaps])
    samples = np.stack([r["samples"] for r in snaps])
    frac = counts / np.maximum(samples, 1)[..., None].astype(np.float64)
    tv = 0.5 * np.abs(frac - frac[-1]).sum(axis=-1).mean(axis=-1)
    return dict(steps=np.array(cps, dtype=np.int64), counts=counts, samples=samples, tops0=np.stack([r["tops0"] for r in snaps]),
                percent=np.stack([percent_from_counts(r["counts"], r["samples"]) for r in snaps]), tv=tv)
