"""Equivalence-class histograms of the HIP path against reference-independent exact enumeration
(toric L=3) and against the reference's own replica-averaged histograms (fixture F3).  The GPU
is also bit-identical to the oracle (test_gpu_parity.py); these tests close the loop to the
physics: the sampled class distribution is the right one."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from util_exact import toric_class_probabilities

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def q():
    import qecmc
    assert qecmc.device_count() >= 1
    return qecmc


def _rand_state(seed, L, p):
    rng = np.random.default_rng(seed)
    m = np.zeros((2, L, L), dtype=np.uint8)
    err = rng.random(m.shape) < p
    m[err] = rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)
    return m


def _mean_sem(frac, ok, scan):
    """mean and standard error over the replicas; scan = "wave": the 64 replicas of a wavefront share their generator picks, so the
    error is taken over the 64 wavefront means (valid whatever the correlation inside a wavefront)"""
    if scan != "wave":
        f = frac[ok]
        return f.mean(axis=0), f.std(axis=0, ddof=1) / np.sqrt(ok.sum())
    g = np.array([frac[i:i + 64][ok[i:i + 64]].mean(axis=0) for i in range(0, len(frac), 64) if ok[i:i + 64].any()])
    return frac[ok].mean(axis=0), g.std(axis=0, ddof=1) / np.sqrt(len(g))


@pytest.mark.parametrize("scan", ["random", "sweep", "colour", "wave"])
@pytest.mark.parametrize("seed,p,Nc", [(1, 0.10, 3), (2, 0.15, 4), (3, 0.12, 4), (4, 0.20, 5)])
def test_exact_enumeration_L3(q, seed, p, Nc, scan):
    from qecmc import toric_model as tm
    init = _rand_state(seed, 3, 0.15)
    P = toric_class_probabilities(init, p, tm.apply_stabilizer, tm.to_class)      # device stencils build the group
    R, steps = 4096, 4000
    res = q.pteq_batch(np.broadcast_to(init, (R,) + init.shape).copy(), p, Nc=Nc, steps=steps, iters=10, tops_burn=5,
                       seed=1000 + seed, scan=scan)
    ok = res["samples"] > steps // 2                  # replicas whose burn-in (tops0 >= 5) ended in the first half
    assert ok.mean() > 0.97
    mean, sem = _mean_sem(res["counts"] / np.maximum(res["samples"], 1)[:, None].astype(np.float64), ok, scan)
    # within Monte-Carlo error of the exact answer (4096 replicas: sem ~ 1e-3); the 2e-4 floor absorbs the
    # residual burn-in transient of a finite run
    assert np.all(np.abs(mean - P) <= 5 * sem + 2e-4), (mean, P, sem)
    assert mean.argmax() == P.argmax()


@pytest.mark.parametrize("name", ["L3", "L5"])
def test_reference_histograms_f3(q, name):
    """The reference's own runs (fixture F3/F5: R=16 replicas per syndrome, 5000 recorded ladder steps) vs
    the GPU (1024 replicas, every step recorded).

    * per-rung mean error counts equilibrate fast and are compared strictly (combined standard error);
    * class histograms: rare class-to-class excursions make the reference's 16 short replicas heavy-tailed
      (e.g. L3 syndrome 0: exact P(class 0) = 0.0357, GPU 0.0346, reference replicas 0.002 ... 0.058 with
      mean 0.0107), so its sample variance understates its error: the comparison allows 3 % absolute on top
      of the combined standard error and requires the same most-likely class.  The strict histogram check is
      test_exact_enumeration_L3.
    """
    from qecmc import toric_model as tm
    g = np.load(os.path.join(GOLDEN, "f3_toric.npz"))
    L, p, Nc, iters, steps, burn = g[f"{name}_par"]
    for s in range(g[f"{name}_init"].shape[0]):
        init = g[f"{name}_init"][s]
        ref = g[f"{name}_hist"][s] / (steps - burn)
        ref_n = g[f"{name}_nerr"][s]
        R = 1024
        res = q.pteq_batch(np.broadcast_to(init, (R,) + init.shape).copy(), float(p), Nc=int(Nc), steps=int(steps),
                           iters=int(iters), tops_burn=0, seed=4242 + s, return_states=True)
        ok = res["samples"] == int(steps)          # tops_burn=0: every step is recorded (the 3 % allowance covers the transient)
        assert ok.all()
        frac = res["counts"][ok] / res["samples"][ok, None].astype(np.float64)
        se = np.sqrt(ref.var(axis=0, ddof=1) / ref.shape[0] + frac.var(axis=0, ddof=1) / ok.sum())
        assert np.all(np.abs(ref.mean(axis=0) - frac.mean(axis=0)) <= 4.5 * se + 0.03), (s, ref.mean(0), frac.mean(0), se)
        assert ref.mean(axis=0).argmax() == frac.mean(axis=0).argmax()
        # F5: per-rung <n_errors>; the GPU side is the ensemble mean over replicas at the final step
        n_fin = np.stack([tm.count_errors(np.ascontiguousarray(res["states"][:, c])) for c in range(int(Nc))], axis=1)
        se_n = np.sqrt(ref_n.var(axis=0, ddof=1) / ref_n.shape[0] + n_fin.var(axis=0, ddof=1) / R)
        assert np.all(np.abs(ref_n.mean(axis=0) - n_fin.mean(axis=0)) <= 4.5 * se_n + 0.05), (s, ref_n.mean(0), n_fin.mean(0))


def test_harness_generate_decodes_low_noise(q, tmp_path):
    """generate_data.generate's recipe, batched: at low p the most likely class is the true one.  The toric
    case runs the reference's default convergence criterion (a fixed 3000-step run from a hidden class has
    not mixed between the 16 classes yet at p = 0.05: the burn-in trap of SURVEY A10)."""
    from qecmc import harness
    for params, kw in (({"code": "toric", "size": 5, "p_error": 0.08, "noise": "depolarizing"}, dict(steps=200000)),
                       ({"code": "rotated", "size": 5, "p_error": 0.05, "noise": "depolarizing"},
                        dict(steps=3000, conv_criteria=None, tops_burn=0)),
                       ({"code": "xzzx", "size": 5, "p_error": 0.05, "noise": "biased", "eta": 10},
                        dict(steps=3000, conv_criteria=None, tops_burn=0, biased_decoder="biased")),      # PTEQ_biased
                       ({"code": "xzzx", "size": 5, "p_error": 0.05, "noise": "biased", "eta": 10},
                        dict(steps=3000, conv_criteria=None, tops_burn=0)),     # PTEQ_alpha, as generate_data.py:142-150 routes it
                       ({"code": "planar", "size": 5, "p_error": 0.03, "noise": "depolarizing"},
                        dict(steps=3000, conv_criteria=None, tops_burn=0)),
                       ({"code": "rotated", "size": 5, "p_error": 0.04, "noise": "alpha", "alpha": 2.0},
                        dict(steps=200000)),                                    # PTEQ_alpha's default error_based criterion
                       ({"code": "xzzx", "size": 5, "p_error": 0.05, "noise": "biased", "eta": 10},
                        dict(steps=200000, scan="wave")),                       # ... on the scan = wave kernels and their work queue
                       ({"code": "toric", "size": 5, "p_error": 0.08, "noise": "depolarizing"}, dict(steps=200000, scan="wave"))):
        f = tmp_path / (params["code"] + ".npz")
        out = harness.generate(params, 256, seed=3, file_path=str(f), **kw)
        assert out["distr"].shape == (256, 16 if params["code"] == "toric" else 4)
        # the reference's criterion can stop before the 16 toric classes have mixed ("argmax differed
        # between repeats"), so the toric bar is lower than for the 4-class codes
        assert out["success"].mean() > (0.75 if params["code"] == "toric" else 0.9), (params, out["success"].mean())
        back = np.load(f)
        assert np.array_equal(back["eq_true"], out["eq_true"]) and back["qubit_matrix"].dtype == np.uint8


def test_convergence_study_prefix_property(q):
    """A shorter run is the exact prefix of a longer one (counter-based RNG): with tops_burn = 0 every step is recorded, so
    the counts are monotone in the run length, sample counts equal the step counts, and the distance to the longest
    run's distribution shrinks."""
    from qecmc import harness
    rng = np.random.default_rng(9)
    raw = harness.draw_errors("rotated", 7, 96, 0.12, rng)
    out = harness.convergence_study(raw, 0.12, [100, 400, 1600, 6400], Nc=7, seed=4, code=q.ROTATED)
    assert out["counts"].shape == (4, 96, 4) and np.array_equal(out["samples"], np.broadcast_to(out["steps"][:, None], (4, 96)))
    assert np.all(np.diff(out["counts"].astype(np.int64), axis=0) >= 0)
    assert out["tv"][0] > out["tv"][2] and out["tv"][-1] == 0


def test_harness_class_representatives_and_rain(q):
    from qecmc import harness, planar_model, toric_model, _surf
    rng = np.random.default_rng(2)
    for name, L in (("toric", 5), ("planar", 5), ("xzzx", 5), ("rotated", 7)):
        code = harness._CODES[name]
        raw = harness.draw_errors(name, L, 20, 0.1, rng)
        reps = harness.class_representatives(name, raw)
        ncls = 16 if name == "toric" else 4
        assert reps.shape[:2] == (20, ncls)
        syn = (lambda m: toric_model.syndrome(m)) if name == "toric" else (lambda m: np.concatenate([d.reshape(len(m), -1) for d in planar_model.syndrome(m)], axis=1)) \
            if name == "planar" else (lambda m: _surf.syndrome(code, m))
        for c in range(ncls):
            assert np.array_equal(np.asarray(harness._class_of(code, reps[:, c])), np.full(20, c))
            assert np.array_equal(syn(np.ascontiguousarray(reps[:, c])), syn(raw))
        if name in ("toric", "planar"):
            wet = harness.rain(name, raw, rng)
            assert np.array_equal(syn(wet), syn(raw)) and np.array_equal(harness._class_of(code, wet), harness._class_of(code, raw))
            assert (wet != raw).any()


@pytest.mark.parametrize("method,params,steps", [
    ("PTDC", dict(code="toric", size=3, p_error=0.05, Nc=3, droplets=2), 1500),
    ("PTRC", dict(code="toric", size=3, p_error=0.05, Nc=3, droplets=2), 1500),
    ("STDC", dict(code="planar", size=3, p_error=0.05, p_sampling=0.2, droplets=3, conv_mult=2.0), 1500),
    ("STRC", dict(code="toric", size=3, p_error=0.05, p_sampling=0.2, droplets=3), 1500),
    ("STDC_N_n", dict(code="xzzx", size=5, p_error=0.05, noise="alpha", alpha=2.0, p_sampling=0.2), 1500)])
def test_harness_unique_chain_methods(q, method, params, steps):
    """generate_data.py:168-196 on a batch: the estimators decode low-noise syndromes, chunking does not change the result."""
    from qecmc import harness
    params = dict(params, method=method)
    out = harness.generate(params, 48, seed=5, steps=steps)
    ncls = 16 if params["code"] == "toric" else 4
    tot = out["distr"].sum(axis=1)                          # (PTRC returns the truncated uint8 percent vector, decoders.py:742)
    assert out["distr"].shape == (48, ncls) and np.all(tot <= 100 + 1e-9) and np.all(tot > (100 - ncls if method == "PTRC" else 100 - 1e-9))
    assert out["success"].mean() > 0.9
    if method != "STDC" and method != "STRC":                 # (rain draws from the host stream, which chunking reorders)
        again = harness.generate(params, 48, seed=5, steps=steps, batch=20)
        assert np.array_equal(again["distr"], out["distr"])


# ---- exact pins for the plaquette codes and the biased weights (the CPU twins: tests/test_stats_cpu.py) -------
# Reference-independent: the stabilizer group of an L = 3 xzzx / rotated syndrome has 2^8 elements x 4 classes; the class law
# is their weights summed.  4096 replicas, 5 sigma -- the sharp statistical pin on the Philox re-parametrisation of these paths
# (one 20-bit generator pick instead of five draws, the packed top-chain words, the 12-bit acceptance lead).

def _surf_api(q):
    import types
    from qecmc import _surf
    from oracle import oracle as orc     # (only the generator ordering, to span the group; the stencils are the device's)
    return types.SimpleNamespace(apply_stabilizer=_surf.apply_stabilizer, apply_logical=_surf.apply_logical, eq_class=_surf.eq_class,
                                 ngen=orc.surf_ngen, gen_rco=orc.surf_gen_rco)


def _rand_surf(seed, L=3, p=0.3):
    rng = np.random.default_rng(seed)
    return (rng.integers(1, 4, size=(L, L)) * (rng.random((L, L)) < p)).astype(np.uint8)


def _class_fractions(res, ok):
    frac = res["counts"][ok] / res["samples"][ok, None].astype(np.float64)
    return frac.mean(axis=0), frac.std(axis=0, ddof=1) / np.sqrt(ok.sum())


@pytest.mark.parametrize("scan", ["random", "colour", "wave"])
@pytest.mark.parametrize("name,seed,p,Nc", [("xzzx", 11, 0.20, 3), ("xzzx", 13, 0.15, 4), ("rotated", 12, 0.25, 4), ("rotated", 14, 0.17, 3)])
def test_plaquette_depolarizing_exact_L3(q, name, seed, p, Nc, scan):
    from util_exact import SurfEnumeration, depolarizing_weight
    code = {"xzzx": q.XZZX, "rotated": q.ROTATED}[name]
    init = _rand_surf(seed)
    P = SurfEnumeration(code, init, _surf_api(q)).class_probabilities(depolarizing_weight(p))
    R, steps = 4096, 4000
    res = q.pteq_batch(np.broadcast_to(init, (R,) + init.shape).copy(), p, Nc=Nc, steps=steps, iters=10, tops_burn=5, seed=2000 + seed, code=code, scan=scan)
    ok = res["samples"] > steps // 2
    assert ok.mean() > 0.97
    mean, sem = _mean_sem(res["counts"] / np.maximum(res["samples"], 1)[:, None].astype(np.float64), ok, scan)
    assert np.all(np.abs(mean - P) <= 5 * sem + 2e-4), (mean, P, sem)
    if np.sort(P)[-1] - np.sort(P)[-2] > 0.01:        # (seed 13 draws the empty lattice at L = 3: the four classes tie exactly)
        assert mean.argmax() == P.argmax()


@pytest.mark.parametrize("name,seed,p,eta", [("xzzx", 21, 0.25, 3.0), ("xzzx", 23, 0.15, 100.0), ("rotated", 22, 0.30, 10.0)])
def test_biased_ladder_exact_L3_iters1(q, name, seed, p, eta):
    """Ladder_biased at iters = 1, where quirk Q3 is vacuous: each rung is a Metropolis chain for px^nx py^ny pz^nz pI^nI
    (mcmc_biased.py:25-31) and the swap rule on total counts is the exact exchange ratio for those weights (the eta factors do
    not depend on the rung), so the bottom rung samples the biased class law."""
    from util_exact import SurfEnumeration, biased_weight
    code = {"xzzx": q.XZZX, "rotated": q.ROTATED}[name]
    init = _rand_surf(seed)
    P = SurfEnumeration(code, init, _surf_api(q)).class_probabilities(biased_weight(p, eta))
    R, steps = 4096, 40000
    # (tops_burn = 50: with the reference's handful of tops the histogram still starts in the class the first arrivals from the top
    # happened to bring -- a transient of 36 / steps on P(class 1) in the first case, 25 sigma here, gone by 50 tops)
    res = q.pteq_batch(np.broadcast_to(init, (R,) + init.shape).copy(), p, Nc=3, steps=steps, iters=1, tops_burn=50, seed=3000 + seed, code=code, eta=eta)
    ok = res["samples"] > steps // 2
    assert ok.mean() > 0.9
    mean, sem = _class_fractions(res, ok)
    assert np.all(np.abs(mean - P) <= 5 * sem + 2e-4), (mean, P, sem)
    assert mean.argmax() == P.argmax()


@pytest.mark.parametrize("name,L,seed,pzt,alpha,Nc", [("xzzx", 3, 41, 0.30, 2.0, 3), ("rotated", 3, 42, 0.25, 3.0, 4), ("xzzx", 5, 43, 0.175, 4.04, 5)])
def test_alpha_ladder_wave_scan_has_the_random_scans_law(q, name, L, seed, pzt, alpha, Nc):
    """Ladder_alpha has no closed-form law to pin (its swap test reads attributes that lag behind the codes, Q4), but whatever law the
    reference's chain has, scan = wave has it per syndrome: the pick is independent of the state, so sharing it among the ladders of a
    wavefront changes no ladder's transition kernel.  4 096 replicas of one syndrome under both scans: class fractions and the bottom rung's
    mean n_z, n_x + n_y agree within Monte-Carlo error (wave: the error over the 64 wavefront means)."""
    code = {"xzzx": q.XZZX, "rotated": q.ROTATED}[name]
    init = _rand_surf(seed, L, 0.2)
    R, steps = 4096, 3000
    out = {}
    for scan in ("random", "wave"):
        res = q.pteq_batch(np.broadcast_to(init, (R,) + init.shape).copy(), pzt, Nc=Nc, steps=steps, iters=10, tops_burn=5, seed=5000 + seed, code=code,
                           alpha=alpha, scan=scan, return_states=True)
        ok = res["samples"] > steps // 2
        assert ok.mean() > 0.9
        frac = res["counts"] / np.maximum(res["samples"], 1)[:, None].astype(np.float64)
        bottom = res["states"][:, 0].reshape(R, -1)
        obs = np.concatenate([frac, (bottom == 3).sum(1, keepdims=True), ((bottom == 1) | (bottom == 2)).sum(1, keepdims=True)], axis=1)
        out[scan] = _mean_sem(obs, ok, scan)
    (m0, s0), (m1, s1) = out["random"], out["wave"]
    assert np.all(np.abs(m0 - m1) <= 5 * np.sqrt(s0 ** 2 + s1 ** 2) + 2e-4), (m0, m1, s0, s1)
    assert m0[:4].argmax() == m1[:4].argmax()


@pytest.mark.parametrize("name,seed,p,eta", [("xzzx", 31, 0.25, 3.0), ("rotated", 32, 0.30, 10.0)])
def test_biased_chain_q3_law_L3_iters10(q, name, seed, p, eta):
    """What iters = 10 converges to instead (quirk Q3: every proposal of an update_chain call is tested against the configuration
    at the START of the call): not the biased law, but a law that is still exactly computable for a single chain
    (SurfEnumeration.q3_class_law).  The sampler sits on it -- and measurably off the biased law."""
    from util_exact import SurfEnumeration, biased_weight
    code = {"xzzx": q.XZZX, "rotated": q.ROTATED}[name]
    init = _rand_surf(seed)
    e = SurfEnumeration(code, init, _surf_api(q))
    w = biased_weight(p, eta)
    Q, P = e.q3_class_law(w, 0.5, 10), e.class_probabilities(w)
    assert 0.5 * np.abs(Q - P).sum() > 0.02
    R, steps = 4096, 6000
    # (a 1-rung ladder counts every step as a top: tops_burn = 500 discards the first 500 calls, the transient from the seed)
    res = q.pteq_batch(np.broadcast_to(init, (R,) + init.shape).copy(), p, Nc=1, steps=steps, iters=10, tops_burn=500, seed=4000 + seed, code=code, eta=eta)
    ok = res["samples"] == steps - 499
    assert ok.all()
    mean, sem = _class_fractions(res, ok)
    assert np.all(np.abs(mean - Q) <= 5 * sem + 2e-4), (mean, Q, sem)
    assert np.abs(mean - P).max() > 10 * sem.max()
