"""Pins the oracle's unique-chain estimator (PTDC_droplet / PTDC, decoders.py:138-233) against the reference on an
injected stream (f_ptdc.npz): N(n) = number of distinct chains of each length found by a class ladder, and PTDC's
percent vector."""
import os
import random

import numpy as np
import pytest

from oracle import oracle as orc
from conftest import GOLDEN


def _load():
    return np.load(os.path.join(GOLDEN, "f_ptdc.npz"))


def _stream(seed, n):
    r = random.Random(seed)
    return np.array([r.random() for _ in range(n)], dtype=np.float64)


def _cases(prefix):
    return [str(c) for c in _load()["cases"] if str(c).startswith(prefix)]


@pytest.mark.parametrize("case", _cases("drop"))
def test_ptdc_droplet_length_histogram(case):
    g = _load()
    code, L, p, Nc, steps, seed, ndraw = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    hist, tab = orc.ptdc_droplet(int(code), g[f"{case}_init"], float(p), int(Nc), int(steps), rng=rng)
    assert rng.consumed == int(ndraw)
    assert np.array_equal(hist, g[f"{case}_hist"])
    assert int((tab != 0).sum()) == int(hist.sum())


@pytest.mark.parametrize("case", _cases("ptdc"))
def test_ptdc_percent(case):
    """PTDC with droplets = 1 runs the 16 class ladders one after the other on the same stream (decoders.py:217-219)."""
    g = _load()
    L, p_error, p_sampling, Nc, steps, seed, ndraw = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    reps = g[f"{case}_classes"]
    assert [orc.toric_eq_class(r) for r in reps] == list(range(16))
    hist = np.stack([orc.ptdc_droplet(orc.TORIC, r, float(p_sampling), int(Nc), int(steps) // int(Nc), rng=rng)[0] for r in reps])
    assert rng.consumed == int(ndraw)
    pct = orc.ptdc_distribution(hist, float(p_error)).astype(np.uint8)
    assert np.array_equal(pct, g[f"{case}_percent"])


@pytest.mark.parametrize("case", _cases("sdrop"))
def test_stdc_droplet_length_histogram(case):
    """STDC_droplet (decoders.py:236-265) = the 1-chain ladder, 5 proposals per step (`update_chain_fast(5)`, :250)."""
    g = _load()
    L, p, steps, seed, ndraw = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    hist, _ = orc.ptdc_droplet(orc.PLANAR, g[f"{case}_init"], float(p), 1, int(steps), iters=5, rng=rng)
    assert rng.consumed == int(ndraw)
    assert np.array_equal(hist, g[f"{case}_hist"])


@pytest.mark.parametrize("case", _cases("stdc"))
def test_stdc_distribution(case):
    g = _load()
    L, p_error, p_sampling, steps, seed, ndraw = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    reps = g[f"{case}_classes"]
    assert [orc.surf_eq_class(orc.PLANAR, r) for r in reps] == [0, 1, 2, 3]
    hist = np.stack([orc.ptdc_droplet(orc.PLANAR, r, float(p_sampling), 1, int(steps), iters=5, rng=rng)[0] for r in reps])
    assert rng.consumed == int(ndraw)
    assert np.allclose(orc.ptdc_distribution(hist, float(p_error)), g[f"{case}_dist"], rtol=1e-12, atol=0)


@pytest.mark.parametrize("case", _cases("strc"))
def test_strc_distribution(case):
    """STRC (decoders.py:745-949): N(n) at the two shortest lengths and m(n) of the single chains, then the host formula
    (qecmc.decoders.strc_distribution, pure NumPy)."""
    from qecmc.decoders import strc_distribution
    g = _load()
    L, p_error, p_sampling, steps, seed, ndraw = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    res = [orc.ptdc_droplet(orc.PLANAR, r, float(p_sampling), 1, int(steps), iters=5, rng=rng, with_m=True) for r in g[f"{case}_classes"]]
    assert rng.consumed == int(ndraw)
    dist = strc_distribution(np.stack([r[0] for r in res]), np.stack([r[1] for r in res]), float(p_error), float(p_sampling))
    assert np.allclose(dist, g[f"{case}_dist"], rtol=1e-12, atol=0)


@pytest.mark.parametrize("case", _cases("ptrc"))
def test_ptrc_percent(case):
    """PTRC (decoders.py:584-742): per-rung N(n) and m(n) of the class ladders, then qecmc.decoders.ptrc_distribution."""
    from qecmc.decoders import ptrc_distribution
    g = _load()
    L, p_error, p_sampling, Nc, steps, seed, ndraw = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    res = [orc.ptdc_droplet(orc.TORIC, r, float(p_sampling), int(Nc), int(steps) // int(Nc), rng=rng, per_rung=True, with_m=True)
           for r in g[f"{case}_classes"]]
    assert rng.consumed == int(ndraw)
    pct = ptrc_distribution(np.stack([r[0] for r in res])[:, None], np.stack([r[1] for r in res])[:, None], float(p_error), float(p_sampling))
    assert np.array_equal(pct, g[f"{case}_percent"])


@pytest.mark.parametrize("case", _cases("stemp"))
def test_single_temp_means(case):
    """single_temp (decoders.py:108-135): mean chain length over the first max_iters - 1 steps; the reference runs (and
    draws for) max_iters steps per class."""
    g = _load()
    L, p, max_iters, seed, ndraw = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    means = []
    for r in g[f"{case}_classes"]:
        _, m_o, _ = orc.ptdc_droplet(orc.PLANAR, r, float(p), 1, int(max_iters) - 1, iters=5, rng=rng, with_m=True)
        orc.chain_update(orc.PLANAR, r, float(p), 0.0, 5, rng)      # the last step's draws (its length is not averaged)
        means.append((m_o * np.arange(m_o.size)).sum() / (int(max_iters) - 1))
    assert rng.consumed == int(ndraw)
    assert np.allclose(means, g[f"{case}_means"], rtol=1e-12)


def test_state_key_and_set():
    r = np.random.default_rng(0)
    states = r.integers(0, 4, size=(2000, 50), dtype=np.uint8)
    keys = {int(orc.lib().orc_state_key(orc._u8(s), s.size)) for s in states}
    assert len(keys) == len({s.tobytes() for s in states}) and 0 not in keys


# ---- the conv_mult early stop (decoders.py:153-162, :256-262, :783-826), f_convmult.npz ----------------------------------------------

def _loadc():
    return np.load(os.path.join(GOLDEN, "f_convmult.npz"))


def _casesc(prefix):
    return [str(c) for c in _loadc()["cases"] if str(c).startswith(prefix)]


@pytest.mark.parametrize("case", _casesc("drop"))
def test_convmult_ptdc_droplet(case):
    """The draws consumed pin the step at which the droplet stopped; N(n) pins what it had seen by then."""
    g = _loadc()
    code, L, p, Nc, steps, seed, ndraw, cm = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    hist, _ = orc.ptdc_droplet(int(code), g[f"{case}_init"], float(p), int(Nc), int(steps), rng=rng, conv_mult=float(cm))
    assert rng.consumed == int(ndraw)
    assert np.array_equal(hist, g[f"{case}_hist"])
    full, _ = orc.ptdc_droplet(int(code), g[f"{case}_init"], float(p), int(Nc), int(steps), rng=orc.Rng.stream(_stream(int(seed), 10 ** 5)))
    assert full.sum() > hist.sum()                       # the stop really cut the run short


@pytest.mark.parametrize("case", _casesc("sdrop"))
def test_convmult_stdc_droplet(case):
    g = _loadc()
    L, p, steps, seed, ndraw, cm = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    hist, _ = orc.ptdc_droplet(orc.PLANAR, g[f"{case}_init"], float(p), 1, int(steps), iters=5, rng=rng, conv_mult=float(cm))
    assert rng.consumed == int(ndraw)
    assert np.array_equal(hist, g[f"{case}_hist"])


@pytest.mark.parametrize("case", _casesc("st"))
def test_convmult_stdc_strc_distribution(case):
    from qecmc.decoders import strc_distribution
    g = _loadc()
    L, p_error, p_sampling, steps, seed, ndraw, cm = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    res = [orc.ptdc_droplet(orc.PLANAR, r, float(p_sampling), 1, int(steps), iters=5, rng=rng, with_m=True, conv_mult=float(cm))
           for r in g[f"{case}_classes"]]
    assert rng.consumed == int(ndraw)
    hist, mh = np.stack([r[0] for r in res]), np.stack([r[1] for r in res])
    dist = orc.ptdc_distribution(hist, float(p_error)) if case.startswith("stdc") else \
        strc_distribution(hist, mh, float(p_error), float(p_sampling))
    assert np.allclose(dist, g[f"{case}_dist"], rtol=1e-12, atol=0)


# ---- (n_x, n_y, n_z) of the distinct chains: STDC_general_noise family and Chain_xyz (decoders.py:325-507, mcmc.py:106-114), f_xyz.npz ----

def _loadx():
    return np.load(os.path.join(GOLDEN, "f_xyz.npz"))


def _casesx(prefix):
    return [str(c) for c in _loadx()["cases"] if str(c).startswith(prefix)]


@pytest.mark.parametrize("case", _casesx("gdrop"))
def test_general_noise_droplet_xyz(case):
    """STDC_droplet_general_noise's dict values, in the order the chains were found; a 3-vector p = Chain_xyz sampling."""
    g = _loadx()
    L, steps, seed, ndraw = g[f"{case}_par"]
    p = g[f"{case}_p"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    hist, xyz, _ = orc.ptdc_droplet(orc.PLANAR, g[f"{case}_init"], p if p.size == 3 else float(p[0]), 1, int(steps), iters=5, rng=rng,
                                    with_xyz=True)
    assert rng.consumed == int(ndraw)
    assert np.array_equal(xyz, g[f"{case}_xyz"])
    assert np.array_equal(np.bincount(xyz.sum(axis=1), minlength=hist.size), hist)


@pytest.mark.parametrize("case", _casesx("gn"))
def test_general_noise_distributions(case):
    from qecmc.decoders import general_noise_distribution
    g = _loadx()
    L, steps, seed, ndraw = g[f"{case}_par"]
    ps = g[f"{case}_ps"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    xyz = [orc.ptdc_droplet(orc.PLANAR, r, ps if ps.size == 3 else float(ps[0]), 1, int(steps), iters=5, rng=rng, with_xyz=True)[1]
           for r in g[f"{case}_classes"]]
    assert rng.consumed == int(ndraw)
    p_xyz = g[f"{case}_pxyz"]
    assert np.allclose(general_noise_distribution(xyz, p_xyz), g[f"{case}_all"], rtol=1e-12, atol=0)
    assert np.allclose(general_noise_distribution(xyz, p_xyz, shortest_only=True), g[f"{case}_short"], rtol=1e-12, atol=0)
    both = g[f"{case}_both"]
    assert np.allclose(general_noise_distribution(xyz, p_xyz), both[0], rtol=1e-12, atol=0)
    assert np.allclose(general_noise_distribution(xyz, p_xyz, shortest_only=True), both[1], rtol=1e-12, atol=0)


# ---- STDC_droplet_alpha / STDC_Nall_n_alpha (decoders.py:510-581): Chain_alpha sampling, weights n_z + alpha (n_x + n_y), f_nalpha.npz ----

def _loada():
    return np.load(os.path.join(GOLDEN, "f_nalpha.npz"))


def _casesa(prefix):
    return [str(c) for c in _loada()["cases"] if str(c).startswith(prefix)]


@pytest.mark.parametrize("case", _casesa("adrop"))
def test_alpha_droplet_effective_lengths(case):
    g = _loada()
    code, L, pzt, alpha, steps, seed, ndraw = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    _, xyz, _ = orc.ptdc_droplet(orc.XZZX if code == 0 else orc.ROTATED, g[f"{case}_init"], float(pzt), 1, int(steps), iters=5, rng=rng,
                                 with_xyz=True, alpha=float(alpha))
    assert rng.consumed == int(ndraw)
    assert np.array_equal(xyz[:, 2] + float(alpha) * (xyz[:, 0] + xyz[:, 1]), g[f"{case}_eff"])     # decoders.py:522


@pytest.mark.parametrize("case", _casesa("nall"))
def test_nall_n_alpha_distribution(case):
    from qecmc.decoders import nall_n_alpha_distribution
    g = _loada()
    code, L, pzs, alpha, pzt, steps, seed, ndraw = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    xyz = [orc.ptdc_droplet(orc.XZZX if code == 0 else orc.ROTATED, r, float(pzs), 1, int(steps), iters=5, rng=rng, with_xyz=True,
                            alpha=float(alpha))[1] for r in g[f"{case}_classes"]]
    assert rng.consumed == int(ndraw)
    assert np.allclose(nall_n_alpha_distribution(xyz, float(alpha), float(pzt)), g[f"{case}_dist"], rtol=1e-12, atol=0)
