"""Pins the oracle's "alpha" noise chain / ladder / PTEQ (src/mcmc_alpha.py, decoders_biasednoise.py:175-238) against
trajectories captured from the reference on an injected random stream (f2_alpha.npz)."""
import math
import os
import random

import numpy as np
import pytest

from oracle import oracle as orc
from conftest import GOLDEN


def _load():
    return np.load(os.path.join(GOLDEN, "f2_alpha.npz"))


def _stream(seed, n):
    r = random.Random(seed)
    return np.array([r.random() for _ in range(n)], dtype=np.float64)


def _cases(prefix):
    return [str(c) for c in _load()["cases"] if str(c).startswith(prefix)]


def _code(ci):
    return orc.XZZX if ci == 0 else orc.ROTATED


@pytest.mark.parametrize("case", _cases("achain"))
def test_alpha_chain_trajectories(case):
    g = _load()
    ci, L, pzt, p_logical, iters, seed, ndraw, alpha, n_eff = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    fin, ne = orc.chain_update_alpha(_code(ci), g[f"{case}_init"], float(pzt), float(alpha), float(p_logical), int(iters), rng)
    assert rng.consumed == int(ndraw)
    assert np.array_equal(fin, g[f"{case}_final"]) and ne == n_eff


@pytest.mark.parametrize("det_pow", [0, 1])
@pytest.mark.parametrize("case", _cases("aladder"))
def test_alpha_ladder_trajectories(case, det_pow):
    """det_pow=1 swaps libm pow for the deterministic exp the GPU uses: same decisions on every captured trajectory."""
    g = _load()
    ci, L, pzt, Nc, iters, nstep, seed, ndraw, alpha = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    ld = orc.Ladder(_code(ci), g[f"{case}_init"], float(pzt), int(Nc), 0.5, noise=orc.ALPHA, alpha=float(alpha), det_pow=det_pow)
    assert np.array_equal(ld.p_ladder, g[f"{case}_p_ladder"])
    tops = []; neff = []
    for _ in range(int(nstep)):
        ld.step(int(iters), rng)
        tops.append(ld.tops0); neff.append(ld.n_eff)
    assert rng.consumed == int(ndraw)
    assert tops == g[f"{case}_tops_hist"].tolist()
    assert np.array_equal(np.array(neff), g[f"{case}_neff_hist"])          # incl. the stale slot-bound values (Q4)
    assert np.array_equal(ld.states, g[f"{case}_states"]) and np.array_equal(ld.flags, g[f"{case}_flags"])


@pytest.mark.parametrize("case", _cases("apteq"))
def test_alpha_pteq_percent(case):
    g = _load()
    ci, L, pzt, Nc, iters, steps, tops_burn, conv, seed, ndraw, SEQ, TOPS, eps, alpha = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    res = orc.pteq(_code(ci), g[f"{case}_init"], float(pzt), Nc=int(Nc), SEQ=int(SEQ), TOPS=int(TOPS), tops_burn=int(tops_burn),
                   eps=float(eps), steps=int(steps), iters=int(iters), conv_criteria="error_based" if conv else None, rng=rng,
                   noise=orc.ALPHA, alpha=float(alpha))
    assert rng.consumed == int(ndraw)
    assert np.array_equal(res["percent"], g[f"{case}_percent"])


def test_det_exp_accuracy():
    r = np.random.default_rng(3)
    ys = -np.abs(r.normal(0, 40, 20000))
    d = np.array([orc.lib().orc_det_exp(float(y)) for y in ys])
    assert np.max(np.abs(d - np.exp(ys)) / np.exp(ys)) < 4e-16
    assert orc.lib().orc_det_exp(0.0) == 1.0 and orc.lib().orc_det_exp(3.0) == 1.0 and orc.lib().orc_det_exp(-800.0) == 0.0


@pytest.mark.parametrize("case", [c for c in np.load(os.path.join(GOLDEN, "f_nalpha.npz"))["cases"] if str(c).startswith("short")])
def test_pteq_alpha_with_shortest_loop(case):
    """PTEQ_alpha_with_shortest (decoders_biasednoise.py:93-172): qecmc's host bookkeeping around the oracle's Ladder_alpha on
    the injected stream reproduces the reference's three outputs and consumes the same number of draws."""
    import random
    from qecmc.decoders_biasednoise import _shortest_loop
    from util_shortest import OracleLadderAlpha
    g = np.load(os.path.join(GOLDEN, "f_nalpha.npz"))
    code, L, pzt, alpha, Nc, steps, tops_burn, conv, SEQ, TOPS, eps, seed, ndraw = g[f"{case}_par"]
    r = random.Random(int(seed))
    rng = orc.Rng.stream(np.array([r.random() for _ in range(int(ndraw))], dtype=np.float64))
    ld = OracleLadderAlpha(orc.XZZX if code == 0 else orc.ROTATED, g[f"{case}_init"], float(pzt), float(alpha), int(Nc), rng)
    pct, eqd, sn = _shortest_loop(ld, float(pzt), int(SEQ), int(TOPS), int(tops_burn), float(eps), int(steps), 10,
                                  "error_based" if conv else None)
    assert rng.consumed == int(ndraw)
    assert np.array_equal(pct, g[f"{case}_percent"])
    assert np.allclose(eqd, g[f"{case}_eqdistr"], rtol=1e-12, atol=0, equal_nan=True)
    assert np.allclose(sn, g[f"{case}_shortn"], rtol=1e-12, atol=0, equal_nan=True)
