"""Pins the oracle's XZZX / rotated-surface-code stencils, the biased chain (src/mcmc_biased.py) and the
generic ladder / PTEQ against vectors captured from the reference (f1_surf.npz, f2_surf.npz)."""
import os
import random

import numpy as np
import pytest

from oracle import oracle as orc
from conftest import GOLDEN

CODE = {"xzzx": orc.XZZX, "rot": orc.ROTATED}


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


def _stream(seed, n):
    r = random.Random(seed)
    return np.array([r.random() for _ in range(n)], dtype=np.float64)


@pytest.mark.parametrize("name", ["xzzx", "rot"])
@pytest.mark.parametrize("L", [3, 5, 9])
@pytest.mark.parametrize("rep", [0, 1])
def test_f1_surf_stencils(name, L, rep):
    g = _load("f1_surf.npz")
    code = CODE[name]
    t = f"{name}_L{L}_{rep}"
    m = g[f"{t}_m"]
    assert orc.count_errors(m) == int(g[f"{t}_count"])
    assert orc.surf_eq_class(code, m) == int(g[f"{t}_class"])
    assert np.array_equal(orc.surf_syndrome(code, m), g[f"{t}_defects"])
    for i, (r, c, op) in enumerate(g[f"{t}_stab_arg"]):
        new, dE = orc.surf_apply_stabilizer(code, m, int(r), int(c), int(op))
        assert dE == int(g[f"{t}_stab_dE"][i]) and np.array_equal(new, g[f"{t}_stab_new"][i])
        assert np.array_equal(orc.surf_syndrome(code, new), g[f"{t}_defects"])     # stabilizers keep the syndrome
    for i, (op, xp, zp) in enumerate(g[f"{t}_log_arg"]):
        new, dE = orc.surf_apply_logical(code, m, int(op), int(xp), int(zp))
        assert dE == int(g[f"{t}_log_dE"][i]) and np.array_equal(new, g[f"{t}_log_new"][i])
        assert orc.surf_eq_class(code, new) == int(g[f"{t}_log_class"][i])


def test_survey_inline_kats():
    Q = np.array([[0, 3, 0], [1, 0, 2], [0, 0, 3]], dtype=np.uint8)
    assert orc.surf_eq_class(orc.XZZX, Q) == 2 and orc.surf_eq_class(orc.ROTATED, Q) == 0
    new, dE = orc.surf_apply_stabilizer(orc.XZZX, Q, 0, 0, 1)
    assert dE == 1 and new.tolist() == [[1, 0, 0], [2, 1, 2], [0, 0, 3]]
    new, dE = orc.surf_apply_stabilizer(orc.ROTATED, Q, 0, 3, 3)
    assert dE == 1 and new.tolist() == [[3, 3, 0], [2, 0, 2], [0, 0, 3]]
    new, dE = orc.surf_apply_logical(orc.ROTATED, Q, 3, 1, 2)
    assert dE == 2 and orc.surf_eq_class(orc.ROTATED, new) == 3


def _cases(prefix):
    g = _load("f2_surf.npz")
    return [str(c) for c in g["cases"] if str(c).startswith(prefix)]


@pytest.mark.parametrize("case", _cases("chain") + _cases("bchain"))
def test_f2_surf_chain_trajectories(case):
    g = _load("f2_surf.npz")
    ci, L, p, p_logical, iters, seed, ndraw, biased, eta = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    fin = orc.chain_update(orc.XZZX if ci == 0 else orc.ROTATED, g[f"{case}_init"], float(p), float(p_logical), int(iters), rng,
                           noise=int(biased), eta=float(eta))
    assert rng.consumed == int(ndraw)
    assert np.array_equal(fin, g[f"{case}_final"])


@pytest.mark.parametrize("case", _cases("ladder"))
def test_f2_surf_ladder_trajectories(case):
    g = _load("f2_surf.npz")
    ci, L, p, Nc, iters, nstep, seed, ndraw, biased, eta = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    ld = orc.Ladder(orc.XZZX if ci == 0 else orc.ROTATED, g[f"{case}_init"], float(p), int(Nc), 0.5, noise=int(biased), eta=float(eta))
    assert np.array_equal(ld.p_ladder, g[f"{case}_p_ladder"]) and np.array_equal(ld.p_diff, g[f"{case}_p_diff"])
    tops = []
    for _ in range(int(nstep)):
        ld.step(int(iters), rng)
        tops.append(ld.tops0)
    assert rng.consumed == int(ndraw)
    assert tops == g[f"{case}_tops_hist"].tolist()
    assert np.array_equal(ld.states, g[f"{case}_states"]) and np.array_equal(ld.flags, g[f"{case}_flags"])


@pytest.mark.parametrize("case", _cases("pteq"))
def test_f2_surf_pteq_percent(case):
    g = _load("f2_surf.npz")
    ci, L, p, Nc, iters, steps, tops_burn, conv, seed, ndraw, SEQ, TOPS, eps, biased, eta = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    res = orc.pteq(orc.XZZX if ci == 0 else orc.ROTATED, g[f"{case}_init"], float(p), Nc=int(Nc), SEQ=int(SEQ), TOPS=int(TOPS),
                   tops_burn=int(tops_burn), eps=float(eps), steps=int(steps), iters=int(iters),
                   conv_criteria="error_based" if conv else None, rng=rng, noise=int(biased), eta=float(eta))
    assert rng.consumed == int(ndraw)
    assert np.array_equal(res["percent"], g[f"{case}_percent"])
