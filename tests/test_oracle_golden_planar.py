"""Pins the oracle's planar-code stencils (src/planar_model.py) and the generic chain / ladder / PTEQ on that code against
vectors captured from the reference (f_planar.npz), incl. `Chain.update_chain_fast`, which the reference hard-wires to
this stencil (mcmc.py:6,152-160)."""
import os
import random

import numpy as np
import pytest

from oracle import oracle as orc
from conftest import GOLDEN


def _load():
    return np.load(os.path.join(GOLDEN, "f_planar.npz"))


def _stream(seed, n):
    r = random.Random(seed)
    return np.array([r.random() for _ in range(n)], dtype=np.float64)


def _cases(prefix):
    return [str(c) for c in _load()["cases"] if str(c).startswith(prefix)]


@pytest.mark.parametrize("t", [str(k) for k in _load()["kats"]])
def test_planar_stencils(t):
    g = _load()
    m = g[f"{t}_m"]
    L = m.shape[-1]
    assert orc.count_errors(m) == int(g[f"{t}_count"])
    assert orc.surf_eq_class(orc.PLANAR, m) == int(g[f"{t}_class"])
    v, q = orc.planar_syndrome(m)
    assert np.array_equal(v, g[f"{t}_vertex"].astype(bool)) and np.array_equal(q, g[f"{t}_plaquette"].astype(bool))
    for i, (r, c, op) in enumerate(g[f"{t}_stab_arg"]):
        new, dE = orc.surf_apply_stabilizer(orc.PLANAR, m, int(r), int(c), int(op))
        assert dE == int(g[f"{t}_stab_dE"][i]) and np.array_equal(new, g[f"{t}_stab_new"][i])
        v2, q2 = orc.planar_syndrome(new)
        assert np.array_equal(v2, v) and np.array_equal(q2, q)                 # stabilizers keep the syndrome
    for i, (op, xp, zp) in enumerate(g[f"{t}_log_arg"]):
        new, dE = orc.surf_apply_logical(orc.PLANAR, m, int(op), int(xp), int(zp))
        assert dE == int(g[f"{t}_log_dE"][i]) and np.array_equal(new, g[f"{t}_log_new"][i])
        assert orc.surf_eq_class(orc.PLANAR, new) == int(g[f"{t}_log_class"][i])
    # the generator order of the sweep / the one-word pick covers every (row, col, op) of the reference exactly once
    seen = {orc.surf_gen_rco(orc.PLANAR, L, k) for k in range(orc.surf_ngen(orc.PLANAR, L))}
    assert seen == {tuple(int(x) for x in a) for a in g[f"{t}_stab_arg"]} and len(seen) == 2 * L * (L - 1)


@pytest.mark.parametrize("case", _cases("chain"))
def test_planar_chain_trajectories(case):
    g = _load()
    L, p, p_logical, iters, seed, ndraw, fast = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    fin = g[f"{case}_init"]
    if fast:                                          # update_chain_fast(5) x iters/5 = the same chain, proposal by proposal
        for _ in range(int(iters) // 5):
            fin = orc.chain_update(orc.PLANAR, fin, float(p), 0.0, 5, rng)
    else:
        fin = orc.chain_update(orc.PLANAR, fin, float(p), float(p_logical), int(iters), rng)
    assert rng.consumed == int(ndraw)
    assert np.array_equal(fin, g[f"{case}_final"])


@pytest.mark.parametrize("case", _cases("ladder"))
def test_planar_ladder_trajectories(case):
    g = _load()
    L, p, Nc, iters, nstep, seed, ndraw = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    ld = orc.Ladder(orc.PLANAR, g[f"{case}_init"], float(p), int(Nc), 0.5)
    tops = []
    for _ in range(int(nstep)):
        ld.step(int(iters), rng)
        tops.append(ld.tops0)
    assert rng.consumed == int(ndraw)
    assert tops == g[f"{case}_tops_hist"].tolist()
    assert np.array_equal(ld.states, g[f"{case}_states"]) and np.array_equal(ld.flags, g[f"{case}_flags"])


@pytest.mark.parametrize("case", _cases("pteq"))
def test_planar_pteq_percent(case):
    g = _load()
    L, p, Nc, iters, steps, tops_burn, conv, seed, ndraw, SEQ, TOPS, eps = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    res = orc.pteq(orc.PLANAR, g[f"{case}_init"], float(p), Nc=int(Nc), SEQ=int(SEQ), TOPS=int(TOPS), tops_burn=int(tops_burn),
                   eps=float(eps), steps=int(steps), iters=int(iters), conv_criteria="error_based" if conv else None, rng=rng)
    assert rng.consumed == int(ndraw)
    assert np.array_equal(res["percent"], g[f"{case}_percent"])
