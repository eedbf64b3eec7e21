"""Pins the CPU oracle against vectors captured from the reference itself
(tests/golden/gen_golden.py).  Exact equality everywhere: these are integer /
byte results, and the stream-injected trajectories fix the draw order."""
import os
import random

import numpy as np
import pytest

from oracle import oracle as orc

from conftest import GOLDEN


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


def _stream(seed, n):
    r = random.Random(seed)
    return np.array([r.random() for _ in range(n)], dtype=np.float64)


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert orc.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert orc.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert orc.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


@pytest.mark.parametrize("L", [3, 5, 9])
@pytest.mark.parametrize("rep", [0, 1])
def test_f1_toric_stencils(L, rep):
    g = _load("f1_toric.npz")
    t = f"L{L}_{rep}"
    m = g[f"{t}_m"]
    assert orc.count_errors(m) == int(g[f"{t}_count"])
    assert orc.toric_eq_class(m) == int(g[f"{t}_class"])
    assert np.array_equal(orc.toric_syndrome(m), g[f"{t}_defects"])
    i = 0
    for op in (1, 3):
        for r in range(L):
            for c in range(L):
                new, dE = orc.toric_apply_stabilizer(m, r, c, op)
                assert dE == int(g[f"{t}_stab_dE"][i]) and np.array_equal(new, g[f"{t}_stab_new"][i])
                i += 1
    for j, (op, layer, xp, zp) in enumerate(g[f"{t}_log_arg"]):
        new, dE = orc.toric_apply_logical(m, int(op), int(layer), int(xp), int(zp))
        assert dE == int(g[f"{t}_log_dE"][j]) and np.array_equal(new, g[f"{t}_log_new"][j])
    for eq in range(16):
        out = orc.toric_to_class(m, eq)
        assert np.array_equal(out, g[f"{t}_to_class"][eq])
        assert orc.toric_eq_class(out) == eq


def test_f1_inline_kats_from_survey():
    # SURVEY.md Appendix C (captured from the reference)
    M = np.array([[[0, 2, 0], [0, 0, 1], [3, 0, 0]], [[1, 0, 0], [0, 3, 0], [0, 0, 2]]], dtype=np.uint8)
    assert orc.count_errors(M) == 6 and orc.toric_eq_class(M) == 0
    assert orc.toric_syndrome(M).tolist() == [[[1, 1, 0], [0, 0, 1], [0, 0, 1]], [[0, 1, 0], [0, 1, 0], [1, 0, 1]]]
    new, dE = orc.toric_apply_stabilizer(M, 2, 1, 1)
    assert dE == 4 and new.tolist() == [[[0, 2, 0], [0, 1, 1], [3, 1, 0]], [[1, 0, 0], [0, 3, 0], [1, 1, 2]]]
    new, dE = orc.toric_apply_logical(M, 2, 1, 1, 2)
    assert dE == 3 and orc.toric_eq_class(new) == 12


def _cases(prefix):
    g = _load("f2_toric.npz")
    return [c for c in g["cases"] if str(c).startswith(prefix)]


@pytest.mark.parametrize("case", _cases("chain"))
def test_f2_chain_trajectories(case):
    g = _load("f2_toric.npz")
    L, p, p_logical, iters, seed, ndraw = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    fin = orc.toric_chain_update(g[f"{case}_init"], float(p), float(p_logical), int(iters), rng)
    assert rng.consumed == int(ndraw)
    assert np.array_equal(fin, g[f"{case}_final"])


@pytest.mark.parametrize("case", _cases("ladder"))
def test_f2_ladder_trajectories(case):
    g = _load("f2_toric.npz")
    L, p, Nc, iters, nstep, seed, ndraw = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    ld = orc.ToricLadder(g[f"{case}_init"], float(p), int(Nc), 0.5)
    assert np.array_equal(ld.p_ladder, g[f"{case}_p_ladder"])
    assert np.array_equal(ld.p_diff, g[f"{case}_p_diff"])
    tops = []
    for _ in range(int(nstep)):
        ld.step(int(iters), rng)
        tops.append(ld.tops0)
    assert rng.consumed == int(ndraw)
    assert tops == g[f"{case}_tops_hist"].tolist()
    assert np.array_equal(ld.states, g[f"{case}_states"])
    assert np.array_equal(ld.flags, g[f"{case}_flags"])


@pytest.mark.parametrize("case", _cases("pteq"))
def test_f2_pteq_percent(case):
    g = _load("f2_toric.npz")
    L, p, Nc, iters, steps, tops_burn, conv, seed, ndraw, SEQ, TOPS, eps = g[f"{case}_par"]
    rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
    res = orc.toric_pteq(g[f"{case}_init"], float(p), Nc=int(Nc), SEQ=int(SEQ), TOPS=int(TOPS),
                         tops_burn=int(tops_burn), eps=float(eps), steps=int(steps), iters=int(iters),
                         conv_criteria="error_based" if conv else None, rng=rng)
    assert rng.consumed == int(ndraw)
    assert np.array_equal(res["percent"], g[f"{case}_percent"])


def test_f2_has_a_converged_case():
    g = _load("f2_toric.npz")
    n = 0
    for case in _cases("pteq"):
        L, p, Nc, iters, steps, tops_burn, conv, seed, ndraw, SEQ, TOPS, eps = g[f"{case}_par"]
        if not conv:
            continue
        rng = orc.Rng.stream(_stream(int(seed), int(ndraw)))
        res = orc.toric_pteq(g[f"{case}_init"], float(p), Nc=int(Nc), SEQ=int(SEQ), TOPS=int(TOPS),
                             tops_burn=int(tops_burn), eps=float(eps), steps=int(steps), iters=int(iters),
                             conv_criteria="error_based", rng=rng)
        n += res["converged"] and res["steps_done"] < int(steps)
    assert n >= 1


def test_f4_config1_plumbing():
    g = _load("f4_config1.npz")
    rng = orc.Rng.stream(_stream(1, int(g["draws"])))
    fin = orc.toric_chain_update(g["init"], 0.10, 0.0, 10000, rng)
    assert np.array_equal(fin, g["final"])
    assert orc.count_errors(fin) == int(g["count"]) == 7
    assert orc.toric_eq_class(fin) == int(g["cls"]) == 10
    assert np.array_equal(orc.toric_syndrome(g["init"]), g["defects"])
    assert np.array_equal(orc.toric_syndrome(fin), g["defects"])   # MCMC never changes the syndrome
