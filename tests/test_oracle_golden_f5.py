"""Pins the oracle's equilibrium observables (per-rung error-count sums, per-pair swap acceptances, class counts, tops0)
against the reference's own long ladder runs of fixture F5 (tests/golden/gen_golden.py:gen_f5): the reference ran under
`random.seed(seed)`, so feeding the oracle the same MT19937 stream replays a whole 20 000-step run exactly -- toric L=9
p=0.15 Nc=8 (BASELINE config 2's shape), rotated L=5/7, Ladder_biased on xzzx L=5/7."""
import os
import random

import numpy as np
import pytest

from oracle import oracle as orc

from conftest import GOLDEN

CASES = [("toric_L9", 0, 0), ("toric_L9", 2, 5), ("rot_L5", 1, 3), ("rot_L7", 0, 1), ("xzzxb_L5", 2, 0), ("xzzxb_L7", 1, 7)]


@pytest.mark.parametrize("name,s,r", CASES)
def test_f5_replica_replayed_exactly(name, s, r):
    g = np.load(os.path.join(GOLDEN, "f5_stats.npz"))
    L, p, eta, Nc, iters, steps, burn = g[f"{name}_par"]
    L, Nc, iters, steps, burn = int(L), int(Nc), int(iters), int(steps), int(burn)
    init = g[f"{name}_init"][s]
    seed = 7000 + 100 * s + r                                     # gen_f5's job seed
    per_prop = 5 if name.startswith("toric") else 7               # draws per proposal, with margin (toric 4-4.5, plaquette codes 5-6)
    mt = random.Random(seed)
    stream = np.array([mt.random() for _ in range(steps * (Nc * iters * per_prop + Nc))], dtype=np.float64)
    rng = orc.Rng.stream(stream)
    if name.startswith("toric"):
        ld, ncls = orc.ToricLadder(init, float(p), Nc, 0.5), 16
        cls = orc.toric_eq_class
    else:
        code = orc.XZZX if name.startswith("xzzx") else orc.ROTATED
        biased = name.startswith("xzzxb")
        ld, ncls = orc.Ladder(code, init, float(p), Nc, 0.5, noise=orc.BIASED if biased else orc.DEPOLARIZING, eta=float(eta)), 4
        cls = lambda m: orc.surf_eq_class(code, m)
    hist = np.zeros(ncls, dtype=np.int64)
    for t in range(steps):
        if t == burn:
            acc0, n0 = ld.swap_accepts.astype(np.int64), ld.nerr_sums.astype(np.int64)
        ld.step(iters, rng)
        if t >= burn:
            hist[cls(ld.states[0])] += 1
    assert np.array_equal(hist, g[f"{name}_hist"][s, r])
    assert ld.tops0 == int(g[f"{name}_tops0"][s, r])
    assert np.array_equal(ld.swap_accepts.astype(np.int64) - acc0, g[f"{name}_swap_acc"][s, r])
    assert np.all(g[f"{name}_swap_att"][s, r] == steps - burn)
    # the fixture stores the time average; the sums are integers, so the average is exact in float64
    assert np.array_equal((ld.nerr_sums.astype(np.int64) - n0) / (steps - burn), g[f"{name}_nerr"][s, r])
