"""scan = "wave" (QECMC_SCAN_WAVE, csrc/ladder_wu.hpp; VERDICT r3 "Next round" item 2): the reference's random-scan chain with a
generator pick shared by the 64 ladders of a wavefront -- states in registers, addressed with the VGPR index mode.  Per syndrome
it IS the reference's Markov chain (toric_model.py:287-296 picks the generator independently of the state), so it is validated
bit for bit against the oracle's restatement of the rule (orc_model.scan = 3) -- final configuration of every rung, class
counts, samples, tops0, the criterion's stopping step -- and statistically against exact enumeration (tests/test_gpu_stats.py
runs its L = 3 cases with this scan too; the oracle side: tests/test_stats_cpu.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def q():
    import qecmc
    assert qecmc.device_count() >= 1
    return qecmc


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def _rand(rng, shape, p):
    return (rng.integers(1, 4, size=shape) * (rng.random(shape) < p)).astype(np.uint8)


def _codes(q, orc, name):
    return {"toric": (q.TORIC, orc.TORIC), "xzzx": (q.XZZX, orc.XZZX), "rotated": (q.ROTATED, orc.ROTATED), "planar": (q.PLANAR, orc.PLANAR)}[name]


def _init(rng, name, N, L, p):
    shape = (N, 2, L, L) if name in ("toric", "planar") else (N, L, L)
    init = _rand(rng, shape, p)
    if name == "planar":
        init[:, 1, -1, :] = 0; init[:, 1, :, -1] = 0
    return init


CASES = [  # name, L, Nc, N, steps, iters, p, p_logical, replicas, first_syndrome
    ("toric", 3, 3, 5, 400, 10, 0.10, 0.5, 1, 0), ("toric", 5, 5, 70, 200, 10, 0.10, 0.5, 1, 128),     # 70: two workgroups, the second ragged
    ("toric", 9, 8, 6, 200, 10, 0.15, 0.5, 1, 64), ("toric", 15, 8, 3, 60, 10, 0.18, 0.5, 1, 0),      # W = 11 -> 12 / 29 -> 32 state words
    ("toric", 4, 4, 4, 150, 7, 0.12, 0.5, 1, 0), ("toric", 9, 16, 2, 50, 3, 0.15, 0.25, 1, 0),         # odd iters: blocks of 8 cut anywhere; 16 rungs
    ("toric", 7, 2, 4, 200, 10, 0.10, 0.5, 1, 0), ("toric", 5, 5, 4, 120, 1, 0.02, 1.0, 3, 192),       # p = 0.02: thresholds below 2^-16
    ("toric", 5, 5, 3, 300, 10, 0.10, 0.0, 1, 0),                                                     # no logical moves (PTDC's ladders)
    ("toric", 16, 4, 2, 40, 25, 0.15, 0.5, 1, 0), ("toric", 9, 9, 3, 80, 10, 0.15, 0.5, 1, 0),        # 32 words exactly; 9 rungs: 1024-thread groups
    ("xzzx", 9, 8, 5, 150, 10, 0.15, 0.5, 1, 0), ("rotated", 7, 7, 4, 150, 10, 0.17, 0.5, 1, 0), ("rotated", 21, 8, 2, 30, 10, 0.17, 0.5, 1, 0),
    ("planar", 5, 5, 4, 150, 10, 0.12, 0.5, 1, 0), ("xzzx", 3, 2, 7, 300, 5, 0.2, 0.5, 2, 64), ("planar", 9, 8, 3, 100, 10, 0.12, 0.5, 1, 0)]


@pytest.mark.parametrize("name,L,Nc,N,steps,iters,p,p_logical,R,first", CASES)
def test_wave_scan_bit_exact(q, orc, name, L, Nc, N, steps, iters, p, p_logical, R, first):
    rng = np.random.default_rng(L * 7 + Nc + N)
    code, ocode = _codes(q, orc, name)
    init = _init(rng, name, N, L, p)
    kw = dict(steps=steps, iters=iters, tops_burn=1, seed=77, first_syndrome=first)
    got = q.pteq_batch(init, p, Nc=Nc, code=code, scan="wave", p_logical=p_logical, return_states=True, replicas=R, **kw)
    ncls = 16 if name == "toric" else 4
    counts = np.zeros((N * R, ncls), np.uint32); samples = np.zeros(N * R, np.uint64); tops0 = np.zeros(N * R, np.uint64)
    states = np.zeros((N * R, Nc) + init.shape[1:], np.uint8)
    for l in range(N * R):
        ld = orc.Ladder(ocode, init[l // R], p, Nc, p_logical, scan=3)
        r = orc.Rng.philox(77, first + l)
        for t in range(steps):
            ld.step(iters, r)
            if ld.tops0 >= 1:
                counts[l, orc.surf_eq_class(ocode, ld.states[0]) if name != "toric" else orc.toric_eq_class(ld.states[0])] += 1
                samples[l] += 1
        tops0[l] = ld.tops0
        states[l] = ld.states
    assert np.array_equal(got["states"], states)
    assert np.array_equal(got["counts"], counts.reshape(N, R, ncls).sum(axis=1))
    assert np.array_equal(got["samples"], samples.reshape(N, R).sum(axis=1).astype(np.uint32))
    assert np.array_equal(got["tops0"], tops0.reshape(N, R).sum(axis=1).astype(np.uint32))


@pytest.mark.parametrize("name,L,Nc,N,steps,iters,R", [("toric", 3, 3, 40, 3000, 10, 1), ("toric", 5, 5, 70, 4000, 10, 1), ("rotated", 5, 4, 25, 3000, 7, 1),
                                                    ("xzzx", 5, 5, 20, 3000, 10, 3), ("planar", 4, 4, 10, 2000, 5, 1)])
def test_wave_scan_with_the_convergence_criterion_bit_exact(q, orc, name, L, Nc, N, steps, iters, R):
    """conv_criteria = 'error_based' (decoders.py:74-82,93-105) in the scan = 3 kernel: stopping step, flag, class counts, samples and
    tops0 are the oracle's (its PTEQ loop around scan = 3); a workgroup leaves when its 64 ladders have all stopped."""
    rng = np.random.default_rng(L + 31 * Nc)
    code, ocode = _codes(q, orc, name)
    init = _init(rng, name, N, L, 0.1)
    kw = dict(steps=steps, iters=iters, tops_burn=2, seed=5, first_syndrome=64, conv_criteria="error_based", SEQ=2, TOPS=4, eps=0.3)
    got = q.pteq_batch(init, 0.1, Nc=Nc, code=code, scan="wave", replicas=R, **kw)
    ref = orc.pteq_batch(ocode, np.repeat(init, R, axis=0), 0.1, Nc, kw.pop("steps"), scan=3, **kw)
    ncls = ref["counts"].shape[1]
    assert ref["converged"].any()                                                     # (the criterion fires within the horizon)
    assert np.array_equal(got["counts"], ref["counts"].reshape(N, R, ncls).sum(axis=1))
    assert np.array_equal(got["samples"], ref["samples"].reshape(N, R).sum(axis=1).astype(np.uint32))
    assert np.array_equal(got["tops0"], ref["tops0"].reshape(N, R).sum(axis=1).astype(np.uint32))
    assert np.array_equal(got["steps_done"], ref["steps_done"].reshape(N, R).max(axis=1).astype(np.uint32))
    assert np.array_equal(got["converged"], ref["converged"].reshape(N, R).all(axis=1))


def test_wave_scan_sharded_equals_whole(q):
    """Results depend on the global ladder index only: a batch cut at a multiple of 64 gives the rows of the whole batch."""
    rng = np.random.default_rng(8)
    init = _rand(rng, (200, 2, 5, 5), 0.1)
    kw = dict(Nc=5, steps=300, iters=10, seed=3, scan="wave", return_states=True)
    whole = q.pteq_batch(init, 0.1, first_syndrome=256, **kw)
    a = q.pteq_batch(init[:128], 0.1, first_syndrome=256, **kw)
    b = q.pteq_batch(init[128:], 0.1, first_syndrome=384, **kw)
    for key in ("counts", "samples", "tops0", "states"):
        assert np.array_equal(whole[key], np.concatenate([a[key], b[key]]))


def test_wave_scan_conserves_the_syndrome(q):
    from qecmc import toric_model as tm
    rng = np.random.default_rng(2)
    init = _rand(rng, (70, 2, 9, 9), 0.15)
    got = q.pteq_batch(init, 0.15, Nc=8, steps=500, scan="wave", return_states=True, seed=5)
    for s in range(0, 70, 7):
        ref = tm.syndrome(init[s])
        for c in range(8):
            assert np.array_equal(tm.syndrome(got["states"][s, c]), ref)
    assert (got["samples"] == got["counts"].sum(axis=1)).all()


def test_wave_scan_rejects_what_it_does_not_do(q):
    init = np.zeros((2, 2, 5, 5), np.uint8)
    with pytest.raises(q.QecmcError):
        q.pteq_batch(np.zeros((1, 5, 5), np.uint8), 0.1, Nc=5, steps=100, scan="wave", code=q.XZZX, eta=10.0)     # biased rule
    with pytest.raises(q.QecmcError):
        q.pteq_batch(init, 0.1, Nc=1, steps=100, scan="wave", p_logical=0.5)          # a 1-rung ladder's top sits below p = 0.75
    with pytest.raises(q.QecmcError):
        q.pteq_batch(init, 0.1, Nc=5, steps=100, scan="wave", first_syndrome=7)       # a wavefront is one pick group
