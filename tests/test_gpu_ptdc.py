"""GPU parity for the direct-counting estimator PTDC (decoders.py:138-233): the unique-chain length histograms N(n) of the
GPU path (ladder kernel + hash-set insertion kernel) against the oracle, bit for bit, and the estimator against exact
enumeration of the stabilizer group."""
import numpy as np
import pytest

from util_exact import toric_class_probabilities

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


def _toric_reps(q, m):
    from qecmc import toric_model as tm
    return np.stack([tm.to_class(m, eq) for eq in range(16)])


def _rand_toric(rng, L, p):
    m = np.zeros((2, L, L), dtype=np.uint8)
    err = rng.random(m.shape) < p
    m[err] = rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)
    return m


@pytest.mark.parametrize("L,p,Nc,steps,droplets,N", [(3, 0.1, 3, 120, 1, 3), (5, 0.1, 5, 60, 3, 2), (5, 0.3, 2, 100, 2, 2), (7, 0.15, 8, 25, 2, 1)])
def test_ptdc_batch_toric_bit_exact(q, orc, L, p, Nc, steps, droplets, N):
    rng = np.random.default_rng(L * 7 + steps)
    init = np.stack([_toric_reps(q, _rand_toric(rng, L, 0.12)) for _ in range(N)])
    assert [orc.toric_eq_class(r) for r in init[0]] == list(range(16))
    got = q.ptdc_batch(init, p, Nc=Nc, steps=steps, droplets=droplets, seed=99, first_syndrome=4)
    ref = orc.ptdc_batch(orc.TORIC, init, p, Nc, steps, droplets=droplets, seed=99, first_syndrome=4)
    assert got.shape == (N, 16, 2 * L * L + 1) and got.sum() > 0
    assert np.array_equal(got, ref)


def test_ptdc_batch_planar_bit_exact(q, orc):
    rng = np.random.default_rng(12)
    L = 5
    from qecmc import planar_model as pm
    m = np.zeros((2, L, L), dtype=np.uint8)
    err = rng.random(m.shape) < 0.1
    m[err] = rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)
    m[1, -1, :] = 0; m[1, :, -1] = 0
    reps = [pm.apply_logical(m, op, 0, 0)[0] for op in range(4)]          # one representative per class
    assert sorted(int(pm.eq_class(r)) for r in reps) == [0, 1, 2, 3]
    init = np.stack(reps)[None]
    got = q.ptdc_batch(init, 0.12, Nc=4, steps=80, droplets=2, seed=5, code=q.PLANAR)
    ref = orc.ptdc_batch(orc.PLANAR, init, 0.12, 4, 80, droplets=2, seed=5)
    assert np.array_equal(got, ref) and got.sum() > 0


def test_ptdc_dropin_matches_exact_classes(q, orc):
    """PTDC with the reference's signature; at L = 3 the sets it collects hold the chains that dominate Z_E, so the
    estimate sits close to the exact class probabilities (and its argmax on the true class)."""
    rng = np.random.default_rng(3)
    m = _rand_toric(rng, 3, 0.12)
    code = q.Toric_code(3)
    code.qubit_matrix = m.copy()
    p = 0.1
    pct = q.PTDC(code, p, droplets=4, steps=6000, seed=17)
    assert pct.dtype == np.uint8 and pct.shape == (16,)
    P = toric_class_probabilities(m, p, orc.toric_apply_stabilizer, orc.toric_to_class) * 100
    assert pct.argmax() == P.argmax() and 0.5 * np.abs(pct.astype(np.float64) - P).sum() < 6.0
    # the same ladders through the oracle
    hist = orc.ptdc_batch(orc.TORIC, _toric_reps(q, m)[None], p, 3, 6000 // 3, droplets=4, seed=17)
    assert np.array_equal(pct, orc.ptdc_distribution(hist[0], p).astype(np.uint8))


def _planar_reps(rng, L, p):
    from qecmc import planar_model as pm
    m = np.zeros((2, L, L), dtype=np.uint8)
    err = rng.random(m.shape) < p
    m[err] = rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)
    m[1, -1, :] = 0; m[1, :, -1] = 0
    reps = sorted((pm.apply_logical(m, op, 0, 0)[0] for op in range(4)), key=lambda r: int(pm.eq_class(r)))
    return m, np.stack(reps)


def test_stdc_is_the_one_chain_case(q, orc):
    """STDC_droplet (decoders.py:236-265): single chains, `update_chain_fast(5)` per step = Nc = 1, iters = 5."""
    rng = np.random.default_rng(21)
    init = np.stack([_planar_reps(rng, 5, 0.1)[1] for _ in range(3)])
    got = q.ptdc_batch(init, 0.15, Nc=1, steps=400, droplets=3, iters=5, seed=8, first_syndrome=2, code=q.PLANAR)
    ref = orc.ptdc_batch(orc.PLANAR, init, 0.15, 1, 400, droplets=3, iters=5, seed=8, first_syndrome=2)
    assert np.array_equal(got, ref) and got.sum() > 1000
    # the drop-in with the list form of init_code (no rain)
    codes = []
    for r in init[0]:
        c = q.Planar_code(5); c.qubit_matrix = r.copy(); codes.append(c)
    dist = q.STDC(codes, 0.1, p_sampling=0.15, droplets=3, steps=400, seed=8)
    ref1 = orc.ptdc_batch(orc.PLANAR, init[:1], 0.15, 1, 400, droplets=3, iters=5, seed=8)
    assert dist.dtype == np.float64 and np.allclose(dist, orc.ptdc_distribution(ref1[0], 0.1), rtol=1e-12)
    assert abs(dist.sum() - 100) < 1e-9


def test_stdc_rain_start(q, orc):
    """A single toric init_code: every droplet starts from apply_stabilizers_uniform() of the class representative
    (decoders.py:246-247,289-294); the per-droplet starts travel through init_per_droplet."""
    rng = np.random.default_rng(4)
    code = q.Toric_code(3)
    code.qubit_matrix = _rand_toric(rng, 3, 0.12)
    np.random.seed(11)
    dist = q.STDC(code, 0.1, droplets=2, steps=500, seed=3)
    np.random.seed(11)                                   # replay the rain on the host to rebuild the droplets' starts
    import copy
    starts = []
    for eq in range(16):
        c = copy.deepcopy(code); c.qubit_matrix = c.to_class(eq)
        starts.append(np.stack([c.apply_stabilizers_uniform() for _ in range(2)]))
    init = np.stack(starts)[None]
    assert all(orc.toric_eq_class(init[0, eq, d]) == eq for eq in range(16) for d in range(2))
    ref = orc.ptdc_batch(orc.TORIC, init, 0.1, 1, 500, droplets=2, iters=5, seed=3)
    assert np.allclose(dist, orc.ptdc_distribution(ref[0], 0.1), rtol=1e-12)
    P = toric_class_probabilities(code.qubit_matrix, 0.1, orc.toric_apply_stabilizer, orc.toric_to_class) * 100
    assert dist.argmax() == P.argmax()


def test_per_rung_sets_and_observation_counts(q, orc):
    """PTRC_droplet's bookkeeping (decoders.py:584-631): one set per (ladder, rung), N(n) and m(n)."""
    rng = np.random.default_rng(31)
    init = np.stack([_toric_reps(q, _rand_toric(rng, 3, 0.12)) for _ in range(2)])
    got_n, got_m = q.ptdc_batch(init, 0.1, Nc=4, steps=150, droplets=2, seed=6, first_syndrome=1, per_rung=True, with_m=True)
    ref_n, ref_m = orc.ptdc_batch(orc.TORIC, init, 0.1, 4, 150, droplets=2, seed=6, first_syndrome=1, per_rung=True, with_m=True)
    assert got_n.shape == (2, 16, 2, 4, 19)
    assert np.array_equal(got_n, ref_n) and np.array_equal(got_m, ref_m)
    assert np.all(got_m.sum(axis=-1) == 150) and np.all(got_n <= got_m)
    # shared sets with m(n): the single-chain form STRC uses
    got_n, got_m = q.ptdc_batch(init, 0.2, Nc=1, steps=300, droplets=3, iters=5, seed=6, with_m=True)
    ref_n, ref_m = orc.ptdc_batch(orc.TORIC, init, 0.2, 1, 300, droplets=3, iters=5, seed=6, with_m=True)
    assert np.array_equal(got_n, ref_n) and np.array_equal(got_m, ref_m) and np.all(got_m.sum(axis=-1) == 900)


def test_strc_ptrc_dropins(q, orc):
    from qecmc.decoders import strc_distribution, ptrc_distribution
    rng = np.random.default_rng(5)
    m = _rand_toric(rng, 3, 0.12)
    p = 0.1
    P = toric_class_probabilities(m, p, orc.toric_apply_stabilizer, orc.toric_to_class) * 100
    codes = []
    for r in _toric_reps(q, m):
        c = q.Toric_code(3); c.qubit_matrix = r.copy(); codes.append(c)
    pct = q.PTRC(codes, p, droplets=3, Nc=3, steps=6000, seed=9)
    n_u, m_o = orc.ptdc_batch(orc.TORIC, _toric_reps(q, m)[None], p, 3, 2000, droplets=3, seed=9, per_rung=True, with_m=True)
    assert pct.dtype == np.uint8 and np.array_equal(pct, ptrc_distribution(n_u[0], m_o[0], p, p))
    assert 0.5 * np.abs(pct.astype(np.float64) - P).sum() < 10.0     # (the two leading classes of this syndrome are degenerate: no argmax test)
    dist = q.STRC(codes, p, p_sampling=0.2, droplets=3, steps=3000, seed=10)
    n_u, m_o = orc.ptdc_batch(orc.TORIC, _toric_reps(q, m)[None], 0.2, 1, 3000, droplets=3, iters=5, seed=10, with_m=True)
    assert np.allclose(dist, strc_distribution(n_u[0], m_o[0], p, 0.2), rtol=1e-12)
    assert 0.5 * np.abs(dist - P).sum() < 10.0


def test_single_temp_dropin(q, orc):
    rng = np.random.default_rng(41)
    _, reps = _planar_reps(rng, 5, 0.1)
    codes = []
    for r in reps:
        c = q.Planar_code(5); c.qubit_matrix = r.copy(); codes.append(c)
    means = q.single_temp(codes, 0.2, 400, seed=2)
    _, m_o = orc.ptdc_batch(orc.PLANAR, reps[None], 0.2, 1, 399, droplets=1, iters=5, seed=2, with_m=True)
    assert np.allclose(means, (m_o[0] * np.arange(m_o.shape[-1])).sum(axis=-1) / 399, rtol=1e-12) and means.shape == (4,)


# ---- the conv_mult early stop (decoders.py:153-162, :256-262, :783-826) -------------------------------------------------------------

@pytest.mark.parametrize("L,p,Nc,steps,droplets,cm,N", [(3, 0.1, 3, 400, 1, 2.0, 2), (5, 0.1, 5, 300, 3, 2.0, 1), (4, 0.3, 2, 1000, 2, 1.5, 2),
                                                        (3, 0.2, 8, 250, 2, 3.0, 1)])
def test_conv_mult_toric_bit_exact(q, orc, L, p, Nc, steps, droplets, cm, N):
    """Every droplet stops on its own (its private dictionary decides), the class sets are the union over the droplets."""
    rng = np.random.default_rng(L * 11 + steps)
    init = np.stack([_toric_reps(q, _rand_toric(rng, L, 0.12)) for _ in range(N)])
    got_n, got_m, sd = q.ptdc_batch(init, p, Nc=Nc, steps=steps, droplets=droplets, seed=41, first_syndrome=3, with_m=True, conv_mult=cm,
                                    return_steps=True)
    ref_n, ref_m = orc.ptdc_batch(orc.TORIC, init, p, Nc, steps, droplets=droplets, seed=41, first_syndrome=3, with_m=True, conv_mult=cm)
    assert np.array_equal(got_n, ref_n) and np.array_equal(got_m, ref_m)
    assert sd.shape == (N, 16, droplets) and sd.min() >= 1 and sd.max() <= steps and sd.min() < steps      # some droplet stopped early
    assert np.array_equal(sd.sum(axis=-1) * Nc, got_m.sum(axis=-1))                  # every recorded step observes every rung once
    assert np.all(sd * 100 >= steps) or np.all(sd[sd * 100 < steps] == steps)        # never before steps / 100 (:160)
    full = q.ptdc_batch(init, p, Nc=Nc, steps=steps, droplets=droplets, seed=41, first_syndrome=3)
    assert np.all(got_n <= full) and got_n.sum() < full.sum()                        # a prefix of the full run


def test_conv_mult_single_chains_and_dropins(q, orc):
    """STDC / STRC form (Nc = 1, iters = 5) with per-droplet starts; the drop-ins pass conv_mult through."""
    rng = np.random.default_rng(77)
    from qecmc.decoders import strc_distribution
    m, reps = _planar_reps(rng, 5, 0.1)
    init = np.stack([np.stack([r, r, r]) for r in reps])[None]                      # [1, 4, droplets = 3, ...]
    got_n, got_m, sd = q.ptdc_batch(init, 0.15, Nc=1, steps=900, droplets=3, iters=5, seed=9, code=q.PLANAR, with_m=True, conv_mult=2.5,
                                    return_steps=True)
    ref_n, ref_m = orc.ptdc_batch(orc.PLANAR, init, 0.15, 1, 900, droplets=3, iters=5, seed=9, with_m=True, conv_mult=2.5)
    assert np.array_equal(got_n, ref_n) and np.array_equal(got_m, ref_m) and sd.min() < 900
    codes = []
    for r in reps:
        c = q.Planar_code(5); c.qubit_matrix = r.copy(); codes.append(c)
    ref_n1, ref_m1 = orc.ptdc_batch(orc.PLANAR, reps[None], 0.15, 1, 900, droplets=3, iters=5, seed=9, with_m=True, conv_mult=2.5)
    assert np.allclose(q.STDC(codes, 0.1, p_sampling=0.15, droplets=3, steps=900, conv_mult=2.5, seed=9),
                       orc.ptdc_distribution(ref_n1[0], 0.1), rtol=1e-12)
    a = q.STRC(codes, 0.1, p_sampling=0.15, droplets=3, steps=900, conv_mult=2.5, seed=9)     # list form: no rain (:846-847)
    assert np.allclose(a, strc_distribution(ref_n1[0], ref_m1[0], 0.1, 0.15), rtol=1e-12) and abs(a.sum() - 100) < 1e-9
    code = q.Toric_code(3); code.qubit_matrix = _rand_toric(rng, 3, 0.12)
    pct = q.PTDC(code, 0.1, droplets=2, steps=3000, conv_mult=2.0, seed=5)
    ref = orc.ptdc_batch(orc.TORIC, _toric_reps(q, code.qubit_matrix)[None], 0.1, 3, 1000, droplets=2, seed=5, conv_mult=2.0)
    assert np.array_equal(pct, orc.ptdc_distribution(ref[0], 0.1).astype(np.uint8))


def test_conv_mult_is_ignored_per_rung(q):
    """PTRC_droplet's stop is commented out in the reference (decoders.py:627-630)."""
    rng = np.random.default_rng(5)
    init = _toric_reps(q, _rand_toric(rng, 3, 0.12))[None]
    a = q.ptdc_batch(init, 0.1, Nc=3, steps=200, droplets=2, seed=1, per_rung=True, conv_mult=2.0)
    b = q.ptdc_batch(init, 0.1, Nc=3, steps=200, droplets=2, seed=1, per_rung=True)
    assert np.array_equal(a, b)


# ---- (n_x, n_y, n_z) of the distinct chains, Chain_xyz sampling: STDC_general_noise family (decoders.py:325-507) ----------------------

def _sorted_sets(orc, xv):
    """oracle output uint32[N, ncls, maxu] -> list[N][ncls] of sorted int64[k, 3]"""
    from qecmc.decoders import unpack_xyz
    return [[unpack_xyz(xv[s, c]) for c in range(xv.shape[1])] for s in range(xv.shape[0])]


def _same_sets(a, b):
    return len(a) == len(b) and all(len(x) == len(y) and all(np.array_equal(u, v) for u, v in zip(x, y)) for x, y in zip(a, b))


def test_xyz_of_distinct_chains_bit_exact(q, orc):
    rng = np.random.default_rng(88)
    init = np.stack([_planar_reps(rng, 5, 0.1)[1] for _ in range(2)])
    hist, xyz = q.ptdc_batch(init, 0.15, Nc=1, steps=300, droplets=3, iters=5, seed=12, first_syndrome=1, code=q.PLANAR, with_xyz=True)
    rh, rx = orc.ptdc_batch(orc.PLANAR, init, 0.15, 1, 300, droplets=3, iters=5, seed=12, first_syndrome=1, with_xyz=True)
    assert np.array_equal(hist, rh) and _same_sets(xyz, _sorted_sets(orc, rx))
    for s in range(2):
        for c in range(4):
            assert np.array_equal(np.bincount(xyz[s][c].sum(axis=1), minlength=hist.shape[-1]), hist[s, c])
    # any ladder: toric, Nc = 3, with the early stop
    ti = _toric_reps(q, _rand_toric(rng, 3, 0.12))[None]
    hist, xyz = q.ptdc_batch(ti, 0.1, Nc=3, steps=200, droplets=2, seed=3, with_xyz=True, conv_mult=2.0)
    rh, rx = orc.ptdc_batch(orc.TORIC, ti, 0.1, 3, 200, droplets=2, seed=3, with_xyz=True, conv_mult=2.0)
    assert np.array_equal(hist, rh) and _same_sets(xyz, _sorted_sets(orc, rx))


@pytest.mark.parametrize("code_name,L,pxyz", [("planar", 5, (0.08, 0.01, 0.03)), ("planar", 3, (0.02, 0.05, 0.11)), ("xzzx", 5, (0.3, 0.2, 0.1)),
                                              ("rotated", 5, (0.01, 0.02, 0.2))])
def test_chain_xyz_sampling_bit_exact(q, orc, code_name, L, pxyz):
    """Chain_xyz (mcmc.py:106-114,162-173): acceptance prod_i factors_i^(change of n_i) from a table of integer thresholds."""
    rng = np.random.default_rng(L + len(code_name))
    if code_name == "planar":
        init = np.stack([_planar_reps(rng, L, 0.1)[1] for _ in range(2)])
        cg, co = q.PLANAR, orc.PLANAR
    else:
        cg, co = (q.XZZX, orc.XZZX) if code_name == "xzzx" else (q.ROTATED, orc.ROTATED)
        m = rng.integers(0, 4, size=(2, 4, L, L), dtype=np.uint8) * (rng.random((2, 4, L, L)) < 0.15)
        init = m.astype(np.uint8)                         # the class labels do not matter for the sampling parity
    pa = np.array(pxyz)
    hist, xyz = q.ptdc_batch(init, pa, Nc=1, steps=250, droplets=2, iters=5, seed=21, code=cg, with_xyz=True)
    rh, rx = orc.ptdc_batch(co, init, pa, 1, 250, droplets=2, iters=5, seed=21, with_xyz=True)
    assert hist.sum() > 50 and np.array_equal(hist, rh) and _same_sets(xyz, _sorted_sets(orc, rx))


def test_general_noise_dropins(q, orc):
    from qecmc.decoders import general_noise_distribution
    rng = np.random.default_rng(61)
    _, reps = _planar_reps(rng, 3, 0.12)
    codes = []
    for r in reps:
        c = q.Planar_code(3); c.qubit_matrix = r.copy(); codes.append(c)
    p_xyz = np.array([0.03, 0.02, 0.08])
    for ps in (None, 0.2, np.array([0.1, 0.06, 0.08])):
        dist = q.STDC_general_noise(codes, p_xyz, p_sampling=ps, droplets=3, steps=500, seed=4)
        _, rx = orc.ptdc_batch(orc.PLANAR, reps[None], p_xyz.sum() if ps is None else ps, 1, 500, droplets=3, iters=5, seed=4, with_xyz=True)
        sets = _sorted_sets(orc, rx)[0]
        assert np.allclose(dist, general_noise_distribution(sets, p_xyz), rtol=1e-12) and abs(dist.sum() - 100) < 1e-9
        both = q.STDC_general_noise_shortest(codes, p_xyz, p_sampling=ps, droplets=3, steps=500, seed=4)
        assert np.allclose(both[0], dist, rtol=1e-12)
        assert np.allclose(both[1], general_noise_distribution(sets, p_xyz, shortest_only=True), rtol=1e-12)
        assert np.allclose(q.STDC_general_noise(codes, p_xyz, p_sampling=ps, droplets=3, steps=500, shortest_only=True, seed=4), both[1], rtol=1e-12)
    with pytest.raises(q.QecmcError, match="Nc=3 must be 1"):
        q.ptdc_batch(reps[None], p_xyz, Nc=3, steps=10, code=q.PLANAR)
    with pytest.raises(q.QecmcError, match="positive"):
        q.ptdc_batch(reps[None], np.array([0.1, 0.0, 0.1]), Nc=1, steps=10, code=q.PLANAR)


@pytest.mark.parametrize("code_name,L,pzs,alpha", [("xzzx", 5, 0.15, 1.7), ("rotated", 5, 0.2, 1.0), ("xzzx", 7, 0.1, 3.0)])
def test_alpha_droplets_bit_exact(q, orc, code_name, L, pzs, alpha):
    """STDC_droplet_alpha (decoders.py:510-534): Chain_alpha single chains, 5 proposals per step."""
    rng = np.random.default_rng(L * 3 + len(code_name))
    cg, co = (q.XZZX, orc.XZZX) if code_name == "xzzx" else (q.ROTATED, orc.ROTATED)
    init = (rng.integers(1, 4, size=(2, 4, L, L)) * (rng.random((2, 4, L, L)) < 0.15)).astype(np.uint8)
    hist, xyz = q.ptdc_batch(init, pzs, Nc=1, steps=300, droplets=2, iters=5, seed=33, first_syndrome=2, code=cg, with_xyz=True, alpha=alpha)
    rh, rx = orc.ptdc_batch(co, init, pzs, 1, 300, droplets=2, iters=5, seed=33, first_syndrome=2, with_xyz=True, alpha=alpha)
    assert hist.sum() > 20 and np.array_equal(hist, rh) and _same_sets(xyz, _sorted_sets(orc, rx))
    with pytest.raises(q.QecmcError, match="must be 1"):
        q.ptdc_batch(init, pzs, Nc=3, steps=10, code=cg, alpha=alpha)


def test_nall_n_alpha_dropin(q, orc):
    from qecmc.decoders import nall_n_alpha_distribution
    rng = np.random.default_rng(19)
    L = 5
    code = q.xzzx_code(L)
    code.qubit_matrix = (rng.integers(1, 4, size=(L, L)) * (rng.random((L, L)) < 0.12)).astype(np.uint8)
    dist = q.STDC_Nall_n_alpha(code, pz_tilde_sampling=0.2, alpha=2.0, pz_tilde=0.1, steps=600, seed=7)
    import copy
    reps = np.stack([copy.deepcopy(code).apply_logical(code.define_equivalence_class() ^ eq)[0] for eq in range(4)])
    assert [orc.surf_eq_class(orc.XZZX, r) for r in reps] == [0, 1, 2, 3]
    _, rx = orc.ptdc_batch(orc.XZZX, reps[None], 0.2, 1, 600, droplets=1, iters=5, seed=7, with_xyz=True, alpha=2.0)
    assert np.allclose(dist, nall_n_alpha_distribution(_sorted_sets(orc, rx)[0], 2.0, 0.1), rtol=1e-12) and abs(dist.sum() - 100) < 1e-9
    codes = []
    for r in reps:
        c = q.xzzx_code(L); c.qubit_matrix = r.copy(); codes.append(c)
    assert np.allclose(q.STDC_Nall_n_alpha(codes, pz_tilde_sampling=0.2, alpha=2.0, pz_tilde=0.1, steps=600, seed=7), dist, rtol=1e-12)


def test_unique_chain_edge_cases(q, orc):
    """empty batch, a single step, a ragged last workgroup (ladders not a multiple of 64), refused shapes"""
    rng = np.random.default_rng(8)
    init = np.stack([_toric_reps(q, _rand_toric(rng, 3, 0.12)) for _ in range(5)])     # 5 x 16 x 3 droplets = 240 ladders
    assert q.ptdc_batch(init[:0], 0.1, Nc=3, steps=10).shape == (0, 16, 19)
    got, sd = q.ptdc_batch(init, 0.1, Nc=3, steps=1, droplets=3, seed=2, conv_mult=2.0, return_steps=True)
    ref = orc.ptdc_batch(orc.TORIC, init, 0.1, 3, 1, droplets=3, seed=2, conv_mult=2.0)
    assert np.array_equal(got, ref) and np.all(sd == 1) and np.all(got.sum(axis=-1) <= 3 * 3)
    got, xyz = q.ptdc_batch(init, 0.2, Nc=2, steps=37, droplets=3, seed=4, first_syndrome=1000, with_xyz=True)
    rh, rx = orc.ptdc_batch(orc.TORIC, init, 0.2, 2, 37, droplets=3, seed=4, first_syndrome=1000, with_xyz=True)
    assert np.array_equal(got, rh) and _same_sets(xyz, _sorted_sets(orc, rx))
    with pytest.raises(q.QecmcError, match="per-class sets"):
        q.ptdc_batch(init, 0.1, Nc=3, steps=5, droplets=3, per_rung=True, with_xyz=True)
    with pytest.raises(ValueError):
        q.ptdc_batch(init[:, :4], 0.1, Nc=3, steps=5)
    big = np.zeros((1, 16, 2, 23, 23), dtype=np.uint8)                                # nq = 1058: the packed counts have 10 bits each
    with pytest.raises(q.QecmcError, match="10 bits"):
        q.ptdc_batch(big, 0.1, Nc=2, steps=2, with_xyz=True)
