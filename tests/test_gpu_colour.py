"""scan = "colour" (QECMC_SCAN_COLOUR, csrc/ladder_colour.hip): the latency layout -- one workgroup per
ladder, one wavefront per rung, a whole colour phase of mutually disjoint generators per wavefront pass.  Not the reference's
Markov chain (a systematic scan), so it is validated like scan = "sweep": bit for bit against the oracle's own restatement of the
rule (orc_model.scan = 2: the same phases, built independently, applied one generator after the other), and statistically
against exact enumeration (tests/test_gpu_stats.py runs its L = 3 cases with this scan too)."""
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


CASES = [  # name, L, Nc, N, steps, iters, p, p_logical, replicas
    ("toric", 3, 3, 5, 400, 10, 0.10, 0.5, 1), ("toric", 5, 5, 9, 300, 10, 0.10, 0.5, 1), ("toric", 9, 8, 6, 200, 10, 0.15, 0.5, 1),
    ("toric", 15, 8, 3, 60, 10, 0.18, 0.5, 1),          # 450 generators: colour classes of more than 64 are cut into two passes
    ("toric", 4, 4, 4, 150, 7, 0.12, 0.5, 1), ("toric", 9, 16, 2, 50, 3, 0.15, 0.25, 1), ("toric", 5, 1, 4, 200, 10, 0.10, 0.0, 1),
    ("toric", 5, 5, 3, 120, 1, 0.10, 1.0, 3),            # replicas: ladder l = s R + r, summed per syndrome
    ("xzzx", 9, 8, 5, 150, 10, 0.15, 0.5, 1), ("rotated", 7, 7, 4, 150, 10, 0.17, 0.5, 1), ("rotated", 21, 8, 2, 30, 10, 0.17, 0.5, 1),
    ("planar", 5, 5, 4, 150, 10, 0.12, 0.5, 1), ("xzzx", 3, 2, 7, 300, 5, 0.2, 0.5, 2)]


@pytest.mark.parametrize("name,L,Nc,N,steps,iters,p,p_logical,R", CASES)
def test_colour_scan_bit_exact(q, orc, name, L, Nc, N, steps, iters, p, p_logical, R):
    rng = np.random.default_rng(L * 7 + Nc + N)
    code, ocode = {"toric": (q.TORIC, orc.TORIC), "xzzx": (q.XZZX, orc.XZZX), "rotated": (q.ROTATED, orc.ROTATED), "planar": (q.PLANAR, orc.PLANAR)}[name]
    shape = (N, 2, L, L) if name in ("toric", "planar") else (N, L, L)
    init = _rand(rng, shape, p)
    if name == "planar":
        init[:, 1, -1, :] = 0; init[:, 1, :, -1] = 0
    kw = dict(steps=steps, iters=iters, tops_burn=1, seed=77, first_syndrome=11)
    got = q.pteq_batch(init, p, Nc=Nc, code=code, scan="colour", p_logical=p_logical, return_states=True, TOPS=2, replicas=R, **kw)
    # the oracle's ladder takes p_logical = 0.5 in its batch call (decoders.py:52): drive its Ladder directly for the other values
    ncls = 16 if name == "toric" else 4
    counts = np.zeros((N * R, ncls), np.uint32); samples = np.zeros(N * R, np.uint64); tops0 = np.zeros(N * R, np.uint64)
    reached = np.zeros(N * R, np.uint64)
    states = np.zeros((N * R, Nc) + shape[1:], np.uint8)
    for l in range(N * R):
        ld = orc.Ladder(ocode, init[l // R], p, Nc, p_logical, scan=2)
        r = orc.Rng.philox(77, 11 + l)
        for t in range(steps):
            ld.step(iters, r)
            if ld.tops0 >= 1:
                counts[l, orc.surf_eq_class(ocode, ld.states[0]) if name != "toric" else orc.toric_eq_class(ld.states[0])] += 1
                samples[l] += 1
            if not reached[l] and ld.tops0 >= 2:
                reached[l] = t + 1
        tops0[l] = ld.tops0
        states[l] = ld.states
    assert np.array_equal(got["states"], states)
    assert np.array_equal(got["counts"], counts.reshape(N, R, ncls).sum(axis=1))
    assert np.array_equal(got["samples"], samples.reshape(N, R).sum(axis=1).astype(np.uint32))
    assert np.array_equal(got["tops0"], tops0.reshape(N, R).sum(axis=1).astype(np.uint32))
    want_done = np.where(reached > 0, reached, steps).reshape(N, R).max(axis=1)
    assert np.array_equal(got["steps_done"], want_done.astype(np.uint32))                 # the first step with tops0 >= TOPS
    assert np.array_equal(got["converged"], (reached > 0).reshape(N, R).all(axis=1))


@pytest.mark.parametrize("name,L,Nc,N,steps,iters,R", [("toric", 3, 3, 40, 3000, 10, 1), ("toric", 5, 5, 30, 4000, 10, 1), ("rotated", 5, 4, 25, 3000, 7, 1),
                                                    ("xzzx", 5, 5, 20, 3000, 10, 3), ("planar", 4, 4, 10, 2000, 5, 1)])
def test_colour_scan_with_the_convergence_criterion_bit_exact(q, orc, name, L, Nc, N, steps, iters, R):
    """conv_criteria = 'error_based' (decoders.py:74-105) in the latency layout: the criterion runs on wave 0 of the ladder's workgroup,
    which leaves when it fires.  Stopping step, flag, class counts, samples and tops0 are the oracle's (its PTEQ loop around scan = 2)."""
    rng = np.random.default_rng(L * 5 + N)
    code, ocode = {"toric": (q.TORIC, orc.TORIC), "xzzx": (q.XZZX, orc.XZZX), "rotated": (q.ROTATED, orc.ROTATED), "planar": (q.PLANAR, orc.PLANAR)}[name]
    shape = (N, 2, L, L) if name in ("toric", "planar") else (N, L, L)
    init = _rand(rng, shape, 0.1)
    if name == "planar":
        init[:, 1, -1, :] = 0; init[:, 1, :, -1] = 0
    kw = dict(steps=steps, iters=iters, tops_burn=1, seed=41, first_syndrome=3, conv_criteria="error_based", SEQ=1, TOPS=4, eps=0.5)
    got = q.pteq_batch(init, 0.1, Nc=Nc, code=code, scan="colour", replicas=R, **kw)
    ref = orc.pteq_batch(ocode, np.repeat(init, R, axis=0), 0.1, Nc, kw.pop("steps"), scan=2, **kw)
    ncls = ref["counts"].shape[1]
    assert np.array_equal(got["counts"], ref["counts"].reshape(N, R, ncls).sum(axis=1))
    assert np.array_equal(got["samples"], ref["samples"].reshape(N, R).sum(axis=1).astype(np.uint32))
    assert np.array_equal(got["tops0"], ref["tops0"].reshape(N, R).sum(axis=1).astype(np.uint32))
    assert np.array_equal(got["steps_done"], ref["steps_done"].reshape(N, R).max(axis=1).astype(np.uint32))
    assert np.array_equal(got["converged"], ref["converged"].reshape(N, R).all(axis=1))
    assert got["converged"].any() and np.unique(got["steps_done"]).size > 3


def test_pteq_dropin_in_the_latency_layout(q, orc):
    """decoders.PTEQ(code, p, scan="colour") with the reference's default criterion: one syndrome, one workgroup"""
    from qecmc import decoders
    rng = np.random.default_rng(8)
    code = q.Toric_code(5)
    code.qubit_matrix = _rand(rng, (2, 5, 5), 0.1)
    pct = q.PTEQ(code, 0.1, seed=19, scan="colour")
    ref = orc.pteq(orc.TORIC, code.qubit_matrix, 0.1, Nc=5, steps=1 << 22, conv_criteria="error_based", rng=orc.Rng.philox(19, 0), scan=2)
    assert ref["converged"] and np.array_equal(pct, ref["percent"])


def test_colour_scan_conserves_the_syndrome(q):
    from qecmc import toric_model as tm
    rng = np.random.default_rng(3)
    init = _rand(rng, (8, 2, 9, 9), 0.15)
    got = q.pteq_batch(init, 0.15, Nc=8, steps=500, scan="colour", return_states=True, seed=5)
    for s in range(8):
        ref = tm.syndrome(init[s])
        for c in range(8):
            assert np.array_equal(tm.syndrome(got["states"][s, c]), ref)


def test_colour_scan_rejects_what_it_does_not_do(q):
    init = np.zeros((1, 2, 5, 5), np.uint8)
    with pytest.raises(q.QecmcError):
        q.pteq_batch(np.zeros((1, 5, 5), np.uint8), 0.1, Nc=5, steps=100, scan="sweep", code=q.XZZX, eta=10.0)     # the sweep scan: depolarizing rule only
    with pytest.raises(q.QecmcError):
        q.pteq_batch(np.zeros((1, 5, 5), np.uint8), 0.1, Nc=1, steps=100, scan="colour", code=q.XZZX, alpha=2.0)   # Ladder_alpha needs its top rung at pz_tilde = 1
    with pytest.raises(q.QecmcError):
        q.pteq_batch(init, 0.1, Nc=1, steps=100, scan="colour", p_logical=0.5)          # a 1-rung ladder's top sits below p = 0.75


# ---- scan = colour under the biased / alpha rules (VERDICT r3 "missing" 1: the latency kernel for config 4's rule and PTEQ_alpha's) ----------
RULE_CASES = [  # name, L, Nc, N, steps, iters, p, rule kwargs
    ("xzzx", 3, 3, 6, 300, 10, 0.25, dict(eta=3.0)), ("xzzx", 9, 8, 4, 120, 10, 0.15, dict(eta=100.0)), ("rotated", 7, 5, 4, 150, 7, 0.2, dict(eta=10.0)),
    ("xzzx", 5, 1, 5, 200, 10, 0.2, dict(eta=10.0)),                                        # a one-rung ladder: the biased top rule with its tested logical operators
    ("xzzx", 5, 5, 6, 250, 10, 0.175, dict(alpha=4.04)), ("rotated", 9, 7, 3, 100, 10, 0.15, dict(alpha=2.0)), ("xzzx", 11, 6, 2, 60, 3, 0.1, dict(alpha=1.3)),
    ("rotated", 21, 8, 2, 30, 10, 0.12, dict(eta=100.0))]


@pytest.mark.parametrize("name,L,Nc,N,steps,iters,p,rule", RULE_CASES)
@pytest.mark.parametrize("conv", [False, True])
def test_colour_scan_biased_and_alpha_bit_exact(q, orc, name, L, Nc, N, steps, iters, p, rule, conv):
    """The colour phases under the rules of src/mcmc_biased.py / src/mcmc_alpha.py: each generator a Metropolis move for the model's own weight (the
    reference's rule at iters = 1, where Q3 is vacuous), the biased top rung's logical operators tested, Ladder_alpha's top on the coin, its swap test on
    slot-bound attributes (Q4) -- against the oracle's restatement (its scan = 2): class counts, samples, tops0, every rung's final configuration, and with
    the error_based criterion the stopping step and flag."""
    rng = np.random.default_rng(L * 11 + Nc + N)
    code, ocode = {"xzzx": (q.XZZX, orc.XZZX), "rotated": (q.ROTATED, orc.ROTATED)}[name]
    init = _rand(rng, (N, L, L), 0.12)
    kw = dict(steps=steps * (10 if conv else 1), iters=iters, tops_burn=1 if conv else 0, seed=123, first_syndrome=5)
    if conv:
        kw.update(conv_criteria="error_based", SEQ=1, TOPS=3, eps=0.6)
    okw = dict(noise=orc.BIASED, eta=rule["eta"]) if "eta" in rule else dict(noise=orc.ALPHA, alpha=rule["alpha"], det_pow=1)
    got = q.pteq_batch(init, p, Nc=Nc, code=code, scan="colour", return_states=not conv, **rule, **kw)
    ref = orc.pteq_batch(ocode, init, p, Nc, kw.pop("steps"), return_states=True, scan=2, **okw, **kw)
    for k in ("counts", "samples", "tops0") + (("steps_done", "converged") if conv else ()):
        assert np.array_equal(np.asarray(got[k]).astype(np.uint64), np.asarray(ref[k]).astype(np.uint64)), k
    if not conv:
        assert np.array_equal(got["states"], ref["states"]) and not np.array_equal(got["states"][:, 0], init)
    assert got["counts"].sum() > 0


@pytest.mark.parametrize("name,seed,p,eta,iters", [("xzzx", 51, 0.25, 3.0, 10), ("xzzx", 53, 0.15, 100.0, 10), ("rotated", 52, 0.30, 10.0, 7)])
def test_colour_scan_biased_exact_law_L3(q, name, seed, p, eta, iters):
    """With every generator a Metropolis move for px^nx py^ny pz^nz pI^nI and the swap rule's exact exchange ratio on total counts, the
    bottom rung samples the biased class law -- at any `iters`: the colour rule has no Q3.  Exact enumeration at L = 3, 4 096 replicas, 5 sigma."""
    from util_exact import SurfEnumeration, biased_weight
    from test_gpu_stats import _rand_surf, _surf_api, _class_fractions
    code = {"xzzx": q.XZZX, "rotated": q.ROTATED}[name]
    init = _rand_surf(seed)
    P = SurfEnumeration(code, init, _surf_api(q)).class_probabilities(biased_weight(p, eta))
    R, steps = 4096, 3000
    res = q.pteq_batch(np.broadcast_to(init, (R,) + init.shape).copy(), p, Nc=3, steps=steps, iters=iters, tops_burn=20, seed=7000 + seed, code=code, eta=eta, scan="colour")
    ok = res["samples"] > steps // 2
    assert ok.mean() > 0.9
    mean, sem = _class_fractions(res, ok)
    assert np.all(np.abs(mean - P) <= 5 * sem + 2e-4), (mean, P, sem)
    if np.sort(P)[-1] - np.sort(P)[-2] > 0.01:
        assert mean.argmax() == P.argmax()


def test_pteq_alpha_dropin_in_the_latency_layout(q, orc):
    """PTEQ_alpha(code, pz_tilde, alpha, scan="colour"): the one-syndrome call generate_data.py:142-150 makes for biased noise, in the layout built for it"""
    rng = np.random.default_rng(3)
    code = q.xzzx_code(5)
    code.qubit_matrix = _rand(rng, (5, 5), 0.1)
    pct = q.PTEQ_alpha(code, 0.175, alpha=4.04, Nc=5, steps=400, conv_criteria=None, seed=21, scan="colour")
    ref = orc.pteq_batch(orc.XZZX, code.qubit_matrix[None], 0.175, 5, 400, seed=21, scan=2, noise=orc.ALPHA, alpha=4.04, det_pow=1)
    assert np.array_equal(pct, (np.divide(ref["counts"][0], max(int(ref["samples"][0]), 1)) * 100).astype(np.uint8))
    pct = q.PTEQ_biased(code, 0.15, eta=100.0, Nc=5, steps=400, conv_criteria=None, seed=22, scan="colour")
    ref = orc.pteq_batch(orc.XZZX, code.qubit_matrix[None], 0.15, 5, 400, seed=22, scan=2, noise=orc.BIASED, eta=100.0)
    assert np.array_equal(pct, (np.divide(ref["counts"][0], max(int(ref["samples"][0]), 1)) * 100).astype(np.uint8))
