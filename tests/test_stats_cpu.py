"""Statistical pins of the oracle in Philox mode (the mode the GPU is compared with bit for bit):
exact enumeration at L=3, and the reference's own replica-averaged histograms (fixture F3)."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from conftest import GOLDEN
from util_exact import toric_class_probabilities


def _rand_state(seed, L, p):
    rng = np.random.default_rng(seed)
    m = np.zeros((2, L, L), dtype=np.uint8)
    err = rng.random(m.shape) < p
    m[err] = rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)
    return m


def test_exact_enumeration_is_representative_independent():
    M = np.array([[[0, 2, 0], [0, 0, 1], [3, 0, 0]], [[1, 0, 0], [0, 3, 0], [0, 0, 2]]], dtype=np.uint8)
    P = toric_class_probabilities(M, 0.1, orc.toric_apply_stabilizer, orc.toric_to_class)
    assert abs(P.sum() - 1) < 1e-12 and (P > 0).all()
    # any other chain with the same syndrome (a stabilizer or a logical operator away) gives the same answer
    M2, _ = orc.toric_apply_stabilizer(M, 1, 2, 3)
    M3 = orc.toric_to_class(M2, 9)
    for other in (M2, M3):
        assert np.allclose(P, toric_class_probabilities(other, 0.1, orc.toric_apply_stabilizer, orc.toric_to_class), rtol=1e-12)


@pytest.mark.parametrize("seed,p,Nc", [(1, 0.10, 3), (2, 0.15, 4)])
def test_oracle_philox_matches_exact_enumeration_L3(seed, p, Nc):
    init = _rand_state(seed, 3, 0.15)
    P = toric_class_probabilities(init, p, orc.toric_apply_stabilizer, orc.toric_to_class)
    R, steps = 96, 6000
    res = orc.toric_pteq_batch(np.broadcast_to(init, (R,) + init.shape).copy(), p, Nc, steps, iters=10, tops_burn=5,
                               seed=100 + seed, n_threads=8)
    assert (res["samples"] > steps // 2).all()
    frac = res["counts"] / res["samples"][:, None].astype(np.float64)
    mean, sem = frac.mean(axis=0), frac.std(axis=0, ddof=1) / np.sqrt(R)
    # classes that carry weight: 5 sigma.  The rare ones (P < 1 %) are visited in a few long bursts per replica, so the
    # spread over 96 replicas underestimates their error (|z| of 5-9 turn up for one class or another with most seeds
    # while the mean over seeds sits on P): they get a relative allowance, and the whole vector a total-variation bound
    big = P >= 0.01
    assert np.all(np.abs(mean - P)[big] <= 5 * sem[big] + 2e-4), (mean, P, sem)
    assert np.all(np.abs(mean - P)[~big] <= 5 * sem[~big] + 0.75 * P[~big] + 2e-4), (mean, P, sem)
    assert 0.5 * np.abs(mean - P).sum() < 0.05           # (the dominant class alone has sem ~ 0.017 here)


def _f3_protocol_oracle(init, p, Nc, iters, steps, burn, seed, syndrome):
    ld = orc.ToricLadder(init, p, Nc, 0.5)
    rng = orc.Rng.philox(seed, syndrome)
    hist = np.zeros(16)
    nerr = np.zeros(Nc)
    for t in range(steps):
        ld.step(iters, rng)
        if t >= burn:
            st = ld.states
            hist[orc.toric_eq_class(st[0])] += 1
            nerr += [orc.count_errors(s) for s in st]
    return hist / (steps - burn), nerr / (steps - burn)


@pytest.mark.parametrize("name", ["L3", "L5"])
def test_oracle_philox_matches_reference_histograms(name):
    """Same protocol as the fixture (fixed burn-in, R replicas): replica means of the bottom-chain class
    histogram and of the per-rung error counts agree with the reference within the combined standard error."""
    g = np.load(os.path.join(GOLDEN, "f3_toric.npz"))
    L, p, Nc, iters, steps, burn = g[f"{name}_par"]
    L, Nc, iters = int(L), int(Nc), int(iters)
    steps_o, burn_o, R = 3000, 500, 12                    # shorter oracle runs keep the CPU suite fast
    for s in range(g[f"{name}_init"].shape[0]):
        ref_h = g[f"{name}_hist"][s] / (steps - burn)
        ref_n = g[f"{name}_nerr"][s]
        runs = [_f3_protocol_oracle(g[f"{name}_init"][s], float(p), Nc, iters, steps_o, burn_o, 900 + s, r) for r in range(R)]
        oh = np.array([h for h, _ in runs]); on = np.array([n for _, n in runs])
        se_h = np.sqrt(ref_h.var(axis=0, ddof=1) / ref_h.shape[0] + oh.var(axis=0, ddof=1) / R)
        se_n = np.sqrt(ref_n.var(axis=0, ddof=1) / ref_n.shape[0] + on.var(axis=0, ddof=1) / R)
        assert np.all(np.abs(ref_h.mean(axis=0) - oh.mean(axis=0)) <= 4.5 * se_h + 5e-3), (s, ref_h.mean(0), oh.mean(0))
        assert np.all(np.abs(ref_n.mean(axis=0) - on.mean(axis=0)) <= 4.5 * se_n + 0.05), (s, ref_n.mean(0), on.mean(0))


@pytest.mark.parametrize("top", [False, True])
@pytest.mark.parametrize("code,L,G", [("toric", 4, 32), ("xzzx", 5, 24), ("rot", 3, 8), ("planar", 4, 24)])
def test_philox_generator_pick_is_uniform_over_generators(code, L, G, top):
    """Philox mode spends one word on the reference's generator choice (toric_model.py:291-295: row, col, op;
    xzzx_model.py:439-452: five draws): every one of the G generators must come up, equally often.  At p = 0.75 each
    proposal is accepted, so one proposal from the empty configuration shows which generator was picked."""
    cid = {"toric": orc.TORIC, "xzzx": orc.XZZX, "rot": orc.ROTATED, "planar": orc.PLANAR}[code]
    n = G * 400
    seen = {}
    zero = np.zeros((2, L, L) if code in ("toric", "planar") else (L, L), np.uint8)
    rng = orc.Rng.philox(77, 5)
    for k in range(n):
        m = orc.chain_update(cid, zero, 0.75, 1e-300 if top else 0.0, 1, rng, slot=3, k0=k)   # tiny p_logical: top-branch addressing
        seen[m.tobytes()] = seen.get(m.tobytes(), 0) + 1
    if top:
        # the top chain's packed layout has a 16-bit select: with p_logical -> 0 a logical operator still comes up once in 2^16
        # proposals (0.2 expected here); such a state is seen once or twice, a generator ~400 times
        rare = [k for k, v in seen.items() if v < 20]
        assert len(rare) <= 2
        for k in rare:
            del seen[k]
    for k in seen:
        assert np.count_nonzero(np.frombuffer(k, np.uint8)) in (2, 3, 4)
    assert len(seen) == G
    cnt = np.array(list(seen.values()), dtype=np.float64)
    chi2 = np.sum((cnt - n / G) ** 2 / (n / G))
    assert chi2 < G + 6 * np.sqrt(2 * G)      # mean G-1, sd sqrt(2(G-1)): a > 6 sigma excess would be a broken decode


@pytest.mark.parametrize("code,L,rates", [("toric", 9, (0.05, 0.05, 0.05)), ("xzzx", 9, (7.4257e-4, 7.4257e-4, 0.148515)), ("rotated", 7, (0.17 / 3,) * 3),
                                          ("planar", 5, (0.03, 0.03, 0.03))])
def test_oracle_syndrome_generator_distribution(code, L, rates):
    """The oracle's Philox-mode restatement of generate_random_error + apply_random_logical (generate_data.py:110-131): Pauli
    rates within binomial error, the hiding operator keeps the syndrome and spreads the classes, the raw class is what is
    reported, planar layer 1 keeps its idle row / column."""
    cid = {"toric": orc.TORIC, "xzzx": orc.XZZX, "rotated": orc.ROTATED, "planar": orc.PLANAR}[code]
    N = 6000
    init, raw, eq = orc.generate_syndromes(cid, L, N, *rates, hide_class=True, seed=11, first_syndrome=100)
    px, py, pz = rates
    live = np.ones(raw.shape[1:], dtype=bool)
    if code == "planar":
        live[1, -1, :] = False; live[1, :, -1] = False
        assert not raw[:, 1, -1, :].any() and not raw[:, 1, :, -1].any()
    n = N * int(live.sum())
    for val, pr in ((1, px), (2, py), (3, pz)):
        k = int((raw[:, live] == val).sum())
        assert abs(k - n * pr) <= 5 * np.sqrt(n * pr * (1 - pr)) + 1, (code, val, k / n, pr)
    ncls = 16 if code == "toric" else 4
    for s in range(0, N, 500):
        if code == "toric":
            assert orc.toric_eq_class(raw[s]) == eq[s] and np.array_equal(orc.toric_syndrome(init[s]), orc.toric_syndrome(raw[s]))
        else:
            assert orc.surf_eq_class(cid, raw[s]) == eq[s]
            if code != "planar":
                assert np.array_equal(orc.surf_syndrome(cid, init[s]), orc.surf_syndrome(cid, raw[s]))
    hidden = np.array([orc.toric_eq_class(m) if code == "toric" else orc.surf_eq_class(cid, m) for m in init[:2000]])
    # class change = the applied operator: uniform over the ncls operators (toric: 4 x 4 per layer pair; others: 4), odd L
    delta = np.bincount(hidden ^ eq[:2000] if code in ("toric", "rotated", "planar") else (hidden - eq[:2000]) % 4, minlength=ncls)
    assert delta.min() > 2000 / ncls * 0.6, delta
    # a second call with another first_syndrome continues the same data set
    a, _, _ = orc.generate_syndromes(cid, L, 10, *rates, hide_class=True, seed=11, first_syndrome=105)
    assert np.array_equal(a, init[5:15])


# ---- exact pins for the plaquette codes and the biased weights ("2^8 elements x 4" at L = 3) -----------------
# These pin the Philox re-parametrisation of the xzzx / rotated paths (one 20-bit pick instead of five draws, packed top-chain
# words) to a reference-independent answer.  The same cases run on the GPU with 4096 replicas (tests/test_gpu_stats.py).
import types

from util_exact import SurfEnumeration, biased_weight, depolarizing_weight

ORC_API = types.SimpleNamespace(apply_stabilizer=orc.surf_apply_stabilizer, apply_logical=orc.surf_apply_logical,
                                eq_class=orc.surf_eq_class, ngen=orc.surf_ngen, gen_rco=orc.surf_gen_rco)


def _rand_surf(seed, L=3, p=0.3):
    rng = np.random.default_rng(seed)
    return (rng.integers(1, 4, size=(L, L)) * (rng.random((L, L)) < p)).astype(np.uint8)


def _check_classes(frac, P, nsig=5.0, floor=3e-4, rare=0.75):
    """class frequencies of R replicas against the exact law: 5 sigma on the classes that carry weight, a relative allowance on the
    rare ones (visited in a few long bursts per replica, so the spread over the replicas understates their error)"""
    mean, sem = frac.mean(axis=0), frac.std(axis=0, ddof=1) / np.sqrt(frac.shape[0])
    big = P >= 0.01
    assert np.all(np.abs(mean - P)[big] <= nsig * sem[big] + floor), (mean, P, sem)
    assert np.all(np.abs(mean - P)[~big] <= nsig * sem[~big] + rare * P[~big] + floor), (mean, P, sem)
    return mean, sem


@pytest.mark.parametrize("code", [orc.XZZX, orc.ROTATED])
def test_plaquette_enumeration_is_representative_independent(code):
    m = _rand_surf(5)
    e = SurfEnumeration(code, m, ORC_API)
    P = e.class_probabilities(depolarizing_weight(0.2))
    assert abs(P.sum() - 1) < 1e-12 and (P > 0).all()
    m2 = orc.surf_apply_logical(code, orc.surf_apply_stabilizer(code, m, 1, 1, 1)[0], 3, 0, 0)[0]     # same syndrome, another class
    assert np.allclose(P, SurfEnumeration(code, m2, ORC_API).class_probabilities(depolarizing_weight(0.2)), rtol=1e-12)
    # the Q3 law at iters = 1 is the plain biased law: one proposal per call is tested against the current configuration
    w = biased_weight(0.25, 3.0)
    assert np.allclose(e.q3_class_law(w, 0.5, 1), e.class_probabilities(w), atol=1e-9)


@pytest.mark.parametrize("code,seed,p,Nc", [(orc.XZZX, 11, 0.20, 3), (orc.ROTATED, 12, 0.25, 4)])
def test_oracle_plaquette_depolarizing_exact_L3(code, seed, p, Nc):
    init = _rand_surf(seed)
    P = SurfEnumeration(code, init, ORC_API).class_probabilities(depolarizing_weight(p))
    R, steps = 96, 6000
    res = orc.pteq_batch(code, np.broadcast_to(init, (R,) + init.shape).copy(), p, Nc, steps, iters=10, tops_burn=5, seed=300 + seed, n_threads=8)
    assert (res["samples"] > steps // 2).all()
    _check_classes(res["counts"] / res["samples"][:, None].astype(np.float64), P)


@pytest.mark.parametrize("code,seed,p,eta", [(orc.XZZX, 21, 0.25, 3.0), (orc.ROTATED, 22, 0.30, 10.0)])
def test_oracle_biased_ladder_exact_L3_iters1(code, seed, p, eta):
    """Ladder_biased at iters = 1, where quirk Q3 is vacuous: every rung's chain is a Metropolis chain for its own weights
    px^nx py^ny pz^nz pI^nI, and the swap rule p_diff^(n_hi - n_lo) on TOTAL counts (mcmc_biased.py:107-113) is the exact
    exchange ratio for them (the eta-dependent factors do not depend on the rung) -- so the bottom rung samples the biased law."""
    init = _rand_surf(seed)
    P = SurfEnumeration(code, init, ORC_API).class_probabilities(biased_weight(p, eta))
    R, steps = 96, 40000
    # (tops_burn = 50: past the transient of the first arrivals from the top, see tests/test_gpu_stats.py)
    res = orc.pteq_batch(code, np.broadcast_to(init, (R,) + init.shape).copy(), p, 3, steps, iters=1, tops_burn=50, seed=400 + seed, n_threads=8,
                         noise=orc.BIASED, eta=eta)
    assert (res["samples"] > steps // 2).all()
    _check_classes(res["counts"] / res["samples"][:, None].astype(np.float64), P)


@pytest.mark.parametrize("code,seed,p,eta", [(orc.XZZX, 31, 0.25, 3.0), (orc.ROTATED, 32, 0.30, 10.0)])
def test_oracle_biased_chain_q3_law_L3_iters10(code, seed, p, eta):
    """What iters = 10 converges to instead: the acceptance of every proposal of a call is relative to the configuration at the
    START of the call (quirk Q3), so detailed balance for the biased weights is lost.  For a single chain the limit is still
    computable exactly (SurfEnumeration.q3_class_law): the sampler must sit on THAT law, which differs visibly from the biased one."""
    init = _rand_surf(seed)
    e = SurfEnumeration(code, init, ORC_API)
    w = biased_weight(p, eta)
    Q, P = e.q3_class_law(w, 0.5, 10), e.class_probabilities(w)
    assert 0.5 * np.abs(Q - P).sum() > 0.02                       # the two laws are not the same thing
    R, steps = 96, 6000
    res = orc.pteq_batch(code, np.broadcast_to(init, (R,) + init.shape).copy(), p, 1, steps, iters=10, tops_burn=500, seed=500 + seed, n_threads=8,
                         noise=orc.BIASED, eta=eta)       # (a 1-rung ladder counts every step as a top: the first 500 calls are discarded)
    assert (res["samples"] == steps - 499).all()
    mean, sem = _check_classes(res["counts"] / res["samples"][:, None].astype(np.float64), Q)
    assert np.abs(mean - P).max() > 10 * sem.max()                # ... and measurably not on the biased law


@pytest.mark.parametrize("code,seed,p,Nc", [(orc.TORIC, 1, 0.10, 3), (orc.XZZX, 11, 0.20, 3)])
def test_oracle_colour_scan_matches_exact_enumeration_L3(code, seed, p, Nc):
    """scan = 2 (one colour phase of mutually disjoint generators at a time, the GPU's latency layout): the same stationary law"""
    if code == orc.TORIC:
        init = _rand_state(seed, 3, 0.15)
        P = toric_class_probabilities(init, p, orc.toric_apply_stabilizer, orc.toric_to_class)
    else:
        init = _rand_surf(seed)
        P = SurfEnumeration(code, init, ORC_API).class_probabilities(depolarizing_weight(p))
    R, steps = 96, 6000
    res = orc.pteq_batch(code, np.broadcast_to(init, (R,) + init.shape).copy(), p, Nc, steps, iters=10, tops_burn=5, seed=600 + seed, n_threads=8, scan=2)
    assert (res["samples"] > steps // 2).all()
    _check_classes(res["counts"] / res["samples"][:, None].astype(np.float64), P)


@pytest.mark.parametrize("code,L", [(orc.TORIC, 3), (orc.TORIC, 4), (orc.TORIC, 9), (orc.XZZX, 5), (orc.ROTATED, 7), (orc.PLANAR, 5)])
def test_oracle_colour_phases_partition_the_generators_into_disjoint_sets(code, L):
    ph = orc.colour_phases(code, L)
    G = 2 * L * L if code == orc.TORIC else orc.surf_ngen(code, L)
    members = ph[ph >= 0]
    assert sorted(members.tolist()) == list(range(G))                    # every generator exactly once
    zero = np.zeros((2, L, L) if code in (orc.TORIC, orc.PLANAR) else (L, L), np.uint8)
    for row in ph:
        seen = np.zeros(zero.size, bool)
        for g in row[row >= 0]:
            if code == orc.TORIC:
                pat = orc.toric_apply_stabilizer(zero, (g % (L * L)) // L, g % L, 1 if g < L * L else 3)[0]
            else:
                pat = orc.surf_apply_stabilizer(code, zero, *orc.surf_gen_rco(code, L, int(g)))[0]
            q = pat.ravel() != 0
            assert not (seen & q).any()                                  # no qubit is touched twice within a phase
            seen |= q


# ---- scan = 3 ("wave"): the reference's random scan with a generator pick shared by the 64 ladders of a GPU wavefront ----------
@pytest.mark.parametrize("code,seed,p,Nc", [(orc.TORIC, 1, 0.10, 3), (orc.XZZX, 11, 0.20, 3), (orc.ROTATED, 12, 0.25, 4)])
def test_oracle_wave_scan_matches_exact_enumeration_L3(code, seed, p, Nc):
    """Every ladder of scan = 3 is the reference's chain: its class histogram sits on the exact law.  The replicas get a pick group
    of their own each (first_syndrome = 64 r), so that they are independent and the standard error over them means what it says."""
    if code == orc.TORIC:
        init = _rand_state(seed, 3, 0.15)
        P = toric_class_probabilities(init, p, orc.toric_apply_stabilizer, orc.toric_to_class)
    else:
        init = _rand_surf(seed)
        P = SurfEnumeration(code, init, ORC_API).class_probabilities(depolarizing_weight(p))
    R, steps = 96, 6000
    counts, samples = [], []
    for r in range(R):
        res = orc.pteq_batch(code, init[None].copy(), p, Nc, steps, iters=10, tops_burn=5, seed=700 + seed, first_syndrome=64 * r, n_threads=1, scan=3)
        counts.append(res["counts"][0]); samples.append(res["samples"][0])
    counts, samples = np.array(counts), np.array(samples)
    assert (samples > steps // 2).all()
    _check_classes(counts / samples[:, None].astype(np.float64), P)


def test_oracle_wave_scan_shares_its_picks_within_a_group_of_64_only():
    """A chain at f = 1 accepts every proposal (mcmc.py:42 with factor 1), so its trajectory IS its sequence of generator picks: the
    same for two ladders of one group of 64, different across groups and across slots; and every generator comes up equally often."""
    L, iters, steps = 5, 10, 300
    zero = np.zeros((2, L, L), np.uint8)
    def run(syndrome, slot):
        m = zero.copy()
        for T in range(steps):
            m = orc.chain_update(orc.TORIC, m, 0.75, 0.0, iters, orc.Rng.philox(5, syndrome), slot=slot, k0=T * iters, scan=3)
        return m
    a, b, c, d = run(64, 0), run(127, 0), run(128, 0), run(64, 1)
    assert np.array_equal(a, b) and not np.array_equal(a, c) and not np.array_equal(a, d)
    # uniform over the 2 L^2 generators: apply one proposal at a time from the empty lattice and identify the generator
    G = 2 * L * L
    hits = np.zeros(G, np.int64)
    pats = {}
    for g in range(G):
        pats[orc.toric_apply_stabilizer(zero, (g % (L * L)) // L, g % L, 1 if g < L * L else 3)[0].tobytes()] = g
    n = 20000
    for k in range(n):
        hits[pats[orc.chain_update(orc.TORIC, zero.copy(), 0.75, 0.0, 1, orc.Rng.philox(9, 0), slot=2, k0=k, scan=3).tobytes()]] += 1
    exp = n / G
    assert np.abs(hits - exp).max() < 5 * np.sqrt(exp)


def test_oracle_wave_scan_acceptance_uniform_is_44_bits_against_the_reference_rule():
    """One ladder step of a cold chain, proposal by proposal: accepted iff a12 2^32 + w32 < ceil(f^dE 2^44) -- checked against a direct
    evaluation of mcmc.py:42 on the same Philox words for a few hundred (syndrome, step) and two values of iters."""
    L, p = 3, 0.2
    f = (p / 3.0) / (1.0 - p)
    rng = np.random.default_rng(3)
    n_acc = n_all = 0
    for it in range(300):
        iters = (10, 7)[it & 1]
        init = (rng.integers(1, 4, size=(2, L, L)) * (rng.random((2, L, L)) < 0.3)).astype(np.uint8)
        syn, T = int(rng.integers(0, 1 << 20)), int(rng.integers(0, 5000))
        out = orc.chain_update(orc.TORIC, init.copy(), p, 0.0, iters, orc.Rng.philox(21, syn), slot=1, k0=T * iters, scan=3)
        S, nch, nc4 = 128 // iters, (iters + 9) // 10, (iters + 3) // 4
        cur = init
        for j in range(iters):
            P = (T % S) * iters + j
            pw = orc.philox4x32_10([(T // S) * 64 + (P >> 1), 9 << 16, syn >> 6, 0x800 + 1], [21, 0])
            g = (int(pw[2 * (P & 1) + 1]) * 2 * L * L) >> 32
            new, dE = orc.toric_apply_stabilizer(cur, (g % (L * L)) // L, g % L, 1 if g < L * L else 3)
            aw = orc.philox4x32_10([T * nch + j // 10, 10 << 16, syn, 1], [21, 0]); rw = orc.philox4x32_10([T * nc4 + (j >> 2), 11 << 16, syn, 1], [21, 0])
            fld = j % 10
            a12 = (aw[fld >> 1] >> (12 * (fld & 1))) & 0xFFF if fld < 8 else (aw[0 if fld == 8 else 2] >> 24) | (((aw[1 if fld == 8 else 3] >> 24) & 0xF) << 8)
            x = (a12 << 32) + int(rw[j & 3])
            thr = (1 << 44) if f ** dE >= 1 else int(np.ceil((f ** dE) * 2.0 ** 44))
            acc = x < thr
            n_acc += acc; n_all += 1
            if acc:
                cur = new
        assert np.array_equal(out, cur)
    assert 0 < n_acc < n_all
