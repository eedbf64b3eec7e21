"""GPU parity for the "alpha" noise model (src/mcmc_alpha.py, PTEQ_alpha): chains / ladders / PTEQ bit for bit against
the oracle on the same Philox stream.  The oracle runs with det_pow=1: the swap test's power is taken as the
deterministic exp(e ln b) the kernel uses (tests/test_oracle_golden_alpha.py shows it makes the reference's decisions)."""
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


def rand_states(rng, N, L, p):
    m = np.zeros((N, L, L), dtype=np.uint8)
    err = rng.random(m.shape) < p
    m[err] = rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)
    return m


def _cls(q, name):
    return (q.xzzx_code, q.XZZX) if name == "xzzx" else (q.RotSurCode, q.ROTATED)


@pytest.mark.parametrize("name,L,pzt,alpha,p_logical,iters", [
    ("xzzx", 5, 0.1, 1.7, 0.0, 600), ("xzzx", 9, 0.5, 1.3, 0.5, 400), ("rot", 7, 0.12, 3.1, 0.0, 400), ("xzzx", 9, 1.0, 2.0, 0.5, 300),
    ("xzzx", 9, 0.01, 2.0, 0.0, 50)])
def test_chain_alpha_bit_exact(q, orc, name, L, pzt, alpha, p_logical, iters):
    cls, cid = _cls(q, name)
    rng = np.random.default_rng(L * 17 + iters)
    m = rand_states(rng, 1, L, 0.15)[0]
    seed, stream, slot, k0 = 0x5EED0A1FA, 4, 1, 321
    code = cls(L)
    code.qubit_matrix = m.copy()
    ch = q.Chain_alpha(pzt, alpha, code, seed=seed, stream=stream)
    ch.p_logical, ch.slot, ch.proposals_done = p_logical, slot, k0
    ch.update_chain(iters)
    ref, ne = orc.chain_update_alpha(cid, m, pzt, alpha, p_logical, iters, orc.Rng.philox(seed, stream), slot=slot, k0=k0)
    assert np.array_equal(ch.code.qubit_matrix, ref)
    assert ch.n_eff == ne                                                   # refreshed iff a move was accepted


@pytest.mark.parametrize("name,L,pzt,alpha,Nc,iters,nstep", [
    ("xzzx", 3, 0.2, 2.0, 3, 5, 60), ("xzzx", 9, 0.08, 2.5, 8, 10, 40), ("rot", 7, 0.12, 3.1, 6, 10, 40), ("xzzx", 5, 0.1, 1.0, 16, 4, 30),
    ("rot", 5, 0.3, 1.5, 1, 10, 20), ("xzzx", 9, 0.02, 1.2, 8, 1, 80)])
def test_ladder_alpha_bit_exact(q, orc, name, L, pzt, alpha, Nc, iters, nstep):
    cls, cid = _cls(q, name)
    rng = np.random.default_rng(L + Nc)
    m = rand_states(rng, 1, L, 0.15)[0]
    seed, stream = 97531, 3
    code = cls(L)
    code.qubit_matrix = m.copy()
    ld = q.Ladder_alpha(pzt, code, alpha, Nc, 0.5, seed=seed, stream=stream)
    ref = orc.Ladder(cid, m, pzt, Nc, 0.5, noise=orc.ALPHA, alpha=alpha, det_pow=1)
    r = orc.Rng.philox(seed, stream)
    assert np.array_equal(ld.pz_tilde_ladder, ref.p_ladder)
    done = 0
    for chunk in (1, 2, nstep - 3):
        ld.step(iters, nsteps=chunk)
        for _ in range(chunk):
            ref.step(iters, r)
        done += chunk
        got = np.stack([c.code.qubit_matrix for c in ld.chains])
        assert np.array_equal(got, ref.states), f"states differ after {done} steps"
        assert [c.flag for c in ld.chains] == ref.flags.tolist() and ld.tops0 == ref.tops0
        assert [c.n_eff for c in ld.chains] == ref.n_eff.tolist()           # slot-bound, possibly stale (Q4)
        assert [[c._nz, c._nxy] for c in ld.chains] == ref.n_eff_counts.tolist()


@pytest.mark.parametrize("name,L,pzt,alpha,Nc,N,steps,tops_burn,conv", [
    ("xzzx", 5, 0.1, 1.7, 5, 70, 200, 1, None), ("xzzx", 9, 0.08, 2.5, 8, 65, 100, 0, None), ("rot", 7, 0.12, 3.1, 6, 33, 100, 2, None),
    ("xzzx", 3, 0.2, 2.0, 3, 50, 4000, 1, "error_based"), ("xzzx", 5, 0.1, 1.7, 5, 40, 3000, 2, "error_based")])
def test_pteq_alpha_batch_bit_exact(q, orc, name, L, pzt, alpha, Nc, N, steps, tops_burn, conv):
    cls, cid = _cls(q, name)
    rng = np.random.default_rng(N * 5 + L)
    init = rand_states(rng, N, L, 0.12)
    kw = dict(steps=steps, iters=10, tops_burn=tops_burn, seed=424242, first_syndrome=11, conv_criteria=conv)
    if conv:
        kw.update(SEQ=1, TOPS=4, eps=0.6)
    got = q.pteq_batch(init, pzt, Nc=Nc, code=cid, alpha=alpha, return_states=conv is None, **kw)
    ref = orc.pteq_batch(cid, init, pzt, Nc, kw.pop("steps"), return_states=True, noise=orc.ALPHA, alpha=alpha, det_pow=1, **kw)
    assert np.array_equal(got["tops0"], ref["tops0"].astype(np.uint32))
    assert np.array_equal(got["samples"], ref["samples"].astype(np.uint32))
    assert np.array_equal(got["counts"], ref["counts"]) and got["counts"].shape == (N, 4)
    assert np.array_equal(got["steps_done"], ref["steps_done"].astype(np.uint32))
    if conv is None:
        assert np.array_equal(got["states"], ref["states"])
    else:
        assert np.array_equal(got["converged"], ref["converged"]) and got["converged"].any()


def test_pteq_alpha_dropin(q, orc):
    """PTEQ_alpha with the reference's signature (decoders_biasednoise.py:175), incl. its default convergence criterion."""
    rng = np.random.default_rng(8)
    code = q.xzzx_code(5)
    code.qubit_matrix = rand_states(rng, 1, 5, 0.15)[0]
    pct = q.PTEQ_alpha(code, 0.1, alpha=1.7, Nc=5, steps=300, conv_criteria=None, seed=21, replicas=1)
    ref = orc.pteq(q.XZZX, code.qubit_matrix, 0.1, Nc=5, steps=300, rng=orc.Rng.philox(21, 0), noise=orc.ALPHA, alpha=1.7, det_pow=1)
    assert pct.shape == (4,) and np.array_equal(pct, ref["percent"])
    pct = q.PTEQ_alpha(code, 0.1, alpha=1.7, Nc=5, steps=200000, seed=22, replicas=1)
    ref = orc.pteq(q.XZZX, code.qubit_matrix, 0.1, Nc=5, steps=200000, conv_criteria="error_based", rng=orc.Rng.philox(22, 0),
                   noise=orc.ALPHA, alpha=1.7, det_pow=1)
    assert np.array_equal(pct, ref["percent"]) and ref["converged"]


def test_alpha_rejects(q):
    with pytest.raises(q.QecmcError):
        q.pteq_batch(np.zeros((1, 2, 5, 5), np.uint8), 0.1, Nc=3, alpha=2.0)          # toric: not built
    with pytest.raises(q.QecmcError):
        q.pteq_batch(np.zeros((1, 5, 5), np.uint8), 1.5, Nc=3, code=q.XZZX, alpha=2.0)   # pz_tilde > 1
    with pytest.raises(q.QecmcError):
        q.pteq_batch(np.zeros((1, 5, 5), np.uint8), 0.1, Nc=3, code=q.XZZX, alpha=0.0)


@pytest.mark.parametrize("name,L,pzt,alpha,Nc,steps,conv,kw", [("xzzx", 5, 0.1, 1.7, 5, 250, None, dict(tops_burn=0)),
                                                               ("rotated", 5, 0.2, 2.0, 4, 600, None, dict(tops_burn=1)),
                                                               ("xzzx", 3, 0.3, 2.0, 3, 3000, "error_based", dict(eps=0.6))])
def test_pteq_alpha_with_shortest(q, orc, name, L, pzt, alpha, Nc, steps, conv, kw):
    """PTEQ_alpha_with_shortest (decoders_biasednoise.py:93-172): the GPU ladder, stepped once per launch under the host
    bookkeeping, against the oracle's ladder under the same bookkeeping (itself pinned to the reference, f_nalpha.npz)."""
    from qecmc.decoders_biasednoise import _shortest_loop
    from util_shortest import OracleLadderAlpha
    rng = np.random.default_rng(L + Nc)
    code = (q.xzzx_code if name == "xzzx" else q.RotSurCode)(L)
    code.qubit_matrix = (rng.integers(1, 4, size=(L, L)) * (rng.random((L, L)) < 0.15)).astype(np.uint8)
    got = q.PTEQ_alpha_with_shortest(code, pzt, alpha=alpha, Nc=Nc, steps=steps, conv_criteria=conv, seed=77, **kw)
    ld = OracleLadderAlpha(orc.XZZX if name == "xzzx" else orc.ROTATED, code.qubit_matrix, pzt, alpha, Nc, orc.Rng.philox(77, 0), det_pow=1)
    ref = _shortest_loop(ld, pzt, kw.get("SEQ", 2), kw.get("TOPS", 10), kw.get("tops_burn", 2), kw.get("eps", 0.1), steps, 10, conv)
    assert got[0].dtype == np.uint8 and np.array_equal(got[0], ref[0])
    assert np.allclose(got[1], ref[1], rtol=1e-12, equal_nan=True) and np.allclose(got[2], ref[2], rtol=1e-12, equal_nan=True)
    assert got[0].sum() > 90
