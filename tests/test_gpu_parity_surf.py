"""GPU parity for the XZZX and rotated surface codes and for the biased chain (BASELINE configs 4, 5):
stencils against the reference's vectors (f1_surf.npz), chains / ladders / PTEQ bit for bit against the
oracle on the same Philox stream."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

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


@pytest.mark.parametrize("name", ["xzzx", "rot"])
@pytest.mark.parametrize("L", [3, 5, 9])
@pytest.mark.parametrize("rep", [0, 1])
def test_f1_surf_stencils_on_device(q, name, L, rep):
    from qecmc import _surf
    g = np.load(os.path.join(GOLDEN, "f1_surf.npz"))
    cls, cid = _cls(q, name)
    t = f"{name}_L{L}_{rep}"
    m = g[f"{t}_m"]
    code = cls(L)
    code.qubit_matrix = m.copy()
    assert code.count_errors() == int(g[f"{t}_count"])
    assert code.define_equivalence_class() == int(g[f"{t}_class"])
    code.syndrome()
    assert np.array_equal(code.plaquette_defects, g[f"{t}_defects"])
    a = g[f"{t}_stab_arg"]
    new, dE = _surf.apply_stabilizer(cid, np.broadcast_to(m, (len(a),) + m.shape), a[:, 0], a[:, 1], a[:, 2])
    assert np.array_equal(new, g[f"{t}_stab_new"]) and np.array_equal(dE, g[f"{t}_stab_dE"])
    a = g[f"{t}_log_arg"]
    new, dE = _surf.apply_logical(cid, np.broadcast_to(m, (len(a),) + m.shape), a[:, 0], a[:, 1], a[:, 2])
    assert np.array_equal(new, g[f"{t}_log_new"]) and np.array_equal(dE, g[f"{t}_log_dE"])
    assert np.array_equal(_surf.eq_class(cid, new), g[f"{t}_log_class"])
    assert np.array_equal(code.qubit_matrix, m)
    with pytest.raises(q.QecmcError):
        code.apply_stabilizer(L - 1, 0, 1)
    with pytest.raises(q.QecmcError):
        code.apply_stabilizer((L - 1) // 2, 0, 3)


@pytest.mark.parametrize("name,L,p,p_logical,iters,eta", [
    ("rot", 5, 0.17, 0.0, 800, None), ("rot", 9, 0.3, 0.5, 600, None), ("rot", 21, 0.17, 0.0, 500, None), ("xzzx", 9, 0.75, 0.5, 500, None),
    ("xzzx", 5, 0.15, 0.0, 600, 100.0), ("xzzx", 9, 0.4, 0.5, 400, 10.0), ("rot", 7, 0.2, 0.5, 400, 3.0), ("xzzx", 9, 0.15, 0.0, 600, 100.0)])
def test_chain_update_bit_exact(q, orc, name, L, p, p_logical, iters, eta):
    cls, cid = _cls(q, name)
    rng = np.random.default_rng(L * 13 + iters)
    m = rand_states(rng, 1, L, 0.15)[0]
    seed, stream, slot, k0 = 0xABCDEF12345, 9, 2, 777
    code = cls(L)
    code.qubit_matrix = m.copy()
    ch = q.Chain(p, code, seed=seed, stream=stream) if eta is None else q.Chain_biased(p, eta, code, seed=seed, stream=stream)
    ch.p_logical, ch.slot, ch.proposals_done = p_logical, slot, k0
    ch.update_chain(iters)
    ref = orc.chain_update(cid, m, p, p_logical, iters, orc.Rng.philox(seed, stream), slot=slot, k0=k0,
                           noise=0 if eta is None else 1, eta=eta or 0.0)
    assert np.array_equal(ch.code.qubit_matrix, ref)
    from qecmc import _surf
    assert np.array_equal(_surf.syndrome(cid, ref), _surf.syndrome(cid, m))


@pytest.mark.parametrize("name,L,p,Nc,iters,nstep,eta", [
    ("rot", 3, 0.3, 4, 5, 60, None), ("rot", 9, 0.17, 8, 10, 50, None), ("xzzx", 5, 0.15, 4, 10, 60, None), ("rot", 21, 0.17, 8, 10, 20, None),
    ("xzzx", 3, 0.3, 3, 5, 60, 10.0), ("xzzx", 9, 0.15, 8, 10, 40, 100.0), ("rot", 5, 0.2, 16, 4, 30, 5.0), ("rot", 5, 0.2, 1, 10, 30, None)])
def test_ladder_step_bit_exact(q, orc, name, L, p, Nc, iters, nstep, eta):
    cls, cid = _cls(q, name)
    rng = np.random.default_rng(L + Nc)
    m = rand_states(rng, 1, L, 0.15)[0]
    seed, stream = 4321, 6
    code = cls(L)
    code.qubit_matrix = m.copy()
    ld = q.Ladder(p, code, Nc, 0.5, seed=seed, stream=stream) if eta is None else q.Ladder_biased(p, code, eta, Nc, 0.5, seed=seed, stream=stream)
    ref = orc.Ladder(cid, m, p, Nc, 0.5, noise=0 if eta is None else 1, eta=eta or 0.0)
    r = orc.Rng.philox(seed, stream)
    assert np.array_equal(ld.p_ladder, ref.p_ladder) and np.array_equal(ld.p_diff, ref.p_diff)
    done = 0
    for chunk in (1, 2, nstep - 3):
        ld.step(iters, nsteps=chunk)
        for _ in range(chunk):
            ref.step(iters, r)
        done += chunk
        got = np.stack([c.code.qubit_matrix for c in ld.chains])
        assert np.array_equal(got, ref.states), f"states differ after {done} steps"
        assert [c.flag for c in ld.chains] == ref.flags.tolist() and ld.tops0 == ref.tops0


@pytest.mark.parametrize("name,L,p,Nc,N,steps,tops_burn,eta,conv", [
    ("rot", 5, 0.17, 5, 70, 200, 1, None, None), ("rot", 9, 0.17, 8, 65, 100, 0, None, None), ("xzzx", 9, 0.15, 8, 64, 100, 0, None, None),
    ("rot", 21, 0.17, 8, 40, 30, 0, None, None), ("xzzx", 9, 0.15, 8, 96, 80, 0, 100.0, None), ("xzzx", 5, 0.15, 5, 33, 150, 2, 100.0, None),
    ("rot", 3, 0.17, 3, 50, 4000, 1, None, "error_based"), ("xzzx", 5, 0.15, 5, 40, 3000, 2, 100.0, "error_based")])
def test_pteq_batch_bit_exact(q, orc, name, L, p, Nc, N, steps, tops_burn, eta, conv):
    cls, cid = _cls(q, name)
    rng = np.random.default_rng(N * 3 + L)
    init = rand_states(rng, N, L, p)
    kw = dict(steps=steps, iters=10, tops_burn=tops_burn, seed=31337, first_syndrome=5, conv_criteria=conv)
    if conv:
        kw.update(SEQ=1, TOPS=4, eps=0.6)
    got = q.pteq_batch(init, p, Nc=Nc, code=cid, eta=eta, return_states=conv is None, **kw)
    ref = orc.pteq_batch(cid, init, p, Nc, kw.pop("steps"), return_states=True, noise=0 if eta is None else 1, eta=eta or 0.0, **kw)
    assert np.array_equal(got["tops0"], ref["tops0"].astype(np.uint32))
    assert np.array_equal(got["samples"], ref["samples"].astype(np.uint32))
    assert np.array_equal(got["counts"], ref["counts"]) and got["counts"].shape == (N, 4)
    assert np.array_equal(got["steps_done"], ref["steps_done"].astype(np.uint32))
    if conv is None:
        assert np.array_equal(got["states"], ref["states"])
    else:
        assert np.array_equal(got["converged"], ref["converged"])


def test_pteq_biased_dropin(q, orc):
    rng = np.random.default_rng(2)
    code = q.xzzx_code(5)
    code.qubit_matrix = rand_states(rng, 1, 5, 0.15)[0]
    pct = q.PTEQ_biased(code, 0.15, eta=100, Nc=5, steps=300, conv_criteria=None, seed=12, replicas=1)
    ref = orc.pteq(q.XZZX, code.qubit_matrix, 0.15, Nc=5, steps=300, rng=orc.Rng.philox(12, 0), noise=1, eta=100)
    assert pct.shape == (4,) and np.array_equal(pct, ref["percent"])
    rot = q.RotSurCode(5)
    rot.qubit_matrix = rand_states(rng, 1, 5, 0.15)[0]
    pct = q.PTEQ(rot, 0.17, steps=300, conv_criteria=None, seed=13, replicas=1)          # decoders.PTEQ works on any code model
    ref = orc.pteq(q.ROTATED, rot.qubit_matrix, 0.17, Nc=5, steps=300, rng=orc.Rng.philox(13, 0))
    assert np.array_equal(pct, ref["percent"])
