"""GPU parity tests proper: the HIP path, called through the C-ABI, against the
golden fixtures and against the CPU oracle on the same seeded inputs.  Integer /
byte work: the bar is bit-exact."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def q():
    import qecmc
    assert qecmc.device_count() >= 1, "no MI355X visible: the product has no CPU fallback"
    return qecmc


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def rand_states(rng, N, L, p):
    m = np.zeros((N, 2, L, L), dtype=np.uint8)
    err = rng.random(m.shape) < p
    m[err] = rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)
    return m


# ------------------------------------------------------------------ F1: stencils vs the reference's vectors
@pytest.mark.parametrize("L", [3, 5, 9])
@pytest.mark.parametrize("rep", [0, 1])
def test_f1_stencils_on_device(q, L, rep):
    from qecmc import toric_model as tm
    g = np.load(os.path.join(GOLDEN, "f1_toric.npz"))
    t = f"L{L}_{rep}"
    m = g[f"{t}_m"]
    code = q.Toric_code(L)
    code.qubit_matrix = m.copy()
    assert code.count_errors() == int(g[f"{t}_count"])
    assert code.define_equivalence_class() == int(g[f"{t}_class"])
    code.syndrom()
    assert np.array_equal(code.defect_matrix, g[f"{t}_defects"])
    # all stabilizers in one batched call
    ops, rows, cols = np.meshgrid([1, 3], np.arange(L), np.arange(L), indexing="ij")
    n = ops.size
    new, dE = tm.apply_stabilizer(np.broadcast_to(m, (n,) + m.shape), rows.ravel(), cols.ravel(), ops.ravel())
    assert np.array_equal(new, g[f"{t}_stab_new"]) and np.array_equal(dE, g[f"{t}_stab_dE"])
    assert np.array_equal(code.qubit_matrix, m)          # apply_* never mutate the input
    a = g[f"{t}_log_arg"]
    new, dE = tm.apply_logical(np.broadcast_to(m, (len(a),) + m.shape), a[:, 0], a[:, 1], a[:, 2], a[:, 3])
    assert np.array_equal(new, g[f"{t}_log_new"]) and np.array_equal(dE, g[f"{t}_log_dE"])
    out = tm.to_class(np.broadcast_to(m, (16,) + m.shape), np.arange(16))
    assert np.array_equal(out, g[f"{t}_to_class"])
    assert np.array_equal(tm.eq_class(out), np.arange(16))
    # single-call forms used by the drop-in API
    one, d1 = code.apply_stabilizer(L - 1, 0, 1)
    assert np.array_equal(one, g[f"{t}_stab_new"][(L - 1) * L]) and d1 == int(g[f"{t}_stab_dE"][(L - 1) * L])


def test_stencil_argument_errors(q):
    code = q.Toric_code(3)
    with pytest.raises(q.QecmcError):
        code.apply_stabilizer(3, 0, 1)
    with pytest.raises(q.QecmcError):
        code.apply_stabilizer(0, 0, 2)
    with pytest.raises(q.QecmcError):
        code.to_class(16)


# ------------------------------------------------------------------ Chain.update_chain vs oracle (same Philox stream)
@pytest.mark.parametrize("L,p,p_logical,iters", [(3, 0.5, 0.0, 50), (5, 0.10, 0.0, 1000), (5, 0.10, 0.0, 10000), (9, 0.15, 0.0, 2000),   # (5, 0.1, 0, 10000): BASELINE config 1 at its stated length
                                                 (9, 0.75, 0.5, 1000), (5, 0.30, 0.5, 1000), (9, 0.20, 0.25, 800),
                                                 (15, 0.18, 0.0, 500), (4, 0.6, 1.0, 300)])
def test_chain_update_bit_exact(q, orc, L, p, p_logical, iters):
    rng = np.random.default_rng(L * 100 + iters)
    m = rand_states(rng, 1, L, 0.15)[0]
    seed, stream, slot, k0 = 0xDEADBEEFCAFE, 7, 3, 12345
    code = q.Toric_code(L)
    code.qubit_matrix = m.copy()
    ch = q.Chain(p, code, seed=seed, stream=stream)
    ch.p_logical, ch.slot, ch.proposals_done = p_logical, slot, k0
    ch.update_chain(iters)
    ref = orc.toric_chain_update(m, p, p_logical, iters, orc.Rng.philox(seed, stream), slot=slot, k0=k0)
    assert np.array_equal(ch.code.qubit_matrix, ref)
    assert np.array_equal(q.toric_model.syndrome(ref), q.toric_model.syndrome(m))


# ------------------------------------------------------------------ Ladder.step vs oracle
@pytest.mark.parametrize("L,p,Nc,iters,nstep", [(3, 0.3, 4, 5, 60), (5, 0.10, 5, 10, 80), (9, 0.15, 8, 10, 60),
                                                (5, 0.25, 3, 7, 70), (3, 0.05, 2, 10, 90), (7, 0.12, 16, 3, 40), (5, 0.12, 3, 1, 40),
                                                (9, 0.15, 8, 13, 20)])
def test_ladder_step_bit_exact(q, orc, L, p, Nc, iters, nstep):
    rng = np.random.default_rng(L + Nc)
    m = rand_states(rng, 1, L, 0.15)[0]
    seed, stream = 99, 5
    code = q.Toric_code(L)
    code.qubit_matrix = m.copy()
    ld = q.Ladder(p, code, Nc, 0.5, seed=seed, stream=stream)
    ref = orc.ToricLadder(m, p, Nc, 0.5)
    r = orc.Rng.philox(seed, stream)
    assert np.array_equal(ld.p_ladder, ref.p_ladder) and np.array_equal(ld.p_diff, ref.p_diff)
    done = 0
    for chunk in (1, 1, 3, nstep - 5):                   # resumable: any chunking gives the same trajectory
        ld.step(iters, nsteps=chunk)
        for _ in range(chunk):
            ref.step(iters, r)
        done += chunk
        got = np.stack([c.code.qubit_matrix for c in ld.chains])
        assert np.array_equal(got, ref.states), f"states differ after {done} steps"
        assert [c.flag for c in ld.chains] == ref.flags.tolist()
        assert ld.tops0 == ref.tops0


def test_interleaved_ladders_share_the_plan_cache_and_block_pool(q, orc):
    """The step entry points keep the tables of the last 8 parameter sets and recycle their small device blocks: 11 ladders of
    different shapes stepped in turn (so entries are evicted and rebuilt, blocks change hands) must each stay on its own trajectory."""
    rng = np.random.default_rng(77)
    shapes = [(3, 0.30, 4), (5, 0.10, 5), (5, 0.12, 5), (7, 0.15, 8), (3, 0.05, 2), (5, 0.20, 3), (9, 0.15, 8), (5, 0.10, 6), (7, 0.10, 3),
              (3, 0.20, 3), (5, 0.15, 5)]
    lads, refs, rngs = [], [], []
    for i, (L, p, Nc) in enumerate(shapes):
        m = rand_states(rng, 1, L, 0.15)[0]
        code = q.Toric_code(L)
        code.qubit_matrix = m.copy()
        lads.append(q.Ladder(p, code, Nc, 0.5, seed=1000 + i, stream=i))
        refs.append(orc.ToricLadder(m, p, Nc, 0.5))
        rngs.append(orc.Rng.philox(1000 + i, i))
    for rnd in range(4):
        for ld, ref, r in zip(lads, refs, rngs):
            ld.step(10, nsteps=1 + rnd)
            for _ in range(1 + rnd):
                ref.step(10, r)
    for ld, ref in zip(lads, refs):
        assert np.array_equal(np.stack([c.code.qubit_matrix for c in ld.chains]), ref.states)
        assert [c.flag for c in ld.chains] == ref.flags.tolist() and ld.tops0 == ref.tops0


def test_ladders_stepped_from_concurrent_threads(q, orc):
    """ctypes releases the GIL around a call: four threads stepping their own ladders (two of them with identical parameters,
    i.e. sharing one cached plan) through the pooled blocks must reproduce the oracle's trajectories."""
    import threading
    rng = np.random.default_rng(78)
    shapes = [(5, 0.10, 5, 7), (5, 0.10, 5, 7), (7, 0.15, 8, 8), (3, 0.30, 4, 9)]
    lads, inits = [], []
    for i, (L, p, Nc, seed) in enumerate(shapes):
        m = rand_states(rng, 1, L, 0.15)[0]
        code = q.Toric_code(L)
        code.qubit_matrix = m.copy()
        lads.append(q.Ladder(p, code, Nc, 0.5, seed=seed, stream=3))
        inits.append(m)
    errs = []

    def run(ld):
        try:
            for _ in range(60):
                ld.step(10)
        except Exception as e:           # noqa: BLE001 -- reported below, in the main thread
            errs.append(e)
    th = [threading.Thread(target=run, args=(ld,)) for ld in lads]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for (L, p, Nc, seed), m, ld in zip(shapes, inits, lads):
        ref, r = orc.ToricLadder(m, p, Nc, 0.5), orc.Rng.philox(seed, 3)
        for _ in range(60):
            ref.step(10, r)
        assert np.array_equal(np.stack([c.code.qubit_matrix for c in ld.chains]), ref.states)
        assert ld.tops0 == ref.tops0


# ------------------------------------------------------------------ PTEQ batch vs oracle
@pytest.mark.parametrize("L,p,Nc,N,steps,tops_burn", [
    (3, 0.10, 3, 1, 300, 2), (3, 0.10, 2, 63, 200, 0), (5, 0.10, 5, 65, 200, 1), (5, 0.10, 5, 130, 150, 2),
    (9, 0.15, 8, 96, 100, 0), (9, 0.15, 8, 64, 120, 2), (7, 0.12, 16, 20, 60, 0), (15, 0.18, 8, 70, 40, 0),
    (5, 0.2, 1, 10, 50, 0)])
def test_pteq_batch_bit_exact(q, orc, L, p, Nc, N, steps, tops_burn):
    rng = np.random.default_rng(N * 7 + L)
    init = rand_states(rng, N, L, p)
    got = q.pteq_batch(init, p, Nc=Nc, steps=steps, iters=10, tops_burn=tops_burn, seed=2020, first_syndrome=11,
                       return_states=True)
    ref = orc.toric_pteq_batch(init, p, Nc, steps, iters=10, tops_burn=tops_burn, seed=2020, first_syndrome=11,
                               return_states=True)
    assert np.array_equal(got["tops0"], ref["tops0"].astype(np.uint32))
    assert np.array_equal(got["samples"], ref["samples"].astype(np.uint32))
    assert np.array_equal(got["counts"], ref["counts"])
    assert np.array_equal(got["states"], ref["states"])
    assert np.array_equal(got["counts"].sum(axis=1), got["samples"])


def test_sharding_invariance(q):
    """Philox is keyed by the GLOBAL syndrome index: a batch split across calls (GPUs) gives the same answer."""
    rng = np.random.default_rng(3)
    init = rand_states(rng, 200, 5, 0.1)
    full = q.pteq_batch(init, 0.1, Nc=5, steps=100, tops_burn=0, seed=5)
    a = q.pteq_batch(init[:77], 0.1, Nc=5, steps=100, tops_burn=0, seed=5, first_syndrome=0)
    b = q.pteq_batch(init[77:], 0.1, Nc=5, steps=100, tops_burn=0, seed=5, first_syndrome=77)
    assert np.array_equal(full["counts"], np.concatenate([a["counts"], b["counts"]]))
    assert np.array_equal(full["tops0"], np.concatenate([a["tops0"], b["tops0"]]))


def test_pteq_dropin_signature(q, orc):
    rng = np.random.default_rng(1)
    code = q.Toric_code(5)
    code.qubit_matrix = rand_states(rng, 1, 5, 0.1)[0]
    pct = q.PTEQ(code, 0.1, Nc=5, steps=400, iters=10, tops_burn=1, conv_criteria=None, seed=77, replicas=1)
    ref = orc.toric_pteq(code.qubit_matrix, 0.1, Nc=5, steps=400, iters=10, tops_burn=1, rng=orc.Rng.philox(77, 0))
    assert pct.dtype == np.uint8 and pct.shape == (16,)
    assert np.array_equal(pct, ref["percent"])


def test_empty_batch(q):
    out = q.pteq_batch(np.zeros((0, 2, 5, 5), dtype=np.uint8), 0.1, Nc=5, steps=10)
    assert out["counts"].shape == (0, 16)


# ------------------------------------------------------------------ size-independent properties at BASELINE size
def test_full_size_properties(q, orc):
    """cfg 2 shape: 65 536 toric L=9 syndromes, Nc=8 (shortened run).  Checks that need no oracle at this size:
    the syndrome of every chain is conserved, counts add up to the sample count, and a random
    subset of syndromes is bit-identical to the oracle."""
    from qecmc import toric_model as tm
    rng = np.random.default_rng(2020)
    N, L, Nc, steps = 65536, 9, 8, 50
    init = rand_states(rng, N, L, 0.15)
    got = q.pteq_batch(init, 0.15, Nc=Nc, steps=steps, iters=10, tops_burn=0, seed=9, return_states=True)
    assert np.array_equal(got["counts"].sum(axis=1), got["samples"]) and np.all(got["samples"] == steps)
    syn0 = tm.syndrome(init)
    for c in range(Nc):
        assert np.array_equal(tm.syndrome(np.ascontiguousarray(got["states"][:, c])), syn0)
    pick = rng.choice(N, size=48, replace=False)
    for s in pick:
        ref = orc.toric_pteq_batch(init[s:s + 1], 0.15, Nc, steps, iters=10, tops_burn=0, seed=9, first_syndrome=int(s),
                                   return_states=True)
        assert np.array_equal(got["counts"][s], ref["counts"][0])
        assert np.array_equal(got["states"][s], ref["states"][0])


# ------------------------------------------------------------------ convergence criterion (decoders.py:74-105)
@pytest.mark.parametrize("L,p,Nc,N,steps,tops_burn,SEQ,TOPS,eps", [
    (3, 0.10, 3, 70, 3000, 1, 1, 4, 0.5), (3, 0.10, 3, 64, 6000, 2, 2, 10, 0.25), (5, 0.10, 5, 40, 4000, 2, 2, 3, 0.4),
    (5, 0.12, 4, 33, 6000, 1, 1, 4, 0.6), (3, 0.3, 2, 10, 500, 0, 0, 1, 1.0)])
def test_pteq_error_based_bit_exact(q, orc, L, p, Nc, N, steps, tops_burn, SEQ, TOPS, eps):
    rng = np.random.default_rng(L * 31 + N)
    init = rand_states(rng, N, L, p)
    kw = dict(steps=steps, iters=10, tops_burn=tops_burn, seed=77, first_syndrome=3, conv_criteria="error_based",
              SEQ=SEQ, TOPS=TOPS, eps=eps)
    got = q.pteq_batch(init, p, Nc=Nc, **kw)
    ref = orc.toric_pteq_batch(init, p, Nc, kw.pop("steps"), **kw)
    assert np.array_equal(got["converged"], ref["converged"])
    assert np.array_equal(got["steps_done"], ref["steps_done"].astype(np.uint32))
    assert np.array_equal(got["samples"], ref["samples"].astype(np.uint32))
    assert np.array_equal(got["tops0"], ref["tops0"].astype(np.uint32))
    assert np.array_equal(got["counts"], ref["counts"])
    assert 0 < got["converged"].sum()                       # the early-exit path really ran


def test_pteq_dropin_default_criterion(q, orc):
    """PTEQ(code, p) with the reference's defaults (error_based, SEQ=2, TOPS=10, eps=0.1, steps=5e7)."""
    rng = np.random.default_rng(8)
    code = q.Toric_code(3)
    code.qubit_matrix = rand_states(rng, 1, 3, 0.1)[0]
    pct = q.PTEQ(code, 0.1, seed=5, replicas=1)
    ref = orc.toric_pteq_batch(code.qubit_matrix[None], 0.1, 3, 1 << 22, seed=5, conv_criteria="error_based")
    assert ref["converged"][0]
    exp = (np.divide(ref["counts"][0], ref["samples"][0]) * 100).astype(np.uint8)
    assert np.array_equal(pct, exp) and 96 <= int(pct.sum()) <= 100
    # one launch, no step run twice: the ladder stopped inside the first horizon
    from qecmc import decoders
    assert decoders.LAST_RUN["launches"] == 1 and decoders.LAST_RUN["replayed_steps"] == 0 and decoders.LAST_RUN["steps_done"] == int(ref["steps_done"][0])
    # ... and a ladder that outlasts a (shrunk) first horizon gives the same answer through the growing horizons, replaying at most 1/15
    old = decoders.PTEQ_FIRST_HORIZON
    try:
        decoders.PTEQ_FIRST_HORIZON = max(16, int(ref["steps_done"][0]) // 40)
        pct2 = q.PTEQ(code, 0.1, seed=5, replicas=1)
    finally:
        decoders.PTEQ_FIRST_HORIZON = old
    assert np.array_equal(pct2, exp) and decoders.LAST_RUN["launches"] >= 2
    assert decoders.LAST_RUN["replayed_steps"] < decoders.LAST_RUN["horizon"] / 14


# ------------------------------------------------------------------ scan = sweep (systematic generator sweep)
@pytest.mark.parametrize("code,L,p,Nc,N,steps,tops_burn,iters", [
    ("toric", 3, 0.10, 3, 70, 200, 1, 10), ("toric", 5, 0.10, 5, 65, 150, 0, 10), ("toric", 9, 0.15, 8, 130, 100, 0, 10),
    ("toric", 9, 0.15, 8, 64, 60, 0, 7), ("toric", 5, 0.2, 1, 20, 80, 0, 10), ("rotated", 5, 0.17, 5, 40, 150, 0, 10),
    ("xzzx", 9, 0.15, 8, 64, 80, 0, 10), ("toric", 15, 0.18, 16, 30, 20, 0, 13)])
def test_sweep_scan_bit_exact(q, orc, code, L, p, Nc, N, steps, tops_burn, iters):
    """scan=sweep is not the reference's chain; the oracle restates the same sweep rule and the GPU must
    reproduce it bit for bit (its physics is checked against exact enumeration in test_gpu_stats.py)."""
    cid = {"toric": q.TORIC, "xzzx": q.XZZX, "rotated": q.ROTATED}[code]
    rng = np.random.default_rng(N + L)
    if code == "toric":
        init = rand_states(rng, N, L, p)
    else:
        init = np.zeros((N, L, L), dtype=np.uint8)
        err = rng.random(init.shape) < p
        init[err] = rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)
    got = q.pteq_batch(init, p, Nc=Nc, steps=steps, iters=iters, tops_burn=tops_burn, seed=99, first_syndrome=2, code=cid,
                       scan="sweep", return_states=True)
    ref = orc.pteq_batch(cid, init, p, Nc, steps, iters=iters, tops_burn=tops_burn, seed=99, first_syndrome=2, scan=1,
                         return_states=True)
    assert np.array_equal(got["tops0"], ref["tops0"].astype(np.uint32))
    assert np.array_equal(got["samples"], ref["samples"].astype(np.uint32))
    assert np.array_equal(got["counts"], ref["counts"])
    assert np.array_equal(got["states"], ref["states"])


@pytest.mark.parametrize("L", [2, 3, 4, 6, 8, 10, 11, 13, 14, 16])
def test_lattice_size_sweep(q, orc, L):
    """Every lattice size the fast top-chain path takes (L <= 16): the logical frame is flushed as a stream of 2L-bit rows
    into 32-bit words, so each L has its own pattern of word boundaries (L = 16: a row is exactly a word); ragged batch,
    several ladder lengths, iters a multiple of 4 or not."""
    rng = np.random.default_rng(100 + L)
    N = 70
    init = (rng.integers(1, 4, size=(N, 2, L, L)) * (rng.random((N, 2, L, L)) < 0.1)).astype(np.uint8)
    for Nc, iters in ((2, 10), (5, 7), (8, 10), (8, 12)):
        got = q.pteq_batch(init, 0.12, Nc=Nc, steps=40, iters=iters, tops_burn=0, seed=5 + L, return_states=True)
        ref = orc.pteq_batch(orc.TORIC, init, 0.12, Nc, 40, iters=iters, tops_burn=0, seed=5 + L, return_states=True)
        assert np.array_equal(got["states"], ref["states"]) and np.array_equal(got["counts"], ref["counts"])
        assert np.array_equal(got["tops0"], ref["tops0"].astype(np.uint32))
