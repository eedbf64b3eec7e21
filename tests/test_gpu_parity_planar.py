"""GPU parity for the planar surface code (src/planar_model.py; SURVEY row f4's code model): device stencils against the
reference's vectors (f_planar.npz), chains (incl. update_chain_fast) / ladders / PTEQ bit for bit against the oracle."""
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
    m = np.zeros((N, 2, L, L), dtype=np.uint8)
    err = rng.random(m.shape) < p
    m[err] = rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)
    m[:, 1, -1, :] = 0
    m[:, 1, :, -1] = 0
    return m


def _kats():
    return [str(k) for k in np.load(os.path.join(GOLDEN, "f_planar.npz"))["kats"]]


@pytest.mark.parametrize("t", _kats())
def test_planar_stencils_on_device(q, t):
    g = np.load(os.path.join(GOLDEN, "f_planar.npz"))
    m = g[f"{t}_m"]
    L = m.shape[-1]
    code = q.Planar_code(L)
    code.qubit_matrix = m.copy()
    assert code.count_errors() == int(g[f"{t}_count"]) and code.define_equivalence_class() == int(g[f"{t}_class"])
    code.syndrom()
    assert np.array_equal(code.vertex_defects, g[f"{t}_vertex"].astype(bool))
    assert np.array_equal(code.plaquette_defects, g[f"{t}_plaquette"].astype(bool))
    from qecmc import planar_model as pm
    args = g[f"{t}_stab_arg"]
    new, dE = pm.apply_stabilizer(np.broadcast_to(m, (len(args),) + m.shape), args[:, 0], args[:, 1], args[:, 2])
    assert np.array_equal(new, g[f"{t}_stab_new"]) and np.array_equal(dE, g[f"{t}_stab_dE"])
    args = g[f"{t}_log_arg"]
    new, dE = pm.apply_logical(np.broadcast_to(m, (len(args),) + m.shape), args[:, 0], args[:, 1], args[:, 2])
    assert np.array_equal(new, g[f"{t}_log_new"]) and np.array_equal(dE, g[f"{t}_log_dE"])
    assert np.array_equal(pm.eq_class(new), g[f"{t}_log_class"])
    assert np.array_equal(code.qubit_matrix, m)                      # never mutated
    with pytest.raises(q.QecmcError):
        code.apply_stabilizer(L - 1, 0, 1)                           # X-type generators live on rows [0, L-1)
    with pytest.raises(q.QecmcError):
        code.apply_stabilizer(0, L - 1, 3)


@pytest.mark.parametrize("L,pxyz,iters", [(3, (0.05, 0.03, 0.04), 200), (5, (0.02, 0.10, 0.01), 1000), (7, (0.10, 0.10, 0.10), 800), (5, (0.30, 0.001, 0.05), 1500)])
def test_chain_xyz_update_bit_exact(q, orc, L, pxyz, iters):
    """src/mcmc.py:106-114 (Chain_xyz, the chain decoders.py:352,442 run STDC_general_noise on): planar code, general (p_x, p_y, p_z)"""
    code = q.Planar_code(L)
    code.qubit_matrix = rand_states(np.random.default_rng(L * 7 + iters), 1, L, 0.2)[0]
    m0 = code.qubit_matrix.copy()
    seed, stream = 0xABCDEF12345, 11
    ch = q.mcmc.Chain_xyz(np.array(pxyz), code, seed=seed, stream=stream)
    assert ch.qubit_errors.shape == (3,)
    for part in (iters // 3, iters - iters // 3):          # two calls continue one proposal stream (k0 carries)
        ch.update_chain_fast(part)
    ref = orc.chain_update(orc.PLANAR, m0, 0.0, 0.0, iters, orc.Rng.philox(seed, stream), pxyz=pxyz)
    assert np.array_equal(ch.code.qubit_matrix, ref) and (L == 3 or not np.array_equal(ref, m0))
    assert np.array_equal(ch.qubit_errors, [np.sum(ref == k) for k in (1, 2, 3)])


@pytest.mark.parametrize("L,p,p_logical,iters,fast", [(5, 0.15, 0.0, 800, False), (7, 0.3, 0.5, 600, False), (4, 0.2, 0.0, 500, True),
                                                       (12, 0.1, 0.0, 400, True), (9, 0.75, 0.5, 500, False)])
def test_chain_bit_exact(q, orc, L, p, p_logical, iters, fast):
    rng = np.random.default_rng(L * 31 + iters)
    m = rand_states(rng, 1, L, 0.12)[0]
    seed, stream, slot, k0 = 0xFEED5, 7, 1, 123
    code = q.Planar_code(L)
    code.qubit_matrix = m.copy()
    ch = q.Chain(p, code, seed=seed, stream=stream)
    ch.p_logical, ch.slot, ch.proposals_done = p_logical, slot, k0
    if fast:                                                         # STDC's inner call (decoders.py:250): 5 proposals at a time
        for _ in range(iters // 5):
            ch.update_chain_fast(5)
    else:
        ch.update_chain(iters)
    ref = orc.chain_update(orc.PLANAR, m, p, 0.0 if fast else p_logical, iters, orc.Rng.philox(seed, stream), slot=slot, k0=k0)
    assert np.array_equal(ch.code.qubit_matrix, ref)
    a, b = orc.planar_syndrome(ref), orc.planar_syndrome(m)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("L,p,Nc,iters,nstep", [(3, 0.3, 3, 5, 60), (5, 0.15, 5, 10, 60), (7, 0.12, 8, 10, 40), (9, 0.15, 16, 3, 30),
                                                (4, 0.2, 1, 10, 30), (12, 0.12, 8, 7, 20)])
def test_ladder_bit_exact(q, orc, L, p, Nc, iters, nstep):
    rng = np.random.default_rng(L + Nc)
    m = rand_states(rng, 1, L, 0.12)[0]
    seed, stream = 13579, 2
    code = q.Planar_code(L)
    code.qubit_matrix = m.copy()
    ld = q.Ladder(p, code, Nc, 0.5, seed=seed, stream=stream)
    ref = orc.Ladder(orc.PLANAR, m, p, Nc, 0.5)
    r = orc.Rng.philox(seed, stream)
    done = 0
    for chunk in (1, 2, nstep - 3):
        ld.step(iters, nsteps=chunk)
        for _ in range(chunk):
            ref.step(iters, r)
        done += chunk
        got = np.stack([c.code.qubit_matrix for c in ld.chains])
        assert np.array_equal(got, ref.states), f"states differ after {done} steps"
        assert [c.flag for c in ld.chains] == ref.flags.tolist() and ld.tops0 == ref.tops0


@pytest.mark.parametrize("L,p,Nc,N,steps,tops_burn,conv,scan", [
    (5, 0.15, 5, 70, 200, 1, None, "random"), (9, 0.15, 8, 65, 100, 0, None, "random"), (12, 0.12, 8, 40, 40, 0, None, "random"),
    (3, 0.17, 3, 50, 4000, 1, "error_based", "random"), (7, 0.15, 6, 64, 100, 0, None, "sweep")])
def test_pteq_batch_bit_exact(q, orc, L, p, Nc, N, steps, tops_burn, conv, scan):
    rng = np.random.default_rng(N * 3 + L)
    init = rand_states(rng, N, L, p)
    kw = dict(steps=steps, iters=10, tops_burn=tops_burn, seed=2468, first_syndrome=9, conv_criteria=conv)
    if conv:
        kw.update(SEQ=1, TOPS=4, eps=0.6)
    got = q.pteq_batch(init, p, Nc=Nc, code=q.PLANAR, scan=scan, return_states=conv is None, **kw)
    ref = orc.pteq_batch(orc.PLANAR, init, p, Nc, kw.pop("steps"), return_states=True, scan=1 if scan == "sweep" else 0, **kw)
    assert np.array_equal(got["tops0"], ref["tops0"].astype(np.uint32))
    assert np.array_equal(got["samples"], ref["samples"].astype(np.uint32))
    assert np.array_equal(got["counts"], ref["counts"]) and got["counts"].shape == (N, 4)
    if conv is None:
        assert np.array_equal(got["states"], ref["states"])
    else:
        assert np.array_equal(got["converged"], ref["converged"]) and np.array_equal(got["steps_done"], ref["steps_done"].astype(np.uint32))


def test_planar_pteq_dropin_and_exact_classes(q, orc):
    """decoders.PTEQ on a Planar_code; and the class histogram against exact enumeration of the stabilizer group at L = 3
    (2^12 generator subsets x 4 classes)."""
    rng = np.random.default_rng(6)
    code = q.Planar_code(5)
    code.qubit_matrix = rand_states(rng, 1, 5, 0.12)[0]
    pct = q.PTEQ(code, 0.12, steps=300, conv_criteria=None, seed=5, replicas=1)
    ref = orc.pteq(orc.PLANAR, code.qubit_matrix, 0.12, Nc=5, steps=300, rng=orc.Rng.philox(5, 0))
    assert pct.shape == (4,) and np.array_equal(pct, ref["percent"])
    L, p = 3, 0.12
    m = rand_states(rng, 1, L, 0.15)[0]
    f = (p / 3) / (1 - p)
    G = orc.surf_ngen(orc.PLANAR, L)
    gens = []
    for g in range(G):
        r, c, o = orc.surf_gen_rco(orc.PLANAR, L, g)
        gens.append(orc.surf_apply_stabilizer(orc.PLANAR, np.zeros_like(m), r, c, o)[0])
    Z = np.zeros(4)
    for cls_op in range(4):
        base = orc.surf_apply_logical(orc.PLANAR, m, cls_op, 0, 0)[0]
        cls = orc.surf_eq_class(orc.PLANAR, base)
        for mask in range(1 << G):
            s = base.copy()
            for g in range(G):
                if (mask >> g) & 1:
                    s ^= gens[g]
            Z[cls] += f ** np.count_nonzero(s)
    P = Z / Z.sum()
    R = 2048
    res = q.pteq_batch(np.broadcast_to(m, (R,) + m.shape).copy(), p, Nc=3, steps=3000, tops_burn=0, code=q.PLANAR, seed=77)
    frac = res["counts"] / res["samples"][:, None].astype(np.float64)
    mean, sem = frac.mean(axis=0), frac.std(axis=0, ddof=1) / np.sqrt(R)
    assert np.all(np.abs(mean - P) <= 5 * sem + 0.01), (mean, P, sem)


@pytest.mark.parametrize("name", ["planar", "xzzx", "rot"])
def test_generate_random_error_draw_order(q, name):
    """generate_random_error consumes one `random.random()` per cell in C order and maps it to Z / X / Y by the open
    intervals of planar_model.py:18-36 / xzzx_model.py:16-30 / rotated_surface_model.py:25-38."""
    import random
    L, (px, py, pz) = 5, (0.07, 0.05, 0.11)
    code = {"planar": q.Planar_code, "xzzx": q.xzzx_code, "rot": q.RotSurCode}[name](L)
    random.seed(123)
    code.generate_random_error(px, py, pz)
    random.seed(123)
    shape = (2, L, L) if name == "planar" else (L, L)
    want = np.zeros(shape, dtype=np.uint8)
    for idx in np.ndindex(*shape):
        r = random.random()
        want[idx] = 3 if r < pz else 1 if pz < r < pz + px else 2 if pz + px < r < pz + px + py else 0
    if name == "planar":
        want[1, -1, :] = 0
        want[1, :, -1] = 0
    assert np.array_equal(code.qubit_matrix, want) and code.qubit_matrix.dtype == np.uint8


@pytest.mark.parametrize("name,L", [("xzzx", 3), ("xzzx", 9), ("xzzx", 15), ("xzzx", 17), ("xzzx", 21), ("xzzx", 31),
                                    ("rotated", 5), ("rotated", 9), ("rotated", 15), ("rotated", 17), ("rotated", 21), ("rotated", 27),
                                    ("planar", 3), ("planar", 4), ("planar", 9), ("planar", 16), ("planar", 17), ("planar", 20)])
def test_plaquette_code_size_sweep(q, orc, name, L):
    """The top chain of the plaquette codes collects its logical operators in a frame and flushes it as a stream of 2L-bit
    rows: one-word rows up to L = 16, two-word rows up to L = 32 (rotated L = 21 is BASELINE config 5's shape)."""
    cg, co = {"xzzx": (q.XZZX, orc.XZZX), "rotated": (q.ROTATED, orc.ROTATED), "planar": (q.PLANAR, orc.PLANAR)}[name]
    rng = np.random.default_rng(7 * L + len(name))
    N = 40
    shape = (N, 2, L, L) if name == "planar" else (N, L, L)
    init = (rng.integers(1, 4, size=shape) * (rng.random(shape) < 0.1)).astype(np.uint8)
    if name == "planar":
        init[:, 1, -1, :] = 0
        init[:, 1, :, -1] = 0
    Nc = 4 if L > 20 else 5
    got = q.pteq_batch(init, 0.12, Nc=Nc, steps=30, iters=10, tops_burn=0, seed=5 + L, code=cg, return_states=True)
    ref = orc.pteq_batch(co, init, 0.12, Nc, 30, iters=10, tops_burn=0, seed=5 + L, return_states=True)
    assert np.array_equal(got["states"], ref["states"]) and np.array_equal(got["counts"], ref["counts"])
    assert np.array_equal(got["tops0"], ref["tops0"].astype(np.uint32))
