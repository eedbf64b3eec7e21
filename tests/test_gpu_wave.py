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
    ("toric", 9, 8, 6, 200, 10, 0.15, 0.5, 1, 64), ("toric", 11, 8, 3, 60, 10, 0.18, 0.5, 1, 0),      # W = 11 -> 12 words / 16 words exactly
    ("toric", 4, 4, 4, 150, 7, 0.12, 0.5, 1, 0), ("toric", 9, 16, 2, 50, 3, 0.15, 0.25, 1, 0),         # odd iters: blocks of 8 cut anywhere; 16 rungs
    ("toric", 7, 2, 4, 200, 10, 0.10, 0.5, 1, 0), ("toric", 5, 5, 4, 120, 1, 0.02, 1.0, 3, 192),       # p = 0.02: thresholds below 2^-16
    ("toric", 5, 5, 3, 300, 10, 0.10, 0.0, 1, 0),                                                     # no logical moves (PTDC's ladders)
    ("toric", 10, 4, 2, 40, 25, 0.15, 0.5, 1, 0), ("toric", 9, 9, 3, 80, 10, 0.15, 0.5, 1, 0),        # 13 of 16 words; 9 rungs: 1024-thread groups
    ("xzzx", 9, 8, 5, 150, 10, 0.15, 0.5, 1, 0), ("rotated", 7, 7, 4, 150, 10, 0.17, 0.5, 1, 0), ("rotated", 15, 8, 2, 30, 10, 0.17, 0.5, 1, 0),
    ("planar", 5, 5, 4, 150, 10, 0.12, 0.5, 1, 0), ("xzzx", 3, 2, 7, 300, 5, 0.2, 0.5, 2, 64), ("planar", 9, 8, 3, 100, 10, 0.12, 0.5, 1, 0),
    # 17 .. 32 state words per rung: the 32-word kernels (6 waves per SIMD, the state through the exchange buffer in two halves)
    ("toric", 15, 8, 3, 40, 10, 0.18, 0.5, 1, 0), ("toric", 16, 6, 2, 30, 7, 0.15, 0.5, 1, 64), ("toric", 12, 8, 70, 30, 10, 0.15, 0.5, 1, 0),   # W = 29, 32, 18
    ("rotated", 21, 8, 3, 40, 10, 0.17, 0.5, 1, 0), ("xzzx", 17, 5, 2, 40, 10, 0.15, 0.5, 2, 0), ("rotated", 19, 2, 3, 50, 3, 0.1, 1.0, 1, 128)]


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


@pytest.mark.parametrize("name,L,Nc,N,steps,iters,grid,R", [("toric", 3, 3, 300, 2500, 10, 1, 1), ("toric", 5, 5, 400, 3000, 10, 2, 1), ("toric", 5, 4, 200, 2000, 7, 3, 1),
                                                         ("rotated", 5, 4, 150, 2500, 10, 1, 1), ("xzzx", 5, 5, 100, 2000, 10, 1, 2), ("planar", 4, 4, 130, 1500, 5, 2, 1),
                                                         ("toric", 9, 8, 140, 1200, 10, 1, 1)])
def test_wave_scan_work_queue_bit_exact(q, orc, name, L, Nc, N, steps, iters, grid, R):
    """The criterion runs on a persistent grid (forced down to 1-3 workgroups here, so that every lane runs several ladders): a lane whose
    ladder has ended takes the next one of its workgroup's share of the batch, in lane order among the lanes that end together.  The
    oracle restates the rule (orc_pteq_wave_queue); class counts, samples, tops0, the stopping step and the flag of every ladder are
    its -- ladders stopped by the criterion and ladders that reach the horizon of `steps` of their own steps alike."""
    rng = np.random.default_rng(L + 17 * Nc + N)
    code, ocode = _codes(q, orc, name)
    init = _init(rng, name, N, L, 0.1)
    kw = dict(iters=iters, tops_burn=2, seed=9, first_syndrome=128, SEQ=2, TOPS=4, eps=0.3)
    got = q.pteq_batch(init, 0.1, Nc=Nc, code=code, scan="wave", replicas=R, steps=steps, conv_criteria="error_based", flags=q.dev_flags(queue_grid=grid), **kw)
    M = N * R
    g_eff = max(1, min(grid, (M + 63) // 64))
    ref = orc.pteq_wave_queue(ocode, np.repeat(init, R, axis=0), 0.1, Nc, steps, g_eff, **kw)
    ncls = ref["counts"].shape[1]
    if name == "toric" and L <= 5 and iters == 10:
        assert ref["converged"].any() and not ref["converged"].all()                  # both ways of ending occur
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
    with pytest.raises(q.QecmcError):
        q.pteq_batch(np.zeros((2, 2, 15, 15), np.uint8), 0.18, Nc=8, steps=10, scan="wave", conv_criteria="error_based")   # 29 words per rung: fixed-length runs only
    with pytest.raises(q.QecmcError):
        q.pteq_batch(np.zeros((2, 2, 15, 15), np.uint8), 0.18, Nc=9, steps=10, scan="wave")   # ... of at most 8 rungs
    with pytest.raises(q.QecmcError):
        q.pteq_batch(np.zeros((2, 2, 17, 17), np.uint8), 0.18, Nc=4, steps=10, scan="wave")   # 37 words per rung
    with pytest.raises(q.QecmcError):                                                  # the criterion runs reuse lanes: no final states
        q.pteq_batch(init, 0.1, Nc=5, steps=100, scan="wave", conv_criteria="error_based", return_states=True)


@pytest.mark.parametrize("kind", ["depolarizing", "wave", "biased", "alpha"])
def test_launch_dev_refuses_an_undersized_workspace(q, kind):
    """ABI 4: qecmc_pteq_launch_dev takes the size of the criterion runs' log next to its pointer and refuses a buffer
    smaller than the launch needs -- QECMC_ERR_INVALID before anything is enqueued -- for the depolarizing, biased and alpha criterion
    plans alike; the size comes from qecmc_plan_workspace_bytes, the one place the formula lives."""
    import ctypes as C
    import torch
    from qecmc import _lib as L_
    N, L, Nc, steps = 200, 5, 5, 500
    kw = dict(L=L, Nc=Nc, p=0.1, p_logical=0.5, iters=10, steps=steps, tops_burn=0, TOPS=4, SEQ=2, eps=0.3, seed=1, conv_mode=L_.CONV_ERROR_BASED)
    if kind == "depolarizing":
        pr, nq, ncls = L_.make_params(code=L_.TORIC, **kw), 50, 16
    elif kind == "wave":
        pr, nq, ncls = L_.make_params(code=L_.TORIC, scan=L_.SCAN_WAVE, **kw), 50, 16
    elif kind == "biased":
        pr, nq, ncls = L_.make_params(code=L_.XZZX, noise=L_.NOISE_BIASED, eta=10.0, **kw), 25, 4
    else:
        pr, nq, ncls = L_.make_params(code=L_.XZZX, noise=L_.NOISE_ALPHA, alpha=2.0, **kw), 25, 4
    plan = C.c_void_p()
    L_.check(L_.lib().qecmc_plan_create(pr, C.byref(plan)))
    try:
        need = C.c_uint64()
        L_.check(L_.lib().qecmc_plan_workspace_bytes(plan, N, 0, C.byref(need)))
        assert need.value == (4 if kind == "alpha" else 2) * 256 * steps          # one column per ladder, whole groups of 64
        dev = torch.device("cuda", 0)
        init = torch.zeros(N * nq, dtype=torch.uint8, device=dev)
        counts = torch.zeros(N * ncls, dtype=torch.int32, device=dev)
        samples = torch.zeros(N, dtype=torch.int32, device=dev)
        ws = torch.zeros(need.value, dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream()
        args = (plan, init.data_ptr(), N, 0, counts.data_ptr(), samples.data_ptr(), None, None, None, None)
        # one ladder per lane here (the persistent grid is larger than the batch), so the launch needs all of it
        rc = L_.lib().qecmc_pteq_launch_dev(*args, ws.data_ptr(), need.value - 1, C.c_void_p(stream.cuda_stream))
        assert rc == -1 and b"workspace" in L_.lib().qecmc_last_error()
        rc = L_.lib().qecmc_pteq_launch_dev(*args, None, 0, C.c_void_p(stream.cuda_stream))
        assert rc == -1
        L_.check(L_.lib().qecmc_pteq_launch_dev(*args, ws.data_ptr(), need.value, C.c_void_p(stream.cuda_stream)))
        torch.cuda.synchronize()
        assert int(samples.sum().item()) > 0
    finally:
        L_.lib().qecmc_plan_destroy(plan)


# ---- scan = wave under the alpha noise model (src/mcmc_alpha.py; what generate_data.py:142-150 routes biased noise to) --------------------
ALPHA_CASES = [  # name, L, Nc, N, steps, iters, pz_tilde, alpha, replicas
    ("xzzx", 3, 3, 5, 300, 10, 0.20, 2.0, 1), ("xzzx", 5, 5, 70, 200, 10, 0.175, 4.04, 1), ("rotated", 5, 4, 6, 200, 7, 0.15, 3.0, 1),
    ("xzzx", 7, 7, 4, 150, 10, 0.12, 4.04, 1), ("rotated", 7, 8, 3, 100, 10, 0.2, 1.3, 2), ("xzzx", 9, 8, 3, 80, 10, 0.15, 2.5, 1),
    ("xzzx", 11, 5, 2, 40, 25, 0.1, 6.0, 1), ("xzzx", 5, 2, 4, 200, 1, 0.3, 1.0, 1), ("rotated", 9, 6, 3, 60, 3, 0.25, 2.0, 1),
    ("xzzx", 9, 9, 3, 80, 10, 0.175, 4.04, 1), ("rotated", 7, 16, 2, 40, 10, 0.2, 2.0, 1), ("xzzx", 5, 12, 3, 60, 7, 0.15, 3.0, 1)]      # Nc = L = 9 (the reference's default); 16 and 12 rungs


@pytest.mark.parametrize("name,L,Nc,N,steps,iters,pzt,alpha,R", ALPHA_CASES)
def test_wave_scan_alpha_bit_exact(q, orc, name, L, Nc, N, steps, iters, pzt, alpha, R):
    """Ladder_alpha (mcmc_alpha.py:75-137) with the generator pick shared by a wavefront: p_b frozen at the step's start (Q3), slot-bound n_eff
    attributes in the swap test (Q4), the top rung at pz_tilde = 1 -- class counts, tops0 and every rung's final configuration equal the oracle's."""
    rng = np.random.default_rng(L * 5 + Nc + N)
    code, ocode = _codes(q, orc, name)
    init = _init(rng, name, N, L, 0.12)
    kw = dict(steps=steps, iters=iters, tops_burn=0, seed=31, first_syndrome=64)
    got = q.pteq_batch(init, pzt, Nc=Nc, code=code, alpha=alpha, scan="wave", return_states=True, replicas=R, **kw)
    ref = orc.pteq_batch(ocode, np.repeat(init, R, axis=0), pzt, Nc, kw.pop("steps"), return_states=True, noise=orc.ALPHA, alpha=alpha, det_pow=1, scan=3, **kw)
    assert np.array_equal(got["counts"], ref["counts"].reshape(N, R, 4).sum(axis=1))
    assert np.array_equal(got["tops0"], ref["tops0"].reshape(N, R).sum(axis=1).astype(np.uint32))
    assert np.array_equal(got["samples"], ref["samples"].reshape(N, R).sum(axis=1).astype(np.uint32))
    if R == 1:
        assert np.array_equal(got["states"], ref["states"])
    assert got["counts"].sum() > 0 and not np.array_equal(got["states"][:, 0], init)


@pytest.mark.parametrize("name,L,Nc,N,steps,iters,grid,pzt,alpha", [("xzzx", 3, 3, 200, 2500, 10, 1, 0.2, 2.0), ("xzzx", 5, 5, 300, 3000, 10, 2, 0.175, 4.04),
                                                                  ("rotated", 5, 4, 150, 2500, 7, 1, 0.15, 3.0), ("xzzx", 7, 7, 100, 1500, 10, 1, 0.15, 4.04),
                                                                  ("xzzx", 5, 9, 150, 2500, 10, 1, 0.175, 4.04)])
def test_wave_scan_alpha_work_queue_bit_exact(q, orc, name, L, Nc, N, steps, iters, grid, pzt, alpha):
    """PTEQ_alpha's default route (decoders_biasednoise.py:175-238: error_based on chains[0].n_eff) on the deterministic work queue"""
    rng = np.random.default_rng(L + 19 * Nc + N)
    code, ocode = _codes(q, orc, name)
    init = _init(rng, name, N, L, 0.1)
    kw = dict(iters=iters, tops_burn=2, seed=13, first_syndrome=0, SEQ=2, TOPS=4, eps=0.3)
    got = q.pteq_batch(init, pzt, Nc=Nc, code=code, alpha=alpha, scan="wave", steps=steps, conv_criteria="error_based", flags=q.dev_flags(queue_grid=grid), **kw)
    g_eff = max(1, min(grid, (N + 63) // 64))
    ref = orc.pteq_wave_queue(ocode, init, pzt, Nc, steps, g_eff, noise=orc.ALPHA, alpha=alpha, det_pow=1, **kw)
    for k in ("counts", "samples", "tops0", "steps_done"):
        assert np.array_equal(got[k], ref[k].astype(got[k].dtype)), k
    assert np.array_equal(got["converged"], ref["converged"])
    assert ref["converged"].any()


@pytest.mark.parametrize("L,N,steps", [(9, 65536, 50), (15, 131072, 12)])
def test_wave_scan_at_the_full_size_of_baseline_configs_2_and_3(q, orc, L, N, steps):
    """The bench lines' own shapes on their own kernels (shortened runs): 65 536 toric L = 9 syndromes (12 words, 8 waves per SIMD) and 131 072 toric L = 15
    (32 words, the exchange in two halves), Nc = 8.  Size-independent properties -- every chain keeps its syndrome, the class counts add up to the samples --,
    sharded equals whole on a slice, and whole wavefronts of the batch bit-identical to the oracle (the pick group is the global index >> 6)."""
    from qecmc import toric_model as tm
    rng = np.random.default_rng(2024 + L)
    Nc = 8
    init = _rand(rng, (N, 2, L, L), 0.15)
    got = q.pteq_batch(init, 0.15, Nc=Nc, steps=steps, iters=10, tops_burn=0, seed=9, return_states=True, scan="wave")
    assert np.array_equal(got["counts"].sum(axis=1), got["samples"]) and np.all(got["samples"] == steps)
    syn0 = tm.syndrome(init)
    for c in (0, Nc - 1):
        assert np.array_equal(tm.syndrome(np.ascontiguousarray(got["states"][:, c])), syn0)
    for g in rng.choice(N // 64, size=2, replace=False):          # two wavefronts of the batch, anywhere in it
        lo = int(g) * 64
        part = q.pteq_batch(init[lo:lo + 64], 0.15, Nc=Nc, steps=steps, iters=10, tops_burn=0, seed=9, return_states=True, scan="wave", first_syndrome=lo)
        assert np.array_equal(part["counts"], got["counts"][lo:lo + 64]) and np.array_equal(part["states"], got["states"][lo:lo + 64])
        for s in (lo, lo + 37):
            ref = orc.pteq_batch(orc.TORIC, init[s:s + 1], 0.15, Nc, steps, iters=10, tops_burn=0, seed=9, first_syndrome=s, return_states=True, scan=3)
            assert np.array_equal(got["counts"][s], ref["counts"][0]) and np.array_equal(got["states"][s], ref["states"][0])
