"""Size-independent properties at the per-GPU sizes of BASELINE configs 3, 4 and 5 (config 2's is tests/test_gpu_parity.py::test_full_size_properties):
what can be checked without the oracle at 10^4-10^5 ladders -- every rung keeps its syndrome, class counts add up to the sample
count, a batch run as two shards with their global indices gives the rows of the whole batch (the multi-GPU partitioning),
a run continued from its own output is the longer run -- plus a random subset of ladders compared with the oracle bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def q():
    import qecmc
    return qecmc


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


CASES = [("toric", 15, 0.18, 8, 131072, None),        # configs[2]: 1M syndromes over 8 GPUs
         ("xzzx", 9, 0.15, 8, 65536, 100.0),          # configs[3]: the biased rule
         ("rotated", 21, 0.17, 8, 32768, None)]       # configs[4]


@pytest.mark.parametrize("name,L,p,Nc,N,eta", CASES)
def test_full_size_properties(q, orc, name, L, p, Nc, N, eta):
    from qecmc import _surf, toric_model as tm
    rng = np.random.default_rng(L * 1000 + N)
    cid = {"toric": q.TORIC, "xzzx": q.XZZX, "rotated": q.ROTATED}[name]
    oid = {"toric": orc.TORIC, "xzzx": orc.XZZX, "rotated": orc.ROTATED}[name]
    shape = (N, 2, L, L) if name == "toric" else (N, L, L)
    init = np.zeros(shape, dtype=np.uint8)
    err = rng.random(shape) < p
    init[err] = rng.integers(1, 4, size=int(err.sum()), dtype=np.uint8)
    steps, seed = 24, 31
    kw = dict(Nc=Nc, iters=10, tops_burn=0, seed=seed, code=cid, eta=eta)
    whole = q.pteq_batch(init, p, steps=steps, **kw)
    assert np.array_equal(whole["counts"].sum(axis=1), whole["samples"]) and np.all(whole["samples"] == steps)
    # two shards of the batch, as two ranks would run them (Philox is keyed by the global ladder index)
    h = N // 2 + 64 * 3 + 5                                   # (a ragged cut: the second shard starts inside a workgroup of the whole run)
    a = q.pteq_batch(init[:h], p, steps=steps, first_syndrome=0, **kw)
    b = q.pteq_batch(init[h:], p, steps=steps, first_syndrome=h, **kw)
    for key in ("counts", "samples", "tops0"):
        assert np.array_equal(np.concatenate([a[key], b[key]]), whole[key]), key
    # states and continuation on a slice (the state output of the whole batch is 0.5 GB at L = 15)
    M = 4096
    first = 12345
    sl = slice(first, first + M)
    one = q.pteq_batch(init[sl], p, steps=steps, first_syndrome=first, return_states=True, **kw)
    assert np.array_equal(one["counts"], whole["counts"][sl]) and np.array_equal(one["tops0"], whole["tops0"][sl])
    syn = (lambda m: tm.syndrome(m)) if name == "toric" else (lambda m: _surf.syndrome(cid, m))
    syn0 = syn(init[sl])
    for c in range(Nc):
        assert np.array_equal(syn(np.ascontiguousarray(one["states"][:, c])), syn0), "rung %d lost its syndrome" % c
    pick = rng.choice(M, size=12, replace=False)
    for s in pick:
        ref = orc.pteq_batch(oid, init[first + s:first + s + 1], p, Nc, steps, iters=10, tops_burn=0, seed=seed, first_syndrome=int(first + s),
                             return_states=True, noise=orc.BIASED if eta else orc.DEPOLARIZING, eta=eta or 0.0)
        assert np.array_equal(one["counts"][s], ref["counts"][0]) and int(one["tops0"][s]) == int(ref["tops0"][0])
        assert np.array_equal(one["states"][s], ref["states"][0])


# ------------------------------------------------------------------ the largest shapes the LDS holds, and what lies beyond
@pytest.mark.parametrize("name,L,Nc,eta", [("toric", 20, 8, None), ("toric", 31, 2, None), ("toric", 32, 1, None), ("rotated", 33, 4, None),
                                          ("xzzx", 21, 4, 30.0)])
def test_largest_shapes_bit_exact(q, orc, name, L, Nc, eta):
    rng = np.random.default_rng(L + Nc)
    cid = {"toric": q.TORIC, "xzzx": q.XZZX, "rotated": q.ROTATED}[name]
    oid = {"toric": orc.TORIC, "xzzx": orc.XZZX, "rotated": orc.ROTATED}[name]
    N = 5
    shape = (N, 2, L, L) if name == "toric" else (N, L, L)
    init = (rng.integers(1, 4, size=shape) * (rng.random(shape) < 0.12)).astype(np.uint8)
    got = q.pteq_batch(init, 0.12, Nc=Nc, steps=12, iters=10, tops_burn=0, seed=3, first_syndrome=2, code=cid, eta=eta, return_states=True)
    ref = orc.pteq_batch(oid, init, 0.12, Nc, 12, iters=10, tops_burn=0, seed=3, first_syndrome=2, return_states=True,
                         noise=orc.BIASED if eta else orc.DEPOLARIZING, eta=eta or 0.0)
    assert np.array_equal(got["counts"], ref["counts"]) and np.array_equal(got["tops0"], ref["tops0"].astype(np.uint32))
    assert np.array_equal(got["states"], ref["states"])


@pytest.mark.parametrize("name,L,Nc,eta,what", [("toric", 24, 8, None, "of LDS per workgroup"), ("toric", 40, 2, None, "generators exceed"),
                                               ("rotated", 41, 8, None, "of LDS per workgroup"), ("xzzx", 23, 4, 10.0, "10-bit fields")])
def test_shapes_beyond_the_lds_are_refused_with_the_reason(q, name, L, Nc, eta, what):
    cid = {"toric": q.TORIC, "xzzx": q.XZZX, "rotated": q.ROTATED}[name]
    init = np.zeros((2, 2, L, L) if name == "toric" else (2, L, L), dtype=np.uint8)
    with pytest.raises(q.QecmcError, match=what):
        q.pteq_batch(init, 0.1, Nc=Nc, steps=5, code=cid, eta=eta)
