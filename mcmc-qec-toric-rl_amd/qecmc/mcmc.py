"""Host-side mirror of the reference's `Chain` / `Ladder` (src/mcmc.py:10-103).

Same constructor arguments and attributes (`code`, `p`, `p_logical`, `flag`,
`factor`; `chains`, `tops0`, `Nc`, `p_ladder`, `p_diff`).  The Metropolis loop
itself runs on the GPU: `Chain.update_chain(iters)` and `Ladder.step(iters)`
are each ONE call through the C-ABI (qecmc_chain_update / qecmc_ladder_step),
drawing from counter-based Philox streams instead of CPython's global MT19937
(the reference is unseeded, quirk Q8: only the distribution is defined).
"""
import copy
import random as rand

import numpy as np

from . import _lib as L_


def _fresh_seed():
    # follows `random.seed(...)`, so a seeded caller gets reproducible runs
    return rand.getrandbits(64)


def _code_id(code):
    name = type(code).__name__
    ids = {"Toric_code": L_.TORIC, "xzzx_code": L_.XZZX, "RotSurCode": L_.ROTATED, "Planar_code": L_.PLANAR}
    if name not in ids:
        raise NotImplementedError(f"no GPU kernels for code model {name} in this build")
    return ids[name]


class Chain:
    _eta = None           # Chain_biased sets the bias

    def __init__(self, p, code, seed=None, stream=0):
        self.code = code
        self.p = p
        self.p_logical = 0
        self.flag = 0
        self.factor = ((self.p / 3.0) / (1.0 - self.p))
        self.seed = _fresh_seed() if seed is None else seed
        self.stream = stream          # Philox syndrome index of this chain
        self.slot = 0                 # Philox stream id (ladder slot)
        self.proposals_done = 0

    def update_chain(self, iters):
        """`iters` Metropolis proposals (src/mcmc.py:19-43) in one kernel launch."""
        m, _ = L_.as_states(self.code.qubit_matrix, self.code.qubit_matrix.ndim)
        m = m.copy()
        if self._eta is None:
            L_.check(L_.lib().qecmc_chain_update(_code_id(self.code), self.code.system_size, 1, L_.u8(m), float(self.p),
                                                 float(self.p_logical), int(iters), self.seed, self.stream, self.slot,
                                                 self.proposals_done))
        else:
            L_.check(L_.lib().qecmc_chain_update_biased(_code_id(self.code), self.code.system_size, 1, L_.u8(m),
                                                        float(self.p), float(self._eta), float(self.p_logical), int(iters),
                                                        self.seed, self.stream, self.slot, self.proposals_done))
        self.proposals_done += int(iters)
        self.code.qubit_matrix = m[0]

    def update_chain_fast(self, iters):
        # the reference's jitted loop (mcmc.py:152-160) is the non-top branch of update_chain, hard-wired to the planar
        # stencil (quirk Q1): on a Planar_code it is that chain; on the other codes the reference would corrupt the
        # state, and the only sensible reading is the same chain as update_chain on the code's own stencil
        p_logical, self.p_logical = self.p_logical, 0
        try:
            self.update_chain(iters)
        finally:
            self.p_logical = p_logical


class Chain_xyz:
    """src/mcmc.py:106-114: a single chain at general noise (p_x, p_y, p_z) -- what decoders.py:352,442 sample STDC_general_noise
    with.  `update_chain_fast(iters)` (:112-114 -> :162-173) is one kernel launch (qecmc_chain_update_xyz): a generator proposal of the
    code's own stencil, accepted with probability prod_i (p_i / (1 - sum p))^(change of n_i); `qubit_errors` follows the chain."""

    def __init__(self, p_xyz, code, seed=None, stream=0):
        self.code = code
        self.p_xyz = np.asarray(p_xyz, dtype=np.float64)
        self.factors = self.p_xyz / (1.0 - self.p_xyz.sum())
        self.qubit_errors = code.count_errors_xyz()
        self.seed = _fresh_seed() if seed is None else seed
        self.stream = stream
        self.slot = 0
        self.proposals_done = 0

    def update_chain_fast(self, iters):
        import ctypes as C
        m, _ = L_.as_states(self.code.qubit_matrix, self.code.qubit_matrix.ndim)
        m = m.copy()
        pxyz = (C.c_double * 3)(*[float(v) for v in self.p_xyz])
        L_.check(L_.lib().qecmc_chain_update_xyz(_code_id(self.code), self.code.system_size, 1, L_.u8(m), pxyz, int(iters), self.seed,
                                                 self.stream, self.slot, self.proposals_done))
        self.proposals_done += int(iters)
        self.code.qubit_matrix = m[0]
        self.qubit_errors = self.code.count_errors_xyz()


class Ladder:
    _eta = None           # Ladder_biased sets the bias
    _chain_cls = None

    def __init__(self, p_bottom, init_code, Nc, p_logical=0, seed=None, stream=0):
        self.p_bottom = p_bottom
        self.init_code = init_code
        self.Nc = Nc
        self.p_logical = p_logical
        p_top = 0.75 if self._eta is None else (self._eta + 1) / (2 * self._eta + 1)
        p_ladder = np.linspace(p_bottom, p_top, Nc)
        self.p_ladder = p_ladder
        self.p_diff = (p_ladder[:-1] * (1 - p_ladder[1:])) / (p_ladder[1:] * (1 - p_ladder[:-1]))
        self.seed = _fresh_seed() if seed is None else seed
        self.stream = stream
        self.chains = [self._make_chain(p, copy.deepcopy(init_code), stream) for p in p_ladder]
        for slot, ch in enumerate(self.chains):
            ch.slot = slot
        self.chains[-1].flag = 1
        self.chains[-1].p_logical = p_logical
        self.tops0 = 0
        self.steps_done = 0
        self.proposals_done = 0

    def _make_chain(self, p, code, stream):
        return Chain(p, code, seed=self.seed, stream=stream)

    def _params(self, iters):
        code = self.chains[0].code
        return L_.make_params(code=_code_id(code), L=code.system_size, Nc=self.Nc, p=float(self.p_bottom),
                              p_logical=float(self.p_logical), iters=int(iters), seed=self.seed,
                              first_syndrome=self.stream, noise=L_.NOISE_DEPOLARIZING if self._eta is None else L_.NOISE_BIASED,
                              eta=0.0 if self._eta is None else float(self._eta))

    def update_ladder(self, iters):
        for ch in self.chains:
            ch.proposals_done = self.proposals_done
            ch.update_chain(iters)
        self.proposals_done += int(iters)

    def r_flip(self, ind_lo):
        """src/mcmc.py:86-92 with _r_flip (:144-149) -- mcmc_biased.py:107-113 likewise: should rungs ind_lo / ind_lo + 1 be swapped?
        (Host-side, on Python's `random` like the reference; `step` runs the whole sweep on the GPU and does not call it.)"""
        ne_lo = self.chains[ind_lo].code.count_errors()
        ne_hi = self.chains[ind_lo + 1].code.count_errors()
        if ne_hi < ne_lo and self._eta is None:
            return True
        return rand.random() < self.p_diff[ind_lo] ** (ne_hi - ne_lo)

    def step(self, iters, nsteps=1):
        """`nsteps` x Ladder.step(iters) (src/mcmc.py:94-103) in one kernel launch."""
        states = np.ascontiguousarray(np.stack([ch.code.qubit_matrix for ch in self.chains])[None], dtype=np.uint8)
        flags = np.array([[ch.flag for ch in self.chains]], dtype=np.uint8)
        tops0 = np.array([self.tops0], dtype=np.uint32)
        pr = self._params(iters)
        L_.check(L_.lib().qecmc_ladder_step(pr, 1, L_.u8(states), L_.u8(flags), L_.u32(tops0), int(iters), int(nsteps),
                                            self.steps_done, self.proposals_done))
        self.steps_done += int(nsteps)
        self.proposals_done += int(iters) * int(nsteps)
        for c, ch in enumerate(self.chains):
            ch.code.qubit_matrix = states[0, c].copy()
            ch.flag = int(flags[0, c])
            ch.proposals_done = self.proposals_done
        self.tops0 = int(tops0[0])
