"""Host-side mirror of `Chain_alpha` / `Ladder_alpha` (src/mcmc_alpha.py:10-137): the "alpha" noise model, where the
ladder variable is pz_tilde = p_z / (1 - p) and p_x = p_y = pz_tilde**alpha (1 - p).  Acceptance is the biased rule
(p_b frozen at loop entry, quirk Q3); swaps compare the chains' `n_eff` attributes, which -- as in the reference -- stay
with the ladder slot when the codes are exchanged and are refreshed by accepted moves only (quirk Q4)."""
import copy

import numpy as np

from . import _lib as L_
from .mcmc import _code_id, _fresh_seed


def _counts(m):
    m = np.asarray(m)
    return int(np.count_nonzero(m == 3)), int(np.count_nonzero(m == 1) + np.count_nonzero(m == 2))


class Chain_alpha:
    def __init__(self, pz_tilde, alpha, code, seed=None, stream=0):
        self.code = code
        self.pz_tilde = pz_tilde
        self.alpha = alpha
        self.p_logical = 0
        self.flag = 0
        self._nz, self._nxy = _counts(code.qubit_matrix)           # mcmc_alpha.py:18-22
        self.seed = _fresh_seed() if seed is None else seed
        self.stream = stream
        self.slot = 0
        self.proposals_done = 0

    @property
    def n_eff(self):
        return self._nz + self.alpha * self._nxy                   # mcmc_alpha.py:22,58

    def update_chain(self, iters):
        """`iters` proposals (src/mcmc_alpha.py:27-70) in one kernel launch."""
        m, _ = L_.as_states(self.code.qubit_matrix, self.code.qubit_matrix.ndim)
        m = m.copy()
        acc = np.zeros(1, dtype=np.uint8)
        L_.check(L_.lib().qecmc_chain_update_alpha(_code_id(self.code), self.code.system_size, 1, L_.u8(m),
                                                   float(self.pz_tilde), float(self.alpha), float(self.p_logical),
                                                   int(iters), self.seed, self.stream, self.slot, self.proposals_done,
                                                   L_.u8(acc)))
        self.proposals_done += int(iters)
        self.code.qubit_matrix = m[0]
        if acc[0]:
            self._nz, self._nxy = _counts(m[0])


    def update_chain_fast(self, iters):
        # mcmc_alpha.py:73-74 reads `self.factor`, which Chain_alpha never sets: the reference raises here too
        raise AttributeError("'Chain_alpha' object has no attribute 'factor'")


class Ladder_alpha:
    def __init__(self, pz_tilde_bottom, init_code, alpha, Nc, p_logical=0, seed=None, stream=0):
        self.alpha = alpha
        self.pz_tilde_bottom = pz_tilde_bottom
        self.init_code = init_code
        self.Nc = Nc
        self.p_logical = p_logical
        pz_tilde_ladder = np.linspace(pz_tilde_bottom, 1, Nc)      # pz_tilde_top = 1, mcmc_alpha.py:94-97
        self.pz_tilde_ladder = pz_tilde_ladder
        with np.errstate(divide="ignore", invalid="ignore"):
            self.pz_tilde_diff = (pz_tilde_ladder[:-1] * (1 - pz_tilde_ladder[1:])) / (pz_tilde_ladder[1:] * (1 - pz_tilde_ladder[:-1]))
        self.seed = _fresh_seed() if seed is None else seed
        self.stream = stream
        self.chains = [Chain_alpha(pz, alpha, copy.deepcopy(init_code), seed=self.seed, stream=stream) for pz in pz_tilde_ladder]
        for slot, ch in enumerate(self.chains):
            ch.slot = slot
        self.chains[-1].flag = 1
        self.chains[-1].p_logical = p_logical
        self.tops0 = 0
        self.steps_done = 0
        self.proposals_done = 0

    def update_ladder(self, iters):
        for ch in self.chains:
            ch.proposals_done = self.proposals_done
            ch.update_chain(iters)
        self.proposals_done += int(iters)

    def r_flip(self, ind_lo):
        """mcmc_alpha.py:117-123: the slots' n_eff attributes (quirk Q4) and the ratio of their pz_tilde's; always draws"""
        import random as rand
        lo, hi = self.chains[ind_lo], self.chains[ind_lo + 1]
        return rand.random() < (lo.pz_tilde / hi.pz_tilde) ** (hi.n_eff - lo.n_eff)

    def step(self, iters, nsteps=1):
        """`nsteps` x Ladder_alpha.step(iters) (src/mcmc_alpha.py:127-137) in one kernel launch."""
        code = self.chains[0].code
        states = np.ascontiguousarray(np.stack([ch.code.qubit_matrix for ch in self.chains])[None], dtype=np.uint8)
        flags = np.array([[ch.flag for ch in self.chains]], dtype=np.uint8)
        tops0 = np.array([self.tops0], dtype=np.uint32)
        neff = np.array([[[ch._nz, ch._nxy] for ch in self.chains]], dtype=np.uint16)
        pr = L_.make_params(code=_code_id(code), L=code.system_size, Nc=self.Nc, p=float(self.pz_tilde_bottom),
                            p_logical=float(self.p_logical), iters=int(iters), seed=self.seed, first_syndrome=self.stream,
                            noise=L_.NOISE_ALPHA, alpha=float(self.alpha))
        L_.check(L_.lib().qecmc_ladder_step_alpha(pr, 1, L_.u8(states), L_.u8(flags), L_.u32(tops0), L_.u16(neff),
                                                  int(iters), int(nsteps), self.steps_done, self.proposals_done))
        self.steps_done += int(nsteps)
        self.proposals_done += int(iters) * int(nsteps)
        for c, ch in enumerate(self.chains):
            ch.code.qubit_matrix = states[0, c].copy()
            ch.flag = int(flags[0, c])
            ch._nz, ch._nxy = int(neff[0, c, 0]), int(neff[0, c, 1])
            ch.proposals_done = self.proposals_done
        self.tops0 = int(tops0[0])
