"""Host-side mirror of `Chain_biased` / `Ladder_biased` (src/mcmc_biased.py:10-124): Z-biased noise,
acceptance from the full (n_x, n_y, n_z) counts.  Reference quirk Q3 (p_b frozen at loop entry) is reproduced."""
from .mcmc import Chain, Ladder


class Chain_biased(Chain):
    def __init__(self, p, eta, code, seed=None, stream=0):
        super().__init__(p, code, seed=seed, stream=stream)
        self.eta = eta
        self._eta = eta


class Ladder_biased(Ladder):
    def __init__(self, p_bottom, init_code, eta, Nc, p_logical=0, seed=None, stream=0):
        self.eta = eta
        self._eta = eta
        super().__init__(p_bottom, init_code, Nc, p_logical, seed=seed, stream=stream)
        self._bottom = p_bottom

    def _make_chain(self, p, code, stream):
        return Chain_biased(p, self.eta, code, seed=self.seed, stream=stream)
