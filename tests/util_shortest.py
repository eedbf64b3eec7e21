"""An oracle-backed stand-in for qecmc's Ladder_alpha, so that the host bookkeeping of PTEQ_alpha_with_shortest
(qecmc.decoders_biasednoise._shortest_loop) can be run on the oracle's ladder: in injected-stream mode that pins the loop
against the reference's fixture; in Philox mode it is the CPU twin of the GPU run."""
from types import SimpleNamespace

from oracle import oracle as orc


class OracleLadderAlpha:
    def __init__(self, code, init, pz_tilde, alpha, Nc, rng, det_pow=0):
        self._code, self._rng = code, rng
        self._ld = orc.Ladder(code, init, pz_tilde, Nc, p_logical=0.5, noise=orc.ALPHA, alpha=alpha, det_pow=det_pow)
        self.chains = [self]                      # only the bottom slot is looked at
        self.code = SimpleNamespace(nbr_eq_classes=4, define_equivalence_class=self._cls, qubit_matrix=None)
        self._refresh()

    def _cls(self):
        return orc.surf_eq_class(self._code, self.code.qubit_matrix)

    def _refresh(self):
        self.code.qubit_matrix = self._ld.states[0]
        self.n_eff = float(self._ld.n_eff[0])
        self.tops0 = self._ld.tops0

    def step(self, iters):
        self._ld.step(iters, self._rng)
        self._refresh()
