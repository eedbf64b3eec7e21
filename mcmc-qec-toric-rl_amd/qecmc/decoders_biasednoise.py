"""Host-side mirror of `PTEQ_biased` (decoders_biasednoise.py:28-75) and `PTEQ_alpha` (:175-226): the PTEQ loop around
Ladder_biased / Ladder_alpha."""
from .decoders import _pteq


def PTEQ_biased(init_code, p, eta=0.5, Nc=None, SEQ=2, TOPS=10, tops_burn=2, eps=0.1, steps=50000000, iters=10,
                conv_criteria='error_based', seed=None):
    return _pteq(init_code, p, eta, Nc, SEQ, TOPS, tops_burn, eps, steps, iters, conv_criteria, seed)


def PTEQ_alpha(init_code, pz_tilde, alpha=1, Nc=None, SEQ=2, TOPS=10, tops_burn=2, eps=0.1, steps=50000000, iters=10,
               conv_criteria='error_based', seed=None):
    return _pteq(init_code, pz_tilde, None, Nc, SEQ, TOPS, tops_burn, eps, steps, iters, conv_criteria, seed, alpha=alpha)
