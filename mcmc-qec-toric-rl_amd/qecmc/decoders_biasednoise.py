"""Host-side mirror of `PTEQ_biased` (decoders_biasednoise.py:28-75) and `PTEQ_alpha` (:175-226): the PTEQ loop around
Ladder_biased / Ladder_alpha."""
from .decoders import _pteq


def PTEQ_biased(init_code, p, eta=0.5, Nc=None, SEQ=2, TOPS=10, tops_burn=2, eps=0.1, steps=50000000, iters=10,
                conv_criteria='error_based', seed=None, replicas=None, scan="random"):
    """decoders_biasednoise.PTEQ_biased (:28-90).  scan="colour": the one-syndrome latency layout (a workgroup per ladder, a colour phase of
    generators per wavefront pass; every generator a Metropolis move for the biased weight -- the reference's rule at iters = 1)."""
    return _pteq(init_code, p, eta, Nc, SEQ, TOPS, tops_burn, eps, steps, iters, conv_criteria, seed, replicas=replicas, scan=scan)


def PTEQ_alpha(init_code, pz_tilde, alpha=1, Nc=None, SEQ=2, TOPS=10, tops_burn=2, eps=0.1, steps=50000000, iters=10,
               conv_criteria='error_based', seed=None, replicas=None, scan="random"):
    """decoders_biasednoise.PTEQ_alpha (:175-238).  scan="colour": the one-syndrome latency layout; scan="wave": the batched throughput layout."""
    return _pteq(init_code, pz_tilde, None, Nc, SEQ, TOPS, tops_burn, eps, steps, iters, conv_criteria, seed, alpha=alpha, replicas=replicas, scan=scan)


def _shortest_loop(ladder, pz_tilde, SEQ, TOPS, tops_burn, eps, steps, iters, conv_criteria):
    """The bookkeeping of PTEQ_alpha_with_shortest (decoders_biasednoise.py:93-172) around any Ladder_alpha-like object
    (`.step(iters)`, `.tops0`, `.chains[0].n_eff`, `.chains[0].code`).  Besides PTEQ_alpha's class histogram it keeps, per
    class, the smallest n_eff the bottom slot showed after burn-in, how often, and the distinct configurations seen with it."""
    import warnings
    import numpy as np
    from math import exp
    nbr_eq_classes = ladder.chains[0].code.nbr_eq_classes
    counts = np.zeros(nbr_eq_classes, dtype=np.uint32)                 # eq[since_burn], kept as a running row (:103,:123-124)
    log = np.zeros(1024)                                               # nbr_errors_bottom_chain, grown on demand (:101)
    since_burn = burn = 0
    conv_start = conv_streak = 0
    unique = [dict() for _ in range(nbr_eq_classes)]                   # :112
    shortest_n = [0] * nbr_eq_classes
    shortest = [100000] * nbr_eq_classes
    for step in range(int(steps)):
        ladder.step(iters)                                             # :120
        bottom = ladder.chains[0]
        cls = int(bottom.code.define_equivalence_class())              # :122
        if ladder.tops0 >= tops_burn:                                  # :124
            since_burn = step - burn
            counts[cls] += 1
            if since_burn >= log.size:
                log = np.concatenate([log, np.zeros(log.size)])
            n_eff = log[since_burn] = bottom.n_eff                     # :128 -- the slot's attribute, possibly stale (quirk Q4)
            if n_eff < shortest[cls]:                                  # :130-138: a new minimum restarts the class's set
                shortest_n[cls], shortest[cls] = 1, n_eff
                unique[cls] = {bottom.code.qubit_matrix.tobytes(): n_eff}
            elif n_eff == shortest[cls]:                               # :139-144
                shortest_n[cls] += 1
                unique[cls].setdefault(bottom.code.qubit_matrix.tobytes(), n_eff)
        else:
            burn += 1                                                  # :147
        if conv_criteria == 'error_based' and ladder.tops0 >= TOPS:    # :149-157; the criterion of :226-238
            l = since_burn + 1
            with np.errstate(invalid="ignore"), warnings.catch_warnings():
                warnings.simplefilter("ignore")                       # an empty quarter averages to nan: not accepted
                err = abs(np.average(log[l // 4: l // 2]) - np.average(log[3 * l // 4: l]))
            if err < eps:
                if conv_streak >= SEQ:
                    break
                conv_streak = ladder.tops0 - conv_start
            else:
                conv_streak, conv_start = 0, ladder.tops0
    beta = -np.log(pz_tilde)                                           # :163
    eqdistr = np.array([sum(exp(-beta * v) for v in u.values()) for u in unique])       # :165-167
    with np.errstate(divide="ignore", invalid="ignore"):
        return ((np.divide(counts, since_burn + 1) * 100).astype(np.uint8), np.divide(eqdistr, sum(eqdistr)) * 100,
                np.array(shortest_n) / sum(shortest_n) * 100)          # :170


def PTEQ_alpha_with_shortest(init_code, pz_tilde, alpha=1, Nc=None, SEQ=2, TOPS=10, tops_burn=2, eps=0.1, steps=50000000, iters=10,
                             conv_criteria='error_based', seed=None):
    """Drop-in for decoders_biasednoise.PTEQ_alpha_with_shortest (:93-172; generate_data.py:162-167, method
    "PTEQ_with_shortest"): returns (PTEQ_alpha's uint8 percent vector, the percent vector from the distinct shortest chains,
    the percent of observations at the shortest n_eff per class).  An analysis variant: the ladder runs on the GPU one
    `Ladder_alpha.step` per launch and the bookkeeping on the host, so it is launch-bound (~10^4 ladder steps/s) -- the
    batched, in-kernel decoders are PTEQ_alpha / pteq_batch."""
    from .mcmc_alpha import Ladder_alpha
    ladder = Ladder_alpha(pz_tilde, init_code, alpha, Nc or init_code.system_size, 0.5, seed=seed)     # :109
    return _shortest_loop(ladder, pz_tilde, SEQ, TOPS, tops_burn, eps, steps, iters, conv_criteria)


# decoders_biasednoise.py:79-90 / :226-237: the same criterion under two more names
from .decoders import conv_crit_error_based_PT as conv_crit_error_based_PT_biased     # noqa: E402
from .decoders import conv_crit_error_based_PT as conv_crit_error_based_PT_alpha      # noqa: E402
