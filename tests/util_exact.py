"""Reference-independent oracle: exact equivalence-class probabilities of a toric-code syndrome by
enumerating the stabilizer group.  P(class) = Z_class / sum Z, Z = sum f^weight over
all chains in the class, f = (p/3)/(1-p).  Feasible at L=3 (2^9 x 2^9 generator subsets x 16 classes)."""
import numpy as np


def _bits(m):
    """uint8[2,L,L] Pauli matrix -> (xbits, zbits) python ints over the flattened qubits."""
    flat = np.asarray(m, dtype=np.uint8).ravel()
    x = sum(1 << i for i, v in enumerate(flat) if v in (1, 2))
    z = sum(1 << i for i, v in enumerate(flat) if v in (2, 3))
    return x, z


def _popcount(a):
    a = a.astype(np.uint64)
    c = np.zeros(a.shape, dtype=np.int64)
    while a.any():
        c += (a & np.uint64(1)).astype(np.int64)
        a >>= np.uint64(1)
    return c


def toric_class_probabilities(init, p, apply_stabilizer, to_class):
    """init: uint8[2,L,L]; apply_stabilizer(m,row,col,op)->(m',dE); to_class(m,eq)->m'."""
    init = np.asarray(init, dtype=np.uint8)
    L = init.shape[1]
    zero = np.zeros_like(init)
    gens_x = [_bits(apply_stabilizer(zero, r, c, 1)[0])[0] for r in range(L) for c in range(L)]
    gens_z = [_bits(apply_stabilizer(zero, r, c, 3)[0])[1] for r in range(L) for c in range(L)]

    def span(gens):
        out = np.zeros(1, dtype=np.uint64)
        for g in gens:
            out = np.concatenate([out, out ^ np.uint64(g)])
        return out                                   # every group element appears the same number of times
    sx, sz = span(gens_x), span(gens_z)
    f = (p / 3.0) / (1.0 - p)
    zsum = np.zeros(16)
    for eq in range(16):
        bx, bz = _bits(to_class(init, eq))
        w = _popcount((np.uint64(bx) ^ sx)[:, None] | (np.uint64(bz) ^ sz)[None, :])
        zsum[eq] = np.sum(f ** w.astype(np.float64))
    return zsum / zsum.sum()


# ---- XZZX / rotated codes (L x L qubits, 4 classes; SURVEY.md 8c: "2^8 elements x 4" at L = 3) -----------------------------------
# Reference-independent like the toric enumeration above: the stabilizer group is spanned from the generators' own Pauli
# patterns (apply_stabilizer on the empty lattice), the class representatives come from the logical operators, and a
# configuration's weight is the noise model's, written out here: f^n for depolarizing noise (src/mcmc.py:16), and
# px^nx py^ny pz^nz pI^nI with pz = p eta / (eta + 1), px = py = p / (2 (eta + 1)) for biased noise (src/mcmc_biased.py:25-31).

def depolarizing_weight(p):
    f = (p / 3.0) / (1.0 - p)
    return lambda cfg: f ** np.count_nonzero(cfg, axis=-1).astype(np.float64)


def biased_weight(p, eta):
    pz, px = p * eta / (eta + 1.0), p / (2.0 * (eta + 1.0))
    py, pi = px, 1.0 - p
    def w(cfg):
        cfg = np.asarray(cfg)
        nx, ny, nz = (cfg == 1).sum(-1), (cfg == 2).sum(-1), (cfg == 3).sum(-1)
        return px ** nx * py ** ny * pz ** nz * pi ** (cfg.shape[-1] - nx - ny - nz)
    return w


class SurfEnumeration:
    """All configurations of one syndrome of an L x L plaquette code: cfg[c, b] = representative of class c, times the group
    element with generator subset b (uint8[4, 2^G, L*L]; Paulis 0..3 compose by XOR).  `api` provides apply_stabilizer(code, m,
    row, col, op) -> (m', dE), apply_logical(code, m, op, xpos, zpos) -> (m', dE), eq_class(code, m), ngen(code, L),
    gen_rco(code, L, g) -> (row, col, op) -- the oracle module or the device-backed qecmc._surf fit."""

    def __init__(self, code, init, api):
        init = np.asarray(init, dtype=np.uint8)
        self.code, self.L, self.api = code, init.shape[-1], api
        L = self.L
        zero = np.zeros((L, L), dtype=np.uint8)
        self.G = api.ngen(code, L)
        assert self.G <= 12, "enumeration is meant for L = 3"
        gens = [np.asarray(api.apply_stabilizer(code, zero, *api.gen_rco(code, L, g))[0], dtype=np.uint8).ravel() for g in range(self.G)]
        grp = np.zeros((1, L * L), dtype=np.uint8)
        for g in gens:                                  # subset b: bit j set <=> generator j applied
            grp = np.concatenate([grp, grp ^ g])
        assert len({e.tobytes() for e in grp}) == 1 << self.G, "generators are not independent"
        reps = [None] * 4
        for k in range(4):                              # apply_logical(m, k, 0, 0) maps class c -> c ^ k
            r = np.asarray(api.apply_logical(code, init, k, 0, 0)[0], dtype=np.uint8)
            reps[int(api.eq_class(code, r))] = r.ravel()
        assert all(r is not None for r in reps)
        self.cfg = np.stack([grp ^ r for r in reps])    # [4, 2^G, nq]
        self.index = {self.cfg[c, b].tobytes(): (c, b) for c in range(4) for b in range(1 << self.G)}
        assert len(self.index) == 4 << self.G

    def class_probabilities(self, weight):
        z = weight(self.cfg).sum(axis=1)
        return z / z.sum()

    def q3_class_law(self, weight, p_logical, iters):
        """Stationary class distribution of ONE chain advanced by update_chain(iters) calls under the biased rule as the reference
        has it (quirk Q3, src/mcmc_biased.py:28-46): every proposal of a call is accepted with min(1, w(new) / w(entry)), `entry`
        being the configuration at the start of the call -- not the current one.  Exact: the transition matrix of one call, row
        by row (row x0 = e_x0 K_x0^iters with K_x0 the one-proposal kernel that tests against w(x0)), then its fixed point.
        The chain is a "top" chain: with probability p_logical a uniformly drawn logical operator (xzzx_model.py:340-357,
        rotated_surface_model.py:331-346: op in 0..3, X_pos drawn iff op in {1,2}, Z_pos iff op in {3,2}), else a uniformly
        drawn generator."""
        api, code, L, G = self.api, self.code, self.L, self.G
        n = 4 << G
        flat = self.cfg.reshape(n, -1)
        w = weight(flat)
        moves, probs = [], []
        for g in range(G):                              # generator g: subset bit flips
            idx = np.arange(n)
            moves.append((idx & ~((1 << G) - 1)) | ((idx & ((1 << G) - 1)) ^ (1 << g)))
            probs.append((1.0 - p_logical) / G)
        if p_logical > 0:
            for op in range(4):
                for xpos in range(L):
                    for zpos in range(L):
                        xp = xpos if op in (1, 2) else 0
                        zp = zpos if op in (3, 2) else 0
                        to = np.empty(n, dtype=np.int64)
                        for i in range(n):
                            y = np.asarray(api.apply_logical(code, flat[i].reshape(L, L), op, xp, zp)[0], dtype=np.uint8)
                            c, b = self.index[y.tobytes()]
                            to[i] = (c << G) | b
                        moves.append(to)
                        probs.append(p_logical / (4.0 * L * L))
        moves, probs = np.stack(moves, axis=1), np.asarray(probs)           # [n, M], [M]
        T = np.zeros((n, n))
        for x0 in range(n):
            acc = np.minimum(1.0, w / w[x0])
            pm = probs[None, :] * acc[moves]                                 # probability of moving x -> moves[x, m]
            stay = 1.0 - pm.sum(axis=1)
            v = np.zeros(n); v[x0] = 1.0
            for _ in range(iters):
                nv = v * stay
                np.add.at(nv, moves.ravel(), (v[:, None] * pm).ravel())
                v = nv
            T[x0] = v
        pi = np.full(n, 1.0 / n)
        for _ in range(200000):
            nxt = pi @ T
            if np.abs(nxt - pi).max() < 1e-15:
                break
            pi = nxt
        return pi.reshape(4, -1).sum(axis=1)
