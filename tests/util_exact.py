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
