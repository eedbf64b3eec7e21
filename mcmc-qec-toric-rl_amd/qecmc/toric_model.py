"""Host-side mirror of the reference's `Toric_code` (src/toric_model.py:7-56).

Same attribute and method names, same argument meaning, same return
conventions (`apply_*` return `(new_matrix, error_change)` and never mutate
`self.qubit_matrix`, toric_model.py:258-259), so `decoders.py` /
`generate_data.py`-style callers run unchanged.  Every stencil executes on the
GPU through the C-ABI (batched kernels, N = 1 here); the state is a plain
NumPy uint8 array, so objects stay deep-copyable and picklable.
"""
import random as rand

import numpy as np

from . import _lib as L_


class Toric_code:
    nbr_eq_classes = 16

    def __init__(self, size):
        self.system_size = size
        self.qubit_matrix = np.zeros((2, size, size), dtype=np.uint8)
        self.defect_matrix = np.zeros((2, size, size), dtype=np.uint8)

    # ---- error generation (host RNG, as the reference: toric_model.py:15-31) ----
    def generate_random_error(self, p_error):
        # each qubit errs with probability p_error, Pauli uniform on {X, Y, Z}
        size = self.system_size
        for layer in range(2):
            draws = np.random.uniform(0, 1, size=(size, size))
            pauli = np.random.randint(3, size=(size, size)) + 1
            self.qubit_matrix[layer] = np.where(draws < p_error, pauli, 0).astype(np.uint8)
        self.syndrom()

    def generate_n_random_errors(self, n):
        size = self.system_size
        flat = np.zeros(2 * size * size, dtype=np.uint8)
        flat[:n] = np.random.randint(3, size=n) + 1
        np.random.shuffle(flat)
        self.qubit_matrix[:, :, :] = flat.reshape(2, size, size)
        self.syndrom()

    # ---- device stencils -------------------------------------------------------
    def count_errors(self):
        return int(count_errors(self.qubit_matrix))

    def apply_logical(self, operator: int, layer: int, X_pos=0, Z_pos=0):
        # (the reference wrapper drops `layer`, quirk Q2; here it is honoured)
        return apply_logical(self.qubit_matrix, operator, layer, X_pos, Z_pos)

    def apply_stabilizer(self, row: int, col: int, operator: int):
        return apply_stabilizer(self.qubit_matrix, row, col, operator)

    def apply_random_logical(self):
        # draw order of _apply_random_logical (toric_model.py:228-253)
        size = self.system_size
        ops = [int(rand.random() * 4), int(rand.random() * 4)]
        m, total = self.qubit_matrix, 0
        for layer, op in enumerate(ops):
            x_pos = int(rand.random() * size) if op in (1, 2) else 0
            z_pos = int(rand.random() * size) if op in (3, 2) else 0
            m, d = apply_logical(m, op, layer, x_pos, z_pos)
            total += d
        return m, total

    def apply_random_stabilizer(self):
        # draw order of _apply_random_stabilizer (toric_model.py:287-296)
        size = self.system_size
        row = int(rand.random() * size)
        col = int(rand.random() * size)
        op = int(rand.random() * 2) or 3
        return apply_stabilizer(self.qubit_matrix, row, col, op)

    def apply_stabilizers_uniform(self, p=0.5):
        # toric_model.py:299-314: every generator independently with probability p
        # (axis-0 index 0 means operator 3, index 1 means operator 1)
        size = self.system_size
        pick = np.random.rand(2, size, size) < p
        idx = np.argwhere(pick)
        if len(idx) == 0:
            return self.qubit_matrix.copy()
        m = self.qubit_matrix
        # generators commute and XOR composes, so the batch can be applied one by one on the device
        for o, r, c in idx:
            m, _ = apply_stabilizer(m, int(r), int(c), 3 if o == 0 else 1)
        return m

    def define_equivalence_class(self):
        return int(eq_class(self.qubit_matrix))

    def to_class(self, eq: int):
        return to_class(self.qubit_matrix, eq)

    def syndrom(self):
        self.defect_matrix = syndrome(self.qubit_matrix)


# ---- functional, batch-capable forms (leading axis N optional) -------------------
def _prep(m):
    a, batched = L_.as_states(m, 3)
    if a.shape[1] != 2 or a.shape[2] != a.shape[3]:
        raise ValueError(f"toric qubit_matrix must have shape (2, L, L), got {a.shape[1:]}")
    return a, batched, a.shape[2], a.shape[0]


def _vec(v, n):
    return np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=np.int32), (n,)))


def apply_stabilizer(m, row, col, operator):
    a, batched, size, n = _prep(m)
    out = np.empty_like(a)
    dE = np.empty(n, dtype=np.int32)
    r, c, o = _vec(row, n), _vec(col, n), _vec(operator, n)
    L_.check(L_.lib().qecmc_apply_stabilizer(L_.TORIC, size, n, L_.u8(a), L_.u8(out), L_.i32(r), L_.i32(c),
                                             L_.i32(o), L_.i32(dE)))
    return (out, dE) if batched else (out[0], int(dE[0]))


def apply_logical(m, operator, layer, X_pos=0, Z_pos=0):
    a, batched, size, n = _prep(m)
    out = np.empty_like(a)
    dE = np.empty(n, dtype=np.int32)
    o, l, x, z = _vec(operator, n), _vec(layer, n), _vec(X_pos, n), _vec(Z_pos, n)
    L_.check(L_.lib().qecmc_apply_logical(L_.TORIC, size, n, L_.u8(a), L_.u8(out), L_.i32(o), L_.i32(l),
                                          L_.i32(x), L_.i32(z), L_.i32(dE)))
    return (out, dE) if batched else (out[0], int(dE[0]))


def count_errors(m):
    a, batched, size, n = _prep(m)
    out = np.empty(n, dtype=np.int64)
    L_.check(L_.lib().qecmc_count_errors(L_.TORIC, size, n, L_.u8(a), out.ctypes.data_as(L_._i64p)))
    return out if batched else int(out[0])


def eq_class(m):
    a, batched, size, n = _prep(m)
    out = np.empty(n, dtype=np.int32)
    L_.check(L_.lib().qecmc_eq_class(L_.TORIC, size, n, L_.u8(a), L_.i32(out)))
    return out if batched else int(out[0])


def to_class(m, eq):
    a, batched, size, n = _prep(m)
    out = np.empty_like(a)
    e = _vec(eq, n)
    L_.check(L_.lib().qecmc_to_class(L_.TORIC, size, n, L_.u8(a), L_.u8(out), L_.i32(e)))
    return out if batched else out[0]


def syndrome(m):
    a, batched, size, n = _prep(m)
    out = np.empty_like(a)
    L_.check(L_.lib().qecmc_syndrome(L_.TORIC, size, n, L_.u8(a), L_.u8(out)))
    return out if batched else out[0]
