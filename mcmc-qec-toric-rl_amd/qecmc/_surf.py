"""Shared implementation of the two L x L plaquette codes (XZZX, rotated surface code): functional,
batch-capable device stencils; the classes in xzzx_model.py / rotated_surface_model.py wrap them."""
import numpy as np

from . import _lib as L_


def _prep(m):
    a, batched = L_.as_states(m, 2)
    if a.shape[1] != a.shape[2]:
        raise ValueError(f"qubit_matrix must have shape (L, L), got {a.shape[1:]}")
    return a, batched, a.shape[1], a.shape[0]


def pauli_field(shape, p_x, p_y, p_z):
    """One uniform per qubit from the `random` module in C order, mapped to a Pauli as the reference's generators do:
    Z below p_z, X in (p_z, p_z + p_x), Y in (p_z + p_x, p_z + p_x + p_y) -- open intervals, xzzx_model.py:16-30 (elif chain),
    rotated_surface_model.py:25-38 (independent ifs: the same outcome, the intervals are disjoint), planar_model.py:18-36."""
    import random as rand
    r = np.array([rand.random() for _ in range(int(np.prod(shape)))], dtype=np.float64).reshape(shape)
    m = np.zeros(shape, dtype=np.uint8)
    m[r < p_z] = 3
    m[(p_z < r) & (r < p_z + p_x)] = 1
    m[(p_z + p_x < r) & (r < p_z + p_x + p_y)] = 2
    return m


def _vec(v, n):
    return np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=np.int32), (n,)))


def apply_stabilizer(code, m, row, col, operator):
    a, batched, size, n = _prep(m)
    out = np.empty_like(a)
    dE = np.empty(n, dtype=np.int32)
    r, c, o = _vec(row, n), _vec(col, n), _vec(operator, n)
    L_.check(L_.lib().qecmc_apply_stabilizer(code, size, n, L_.u8(a), L_.u8(out), L_.i32(r), L_.i32(c), L_.i32(o), L_.i32(dE)))
    return (out, dE) if batched else (out[0], int(dE[0]))


def apply_logical(code, m, operator, X_pos=0, Z_pos=0):
    a, batched, size, n = _prep(m)
    out = np.empty_like(a)
    dE = np.empty(n, dtype=np.int32)
    o, x, z, lay = _vec(operator, n), _vec(X_pos, n), _vec(Z_pos, n), _vec(0, n)
    L_.check(L_.lib().qecmc_apply_logical(code, size, n, L_.u8(a), L_.u8(out), L_.i32(o), L_.i32(lay), L_.i32(x), L_.i32(z),
                                          L_.i32(dE)))
    return (out, dE) if batched else (out[0], int(dE[0]))


def count_errors(code, m):
    a, batched, size, n = _prep(m)
    out = np.empty(n, dtype=np.int64)
    L_.check(L_.lib().qecmc_count_errors(code, size, n, L_.u8(a), out.ctypes.data_as(L_._i64p)))
    return out if batched else int(out[0])


def eq_class(code, m):
    a, batched, size, n = _prep(m)
    out = np.empty(n, dtype=np.int32)
    L_.check(L_.lib().qecmc_eq_class(code, size, n, L_.u8(a), L_.i32(out)))
    return out if batched else int(out[0])


def syndrome(code, m):
    a, batched, size, n = _prep(m)
    out = np.empty((n, size + 1, size + 1), dtype=np.uint8)
    L_.check(L_.lib().qecmc_syndrome(code, size, n, L_.u8(a), L_.u8(out)))
    return out if batched else out[0]


class PlaquetteCode:
    """Common host-side mirror of xzzx_code (src/xzzx_model.py:8-58) and RotSurCode
    (src/rotated_surface_model.py:8-106): same attributes, methods, return conventions."""
    nbr_eq_classes = 4
    _code = None

    def __init__(self, size):
        if size < 3 or size % 2 == 0:
            raise ValueError("the xzzx / rotated models need odd size >= 3 (half-plaquette indexing of the reference)")
        self.system_size = size
        self.qubit_matrix = np.zeros((size, size), dtype=np.uint8)
        self.plaquette_defects = np.zeros((size + 1, size + 1))

    def generate_random_error(self, p_x, p_y, p_z):
        self.qubit_matrix = pauli_field((self.system_size, self.system_size), p_x, p_y, p_z)
        self.syndrome()

    def count_errors(self):
        return count_errors(self._code, self.qubit_matrix)

    def chain_lengths(self):
        m = self.qubit_matrix
        return int((m == 1).sum()), int((m == 2).sum()), int((m == 3).sum())

    def apply_logical(self, operator: int, X_pos=0, Z_pos=0):
        return apply_logical(self._code, self.qubit_matrix, operator, X_pos, Z_pos)

    def apply_stabilizer(self, row: int, col: int, operator: int):
        return apply_stabilizer(self._code, self.qubit_matrix, row, col, operator)

    def apply_random_logical(self):
        import random as rand
        size = self.system_size
        op = int(rand.random() * 4)
        x_pos = int(rand.random() * size) if op in (1, 2) else 0
        z_pos = int(rand.random() * size) if op in (3, 2) else 0
        return self.apply_logical(op, x_pos, z_pos)

    def apply_random_stabilizer(self):
        # five draws, always (xzzx_model.py:439-452)
        import random as rand
        size = self.system_size
        rows = int((size - 1) * rand.random())
        cols = int((size - 1) * rand.random())
        rows2 = int(((size - 1) / 2) * rand.random())
        cols2 = int(4 * rand.random())
        phalf = (size ** 2 - (size - 1) ** 2 - 1) / (size ** 2 - 1)
        if rand.random() > phalf:
            return self.apply_stabilizer(rows, cols, 1)
        return self.apply_stabilizer(rows2, cols2, 3)

    def define_equivalence_class(self):
        return eq_class(self._code, self.qubit_matrix)

    def syndrome(self):
        self.plaquette_defects = syndrome(self._code, self.qubit_matrix).astype(np.float64)
