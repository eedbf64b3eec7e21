"""Host-side mirror of the reference's `Planar_code` (src/planar_model.py:9-153) over the C-ABI: the unrotated surface
code on uint8[2, L, L] (layer 1 uses its first L-1 rows / columns), 4 equivalence classes -- the model
`Chain.update_chain_fast` is hard-wired to in the reference (src/mcmc.py:6,152-160)."""
import random as rand

import numpy as np

from . import _lib as L_


def _prep(m):
    a, batched = L_.as_states(m, 3)
    if a.shape[1] != 2 or a.shape[2] != a.shape[3]:
        raise ValueError(f"qubit_matrix must have shape (2, L, L), got {a.shape[1:]}")
    return a, batched, a.shape[2], a.shape[0]


def _vec(v, n):
    return np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=np.int32), (n,)))


def apply_stabilizer(m, row, col, operator):
    a, batched, size, n = _prep(m)
    out = np.empty_like(a)
    dE = np.empty(n, dtype=np.int32)
    r, c, o = _vec(row, n), _vec(col, n), _vec(operator, n)
    L_.check(L_.lib().qecmc_apply_stabilizer(L_.PLANAR, size, n, L_.u8(a), L_.u8(out), L_.i32(r), L_.i32(c), L_.i32(o), L_.i32(dE)))
    return (out, dE) if batched else (out[0], int(dE[0]))


def apply_logical(m, operator, X_pos=0, Z_pos=0):
    a, batched, size, n = _prep(m)
    out = np.empty_like(a)
    dE = np.empty(n, dtype=np.int32)
    o, x, z, lay = _vec(operator, n), _vec(X_pos, n), _vec(Z_pos, n), _vec(0, n)
    L_.check(L_.lib().qecmc_apply_logical(L_.PLANAR, size, n, L_.u8(a), L_.u8(out), L_.i32(o), L_.i32(lay), L_.i32(x), L_.i32(z),
                                          L_.i32(dE)))
    return (out, dE) if batched else (out[0], int(dE[0]))


def count_errors(m):
    a, batched, size, n = _prep(m)
    out = np.empty(n, dtype=np.int64)
    L_.check(L_.lib().qecmc_count_errors(L_.PLANAR, size, n, L_.u8(a), out.ctypes.data_as(L_._i64p)))
    return out if batched else int(out[0])


def eq_class(m):
    a, batched, size, n = _prep(m)
    out = np.empty(n, dtype=np.int32)
    L_.check(L_.lib().qecmc_eq_class(L_.PLANAR, size, n, L_.u8(a), L_.i32(out)))
    return out if batched else int(out[0])


def syndrome(m):
    """(vertex_defects bool[..., L-1, L], plaquette_defects bool[..., L, L-1]), planar_model.py:134-153."""
    a, batched, size, n = _prep(m)
    out = np.empty((n, 2 * size * (size - 1)), dtype=np.uint8)
    L_.check(L_.lib().qecmc_syndrome(L_.PLANAR, size, n, L_.u8(a), L_.u8(out)))
    v = out[:, :size * (size - 1)].reshape(n, size - 1, size).astype(bool)
    q = out[:, size * (size - 1):].reshape(n, size, size - 1).astype(bool)
    return (v, q) if batched else (v[0], q[0])


class Planar_code:
    nbr_eq_classes = 4

    def __init__(self, size):
        if size < 2:
            raise ValueError("size must be >= 2")
        self.system_size = size
        self.qubit_matrix = np.zeros((2, size, size), dtype=np.uint8)
        self.plaquette_defects = np.zeros((size, size - 1), dtype=bool)
        self.vertex_defects = np.zeros((size - 1, size), dtype=bool)

    def generate_random_error(self, p_x, p_y, p_z):
        # planar_model.py:18-40: a Pauli per cell of the 2 x L x L array, then layer 1's unused last row / column cleared
        from ._surf import pauli_field
        size = self.system_size
        self.qubit_matrix = pauli_field((2, size, size), p_x, p_y, p_z)
        self.qubit_matrix[1, -1, :] = 0
        self.qubit_matrix[1, :, -1] = 0
        self.syndrom()

    def generate_general_noise_error(self, p_xyz):
        # planar_model.py:43-59: one NumPy uniform per cell cut at p_x, p_x + p_y, p_x + p_y + p_z (general noise: STDC_general_noise's input)
        size = self.system_size
        p_xyz = np.asarray(p_xyz, dtype=np.float64)
        u = np.random.uniform(0, 1, size=(2, size, size))
        m = np.zeros((2, size, size), dtype=np.uint8)
        m[u < p_xyz.sum()] = 3
        m[u < p_xyz[0:2].sum()] = 2
        m[u < p_xyz[0]] = 1
        m[1, -1, :] = 0
        m[1, :, -1] = 0
        self.qubit_matrix = m
        self.syndrom()

    def _fill_rows(self, p_x, p_y, p_z):
        # planar_model.py:66-77 / :88-99 index the (2, L, L) array with [i, j] for i, j < L: each draw sets a whole ROW of layer i, and
        # i = 2 is out of bounds -- the loops of the one-layer models pasted onto the two-layer one.  Mirrored as written: only
        # size = 2 runs through; any larger lattice raises the reference's IndexError.
        size = self.system_size
        for i in range(size):
            for j in range(size):
                q = 0
                r = rand.random()
                if r < p_z:
                    q = 3
                elif p_z < r < (p_z + p_x):
                    q = 1
                elif (p_z + p_x) < r < (p_z + p_x + p_y):
                    q = 2
                self.qubit_matrix[i, j] = q

    def generate_biased_error(self, p_error, eta):
        p_z = p_error * eta / (eta + 1)                                    # planar_model.py:61-65
        p_x = p_error / (2 * (eta + 1))
        self._fill_rows(p_x, p_x, p_z)

    def generate_alpha_error(self, p_error, alpha):
        from scipy import optimize                                         # planar_model.py:79-86
        p_tilde = p_error / (1 + p_error)
        pz_tilde = optimize.fsolve(lambda x: x + 2 * x ** alpha - p_tilde, 0.5)[0]
        px_tilde = pz_tilde ** alpha
        self._fill_rows(px_tilde * (1 - p_error), px_tilde * (1 - p_error), pz_tilde * (1 - p_error))

    def chain_lengths(self):
        m = self.qubit_matrix
        return int((m == 1).sum()), int((m == 2).sum()), int((m == 3).sum())

    def count_errors(self):
        return count_errors(self.qubit_matrix)

    def count_errors_xyz(self):
        return np.array(self.chain_lengths(), dtype=np.float64)          # planar_model.py:224-229

    def apply_logical(self, operator: int, X_pos=0, Z_pos=0):
        return apply_logical(self.qubit_matrix, operator, X_pos, Z_pos)

    def apply_stabilizer(self, row: int, col: int, operator: int):
        return apply_stabilizer(self.qubit_matrix, row, col, operator)

    def apply_random_logical(self):
        size = self.system_size                                            # planar_model.py:272-288
        op = int(rand.random() * 4)
        x_pos = int(rand.random() * size) if op in (1, 2) else 0
        z_pos = int(rand.random() * size) if op in (3, 2) else 0
        return self.apply_logical(op, x_pos, z_pos)

    def apply_random_stabilizer(self):
        size = self.system_size                                            # planar_model.py:343-352
        short_side = int((size - 1) * rand.random())
        long_side = int(size * rand.random())
        if rand.random() < 0.5:
            return self.apply_stabilizer(short_side, long_side, 1)
        return self.apply_stabilizer(long_side, short_side, 3)

    def apply_stabilizers_uniform(self, p=0.5):
        # planar_model.py:355-376: every legal generator independently with probability p (NumPy's global RNG);
        # axis-0 index 0 means operator 3, index 1 operator 1; no X-type generator in the last row, no Z-type in the last column
        size = self.system_size
        pick = np.random.rand(2, size, size) < p
        pick[1, size - 1, :] = False
        pick[0, :, size - 1] = False
        m = self.qubit_matrix.copy()
        for o, r, c in np.argwhere(pick):
            m, _ = apply_stabilizer(m, int(r), int(c), 3 if o == 0 else 1)
        return m

    def define_equivalence_class(self):
        return eq_class(self.qubit_matrix)

    def syndrom(self):
        self.vertex_defects, self.plaquette_defects = syndrome(self.qubit_matrix)
