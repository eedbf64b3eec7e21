"""Host-side mirror of the reference's `RotSurCode` (src/rotated_surface_model.py:8-106) over the C-ABI."""
import random as rand

from . import _lib as L_
from ._surf import PlaquetteCode


class RotSurCode(PlaquetteCode):
    _code = L_.ROTATED

    def generate_random_error(self, p_x, p_y, p_z):
        # rotated_surface_model.py:25-38 uses independent `if`s (not elif): same outcome for r off the boundaries
        size = self.system_size
        for i in range(size):
            for j in range(size):
                r = rand.random()
                q = 0
                if r < p_z:
                    q = 3
                if p_z < r < (p_z + p_x):
                    q = 1
                if (p_z + p_x) < r < (p_z + p_x + p_y):
                    q = 2
                self.qubit_matrix[i, j] = q
        self.syndrome()

    def generate_zbiased_error(self, p_error, eta):
        p_z = p_error * eta / (eta + 1)
        p_x = p_error / (2 * (eta + 1))
        PlaquetteCode.generate_random_error(self, p_x, p_x, p_z)
