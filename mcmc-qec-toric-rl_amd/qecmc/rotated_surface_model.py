"""Host-side mirror of the reference's `RotSurCode` (src/rotated_surface_model.py:8-106) over the C-ABI."""
from . import _lib as L_
from ._surf import PlaquetteCode


class RotSurCode(PlaquetteCode):
    _code = L_.ROTATED

    def generate_zbiased_error(self, p_error, eta):
        # rotated_surface_model.py:40-46
        self.generate_random_error(p_error / (2 * (eta + 1)), p_error / (2 * (eta + 1)), p_error * eta / (eta + 1))

    def generate_known_error(self, p_error, eta):
        # rotated_surface_model.py:79-82: a fixed two-qubit X error (the arguments are ignored there too)
        self.qubit_matrix[2, 2] = 1
        self.qubit_matrix[1, 0] = 1
        self.syndrome()
