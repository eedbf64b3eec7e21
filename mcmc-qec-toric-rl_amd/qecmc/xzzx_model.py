"""Host-side mirror of the reference's `xzzx_code` (src/xzzx_model.py:8-58) over the C-ABI."""
from . import _lib as L_
from ._surf import PlaquetteCode


class xzzx_code(PlaquetteCode):
    _code = L_.XZZX

    def generate_known_error(self, p_error):
        self.qubit_matrix[0, 1] = 1
        self.qubit_matrix[1, 1] = 1
        self.syndrome()
