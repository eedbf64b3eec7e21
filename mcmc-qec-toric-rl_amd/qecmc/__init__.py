"""qecmc -- MI355X-native MCMC equivalence-class sampler (host-side mirror of the
reference's Toric_code / Chain / Ladder / PTEQ API over the libqecmc C-ABI)."""
from ._lib import QecmcError, device_count, lib, dev_flags, use_library
from ._lib import TORIC, XZZX, ROTATED, PLANAR, SCANS
from .toric_model import Toric_code
from .xzzx_model import xzzx_code
from .rotated_surface_model import RotSurCode
from .planar_model import Planar_code
from .mcmc import Chain, Ladder, Chain_xyz
from .mcmc_biased import Chain_biased, Ladder_biased
from .decoders import PTEQ, PTDC, STDC, STRC, PTRC, STDC_general_noise, STDC_general_noise_shortest, STDC_Nall_n_alpha, single_temp, pteq_batch, ptdc_batch, ptdc_distribution, percent_from_counts
from .mcmc_alpha import Chain_alpha, Ladder_alpha
from .decoders_biasednoise import PTEQ_biased, PTEQ_alpha, PTEQ_alpha_with_shortest

__all__ = ["QecmcError", "device_count", "lib", "TORIC", "XZZX", "ROTATED", "PLANAR", "Toric_code", "xzzx_code", "RotSurCode", "Planar_code",
           "Chain", "Ladder", "Chain_xyz", "Chain_biased", "Ladder_biased", "Chain_alpha", "Ladder_alpha", "PTEQ", "PTDC", "STDC", "STRC", "PTRC", "STDC_general_noise", "STDC_general_noise_shortest", "STDC_Nall_n_alpha", "single_temp", "PTEQ_biased", "PTEQ_alpha", "PTEQ_alpha_with_shortest", "ptdc_batch", "ptdc_distribution", "pteq_batch", "percent_from_counts"]
