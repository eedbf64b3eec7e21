"""qecmc -- MI355X-native MCMC equivalence-class sampler (host-side mirror of the
reference's Toric_code / Chain / Ladder / PTEQ API over the libqecmc C-ABI)."""
from ._lib import QecmcError, device_count, lib
from .toric_model import Toric_code
from .mcmc import Chain, Ladder
from .decoders import PTEQ, pteq_batch, percent_from_counts

__all__ = ["QecmcError", "device_count", "lib", "Toric_code", "Chain", "Ladder", "PTEQ", "pteq_batch",
           "percent_from_counts"]
