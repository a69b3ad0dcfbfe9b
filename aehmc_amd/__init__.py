"""aehmc_amd -- MI355X-native many-chain HMC/NUTS trajectory engine.

Keeps aehmc's ``hmc.new_kernel`` / ``nuts.new_kernel`` / ``new_state`` call shapes
(reference: aehmc/hmc.py:16,43,77 and aehmc/nuts.py:14,17,56) over hand-written gfx950
HIP kernels reached through the C-ABI of ``include/aehmc_hip.h``.  There is no CPU
fallback: importing works anywhere, but every computation needs ``libaehmc_hip.so`` and
a GPU and fails loudly otherwise.
"""
from . import algorithms, hmc, mass_matrix, nuts, step_size, targets, tracing, utils, window_adaptation  # noqa: F401
from .engine import PerChain  # noqa: F401
from .integrators import IntegratorState  # noqa: F401
from .random import RandomStream  # noqa: F401
from .trajectory import Diagnostics  # noqa: F401

__all__ = ["algorithms", "hmc", "mass_matrix", "nuts", "step_size", "targets", "tracing", "utils", "window_adaptation", "PerChain", "IntegratorState", "RandomStream", "Diagnostics"]
