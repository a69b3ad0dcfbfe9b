"""Value types of the integrator (reference: aehmc/integrators.py:7-11)."""
from typing import Any, NamedTuple, Optional


class IntegratorState(NamedTuple):
    """Same fields as aehmc.integrators.IntegratorState; values are eager device arrays
    with a leading chain axis instead of symbolic TensorVariables."""

    position: Any
    momentum: Optional[Any]
    potential_energy: Any
    potential_energy_grad: Any
