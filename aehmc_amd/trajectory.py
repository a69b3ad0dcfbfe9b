"""Per-transition output record (reference: aehmc/trajectory.py:379-384)."""
from typing import Any, NamedTuple, Optional

from .integrators import IntegratorState


class Diagnostics(NamedTuple):
    """aehmc.trajectory.Diagnostics plus ``n_leapfrog`` (integrator calls that belong to
    the chain's trajectory; the unit of the throughput metric)."""

    state: IntegratorState
    acceptance_probability: Any
    num_doublings: Optional[Any]
    is_turning: Optional[Any]
    is_diverging: Any
    n_leapfrog: Any = None
