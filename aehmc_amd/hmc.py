"""HMC kernel -- thin wrapper over the HIP engine (reference: aehmc/hmc.py)."""
from __future__ import annotations

from typing import Callable, Dict, Tuple

from ._common import Layout, new_state as _new_state, state_rows
from .engine import get_engine, rng_to_device
from .integrators import IntegratorState
from .random import RandomStream
from .trajectory import Diagnostics

new_state = _new_state


def new_kernel(srng: RandomStream, logprob_fn, divergence_threshold: int = 1000) -> Callable:
    """Build a HMC kernel (reference: aehmc/hmc.py:43-126).

    Same arguments as the reference; ``logprob_fn`` is a ``targets.Target``.  The two RNG
    call sites of the reference graph (momentum hmc.py:122, accept hmc.py:194) are taken
    from ``srng`` here, in that order."""
    rng_host = srng.sites(2)
    holder = {}

    def step(state: IntegratorState, step_size, inverse_mass_matrix,
             num_integration_steps: int) -> Tuple[Diagnostics, Dict]:
        """One HMC transition for every chain (reference: aehmc/hmc.py:77-124)."""
        eng = get_engine()
        shape = tuple(state.position.shape)
        layout = Layout(shape, srng.batched, srng.num_chains)
        if "rng" not in holder:
            holder["rng"] = rng_to_device(rng_host, eng.device)
        q, U, g = state_rows(state, layout, eng.device)
        eng.set_target(logprob_fn, layout.D)
        eng.set_metric(inverse_mass_matrix, layout.D)
        out = eng.hmc_step(holder["rng"], float(step_size), int(num_integration_steps),
                           float(divergence_threshold), q, U, g)
        info = Diagnostics(
            state=IntegratorState(position=layout.vec(q), momentum=layout.vec(out["momentum"]),
                                  potential_energy=layout.per_chain(U),
                                  potential_energy_grad=layout.vec(g)),
            acceptance_probability=layout.per_chain(out["acceptance_probability"]),
            num_doublings=None, is_turning=None,
            is_diverging=layout.per_chain(out["is_diverging"].bool()),
            n_leapfrog=layout.per_chain(out["n_leapfrog"]))
        return info, {srng: holder["rng"]}

    return step
