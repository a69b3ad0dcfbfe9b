"""HMC kernel -- thin wrapper over the HIP engine (reference: aehmc/hmc.py)."""
from __future__ import annotations

from typing import Callable, Dict, Tuple

import torch

from ._common import Layout, diagnostics, histories, new_state as _new_state, state_rows
from .engine import get_engine, rng_to_device
from .integrators import IntegratorState
from .random import RandomStream
from .trajectory import Diagnostics

new_state = _new_state


def new_kernel(srng: RandomStream, logprob_fn, divergence_threshold: int = 1000) -> Callable:
    """Build a HMC kernel (reference: aehmc/hmc.py:43-126).

    Same arguments as the reference; ``logprob_fn`` is a ``targets.Target`` or, as in the reference, a Python function of the position (traced once: ``targets.from_callable``).  The two RNG
    call sites of the reference graph (momentum hmc.py:122, accept hmc.py:194) are taken
    from ``srng`` here, in that order."""
    rng_host = srng.sites(2)
    holder = {}
    if torch.cuda.is_available():  # the generator states go to the device with the kernel, not with its first call
        holder["rng"] = rng_to_device(rng_host, get_engine().device)

    def step(state: IntegratorState, step_size, inverse_mass_matrix,
             num_integration_steps: int) -> Tuple[Diagnostics, Dict]:
        """One HMC transition for every chain (reference: aehmc/hmc.py:77-124)."""
        eng = get_engine()
        shape = tuple(state.position.shape)
        layout = Layout(shape, srng.batched, srng.num_chains)
        if "rng" not in holder or holder["rng"].device != eng.device:  # (uploaded at construction when a GPU is there)
            holder["rng"] = holder["rng"].to(eng.device) if "rng" in holder else rng_to_device(rng_host, eng.device)
        q, U, g = state_rows(state, layout, eng.device)
        eng.set_target(logprob_fn, layout.D, scalar=layout.scalar)
        eng.set_metric(inverse_mass_matrix, layout.D)
        out = eng.hmc_step(holder["rng"], eng.set_step_sizes(step_size), int(num_integration_steps),
                           float(divergence_threshold), q, U, g)
        info = diagnostics(layout, q, U, g, out, False)
        return info, {srng: holder["rng"]}

    def sample(state: IntegratorState, step_size, inverse_mass_matrix, num_integration_steps: int,
               num_samples: int, keep_samples: bool = True):
        """``num_samples`` consecutive transitions per chain in one engine call -- the
        user-level ``aesara.scan(kernel, n_steps=N)`` loop of the reference
        (tests/test_hmc.py:138-148).  Returns ``(samples [N, ...], Diagnostics of the last
        transition, acceptance history [N, ...], divergence history [N, ...])``."""
        eng = get_engine()
        layout = Layout(tuple(state.position.shape), srng.batched, srng.num_chains)
        if "rng" not in holder or holder["rng"].device != eng.device:  # (uploaded at construction when a GPU is there)
            holder["rng"] = holder["rng"].to(eng.device) if "rng" in holder else rng_to_device(rng_host, eng.device)
        q, U, g = state_rows(state, layout, eng.device)
        eng.set_target(logprob_fn, layout.D, scalar=layout.scalar)
        eng.set_metric(inverse_mass_matrix, layout.D)
        out = eng.hmc_sample(holder["rng"], eng.set_step_sizes(step_size), int(num_integration_steps),
                             float(divergence_threshold), int(num_samples), q, U, g, keep_samples)
        info = diagnostics(layout, q, U, g, out, False)
        samples, acc_hist, div_hist = histories(layout, out, int(num_samples), keep_samples)
        return samples, info, acc_hist, div_hist

    step.sample = sample
    step.num_chains, step.batched = srng.num_chains, srng.batched
    step._hmc = dict(srng=srng, rng_host=rng_host, holder=holder, logprob_fn=logprob_fn,  # (for tests / resuming /
                     divergence_threshold=float(divergence_threshold))                    #  the one-call warm-up)
    return step
