"""NUTS kernel -- thin wrapper over the HIP engine (reference: aehmc/nuts.py)."""
from __future__ import annotations

from typing import Callable, Dict, Tuple

import torch

from ._common import Layout, diagnostics, histories, new_state as _new_state, state_rows
from .engine import get_engine, rng_to_device
from .integrators import IntegratorState
from .random import RandomStream
from .trajectory import Diagnostics

new_state = _new_state  # reference: aehmc/nuts.py:14


def new_kernel(srng: RandomStream, logprob_fn, max_num_expansions: int = 10,
               divergence_threshold: int = 1000) -> Callable:
    """Build an iterative NUTS kernel (reference: aehmc/nuts.py:17-155).

    RNG call sites, in the reference's graph-construction order: momentum (nuts.py:113),
    direction (trajectory.py:516), uniform progressive sampling (proposals.py:99), biased
    progressive sampling (proposals.py:131)."""
    rng_host = srng.sites(4)
    holder = {}
    if torch.cuda.is_available():  # the generator states go to the device with the kernel, not with its first call
        holder["rng"] = rng_to_device(rng_host, get_engine().device)

    def step(state: IntegratorState, step_size, inverse_mass_matrix) -> Tuple[Diagnostics, Dict]:
        """One NUTS transition for every chain (reference: aehmc/nuts.py:56-153)."""
        eng = get_engine()
        shape = tuple(state.position.shape)
        layout = Layout(shape, srng.batched, srng.num_chains)
        if "rng" not in holder or holder["rng"].device != eng.device:  # (uploaded at construction when a GPU is there)
            holder["rng"] = holder["rng"].to(eng.device) if "rng" in holder else rng_to_device(rng_host, eng.device)
        q, U, g = state_rows(state, layout, eng.device)
        eng.set_target(logprob_fn, layout.D, scalar=layout.scalar)
        eng.set_metric(inverse_mass_matrix, layout.D)
        out = eng.nuts_step(holder["rng"], eng.set_step_sizes(step_size), int(max_num_expansions),
                            float(divergence_threshold), q, U, g)
        info = diagnostics(layout, q, U, g, out, True)
        return info, {srng: holder["rng"]}

    def sample(state: IntegratorState, step_size, inverse_mass_matrix, num_samples: int,
               keep_samples: bool = True):
        """``num_samples`` consecutive transitions per chain in one engine call (the
        reference's user-level ``aesara.scan(kernel, n_steps=N)``, tests/test_hmc.py:296-324).
        Returns ``(samples [N, ...], Diagnostics of the last transition with the leapfrog
        TOTAL in n_leapfrog, acceptance history, divergence history)``."""
        eng = get_engine()
        layout = Layout(tuple(state.position.shape), srng.batched, srng.num_chains)
        if "rng" not in holder or holder["rng"].device != eng.device:  # (uploaded at construction when a GPU is there)
            holder["rng"] = holder["rng"].to(eng.device) if "rng" in holder else rng_to_device(rng_host, eng.device)
        q, U, g = state_rows(state, layout, eng.device)
        eng.set_target(logprob_fn, layout.D, scalar=layout.scalar)
        eng.set_metric(inverse_mass_matrix, layout.D)
        out = eng.nuts_sample(holder["rng"], eng.set_step_sizes(step_size), int(max_num_expansions),
                              float(divergence_threshold), int(num_samples), q, U, g, keep_samples)
        info = diagnostics(layout, q, U, g, out, True)
        samples, acc_hist, div_hist = histories(layout, out, int(num_samples), keep_samples)
        return samples, info, acc_hist, div_hist

    step.sample = sample
    step.num_chains, step.batched = srng.num_chains, srng.batched
    # what window_adaptation.run needs to drive the warm-up loop inside the engine
    step._nuts = dict(srng=srng, rng_host=rng_host, holder=holder, logprob_fn=logprob_fn,
                      max_num_expansions=int(max_num_expansions), divergence_threshold=float(divergence_threshold))
    return step
