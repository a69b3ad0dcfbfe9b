"""Shape plumbing shared by hmc.py / nuts.py: the reference works on one chain
(position of shape () or (D,)); the engine adds a leading chain axis."""
from __future__ import annotations

import numpy as np
import torch

from .engine import _dev_f64, get_engine
from .integrators import IntegratorState
from .trajectory import Diagnostics


class Layout:
    """How user-facing arrays map to the engine's [C, D] rows."""

    def __init__(self, shape, batched: bool, num_chains: int):
        shape = tuple(shape)
        self.user_shape = shape
        if batched:
            if len(shape) == 0 or shape[0] != num_chains:
                raise ValueError(f"position must have leading dimension {num_chains} (one row per "
                                 f"chain of the RandomStream), got shape {shape}")
            self.C = num_chains
            self.D = int(np.prod(shape[1:])) if len(shape) > 1 else 1
            if len(shape) > 2:
                raise ValueError("position must be [C] or [C, D]")
        else:
            if len(shape) > 1:
                raise ValueError("position must be a scalar or a vector (use RandomStream(seeds=...) "
                                 "for many chains)")
            self.C = 1
            self.D = shape[0] if len(shape) == 1 else 1
        self.scalar_chain_shape = shape[:1] if batched else ()

    @property
    def scalar(self):  # a chain's position is a scalar (the reference's size-() random variable)
        return len(self.user_shape) == len(self.scalar_chain_shape)

    def rows(self, x, device):
        return _dev_f64(x, device).reshape(self.C, self.D).clone()

    def vec(self, t):  # [C, D] -> user shape
        return t.reshape(self.user_shape)

    def per_chain(self, t):  # [C] -> () or [C]
        return t.reshape(self.scalar_chain_shape)


def state_rows(state: IntegratorState, layout: Layout, device):
    q = layout.rows(state.position, device)
    U = _dev_f64(state.potential_energy, device).reshape(layout.C).clone()
    g = layout.rows(state.potential_energy_grad, device)
    return q, U, g


def new_state(q, logprob_fn, num_chains=None) -> IntegratorState:
    """Create a new HMC/NUTS state from a position (reference: aehmc/hmc.py:16-40):
    ``potential_energy = -logprob_fn(q)`` and its gradient, ``momentum=None``.

    ``q``: scalar / [D] for one chain, [C] / [C, D] for many (pass ``num_chains=C`` to mark
    a leading chain axis; a 2-D position is always read as [C, D])."""
    eng = get_engine()
    shape = tuple(q.shape) if hasattr(q, "shape") else np.shape(q)
    batched = (num_chains is not None) or len(shape) == 2
    C = shape[0] if batched else 1
    layout = Layout(shape, batched, C)
    rows = layout.rows(q, eng.device)
    from .targets import as_target
    logprob_fn = as_target(logprob_fn, layout.D, layout.scalar)  # (a Python function of the position: traced once)
    eng.set_target(logprob_fn, layout.D)
    if eng.metric_ndim is None or eng.metric_D != layout.D:
        # new_state needs no metric; bind a unit one so that the ctx is complete
        eng.set_metric(torch.ones(layout.D, dtype=torch.float64, device=eng.device), layout.D)
    eng.ensure_workspace(layout.C, 1)
    if not getattr(logprob_fn, "gradient_checked", True):  # a hand-written gradient: verified where it is first evaluated
        eng.check_gradient(rows)
        logprob_fn.gradient_checked = True
    U, g = eng.new_state(rows)
    return IntegratorState(position=layout.vec(rows), momentum=None,
                           potential_energy=layout.per_chain(U),
                           potential_energy_grad=layout.vec(g))


def diagnostics(layout: Layout, q, U, g, out, tree: bool) -> Diagnostics:
    """trajectory.py:379-384 record from the engine's [C, ...] outputs (``tree``: NUTS fields;
    HMC returns None for them as hmc.py:199-204 does)."""
    flags = out["flags"].bool()  # [2, C]: is_turning (HMC: the accept flag), is_diverging
    return Diagnostics(
        state=IntegratorState(position=layout.vec(q), momentum=layout.vec(out["momentum"]),
                              potential_energy=layout.per_chain(U),
                              potential_energy_grad=layout.vec(g)),
        acceptance_probability=layout.per_chain(out["acceptance_probability"]),
        num_doublings=layout.per_chain(out["num_doublings"]) if tree else None,
        is_turning=layout.per_chain(flags[0]) if tree else None,
        is_diverging=layout.per_chain(flags[1]),
        n_leapfrog=layout.per_chain(out["n_leapfrog"]))


def histories(layout: Layout, out, n: int, keep_samples: bool):
    """(samples [N, ...], acceptance history, divergence history) of a sample() call."""
    samples = out["samples"].reshape((n,) + layout.user_shape) if keep_samples else None
    hist_shape = (n,) + layout.scalar_chain_shape
    return (samples, out["acceptance_history"].reshape(hist_shape),
            out["divergence_history"].bool().reshape(hist_shape))
