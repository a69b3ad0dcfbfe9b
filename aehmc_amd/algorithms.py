"""The adaptation algorithms as stand-alone building blocks (reference: aehmc/algorithms.py): ``dual_averaging``
(algorithms.py:17-115) and ``welford_covariance`` (algorithms.py:120-204) with the reference's ``(init, update[,
final])`` call shapes.  States are eager device arrays; every ``update`` / ``final`` is one HIP launch of the kernels
the warm-up itself uses (``aehmc_dual_averaging_update``, ``aehmc_welford_update``, ``aehmc_covariance_final``:
the same arithmetic, instruction for instruction), and returns NEW arrays -- the reference's states are values.

The reference estimates for ONE chain.  ``num_chains=C`` (a keyword the reference does not have) runs C independent
estimators on arrays with a leading chain axis."""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch

from .engine import _dev_f64, get_engine
from .step_size import DualAveragingState  # noqa: F401  (algorithms.py:9-14)


def dual_averaging(gamma: float = 0.05, t0: int = 10, kappa: float = 0.75) -> Tuple[Callable, Callable]:
    """(init, update) -- reference: aehmc/algorithms.py:17-115.  ``update(gradient, state)`` takes the gradient
    itself (``step_size.dual_averaging_adaptation`` passes ``target_acceptance_rate - acceptance_probability``)."""

    def init(mu) -> DualAveragingState:
        """algorithms.py:56-76: step 1, iterates and averages 0, shrinkage points ``mu`` (scalar or one per chain)."""
        eng = get_engine()
        mu = _dev_f64(mu, eng.device).reshape(-1).clone()
        z = torch.zeros_like(mu)
        return DualAveragingState(step=torch.ones(mu.numel(), dtype=torch.int64, device=eng.device), iterates=z,
                                  iterates_avg=z.clone(), gradient_avg=z.clone(), shrinkage_pts=mu)

    def update(gradient, state: DualAveragingState) -> DualAveragingState:
        """algorithms.py:79-115.  (The kernel forms ``target - p``: with target 0 and p = -gradient that IS the
        gradient, bit for bit.)"""
        eng = get_engine()
        g = _dev_f64(gradient, eng.device).reshape(-1)
        if g.numel() != state.step.numel():
            raise ValueError(f"{g.numel()} gradients for {state.step.numel()} dual-averaging states")
        new = DualAveragingState(state.step.clone(), state.iterates.clone(), state.iterates_avg.clone(),
                                 state.gradient_avg.clone(), state.shrinkage_pts)
        eng.dual_averaging_update(0.0, gamma, t0, kappa, -g, new.step, new.iterates, new.iterates_avg,
                                  new.gradient_avg, new.shrinkage_pts, None)
        return new

    return init, update


class _Shapes:
    """User-facing shapes of a Welford state <-> the engine's [C, D] rows."""

    def __init__(self, mean: torch.Tensor, num_chains: Optional[int]):
        self.batched = num_chains is not None
        lead = 1 if self.batched else 0
        if mean.ndim not in (lead, lead + 1):
            raise ValueError(f"mean has shape {tuple(mean.shape)}: expected {'[C] or [C, D]' if self.batched else '() or [D]'}")
        self.n_dims = mean.shape[lead] if mean.ndim == lead + 1 else 0
        self.C = num_chains if self.batched else 1
        if self.batched and mean.shape[0] != num_chains:
            raise ValueError(f"state has {mean.shape[0]} rows, num_chains is {num_chains}")
        self.D = max(self.n_dims, 1)
        self.mean_shape = tuple(mean.shape)
        self.n_shape = (self.C,) if self.batched else ()


def welford_covariance(compute_covariance: bool, num_chains: Optional[int] = None
                       ) -> Tuple[Callable, Callable, Callable]:
    """Welford's online estimator of variance / covariance: (init, update, final) -- reference:
    aehmc/algorithms.py:120-204.  ``compute_covariance``: m2 is [D, D] and grows by
    ``outer(updated_delta, delta)``; otherwise [D].  A scalar problem (``n_dims == 0``) stays scalar either way
    (algorithms.py:151-156,193)."""

    def init(n_dims: int):
        eng = get_engine()
        lead = (num_chains,) if num_chains is not None else ()
        f64 = dict(dtype=torch.float64, device=eng.device)
        mean = torch.zeros(lead + ((n_dims,) if n_dims else ()), **f64)
        m2 = torch.zeros(lead + ((n_dims, n_dims) if (compute_covariance and n_dims) else ((n_dims,) if n_dims else ())),
                         **f64)
        return mean, m2, torch.zeros(lead, dtype=torch.int64, device=eng.device)

    def update(value, mean, m2, sample_size):
        eng = get_engine()
        sh = _Shapes(mean, num_chains)
        full = bool(compute_covariance) and sh.n_dims > 0
        v = _dev_f64(value, eng.device).reshape(sh.C, sh.D).contiguous()
        mean2 = mean.reshape(sh.C, sh.D).clone()
        m22 = m2.reshape((sh.C, sh.D, sh.D) if full else (sh.C, sh.D)).clone()
        n2 = sample_size.reshape(sh.C).to(torch.int64).clone()
        eng.welford_update(v, mean2, m22, n2, full)
        return mean2.reshape(sh.mean_shape), m22.reshape(m2.shape), n2.reshape(sh.n_shape)

    def final(m2, sample_size):
        """algorithms.py:199-202: m2 / (sample_size - 1)."""
        eng = get_engine()
        n = sample_size.reshape(-1).to(torch.int64).contiguous()
        C = n.numel()
        per = m2.numel() // C
        lead = 1 if num_chains is not None else 0
        full = bool(compute_covariance) and m2.ndim == lead + 2
        D = int(round(per ** 0.5)) if full else per
        return eng.covariance_final(m2.contiguous(), n, D, full, False).reshape(m2.shape)

    return init, update, final
