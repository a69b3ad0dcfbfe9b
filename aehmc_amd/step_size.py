"""Dual-averaging step-size adaptation as a stand-alone building block (reference: aehmc/step_size.py:9-100
over aehmc/algorithms.py:17-115).  ``dual_averaging_adaptation()`` returns ``(init, update)`` with the
reference's call shapes; the states are per-chain device arrays and ``update`` is one HIP launch
(``aehmc_dual_averaging_update``: the same arithmetic, instruction for instruction, as the warm-up kernels of
window_adaptation).  The reference's own test wraps an HMC kernel with it (tests/test_step_size.py:13-88):

    init, update = dual_averaging_adaptation()
    da = init(torch.ones(C))                       # shrinkage points mu (the test passes the step size 1.0)
    for _ in range(n):
        info, _ = kernel(state, PerChain(torch.exp(da.iterates)), imm, L)
        da = update(info.acceptance_probability, da)
"""
from __future__ import annotations

from typing import Callable, NamedTuple, Tuple

import torch

from .engine import _dev_f64, get_engine


class DualAveragingState(NamedTuple):  # algorithms.py:9-14
    step: torch.Tensor           # [C] int64
    iterates: torch.Tensor       # [C] x = log step size
    iterates_avg: torch.Tensor   # [C]
    gradient_avg: torch.Tensor   # [C]
    shrinkage_pts: torch.Tensor  # [C] mu


def dual_averaging_adaptation(target_acceptance_rate: float = 0.8, gamma: float = 0.05, t0: int = 10,
                              kappa: float = 0.75) -> Tuple[Callable, Callable]:
    """(init, update) -- reference: aehmc/step_size.py:9-100."""

    def init(mu) -> DualAveragingState:
        """algorithms.py:56-76: step 1, iterate 0 (so the first step size is exp(0) = 1 whatever ``mu``),
        averages 0, shrinkage points ``mu`` (scalar or one per chain)."""
        eng = get_engine()
        mu = _dev_f64(mu, eng.device).reshape(-1).clone()
        z = torch.zeros_like(mu)
        return DualAveragingState(step=torch.ones(mu.numel(), dtype=torch.int64, device=eng.device), iterates=z,
                                  iterates_avg=z.clone(), gradient_avg=z.clone(), shrinkage_pts=mu)

    def update(acceptance_probability, state: DualAveragingState) -> DualAveragingState:
        """step_size.py:75-98 (gradient = target - acceptance probability) + algorithms.py:79-115;
        returns a NEW state (the reference's states are values)."""
        eng = get_engine()
        p = _dev_f64(acceptance_probability, eng.device).reshape(-1)
        if p.numel() != state.step.numel():
            raise ValueError(f"{p.numel()} acceptance probabilities for {state.step.numel()} adaptation states")
        new = DualAveragingState(state.step.clone(), state.iterates.clone(), state.iterates_avg.clone(),
                                 state.gradient_avg.clone(), state.shrinkage_pts)
        eng.dual_averaging_update(target_acceptance_rate, gamma, t0, kappa, p, new.step, new.iterates,
                                  new.iterates_avg, new.gradient_avg, new.shrinkage_pts, None)
        return new

    return init, update
