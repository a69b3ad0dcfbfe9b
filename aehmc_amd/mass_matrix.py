"""Mass-matrix adaptation as a stand-alone building block (reference: aehmc/mass_matrix.py:12-120):
``covariance_adaptation(is_mass_matrix_full)`` -> ``(init, update, final)`` over Welford's estimator
(``algorithms.welford_covariance``), ``final`` applying Stan's shrinkage -- the arithmetic of the warm-up kernels
(``aehmc_covariance_final`` with ``shrink``).  ``num_chains=C`` (not in the reference) adapts C chains at once."""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch

from . import algorithms
from .engine import get_engine


def covariance_adaptation(is_mass_matrix_full: bool = False, num_chains: Optional[int] = None
                          ) -> Tuple[Callable, Callable, Callable]:
    wc_init, wc_update, _ = algorithms.welford_covariance(is_mass_matrix_full, num_chains)
    lead = (num_chains,) if num_chains is not None else ()

    def init(n_dims: int):
        """mass_matrix.py:37-61: the identity (1.0 / ones / eye) and a fresh Welford state."""
        eng = get_engine()
        f64 = dict(dtype=torch.float64, device=eng.device)
        if n_dims == 0:
            imm = torch.ones(lead, **f64)
        elif is_mass_matrix_full:
            imm = torch.eye(n_dims, **f64).expand(lead + (n_dims, n_dims)).clone()
        else:
            imm = torch.ones(lead + (n_dims,), **f64)
        return imm, wc_init(n_dims)

    def update(position, wc_state):
        """mass_matrix.py:63-81."""
        return wc_update(position, *wc_state)

    def final(wc_state):
        """mass_matrix.py:83-118: (n / (n + 5)) cov + 1e-3 (5 / (n + 5)) -- on the diagonal only for a full matrix."""
        eng = get_engine()
        _, m2, sample_size = wc_state
        n = sample_size.reshape(-1).to(torch.int64).contiguous()
        C = n.numel()
        per = m2.numel() // C
        full = bool(is_mass_matrix_full) and m2.ndim == len(lead) + 2
        D = int(round(per ** 0.5)) if full else per
        return eng.covariance_final(m2.contiguous(), n, D, full, True).reshape(m2.shape)

    return init, update, final
