"""Stand-ins for the reference's symbolic ``logprob_fn``.

The reference takes any Python callable building an Aesara graph and differentiates it
(aehmc/hmc.py:33-34, integrators.py:64-65).  Arbitrary callables cannot be compiled to
HIP, so the engine takes Target objects naming a device function (``kind``) plus its
parameter buffers.  ``potential = -logprob``.
"""
from __future__ import annotations

import numpy as np

# must match enum aehmc_target_kind in include/aehmc_hip.h
T_STD_NORMAL, T_ISO_GAUSSIAN, T_DIAG_GAUSSIAN, T_DENSE_MVN, T_LINREG = range(5)


class Target:
    kind = -1
    dim = None  # None: inferred from the position

    def params(self):
        """dict of name -> array-like (float64) device parameters"""
        return {}


class StdNormal(Target):
    """aeppl logprob of independent N(0,1) coordinates (README.md:27-36):
    logp = sum(-0.5*y**2 - log(sqrt(2*pi)))."""

    kind = T_STD_NORMAL


class IsoGaussian(Target):
    """U = 0.5*||q||^2 (tests/test_trajectory.py:150-151)."""

    kind = T_ISO_GAUSSIAN


class DiagGaussian(Target):
    """Independent N(mu_i, sigma_i^2)."""

    kind = T_DIAG_GAUSSIAN

    def __init__(self, mu, sigma):
        self.mu, self.sigma = mu, sigma
        self.dim = int(np.size(mu)) if not hasattr(mu, "numel") else int(mu.numel())

    def params(self):
        return {"mu": self.mu, "sigma": self.sigma}


class DenseMVN(Target):
    """U = 0.5*(q-mu)^T P (q-mu) with a dense symmetric precision P [D,D]."""

    kind = T_DENSE_MVN

    def __init__(self, mu, precision):
        self.mu, self.precision = mu, precision
        self.dim = int(np.size(mu)) if not hasattr(mu, "numel") else int(mu.numel())

    def params(self):
        return {"mu": self.mu, "prec": self.precision}


class LinearRegression(Target):
    """examples/LinearRegression.ipynb:126-166: w~N(0,1), n~Gamma(2,1), y~N(X w, n),
    sampled in q = [w, log n]."""

    kind = T_LINREG
    dim = 2

    def __init__(self, X, y):
        self.X, self.y = X, y

    def params(self):
        return {"X": self.X, "y": self.y}
