"""Stand-ins for the reference's symbolic ``logprob_fn``.

The reference takes any Python callable building an Aesara graph and differentiates it
(aehmc/hmc.py:33-34, integrators.py:64-65).  Arbitrary callables cannot be compiled to
HIP, so the engine takes Target objects naming a device function (``kind``) plus its
parameter buffers.  ``potential = -logprob``.
"""
from __future__ import annotations

import numpy as np

# must match enum aehmc_target_kind in include/aehmc_hip.h
T_STD_NORMAL, T_ISO_GAUSSIAN, T_DIAG_GAUSSIAN, T_DENSE_MVN, T_LINREG, T_CUSTOM = range(6)


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


class Custom(Target):
    """A user-defined coordinate-wise ``logprob_fn`` (reference: aehmc/hmc.py:16-40 takes any callable and
    differentiates it, integrators.py:61-65).  ``source`` is HIP source defining

        __device__ void aehmc_custom_elem(double q, long long i, const double *const *prm, double &u, double &g)

    -- ``u`` = coordinate i's contribution to the potential energy U = -logprob(q) = sum_i u_i, ``g`` = du_i/dq_i --
    with ``prm[k]`` the k-th array of ``params`` (device float64 arrays, e.g. one value per coordinate).  The engine
    compiles its kernel templates against it with hipRTC on first use (a few seconds; cached per source): the lock-step
    engine for any metric and dimension, the register-resident NUTS kernel (D <= 512) and the fused HMC kernel
    (D <= 1024) for diagonal / scalar metrics.  Example (independent Student-t coordinates, nu_i = prm[0][i]):

        Custom('''__device__ void aehmc_custom_elem(double q, long long i, const double *const *prm, double &u, double &g) {
                     const double nu = prm[0][i];
                     u = 0.5 * (nu + 1.0) * log1p(q * q / nu);
                     g = (nu + 1.0) * q / (nu + q * q);
                   }''', params=[nu])"""

    kind = T_CUSTOM

    def __init__(self, source: str, params=(), dim=None):
        self.source, self.param_list, self.dim = str(source), list(params), dim

    def params(self):
        return {f"p{k}": v for k, v in enumerate(self.param_list)}
