"""Stand-ins for the reference's symbolic ``logprob_fn``.

The reference takes any Python callable building an Aesara graph and differentiates it
(aehmc/hmc.py:33-34, integrators.py:64-65).  Arbitrary callables cannot be compiled to
HIP, so the engine takes Target objects naming a device function (``kind``) plus its
parameter buffers.  ``potential = -logprob``.
"""
from __future__ import annotations

import numpy as np

# must match enum aehmc_target_kind in include/aehmc_hip.h
T_STD_NORMAL, T_ISO_GAUSSIAN, T_DIAG_GAUSSIAN, T_DENSE_MVN, T_LINREG, T_CUSTOM, T_GLM = range(7)


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


class CustomGLM(Target):
    """A user-defined row-reduction ("GLM-type") ``logprob_fn`` over a data matrix ``X`` [N, D] and responses ``y`` [N]:
    U(q) = sum_n loss(x_n . q, y_n) + sum_i prior(q_i).  ``source`` is HIP source defining

        __device__ void aehmc_glm_row(double z, double y, long long n, const double *const *prm, double &loss, double &dloss_dz)
        __device__ void aehmc_glm_prior(double q, long long i, const double *const *prm, double &u, double &g)

    The two products with X per leapfrog run as chain-batched fp64 MFMA GEMMs, the user's functions in kernels compiled
    with hipRTC on first use (lock-step engine, any metric).  Logistic regression with a N(0, tau^2) prior:

        CustomGLM('''
          __device__ void aehmc_glm_row(double z, double y, long long n, const double *const *prm, double &l, double &d) {
            l = (z > 0 ? z + log1p(exp(-z)) : log1p(exp(z))) - y * z;
            d = 1.0 / (1.0 + exp(-z)) - y;
          }
          __device__ void aehmc_glm_prior(double q, long long i, const double *const *prm, double &u, double &g) {
            const double tau = prm[0][0];
            u = 0.5 * q * q / (tau * tau);
            g = q / (tau * tau);
          }''', X, y, params=[[2.0]])"""

    kind = T_GLM

    def __init__(self, source: str, X, y, params=()):
        self.source, self.X, self.y, self.param_list = str(source), X, y, list(params)
        self.dim = int(X.shape[1])

    def params(self):
        d = {f"p{k}": v for k, v in enumerate(self.param_list)}
        d.update(X=self.X, y=self.y)
        return d
