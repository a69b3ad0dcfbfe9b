"""The reference's ``logprob_fn``.

The reference takes any Python callable building an Aesara graph and differentiates it
(aehmc/hmc.py:33-34, integrators.py:64-65).  Here a ``logprob_fn`` is either a Target object
naming a device function (``kind``) plus its parameter buffers, or -- as in the reference --
a Python function of the position: ``from_callable`` traces it once (aehmc_amd/tracing.py)
and emits the HIP source of a ``Custom`` / ``CustomJoint`` target, which the engine compiles
with hipRTC and differentiates.  ``potential = -logprob``.
"""
from __future__ import annotations

import numpy as np

# must match enum aehmc_target_kind in include/aehmc_hip.h
T_STD_NORMAL, T_ISO_GAUSSIAN, T_DIAG_GAUSSIAN, T_DENSE_MVN, T_LINREG, T_CUSTOM, T_GLM, T_JOINT = range(8)


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


_DUAL = '#include "dual.cuh"\n'
# the engine's entry points over a density written ONCE as a template: instantiated with aehmc::Dual (csrc/dual.cuh)
_ELEM_FROM_LOGP = """
__device__ void aehmc_custom_elem(double q, long long i, const double *const *prm, double &u, double &g) {
  const aehmc::Dual r = aehmc_logp(aehmc::Dual(q, 1.0), i, prm);
  u = -r.v;
  g = -r.d;
}
"""
_GLM_FROM_LOGP = """
__device__ void aehmc_glm_row(double z, double y, long long n, const double *const *prm, double &loss, double &dloss) {
  const aehmc::Dual r = aehmc_glm_loglik(aehmc::Dual(z, 1.0), y, n, prm);
  loss = -r.v;
  dloss = -r.d;
}
__device__ void aehmc_glm_prior(double q, long long i, const double *const *prm, double &u, double &g) {
  const aehmc::Dual r = aehmc_glm_logprior(aehmc::Dual(q, 1.0), i, prm);
  u = -r.v;
  g = -r.d;
}
"""


class Custom(Target):
    """A user-defined coordinate-wise ``logprob_fn`` (reference: aehmc/hmc.py:16-40 takes any callable and
    differentiates it, hmc.py:33-34, integrators.py:61-65).  ``source`` is HIP source in ONE of two forms:

    * the log-density only -- the engine differentiates it (forward mode, ``csrc/dual.cuh``): a function template over
      its arithmetic type,

          template <class T> __device__ T aehmc_logp(T q, long long i, const double *const *prm)

      returning coordinate i's term of logprob(q) = sum_i logp_i(q_i); ``prm[k]`` is the k-th array of ``params``
      (device float64 arrays, e.g. one value per coordinate).  Independent Student-t coordinates, nu_i = prm[0][i]:

          Custom('''template <class T> __device__ T aehmc_logp(T q, long long i, const double *const *prm) {
                       const double nu = prm[0][i];
                       return -0.5 * (nu + 1.0) * log1p(q * q / nu);
                     }''', params=[nu])

    * potential and gradient by hand,

          __device__ void aehmc_custom_elem(double q, long long i, const double *const *prm, double &u, double &g)

      -- ``u`` = coordinate i's contribution to U = -logprob(q), ``g`` = du_i/dq_i.  A hand-written gradient is checked
      against central differences of ``u`` at the first position it is evaluated at (``new_state``): a wrong one is an
      error, not a wrong posterior.

    The engine compiles its kernel templates against the source with hipRTC on first use (a few seconds; cached per
    source): the lock-step engine for any metric and dimension; for diagonal / scalar metrics the register-resident NUTS
    kernel (D <= 512), the fused HMC kernel (D <= 1024) and the workgroup-per-chain NUTS / HMC kernels (D <= 10176 /
    10240); for a shared dense metric the block-resident NUTS / HMC kernels (64 < D <= 512)."""

    kind = T_CUSTOM

    def __init__(self, source: str, params=(), dim=None):
        self.user_source = str(source)
        self.hand_gradient = "aehmc_custom_elem" in self.user_source
        if not self.hand_gradient and "aehmc_logp" not in self.user_source:
            raise ValueError("Custom: the source must define aehmc_logp (log-density, differentiated by the engine) or "
                             "aehmc_custom_elem (potential and gradient by hand)")
        self.source = self.user_source if self.hand_gradient else _DUAL + self.user_source + _ELEM_FROM_LOGP
        self.param_list, self.dim = list(params), dim
        self.gradient_checked = not self.hand_gradient

    def params(self):
        return {f"p{k}": v for k, v in enumerate(self.param_list)}


class CustomJoint(Target):
    """A user-defined JOINT (non-separable) ``logprob_fn`` of up to 2048 coordinates -- hierarchical models, funnels:
    what the reference samples through aeppl's ``joint_logprob`` (tests/test_hmc.py:170-264).  ``source`` is HIP source
    defining the log-DENSITY only,

        template <class V> __device__ auto aehmc_logp(const V &q, const double *const *prm)

    with ``q[i]`` the coordinates and ``q.size()`` their number; the engine differentiates it in forward mode
    (``csrc/dual.cuh``): lane i of the chain's wavefront evaluates the density with the derivative seeded at coordinate
    i, so one evaluation per leapfrog gives U = -logp and the whole gradient.  Neal's funnel (v = q[0], x = q[1:]):

        CustomJoint('''template <class V> __device__ auto aehmc_logp(const V &q, const double *const *prm) {
                            auto v = q[0];
                            auto lp = -v * v / 18.0;
                            for (int i = 1; i < q.size(); i++) lp += -0.5 * q[i] * q[i] * exp(-v) - 0.5 * v;
                            return lp;
                          }''', dim=10)

    Up to 64 coordinates NUTS and HMC run in single launches (any number of transitions) with a scalar, diagonal or
    dense metric, shared or per chain, and under ``window_adaptation``.  Above 64 (round 5) the target runs on the
    lock-step path: the chain's row waits in LDS and the wavefront evaluates the density ceil(dim / 64) times per
    gradient, lane l seeding coordinate l + 64 k in pass k -- O(dim^2 / 64) density terms per leapfrog and chain.
    A density traced from a Python function (``from_callable``) brings its reverse-mode program (``grad_source``):
    one sweep per gradient whatever the dimension, its loops spread over the wavefront's lanes."""

    kind = T_JOINT

    def __init__(self, source: str, dim: int, params=(), grad_source=None):
        if "aehmc_logp" not in source:
            raise ValueError("CustomJoint: the source must define aehmc_logp(const V &q, const double *const *prm)")
        if not 1 <= int(dim) <= 2048:
            raise ValueError("CustomJoint: 1 <= dim <= 2048")
        self.user_source = str(source)
        # grad_source: the density's reverse-mode program (aehmc_logp_grad + AEHMC_JOINT_GRAD), emitted by
        # aehmc_amd/tracing.py for a traced Python function: above 64 coordinates ONE sweep per gradient
        self.source = _DUAL + self.user_source + (grad_source or "")
        self.param_list, self.dim = list(params), int(dim)
        self.hand_gradient, self.gradient_checked = False, True

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
          }''', X, y, params=[[2.0]])

    The density-only form (the engine differentiates: ``csrc/dual.cuh``) defines instead

        template <class T> __device__ T aehmc_glm_loglik(T z, double y, long long n, const double *const *prm)
        template <class T> __device__ T aehmc_glm_logprior(T q, long long i, const double *const *prm)

    e.g. ``return y * z - softplus(z);`` and ``return -0.5 * q * q / (tau * tau);`` for the model above.  A hand-written
    pair of gradients is checked against central differences at the first ``new_state``."""

    kind = T_GLM

    def __init__(self, source: str, X, y, params=()):
        self.user_source = str(source)
        self.hand_gradient = "aehmc_glm_row" in self.user_source
        if not self.hand_gradient and not ("aehmc_glm_loglik" in self.user_source and "aehmc_glm_logprior" in self.user_source):
            raise ValueError("CustomGLM: the source must define aehmc_glm_loglik and aehmc_glm_logprior (log-densities, "
                             "differentiated by the engine) or aehmc_glm_row and aehmc_glm_prior (losses and gradients by hand)")
        self.source = self.user_source if self.hand_gradient else _DUAL + self.user_source + _GLM_FROM_LOGP
        self.X, self.y, self.param_list = X, y, list(params)
        self.gradient_checked = not self.hand_gradient
        self.dim = int(X.shape[1])

    def params(self):
        d = {f"p{k}": v for k, v in enumerate(self.param_list)}
        d.update(X=self.X, y=self.y)
        return d


def from_callable(fn, dim, scalar=False, args=(), reverse="auto"):
    """A Python ``logprob_fn`` (reference: README.md:27-36, aehmc/hmc.py:16-40 -- any function of the position) as a
    Target: ``fn`` is called ONCE on a proxy of one chain's position (a scalar proxy if ``scalar``, else a vector of
    ``dim`` entries; see ``aehmc_amd.tracing`` for what it may do with it) and the recorded expression is emitted as the
    ``aehmc_logp`` template -- a sum of per-coordinate terms becomes a ``Custom`` target (every kernel family),
    anything else a ``CustomJoint`` target (dim <= 2048).  numpy arrays the function closes over are captured as the
    target's parameter arrays AT TRACE TIME (trace again after changing them).  An operation that cannot be traced
    raises ``TypeError`` here, not in the compiler.

        logprob_fn = lambda y: -0.5 * y**2 - 0.5 * np.log(2 * np.pi)             # README.md:27-36, scalar position
        target = targets.from_callable(logprob_fn, 1, scalar=True)

    ``hmc.new_state`` / ``new_kernel`` and ``nuts.new_state`` / ``new_kernel`` call this themselves when they are handed a
    function instead of a Target (``as_target``).  ``reverse``: how a joint density is differentiated -- "auto" (forward
    mode up to 64 coordinates unless its reductions are long, one reverse sweep above), True (reverse at every size),
    False (forward mode only)."""
    from . import tracing
    tr = tracing.trace(fn, dim, scalar=scalar, args=args, reverse=reverse)
    try:  # the captured arrays are fixed from here on: on the device once (a host array would be hashed and uploaded per call)
        import torch
        if torch.cuda.is_available():
            tr.params = [torch.as_tensor(p, dtype=torch.float64, device="cuda") for p in tr.params]
    except ImportError:
        pass
    tgt = (Custom(tr.source, params=tr.params, dim=tr.dim) if tr.elementwise
           else CustomJoint(tr.source, tr.dim, params=tr.params, grad_source=tr.grad_source))
    tgt.traced_from = fn
    return tgt


_TRACED = {}  # (id(fn), dim, scalar) -> (fn, Target): one trace (and one compilation) per function and shape


def as_target(logprob_fn, dim, scalar=None):
    """``logprob_fn`` itself if it is a Target, else the Target traced from the Python function (cached per function).
    ``scalar``: the position of a chain is a scalar (the function is called on a scalar proxy); None: as it was traced
    before for this dimension, else a vector."""
    if isinstance(logprob_fn, Target):
        return logprob_fn
    if not callable(logprob_fn):
        raise TypeError(f"logprob_fn must be a targets.Target or a Python function of the position, got {type(logprob_fn).__name__}")
    for sc in ((True, False) if scalar is None else (bool(scalar),)):
        hit = _TRACED.get((id(logprob_fn), int(dim), sc))
        if hit is not None and hit[0] is logprob_fn:
            return hit[1]
    sc = bool(scalar)
    tgt = from_callable(logprob_fn, dim, scalar=sc)
    _TRACED[(id(logprob_fn), int(dim), sc)] = (logprob_fn, tgt)
    return tgt
