#!/usr/bin/env python3
"""NUTS throughput of Neal's funnel written as a PYTHON logprob_fn (traced: aehmc_amd/tracing.py) -- above 64 coordinates its
reverse-mode program, one sweep per gradient -- beside the same density as a hand-written HIP template (forward mode:
ceil(D / 64) passes).  usage: joint_traced_time.py [C]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aehmc_amd import RandomStream, nuts, targets
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
SRC = """
template <class V> __device__ auto aehmc_logp(const V &q, const double *const *prm) {
  auto v = q[0];
  auto lp = -v * v / 18.0;
  for (int i = 1; i < q.size(); i++) lp += -0.5 * q[i] * q[i] * exp(-v) - 0.5 * v;
  return lp;
}
"""


def funnel(q):
    v, x = q[0], q[1:]
    return -v * v / 18.0 + (-0.5 * x * x * np.exp(-v) - 0.5 * v).sum()


for D in (10, 64, 100, 256, 1000):
    for form in ("python", "template"):
        if form == "template" and D > 1000:
            continue
        r = np.random.default_rng(D)
        tgt = targets.from_callable(funnel, D) if form == "python" else targets.CustomJoint(SRC, dim=D)
        q0 = torch.as_tensor(0.3 * r.standard_normal((C, D)), device="cuda")
        imm = torch.ones(D, dtype=torch.float64, device="cuda")
        kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=6)
        state = nuts.new_state(q0, tgt)
        eps = 0.05
        for _ in range(2):
            state = kernel(state, eps, imm)[0].state._replace(momentum=None)
        T = 5
        state = kernel.sample(state, eps, imm, 2, keep_samples=False)[1].state._replace(momentum=None)  # (compiles the sample() program)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        samples, info, acc, div = kernel.sample(state, eps, imm, T, keep_samples=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        nl = int(info.n_leapfrog.sum())
        print(f"funnel D={D} C={C} {form:8s}: {dt / T * 1e3:.2f} ms/transition, {nl / T / C:.1f} leapfrogs/chain, {nl / dt:.3e} leapfrog/s", flush=True)

# ---- the notebook's regression (examples/LinearRegression.ipynb) written as a Python function over N data rows, beside
#      the built-in LinearRegression target (k_nuts_linreg: chains of a workgroup share each pass over the data)
for N, Cr in ((10_000, 4096), (100_000, 1024)):
    rng = np.random.default_rng(0)
    X = rng.normal(0, 1, size=(N,)); y = 3 * X + rng.normal(0, 1)
    h = 0.5 * np.log(2 * np.pi)

    def regression(q):
        w, ls = q[0], q[1]
        n = np.exp(ls)
        r = y - X * w
        return (-0.5 * w * w - h) + (ls - n) + ls + (-0.5 * (r / n) ** 2 - ls - h).sum()

    for form, tgt in (("python", targets.from_callable(regression, 2)), ("builtin", targets.LinearRegression(X, y))):
        q0 = torch.as_tensor(np.array([3.0, 0.0]) + 0.01 * rng.standard_normal((Cr, 2)), device="cuda")
        imm = np.array([1.0 / N, 0.5 / N])
        kernel = nuts.new_kernel(RandomStream(seeds=list(range(Cr))), tgt, max_num_expansions=6)
        state = nuts.new_state(q0, tgt)
        state = kernel.sample(state, 0.5, imm, 3, keep_samples=False)[1].state._replace(momentum=None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        samples, info, acc, div = kernel.sample(state, 0.5, imm, 5, keep_samples=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        nl = int(info.n_leapfrog.sum())
        print(f"regression N={N} C={Cr} {form:8s}: {dt / 5 * 1e3:.2f} ms/transition, {nl / 5 / Cr:.1f} leapfrogs/chain, {nl / dt:.3e} leapfrog/s", flush=True)

# ---- logistic regression: a Python function over a captured data matrix (z = X @ q) beside targets.CustomGLM (HIP source,
#      density only: the one-launch row sweep / the GEMM path)
from aehmc_amd import tracing
for N, D, Cr in ((10_000, 8, 4096), (100_000, 8, 1024)):
    rng = np.random.default_rng(0)
    X = rng.normal(size=(N, D)); w = rng.normal(size=D)
    y = (rng.random(N) < 1.0 / (1.0 + np.exp(-X @ w))).astype(np.float64)

    def logistic(q):
        z = X @ q
        return (y * z - tracing.softplus(z)).sum() - 0.5 * (q @ q) / 4.0

    GLM = """
template <class T> __device__ T aehmc_glm_loglik(T z, double y, long long n, const double *const *prm) { return y * z - softplus(z); }
template <class T> __device__ T aehmc_glm_logprior(T q, long long i, const double *const *prm) { return -0.5 * q * q / 4.0; }
"""
    for form, tgt in (("python", targets.from_callable(logistic, D)),
                      ("CustomGLM", targets.CustomGLM(GLM, torch.as_tensor(X, device="cuda"), torch.as_tensor(y, device="cuda")))):
        q0 = torch.as_tensor(w + 0.1 * rng.standard_normal((Cr, D)), device="cuda")
        imm = torch.ones(D, dtype=torch.float64, device="cuda")
        eps = 0.3 / np.sqrt(N)
        kernel = nuts.new_kernel(RandomStream(seeds=list(range(Cr))), tgt, max_num_expansions=6)
        state = nuts.new_state(q0, tgt)
        state = kernel.sample(state, eps, imm, 3, keep_samples=False)[1].state._replace(momentum=None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        samples, info, acc, div = kernel.sample(state, eps, imm, 6, keep_samples=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        nl = int(info.n_leapfrog.sum())
        print(f"logistic N={N} D={D} C={Cr} {form:9s}: {dt / 6 * 1e3:.2f} ms/transition, {nl / 6 / Cr:.1f} leapfrogs/chain, {nl / dt:.3e} leapfrog/s", flush=True)
