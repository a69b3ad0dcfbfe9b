import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from aehmc_amd import RandomStream, nuts, targets, tracing
GLM = """
template <class T> __device__ T aehmc_glm_loglik(T z, double y, long long n, const double *const *prm) { return y * z - softplus(z); }
template <class T> __device__ T aehmc_glm_logprior(T q, long long i, const double *const *prm) { return -0.5 * q * q / 4.0; }
"""
for N, D, C in ((10000, 40, 1024), (2000, 64, 4096), (10000, 20, 4096)):
    rng = np.random.default_rng(0)
    X = rng.normal(size=(N, D)) / np.sqrt(D); w = rng.normal(size=D)
    y = (rng.random(N) < 1.0 / (1.0 + np.exp(-X @ w))).astype(np.float64)
    def logistic(q):
        z = X @ q
        return (y * z - tracing.softplus(z)).sum() - 0.5 * (q @ q) / 4.0
    for form, mk in (("python", lambda: targets.from_callable(logistic, D)),
                     ("CustomGLM", lambda: targets.CustomGLM(GLM, torch.as_tensor(X, device="cuda"), torch.as_tensor(y, device="cuda")))):
        t0 = time.perf_counter(); tgt = mk()
        q0 = torch.as_tensor(w + 0.1 * rng.standard_normal((C, D)), device="cuda")
        imm = torch.ones(D, dtype=torch.float64, device="cuda")
        eps = 0.3 / np.sqrt(N)
        kernel = nuts.new_kernel(RandomStream(seeds=list(range(C))), tgt, max_num_expansions=6)
        state = nuts.new_state(q0, tgt)
        state = kernel.sample(state, eps, imm, 2, keep_samples=False)[1].state._replace(momentum=None)
        torch.cuda.synchronize(); tc = time.perf_counter() - t0; t0 = time.perf_counter()
        samples, info, acc, div = kernel.sample(state, eps, imm, 4, keep_samples=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        nl = int(info.n_leapfrog.sum())
        print(f"logistic N={N} D={D} C={C} {form:9s}: {dt / 4 * 1e3:.2f} ms/transition, {nl / 4 / C:.1f} leapfrogs/chain, {nl / dt:.3e} leapfrog/s (trace+compile+warm-up {tc:.1f} s)", flush=True)
